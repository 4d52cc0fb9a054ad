"""Code-generation checks on the built gfx950 objects (no GPU needed: llvm-objdump on the device code of
meilisearch_amd/csrc/*.o).  ADVICE r2: the completion protocols of msi_vm.hip / msi_bits.hip order their set-word
stores before the ticket atomic with `s_waitcnt vmcnt(0)` (MSI_ORDER_ATOMICS, msi_common.h) — that instruction must be
IN the ISA, whatever the compiler makes of the fence beside it."""
import os
import re
import shutil
import subprocess
import tempfile

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"


def device_isa(obj):
    tmp = tempfile.mkdtemp(prefix="msi_isa_")
    try:
        local = os.path.join(tmp, os.path.basename(obj))
        shutil.copy(obj, local)
        subprocess.run([OBJDUMP, "--offloading", local], capture_output=True, text=True, check=True)
        co = [f for f in os.listdir(tmp) if "gfx950" in f]
        assert co, "no gfx950 code object in " + obj
        return subprocess.run([OBJDUMP, "-d", os.path.join(tmp, co[0])], capture_output=True, text=True, check=True).stdout
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


@pytest.mark.parametrize("unit,kernels", [("msi_vm", ["vm_kernel"]),
                                          ("msi_bits", ["bits_"])])
def test_ticket_atomics_wait_for_the_stores_before_them(unit, kernels):
    obj = os.path.join(ROOT, "meilisearch_amd", "csrc", unit + ".o")
    if not os.path.exists(obj) or not os.path.exists(OBJDUMP):
        pytest.skip("objects not built / no llvm-objdump")
    isa = device_isa(obj)
    funcs = re.split(r"\n(?=[0-9a-f]{16} <)", isa)      # one piece per function symbol
    checked = 0
    for fn in funcs:
        head = fn.split("\n", 1)[0]
        if not any(k in head for k in kernels):
            continue
        lines = fn.split("\n")
        for i, l in enumerate(lines):
            # a returning (sc0) global atomic add = a ticket / completion counter of the protocol
            if re.search(r"\bglobal_atomic_add(_x2)?\b.*\bsc0\b", l):
                window = lines[max(0, i - 60):i]
                last_wait = max((j for j, w in enumerate(window) if "s_waitcnt" in w and "vmcnt(0)" in w), default=None)
                assert last_wait is not None, (head, l)
                # no store may sit between the last vmcnt(0) wait and the ticket
                between = window[last_wait + 1:]
                assert not any(re.search(r"\b(global|buffer|flat)_store", b) for b in between), (head, between)
                checked += 1
    assert checked >= 1, "no ticket atomic found in " + unit


def kernel_vgprs(obj):
    """{kernel symbol: vgpr_count} from the code object's metadata notes."""
    readelf = os.path.join(os.path.dirname(OBJDUMP), "llvm-readelf")
    tmp = tempfile.mkdtemp(prefix="msi_isa_")
    try:
        local = os.path.join(tmp, os.path.basename(obj))
        shutil.copy(obj, local)
        subprocess.run([OBJDUMP, "--offloading", local], capture_output=True, text=True, check=True)
        co = [f for f in os.listdir(tmp) if "gfx950" in f]
        notes = subprocess.run([readelf, "--notes", os.path.join(tmp, co[0])], capture_output=True, text=True, check=True).stdout
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    out, name = {}, None
    for line in notes.split("\n"):
        m = re.search(r"\.name:\s+(\S+)", line)
        if m:
            name = m.group(1)
        m = re.search(r"\.vgpr_count:\s+(\d+)", line)
        if m and name:
            out[name] = int(m.group(1))
    return out


def test_occupancy_the_host_code_counts_on():
    """msi_dict.hip launches `6 x CUs` workgroups of the bit-parallel matcher kernel and `4 x CUs` of the banded one (4 waves
    each: 6 / 4 waves per SIMD) — the kernels are bound by how many range-scanning waves a CU holds — and the command-list
    interpreter is built for 4 waves per SIMD (__launch_bounds__(256, 4)).  The register counts behind those numbers, on the built objects."""
    csrc = os.path.join(ROOT, "meilisearch_amd", "csrc")
    if not os.path.exists(os.path.join(csrc, "msi_dict.o")) or not os.path.exists(OBJDUMP):
        pytest.skip("objects not built / no llvm-objdump")
    d = kernel_vgprs(os.path.join(csrc, "msi_dict.o"))
    bits = [v for k, v in d.items() if "dict_lookup_kernelILb1" in k]
    banded = [v for k, v in d.items() if "dict_lookup_kernelILb0" in k]
    assert bits and banded, d
    assert bits[0] <= 80 and banded[0] <= 128, d           # 512 / 80 = 6, 512 / 128 = 4 waves per SIMD
    v = kernel_vgprs(os.path.join(csrc, "msi_vm.o"))
    vm = [x for k, x in v.items() if "vm_kernel" in k]
    assert vm and vm[0] <= 128, v                          # 4 waves per SIMD = four 4-wave workgroups per CU (LDS: 36.8 KB each)


def test_the_int8_sweep_as_it_was_compiled():
    """Round 5's dominant kernel on the built object: the default shapes of vs_scan_i8_kernel (8 waves; 8 / 6 / 4 query tiles)
    keep their accumulators and row buffers in registers (a 512-thread workgroup leaves 256 per wave: no scratch in the sparse
    main pass), the matrix work is v_mfma_i32_16x16x64_i8, and the rows are streamed with non-temporal 16-byte loads."""
    csrc = os.path.join(ROOT, "meilisearch_amd", "csrc")
    obj = os.path.join(csrc, "msi_vs.o")
    if not os.path.exists(obj) or not os.path.exists(OBJDUMP):
        pytest.skip("objects not built / no llvm-objdump")
    readelf = os.path.join(os.path.dirname(OBJDUMP), "llvm-readelf")
    tmp = tempfile.mkdtemp(prefix="msi_isa_")
    try:
        local = os.path.join(tmp, "msi_vs.o")
        shutil.copy(obj, local)
        subprocess.run([OBJDUMP, "--offloading", local], capture_output=True, text=True, check=True)
        co = os.path.join(tmp, [f for f in os.listdir(tmp) if "gfx950" in f][0])
        notes = subprocess.run([readelf, "--notes", co], capture_output=True, text=True, check=True).stdout
        asm = subprocess.run([OBJDUMP, "-d", co], capture_output=True, text=True, check=True).stdout
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    meta, name = {}, None
    for line in notes.split("\n"):
        m = re.search(r"\.name:\s+(\S+)", line)
        if m:
            name = m.group(1)
            meta[name] = {}
        for key in ("vgpr_count", "vgpr_spill_count"):
            m = re.search(r"\." + key + r":\s+(\d+)", line)
            if m and name:
                meta[name][key] = int(m.group(1))
    # vs_scan_i8_kernel<WAVES = 8, NQT, RT, GS, DENSE = false, FILT = false>: the sparse main pass of the shapes the library
    # launches by default (round 6 added FILT — the row-granular filtered sweep — and the unfiltered kernel kept two always-null
    # arguments because without their uniform branches the same source spilled 22 registers: Scan8Args::list)
    main = {k: v for k, v in meta.items() if "vs_scan_i8_kernelILi8E" in k and k.endswith("Lb0ELb0EEEvNS_9Scan8ArgsE")}
    for nqt, rt in ((8, 2), (6, 2), (4, 3)):
        hit = {k: v for k, v in main.items() if f"ILi8ELi{nqt}ELi{rt}ELi" in k}
        assert len(hit) == 3, (nqt, rt, sorted(main))                         # one per pipeline depth GS = 4 / 3 / 2
        for k, v in hit.items():
            assert v["vgpr_count"] <= 256 and v["vgpr_spill_count"] == 0, (k, v)
    body = asm.split("<_ZN12_GLOBAL__N_117vs_scan_i8_kernelILi8ELi8ELi2ELi4ELb0ELb0EEEvNS_9Scan8ArgsE>:")[1].split("s_endpgm")[0]
    assert body.count("v_mfma_i32_16x16x64_i8") == 8 * 2 * 4 * 2              # NQT x RT x GS, for each of the two row buffers
    loads = [ln for ln in body.split("\n") if "global_load_dwordx4" in ln and " nt" in ln]
    assert len(loads) >= 2 * 4 * 2                                            # RT x GS per buffer, non-temporal
