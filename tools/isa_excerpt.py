"""Writes the ISA evidence DESIGN.md cites for the int8 sweep (profiles/r6_i8_sweep_isa.txt): the kernel's register and LDS
figures from the code object's metadata, the residency they allow, and the steady-state loop of vs_scan_i8_kernel<8,8,2,4,false>
with the `s_waitcnt vmcnt` placement relative to the stage's loads and MFMAs (the wait is forced BEFORE the next
stage's loads).  No GPU needed: llvm-objdump / llvm-readelf on meilisearch_amd/csrc/msi_vs.o.

    python tools/isa_excerpt.py > profiles/r6_i8_sweep_isa.txt"""
import os
import re
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = "/opt/rocm/lib/llvm/bin"


def main():
    obj = os.path.join(ROOT, "meilisearch_amd", "csrc", "msi_vs.o")
    tmp = tempfile.mkdtemp(prefix="msi_isa_")
    try:
        local = os.path.join(tmp, "msi_vs.o")
        shutil.copy(obj, local)
        subprocess.run([BIN + "/llvm-objdump", "--offloading", local], capture_output=True, text=True, check=True)
        co = os.path.join(tmp, [f for f in os.listdir(tmp) if "gfx950" in f][0])
        notes = subprocess.run([BIN + "/llvm-readelf", "--notes", co], capture_output=True, text=True, check=True).stdout
        isa = subprocess.run([BIN + "/llvm-objdump", "-d", "--demangle", co], capture_output=True, text=True, check=True).stdout
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    want = "vs_scan_i8_kernel<8, 8, 2, 4, false, false>"
    # metadata of every int8 sweep instantiation
    print("# code object metadata (llvm-readelf --notes), int8 sweep instantiations of meilisearch_amd/csrc/msi_vs.o")
    blocks = re.split(r"\n\s*- \.agpr_count:", notes)
    for b in blocks[1:]:
        name = re.search(r"\.name:\s+(\S+)", b)
        if not name or "vs_scan_i8_kernel" not in name.group(1):
            continue
        dem = subprocess.run(["c++filt", name.group(1)], capture_output=True, text=True).stdout.strip()
        f = {k: re.search(r"\.%s:\s+(\d+)" % k, b) for k in ("vgpr_count", "sgpr_count", "vgpr_spill_count", "sgpr_spill_count",
                                                              "group_segment_fixed_size", "private_segment_fixed_size", "max_flat_workgroup_size")}
        agpr = re.match(r"\s*(\d+)", b)
        vals = {k: (int(v.group(1)) if v else None) for k, v in f.items()}
        vg = vals["vgpr_count"] or 0
        # gfx950: 512 VGPRs (arch + acc, unified) per SIMD lane; a 512-thread workgroup = 2 waves per SIMD
        waves_per_simd = 512 // max(1, ((vg + 7) // 8) * 8)
        short = re.search(r"vs_scan_i8_kernel<[^>]*>", dem)
        print(f"{short.group(0) if short else dem}: vgpr {vg} (agpr {agpr.group(1) if agpr else '?'}), sgpr {vals['sgpr_count']}, vgpr spills {vals['vgpr_spill_count']}, "
              f"scratch {vals['private_segment_fixed_size']} B, static LDS {vals['group_segment_fixed_size']} B (+ dynamic: the query fragments), "
              f"waves per SIMD by registers: {waves_per_simd}")
    funcs = re.split(r"\n(?=[0-9a-f]{16} <)", isa)
    body = next((f for f in funcs if want in f.split("\n", 1)[0]), None)
    if body is None:
        print("kernel not found:", want)
        sys.exit(1)
    lines = body.split("\n")
    n_mfma = sum("v_mfma_i32_16x16x64_i8" in l for l in lines)
    n_load = sum(re.search(r"\bglobal_load_dwordx4\b", l) is not None for l in lines)
    n_wait = [l for l in lines if "s_waitcnt" in l and "vmcnt" in l]
    print(f"\n# {want}: {len(lines)} lines of ISA, {n_mfma} v_mfma_i32_16x16x64_i8, {n_load} global_load_dwordx4, {len(n_wait)} s_waitcnt with a vmcnt")
    print("# every s_waitcnt vmcnt(...) of the kernel with the number of row loads (global_load_dwordx4 ... nt) and MFMAs since the previous one:")
    loads = mfmas = quiet = 0
    for l in lines:
        if re.search(r"\bglobal_load_dwordx4\b", l):
            loads += 1
        if "v_mfma_i32_16x16x64_i8" in l:
            mfmas += 1
        if "s_waitcnt" in l and "vmcnt" in l:
            ins = l.split("//")[0].strip()
            if loads or mfmas:
                if quiet:
                    print(f"  ({quiet} more waits with no row load and no MFMA in between: the epilogue's conditional survivor stores)")
                    quiet = 0
                print(f"  after {loads:3d} loads, {mfmas:3d} MFMAs: {ins}")
            else:
                quiet += 1
            loads = mfmas = 0
    if quiet:
        print(f"  ({quiet} more waits with no row load and no MFMA in between)")
    # the steady-state loop: the longest backward branch target region containing MFMAs
    print("\n# the main loop (from the first MFMA to the backward branch), instruction mnemonics with operands trimmed:")
    first = next(i for i, l in enumerate(lines) if "v_mfma_i32_16x16x64_i8" in l)
    last = max(i for i, l in enumerate(lines) if "v_mfma_i32_16x16x64_i8" in l)
    kept = 0
    for l in lines[max(0, first - 40):last + 12]:
        ins = l.split("//")[0].strip()
        if not ins or kept >= 260:
            continue
        if re.search(r"global_load|s_waitcnt|v_mfma|s_cbranch|s_branch|ds_read|ds_load|s_barrier|global_store|global_atomic|buffer_", ins):
            print("  " + ins[:110])
            kept += 1
    print(f"# ({kept} memory / matrix / wait / branch instructions shown; VALU and SALU lines omitted)")


if __name__ == "__main__":
    main()
