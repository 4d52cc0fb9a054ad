"""TEST INFRASTRUCTURE — runs pytest with libmsi replaced by its CPU-emulated build.

    python tests/emu/run_emulated.py <pytest arguments>

Builds tests/emu/_build/libmsi_emu.so from the product's own sources (every meilisearch_amd/csrc/*.hip, compiled as
plain C++ by the ROCm clang against tests/emu/hip/hip_runtime.h: fibers for the threads of a workgroup, wave
collectives, MFMA, atomics, LDS, the stream / memcpy API), makes it what meilisearch_amd._lib.lib() returns IN THIS
PROCESS ONLY, and hands over to pytest.  tests/test_kernels_emulated_cpu.py starts it as a subprocess so that the GPU
test files run unchanged — same bodies, fixtures and parameters — in the CPU tier; the product never loads this
build (meilisearch_amd has no CPU path: on the real libmsi.so `ma.Context` fails without an MI355X).
It checks kernel LOGIC (indexing, fragment layouts, ballots, completion protocol, stale-slot handling); speed,
inter-workgroup memory ordering and code generation are the GPU tier's."""
import ctypes as C
import glob
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CSRC = os.path.join(ROOT, "meilisearch_amd", "csrc")
# MSI_EMU_OPT (e.g. "-O2"): a second build in its own directory, for HOST CPU profiles of the search threads (tools/kw_leg.py
# --emulated); the test tier always runs the -O1 build
OPT = os.environ.get("MSI_EMU_OPT", "-O1")
BUILD = os.path.join(ROOT, "tests", "emu", "_build" if OPT == "-O1" else "_build" + OPT.replace("-", "_"))
SO = os.path.join(BUILD, "libmsi_emu.so")
CLANG = "/opt/rocm/lib/llvm/bin/clang++"   # plain C++ mode: ext_vector_type and __bf16 as the kernels spell them


def build():
    # (msi_group.hip included: MSI_EMU_DEVICES emulated devices, RCCL replaced by tests/emu/rccl_emu.cpp — build_rccl())
    sources = sorted(glob.glob(os.path.join(CSRC, "*.hip")))
    deps = sources + sorted(glob.glob(os.path.join(CSRC, "*.h"))) + [os.path.join(ROOT, "include", "msi.h"),
                                                                     os.path.join(ROOT, "tests", "emu", "hip", "hip_runtime.h")]
    if os.path.exists(SO) and all(os.path.getmtime(d) <= os.path.getmtime(SO) for d in deps):
        return SO
    os.makedirs(BUILD, exist_ok=True)
    tmp = SO + f".{os.getpid()}.tmp"
    subprocess.check_call([CLANG, "-std=c++17", OPT, "-g", "-fPIC", "-shared", "-ffp-contract=off", "-x", "c++",
                           "-I" + os.path.join(ROOT, "tests", "emu"), "-I" + CSRC, "-I" + os.path.join(ROOT, "include")]
                          + sources + ["-Wl,-Bsymbolic", "-o", tmp, "-lpthread", "-ldl"])
    os.replace(tmp, SO)
    return SO


RCCL_SO = os.path.join(BUILD, "librccl_emu.so")


def build_rccl():
    """The stand-in for librccl.so (tests/emu/rccl_emu.cpp): collectives between emulated devices are memcpy."""
    src = os.path.join(ROOT, "tests", "emu", "rccl_emu.cpp")
    if os.path.exists(RCCL_SO) and os.path.getmtime(src) <= os.path.getmtime(RCCL_SO):
        return RCCL_SO
    os.makedirs(BUILD, exist_ok=True)
    tmp = RCCL_SO + f".{os.getpid()}.tmp"
    subprocess.check_call([CLANG, "-std=c++17", OPT, "-g", "-fPIC", "-shared", src, "-o", tmp, "-lpthread"])
    os.replace(tmp, RCCL_SO)
    return RCCL_SO


RUNNER_SO = os.path.join(BUILD, "libmsi_rankedbench_emu.so")


def build_runner():
    """tools/ranked_bench.cpp (the synthetic inverted index + caller threads of bench.py's keyword leg) linked against the
    emulated build instead of libmsi.so: tests/test_configs_gpu.py::test_c4_keyword_leg runs through it at a reduced size
    (oracle/synth_index.py loads $MSI_RUNNER_SO when it is set)."""
    src = os.path.join(ROOT, "tools", "ranked_bench.cpp")
    if os.path.exists(RUNNER_SO) and all(os.path.getmtime(d) <= os.path.getmtime(RUNNER_SO) for d in (src, SO)):
        return RUNNER_SO
    tmp = RUNNER_SO + f".{os.getpid()}.tmp"
    subprocess.check_call([CLANG, "-std=c++17", OPT, "-g", "-fPIC", "-shared", "-DRANKED_BENCH_LIB", "-x", "c++",
                           "-I" + os.path.join(ROOT, "include"), src, "-L" + BUILD, "-lmsi_emu", "-Wl,-rpath," + BUILD,
                           "-lpthread", "-o", tmp])
    os.replace(tmp, RUNNER_SO)
    return RUNNER_SO


class EmulatedLib:
    def __init__(self, path):
        self._L = C.CDLL(path)

    def __getattr__(self, name):
        if name.startswith("_"):
            raise AttributeError(name)
        from meilisearch_amd import _lib
        fn = getattr(self._L, name)
        if name in _lib.PROTOTYPES:
            fn.restype, fn.argtypes = _lib.PROTOTYPES[name]
        setattr(self, name, fn)
        return fn


def main(argv):
    if ROOT not in sys.path:
        sys.path.insert(0, ROOT)
    from meilisearch_amd import _lib
    _lib._LIB = EmulatedLib(build())
    assert _lib.lib().msi_abi_version() == 3
    os.environ["MSI_RUNNER_SO"] = build_runner()
    os.environ["MSI_RCCL_LIBRARY"] = build_rccl()
    import pytest
    return pytest.main(argv)


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
