"""CPU tier for the KERNELS of the docid-set pool: msi_ctx.hip + msi_bits.hip + msi_search.hip are compiled with g++
against tests/emu/hip/hip_runtime.h — a CPU emulation of the HIP runtime (fibers for the threads of a workgroup,
wave ballots / shuffles, __syncthreads, atomics) — so the very kernel source hipcc compiles for gfx950 is executed
here: the set algebra, the CboRoaringBitmap decoders, first_k, the cost-level kernels, the order-key kernels and
every launch the ranked keyword search enqueues.

TEST INFRASTRUCTURE ONLY.  The emulated build is loaded into a private handle that replaces meilisearch_amd._lib._LIB
for the duration of this module and is restored afterwards; the product never loads it (meilisearch_amd has no CPU
path: `ma.Context` on the real libmsi.so fails without an MI355X).  The GPU tier runs the same test bodies on the
device."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from meilisearch_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BUILD = os.path.join(ROOT, "tests", "emu", "_build")
SO = os.path.join(BUILD, "libmsi_emu.so")
CSRC = os.path.join(ROOT, "meilisearch_amd", "csrc")
SOURCES = [os.path.join(CSRC, "msi_ctx.hip"), os.path.join(CSRC, "msi_bits.hip"), os.path.join(CSRC, "msi_search.hip"),
           os.path.join(ROOT, "tests", "hostlogic", "mock_device.cpp")]
DEPS = SOURCES + [os.path.join(CSRC, "msi_common.h"), os.path.join(ROOT, "include", "msi.h"),
                  os.path.join(ROOT, "tests", "emu", "hip", "hip_runtime.h")]


class EmulatedLib:
    """Resolves the symbols the emulated build has; anything else (vector store, dictionary kernels) is an error."""

    def __init__(self, path):
        self._L = C.CDLL(path)

    def __getattr__(self, name):
        if name.startswith("_"):
            raise AttributeError(name)
        fn = getattr(self._L, name)
        if name in _lib.PROTOTYPES:
            fn.restype, fn.argtypes = _lib.PROTOTYPES[name]
        setattr(self, name, fn)
        return fn


@pytest.fixture(scope="module")
def emu():
    if not os.path.exists(SO) or any(os.path.getmtime(s) > os.path.getmtime(SO) for s in DEPS):
        os.makedirs(BUILD, exist_ok=True)
        subprocess.check_call(["g++", "-std=c++17", "-O1", "-g", "-fPIC", "-shared", "-x", "c++", "-DMOCK_DICT_ONLY",
                               "-I" + os.path.join(ROOT, "tests", "emu"), "-I" + CSRC, "-I" + os.path.join(ROOT, "include")]
                              + SOURCES + ["-Wl,-Bsymbolic", "-o", SO, "-lpthread"])
    _lib.lib()                      # the product library stays what every other module sees
    saved = _lib._LIB
    L = EmulatedLib(SO)
    from tests.test_search_hostlogic_cpu import LOOKUP_FN
    L.mock_dict_create.restype, L.mock_dict_create.argtypes = C.c_void_p, [C.c_void_p, C.c_void_p, C.c_uint32, LOOKUP_FN]
    L.mock_dict_destroy.restype, L.mock_dict_destroy.argtypes = None, [C.c_void_p]
    L.harness_cls = EmuHarness
    _lib._LIB = L
    try:
        import meilisearch_amd as ma
        L.ctx = ma.Context(0)
        yield L
        L.ctx.close()
    finally:
        import gc
        gc.collect()                # nothing created on the emulated build may be finalised by the product library
        _lib._LIB = saved


def _emu_harness_base():
    from tests.test_search_hostlogic_cpu import MockHarness
    return MockHarness


class EmuHarness(_emu_harness_base()):
    """The host-logic harness with the REAL pool and key arrays (emulated kernels); only the dictionary is a double."""

    def pool_create(self, n_docs, n_slots):
        import meilisearch_amd as ma
        return ma.BitsPool(self.L.ctx, n_docs, n_slots)

    def pool_destroy(self, pool):
        pool.close()

    def keys_create(self, arr):
        import meilisearch_amd as ma
        return ma.DocKeys(self.L.ctx, arr)

    def keys_destroy(self, h):
        h.close()

    def values_create(self, per_doc, n_values):
        import meilisearch_amd as ma
        return ma.DocValues(self.L.ctx, per_doc, n_values)

    def values_destroy(self, h):
        h.close()

    def points_create(self, lat_lng):
        import meilisearch_amd as ma
        return ma.GeoPoints(self.L.ctx, lat_lng)

    def points_destroy(self, h):
        h.close()


# ---- the kernels against numpy: the bodies of the GPU tier ------------------------------------------------------------
@pytest.mark.parametrize("n_docs", [1, 63, 64, 1000, 20003])
def test_set_algebra(emu, n_docs):
    import tests.test_bits_gpu as TB
    TB.test_algebra_vs_numpy(emu.ctx, n_docs)


def test_cbo_decoders(emu):
    import tests.test_bits_gpu as TB
    TB.test_cbo_decode(emu.ctx)


@pytest.mark.parametrize("n_docs", [1, 63, 64, 65, 1000, 5003])
def test_order_keys(emu, n_docs):
    import tests.test_zz_order_keys_gpu as TO
    TO.test_order_next_against_numpy(n_docs)


@pytest.mark.parametrize("kind", ["single", "multi", "chain", "same"])
@pytest.mark.parametrize("n_docs", [1, 63, 64, 65, 1000, 2003])
def test_distinct_kernels(emu, n_docs, kind):
    import tests.test_zzz_distinct_gpu as TD
    TD.test_distinct_against_the_sequential_loop(n_docs, kind)


def test_distinct_scratch_stamps(emu):
    import tests.test_zzz_distinct_gpu as TD
    TD.test_many_calls_share_the_scratch_without_clearing_it()


@pytest.mark.parametrize("ascending", [True, False], ids=["asc", "desc"])
@pytest.mark.parametrize("n_docs", [1, 63, 64, 65, 1000])
def test_geo_kernels(emu, n_docs, ascending):
    import tests.test_zzz_geo_gpu as TG
    TG.test_geo_next_against_the_bucket_rule(n_docs, ascending)


# ---- the ranked keyword search over the emulated kernels ---------------------------------------------------------------
@pytest.mark.parametrize("fused,per_wait", [("1", "1"), ("0", "1"), ("1", "4")],
                         ids=["level-at-once", "path-by-path", "4-levels-per-wait"])
def test_reference_snapshots_over_emulated_kernels(emu, monkeypatch, fused, per_wait):
    import tests.test_search_hostlogic_cpu as H
    H.test_reference_snapshots_through_the_host_logic(emu, monkeypatch, fused, per_wait)


def test_random_corpora_over_emulated_kernels(emu, monkeypatch):
    import tests.test_search_hostlogic_cpu as H
    H.test_host_logic_matches_oracle_on_random_corpora(emu, monkeypatch, "3")


def test_sort_rules_over_emulated_kernels(emu, monkeypatch):
    import tests.test_search_hostlogic_cpu as H
    H.test_sort_rules_match_the_oracle(emu, monkeypatch, "1")


def test_distinct_over_emulated_kernels(emu, monkeypatch):
    import tests.test_search_hostlogic_cpu as H
    H.test_distinct_matches_the_oracle(emu, monkeypatch, "1", fields=("color", "sizes"), setups=H.DISTINCT_SETUPS[1:])


def test_geo_sort_over_emulated_kernels(emu):
    import tests.test_search_hostlogic_cpu as H
    H.test_geo_sort_rs_through_the_host_logic(emu)
    H.test_geo_sort_matches_the_oracle(emu, setups=H.GEO_SETUPS[1:4])


def test_the_product_library_is_back(emu):
    """Nothing of the emulation may leak: after this module other tests see libmsi.so again (checked at teardown by
    the fixture; here: the emulated build really is a different object)."""
    assert _lib._LIB is emu and emu._L._name == SO
