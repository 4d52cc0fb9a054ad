"""ctypes loader for oracle/libmsi_oracle.so — TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this module.  The product package (meilisearch_amd/) never does.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE])


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "libmsi_oracle.so")
        if not os.path.exists(path):
            build()
        L = C.CDLL(path)
        f32p = C.POINTER(C.c_float)
        u32p = C.POINTER(C.c_uint32)
        u64p = C.POINTER(C.c_uint64)
        u8p = C.POINTER(C.c_uint8)
        f64p = C.POINTER(C.c_double)
        L.orc_dot_f32.restype = C.c_float
        L.orc_dot_f32.argtypes = [f32p, f32p, C.c_uint32]
        L.orc_cosine_distance.restype = C.c_float
        L.orc_cosine_distance.argtypes = [f32p, f32p, C.c_uint32]
        L.orc_similarity.restype = C.c_float
        L.orc_similarity.argtypes = [C.c_float]
        L.orc_vs_topk.restype = None
        L.orc_vs_topk.argtypes = [f32p, u32p, C.c_uint64, C.c_uint32, f32p, C.c_uint32, u64p,
                                  C.c_uint64, u32p, f32p, u32p]
        L.orc_distribution_shift.restype = C.c_float
        L.orc_distribution_shift.argtypes = [C.c_float, C.c_float, C.c_float]
        L.orc_rank_global_score.restype = C.c_double
        L.orc_rank_global_score.argtypes = [u32p, u32p, C.c_uint32]
        L.orc_compare_scores.restype = C.c_int32
        L.orc_compare_scores.argtypes = [f64p, C.c_uint32, C.c_float, f64p, C.c_uint32, C.c_float]
        L.orc_typo_budget.restype = C.c_uint8
        L.orc_typo_budget.argtypes = [u8p, C.c_uint32, C.c_uint32, C.c_uint32]
        L.orc_osa_distance.restype = C.c_uint32
        L.orc_osa_distance.argtypes = [u8p, C.c_uint32, u8p, C.c_uint32, C.c_int]
        L.orc_typo_lookup.restype = None
        L.orc_typo_lookup.argtypes = [u8p, u32p, C.c_uint32, u8p, C.c_uint32, C.c_uint32, C.c_int,
                                      C.c_uint32, C.c_uint32, u32p, u32p, u32p, u32p]
        L.orc_bq_topk.restype = None
        L.orc_bq_topk.argtypes = [f32p, u32p, C.c_uint64, C.c_uint32, f32p, C.c_uint32, u64p, C.c_uint64, u32p, f32p, u32p]
        _LIB = L
    return _LIB


def _p(a, t):
    return a.ctypes.data_as(C.POINTER(t))


def cosine_distance(q, x):
    q = np.ascontiguousarray(q, dtype=np.float32)
    x = np.ascontiguousarray(x, dtype=np.float32)
    return float(np.float32(lib().orc_cosine_distance(_p(q, C.c_float), _p(x, C.c_float), q.size)))


def similarity(distance):
    return float(np.float32(lib().orc_similarity(C.c_float(distance))))


def vs_topk(rows, docids, q, k, filter_bits=None, filter_nbits=0):
    """Exact top-k of one store: returns (docids[u32], dist[f32])."""
    rows = np.ascontiguousarray(rows, dtype=np.float32)
    docids = np.ascontiguousarray(docids, dtype=np.uint32)
    q = np.ascontiguousarray(q, dtype=np.float32)
    n, d = rows.shape if rows.ndim == 2 else (0, q.size)
    out_d = np.zeros(max(k, 1), dtype=np.uint32)
    out_s = np.zeros(max(k, 1), dtype=np.float32)
    cnt = C.c_uint32(0)
    fb = None
    if filter_bits is not None:
        filter_bits = np.ascontiguousarray(filter_bits, dtype=np.uint64)
        fb = _p(filter_bits, C.c_uint64)
    lib().orc_vs_topk(_p(rows, C.c_float), _p(docids, C.c_uint32), n, d, _p(q, C.c_float), k, fb,
                      filter_nbits, _p(out_d, C.c_uint32), _p(out_s, C.c_float), C.byref(cnt))
    return out_d[:cnt.value].copy(), out_s[:cnt.value].copy()


def distribution_shift(mean, sigma, score):
    return float(np.float32(lib().orc_distribution_shift(mean, sigma, score)))


def rank_global_score(pairs):
    r = np.array([p[0] for p in pairs], dtype=np.uint32)
    m = np.array([p[1] for p in pairs], dtype=np.uint32)
    return float(lib().orc_rank_global_score(_p(r, C.c_uint32), _p(m, C.c_uint32), len(pairs)))


def compare_scores(left, lratio, right, rratio):
    l = np.array(left, dtype=np.float64)
    r = np.array(right, dtype=np.float64)
    return int(lib().orc_compare_scores(_p(l, C.c_double), len(left), lratio, _p(r, C.c_double),
                                        len(right), rratio))


def typo_budget(word, min_one=5, min_two=9):
    b = np.frombuffer(word.encode("utf-8") if isinstance(word, str) else word, dtype=np.uint8)
    return int(lib().orc_typo_budget(_p(b, C.c_uint8), b.size, min_one, min_two))


def osa_distance(a, b, prefix=False):
    a = np.frombuffer(a.encode("utf-8") if isinstance(a, str) else a, dtype=np.uint8)
    b = np.frombuffer(b.encode("utf-8") if isinstance(b, str) else b, dtype=np.uint8)
    return int(lib().orc_osa_distance(_p(a, C.c_uint8), a.size, _p(b, C.c_uint8), b.size,
                                      1 if prefix else 0))


class Dictionary:
    """Flat sorted dictionary: concatenated UTF-8 bytes + offsets (fst stream order)."""

    def __init__(self, words):
        bs = [w.encode("utf-8") if isinstance(w, str) else bytes(w) for w in words]
        assert all(bs[i] < bs[i + 1] for i in range(len(bs) - 1)), "dictionary must be sorted+unique"
        self.words = bs
        self.concat = np.frombuffer(b"".join(bs), dtype=np.uint8).copy() if bs else np.zeros(0, np.uint8)
        off = np.zeros(len(bs) + 1, dtype=np.uint32)
        np.cumsum([len(b) for b in bs], out=off[1:])
        self.offsets = off

    @classmethod
    def from_flat(cls, concat, offsets):
        self = cls.__new__(cls)
        self.concat = np.ascontiguousarray(concat, dtype=np.uint8)
        self.offsets = np.ascontiguousarray(offsets, dtype=np.uint32)
        self.words = None
        return self

    def __len__(self):
        return self.offsets.size - 1

    def word(self, i):
        return bytes(self.concat[self.offsets[i]:self.offsets[i + 1]])


def typo_lookup(dic, word, max_typos, is_prefix, cap_one=150, cap_two=50):
    """Returns (one_typo_indices, two_typo_indices) as numpy u32 arrays."""
    q = np.frombuffer(word.encode("utf-8") if isinstance(word, str) else word, dtype=np.uint8)
    one = np.zeros(cap_one + 1, dtype=np.uint32)
    two = np.zeros(cap_two + 1, dtype=np.uint32)
    n1, n2 = C.c_uint32(0), C.c_uint32(0)
    concat = dic.concat if dic.concat.size else np.zeros(1, np.uint8)
    lib().orc_typo_lookup(_p(concat, C.c_uint8), _p(dic.offsets, C.c_uint32), len(dic),
                          _p(q, C.c_uint8), q.size, max_typos, 1 if is_prefix else 0, cap_one,
                          cap_two, _p(one, C.c_uint32), C.byref(n1), _p(two, C.c_uint32),
                          C.byref(n2))
    return one[:n1.value].copy(), two[:n2.value].copy()


def bq_topk(rows, docids, q, k, filter_bits=None, filter_nbits=0):
    """Exact top-k of a binary-quantised store (orc_bq_topk: bit = x > 0, distance = hamming / dim, (distance, docid))."""
    rows = np.ascontiguousarray(rows, dtype=np.float32)
    docids = np.ascontiguousarray(docids, dtype=np.uint32)
    q = np.ascontiguousarray(q, dtype=np.float32)
    n, d = rows.shape if rows.ndim == 2 else (0, q.size)
    out_d = np.zeros(max(k, 1), dtype=np.uint32)
    out_s = np.zeros(max(k, 1), dtype=np.float32)
    cnt = C.c_uint32(0)
    fb = None
    if filter_bits is not None:
        filter_bits = np.ascontiguousarray(filter_bits, dtype=np.uint64)
        fb = _p(filter_bits, C.c_uint64)
    lib().orc_bq_topk(_p(rows, C.c_float), _p(docids, C.c_uint32), n, d, _p(q, C.c_float), k, fb, filter_nbits,
                      _p(out_d, C.c_uint32), _p(out_s, C.c_float), C.byref(cnt))
    return out_d[:cnt.value].copy(), out_s[:cnt.value].copy()


def merge_positioned_hits_into_page(pins, skip, take, organic_hits):
    """crates/milli/src/search/mod.rs:579-625, restated line for line.  pins: [(position, hit)] in the order the caller
    resolved them; organic_hits: the bucket sort's prefix [0, skip + take).  -> the page [skip, skip + take)."""
    if not pins:
        return list(organic_hits)
    page_end = skip + take
    merged, organic, pin_i, combined = [], iter(organic_hits), 0, 0
    while combined < page_end:
        if pin_i < len(pins):
            if pins[pin_i][0] <= combined:
                hit = pins[pin_i][1]
                pin_i += 1
            else:
                hit = next(organic, None)
                if hit is None:
                    hit = pins[pin_i][1]
                    pin_i += 1
        else:
            hit = next(organic, None)
        if hit is None:
            break
        if combined >= skip:
            merged.append(hit)
        combined += 1
    return merged
