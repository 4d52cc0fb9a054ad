"""Python binding + host-side mirror of milli's typo derivation (S2 seam).

`GpuDictionary` wraps one `msi_dict` (the words FST staged flat in HBM).
`number_of_typos_allowed`, `find_one_typo_derivations` and
`find_one_two_typo_derivations` mirror
crates/milli/src/search/new/query_term/{parse_query.rs:204-225,
compute_derivations.rs:75-168}; the batched entry point `lookup` is what a
micro-batching Rust shim would call.
"""
import ctypes as C

import numpy as np

from ._lib import DictStats, TypoQuery, check, lib
from .device import np_ptr

MAX_ONE_TYPO_COUNT = 150   # search/new/limits.rs:7
MAX_TWO_TYPOS_COUNT = 50   # search/new/limits.rs:9
MAX_WORD_LENGTH = 250      # crates/milli/src/lib.rs


def number_of_typos_allowed(word, authorize_typos=True, min_len_one_typo=5, min_len_two_typos=9,
                            exact_words=()):
    """parse_query.rs:204-225 — thresholds count chars, not bytes."""
    n = len(word)
    if not authorize_typos or n < min_len_one_typo or word in exact_words:
        return 0
    if n < min_len_two_typos:
        return 1
    return 2


def pack_queries(queries):
    """[(word, max_typos, is_prefix)] -> (bytes u8, offsets u32, flags u8): the packed
    device-side form of msi_dict_lookup_device (flags = max_typos | is_prefix << 2)."""
    bs = [w.encode("utf-8") if isinstance(w, str) else bytes(w) for w, _, _ in queries]
    qb = np.frombuffer(b"".join(bs), dtype=np.uint8).copy() if bs else np.zeros(0, np.uint8)
    if qb.size == 0:
        qb = np.zeros(1, np.uint8)
    off = np.zeros(len(bs) + 1, dtype=np.uint32)
    if bs:
        np.cumsum([len(b) for b in bs], out=off[1:])
    fl = np.array([(min(mt, 2) & 3) | (4 if pf else 0) for _, mt, pf in queries], dtype=np.uint8)
    return qb, off, fl


FST_SKIP_CHECKSUM = 1


def fst_decode(fst_bytes, flags=0):
    """Keys of an `fst::Set` blob (main["words-fst"], index.rs:1225-1243) in stream order, flat:
    (concat u8, offsets u32[n+1]) — host-side msi_fst_decode, two calls (sizes, then data)."""
    buf = np.frombuffer(bytes(fst_bytes), dtype=np.uint8)
    n, nb = C.c_uint32(0), C.c_uint64(0)
    check(lib().msi_fst_decode(np_ptr(buf) if buf.size else None, buf.size, flags, None, 0, None, 0, C.byref(n), C.byref(nb)))
    concat = np.zeros(max(nb.value, 1), dtype=np.uint8)
    offsets = np.zeros(n.value + 1, dtype=np.uint32)
    check(lib().msi_fst_decode(np_ptr(buf), buf.size, flags, np_ptr(concat), nb.value, np_ptr(offsets), n.value,
                               C.byref(n), C.byref(nb)))
    return concat[:nb.value], offsets


class GpuDictionary:
    """Sorted, unique word list (the keys of `word_docids`, index.rs:1238-1243)."""

    @classmethod
    def from_fst(cls, ctx, fst_bytes, facet_values=False):
        """The dictionary straight from milli's `fst::Set` bytes (msi_dict_create_from_fst /
        msi_dict_create_values_from_fst): no key list crosses the boundary."""
        self = cls.__new__(cls)
        self.ctx, self.facet_values = ctx, facet_values
        buf = np.frombuffer(bytes(fst_bytes), dtype=np.uint8)
        self._h = C.c_void_p()
        create = lib().msi_dict_create_values_from_fst if facet_values else lib().msi_dict_create_from_fst
        check(create(ctx.handle, np_ptr(buf), buf.size, C.byref(self._h)))
        self.concat, self.offsets = fst_decode(fst_bytes, FST_SKIP_CHECKSUM)   # only for word(i) on the Python side
        return self

    def __init__(self, ctx, words=None, concat=None, offsets=None, facet_values=False):
        """facet_values=True stages the (sorted, unique, normalised) values of one facet for `search_values`
        (msi_dict_create_values)."""
        self.ctx = ctx
        self.facet_values = facet_values
        if words is not None:
            bs = [w.encode("utf-8") if isinstance(w, str) else bytes(w) for w in words]
            concat = np.frombuffer(b"".join(bs), dtype=np.uint8).copy() if bs else np.zeros(0, np.uint8)
            offsets = np.zeros(len(bs) + 1, dtype=np.uint32)
            if bs:
                np.cumsum([len(b) for b in bs], out=offsets[1:])
        self.concat = np.ascontiguousarray(concat, dtype=np.uint8)
        self.offsets = np.ascontiguousarray(offsets, dtype=np.uint32)
        self._h = C.c_void_p()
        cc = self.concat if self.concat.size else np.zeros(1, np.uint8)
        create = lib().msi_dict_create_values if facet_values else lib().msi_dict_create
        check(create(ctx.handle, np_ptr(cc), np_ptr(self.offsets), self.offsets.size - 1, C.byref(self._h)))

    def __len__(self):
        return int(lib().msi_dict_len(self._h))

    def word(self, i):
        return bytes(self.concat[self.offsets[i]:self.offsets[i + 1]]).decode("utf-8")

    def search_values(self, query, max_typos, cap=1000):
        """Facet search (search/facet/search.rs:122-190): indices of the values with a prefix within `max_typos`
        edits of `query`, in stream order; -> (indices, truncated)."""
        q = np.frombuffer(query.encode("utf-8") if isinstance(query, str) else bytes(query), dtype=np.uint8)
        out = np.zeros(max(cap, 1), dtype=np.uint32)
        n, trunc = C.c_uint32(0), C.c_int32(0)
        check(lib().msi_dict_search_values(self._h, np_ptr(q) if q.size else None, q.size, max_typos, cap, np_ptr(out),
                                           C.byref(n), C.byref(trunc)))
        return out[:n.value].copy(), bool(trunc.value)

    def lookup(self, queries, cap_one=MAX_ONE_TYPO_COUNT, cap_two=MAX_TWO_TYPOS_COUNT):
        """queries: list of (word, max_typos, is_prefix).  Returns a list of
        (one_typo_indices, two_typo_indices) numpy arrays in dictionary order."""
        n = len(queries)
        if n == 0:
            return []
        arr = (TypoQuery * n)()
        keep = []
        for i, (w, mt, pf) in enumerate(queries):
            b = w.encode("utf-8") if isinstance(w, str) else bytes(w)
            buf = C.create_string_buffer(b, len(b))
            keep.append(buf)
            arr[i].word = C.cast(buf, C.c_void_p)
            arr[i].len = len(b)
            arr[i].max_typos = mt
            arr[i].is_prefix = 1 if pf else 0
        one = np.zeros((n, cap_one), dtype=np.uint32)
        two = np.zeros((n, cap_two), dtype=np.uint32)
        c1 = np.zeros(n, dtype=np.uint32)
        c2 = np.zeros(n, dtype=np.uint32)
        check(lib().msi_dict_lookup(self._h, arr, n, cap_one, cap_two, np_ptr(one), np_ptr(c1),
                                    np_ptr(two), np_ptr(c2)))
        return [(one[i, :c1[i]].copy(), two[i, :c2[i]].copy()) for i in range(n)]

    def lookup_device(self, qbytes_t, qoff_t, qflags_t, n, one_t, one_cnt_t, two_t, two_cnt_t,
                      cap_one=MAX_ONE_TYPO_COUNT, cap_two=MAX_TWO_TYPOS_COUNT):
        check(lib().msi_dict_lookup_device(
            self._h, C.c_void_p(qbytes_t.data_ptr()), C.c_void_p(qoff_t.data_ptr()),
            C.c_void_p(qflags_t.data_ptr()), n, cap_one, cap_two, C.c_void_p(one_t.data_ptr()),
            C.c_void_p(one_cnt_t.data_ptr()), C.c_void_p(two_t.data_ptr()),
            C.c_void_p(two_cnt_t.data_ptr())))

    def set_microbatch(self, max_wait_us, target_words=256):
        """Fuse concurrent small `lookup` calls (other threads) into shared launches."""
        check(lib().msi_dict_set_microbatch(self._h, int(max_wait_us), int(target_words)))

    def microbatch_stats(self):
        a, b = C.c_uint64(0), C.c_uint64(0)
        check(lib().msi_dict_microbatch_stats(self._h, C.byref(a), C.byref(b)))
        return {"fused_calls": int(a.value), "fused_launches": int(b.value)}

    def enable_posting_cache(self, capacity_bytes):
        """HBM cache of the index version's stored postings for msi_keyword_search_ranked (msi.h)."""
        check(lib().msi_dict_enable_posting_cache(self._h, int(capacity_bytes)))

    def posting_cache_stats(self):
        out = (C.c_uint64 * 4)()
        check(lib().msi_dict_posting_cache_stats(self._h, out))
        return {"hits": int(out[0]), "misses": int(out[1]), "bytes_used": int(out[2]), "capacity": int(out[3])}

    def match_time(self):
        n, ms = C.c_uint64(0), C.c_double(0.0)
        check(lib().msi_dict_match_time(self._h, C.byref(n), C.byref(ms)))
        return int(n.value), float(ms.value)

    def stats(self):
        s = DictStats()
        check(lib().msi_dict_get_stats(self._h, C.byref(s)))
        return {"lookup_launches": s.lookup_launches, "pairs_scanned": s.pairs_scanned,
                "dict_bytes": s.dict_bytes}

    # -- mirrors of the reference functions (single query) ---------------------
    def find_one_typo_derivations(self, word, is_prefix):
        """compute_derivations.rs:75-107 → list of derived words (stream order)."""
        one, _ = self.lookup([(word, 1, is_prefix)])[0]
        return [self.word(i) for i in one]

    def find_one_two_typo_derivations(self, word, is_prefix):
        """compute_derivations.rs:109-168 → (one_typo_words, two_typo_words)."""
        one, two = self.lookup([(word, 2, is_prefix)])[0]
        return [self.word(i) for i in one], [self.word(i) for i in two]

    def close(self):
        if self._h:
            lib().msi_dict_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
