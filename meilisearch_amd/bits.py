"""Python binding of the dense docid-set pool (S3 seam, msi_bits_*)."""
import ctypes as C

import numpy as np

from ._lib import check, lib
from .device import np_ptr

AND, OR, ANDNOT, XOR = 0, 1, 2, 3
NO_UNIVERSE = 0xFFFFFFFF


class BitsPool:
    def __init__(self, ctx, n_docs, n_slots, private_stream=False):
        self.ctx = ctx
        self.n_docs = int(n_docs)
        self.n_slots = int(n_slots)
        self._h = C.c_void_p()
        check(lib().msi_bits_create(ctx.handle, self.n_docs, self.n_slots, C.byref(self._h)))
        if private_stream:
            check(lib().msi_bits_use_private_stream(self._h))

    def set_from_docids(self, slot, docids):
        d = np.ascontiguousarray(docids, dtype=np.uint32)
        check(lib().msi_bits_set_from_docids(self._h, slot, np_ptr(d) if d.size else None, d.size))

    def set_from_docid_lists_device(self, first_slot, slot_stride, docids_t, counts_t):
        """docids_t: cuda int32/uint32 [n_lists, list_stride], counts_t: cuda int32 [n_lists] (same stream as the pool)."""
        n_lists, list_stride = docids_t.shape
        check(lib().msi_bits_set_from_docid_lists_device(self._h, first_slot, slot_stride, C.c_void_p(docids_t.data_ptr()),
                                                        list_stride, C.c_void_p(counts_t.data_ptr()), n_lists))

    def set_from_cbo(self, slot, data):
        b = np.frombuffer(bytes(data), dtype=np.uint8)
        check(lib().msi_bits_set_from_cbo(self._h, slot, np_ptr(b) if b.size else None, b.size))

    def set_from_words(self, slot, words):
        w = np.ascontiguousarray(words, dtype=np.uint64)
        check(lib().msi_bits_set_from_words(self._h, slot, np_ptr(w) if w.size else None, w.size))

    def fill(self, slot, ones):
        check(lib().msi_bits_fill(self._h, slot, 1 if ones else 0))

    def op(self, dst, a, b, op):
        check(lib().msi_bits_op(self._h, dst, a, b, op))

    def union_many_and(self, dst, srcs, universe=NO_UNIVERSE):
        s = np.ascontiguousarray(srcs, dtype=np.uint32)
        check(lib().msi_bits_union_many_and(self._h, dst, np_ptr(s) if s.size else None, s.size, universe))

    def order_next(self, keys, universe, bucket):
        """The Sort rule's next bucket: bucket := the documents of `universe` with its smallest key, universe -= bucket.
        -> (key, count)."""
        key, n = C.c_uint32(0), C.c_uint64(0)
        check(lib().msi_bits_order_next(self._h, keys._h, universe, bucket, C.byref(key), C.byref(n)))
        return int(key.value), int(n.value)

    def facet_range(self, keys, lo, hi, dst, accumulate=False):
        """dst (|)= the documents with a facet key in [lo, hi] (msi_bits_facet_range; index_filter.rs:139-153)."""
        check(lib().msi_bits_facet_range(self._h, keys._h, int(lo), int(hi), dst, 1 if accumulate else 0))

    def facet_in(self, keys, sorted_keys, dst, accumulate=False):
        k = np.ascontiguousarray(sorted_keys, dtype=np.uint64)
        check(lib().msi_bits_facet_in(self._h, keys._h, np_ptr(k) if k.size else None, k.size, dst, 1 if accumulate else 0))

    def geo_within(self, points, src, lat, lng, radius_m, dst):
        """dst := the documents of src within radius_m metres of (lat, lng): `_geoRadius` (index_filter.rs:465-503)."""
        check(lib().msi_bits_geo_within(self._h, points._h, src, float(lat), float(lng), float(radius_m), dst))

    VECTOR_FILTER_KINDS = {"none": 0, "fragment": 1, "documentTemplate": 2, "userProvided": 3, "regenerate": 4}

    def vector_filter(self, dst, kind, stores=(), bq_stores=(), has_fragments=False, user_provided=0, skip_regenerate=0,
                      scratch=0, accumulate=False):
        """`_vectors.<embedder>[...]` for one embedder (search/facet/filter/vector.rs:78-158): the items of its stores
        minus the user-provided / skip-regenerate bitmaps as `kind` asks; accumulate=True ORs into dst."""
        import ctypes as C
        vs = (C.c_void_p * max(1, len(stores)))(*[s._h for s in stores])
        bq = (C.c_void_p * max(1, len(bq_stores)))(*[s._h for s in bq_stores])
        check(lib().msi_bits_vector_filter(self._h, dst, self.VECTOR_FILTER_KINDS[kind], 1 if has_fragments else 0, vs, len(stores),
                                           bq, len(bq_stores), user_provided, skip_regenerate, scratch, 1 if accumulate else 0))

    def geo_next(self, points, universe, bucket, scratch, lat, lng, ascending=True, max_bucket_size=1000, margin=1.0):
        """GeoSort's next bucket (documents/geo_sort.rs:150-224): the documents of `universe` within `margin` metres of
        the nearest (farthest) one, at most max_bucket_size; universe -= bucket.
        -> (docid whose point is the bucket's value | None when no document of the universe has a point, count)."""
        first, n = C.c_uint32(0), C.c_uint64(0)
        check(lib().msi_bits_geo_next(self._h, points._h, universe, bucket, scratch, float(lat), float(lng),
                                      1 if ascending else 0, int(max_bucket_size), float(margin), C.byref(first), C.byref(n)))
        return (None if first.value == 0xFFFFFFFF else int(first.value)), int(n.value)

    def distinct(self, values, candidates, remaining, excluded=NO_UNIVERSE):
        """apply_distinct_rule (search/new/distinct.rs:19-36): remaining := the candidates kept (one per value of the
        distinct field, smallest docid first), excluded := every document that shares a value with a kept one;
        `candidates` is consumed.  -> (|remaining|, parallel rounds, finished by the sequential kernel?)."""
        n, rounds = C.c_uint64(0), C.c_uint32(0)
        check(lib().msi_bits_distinct(self._h, values._h, candidates, remaining, excluded, C.byref(n), C.byref(rounds)))
        return int(n.value), int(rounds.value & 0x7FFFFFFF), bool(rounds.value >> 31)

    def distinct_excluded(self, values, kept, excluded):
        check(lib().msi_bits_distinct_excluded(self._h, values._h, kept, excluded))

    def andnot_many_count(self, removed, slots):
        s = np.ascontiguousarray(slots, dtype=np.uint32)
        out = np.zeros(s.size, dtype=np.uint64)
        check(lib().msi_bits_andnot_many_count(self._h, removed, s.size, np_ptr(s), np_ptr(out)))
        return [int(x) for x in out]

    def count(self, slot):
        out = C.c_uint64(0)
        check(lib().msi_bits_count(self._h, slot, C.byref(out)))
        return int(out.value)

    def first_k(self, slot, k):
        out = np.zeros(max(k, 1), dtype=np.uint32)
        n = C.c_uint32(0)
        check(lib().msi_bits_first_k(self._h, slot, k, np_ptr(out), C.byref(n)))
        return out[:n.value].copy()

    def read_words(self, slot):
        out = np.zeros((self.n_docs + 63) // 64 or 1, dtype=np.uint64)
        check(lib().msi_bits_read_words(self._h, slot, np_ptr(out)))
        return out[:(self.n_docs + 63) // 64]

    def to_docids(self, slot):
        w = self.read_words(slot)
        bits = np.unpackbits(w.view(np.uint8), bitorder="little")[:self.n_docs]
        return np.nonzero(bits)[0].astype(np.uint32)

    def device_ptr(self, slot):
        return lib().msi_bits_device_ptr(self._h, slot)

    def close(self):
        if self._h:
            lib().msi_bits_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class DocKeys:
    """One u32 order key per document, resident in HBM (msi_doc_keys_create): the Sort / Asc / Desc ranking rules
    (search/new/sort.rs:95-233) read it instead of walking the facet databases per query."""

    def __init__(self, ctx, keys):
        self.ctx = ctx
        self.keys = np.ascontiguousarray(keys, dtype=np.uint32)
        self._h = C.c_void_p()
        check(lib().msi_doc_keys_create(ctx.handle, np_ptr(self.keys), self.keys.size, C.byref(self._h)))

    def close(self):
        if self._h:
            lib().msi_doc_keys_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class DocValues:
    """The facet values of the distinct field per document, CSR in HBM (msi_doc_values_create): what
    apply_distinct_rule (search/new/distinct.rs:19-62) reads instead of walking the facet databases per candidate.
    per_doc: one sequence of value ids (< n_values) per document."""

    def __init__(self, ctx, per_doc, n_values):
        self.ctx = ctx
        self.offsets = np.zeros(len(per_doc) + 1, dtype=np.uint64)
        np.cumsum([len(v) for v in per_doc], out=self.offsets[1:])
        flat = [x for v in per_doc for x in v]
        self.values = np.ascontiguousarray(flat if flat else [0], dtype=np.uint32)
        self._h = C.c_void_p()
        check(lib().msi_doc_values_create(ctx.handle, np_ptr(self.offsets), np_ptr(self.values), len(per_doc), int(n_values),
                                          C.byref(self._h)))

    def close(self):
        if self._h:
            lib().msi_doc_values_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class GeoPoints:
    """The _geo point of every document in HBM (msi_geo_points_create): lat_lng [n_docs][2] f64, NaN = no point."""

    def __init__(self, ctx, lat_lng):
        self.ctx = ctx
        self.lat_lng = np.ascontiguousarray(lat_lng, dtype=np.float64).reshape(-1, 2)
        self._h = C.c_void_p()
        check(lib().msi_geo_points_create(ctx.handle, np_ptr(self.lat_lng), self.lat_lng.shape[0], C.byref(self._h)))

    def close(self):
        if self._h:
            lib().msi_geo_points_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def facet_number_key(x):
    """The monotone f64 -> u64 map of the number tables (msi_facet_number_key)."""
    return int(lib().msi_facet_number_key(float(x)))


class FacetKeys:
    """The facet values of one filterable field and kind per document, CSR of u64 sort keys in HBM
    (msi_facet_keys_create).  per_doc: one sequence of keys per document."""

    def __init__(self, ctx, per_doc):
        self.ctx = ctx
        self.offsets = np.zeros(len(per_doc) + 1, dtype=np.uint64)
        np.cumsum([len(v) for v in per_doc], out=self.offsets[1:])
        flat = [int(x) for v in per_doc for x in v]
        self.keys = np.array(flat if flat else [0], dtype=np.uint64)
        self._h = C.c_void_p()
        check(lib().msi_facet_keys_create(ctx.handle, np_ptr(self.offsets), np_ptr(self.keys), len(per_doc), C.byref(self._h)))

    def close(self):
        if self._h:
            lib().msi_facet_keys_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
