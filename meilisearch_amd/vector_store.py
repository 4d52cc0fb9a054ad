"""Python binding + host-side mirror of milli's VectorStore (S1 seam).

`GpuStore` wraps one `msi_vs` (one arroy/hannoy index = one (embedder, store)
pair).  `VectorStore` mirrors `crates/milli/src/vector/store.rs`: up to 256
stores per embedder (store.rs:1427-1434), `nns_by_vector` / `nns_by_item`
concatenating the per-store results and sorting them by distance
(store.rs:980-1093), `item_vectors` (store.rs:676-720).
"""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import VsStats, check, lib
from .device import np_ptr


def dense_filter(docids, nbits=None):
    """Dense bitset (u64 words, LSB first) of a docid collection — the boundary
    form of the RoaringBitmap `filter` (store.rs:641-643)."""
    ids = np.asarray(sorted(set(int(d) for d in docids)), dtype=np.uint64)
    if nbits is None:
        nbits = int(ids.max()) + 1 if ids.size else 0
    words = np.zeros((nbits + 63) // 64 or 1, dtype=np.uint64)
    ids = ids[ids < nbits]
    np.bitwise_or.at(words, (ids >> np.uint64(6)).astype(np.int64),
                     np.uint64(1) << (ids & np.uint64(63)))
    return words, nbits


class GpuStore:
    """One vector store resident in HBM (msi_vs)."""

    def __init__(self, ctx, dim, storage="f32"):
        self.ctx = ctx
        self.dim = int(dim)
        self.storage = storage
        self._h = C.c_void_p()
        check(lib().msi_vs_create_typed(ctx.handle, self.dim, {"f32": 0, "bf16": 1}[storage], C.byref(self._h)))

    def upload(self, docids, rows):
        docids = np.ascontiguousarray(docids, dtype=np.uint32)
        rows = np.ascontiguousarray(rows, dtype=np.float32).reshape(-1, self.dim)
        assert rows.shape[0] == docids.shape[0]
        check(lib().msi_vs_upload(self._h, np_ptr(docids), np_ptr(rows), docids.shape[0]))

    def update(self, remove_docids=(), add_docids=(), add_rows=None):
        """msi_vs_update: the documents of remove_docids leave the store, the rows of add_docids enter it (an existing
        docid is replaced); both lists strictly ascending.  Only the delta travels over PCIe."""
        rm = np.ascontiguousarray(remove_docids, dtype=np.uint32)
        ad = np.ascontiguousarray(add_docids, dtype=np.uint32)
        rows = np.ascontiguousarray(add_rows if add_rows is not None else np.zeros((0, self.dim)), dtype=np.float32).reshape(-1, self.dim)
        assert rows.shape[0] == ad.shape[0]
        check(lib().msi_vs_update(self._h, np_ptr(rm) if rm.size else None, rm.size, np_ptr(ad) if ad.size else None,
                                  np_ptr(rows) if ad.size else None, ad.size))

    def upload_device(self, docids_t, rows_t):
        """docids_t: cuda int32/uint32-compatible tensor, rows_t: cuda f32 [n, dim]."""
        assert _lib.on_device(rows_t) and rows_t.is_contiguous() and _lib.on_device(docids_t)
        import torch
        torch.cuda.current_stream(rows_t.device).synchronize()   # libmsi works on its own stream
        n = rows_t.shape[0]
        check(lib().msi_vs_upload_device(self._h, C.c_void_p(docids_t.data_ptr()),
                                         C.c_void_p(rows_t.data_ptr()), n))

    def __len__(self):
        return int(lib().msi_vs_len(self._h))

    @property
    def max_batch(self):
        """Queries answered by one HBM sweep (16/32/48, by dimension)."""
        return int(lib().msi_vs_max_batch(self._h))

    def get_vector(self, docid):
        out = np.zeros(self.dim, dtype=np.float32)
        found = C.c_int32(0)
        check(lib().msi_vs_get_vector(self._h, int(docid), np_ptr(out), C.byref(found)))
        return out if found.value else None

    def search(self, queries, k, filter_bits=None, filter_nbits=0, cancel=None):
        """queries [B, dim] -> (docids [B,k] u32, dist [B,k] f32, counts [B] u32)."""
        q = np.ascontiguousarray(queries, dtype=np.float32).reshape(-1, self.dim)
        b = q.shape[0]
        out_d = np.full((b, max(k, 1)), 0xFFFFFFFF, dtype=np.uint32)
        out_s = np.full((b, max(k, 1)), np.inf, dtype=np.float32)
        cnt = np.zeros(b, dtype=np.uint32)
        fb = None
        if filter_bits is not None:
            fb = np.ascontiguousarray(filter_bits, dtype=np.uint64)
            if not filter_nbits:
                filter_nbits = 64 * fb.size   # a filter without a bit count means "all of its words"
        cancel_p = None
        if cancel is not None:
            cancel_p = cancel.ctypes.data_as(C.c_void_p)
        check(lib().msi_vs_search(self._h, np_ptr(q), b, k, np_ptr(fb), filter_nbits, cancel_p,
                                  np_ptr(out_d), np_ptr(out_s), np_ptr(cnt)))
        return out_d[:, :k], out_s[:, :k], cnt

    def search_device(self, q_t, k, out_docids_t, out_dist_t, out_counts_t, inexact_t=None,
                      filter_ptr=None, filter_nbits=0):
        """Enqueue a search on the context stream (chunks of `max_batch` queries
        per HBM sweep); no sync."""
        nq = q_t.shape[0]
        check(lib().msi_vs_search_device(
            self._h, C.c_void_p(q_t.data_ptr()), nq, k, filter_ptr, filter_nbits,
            C.c_void_p(out_docids_t.data_ptr()), C.c_void_p(out_dist_t.data_ptr()),
            C.c_void_p(out_counts_t.data_ptr()),
            C.c_void_p(inexact_t.data_ptr()) if inexact_t is not None else None))

    def search_by_item(self, docid, k, filter_bits=None, filter_nbits=0):
        """nns_by_item for this store -> (docids, dist) or None when the item has no vector here."""
        out_d = np.zeros(max(k, 1), dtype=np.uint32)
        out_s = np.zeros(max(k, 1), dtype=np.float32)
        cnt, found = C.c_uint32(0), C.c_int32(0)
        fb = None if filter_bits is None else np.ascontiguousarray(filter_bits, dtype=np.uint64)
        if fb is not None and not filter_nbits:
            filter_nbits = 64 * fb.size
        check(lib().msi_vs_search_by_item(self._h, int(docid), k, np_ptr(fb), filter_nbits, np_ptr(out_d), np_ptr(out_s),
                                          C.byref(cnt), C.byref(found)))
        if not found.value:
            return None
        return out_d[:cnt.value].copy(), out_s[:cnt.value].copy()

    def set_sweep_split(self, n):
        """A full sweep as n times the workgroups (short workgroups: for a host that runs other device work beside the sweeps)."""
        check(lib().msi_vs_set_sweep_split(self._h, int(n)))

    def set_microbatch(self, max_wait_us):
        """Fuse concurrent unfiltered `search` calls (other threads) into shared HBM sweeps."""
        check(lib().msi_vs_set_microbatch(self._h, int(max_wait_us)))

    def microbatch_stats(self):
        a, b = C.c_uint64(0), C.c_uint64(0)
        check(lib().msi_vs_microbatch_stats(self._h, C.byref(a), C.byref(b)))
        return {"fused_calls": int(a.value), "fused_sweeps": int(b.value)}

    def debug_fast_scores(self, queries):
        """(scores [B, len] of the fast scan = dot/|row|, eps assumed by the proof)."""
        q = np.ascontiguousarray(queries, dtype=np.float32).reshape(-1, self.dim)
        out = np.zeros((q.shape[0], len(self)), dtype=np.float32)
        eps = C.c_float(0.0)
        check(lib().msi_vs_debug_fast_scores(self._h, np_ptr(q), q.shape[0], np_ptr(out), C.byref(eps)))
        return out, float(eps.value)

    def scan_time(self):
        """(launches, total ms) of the main-pass vs_scan kernel since the last call
        (HIP events on the launch stream; needs Context.set_profiling(True))."""
        n, ms = C.c_uint64(0), C.c_double(0.0)
        check(lib().msi_vs_scan_time(self._h, C.byref(n), C.byref(ms)))
        return int(n.value), float(ms.value)

    def filter_stats(self):
        """The last filtered search as the device counted it: items visited (16 rows each), allowed rows, items written as
        compacted allowed rows / as whole tiles (msi_vs_filter_stats)."""
        o = (C.c_uint64 * 4)()
        check(lib().msi_vs_filter_stats(self._h, o))
        return {"items": int(o[0]), "allowed_rows": int(o[1]), "compact_items": int(o[2]), "tile_items": int(o[3])}

    def stats(self):
        s = VsStats()
        check(lib().msi_vs_get_stats(self._h, C.byref(s)))
        return {"scan_launches": s.scan_launches, "scan_tiles": s.scan_tiles,
                "exhaustive_reruns": s.exhaustive_reruns, "bytes_per_tile": s.bytes_per_tile,
                "second_opinion_queries": s.second_opinion_queries, "x3_first_sweeps": s.x3_first_sweeps,
                "x2_sweeps": s.x2_sweeps, "level_sweeps": [int(x) for x in s.level_sweeps],
                "i8_bytes_per_tile": int(s.i8_bytes_per_tile), "i8_sweeps": int(s.i8_sweeps),
                "device_rerun_queries": int(s.device_rerun_queries), "i8_queries_per_sweep": int(s.i8_queries_per_sweep),
                "f32_queries_per_sweep": int(s.f32_queries_per_sweep), "i8_scan_tiles": int(s.i8_scan_tiles)}

    def close(self):
        if self._h:
            lib().msi_vs_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class GpuBqStore:
    """One binary-quantised vector store resident in HBM (msi_bq: sign bits as bit planes, exact Hamming k-NN)."""

    def __init__(self, ctx, dim):
        self.ctx, self.dim = ctx, int(dim)
        self._h = C.c_void_p()
        check(lib().msi_bq_create(ctx.handle, self.dim, C.byref(self._h)))

    def upload(self, docids, rows):
        docids = np.ascontiguousarray(docids, dtype=np.uint32)
        rows = np.ascontiguousarray(rows, dtype=np.float32).reshape(-1, self.dim)
        assert rows.shape[0] == docids.shape[0]
        check(lib().msi_bq_upload(self._h, np_ptr(docids), np_ptr(rows), docids.shape[0]))

    def upload_device(self, docids_t, rows_t):
        import torch
        torch.cuda.current_stream(rows_t.device).synchronize()
        check(lib().msi_bq_upload_device(self._h, C.c_void_p(docids_t.data_ptr()), C.c_void_p(rows_t.data_ptr()), rows_t.shape[0]))

    def __len__(self):
        return int(lib().msi_bq_len(self._h))

    def get_vector(self, docid):
        out = np.zeros(self.dim, dtype=np.float32)
        found = C.c_int32(0)
        check(lib().msi_bq_get_vector(self._h, int(docid), np_ptr(out), C.byref(found)))
        return out if found.value else None

    def search(self, queries, k, filter_bits=None, filter_nbits=0):
        q = np.ascontiguousarray(queries, dtype=np.float32).reshape(-1, self.dim)
        b = q.shape[0]
        out_d = np.full((b, max(k, 1)), 0xFFFFFFFF, dtype=np.uint32)
        out_s = np.full((b, max(k, 1)), np.inf, dtype=np.float32)
        cnt = np.zeros(b, dtype=np.uint32)
        fb = None
        if filter_bits is not None:
            fb = np.ascontiguousarray(filter_bits, dtype=np.uint64)
            if not filter_nbits:
                filter_nbits = 64 * fb.size
        check(lib().msi_bq_search(self._h, np_ptr(q), b, k, np_ptr(fb), filter_nbits, np_ptr(out_d), np_ptr(out_s), np_ptr(cnt)))
        return out_d[:, :k], out_s[:, :k], cnt

    def close(self):
        if self._h:
            lib().msi_bq_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class VectorStore:
    """Mirror of milli's `VectorStore` for one embedder (store.rs:29-130).

    A document with several embeddings has one row in each of the first stores
    (store.rs:752-786), so each store holds at most one row per docid.
    """

    MAX_STORES = 256  # vector_store_range_for_embedder, store.rs:1427-1429

    def __init__(self, ctx, dim):
        self.ctx = ctx
        self.dim = int(dim)
        self.stores = {}  # store_id -> GpuStore

    def set_store(self, store_id, docids, rows):
        assert 0 <= store_id < self.MAX_STORES
        docids = np.asarray(docids, dtype=np.uint32)
        rows = np.asarray(rows, dtype=np.float32).reshape(-1, self.dim)
        order = np.argsort(docids, kind="stable")
        st = self.stores.get(store_id) or GpuStore(self.ctx, self.dim)
        st.upload(docids[order], rows[order])
        self.stores[store_id] = st

    def add_documents(self, embeddings_by_docid):
        """embeddings_by_docid: {docid: [vec, vec, ...]} — i-th vector goes to store i
        (add_items, store.rs:752-786)."""
        per_store = {}
        for docid, vecs in embeddings_by_docid.items():
            for i, v in enumerate(vecs):
                per_store.setdefault(i, []).append((docid, v))
        for sid, items in per_store.items():
            self.set_store(sid, [d for d, _ in items], np.stack([v for _, v in items]))

    def _readers(self):
        # _arroy_readers/_hannoy_readers skip empty stores (store.rs:1111-1150)
        return [self.stores[s] for s in sorted(self.stores) if len(self.stores[s])]

    def nns_by_vector(self, vector, limit, filter_docids=None):
        """store.rs:638-675,1036-1093 → list of (docid, distance) ascending."""
        fb, nb = (None, 0)
        if filter_docids is not None:
            fb, nb = dense_filter(filter_docids)
        results = []
        for st in self._readers():
            d, s, c = st.search(np.asarray(vector, dtype=np.float32)[None, :], limit, fb, nb)
            results += [(int(d[0, i]), float(s[0, i])) for i in range(int(c[0]))]
        results.sort(key=lambda t: (t[1], t[0]))
        return results

    def nns_by_item(self, item, limit, filter_docids=None):
        """store.rs:615-637,980-1034: per store, query = the item's vector there."""
        fb, nb = (None, 0)
        if filter_docids is not None:
            fb, nb = dense_filter(filter_docids)
        results = []
        for st in self._readers():
            r = st.search_by_item(item, limit, fb, nb)
            if r is None:
                continue
            results += [(int(a), float(b)) for a, b in zip(*r)]
        results.sort(key=lambda t: (t[1], t[0]))
        return results

    def item_vectors(self, docid):
        out = []
        for st in self._readers():
            v = st.get_vector(docid)
            if v is not None:
                out.append(v)
        return out
