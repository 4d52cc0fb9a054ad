"""Multi-GPU host logic: one process per GPU (torch.distributed; backend "nccl" =
RCCL over xGMI on the GPU box, "gloo" in the CPU tests).

Two ways the path shards:

* `shard_queries` — the default: every rank holds a replica of the index and
  answers its own contiguous slice of the query batch; the per-rank result lists
  are all-gathered.  No data-path collective, weak scaling.
* `ShardedSearch` — for stores larger than one GPU: every rank scans its
  contiguous ROW range for the whole batch, the (distance, docid)[B, k] lists are
  all-gathered and merged by (distance, docid) — `msi_merge_topk`, the reference's
  concatenate + sort tail (crates/milli/src/vector/store.rs:1059,1090).

`local_search(queries, k) -> (docids [B,k] u32, dist [B,k] f32, counts [B] u32)` is
the device call (`GpuStore.search`) in production; the CPU tests inject a stand-in
so that the partitioning, the collective and the merge are exercised without a GPU.
"""
import ctypes as C

import numpy as np

from ._lib import lib
from .device import np_ptr


def row_range(n_rows, rank, world):
    """Contiguous row range [r0, r1) of `rank`; ranges differ by at most one row."""
    base, rem = divmod(int(n_rows), int(world))
    r0 = rank * base + min(rank, rem)
    return r0, r0 + base + (1 if rank < rem else 0)


def query_range(n_queries, rank, world):
    return row_range(n_queries, rank, world)


def merge_topk(docids, dist, counts, k_out):
    """Host merge of ascending lists: docids/dist [L, stride], counts [L]."""
    docids = np.ascontiguousarray(docids, dtype=np.uint32)
    dist = np.ascontiguousarray(dist, dtype=np.float32)
    counts = np.ascontiguousarray(counts, dtype=np.uint32)
    n_lists, stride = docids.shape
    out_d = np.zeros(max(k_out, 1), dtype=np.uint32)
    out_s = np.zeros(max(k_out, 1), dtype=np.float32)
    n = lib().msi_merge_topk(np_ptr(docids), np_ptr(dist), np_ptr(counts), n_lists, stride, k_out,
                             np_ptr(out_d), np_ptr(out_s))
    return out_d[:n].copy(), out_s[:n].copy()


def _all_gather(t, world, dist_mod):
    import torch
    t = t.contiguous()
    parts = [torch.empty_like(t) for _ in range(world)]
    dist_mod.all_gather(parts, t)   # RCCL all-gather on the GPU box, gloo in the CPU tests
    return torch.stack(parts)


class ShardedSearch:
    """Row-sharded exact k-NN.  `local_search` sees only this rank's rows."""

    def __init__(self, local_search, device=None):
        import torch.distributed as dist
        self.dist = dist
        self.rank = dist.get_rank()
        self.world = dist.get_world_size()
        self.local_search = local_search
        self.device = device

    def search(self, queries, k):
        import torch
        d, s, c = self.local_search(queries, k)
        b = d.shape[0]
        dev = self.device or torch.device("cpu")
        # one exchange step: B*k*8 bytes (+ counts) per rank
        d_t = torch.from_numpy(np.ascontiguousarray(d).view(np.int32)).to(dev)
        s_t = torch.from_numpy(np.ascontiguousarray(s)).to(dev)
        c_t = torch.from_numpy(np.ascontiguousarray(c).view(np.int32)).to(dev)
        gd = _all_gather(d_t, self.world, self.dist).cpu().numpy().view(np.uint32)
        gs = _all_gather(s_t, self.world, self.dist).cpu().numpy()
        gc = _all_gather(c_t, self.world, self.dist).cpu().numpy().view(np.uint32)
        out_d = np.full((b, k), 0xFFFFFFFF, dtype=np.uint32)
        out_s = np.full((b, k), np.inf, dtype=np.float32)
        out_c = np.zeros(b, dtype=np.uint32)
        for j in range(b):
            md, ms = merge_topk(gd[:, j, :], gs[:, j, :], gc[:, j], k)
            out_d[j, :md.size] = md
            out_s[j, :ms.size] = ms
            out_c[j] = md.size
        return out_d, out_s, out_c


def shard_queries(local_search, queries, k, device=None):
    """Query-sharded search over replicas: rank r answers queries[q0:q1]; every rank
    returns the full [B, k] result (all-gather of the per-rank slices, padded to the
    largest slice)."""
    import torch
    import torch.distributed as dist
    rank, world = dist.get_rank(), dist.get_world_size()
    b = queries.shape[0]
    q0, q1 = query_range(b, rank, world)
    per = -(-b // world)
    d = np.full((per, k), 0xFFFFFFFF, dtype=np.uint32)
    s = np.full((per, k), np.inf, dtype=np.float32)
    c = np.zeros(per, dtype=np.uint32)
    if q1 > q0:
        ld, ls, lc = local_search(queries[q0:q1], k)
        d[:q1 - q0], s[:q1 - q0], c[:q1 - q0] = ld, ls, lc
    dev = device or torch.device("cpu")
    gd = _all_gather(torch.from_numpy(d.view(np.int32)).to(dev), world, dist).cpu().numpy().view(np.uint32)
    gs = _all_gather(torch.from_numpy(s).to(dev), world, dist).cpu().numpy()
    gc = _all_gather(torch.from_numpy(c.view(np.int32)).to(dev), world, dist).cpu().numpy().view(np.uint32)
    out_d = np.concatenate([gd[r, :query_range(b, r, world)[1] - query_range(b, r, world)[0]] for r in range(world)])
    out_s = np.concatenate([gs[r, :query_range(b, r, world)[1] - query_range(b, r, world)[0]] for r in range(world)])
    out_c = np.concatenate([gc[r, :query_range(b, r, world)[1] - query_range(b, r, world)[0]] for r in range(world)])
    return out_d, out_s, out_c


def merge_topk_device(ctx, docids_t, dist_t, counts_t, out_docids_t, out_dist_t, out_counts_t):
    """Device k-way merge (msi_merge_topk_device): docids_t/dist_t [L, B, k], counts_t [L, B]
    torch CUDA tensors (the all-gathered per-shard lists); outputs [B, k] / [B]."""
    from ._lib import check
    n_lists, b, k = docids_t.shape
    check(lib().msi_merge_topk_device(ctx.handle, C.c_void_p(docids_t.data_ptr()), C.c_void_p(dist_t.data_ptr()),
                                      C.c_void_p(counts_t.data_ptr()), n_lists, b, k,
                                      C.c_void_p(out_docids_t.data_ptr()), C.c_void_p(out_dist_t.data_ptr()),
                                      C.c_void_p(out_counts_t.data_ptr())))
