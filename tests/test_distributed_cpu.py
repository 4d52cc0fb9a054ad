"""N>1 path on CPU: world_size-2 (and 3) gloo process groups exercise the row and
query partitioning, the one exchange step (all-gather) and the host merge
(msi_merge_topk).  The device scan is replaced by the oracle as `local_search`
(test infrastructure standing in for GpuStore.search; the product code under test
is meilisearch_amd/distributed.py + libmsi's host merge)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, mode, out_dir):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from meilisearch_amd import distributed as D
    from meilisearch_amd import synth
    from oracle import oracle as orc
    n, dim, k, b = 2003, 24, 10, 7
    rows = synth.make_embeddings(n, dim, seed=5)
    rows[100:140] = rows[100]                       # ties across the shard boundary ordering rule
    ids = np.arange(n, dtype=np.uint32) * 3 + 1
    qs = synth.make_embeddings(b, dim, seed=6)

    def make_local(r0, r1):
        def local(q, kk):
            d = np.full((q.shape[0], kk), 0xFFFFFFFF, dtype=np.uint32)
            s = np.full((q.shape[0], kk), np.inf, dtype=np.float32)
            c = np.zeros(q.shape[0], dtype=np.uint32)
            for j in range(q.shape[0]):
                e_ids, e_dist = orc.vs_topk(rows[r0:r1], ids[r0:r1], q[j], kk)
                d[j, :e_ids.size], s[j, :e_ids.size], c[j] = e_ids, e_dist, e_ids.size
            return d, s, c
        return local

    if mode == "rows":
        r0, r1 = D.row_range(n, rank, world)
        got = D.ShardedSearch(make_local(r0, r1)).search(qs, k)
    else:
        got = D.shard_queries(make_local(0, n), qs, k)
    exp = make_local(0, n)(qs, k)
    ok = all((g == e).all() if g.dtype != np.float32 else (g.view(np.uint32) == e.view(np.uint32)).all()
             for g, e in zip(got, exp))
    with open(os.path.join(out_dir, f"ok_{mode}_{rank}"), "w") as f:
        f.write("1" if ok else "0")
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,mode", [(2, "rows"), (2, "queries"), (3, "rows"), (3, "queries")])
def test_sharded_search_matches_single_process(tmp_path, world, mode, oracle):
    port = _free_port()
    mp.spawn(_worker, args=(world, port, mode, str(tmp_path)), nprocs=world, join=True)
    for r in range(world):
        assert open(tmp_path / f"ok_{mode}_{r}").read() == "1", (mode, r)


def test_row_and_query_ranges():
    from meilisearch_amd import distributed as D
    for n in (0, 1, 7, 8, 1000003):
        for w in (1, 2, 3, 8):
            rs = [D.row_range(n, r, w) for r in range(w)]
            assert rs[0][0] == 0 and rs[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(rs, rs[1:]))
            assert max(b - a for a, b in rs) - min(b - a for a, b in rs) <= 1


def test_merge_topk_host():
    from meilisearch_amd import distributed as D
    d = np.array([[5, 9, 11, 0], [2, 7, 0, 0], [1, 0, 0, 0]], dtype=np.uint32)
    s = np.array([[0.1, 0.2, 0.3, 0], [0.1, 0.25, 0, 0], [0.05, 0, 0, 0]], dtype=np.float32)
    c = np.array([3, 2, 1], dtype=np.uint32)
    md, ms = D.merge_topk(d, s, c, 4)
    assert md.tolist() == [1, 2, 5, 9]            # 0.1 tie -> ascending docid
    assert ms.tolist() == [np.float32(0.05), np.float32(0.1), np.float32(0.1), np.float32(0.2)]
    md, ms = D.merge_topk(d, s, c, 100)
    assert md.size == 6
