"""SURVEY §8 e — the multi-GPU entry points of the C ABI (msi_group / msi_vs_group, RCCL inside libmsi), on the one
device a test box has: an in-process group of one device in both modes must answer exactly like a plain store (and like
the oracle), and the per-rank form (the one bench.py uses under torchrun) must join a world of one and all-gather a
device buffer onto itself.  World sizes > 1 need more devices than the test tier has; the exchange and merge logic for
them is covered by the shard-emulation test of tests/test_vs_gpu.py and the gloo tests of tests/test_distributed_cpu.py."""
import ctypes as C
import os
import threading

import numpy as np
import pytest

import meilisearch_amd as ma
from meilisearch_amd import synth
from meilisearch_amd._lib import check, lib
from meilisearch_amd.device import np_ptr

pytestmark = pytest.mark.gpu


def devices_of_the_box():
    """Devices libmsi can open: torch's count on a GPU box; the emulated tier (tests/test_kernels_emulated_cpu.py, group
    "multi-device") sets MSI_EMU_DEVICES and has no torch device at all."""
    if os.environ.get("MSI_EMU_DEVICES"):
        return int(os.environ["MSI_EMU_DEVICES"])
    import torch
    return torch.cuda.device_count()


@pytest.mark.parametrize("mode", [0, 1])   # MSI_GROUP_REPLICATE, MSI_GROUP_SHARD_ROWS
def test_in_process_group_of_one_device(oracle, mode):
    devs = (C.c_int32 * 1)(0)
    g = C.c_void_p()
    check(lib().msi_group_create(devs, 1, C.byref(g)))
    assert lib().msi_group_size(g) == 1 and lib().msi_group_ctx(g, 0)
    vs = C.c_void_p()
    check(lib().msi_vs_group_create(g, 96, 0, mode, C.byref(vs)))
    rows = synth.make_embeddings(7000, 96, seed=3)
    ids = (np.arange(7000, dtype=np.uint32) * 2 + 1)
    check(lib().msi_vs_group_upload(vs, np_ptr(ids), np_ptr(rows), 7000))
    q = synth.make_embeddings(37, 96, seed=4)
    k = 20
    out_d = np.zeros((37, k), np.uint32)
    out_s = np.zeros((37, k), np.float32)
    cnt = np.zeros(37, np.uint32)
    check(lib().msi_vs_group_search(vs, np_ptr(q), 37, k, np_ptr(out_d), np_ptr(out_s), np_ptr(cnt)))
    for j in range(37):
        e_ids, e_dist = oracle.vs_topk(rows, ids, q[j], k)
        assert int(cnt[j]) == k and out_d[j].tolist() == e_ids.tolist()
        assert out_s[j].view(np.uint32).tolist() == e_dist.view(np.uint32).tolist()
    # duplicates force the exactness proof to fail on the shard: the group falls back to the exhaustive host path
    rows2 = np.repeat(rows[:50], 40, axis=0)
    ids2 = np.arange(2000, dtype=np.uint32)
    check(lib().msi_vs_group_upload(vs, np_ptr(ids2), np_ptr(rows2), 2000))
    check(lib().msi_vs_group_search(vs, np_ptr(q[:3]), 3, k, np_ptr(out_d), np_ptr(out_s), np_ptr(cnt)))
    for j in range(3):
        e_ids, e_dist = oracle.vs_topk(rows2, ids2, q[j], k)
        assert out_d[j].tolist() == e_ids.tolist()
    lib().msi_vs_group_destroy(vs)
    lib().msi_group_destroy(g)


def test_per_rank_group_world_of_one(ctx):
    import torch
    uid = (C.c_uint8 * 128)()
    check(lib().msi_group_unique_id(uid))
    g = C.c_void_p()
    check(lib().msi_group_create_rank(ctx.handle, 0, 1, uid, C.byref(g)))
    send = torch.arange(1000, dtype=torch.int32, device="cuda:0")
    recv = torch.zeros(1000, dtype=torch.int32, device="cuda:0")
    torch.cuda.synchronize()
    check(lib().msi_group_allgather(g, C.c_void_p(send.data_ptr()), 4000, C.c_void_p(recv.data_ptr())))
    ctx.synchronize()
    assert torch.equal(send, recv)
    lib().msi_group_destroy(g)


@pytest.mark.parametrize("mode", [0, 1])   # MSI_GROUP_REPLICATE, MSI_GROUP_SHARD_ROWS
def test_in_process_group_over_every_device_of_the_box(oracle, mode):
    """VERDICT r2 #6e: on a box with >= 2 devices (the driver's 8-GPU node) the in-process group spans all of them — RCCL
    with N > 1 ranks (ncclCommInitAll), rows sharded over the devices or replicated with the query batch split, one packed
    all-gather, device merge — and must still equal the oracle bit for bit.  Skipped on the one-device test boxes."""
    n_dev = devices_of_the_box()
    if n_dev < 2:
        pytest.skip("one device: the world-of-one forms above are what this box can run")
    emulated = bool(os.environ.get("MSI_EMU_DEVICES"))
    devs = (C.c_int32 * n_dev)(*range(n_dev))
    g = C.c_void_p()
    check(lib().msi_group_create(devs, n_dev, C.byref(g)))
    assert lib().msi_group_size(g) == n_dev
    vs = C.c_void_p()
    check(lib().msi_vs_group_create(g, 128, 0, mode, C.byref(vs)))
    n = 6_000 if emulated else 50_000    # (the CPU emulation runs every lane of every workgroup in turn)
    rows = synth.make_embeddings(n, 128, seed=13)
    ids = (np.arange(n, dtype=np.uint32) * 3 + 2)
    check(lib().msi_vs_group_upload(vs, np_ptr(ids), np_ptr(rows), n))
    q = synth.make_embeddings(53, 128, seed=14)        # not a multiple of the device count
    k = 20
    out_d = np.zeros((53, k), np.uint32)
    out_s = np.zeros((53, k), np.float32)
    cnt = np.zeros(53, np.uint32)
    check(lib().msi_vs_group_search(vs, np_ptr(q), 53, k, np_ptr(out_d), np_ptr(out_s), np_ptr(cnt)))
    for j in range(53):
        e_ids, e_dist = oracle.vs_topk(rows, ids, q[j], k)
        assert int(cnt[j]) == k and out_d[j].tolist() == e_ids.tolist(), (mode, j)
        assert out_s[j].view(np.uint32).tolist() == e_dist.view(np.uint32).tolist()
    lib().msi_vs_group_destroy(vs)
    lib().msi_group_destroy(g)


def test_per_rank_groups_of_every_device_all_gather(oracle):
    """The one-process-per-GPU form with a world > 1, here as one thread per device: every rank joins with its own context
    and the unique id of rank 0 (msi_group_create_rank blocks until all have), contributes its own packed buffer and must
    receive every rank's in rank order — the exchange bench.py runs under torchrun.  Then the row-sharded search as that
    form does it: each rank searches its rows, the packed lists travel in ONE all-gather, every rank merges and all must
    hold the single-store answer."""
    n_dev = devices_of_the_box()
    if n_dev < 2:
        pytest.skip("one device")
    L = lib()
    uid = (C.c_uint8 * 128)()
    check(L.msi_group_unique_id(uid))
    n, d, k, nq = 4000, 64, 10, 9
    rows = synth.make_embeddings(n, d, seed=21)
    ids = np.arange(n, dtype=np.uint32) * 5 + 1
    q = synth.make_embeddings(nq, d, seed=22)
    per = 2 * nq * k + nq
    merged, errors = [None] * n_dev, []

    def rank_main(r):
        try:
            ctx = ma.Context(r)
            g = C.c_void_p()
            check(L.msi_group_create_rank(ctx.handle, r, n_dev, uid, C.byref(g)))
            assert L.msi_group_size(g) == n_dev
            r0, r1 = n * r // n_dev, n * (r + 1) // n_dev
            store = ma.vector_store.GpuStore(ctx, d)
            store.upload(ids[r0:r1], rows[r0:r1])
            o_ids, o_dist, o_cnt = store.search(q, k)
            send_h = np.concatenate([o_dist.view(np.uint32).reshape(-1), o_ids.reshape(-1), o_cnt.astype(np.uint32)])
            if os.environ.get("MSI_EMU_DEVICES"):     # emulated devices: host memory is device memory
                send, recv = send_h.copy(), np.zeros(per * n_dev, np.uint32)
                send_p, recv_p = np_ptr(send), np_ptr(recv)
            else:
                import torch
                send = torch.from_numpy(send_h.view(np.int32)).to(f"cuda:{r}")
                recv = torch.zeros(per * n_dev, dtype=torch.int32, device=f"cuda:{r}")
                torch.cuda.synchronize(r)
                send_p, recv_p = C.c_void_p(send.data_ptr()), C.c_void_p(recv.data_ptr())
            check(L.msi_group_allgather(g, send_p, per * 4, recv_p))
            ctx.synchronize()
            got = (recv if isinstance(recv, np.ndarray) else recv.cpu().numpy().view(np.uint32)).reshape(n_dev, per)
            all_dist = got[:, :nq * k].copy().view(np.float32).reshape(n_dev, nq, k)
            all_ids = got[:, nq * k:2 * nq * k].reshape(n_dev, nq, k)
            all_cnt = got[:, 2 * nq * k:].reshape(n_dev, nq)
            out = []
            for j in range(nq):
                m_ids, m_dist = np.zeros(k, np.uint32), np.zeros(k, np.float32)
                c = L.msi_merge_topk(np_ptr(np.ascontiguousarray(all_ids[:, j])), np_ptr(np.ascontiguousarray(all_dist[:, j])),
                                     np_ptr(np.ascontiguousarray(all_cnt[:, j])), n_dev, k, k, np_ptr(m_ids), np_ptr(m_dist))
                out.append((m_ids[:c].tolist(), m_dist[:c].view(np.uint32).tolist()))
            merged[r] = out
            L.msi_group_destroy(g)
        except Exception as e:   # noqa: BLE001 - reported by the main thread
            errors.append((r, repr(e)))

    th = [threading.Thread(target=rank_main, args=(r,)) for r in range(n_dev)]
    for t in th:
        t.start()
    for t in th:
        t.join(timeout=600)
    assert not errors, errors
    for j in range(nq):
        e_ids, e_dist = oracle.vs_topk(rows, ids, q[j], k)
        for r in range(n_dev):
            assert merged[r][j] == (e_ids.tolist(), e_dist.view(np.uint32).tolist()), (r, j)
