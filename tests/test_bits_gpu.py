"""S3 parity: dense docid-set algebra and CboRoaringBitmap decoding against numpy."""
import struct

import numpy as np
import pytest

import meilisearch_amd as ma
from meilisearch_amd import bits as B

pytestmark = pytest.mark.gpu


def roaring_serialize(ids, use_runs=False):
    """Standard portable Roaring serialisation (RoaringFormatSpec), as written by
    roaring-rs' serialize_into: cookie 12346, containers array (<=4096) / bitmap;
    with use_runs: cookie 12347 and run containers."""
    ids = np.unique(np.asarray(ids, dtype=np.uint32))
    keys = np.unique(ids >> 16)
    conts = [(int(k), (ids[(ids >> 16) == k] & 0xFFFF).astype(np.uint16)) for k in keys]
    n = len(conts)
    out = bytearray()
    if use_runs:
        out += struct.pack("<I", 12347 | ((n - 1) << 16))
        out += bytes([0xFF] * ((n + 7) // 8))
    else:
        out += struct.pack("<II", 12346, n)
    for k, v in conts:
        out += struct.pack("<HH", k, len(v) - 1)
    if not use_runs or n >= 4:
        out += b"\0" * (4 * n)   # offsets (ignored by the decoder, recomputed)
    for k, v in conts:
        if use_runs:
            runs = []
            start = prev = int(v[0])
            for x in v[1:]:
                x = int(x)
                if x != prev + 1:
                    runs.append((start, prev - start))
                    start = x
                prev = x
            runs.append((start, prev - start))
            out += struct.pack("<H", len(runs))
            for s, l in runs:
                out += struct.pack("<HH", s, l)
        elif len(v) <= 4096:
            out += v.astype("<u2").tobytes()
        else:
            words = np.zeros(1024, dtype=np.uint64)
            np.bitwise_or.at(words, (v >> 6).astype(np.int64), np.uint64(1) << (v & 63).astype(np.uint64))
            out += words.astype("<u8").tobytes()
    return bytes(out)


def cbo_serialize(ids, use_runs=False):
    # CboRoaringBitmapCodec::serialize_into_writer, cbo_roaring_bitmap_codec.rs:33-51
    ids = np.unique(np.asarray(ids, dtype=np.uint32))
    if ids.size <= 7:
        return ids.astype("=u4").tobytes()
    return roaring_serialize(ids, use_runs)


def test_cbo_threshold_literals():
    # cbo_roaring_bitmap_codec.rs:186-220: <= 7 integers are raw u32s
    assert len(cbo_serialize(range(7))) == 28
    assert len(cbo_serialize(range(8))) > 28


@pytest.mark.parametrize("n_docs", [1, 63, 64, 1000, 200003])
def test_algebra_vs_numpy(ctx, n_docs):
    rng = np.random.default_rng(n_docs)
    pool = ma.BitsPool(ctx, n_docs, 8)
    sets = [np.unique(rng.integers(0, n_docs, size=max(1, n_docs // d))).astype(np.uint32) for d in (2, 3, 50)]
    for i, s in enumerate(sets):
        pool.set_from_docids(i, s)
        assert pool.to_docids(i).tolist() == s.tolist()
        assert pool.count(i) == s.size
    a, b, c = (set(s.tolist()) for s in sets)
    pool.op(3, 0, 1, B.AND)
    assert set(pool.to_docids(3).tolist()) == a & b
    pool.op(3, 0, 1, B.OR)
    assert set(pool.to_docids(3).tolist()) == a | b
    pool.op(3, 0, 1, B.ANDNOT)
    assert set(pool.to_docids(3).tolist()) == a - b
    pool.op(3, 0, 1, B.XOR)
    assert set(pool.to_docids(3).tolist()) == a ^ b
    pool.union_many_and(4, [0, 1, 2], universe=B.NO_UNIVERSE)
    assert set(pool.to_docids(4).tolist()) == a | b | c
    pool.union_many_and(4, [1, 2], universe=0)
    assert set(pool.to_docids(4).tolist()) == (b | c) & a
    pool.fill(5, True)
    assert pool.count(5) == n_docs
    pool.fill(5, False)
    assert pool.count(5) == 0
    for k in (0, 1, 5, 10 ** 6):
        assert pool.first_k(0, k).tolist() == sets[0][:k].tolist()


def test_cbo_decode(ctx):
    n_docs = 400000
    rng = np.random.default_rng(9)
    pool = ma.BitsPool(ctx, n_docs, 2)
    cases = [[], [5], [1, 2, 3, 70000, 399999], list(range(7)), list(range(8)),
             rng.integers(0, n_docs, 3000).tolist(),                     # array containers
             rng.integers(0, n_docs, 300000).tolist(),                   # bitmap containers
             list(range(65530, 65600)) + list(range(200000, 210000))]    # runs across a container edge
    for ids in cases:
        exp = np.unique(np.asarray(ids, dtype=np.uint32))
        pool.set_from_cbo(0, cbo_serialize(ids))
        assert pool.to_docids(0).tolist() == exp.tolist()
        runs = cbo_serialize(ids, use_runs=True)
        # the codec tells the two encodings apart by LENGTH (cbo_roaring_bitmap_codec.rs:53-58):
        # a Roaring body of <= 28 bytes cannot be a CboRoaringBitmap value
        if exp.size > 7 and len(runs) > 28:
            pool.set_from_cbo(1, runs)
            assert pool.to_docids(1).tolist() == exp.tolist()
    with pytest.raises(ma.MsiError):
        pool.set_from_cbo(0, b"\x01" * 40)


def test_bits_as_vector_filter(ctx, oracle):
    # a device-resident candidate set feeds the vector scan directly (no PCIe hop)
    import torch
    from meilisearch_amd import synth
    n, dim = 5000, 32
    rows = synth.make_embeddings(n, dim, seed=3)
    ids = np.arange(n, dtype=np.uint32)
    st = ma.GpuStore(ctx, dim)
    st.upload(ids, rows)
    pool = ma.BitsPool(ctx, n, 1)
    allowed = np.arange(0, n, 7, dtype=np.uint32)
    pool.set_from_docids(0, allowed)
    dev = torch.device("cuda", ctx.device)
    q = synth.make_embeddings(3, dim, seed=4)
    q_t = torch.from_numpy(q).to(dev)
    out_d = torch.zeros((3, 10), dtype=torch.int32, device=dev)
    out_s = torch.zeros((3, 10), dtype=torch.float32, device=dev)
    out_c = torch.zeros(3, dtype=torch.int32, device=dev)
    inex = torch.zeros(3, dtype=torch.int32, device=dev)
    torch.cuda.synchronize()
    import ctypes
    st.search_device(q_t, 10, out_d, out_s, out_c, inex, filter_ptr=ctypes.c_void_p(pool.device_ptr(0)), filter_nbits=n)
    ctx.synchronize()
    fb, nb = ma.dense_filter(allowed, nbits=n)
    for j in range(3):
        e_ids, e_dist = oracle.vs_topk(rows, ids, q[j], 10, fb, nb)
        assert out_d[j].cpu().numpy().astype(np.uint32).tolist() == e_ids.tolist()
        assert inex[j].item() == 0


def test_set_from_docid_lists_device():
    """Device-resident docid lists (a vector search's top-k) -> one slot per list, in two launches."""
    import torch
    import meilisearch_amd as ma
    ctx = ma.Context(0)
    n_docs, n_lists, stride = 100_000, 7, 300
    pool = ma.BitsPool(ctx, n_docs, 2 + 3 * n_lists)
    rng = np.random.default_rng(3)
    ids = rng.integers(0, n_docs, (n_lists, stride), dtype=np.int64).astype(np.uint32)
    ids[2, 5] = 0xFFFFFFFF                       # a padding entry is ignored
    counts = np.array([300, 0, 17, 299, 1, 150, 300], dtype=np.int32)
    for s_ in range(2 + 3 * n_lists):
        pool.fill(s_, True)                       # stale content must be cleared
    dev = torch.device("cuda", 0)
    ids_t = torch.from_numpy(ids.view(np.int32)).to(dev)
    cnt_t = torch.from_numpy(counts).to(dev)
    torch.cuda.synchronize()
    pool.set_from_docid_lists_device(2, 3, ids_t, cnt_t)
    for i in range(n_lists):
        want = sorted(set(int(x) for x in ids[i, :counts[i]] if x < n_docs))
        assert pool.to_docids(2 + 3 * i).tolist() == want
        assert pool.count(2 + 3 * i + 1) == n_docs   # neighbours untouched


def test_decode_batch_larger_than_the_staging_ring(ctx):
    """One posting of 1100 bitmap containers is 9 MB of serialised bytes plus 17.6 KB of descriptors — more than the 8 MB
    the pinned staging ring starts with: the direct decode path takes ONE ring block for both (growing the ring), so the
    descriptors can neither overlap the not-yet-copied bytes nor dangle (msi_bits_decode_batch)."""
    n_cont = 1100
    rng = np.random.default_rng(77)
    words = rng.integers(0, 2 ** 63, size=(n_cont, 1024), dtype=np.uint64) | (rng.integers(0, 2, size=(n_cont, 1024), dtype=np.uint64) << np.uint64(63))
    card = np.array([int(np.unpackbits(w.view(np.uint8)).sum()) for w in words])
    assert (card > 4096).all()
    out = bytearray(struct.pack("<II", 12346, n_cont))
    for k in range(n_cont):
        out += struct.pack("<HH", k, (card[k] - 1) & 0xFFFF)
    out += b"\0" * (4 * n_cont)
    out += words.astype("<u8").tobytes()
    assert len(out) > 8 << 20
    n_docs = n_cont * 65536
    pool = ma.BitsPool(ctx, n_docs, 2)
    pool.set_from_cbo(0, bytes(out))
    got = pool.read_words(0)
    assert np.array_equal(got[:n_cont * 1024], words.reshape(-1))
    assert pool.count(0) == int(card.sum())
    pool.close()
