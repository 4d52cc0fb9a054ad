"""msi_vs_update: a store brought up to date by a sequence of committed updates — removals,
additions, replacements, empty updates, down to an empty store and back — answers every search exactly like a store
freshly uploaded with the resulting rows (docids and f32 distances bit for bit, which in turn equal the oracle's),
for f32 and bf16 row storage, ragged tile counts and dimensions that are not multiples of the tile width."""
import numpy as np
import pytest

import meilisearch_amd as ma
from meilisearch_amd import synth

pytestmark = pytest.mark.gpu


def check_same(oracle, st, model, dim, k, seed, storage):
    ids = np.array(sorted(model), dtype=np.uint32)
    rows = np.stack([model[d] for d in ids.tolist()]) if len(ids) else np.zeros((0, dim), np.float32)
    assert len(st) == len(ids)
    qs = synth.make_embeddings(5, dim, seed=seed)
    d, s, c = st.search(qs, k)
    fresh = ma.GpuStore(st.ctx, dim, storage=storage)
    fresh.upload(ids, rows)
    d2, s2, c2 = fresh.search(qs, k)
    ref_rows = synth.round_to_bf16(rows) if storage == "bf16" else rows
    for j in range(len(qs)):
        assert c[j] == c2[j] == min(k, len(ids))
        assert d[j][:c[j]].tolist() == d2[j][:c[j]].tolist()
        assert s[j][:c[j]].view(np.uint32).tolist() == s2[j][:c[j]].view(np.uint32).tolist()
        if len(ids):
            e_ids, e_dist = oracle.vs_topk(ref_rows, ids, qs[j], k)
            assert d[j][:c[j]].tolist() == e_ids.tolist()
            assert s[j][:c[j]].view(np.uint32).tolist() == e_dist.view(np.uint32).tolist()
    if len(ids):
        probe = int(ids[len(ids) // 2])
        assert (st.get_vector(probe) == ref_rows[len(ids) // 2]).all()
    fresh.close()


@pytest.mark.parametrize("storage", ["f32", "bf16"])
@pytest.mark.parametrize("n,dim", [(37, 5), (500, 96), (2100, 384)])
def test_updates_equal_a_fresh_upload(ctx, oracle, n, dim, storage):
    rng = np.random.default_rng(n + dim)
    universe = np.arange(0, 4 * n, dtype=np.uint32)
    ids = np.sort(rng.choice(universe, n, replace=False)).astype(np.uint32)
    rows = synth.make_embeddings(n, dim, seed=n)
    model = {int(d): rows[i] for i, d in enumerate(ids)}
    st = ma.GpuStore(ctx, dim, storage=storage)
    st.upload(ids, rows)
    k = 12
    for step in range(6):
        have = np.array(sorted(model), dtype=np.uint32)
        n_rm = int(rng.integers(0, max(2, len(have) // 3)))
        rm = np.sort(rng.choice(have, min(n_rm, len(have)), replace=False)) if len(have) else np.zeros(0, np.uint32)
        rm = np.union1d(rm, rng.choice(universe, 3)).astype(np.uint32)          # unknown docids are ignored
        n_add = int(rng.integers(0, max(2, n // 3)))
        ad = np.sort(rng.choice(universe, n_add, replace=False)).astype(np.uint32)   # some replace existing rows
        new_rows = synth.make_embeddings(max(n_add, 1), dim, seed=1000 * step + n)[:n_add]
        st.update(rm, ad, new_rows)
        for d in rm.tolist():
            if d not in set(ad.tolist()):
                model.pop(d, None)
        for i, d in enumerate(ad.tolist()):
            model[d] = new_rows[i]
        check_same(oracle, st, model, dim, k, seed=step, storage=storage)
    st.update()                                                                  # nothing changes
    check_same(oracle, st, model, dim, k, seed=77, storage=storage)
    st.update(np.array(sorted(model), dtype=np.uint32))                          # everything leaves
    model.clear()
    check_same(oracle, st, model, dim, k, seed=78, storage=storage)
    back = synth.make_embeddings(20, dim, seed=5)
    st.update((), np.arange(20, dtype=np.uint32) * 3, back)                      # and an empty store takes rows again
    model.update({3 * i: back[i] for i in range(20)})
    check_same(oracle, st, model, dim, k, seed=79, storage=storage)
    with pytest.raises(ma.MsiError):
        st.update([5, 4])
    with pytest.raises(ma.MsiError):
        st.update((), [9, 9], back[:2])
    st.close()


def test_update_of_a_store_that_was_never_uploaded(ctx, oracle):
    dim = 40
    st = ma.GpuStore(ctx, dim)
    st.update([1, 2, 3])                                     # removing from nothing
    assert len(st) == 0
    rows = synth.make_embeddings(50, dim, seed=3)
    st.update((), np.arange(50, dtype=np.uint32) * 2, rows)
    check_same(oracle, st, {2 * i: rows[i] for i in range(50)}, dim, 7, seed=1, storage="f32")
    st.close()
