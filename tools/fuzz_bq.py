#!/usr/bin/env python
"""Differential fuzzing of the binary-quantised k-NN (msi_bq.hip: one bounded sweep behind a sample's k-th distance, the
three-sweep exhaustive form as fallback) against the oracle (orc_bq_topk): random sizes, dimensions (not multiples of 32 /
64 included), k, filters, clustered and low-entropy codes (thousands of ties: the docid rule decides), skewed data that
makes the sample a poor predictor.  MSI_BQ_SAMPLE_ROWS is drawn small so that the one-sweep form runs at these sizes.

    python tools/fuzz_bq.py [first_seed] [seconds] [--emulated-kernels]     (without the flag: on the MI355X)
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from meilisearch_amd import _lib
EMU = "--emulated-kernels" in sys.argv
if EMU:
    sys.argv.remove("--emulated-kernels")
    sys.path.insert(0, os.path.join(ROOT, "tests", "emu"))
    import run_emulated
    _lib._LIB = run_emulated.EmulatedLib(run_emulated.build())
import meilisearch_amd as ma
from oracle import oracle as O

seed = int(sys.argv[1]) if len(sys.argv) > 1 else 0
budget = float(sys.argv[2]) if len(sys.argv) > 2 else 60
ctx = ma.Context(0)
t_end = time.time() + budget
n_q = bad = 0
while time.time() < t_end:
    seed += 1
    rng = np.random.default_rng(seed)
    os.environ["MSI_BQ_SAMPLE_ROWS"] = str(int(rng.choice([64, 256, 1024])))
    n = int(rng.choice([1, 5, 70, 700, 3000] + ([] if EMU else [50000])))
    dim = int(rng.choice([1, 3, 8, 31, 64, 65, 130, 256] + ([] if EMU else [768, 1024])))
    k = int(rng.choice([1, 5, 20, 200]))
    kind = int(rng.integers(0, 4))
    if kind == 0:
        rows = rng.standard_normal((n, dim))
    elif kind == 1:    # few distinct codes
        c = rng.standard_normal((max(1, n // 50), dim))
        rows = c[rng.integers(0, c.shape[0], n)]
    elif kind == 2:    # the first rows (the sample) look nothing like the rest
        rows = rng.standard_normal((n, dim))
        rows[: n // 3] = np.abs(rows[: n // 3])
    else:              # zeros and ones
        rows = rng.integers(-1, 2, size=(n, dim)).astype(np.float64)
    rows = rows.astype(np.float32)
    ids = np.sort(rng.choice(np.arange(3 * n + 10, dtype=np.uint32), n, replace=False)).astype(np.uint32)
    nq = int(rng.choice([1, 5, 33]))
    qs = rng.standard_normal((nq, dim)).astype(np.float32)
    if rng.random() < 0.5:
        qs[0] = rows[int(rng.integers(n))]
    flt = ()
    if rng.random() < 0.4:
        keep = ids[rng.random(n) < rng.choice([0.02, 0.5, 0.98])]
        flt = ma.dense_filter(keep.tolist(), int(ids.max()) + 1)
    st = ma.GpuBqStore(ctx, dim)
    st.upload(ids, rows)
    d, s, c = st.search(qs, k, *flt)
    for j in range(nq):
        e_ids, e_dist = O.bq_topk(rows, ids, qs[j], k, *flt)
        m = int(c[j])
        n_q += 1
        if m != e_ids.size or d[j, :m].tolist() != e_ids.tolist() or s[j, :m].view(np.uint32).tolist() != e_dist.view(np.uint32).tolist():
            bad += 1
            print("MISMATCH seed", seed, "n", n, "dim", dim, "k", k, "kind", kind, "query", j, "filter", bool(flt), d[j, :m][:6], e_ids[:6])
    st.close()
print("queries", n_q, "bad", bad)
