#!/usr/bin/env python
"""Differential fuzzing of the vector k-NN (msi_vs.hip: bf16 MFMA candidate scan + exactness proof + f32 rescoring +
exhaustive fallback) against the oracle's sequential-f32 scan (oracle/msi_oracle.c vs_topk): row counts, dimensions and k
at random, and data chosen to sit where a narrow candidate pass goes wrong — near-duplicate clusters (distances that
differ in the last f32 bits), exact duplicates (docid tie order), huge and tiny magnitudes, zero rows, quantised
components, queries equal to rows — with and without candidate filters.  Bar: same docids, same order, the same f32 bits.

    python tools/fuzz_vs.py [first_seed] [seconds] [--emulated-kernels]     (without the flag: on the MI355X)
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

f32 = np.float32
seed = int(sys.argv[1]) if len(sys.argv) > 1 else 0
budget = float(sys.argv[2]) if len(sys.argv) > 2 else 60
ctx = ma.Context(0)
t_end = time.time() + budget
n_q = bad = inexact = 0
while time.time() < t_end:
    seed += 1
    rng = np.random.default_rng(seed)
    n = int(rng.choice([1, 3, 17, 64, 200, 700] + ([] if EMU else [5000, 40000])))
    dim = int(rng.choice([1, 2, 3, 7, 16, 31, 64, 96, 130] + ([] if EMU else [384, 768, 1024])))
    k = int(rng.choice([1, 2, 5, 20, 100]))
    kind = int(rng.integers(0, 6))
    if kind == 0:      # plain
        rows = rng.standard_normal((n, dim))
    elif kind == 1:    # clusters of near-duplicates
        c = rng.standard_normal((max(1, n // 16), dim))
        rows = c[rng.integers(0, c.shape[0], n)] * (1.0 + rng.standard_normal((n, 1)) * 1e-7) + rng.standard_normal((n, dim)) * rng.choice([0.0, 1e-7, 1e-4])
    elif kind == 2:    # exact duplicates and scaled copies (cosine ties)
        c = rng.standard_normal((max(1, n // 8), dim))
        rows = c[rng.integers(0, c.shape[0], n)] * rng.choice([1.0, 2.0, 0.5, 1024.0], size=(n, 1))
    elif kind == 3:    # wild magnitudes
        rows = rng.standard_normal((n, dim)) * np.exp(rng.uniform(-20, 20, size=(n, 1))) * np.exp(rng.uniform(-3, 3, size=(1, dim)))
    elif kind == 4:    # quantised components, many zeros
        rows = rng.integers(-2, 3, size=(n, dim)).astype(np.float64)
    else:              # one dominant component
        rows = rng.standard_normal((n, dim)) * 1e-3
        rows[:, int(rng.integers(dim))] += rng.choice([-1.0, 1.0], size=n)
    rows = rows.astype(f32)
    if rng.random() < 0.3:
        rows[rng.integers(0, n, max(1, n // 10))] = 0
    ids = np.sort(rng.choice(np.arange(4 * n + 10, dtype=np.uint32), n, replace=False)).astype(np.uint32)
    nq = int(rng.choice([1, 3, 7]))
    qs = rng.standard_normal((nq, dim)).astype(f32)
    for j in range(nq):
        r = rng.random()
        if r < 0.4:
            qs[j] = rows[int(rng.integers(n))]
        elif r < 0.6:
            qs[j] = rows[int(rng.integers(n))] * f32(rng.choice([3.0, 1e-3])) + (rng.standard_normal(dim) * 1e-6).astype(f32)
        elif r < 0.65:
            qs[j] = 0
    fb, nb = None, 0
    if rng.random() < 0.4:
        keep = ids[rng.random(n) < rng.choice([0.05, 0.5, 0.95])]
        fb, nb = ma.dense_filter(keep.tolist())
    st = ma.GpuStore(ctx, dim)
    st.upload(ids, rows)
    d, s, c = st.search(qs, k, fb, nb)
    inexact += int(st.stats()["exhaustive_reruns"])
    for j in range(nq):
        e_ids, e_dist = O.vs_topk(rows, ids, qs[j], k, fb, nb)
        m = int(c[j])
        n_q += 1
        if m != e_ids.size or d[j, :m].tolist() != e_ids.tolist() or s[j, :m].view(np.uint32).tolist() != e_dist.view(np.uint32).tolist():
            bad += 1
            print("MISMATCH seed", seed, "n", n, "dim", dim, "k", k, "kind", kind, "query", j, "filter", nb, d[j, :m][:6], e_ids[:6], s[j, :m][:4], e_dist[:4])
    st.close()
print("queries", n_q, "bad", bad, "exhaustive re-runs", inexact)
