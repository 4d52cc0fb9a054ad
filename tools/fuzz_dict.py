#!/usr/bin/env python
"""Differential fuzzing of the typo lookup (msi_dict.hip) against the oracle's literal loop (oracle/msi_oracle.c,
compute_derivations.rs:75-168): random alphabets (ASCII / multi-byte), dictionary sizes, word lengths, queries (words of
the dictionary with 0-3 edits, prefixes, random strings), typo budgets, prefix flags and caps.

    python tools/fuzz_dict.py [first_seed] [seconds] [--emulated-kernels]     (without the flag: on the MI355X)
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from meilisearch_amd import _lib
if "--emulated-kernels" in sys.argv:
    sys.argv.remove("--emulated-kernels")
    sys.path.insert(0, os.path.join(ROOT, "tests", "emu"))
    import run_emulated
    _lib._LIB = run_emulated.EmulatedLib(run_emulated.build())
import meilisearch_amd as ma
from meilisearch_amd import synth
from oracle import oracle as O

ALPHABETS = ["ab", "abc", "abcdefgh", "abcdefghijklmnopqrstuvwxyz", "aé", "aбc日", "xyzé日😀", "0123456789ab"]
seed = int(sys.argv[1]) if len(sys.argv) > 1 else 0
budget = float(sys.argv[2]) if len(sys.argv) > 2 else 60
ctx = ma.Context(0)
t_end = time.time() + budget
n = bad = 0
while time.time() < t_end:
    seed += 1
    rng = np.random.default_rng(seed)
    alpha = ALPHABETS[int(rng.integers(len(ALPHABETS)))]
    n_words = int(rng.choice([1, 2, 30, 300, 2000, 6000]))
    lo, hi = (1, 6) if rng.random() < 0.2 else (3, int(rng.choice([8, 14, 40, 90])))   # 90: queries past 64 chars take the banded matcher
    words = set()
    tries = 0
    while len(words) < n_words and tries < 20 * n_words + 100:
        tries += 1
        L = int(rng.integers(lo, hi + 1))
        words.add("".join(alpha[int(i)] for i in rng.integers(0, len(alpha), L)))
    words = sorted(words, key=lambda w: w.encode())

    def edit(w):
        w = list(w)
        for _ in range(int(rng.integers(0, 4))):
            k = int(rng.integers(0, len(w) + 1))
            r = rng.random()
            c = alpha[int(rng.integers(len(alpha)))]
            if r < 0.3 and k < len(w): w[k] = c
            elif r < 0.55: w.insert(k, c)
            elif r < 0.8 and k < len(w) and len(w) > 1: del w[k]
            elif k + 1 < len(w): w[k], w[k + 1] = w[k + 1], w[k]
        return "".join(w)

    queries = []
    for _ in range(int(rng.choice([1, 8, 60]))):
        r = rng.random()
        if r < 0.7:
            w = edit(words[int(rng.integers(len(words)))])
        elif r < 0.85:
            w = words[int(rng.integers(len(words)))]
            w = w[:int(rng.integers(0, len(w) + 1))]
        else:
            w = "".join(alpha[int(i)] for i in rng.integers(0, len(alpha), int(rng.integers(0, 20))))
        queries.append((w, int(rng.integers(1, 3)), bool(rng.random() < 0.4)))
    caps = [(150, 50), (4, 3), (1000, 1000), (1, 1)][int(rng.integers(4))]
    concat, off = synth.flatten_words(words)
    odic = O.Dictionary.from_flat(concat, off)
    gdic = ma.GpuDictionary(ctx, concat=concat, offsets=off)
    got = gdic.lookup(queries, cap_one=caps[0], cap_two=caps[1])
    for (w, b, p), (g1, g2) in zip(queries, got):
        e1, e2 = O.typo_lookup(odic, w, b, p, cap_one=caps[0], cap_two=caps[1])
        n += 1
        if g1.tolist() != e1.tolist() or g2.tolist() != e2.tolist():
            bad += 1
            print("MISMATCH seed", seed, repr(alpha), len(words), repr(w), b, p, caps, g1[:6], e1[:6], g2[:6], e2[:6])
    gdic.close()
print("lookups", n, "bad", bad)
