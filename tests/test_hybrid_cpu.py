"""Host-side tail of semantic / hybrid search (no GPU): msi_vector_sort,
msi_hybrid_merge, msi_results_good_enough against literal Python restatements of
crates/milli/src/search/new/vector_sort.rs:58-168 and search/hybrid.rs:32-235,367-386,
plus the reference's own literals where they exist."""
import numpy as np

f32 = np.float32


def py_compare_scores(l, lr, r, rr):          # hybrid.rs:32-80 restricted to Score values
    i = 0
    while True:
        hl, hr = i < len(l), i < len(r)
        if not hl and not hr:
            return 0
        if not hl:
            return -1
        if not hr:
            return 1
        a, b = l[i] * float(f32(lr)), r[i] * float(f32(rr))
        i += 1
        if abs(a - b) <= np.finfo(np.float64).eps:
            continue
        return -1 if a < b else 1


def py_merge(vh, kh, ratio, off, lim):        # hybrid.rs:102-235 (no pins / distinct)
    out, seen, iv, ik = [], set(), 0, 0
    merged = []
    while iv < len(vh) or ik < len(kh):
        if iv >= len(vh):
            take_v = False
        elif ik >= len(kh):
            take_v = True
        else:
            take_v = py_compare_scores(vh[iv][1], ratio, kh[ik][1], f32(1.0) - f32(ratio)) >= 0
        if take_v:
            merged.append((vh[iv][0], True))
            iv += 1
        else:
            merged.append((kh[ik][0], False))
            ik += 1
    for d, s in merged:
        if d in seen:
            continue
        seen.add(d)
        out.append((d, s))
    page = out[off:off + lim]
    return page, sum(1 for _, s in page if s)


def test_hybrid_merge_vs_restatement():
    import meilisearch_amd as ma
    rng = np.random.default_rng(3)
    for trial in range(200):
        nv, nk = int(rng.integers(0, 12)), int(rng.integers(0, 12))
        vd = rng.choice(30, nv, replace=False).tolist()
        kd = rng.choice(30, nk, replace=False).tolist()
        vs = sorted(rng.random(nv).round(3).tolist(), reverse=True)
        ks = sorted(rng.choice([0.25, 0.5, 0.75, 1.0, 0.9848484848484848, 0.9242424242424242], nk).tolist(), reverse=True)
        vh = [(d, [s]) for d, s in zip(vd, vs)]
        kh = [(d, [s] if rng.random() < 0.8 else [s, 0.5]) for d, s in zip(kd, ks)]
        ratio = float(rng.choice([0.0, 0.2, 0.5, 0.8, 1.0]))
        off, lim = int(rng.integers(0, 4)), int(rng.integers(1, 10))
        assert ma.scoring.hybrid_merge(vh, kh, ratio, off, lim) == py_merge(vh, kh, ratio, off, lim), trial


def test_hybrid_ratio_extremes():
    import meilisearch_amd as ma
    # ratio 1.0: keyword scores are weighted by 0 -> every vector hit first
    # (crates/meilisearch/tests/search/hybrid.rs "semanticRatio": 1.0 cases)
    vh = [(1, [0.9]), (2, [0.5])]
    kh = [(3, [1.0]), (1, [0.8])]
    got, sem = ma.scoring.hybrid_merge(vh, kh, 1.0, 0, 10)
    assert [d for d, _ in got] == [1, 2, 3] and sem == 2
    got, sem = ma.scoring.hybrid_merge(vh, kh, 0.0, 0, 10)
    assert [d for d, _ in got][:2] == [3, 1] and sem == 1   # doc 1 reached through the keyword list first


def test_vector_sort():
    import meilisearch_amd as ma
    # cutoff.rs:507-626 literals: IDs [2,0,3,1], similarities 1.0, 0.5, 0.5, 0.0
    d, s = ma.scoring.vector_sort([2, 0, 3, 1], f32([0.0, 0.5, 0.5, 1.0]), 0, 10)
    assert d.tolist() == [2, 0, 3, 1] and s.tolist() == [1.0, 0.5, 0.5, 0.0]
    # several embeddings per document: first (smallest distance) occurrence wins
    d, s = ma.scoring.vector_sort([5, 7, 5, 9, 7], f32([0.1, 0.2, 0.3, 0.4, 0.5]), 1, 2)
    assert d.tolist() == [7, 9]
    # DistributionShift on top (hybrid.rs:540-568 literals)
    dist = f32(1.0) - f32([0.990290343761444, 0.974341630935669, 0.9472135901451112])
    d, s = ma.scoring.vector_sort([1, 2, 3], dist, 0, 3, distribution=(0.998, 0.01))
    assert s.tolist() == [f32(0.19161224365234375), f32(1.1920928955078125e-07), f32(1.1920928955078125e-07)]


def test_results_good_enough():
    import meilisearch_amd as ma
    assert ma.scoring.results_good_enough([0.95, 0.91], 2, 0.5)
    assert not ma.scoring.results_good_enough([0.95, 0.89], 2, 0.5)      # 0.89 * 0.5 < 0.45
    assert not ma.scoring.results_good_enough([0.95], 2, 0.5)            # not enough hits
    assert not ma.scoring.results_good_enough([1.0, 1.0], 2, 0.9)


def test_hybrid_merge_batch_equals_per_query():
    import meilisearch_amd as ma
    rng = np.random.default_rng(9)
    Q, k = 11, 8
    v_ids = np.stack([rng.choice(50, k, replace=False) for _ in range(Q)]).astype(np.uint32)
    v_dist = np.sort(rng.random((Q, k)).astype(np.float32) * 0.5, axis=1)
    v_cnt = rng.integers(0, k + 1, Q).astype(np.uint32)
    k_ids = np.stack([rng.choice(50, k, replace=False) for _ in range(Q)]).astype(np.uint32)
    n_terms = rng.integers(1, 5, Q).astype(np.uint32)
    k_words = np.sort(np.stack([rng.integers(1, n_terms[q] + 1, k) for q in range(Q)]), axis=1)[:, ::-1].astype(np.uint32)
    k_maxt = np.full((Q, k), 3, dtype=np.uint32)
    k_typos = rng.integers(0, 4, (Q, k)).astype(np.uint32)
    k_cnt = rng.integers(0, k + 1, Q).astype(np.uint32)
    for ratio in (0.0, 0.5, 0.9):
        d, s, c, h = ma.scoring.hybrid_merge_batch(v_ids, v_dist, v_cnt, k_ids, k_words, k_typos, k_maxt, k_cnt,
                                                    n_terms, ratio, 1, 6)
        for q in range(Q):
            vh = [(int(v_ids[q, i]), [float(f32(1.0) - v_dist[q, i])]) for i in range(int(v_cnt[q]))]
            kh = [(int(k_ids[q, i]), [ma.scoring.rank_global_score([(int(k_words[q, i]), int(n_terms[q])),
                                                                      (int(k_maxt[q, i] + 1 - k_typos[q, i]), int(k_maxt[q, i] + 1))])])
                  for i in range(int(k_cnt[q]))]
            exp, sem = ma.scoring.hybrid_merge(vh, kh, ratio, 1, 6)
            assert c[q] == len(exp) and h[q] == sem
            assert [(int(d[q, i]), bool(s[q, i])) for i in range(int(c[q]))] == exp


def test_score_details_global_score_matches_reference_literals():
    """cutoff.rs:330-470: Words 3/3 then Typo k of 3 -> 1.0000 0.9167 0.8333 0.7500 (Rank::merge); and the
    oracle's ScoreDetails::global_score on every variant."""
    from meilisearch_amd import ranking as R
    from oracle import ranking_oracle as RO
    for typo, want in ((0, "1.0000"), (1, "0.9167"), (2, "0.8333"), (3, "0.7500")):
        got = R.score_details_global_score([("Words", 3, 3), ("Typo", typo, 3)])
        assert f"{got:.4f}" == want
    details = [("Words", 2, 3), ("Typo", 1, 4), ("Proximity", 5, 8), ("Fid", 3, 7), ("Position", 11, 21),
               ("ExactAttribute", 2, 3), ("ExactWords", 1, 3)]
    want = RO.global_score([("Words", 2, 3), ("Typo", 1, 4), ("Proximity", 5, 8), ("Fid", 3, 7), ("Position", 11, 21),
                            ("ExactAttribute", "MatchesStart"), ("ExactWords", 1, 3)])
    assert R.score_details_global_score(details) == want
