"""Bindings of the host-side scoring arithmetic of libmsi (a3, a14, a15)."""
import ctypes as C

import numpy as np

from ._lib import lib
from .device import np_ptr


def distribution_shift(mean, sigma, score):
    """DistributionShift::shift — crates/milli/src/vector/distribution.rs:103-130."""
    return float(np.float32(lib().msi_distribution_shift(mean, sigma, score)))


def rank_global_score(pairs):
    """Rank::global_score — crates/milli/src/score_details.rs:517-546."""
    r = np.array([p[0] for p in pairs], dtype=np.uint32)
    m = np.array([p[1] for p in pairs], dtype=np.uint32)
    return float(lib().msi_rank_global_score(np_ptr(r) if r.size else None,
                                             np_ptr(m) if m.size else None, len(pairs)))


def compare_scores(left, left_ratio, right, right_ratio):
    """compare_scores over Score sequences — crates/milli/src/search/hybrid.rs:32-80."""
    l = np.array(left, dtype=np.float64)
    r = np.array(right, dtype=np.float64)
    return int(lib().msi_compare_scores(np_ptr(l) if l.size else None, len(left), left_ratio,
                                        np_ptr(r) if r.size else None, len(right), right_ratio))


def vector_sort(docids, dist, offset=0, limit=20, distribution=None):
    """VectorSort as the only ranking rule — search/new/vector_sort.rs:58-168."""
    d = np.ascontiguousarray(docids, dtype=np.uint32)
    s = np.ascontiguousarray(dist, dtype=np.float32)
    out_d = np.zeros(max(limit, 1), dtype=np.uint32)
    out_s = np.zeros(max(limit, 1), dtype=np.float32)
    mean, sigma = distribution if distribution else (0.0, 1.0)
    n = lib().msi_vector_sort(np_ptr(d) if d.size else None, np_ptr(s) if s.size else None, d.size,
                              1 if distribution else 0, mean, sigma, offset, limit, np_ptr(out_d), np_ptr(out_s))
    return out_d[:n].copy(), out_s[:n].copy()


def _flatten(score_lists):
    off = np.zeros(len(score_lists) + 1, dtype=np.uint32)
    if score_lists:
        np.cumsum([len(x) for x in score_lists], out=off[1:])
    flat = np.array([v for x in score_lists for v in x], dtype=np.float64)
    if flat.size == 0:
        flat = np.zeros(1, dtype=np.float64)
    return flat, off


def hybrid_merge(vector_hits, keyword_hits, semantic_ratio, offset=0, limit=20):
    """ScoreWithRatioResult::merge — search/hybrid.rs:102-235.  hits: [(docid, [score values])].
    Returns ([(docid, is_semantic)], semantic_hit_count)."""
    vd = np.array([h[0] for h in vector_hits], dtype=np.uint32)
    kd = np.array([h[0] for h in keyword_hits], dtype=np.uint32)
    vs, vo = _flatten([h[1] for h in vector_hits])
    ks, ko = _flatten([h[1] for h in keyword_hits])
    out_d = np.zeros(max(limit, 1), dtype=np.uint32)
    out_s = np.zeros(max(limit, 1), dtype=np.uint8)
    cnt = C.c_uint32(0)
    n = lib().msi_hybrid_merge(np_ptr(vd) if vd.size else None, np_ptr(vs), np_ptr(vo), vd.size,
                               np.float32(semantic_ratio), np_ptr(kd) if kd.size else None, np_ptr(ks), np_ptr(ko),
                               kd.size, np.float32(1.0) - np.float32(semantic_ratio), offset, limit, np_ptr(out_d),
                               np_ptr(out_s), C.byref(cnt))
    return [(int(out_d[i]), bool(out_s[i])) for i in range(n)], int(cnt.value)


def results_good_enough(keyword_global_scores, limit_plus_offset, semantic_ratio):
    s = np.array(keyword_global_scores, dtype=np.float64)
    return bool(lib().msi_results_good_enough(np_ptr(s) if s.size else None, s.size, limit_plus_offset,
                                              semantic_ratio))


def hybrid_merge_batch(v_docids, v_dist, v_counts, k_docids, k_words, k_typos, k_maxt, k_counts, n_terms,
                       semantic_ratio, offset=0, limit=20):
    """msi_hybrid_merge_batch over [Q, stride] arrays -> (docids [Q, limit], is_semantic, counts, semantic_hits)."""
    from ._lib import check
    q = v_docids.shape[0]
    arrs = [np.ascontiguousarray(a, dtype=t) for a, t in
            ((v_docids, np.uint32), (v_dist, np.float32), (v_counts, np.uint32), (k_docids, np.uint32),
             (k_words, np.uint32), (k_typos, np.uint32), (k_maxt, np.uint32), (k_counts, np.uint32),
             (n_terms, np.uint32))]
    out_d = np.zeros((q, max(limit, 1)), dtype=np.uint32)
    out_s = np.zeros((q, max(limit, 1)), dtype=np.uint8)
    out_c = np.zeros(q, dtype=np.uint32)
    out_h = np.zeros(q, dtype=np.uint32)
    check(lib().msi_hybrid_merge_batch(np_ptr(arrs[0]), np_ptr(arrs[1]), np_ptr(arrs[2]), arrs[0].shape[1],
                                       np_ptr(arrs[3]), np_ptr(arrs[4]), np_ptr(arrs[5]), np_ptr(arrs[6]),
                                       np_ptr(arrs[7]), arrs[3].shape[1], np_ptr(arrs[8]), q,
                                       np.float32(semantic_ratio), offset, limit, np_ptr(out_d), np_ptr(out_s),
                                       np_ptr(out_c), np_ptr(out_h)))
    return out_d, out_s, out_c, out_h


def inject_pins(pins, organic, offset=0, limit=20):
    """inject_pins / merge_positioned_hits_into_page — search/new/bucket_sort.rs:345-377, search/mod.rs:579-625.
    pins: [(position, docid)] in resolve_pins' order; organic: [(docid, [(kind, a, b), ...])] — the bucket sort's hits for
    from = 0, length = offset + limit.  Returns the page: [(docid, [(kind, a, b), ...])] (a pin: [(MSI_SCORE_PIN, position, 0)])."""
    MAXD = 16
    p = np.array([(a, b) for a, b in pins], dtype=np.uint32).reshape(-1, 2)
    d = np.array([h[0] for h in organic], dtype=np.uint32)
    sc = np.zeros((max(1, len(organic)), MAXD, 3), dtype=np.uint32)
    ns = np.zeros(max(1, len(organic)), dtype=np.uint32)
    for i, (_, det) in enumerate(organic):
        ns[i] = len(det)
        for j, t in enumerate(det):
            sc[i, j] = t
    out_d = np.zeros(max(1, limit), dtype=np.uint32)
    out_s = np.zeros((max(1, limit), MAXD, 3), dtype=np.uint32)
    out_n = np.zeros(max(1, limit), dtype=np.uint32)
    n = lib().msi_inject_pins(np_ptr(p) if p.size else None, len(pins), offset, limit, np_ptr(d) if d.size else None,
                              np_ptr(sc), np_ptr(ns), len(organic), np_ptr(out_d), np_ptr(out_s), np_ptr(out_n))
    return [(int(out_d[i]), [tuple(int(x) for x in out_s[i, j]) for j in range(int(out_n[i]))]) for i in range(n)]
