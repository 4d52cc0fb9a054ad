"""Binary-quantised k-NN: what IS pinned when the distance value is not.

The reference's tests hold no `_rankingScore` produced by a `binaryQuantized: true` embedder (searched: every file under
crates/meilisearch/tests/ that mentions binaryQuantized — settings/vectors.rs, vector/binary_quantized.rs,
vector/settings.rs — snapshots `_vectors` read-backs and settings only), and the distance lives in arroy / hannoy, which
are not under /root/reference.  The two candidate definitions of the crates:

  * hannoy `Hamming`: the number of differing sign bits h (possibly divided by the dimension);
  * arroy `BinaryQuantizedCosine`: the cosine distance (1 - cos) / 2 of the two sign vectors read as +-1:
    cos = (d - 2h) / d, so the distance is h / d — the same number.

Whatever the crate returns is a strictly increasing function of h, so the ORDER of the hits — (distance, docid) ascending,
ties by docid — is the same under every candidate: docid order is pinned even though the value is not.  This test holds
the oracle's order (orc_bq_topk: h / d) equal to the order under each candidate computed independently in numpy."""
import numpy as np

from oracle import oracle as O


def test_every_candidate_distance_orders_the_hits_alike():
    rng = np.random.default_rng(3)
    for n, d, k in ((500, 64, 50), (2000, 96, 200), (300, 7, 300), (1000, 768, 20)):
        rows = rng.standard_normal((n, d)).astype(np.float32)
        rows[rng.random((n, d)) < 0.05] = 0.0          # exact zeros quantise to 0 (bit = x > 0)
        ids = (np.arange(n, dtype=np.uint32) * 3 + 1)
        q = rng.standard_normal(d).astype(np.float32)
        got_ids, got_dist = O.bq_topk(rows, ids, q, k)
        rb, qb = rows > 0, q > 0
        h = (rb != qb[None, :]).sum(axis=1)                                   # hannoy Hamming (raw)
        sr, sq = np.where(rb, 1.0, -1.0).astype(np.float32), np.where(qb, 1.0, -1.0).astype(np.float32)
        cos = (sr @ sq) / (np.linalg.norm(sr, axis=1) * np.linalg.norm(sq))     # arroy BinaryQuantizedCosine on +-1
        cand = {"hamming": h.astype(np.float64), "hamming/dim": h / d, "(1-cos)/2 of the sign vectors": (1.0 - cos.astype(np.float64)) / 2.0}
        for name, dist in cand.items():
            # (distance, docid) ascending; equal h must stay equal under the candidate (no rounding splits a tie)
            key = np.round(dist * d * 2).astype(np.int64) if name != "hamming" else dist.astype(np.int64)
            order = np.lexsort((ids, key))[:k]
            assert ids[order].tolist() == got_ids.tolist(), name
        assert np.allclose(got_dist, (h / d)[np.lexsort((ids, h))[:k]].astype(np.float32))
