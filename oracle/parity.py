"""Parity checks at BASELINE.json's full sizes — TEST INFRASTRUCTURE ONLY (like the rest of oracle/).

Used by tests/test_configs_gpu.py and by the untimed post-run check of bench.py's cpu_baseline leg.  The product
(meilisearch_amd/, libmsi.so) never imports this.

The scalar oracle (msi_oracle.c) takes ~2 ms per 768-d row, so at 10 M rows it cannot score every row for every
query in a test.  The check therefore runs in two steps, both on the CPU:

  1. candidates: the multi-threaded SIMD scan of msi_cpubase.c (relaxed summation order; cross-checked against
     the oracle in tests/test_cpubase_vs_oracle.py) returns, per chunk of rows and per query, the k + `extra` best
     rows.  The union over chunks contains the exact top-k unless more than `extra` rows of one chunk sit within
     the ~1e-6 summation noise of the k-th distance — the check reports how far the nearest non-candidate is
     (`margin`) so that this is visible, and fails when the margin is inside the noise;
  2. verdict: the ORACLE (orc_vs_topk: reference arithmetic of store.rs:1036-1093, (distance, docid) order) ranks
     the candidate rows; docids must be identical and in identical order, distances bit-identical.

Rows arrive chunk by chunk (the caller copies them from HBM), so host memory stays at one chunk.
"""
import concurrent.futures as cf

import numpy as np

from . import cpubase
from . import oracle as orc


class TopkChecker:
    def __init__(self, queries, k, extra=64):
        self.q = np.ascontiguousarray(queries, dtype=np.float32)
        self.nq, self.d = self.q.shape
        self.k, self.kc = int(k), int(k) + int(extra)
        self.cand_ids = [[] for _ in range(self.nq)]
        self.cand_rows = [[] for _ in range(self.nq)]
        self.worst_kept = np.full(self.nq, -np.inf, dtype=np.float64)  # per chunk: the last candidate's distance
        self.rows_seen = 0

    def add_chunk(self, docids, rows):
        """docids u32 [n] ascending, rows f32 [n, d] (host)."""
        rows = np.ascontiguousarray(rows, dtype=np.float32)
        docids = np.ascontiguousarray(docids, dtype=np.uint32)
        n = rows.shape[0]
        self.rows_seen += n
        if n == 0:
            return
        scan = cpubase.CpuVectorScan(rows, docids)
        ids, dist, cnt = scan.search(self.q, min(self.kc, n))
        first = int(docids[0])
        contiguous = int(docids[-1]) - first + 1 == n
        for j in range(self.nq):
            c = int(cnt[j])
            sel = ids[j, :c]
            pos = (sel - first).astype(np.int64) if contiguous else np.searchsorted(docids, sel)
            self.cand_ids[j].append(sel.copy())
            self.cand_rows[j].append(rows[pos].copy())
            if c == self.kc and n > self.kc:
                # distance of the best row this chunk did NOT hand over is >= dist[j, c-1]
                self.worst_kept[j] = max(self.worst_kept[j], -float(dist[j, c - 1]))

    def verdict(self, got_ids, got_dist, got_cnt):
        """Compare with the product's results ([nq, k] docids / distances, [nq] counts)."""
        got_ids = np.asarray(got_ids).astype(np.uint32).reshape(self.nq, -1)
        got_dist = np.asarray(got_dist, dtype=np.float32).reshape(self.nq, -1)
        got_cnt = np.asarray(got_cnt).astype(np.int64).reshape(self.nq)
        mismatches, min_margin, details = 0, np.inf, []
        for j in range(self.nq):
            ids = np.concatenate(self.cand_ids[j]) if self.cand_ids[j] else np.zeros(0, np.uint32)
            rows = np.concatenate(self.cand_rows[j]) if self.cand_rows[j] else np.zeros((0, self.d), np.float32)
            order = np.argsort(ids, kind="stable")
            e_ids, e_dist = orc.vs_topk(rows[order], ids[order], self.q[j], self.k)
            want = min(self.k, self.rows_seen)
            ok = (int(got_cnt[j]) == e_ids.size == want
                  and got_ids[j, :want].tolist() == e_ids.tolist()
                  and got_dist[j, :want].view(np.uint32).tolist() == e_dist.view(np.uint32).tolist())
            if e_dist.size and np.isfinite(self.worst_kept[j]):
                # every row outside the candidate set is at least this far beyond the k-th result
                min_margin = min(min_margin, -self.worst_kept[j] - float(e_dist[-1]))
            if not ok:
                mismatches += 1
                if len(details) < 4:
                    details.append({"query": j, "got": got_ids[j, :want].tolist()[:8], "expected": e_ids.tolist()[:8]})
        return {"checked_queries": self.nq, "mismatches": mismatches, "rows": self.rows_seen, "k": self.k,
                "candidate_margin": None if not np.isfinite(min_margin) else float(min_margin),
                "checker": "oracle/msi_oracle.c orc_vs_topk over the k+%d candidates per chunk of the multi-threaded "
                           "oracle/msi_cpubase.c scan; docids identical in order, f32 distances bit-identical"
                           % (self.kc - self.k),
                **({"first_mismatches": details} if details else {})}


def check_typo_lookup(concat, offsets, queries, got, threads=None, cap_one=150, cap_two=50):
    """Every query of `queries` [(word, max_typos, is_prefix)] through the literal loops of the oracle
    (orc_typo_lookup: compute_derivations.rs:75-168), on `threads` host threads (ctypes releases the GIL).
    `got` = [(one_idx, two_idx)] from the product.  Returns the parity object."""
    dic = orc.Dictionary.from_flat(concat, offsets)
    orc.lib()
    threads = threads or max(1, cpubase.host_threads())

    def one(i):
        w, b, p = queries[i]
        e1, e2 = orc.typo_lookup(dic, w, b, p, cap_one, cap_two)
        g1, g2 = got[i]
        return np.asarray(g1).tolist() == e1.tolist() and np.asarray(g2).tolist() == e2.tolist()

    with cf.ThreadPoolExecutor(max_workers=min(threads, 128)) as ex:
        ok = list(ex.map(one, range(len(queries))))
    bad = [i for i, v in enumerate(ok) if not v]
    return {"checked_words": len(queries), "mismatches": len(bad), "dictionary_words": int(len(offsets) - 1),
            "checker": "oracle/msi_oracle.c orc_typo_lookup (literal loops of find_one_typo_derivations / "
                       "find_one_two_typo_derivations); index lists identical",
            **({"first_mismatches": [queries[i] for i in bad[:4]]} if bad else {})}


# ---------------------------------------------------------------------------------------------------------------
# Keyword leg at the headline size: msi_keyword_search_ranked over the synthetic 10 M-document index of
# tools/ranked_bench.cpp against oracle/ranking_oracle.py (the restatement the reference's snapshots pin, run over
# oracle/docset.py's numpy docid sets) reading the same stored posting bytes through oracle/synth_index.py.
class KeywordLegChecker:
    def __init__(self, runner_lib, runner_handle, n_docs):
        from . import synth_index as SI
        self.SI = SI
        self.lib, self.h = runner_lib, runner_handle
        SI_lib = SI.runner_lib()        # the same shared object, with the checker's prototypes
        self.index = SI.SynthIndex(SI_lib, runner_handle, n_docs)
        self.oracle = SI.KeywordOracle(self.index)

    def run_product(self, first, n, limit, max_details=16, universes=None):
        """The product's answers for prepared queries [first, first + n): rb_run_detailed (caller threads of the
        runner, msi_keyword_search_ranked each).  universes = (docids [n, stride] u32, counts [n] u32): every search
        restricted to its own candidate set (the rerank of a vector search's top-k, config 5)."""
        ids = np.zeros((n, limit), np.uint32)
        cnt = np.zeros(n, np.uint32)
        scores = np.zeros((n, limit), np.float64)
        det = np.zeros((n, limit, max_details, 3), np.uint32)
        ndet = np.zeros((n, limit), np.uint32)
        cand = np.zeros(n, np.uint64)
        if universes is not None:
            u_ids = np.ascontiguousarray(universes[0], dtype=np.uint32)
            u_cnt = np.ascontiguousarray(universes[1], dtype=np.uint32)
            st = self.index.lib.rb_run_universes(self.h, first, n, limit, u_ids.ctypes.data, u_cnt.ctypes.data, u_ids.shape[1],
                                                 ids.ctypes.data, cnt.ctypes.data, scores.ctypes.data, det.ctypes.data,
                                                 ndet.ctypes.data, cand.ctypes.data)
        else:
            st = self.index.lib.rb_run_detailed(self.h, first, n, limit, ids.ctypes.data, cnt.ctypes.data, scores.ctypes.data,
                                                det.ctypes.data, ndet.ctypes.data, cand.ctypes.data)
        assert st == 0, "msi_keyword_search_ranked failed"
        return ids, cnt, scores, det, ndet, cand

    def verdict(self, first, n, limit, product=None, universes=None):
        """-> dict for the bench line's parity object / the test's assertion."""
        SI = self.SI
        ids, cnt, scores, det, ndet, cand = product if product is not None else self.run_product(first, n, limit, universes=universes)
        bad, first_bad, hits, n_details = 0, None, 0, 0
        for i in range(n):
            q = self.index.query(first + i)
            uni = None if universes is None else np.asarray(universes[0][i][:int(universes[1][i])])
            want_ids, want_sc, want_cand = self.oracle.search(q, limit=limit, detailed=True, universe=uni,
                                                              negatives=self.index.negatives(first + i))
            got = SI.product_details(ids[i], cnt[i], det[i], ndet[i], limit, det.shape[2])
            want = [(d, [SI.oracle_detail(s) for s in sc]) for d, sc in zip(want_ids, want_sc)]
            hits += len(want)
            n_details += sum(len(sc) for _, sc in want)
            if got != want or int(cand[i]) != want_cand:
                bad += 1
                if first_bad is None:
                    first_bad = {"query": q, "product": str(got[:3]), "oracle": str(want[:3]),
                                 "candidates": [int(cand[i]), want_cand]}
        out = {"checked_queries": n, "mismatches": bad, "hits_compared": hits, "score_details_compared": n_details,
               "documents": self.index.n_docs,
               "checker": "oracle/ranking_oracle.py (pinned to the reference's 108 snapshot searches, here over oracle/docset.py's "
                          "numpy docid sets) reading the synthetic index's stored CboRoaringBitmap bytes through "
                          "oracle/synth_index.py; typo derivations from oracle/msi_oracle.c: docids in order, every hit's score "
                          "details and the candidate counts identical"}
        if first_bad is not None:
            out["first_mismatch"] = first_bad
        return out
