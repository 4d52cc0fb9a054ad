"""msi_federated_compare / msi_federated_merge (host): crates/meilisearch/src/search/federated/weighted_scores.rs:1-46 and
the k-way merge of perform.rs:545-610, against a literal Python restatement of the Rust code."""
import ctypes as C
import functools
import random

import numpy as np

from meilisearch_amd._lib import lib


class WV(C.Structure):
    _fields_ = [("kind", C.c_uint32), ("asc", C.c_uint32), ("value", C.c_double)]


EPS = 2.220446049250313e-16


def partial_cmp(l, r):          # WeightedScoreValue::partial_cmp, score_details.rs:57-101; None = not comparable
    (lk, la, lv), (rk, ra, rv) = l, r
    if lk == 0 and rk == 0:
        return 0 if abs(lv - rv) <= EPS else (-1 if lv < rv else 1)
    if lk == 1 and rk == 1:
        if la != ra:
            return None
        ln, rn = lv != lv, rv != rv
        if ln or rn:
            return 0 if ln and rn else (-1 if ln else 1)
        o = 0 if lv == rv else (-1 if lv < rv else 1)
        return -o if la else o
    if lk == 2 and rk == 2:
        if la != ra:
            return None
        ln, rn = lv != lv, rv != rv
        if ln or rn:
            return 0 if ln and rn else (-1 if ln else 1)
        return 0 if abs(lv - rv) <= EPS else (-1 if lv < rv else 1)
    return None


def compare(left, lg, right, rg):   # weighted_scores::compare
    i = 0
    while True:
        hl, hr = i < len(left), i < len(right)
        if not hl and not hr:
            return 0
        if not hl:
            return -1
        if not hr:
            return 1
        c = partial_cmp(left[i], right[i])
        if c == 0:
            i += 1
            continue
        if c is not None:
            return c
        lc, rc = len(left) - i - 1, len(right) - i - 1
        if lc != rc:
            return -1 if lc < rc else 1
        break
    return 0 if lg == rg else (-1 if lg < rg else 1)


def c_compare(left, lg, right, rg):
    L = (WV * max(1, len(left)))(*[WV(*v) for v in left])
    R = (WV * max(1, len(right)))(*[WV(*v) for v in right])
    return lib().msi_federated_compare(L, len(left), lg, R, len(right), rg)


def rand_value(rng):
    kind = rng.choice([0, 0, 0, 1, 2])
    v = rng.choice([0.0, 0.25, 0.5, 0.5 + 1e-17, 0.75, 1.0, float("nan")]) if kind else rng.choice([0.1, 0.5, 0.5 + 1e-17, 0.9])
    return (kind, rng.randint(0, 1), v)


def test_compare_matches_the_reference_logic():
    rng = random.Random(3)
    for _ in range(20000):
        left = [rand_value(rng) for _ in range(rng.randint(0, 4))]
        right = [rand_value(rng) for _ in range(rng.randint(0, 4))]
        lg, rg = rng.choice([0.2, 0.5, 0.8]), rng.choice([0.2, 0.5, 0.8])
        assert c_compare(left, lg, right, rg) == compare(left, lg, right, rg), (left, lg, right, rg)
    # hybrid.rs-style literals: a keyword hit 0.9848... against a semantic hit 0.99 at equal weights
    assert c_compare([(0, 0, 0.9848484848484848)], 0.98, [(0, 0, 0.990290343761444)], 0.99) == -1


def test_merge_is_the_k_way_merge_by_compare():
    rng = random.Random(9)
    for _ in range(300):
        n_lists = rng.randint(1, 4)
        lists = []
        for _l in range(n_lists):
            hits = [([(0, 0, round(rng.random(), 2))], round(rng.random(), 2)) for _ in range(rng.randint(0, 6))]
            # each list is already in ITS ranking order (best first)
            hits.sort(key=functools.cmp_to_key(lambda a, b: -compare(a[0], a[1], b[0], b[1])))
            lists.append(hits)
        # expected: repeated pick of the best head; ties keep the lower list index (query_index), perform.rs:557-567
        at, exp = [0] * n_lists, []
        while True:
            best = None
            for l in range(n_lists):
                if at[l] >= len(lists[l]):
                    continue
                if best is None or compare(lists[l][at[l]][0], lists[l][at[l]][1], lists[best][at[best]][0], lists[best][at[best]][1]) > 0:
                    best = l
            if best is None:
                break
            exp.append((best, at[best]))
            at[best] += 1
        offset, limit = rng.randint(0, 3), rng.randint(1, 8)
        lens = (C.c_uint32 * n_lists)(*[len(h) for h in lists])
        keep, vals, offs, glob = [], (C.POINTER(WV) * n_lists)(), (C.POINTER(C.c_uint32) * n_lists)(), (C.POINTER(C.c_double) * n_lists)()
        for l, hits in enumerate(lists):
            flat = [WV(*v) for h in hits for v in h[0]] or [WV(0, 0, 0.0)]
            va = (WV * len(flat))(*flat)
            oa = (C.c_uint32 * (len(hits) + 1))(*np.cumsum([0] + [len(h[0]) for h in hits]).tolist())
            ga = (C.c_double * max(1, len(hits)))(*[h[1] for h in hits] or [0.0])
            keep += [va, oa, ga]
            vals[l], offs[l], glob[l] = va, oa, ga
        ol, op = (C.c_uint32 * limit)(), (C.c_uint32 * limit)()
        lib().msi_federated_merge.restype = C.c_uint32
        n = lib().msi_federated_merge(n_lists, lens, vals, offs, glob, offset, limit, ol, op)
        assert [(ol[i], op[i]) for i in range(n)] == exp[offset:offset + limit]


def test_reference_federated_orders_replay_through_the_merge():
    """crates/meilisearch/tests/search/multi/mod.rs (tests/golden/federated_fixtures.json): the merged hit order of every
    plain-keyword federated snapshot — 11 requests, 69 hits, 27 ties between hits of different queries.  Each query's own
    list (its hits in the order the response shows them) goes back through msi_federated_merge_q with the query's number:
    the interleaving must be the reference's, ties included (`left.query_index < right.query_index`, perform.rs:566,609).
    The plain merge (ties by list order) is what ADVICE r2 found wrong: with the lists handed over in another order than the
    queries it must FAIL on the same fixtures."""
    import json
    import os
    fix = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "federated_fixtures.json")))
    L = lib()
    L.msi_federated_merge_q.restype = C.c_uint32
    n_ties = 0
    wrong_without_query_index = 0
    for case in fix["cases"]:
        hits = case["hits"]
        queries = sorted({q for q, _ in hits})
        # lists in REVERSE query order: only the query numbers can restore the reference's tie order
        order = list(reversed(queries))
        per = {q: [(i, s) for i, (qq, s) in enumerate(hits) if qq == q] for q in queries}
        n_lists = len(order)
        lens = (C.c_uint32 * n_lists)(*[len(per[q]) for q in order])
        keep, vals, offs, glob, qidx = [], (C.c_void_p * n_lists)(), (C.c_void_p * n_lists)(), (C.c_void_p * n_lists)(), (C.c_void_p * n_lists)()
        for li, q in enumerate(order):
            n = len(per[q])
            v = (WV * max(n, 1))()
            o = (C.c_uint32 * (n + 1))(*range(n + 1))
            g = (C.c_double * max(n, 1))()
            qi = (C.c_uint32 * max(n, 1))()
            for j, (_, s) in enumerate(per[q]):
                v[j].kind, v[j].asc, v[j].value = 0, 0, s
                g[j] = s
                qi[j] = q
            keep += [v, o, g, qi]
            vals[li], offs[li], glob[li], qidx[li] = C.addressof(v), C.addressof(o), C.addressof(g), C.addressof(qi)
        total = len(hits)
        ol, op = (C.c_uint32 * total)(), (C.c_uint32 * total)()
        n = L.msi_federated_merge_q(n_lists, lens, vals, offs, glob, qidx, 0, total, ol, op)
        assert n == total
        got = [per[order[ol[i]]][op[i]][0] for i in range(n)]
        assert got == list(range(total)), (case["src"], got)
        n_ties += sum(1 for a, b in zip(hits, hits[1:]) if a[1] == b[1] and a[0] != b[0])
        n2 = L.msi_federated_merge(n_lists, lens, vals, offs, glob, 0, total, ol, op)
        if [per[order[ol[i]]][op[i]][0] for i in range(n2)] != list(range(total)):
            wrong_without_query_index += 1
    assert n_ties >= 20 and wrong_without_query_index >= 1
