#!/usr/bin/env python
"""Differential fuzzing of the ranked keyword search's HOST logic (msi_search.hip compiled against the test double
of the device, tests/hostlogic) against the CPU oracle: random corpora, index settings (exact attributes / words,
prefix databases, synonyms, stop words, typo thresholds), criteria lists, queries (typos, prefixes, phrases),
strategies, offsets, limits, score thresholds, deadlines, and — every other corpus — facet fields with Sort / Asc / Desc
rules, `_geo` points with GeoSort rules (bucket caps, error margins) and a `distinct` field.

    python tools/fuzz_ranked_hostlogic.py [first_seed] [seconds] [--emulated-kernels | --device]
"""
import os
import random
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C
import tests.test_search_hostlogic_cpu as H
import tests.test_search_gpu as G
from oracle import oracle as O, ranking_oracle as RO
from meilisearch_amd import _lib, ranking as R
from tests.toy_milli import ToyMilli, query_terms
EMULATED = "--emulated-kernels" in sys.argv
if EMULATED:
    # the same cases through the product's kernels on the CPU emulation of the HIP runtime (tests/emu): every launch of
    # the search is executed, the dictionary lookups included (test infrastructure; the product never loads that build)
    sys.argv.remove("--emulated-kernels")
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "emu"))
    import run_emulated
    _lib._LIB = run_emulated.EmulatedLib(run_emulated.build())
    from tests.test_zzz_distinct_gpu import device_lib
    L = device_lib()
elif "--device" in sys.argv:
    # ... and through the product itself on the MI355X
    sys.argv.remove("--device")
    from tests.test_zzz_distinct_gpu import device_lib
    L = device_lib()
else:
    L = H.load_hostlib()
# FUZZ_SPREAD=<stride>: every internal docid times the stride on the product's side (the postings it is handed, its pool,
# its universe), so that a 300-document corpus spans several 65 536-document chunks of the command lists — chunk
# summaries, per-chunk path skipping, first-k across chunks.  Corpora without facet fields only (per-document arrays
# are not spread).
SPREAD = int(os.environ.get("FUZZ_SPREAD", "0"))
if SPREAD:
    import copy
    import tests.toy_milli as T
    _plain_cbo = T.cbo_bytes
    T.cbo_bytes = lambda s_: _plain_cbo({d * SPREAD for d in s_})
seed0 = int(sys.argv[1]) if len(sys.argv)>1 else 0
budget = float(sys.argv[2]) if len(sys.argv)>2 else 120
ALLC = ["words","typo","proximity","attribute","attributeRank","wordPosition","exactness","sort"]
t_end = time.time()+budget; n=0; bad=0
seed = seed0
while time.time() < t_end:
    seed += 1
    rng = random.Random(seed)
    docs = G.random_corpus(seed, rng.choice([40, 120, 300]))
    if rng.random()<0.5:
        for d in docs: d["tags"] = " ".join(rng.choice(G.VOCAB) for _ in range(rng.randint(0,3)))
    faceted = rng.random() < 0.5 and not SPREAD
    if faceted:
        places = [(rng.uniform(-80, 80), rng.uniform(-179, 179)) for _ in range(6)]
        for d in docs:
            if rng.random() < 0.8: d["price"] = rng.choice([1, 2, 2.5, 3, 10, 10, 99.5])
            if rng.random() < 0.7: d["color"] = rng.choice(["red", "green", "blue", "Blue"])
            if rng.random() < 0.5: d["sizes"] = [rng.choice([36, 38, 40, "xl"]) for _ in range(rng.randint(1, 3))]
            if rng.random() < 0.75:
                lat, lng = rng.choice(places) if rng.random() < 0.6 else (rng.uniform(-89, 89), rng.uniform(-180, 180))
                d["_geo"] = {"lat": lat + rng.choice([0, 0, 5e-6, 2e-5]), "lng": lng}
    fields = [f for f in ("title","body","tags") if f in docs[0]]
    rng.shuffle(fields)
    kw = {}
    if rng.random()<0.3: kw["exact_attributes"]=[rng.choice(fields)]
    if rng.random()<0.3: kw["exact_words"]=rng.sample(G.VOCAB, 3)
    if rng.random()<0.3: kw["prefix_threshold"]=rng.choice([2,3,5])
    if rng.random()<0.3: kw["synonyms"]={"fast":["quick"],"sunflower":["sun flower"],"lazy dog":["sleepy hound","dogs"]}
    if rng.random()<0.3: kw["stop_words"]=rng.sample(["the","over","sun","dog"],2)
    if rng.random()<0.2: kw["authorize_typos"]=False
    if rng.random()<0.2: kw["min_one"],kw["min_two"]=3,6
    criteria = rng.sample(ALLC, rng.randint(1,6))
    if faceted and rng.random() < 0.4:
        criteria.insert(rng.randrange(len(criteria) + 1), rng.choice(["asc:price", "desc:color", "asc:sizes"]))
    index = ToyMilli(docs, searchable=fields if rng.random()<0.8 else None, criteria=criteria, **kw)
    dic = O.Dictionary(index.words)
    def lookup(w,m,p):
        a,b=O.typo_lookup(dic,w,m,p); return [index.words[i] for i in a],[index.words[i] for i in b]
    if SPREAD:
        wide = copy.copy(index)
        wide.n_docs = index.n_docs * SPREAD
        h = H.make_harness(L, wide, n_slots=int(os.environ.get("FUZZ_SLOTS", "1024")))
        spread_kw = {"universe_cbo": _plain_cbo({d * SPREAD for d in range(index.n_docs)})}
    else:
        h = H.make_harness(L, index, n_slots=int(os.environ.get("FUZZ_SLOTS", "1024")))
        spread_kw = {}
    for _ in range(6):
        nt = rng.randint(1,5)
        ws = []
        for i in range(nt):
            w = rng.choice(G.VOCAB)
            r = rng.random()
            if r<0.15 and len(w)>3:   # typo
                k=rng.randrange(len(w)); w=w[:k]+rng.choice("abcdefghijklmnop")+w[k+1:]
            elif r<0.25: w = w[:rng.randint(1,len(w))]
            ws.append(w)
        q = " ".join(ws)
        if rng.random()<0.25 and nt>=2:
            k=rng.randrange(nt-1); ws2=ws[:]; ws2[k]='"'+ws2[k]; ws2[k+1]=ws2[k+1]+'"'; q=" ".join(ws2)
        if rng.random()<0.2: q += " "
        tms = rng.choice(["last","all","frequency"]); detailed=rng.random()<0.5; offset=rng.choice([0,0,1,5]); limit=rng.choice([1,5,20,100])
        thr = rng.choice([None,None,0.3,0.7,0.9]); sa = rng.choice([None,None,None,0,1,2,4])
        negs = []
        if rng.random() < 0.2:
            negs.append(rng.choice(G.VOCAB))
        if rng.random() < 0.1:
            negs.append((rng.choice(G.VOCAB), rng.choice(G.VOCAB)))
        sort, distinct, geo = None, None, {}
        if faceted:
            if rng.random() < 0.6:
                sort = []
                for _ in range(rng.randint(1, 2)):
                    if rng.random() < 0.45:
                        sort.append((("_geoPoint", rng.uniform(-60, 60), rng.uniform(-170, 170)), rng.choice(["asc", "desc"])))
                    else:
                        sort.append((rng.choice(["price", "color", "sizes"]), rng.choice(["asc", "desc"])))
            if rng.random() < 0.4:
                distinct = rng.choice(["color", "sizes", "price"])
            if rng.random() < 0.3:
                geo["geo_max_bucket_size"] = rng.choice([1, 3, 50])
            if rng.random() < 0.2:
                geo["geo_distance_error_margin"] = rng.choice([0.0, 10.0, 3e6])
        exh = rng.random() < 0.3
        mth = rng.choice([None, 1000, 1000, 7]) if exh else None
        if q.strip() == "" and negs:
            negs = []
        if os.environ.get("FUZZ_KEEP_OLD_SORT_DISTINCT_FENCE") and sa is not None and distinct and (sort or any(c.startswith(("asc:", "desc:")) for c in criteria)):
            sa = None   # (until round 3 the product's Sort rule skipped the values `distinct` had emptied: sort.rs:214-217)
        if os.environ.get("FUZZ_ONLY") and os.environ["FUZZ_ONLY"] != q:      # debugging: replay one query of a seed (every draw above still happens)
            continue
        try:
            RO.GEO_PARAMS.clear()
            geo_strategy = rng.choice([("dynamic", 1000), ("dynamic", 1000), ("rtree", 1000), ("iterative", 1000)]) if geo or sort else ("dynamic", 1000)
            RO.GEO_PARAMS.update(strategy=geo_strategy)
            if "geo_max_bucket_size" in geo: RO.GEO_PARAMS["max_bucket_size"] = geo["geo_max_bucket_size"]
            if "geo_distance_error_margin" in geo: RO.GEO_PARAMS["distance_error_margin"] = geo["geo_distance_error_margin"]
            want = RO.search(RO.Ctx(index,lookup), q, tms=tms, offset=offset, length=limit, detailed=detailed, threshold=thr,
                             stop_after=sa, negatives=negs, sort=sort, distinct=distinct, exhaustive=exh, max_total_hits=mth)
            RO.GEO_PARAMS.clear()
            deg = RO.bucket_sort.degraded if hasattr(RO.bucket_sort,'degraded') else False
            extra = [(([ng], False, 0, 0, False, True) if isinstance(ng, str) else (list(ng), True, 0, 0, False, True)) for ng in negs]
            hits, cand, gdeg = h.search(q, tms=tms, offset=offset, limit=limit, detailed=detailed, stop_after=sa, sort=sort,
                                        distinct=distinct, extra_terms=extra, score_threshold=thr, return_degraded=True, exhaustive=exh,
                                        max_total_hits=mth, geo_strategy=geo_strategy, **geo, **spread_kw)
            if SPREAD:
                assert all(d % SPREAD == 0 for d, _ in hits), hits
                hits = [(d // SPREAD, sc) for d, sc in hits]
        except Exception as e:
            print("EXC", seed, repr(q), criteria, kw, e); bad+=1; continue
        n+=1
        def same_detail(g_, w_):
            if g_ == w_:
                return True
            # GeoSort's value is the point of the bucket's first document: documents whose distances agree to the
            # millimetre (the resolution of distance_between_two_points) are interchangeable there
            if g_[0] == "GeoSort" and w_[0] == "GeoSort" and g_[:3] == w_[:3] and g_[3] is not None and w_[3] is not None:
                return abs(RO.distance_between_two_points(g_[1], g_[3]) - RO.distance_between_two_points(g_[1], w_[3])) <= 1e-3
            return False
        got_sc = [[H.geo_score(s) for s in sc] for _, sc in hits]
        want_sc = [[H.geo_score(G.oracle_score(s)) if s[0] != "Skipped" else ("Skipped", 0, 1) for s in sc] for sc in want[1]]
        sc_ok = len(got_sc) == len(want_sc) and all(len(a_) == len(b_) and all(same_detail(x, y) for x, y in zip(a_, b_))
                                                     for a_, b_ in zip(got_sc, want_sc))
        ok = [d for d,_ in hits]==want[0] and cand==len(want[2]) and sc_ok
        if not ok:
            bad+=1
            print("MISMATCH seed",seed,repr(q),tms,detailed,offset,limit,thr,sa,criteria,kw,"sort",sort,"distinct",distinct,geo,"exhaustive",exh,mth)
            print("  want",want[0][:10],len(want[2])); print("  got ",[d for d,_ in hits][:10],cand)
            if os.environ.get("FUZZ_VERBOSE"):   # score details side by side
                for i in range(max(len(hits), len(want[0]))):
                    print("   ", i, (hits[i][0], got_sc[i]) if i < len(hits) else None, "|", (want[0][i], want_sc[i]) if i < len(want[0]) else None)
            if bad>5: sys.exit(1)
    h.close()
print("cases",n,"bad",bad)
