#!/usr/bin/env python
"""Differential fuzzing of the ranked keyword search's HOST logic (msi_search.hip compiled against the test double
of the device, tests/hostlogic) against the CPU oracle: random corpora, index settings (exact attributes / words,
prefix databases, synonyms, stop words, typo thresholds), criteria lists, queries (typos, prefixes, phrases),
strategies, offsets, limits, score thresholds and deadlines.

    python tools/fuzz_ranked_hostlogic.py [first_seed] [seconds]
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
_lib.lib()
L = C.CDLL(H.SO)
L.msi_keyword_search_ranked.restype, L.msi_keyword_search_ranked.argtypes = _lib.PROTOTYPES["msi_keyword_search_ranked"]
L.mock_bits_create.restype, L.mock_bits_create.argtypes = C.c_void_p, [C.c_uint64, C.c_uint32]
L.mock_bits_destroy.restype, L.mock_bits_destroy.argtypes = None, [C.c_void_p]
L.mock_dict_create.restype, L.mock_dict_create.argtypes = C.c_void_p, [C.c_void_p, C.c_void_p, C.c_uint32, H.LOOKUP_FN]
L.mock_dict_destroy.restype, L.mock_dict_destroy.argtypes = None, [C.c_void_p]
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
    index = ToyMilli(docs, searchable=fields if rng.random()<0.8 else None, criteria=criteria, **kw)
    dic = O.Dictionary(index.words)
    def lookup(w,m,p):
        a,b=O.typo_lookup(dic,w,m,p); return [index.words[i] for i in a],[index.words[i] for i in b]
    h = H.MockHarness(L, index, n_slots=1024)
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
        tms = rng.choice(["last","all"]); detailed=rng.random()<0.5; offset=rng.choice([0,0,1,5]); limit=rng.choice([1,5,20,100])
        thr = rng.choice([None,None,0.3,0.7,0.9]); sa = rng.choice([None,None,None,0,1,2,4])
        negs = []
        if rng.random() < 0.2:
            negs.append(rng.choice(G.VOCAB))
        if rng.random() < 0.1:
            negs.append((rng.choice(G.VOCAB), rng.choice(G.VOCAB)))
        try:
            want = RO.search(RO.Ctx(index,lookup), q, tms=tms, offset=offset, length=limit, detailed=detailed, threshold=thr, stop_after=sa, negatives=negs)
            deg = RO.bucket_sort.degraded if hasattr(RO.bucket_sort,'degraded') else False
            terms = query_terms(q, stop_words=index.stop_words)
            for ng in negs:
                terms.append(([ng], False, 0, 0, False, True) if isinstance(ng, str) else (list(ng), True, 0, 0, False, True))
            hits, cand, gdeg = R.keyword_search_ranked(
                h.dict, h.pool, h.cb, terms, index.criteria, strategy=R.TERMS_ALL if tms == "all" else R.TERMS_LAST,
                offset=offset, limit=limit, detailed=detailed, searchable_fids=index.searchable_fids,
                searchable_weights=[index.weights[f] for f in index.searchable_fids], max_weight=index.max_weight,
                authorize_typos=index.authorize_typos, min_one=index.min_one, min_two=index.min_two, stop_after=sa,
                score_threshold=thr, return_degraded=True, _entry=L.msi_keyword_search_ranked)
        except Exception as e:
            print("EXC", seed, repr(q), criteria, kw, e); bad+=1; continue
        n+=1
        ok = [d for d,_ in hits]==want[0] and cand==len(want[2]) and [[tuple(s) for s in sc] for _,sc in hits]==[[G.oracle_score(s) if s[0]!="Skipped" else ("Skipped",0,1) for s in sc] for sc in want[1]]
        if not ok:
            bad+=1
            print("MISMATCH seed",seed,repr(q),tms,detailed,offset,limit,thr,sa,criteria,kw)
            print("  want",want[0][:10],len(want[2])); print("  got ",[d for d,_ in hits][:10],cand)
            if bad>5: sys.exit(1)
    h.close()
print("cases",n,"bad",bad)
