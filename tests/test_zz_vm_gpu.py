"""The command-list back end of the ranked keyword search (msi_vm.hip) and the HBM posting cache:
  * every reference snapshot search gives the same hits with the cache cold (postings decoded out of the pinned staging
    buffer, bodies stored into the cache by the decoding workgroups) and warm (decoded from HBM), and the cache reports
    hits on the second pass;
  * the direct back end (one launch per set operation, MSI_SEARCH_VM=0) and the command lists agree hit for hit,
    score detail for score detail;
  * searches running concurrently (one pool per thread, one shared dictionary + cache, lists combined into shared
    launches) return what they return alone."""
import json
import os
import threading

import pytest

from tests.test_search_gpu import CASES, FIX, Harness, build_index

pytestmark = pytest.mark.gpu


def _search(h, case):
    hits, _ = h.search(case["query"], tms=case["tms"], offset=case["offset"], limit=case["limit"], detailed=True,
                       stop_after=case.get("stop_after"))
    return [(d, [tuple(s) for s in sc]) for d, sc in hits]


def _by_index():
    groups = {}
    for c in CASES:
        groups.setdefault(c["index"], []).append(c)
    return groups


def test_cold_and_warm_posting_cache_and_direct_backend_agree(monkeypatch):
    checked = 0
    for key, cases in _by_index().items():
        h = Harness(build_index(FIX["indexes"][key]))
        monkeypatch.setenv("MSI_SEARCH_VM", "0")
        direct = [_search(h, c) for c in cases]
        monkeypatch.setenv("MSI_SEARCH_VM", "1")
        plain = [_search(h, c) for c in cases]
        h.dict.enable_posting_cache(8 << 20)
        cold = [_search(h, c) for c in cases]
        s0 = h.dict.posting_cache_stats()
        warm = [_search(h, c) for c in cases]
        s1 = h.dict.posting_cache_stats()
        assert direct == plain == cold == warm
        for c, got in zip(cases, warm):
            if c["ids"] is not None:
                assert [d for d, _ in got] == c["ids"]
        assert s1["hits"] >= s0["hits"]
        assert s1["bytes_used"] <= s1["capacity"]
        checked += len(cases)
    assert checked >= 90


def test_concurrent_searches_share_launches_and_the_cache():
    key, cases = max(_by_index().items(), key=lambda kv: len(kv[1]))
    index = build_index(FIX["indexes"][key])
    base = Harness(index)
    expected = [_search(base, c) for c in cases]
    base.dict.enable_posting_cache(4 << 20)
    import meilisearch_amd as ma
    n_threads = 6
    out = [None] * n_threads
    errs = []

    def worker(t):
        try:
            h = Harness.__new__(Harness)
            h.R, h.index, h.ctx, h.dict, h.cb = base.R, index, base.ctx, base.dict, base.R.IndexCallbacks(index)
            h.pool = ma.BitsPool(base.ctx, max(index.n_docs, 1), 512, private_stream=True)
            out[t] = [[_search(h, c) for c in cases] for _ in range(3)]
        except Exception as e:   # noqa: BLE001
            errs.append(repr(e))
    ths = [threading.Thread(target=worker, args=(t,)) for t in range(n_threads)]
    for th in ths:
        th.start()
    for th in ths:
        th.join()
    assert not errs, errs
    for t in range(n_threads):
        for rep in out[t]:
            assert rep == expected
    st = base.dict.posting_cache_stats()
    assert st["bytes_used"] <= st["capacity"]     # (postings of <= 7 documents are raw ids: never cached)


def test_bucket_sort_tasks_equal_the_sequential_loop_also_when_the_pool_runs_out_of_slots(monkeypatch):
    """The bucket sort runs sibling buckets' sub-trees as cooperative tasks that share one command list
    (MSI_SEARCH_TASKS, default 24).  Same hits and score details as the sequential loop (MSI_SEARCH_TASKS=0); with the
    slot gate off and a small pool the tasks run out of slots and the search is re-run one bucket at a time
    (MSI_SEARCH_TASKS_NO_GATE is a test knob) — still the same answers."""
    import meilisearch_amd as ma
    checked = 0
    for key, cases in _by_index().items():
        index = build_index(FIX["indexes"][key])
        h = Harness(index)
        monkeypatch.setenv("MSI_SEARCH_TASKS", "0")
        sequential = [_search(h, c) for c in cases]
        monkeypatch.setenv("MSI_SEARCH_TASKS", "24")
        tasks = [_search(h, c) for c in cases]
        small = Harness.__new__(Harness)
        small.R, small.index, small.ctx, small.dict, small.cb = h.R, index, h.ctx, h.dict, h.R.IndexCallbacks(index)
        small.pool = ma.BitsPool(h.ctx, max(index.n_docs, 1), 256)
        monkeypatch.setenv("MSI_SEARCH_TASKS_NO_GATE", "1")
        starved = [_search(small, c) for c in cases]
        monkeypatch.delenv("MSI_SEARCH_TASKS_NO_GATE")
        assert sequential == tasks == starved
        checked += len(cases)
    assert checked >= 90


def test_starved_tasks_are_rerun_sequentially_and_still_match_the_oracle(monkeypatch):
    """300-document random corpora rank into many buckets: without the slot gate the bucket sort's tasks exhaust the
    harness's pool, the search is cut off and re-run one bucket at a time — the results still equal the oracle's."""
    import tests.test_search_gpu as G
    monkeypatch.setenv("MSI_SEARCH_TASKS_NO_GATE", "1")
    G.test_matches_oracle_on_random_corpora(1, 100)


def test_a_search_that_cannot_allocate_fails_with_an_error_wherever_hbm_runs_out(monkeypatch):
    """What a search allocates when it first needs it — the list arenas, the compact space's companion pool and rank tables,
    staging buffers — fails when HBM is exhausted.  Wherever that happens the search must come back with MSI_E_OOM (or
    MSI_E_HIP) and the pool must serve the next search: commands recorded against a null rank table faulted the device (a
    384-caller run beside the C4 store dumped a GPU core, profiles/r6_step_callers_and_slots.log).  The CPU emulation of
    HIP injects the failure after k more allocations, k = 0, 1, 2, ... until the search gets through: the test is skipped
    on a device."""
    import tests.test_search_gpu as G
    from meilisearch_amd import _lib
    if type(_lib.lib()).__name__ != "EmulatedLib":
        pytest.skip("allocation failures are injected by the CPU emulation of HIP (tests/emu)")
    monkeypatch.setenv("MSI_SEARCH_COMPACT", "2")          # every universe is compacted
    case = [c for c in G.CASES if c["ids"] and c["detailed"]][0]
    kw = dict(tms=case["tms"], offset=case["offset"], limit=case["limit"], detailed=case["detailed"])
    failures = set()
    for k in range(64):
        h = G.Harness(G.build_index(G.FIX["indexes"][case["index"]]))   # a fresh pool: nothing of its compact space exists yet
        monkeypatch.setenv("MSI_EMU_FAIL_MALLOC", str(k))
        try:
            hits, _ = h.search(case["query"], **kw)
            failed = None
        except _lib.MsiError as e:
            failed = str(e)
        monkeypatch.delenv("MSI_EMU_FAIL_MALLOC")
        if failed is not None:
            assert failed.startswith(("MSI_E_OOM", "MSI_E_HIP")), failed
            failures.add(failed)
            hits, _ = h.search(case["query"], **kw)       # the same pool, HBM available again
        assert [d for d, _ in hits] == case["ids"], (k, failed)
        if failed is None:
            break
    else:
        raise AssertionError("the search never got through")
    assert len(failures) >= 3, failures                   # (arena, rank tables, companion pool, ...: several places)


def test_documents_spread_over_many_chunks(monkeypatch):
    """The command lists work chunk by chunk (65 536 documents each) and skip, per chunk, the sets and the paths whose
    chunk summary says "empty here".  The reference's snapshot indexes hold a few dozen documents — one chunk — so here
    every internal docid is multiplied by a stride that puts each document into a chunk of its own: the same searches
    must return the same documents (times the stride), in the same order, with the same score details — also a second
    time, on the warm posting cache."""
    import meilisearch_amd as ma
    import tests.toy_milli as T
    from tests.toy_milli import query_terms
    STRIDE = 70001
    plain_cbo = T.cbo_bytes
    checked = multi = 0
    for key, cases in _by_index().items():
        index = build_index(FIX["indexes"][key])
        if index.n_docs < 2:
            continue
        h = Harness(index)
        expected = [_search(h, c) for c in cases]
        R = h.R
        pool = ma.BitsPool(h.ctx, index.n_docs * STRIDE, 512)
        dictionary = ma.GpuDictionary(h.ctx, [w.encode() for w in index.words])
        dictionary.enable_posting_cache(8 << 20)
        monkeypatch.setattr(T, "cbo_bytes", lambda s: plain_cbo({d * STRIDE for d in s}))
        cb = R.IndexCallbacks(index)
        universe = plain_cbo({d * STRIDE for d in range(index.n_docs)})

        def spread(case):
            hits, _ = R.keyword_search_ranked(
                dictionary, pool, cb, query_terms(case["query"], stop_words=index.stop_words), index.criteria,
                strategy=R.strategy_of(case["tms"]), offset=case["offset"], limit=case["limit"],
                detailed=True, searchable_fids=index.searchable_fids,
                searchable_weights=[index.weights[f] for f in index.searchable_fids], max_weight=index.max_weight,
                authorize_typos=index.authorize_typos, min_one=index.min_one, min_two=index.min_two,
                stop_after=case.get("stop_after"), universe_cbo=universe)
            return [(d, [tuple(s) for s in sc]) for d, sc in hits]

        for c, e in zip(cases, expected):   # (what the comparison stands on: the reference's own snapshot)
            if c["ids"] is not None:
                assert [d for d, _ in e] == c["ids"]
        want = [[(d * STRIDE, sc) for d, sc in e] for e in expected]
        assert [spread(c) for c in cases] == want
        assert [spread(c) for c in cases] == want          # postings decoded out of the HBM cache this time
        monkeypatch.setattr(T, "cbo_bytes", plain_cbo)
        checked += len(cases)
        multi += index.n_docs * STRIDE > 4 * 65536
    assert checked >= 90 and multi >= 5


def test_fuzz_regressions_through_the_product():
    """The searches that differential fuzzing once caught (tests/test_search_hostlogic_cpu.py::FUZZ_REGRESSIONS), through the
    product's kernels: on the device — or, when this file runs in the CPU tier, on the emulated build."""
    from meilisearch_amd import _lib
    from tests.test_search_hostlogic_cpu import FUZZ_REGRESSIONS, run_fuzz_seeds
    run_fuzz_seeds(FUZZ_REGRESSIONS, "--emulated-kernels" if type(_lib.lib()).__name__ == "EmulatedLib" else "--device")


def test_random_corpora_spread_over_chunks_match_the_oracle():
    """A short run of the differential fuzzer with every docid multiplied on the product's side: random corpora, settings,
    criteria and queries against the oracle while each list works over many 65 536-document chunks (on the device: 33
    chunks; in the CPU tier's emulation: 4)."""
    import subprocess
    import sys
    from meilisearch_amd import _lib
    emulated = type(_lib.lib()).__name__ == "EmulatedLib"
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, FUZZ_SPREAD="700" if emulated else "7001")
    out = subprocess.run([sys.executable, os.path.join(root, "tools", "fuzz_ranked_hostlogic.py"), "31000", "10",
                          "--emulated-kernels" if emulated else "--device"], cwd=root, env=env, capture_output=True, text=True,
                         timeout=400)
    tail = out.stdout[-2000:] + out.stderr[-2000:]
    assert out.returncode == 0 and " bad 0" in out.stdout, tail
    assert int(out.stdout.split("cases")[-1].split()[0]) >= 30, tail


def test_index_views_do_not_share_what_the_engine_remembers():
    """`attributesToSearchOn` reads the index through a restricted view (tests/toy_milli.py: ToyMilli.restricted): the same
    keys answer with other values.  The posting cache in HBM and what the engine knows about absent keys are kept per
    (msi_search_params::index_view, key): searches that alternate between the index and two views of it — one dictionary,
    one cache, one pool — each equal the oracle reading through their own view, cold and warm.  Control: with the views
    unnamed (index_view 0 for all) the same alternation reads another view's postings and goes wrong."""
    from oracle import oracle as O
    from oracle import ranking_oracle as RO
    import tests.test_search_gpu as G
    from tests.toy_milli import ToyMilli, query_terms
    docs = G.random_corpus(31, 300)
    for i, d in enumerate(docs):
        d["tags"] = " ".join(G.VOCAB[(i * 5 + k * 11) % len(G.VOCAB)] for k in range(i % 3))
    index = ToyMilli(docs, searchable=["title", "body", "tags"], prefix_threshold=3)
    dic = O.Dictionary(index.words)

    def lookup(word, max_typos, is_prefix):
        one, two = O.typo_lookup(dic, word, max_typos, is_prefix)
        return [index.words[i] for i in one], [index.words[i] for i in two]
    views = [index, index.restricted(["title"]), index.restricted(["body", "tags"])]
    h = Harness(index)
    h.dict.enable_posting_cache(8 << 20)
    cbs = [h.R.IndexCallbacks(v) for v in views]
    queries = ["quick fox", "the lazy dog", "sun fl", "brown fox jumps", "summer", "su"]

    def run(named):
        bad = n = 0
        for rep in range(2):                       # cold, then warm
            for q in queries:
                for v, cb in zip(views, cbs):
                    R = h.R
                    hits, cand = R.keyword_search_ranked(
                        h.dict, h.pool, cb, query_terms(q, stop_words=index.stop_words), index.criteria, strategy=R.strategy_of("last"),
                        offset=0, limit=30, detailed=True, searchable_fids=index.searchable_fids,
                        searchable_weights=[index.weights[f] for f in index.searchable_fids], max_weight=index.max_weight,
                        authorize_typos=index.authorize_typos, min_one=index.min_one, min_two=index.min_two,
                        index_view=getattr(v, "index_view", 0) if named else 0)
                    want = RO.search(RO.Ctx(v, lookup), q, tms="last", offset=0, length=30, detailed=True)
                    ok = [d for d, _ in hits] == want[0] and cand == len(want[2]) and \
                        [[tuple(s_) for s_ in sc] for _, sc in hits] == [[G.oracle_score(s_) for s_ in sc] for sc in want[1]]
                    bad += int(not ok)
                    n += 1
        return bad, n
    bad, n = run(named=True)
    assert bad == 0 and n == 36
    stats = h.dict.posting_cache_stats()
    assert stats["hits"] > 0
    bad_unnamed, _ = run(named=False)              # (after the named pass: the plain index's keys are warm)
    assert bad_unnamed > 0


def test_a_page_past_what_one_list_reads_back():
    """offset + limit above MSI_VM_MAX_FIRSTK (8192 ids per list): the ids come from the direct first-k kernel, which reads
    docids — so such a search must not have moved into the compact space, where a set holds ranks (with
    MSI_SEARCH_COMPACT=2 this returned "a bucket held fewer documents than its cardinality said", or ranks as docids).
    The CPU tier runs this file with compaction forced."""
    from tests.toy_milli import ToyMilli
    docs = [{"id": i, "title": "apple" if i % 2 == 0 else "pear"} for i in range(17000)]
    index = ToyMilli(docs, searchable=["title"], criteria=["words"])
    h = Harness(index, n_slots=128)
    hits, cand = h.search("apple", criteria=["words"], offset=8200, limit=5)
    assert [d for d, _ in hits] == [16400, 16402, 16404, 16406, 16408]
    assert cand == 8500
    hits, _ = h.search("apple", criteria=["words"], offset=8180, limit=5)     # still inside one list: the compact path
    assert [d for d, _ in hits] == [16360, 16362, 16364, 16366, 16368]


def test_the_universe_of_an_unfiltered_search_is_not_materialised():
    """Round 5: without a filter and without negative terms the universe of a search is what its query graph matches — the
    engine records no "every document" set, no copy of it and no zeroed set per graph node (resolve_universe,
    search/new/mod.rs:273-301, computes the same documents).  A one-word search used to sweep 18 more sets over the whole
    index before its first wait: fill, five operations, two clears (byte model of the lists, msi_bits_vm_bytes)."""
    import ctypes as C

    from meilisearch_amd import _lib
    ix = build_index(FIX["indexes"][sorted(FIX["indexes"])[0]])
    h = Harness(ix)
    lib = _lib.lib()
    word = ix.words[len(ix.words) // 2]
    a, b = (C.c_uint64 * 3)(), (C.c_uint64 * 3)()
    lib.msi_bits_vm_bytes(a)
    hits, _ = h.search(word, criteria=["words"])
    lib.msi_bits_vm_bytes(b)
    assert hits
    set_bytes = 16   # (a pool of <= 128 documents: two 64-bit words per set)
    assert ix.n_docs <= 128
    sets, lists = (b[0] - a[0]) // set_bytes, b[2] - a[2]
    # 14 sets in 2 lists by default; a few more with the compact space forced (MSI_SEARCH_COMPACT=2: rank tables, the images
    # of the cached sets); 32 and more before
    assert lists <= 3 and sets <= 22, (sets, lists)
