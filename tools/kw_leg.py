"""The keyword leg of bench.py's C4 step on its own — the same coherent corpus, the same query generator, the same
caller threads (tools/bin/libmsi_rankedbench.so) — as a process of its own: what bench.py's `keyword_roofline` runs plain
(MSI_SEARCH_CPU_PROFILE=1) and under `rocprofv3 --pmc ...` (counters over every vm_kernel dispatch), so that the counters
describe the HEADLINE's workload and not the round-3 hashed index.

    python tools/kw_leg.py [--docs 10000000] [--words 2000000] [--callers 160] [--queries 3072] [--passes 2] [--corpus coherent|hashed]

Prints one JSON line: queries/s of the measured passes, lists / rounds / microseconds per round (msi_bits_vm_stats), host CPU per
query by where it is spent (msi_search_cpu_profile), posting-cache hit rate, compaction counters, searches run in total
(so that summed counters can be divided per query)."""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

import meilisearch_amd as ma  # noqa: E402  (loads libmsi.so: GPU_MAX_HW_QUEUES is set by its constructor)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--docs", type=int, default=10_000_000)
    ap.add_argument("--words", type=int, default=2_000_000)
    ap.add_argument("--callers", type=int, default=160)
    ap.add_argument("--queries", type=int, default=3072, help="distinct queries (seeded as bench.py's)")
    ap.add_argument("--terms", type=int, default=3)
    ap.add_argument("--passes", type=int, default=2, help="measured passes over the queries (after one untimed pass)")
    ap.add_argument("--corpus", choices=["coherent", "hashed"], default="coherent")
    ap.add_argument("--k", type=int, default=20)
    ap.add_argument("--slots", type=int, default=512)
    ap.add_argument("--cache-mb", type=int, default=8192)
    ap.add_argument("--fresh", type=int, default=0, help="measure on this many queries BEHIND the --queries primer, each met for the "
                    "first time by the engine (the index has derived their keys in an untimed pass; the posting cache holds what the "
                    "primer left) — bench.py's keyword stream")
    ap.add_argument("--flags", type=int, default=0, help="rb_prepare_queries_ex flags (1 phrases, 2 short prefixes, 4 synonyms, 8 negatives)")
    ap.add_argument("--sweep", default="", help="'reaper:callers,...' e.g. '1:256,0:256,1:384': each configuration (MSI_VM_REAPER, "
                    "callers taking searches) measured on its own --segment of fresh queries behind the primer; --callers = the largest")
    ap.add_argument("--segment", type=int, default=1536)
    ap.add_argument("--emulated", action="store_true", help="no MI355X: the CPU-emulated build of libmsi (tests/emu) — for HOST CPU "
                    "profiles of the search threads on a small corpus (kernel time means nothing there)")
    ap.add_argument("--stage", type=int, default=1, help="1 (default): the index's word / fid / position / word-count databases are "
                    "staged into HBM when the index opens (msi_dict_stage_postings; coherent corpus only); 0: the engine meets "
                    "every posting through the index callbacks, as before round 6")
    ap.add_argument("--stage-threads", type=int, default=16)
    ap.add_argument("--cold", action="store_true", help="--fresh: no primer pass — the measured queries meet the cache as "
                    "msi_dict_reset_posting_cache leaves it (only what was staged)")
    a = ap.parse_args()
    runner = os.path.join(ROOT, "tools", "bin", "libmsi_rankedbench.so")
    if a.emulated:
        sys.path.insert(0, os.path.join(ROOT, "tests", "emu"))
        import run_emulated as E
        ma._lib._LIB = E.EmulatedLib(E.build())
        runner = E.build_runner()
    L = C.CDLL(runner)
    L.rb_create.restype = C.c_void_p
    L.rb_create.argtypes = [C.c_uint64, C.c_uint32]
    L.rb_create_corpus.restype = C.c_void_p
    L.rb_create_corpus.argtypes = [C.c_uint64, C.c_uint32, C.c_uint64]
    L.rb_attach.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint64]
    L.rb_prepare_queries_ex.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint64, C.c_uint32]
    L.rb_run.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p]
    L.rb_enable_prefix_dbs.argtypes = [C.c_void_p, C.c_uint32]
    L.rb_enable_synonyms.argtypes = [C.c_void_p]
    L.rb_pool.restype = C.c_void_p
    L.rb_pool.argtypes = [C.c_void_p, C.c_uint32]
    L.rb_dict.restype = C.c_void_p
    L.rb_dict.argtypes = [C.c_void_p]
    L.rb_destroy.argtypes = [C.c_void_p]
    lib = ma._lib.lib()
    ctx = ma.Context(0)
    h = L.rb_create_corpus(a.docs, a.words, 42) if a.corpus == "coherent" else L.rb_create(a.docs, a.words)
    if a.flags & 2:
        assert L.rb_enable_prefix_dbs(h, 50) == 0
    if a.flags & 4:
        assert L.rb_enable_synonyms(h) == 0
    assert L.rb_attach(h, ctx.handle, a.callers, a.slots, a.cache_mb) == 0
    staged = None
    if a.stage and a.corpus == "coherent":
        L.rb_stage_postings.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p]
        sec, cnts = C.c_double(0), (C.c_uint64 * 4)()
        st = L.rb_stage_postings(h, a.stage_threads, C.byref(sec), cnts)
        assert st == 0, "rb_stage_postings failed: %d" % st
        staged = {"seconds": round(sec.value, 1), "values": int(cnts[0]), "bodies_in_hbm": int(cnts[1]), "kept_on_host": int(cnts[2]),
                  "stored_bytes_in_hbm": int(cnts[3])}
    sweep = []
    if a.sweep:   # "reaper:callers,..." — every configuration on its own segment of fresh queries, one process, one posting cache
        for item in a.sweep.split(","):
            r_, c_ = item.split(":")
            sweep.append((r_, int(c_)))
        assert max(c_ for _, c_ in sweep) <= a.callers, "--callers must cover the sweep's largest number of callers"
        a.fresh = len(sweep) * a.segment
    total = a.queries + a.fresh
    L.rb_prepare_queries_ex(h, total, a.terms, 4242, a.flags)
    ids = np.zeros((total, a.k), np.uint32)
    cnt = np.zeros(total, np.uint32)
    sc = np.zeros((total, a.k), np.float64)

    def run(first, n):
        assert L.rb_run(h, first, n, a.k, ids.ctypes.data, cnt.ctypes.data, sc.ctypes.data) == 0

    t_d = time.perf_counter()
    run(0, total)                                # untimed: the index derives what the queries read; pools create their companions
    derive_s = time.perf_counter() - t_d
    L.rb_freeze.argtypes = [C.c_void_p]
    assert L.rb_freeze(h) == 0                   # the index's derived databases: read without a lock from here on (as LMDB's are)
    if a.fresh:
        a.passes = 1
        ma._lib.check(lib.msi_dict_reset_posting_cache(C.c_void_p(L.rb_dict(h))))
        if not a.cold:
            run(0, a.queries)                    # the primer: what a serving process has in HBM
    searches_so_far = total + (a.queries if a.fresh and not a.cold else 0)

    def measure(first, n, passes, label=None):
        """`passes` timed passes over queries [first, first + n): one result object"""
        nonlocal searches_so_far
        pc0 = (C.c_uint64 * 4)()
        lib.msi_dict_posting_cache_stats(C.c_void_p(L.rb_dict(h)), pc0)
        cp0, cp1 = (C.c_uint64 * 8)(), (C.c_uint64 * 8)()
        vs0, vs1 = (C.c_uint64 * 6)(), (C.c_uint64 * 6)()
        lib.msi_search_cpu_profile(cp0)
        lib.msi_bits_vm_stats(C.c_void_p(L.rb_pool(h, 0)), vs0)
        prof_path = os.environ.get("KW_PROFILE")      # a sampling CPU profile of the measured passes (tools/r3_symbolize.py reads it)
        if prof_path:
            L.rb_profile_stop.argtypes = [C.c_char_p]
            L.rb_profile_start()
        c0 = os.times()
        t0 = time.perf_counter()
        for _ in range(passes):
            run(first, n)
        dt = time.perf_counter() - t0
        c1 = os.times()
        if prof_path:
            L.rb_profile_stop(prof_path.encode())
        lib.msi_search_cpu_profile(cp1)
        lib.msi_bits_vm_stats(C.c_void_p(L.rb_pool(h, 0)), vs1)
        nq = passes * n
        searches_so_far += nq
        L.rb_last_latencies.restype = C.c_uint32
        L.rb_last_latencies.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32]
        lat = np.zeros(n, np.float64)
        n_lat = L.rb_last_latencies(h, lat.ctypes.data, n)
        pc = (C.c_uint64 * 4)()
        lib.msi_dict_posting_cache_stats(C.c_void_p(L.rb_dict(h)), pc)
        pc = [int(pc[0] - pc0[0]), int(pc[1] - pc0[1]), int(pc[2]), 0]
        cst, lst = (C.c_uint64 * 3)(), (C.c_uint64 * 2)()
        lib.msi_search_compaction_stats(cst)
        lib.msi_search_late_compaction_stats(lst)
        lists, rounds = vs1[1] - vs0[1], vs1[0] - vs0[0]
        out = {"corpus": a.corpus, "docs": a.docs, "dictionary_words": a.words, "callers": a.callers, "distinct_queries": a.queries,
               "measured_searches": nq, "searches_in_this_process": searches_so_far,
               "fresh_stream": bool(a.fresh), "index_derivation_seconds": round(derive_s, 1), "queries_per_s": round(nq / dt, 1),
               "host_cpus_used": round((c1[0] + c1[1] - c0[0] - c0[1]) / dt, 2),
               "p50_ms_at_load": round(float(np.median(lat[:n_lat])), 2) if n_lat else None,
               "vm": {"lists_per_query": round(lists / nq, 2), "lists_per_launch": round(lists / max(1, rounds), 2),
                      "us_queued_per_list": round((vs1[2] - vs0[2]) / 1e3 / max(1, lists), 1),
                      "us_waiting_for_company_per_list": round((vs1[3] - vs0[3]) / 1e3 / max(1, lists), 1),
                      "us_launch_to_wake_up_per_list": round((vs1[5] - vs0[5]) / 1e3 / max(1, lists), 1)},
               "posting_cache": {"hits": int(pc[0]), "misses": int(pc[1]), "bytes_used": int(pc[2]),
                                 "hit_rate": round(pc[0] / max(1, pc[0] + pc[1]), 4)},
               "compact_space": {"searches": int(cst[0]), "universe_compacted": int(cst[1]), "bucket_sub_trees_moved": int(lst[0])}}
        if staged:
            ss = (C.c_uint64 * 4)()
            lib.msi_dict_staged_stats(C.c_void_p(L.rb_dict(h)), ss)
            out["staged_at_index_open"] = dict(staged, absent_answers_from_complete_dbs=int(ss[3]))
        if a.cold:
            out["cold_posting_cache"] = True
        if label:
            out = dict(label, **out)
        if cp1[0] > cp0[0]:
            m = float(cp1[0] - cp0[0])
            d = [(cp1[i] - cp0[i]) / 1e3 / m for i in range(8)]
            out["host_cpu_us_per_query"] = {"search_threads": round(d[1], 1), "list_submit_and_wait": round(d[2], 1),
                                            "of_it_finalising_lists": round(d[3], 1), "typo_derivations": round(d[4], 1),
                                            "index_callbacks": round(d[5], 1), "host_logic": round(d[1] - d[2] - d[4] - d[5], 1),
                                            "combiner_thread": round(d[6], 1), "lists_per_query": round((cp1[7] - cp0[7]) / m, 2)}
        print(json.dumps(out), flush=True)
        return out

    if sweep:
        L.rb_set_active_callers.argtypes = [C.c_void_p, C.c_uint32]
        for i, (reaper, callers) in enumerate(sweep):
            os.environ["MSI_VM_REAPER"] = reaper          # (msi_vm.hip reads it per round)
            L.rb_set_active_callers(h, callers)
            a.callers = callers
            measure(a.queries + i * a.segment, a.segment, 1, {"MSI_VM_REAPER": reaper})
    elif a.fresh:
        measure(a.queries, a.fresh, 1)
    else:
        measure(0, a.queries, a.passes)
    L.rb_destroy(h)


if __name__ == "__main__":
    main()
