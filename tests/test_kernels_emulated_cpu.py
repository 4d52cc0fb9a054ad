"""CPU tier for the KERNELS: the GPU test files run unchanged — same bodies, fixtures, parameters — against libmsi
built from the product's own .hip sources for a CPU emulation of the HIP runtime (tests/emu: fibers for the threads
of a workgroup, wave ballots / shuffles / MFMA with the CDNA fragment layouts, LDS, atomics, the completion protocol
through "pinned" memory).  Every kernel of the library executes here: the vector scan, selection, rescoring and
merge; the dictionary matcher; the docid-set algebra, Roaring decoders, order keys, distinct and GeoSort; the
bit-sliced ranking kernels; and every launch the ranked keyword search enqueues.

TEST INFRASTRUCTURE ONLY (tests/emu/run_emulated.py says how): each group is a subprocess, so the emulated build never
shares a process with the product library.  Deselected: the tests that hand torch CUDA tensors to the library, and
the largest sizes (the emulation runs ~10^5 thread-steps per millisecond)."""
import os
import re
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NEEDS_TORCH_CUDA = ("row_sharded_search_with_device_merge or bits_as_vector_filter or set_from_docid_lists_device "
                    "or hybrid_end_to_end or device_resident or rerank_universes")

GROUPS = {
    # name: (files, -k expression, minimum number of tests that must have run)
    "vector-scan": (["tests/test_vs_gpu.py", "tests/test_zzz_vs_update_gpu.py", "tests/test_zz_bq_gpu.py", "tests/test_zz_i8_proof_gpu.py"],
                    "not 70000 and not 40000 and not 20000 and not three_query_tiles and not large_scan and not large_k", 32),
    "dictionary": (["tests/test_dict_gpu.py", "tests/test_zz_fst_gpu.py"], "not synthetic_dictionary_all_paths", 12),
    # round 6: batches of 512 words and more run dict_other_kernel on a stream of its own beside the range scans and the cap
    # logic in dict_caps_kernel behind both (DictArgs::defer_caps) — here forced for every batch size
    "dictionary-deferred-caps": (["tests/test_dict_gpu.py"], "not synthetic_dictionary_all_paths", 8, {"MSI_DICT_DEFER_CAPS_MIN": "1"}),
    "docid-sets": (["tests/test_bits_gpu.py", "tests/test_zz_order_keys_gpu.py::test_order_next_against_numpy",
                    "tests/test_zzz_distinct_gpu.py::test_distinct_against_the_sequential_loop",
                    "tests/test_zzz_distinct_gpu.py::test_many_calls_share_the_scratch_without_clearing_it",
                    "tests/test_zzz_geo_gpu.py::test_geo_next_against_the_bucket_rule", "tests/test_zzz_filter_gpu.py"],
                   "not 200003 and not 3001", 46),
    "ranking-kernels": (["tests/test_rank_gpu.py"], "not 50000", 8),
    "ranked-search": (["tests/test_search_gpu.py", "tests/test_zz_order_keys_gpu.py::test_sort_rs_snapshots",
                       "tests/test_zzz_distinct_gpu.py::test_reference_snapshots_with_distinct_and_sort_on_the_device",
                       "tests/test_zzz_distinct_gpu.py::test_reference_criteria_tests_on_the_device",
                       "tests/test_zzz_distinct_gpu.py::test_reference_distinct_integration_tests_on_the_device",
                       "tests/test_zzz_distinct_gpu.py::test_reference_typo_tolerance_and_phrase_integration_tests_on_the_device",
                       "tests/test_zzz_distinct_gpu.py::test_concurrent_searches_with_distinct_sort_and_geo",
                       "tests/test_zzz_geo_gpu.py::test_geo_sort_rs_on_the_device"],
                      "not matches_oracle_on_random_corpora and not under_index_settings", 100),
    # the command lists themselves: levels per wait, task scheduling when the pool runs out of slots, documents spread over
    # many chunks, the posting cache cold / warm / off (a group of its own: the groups run side by side)
    "ranked-search-lists": (["tests/test_zz_levels_per_wait_gpu.py", "tests/test_zz_vm_gpu.py"],
                            "not random_corpora_with_levels", 8),
    "ranked-search-vs-oracle": (["tests/test_zzz_distinct_gpu.py::test_distinct_matches_the_oracle_on_the_device",
                                 "tests/test_zzz_geo_gpu.py::test_geo_sort_matches_the_oracle_on_the_device",
                                 "tests/test_zz_order_keys_gpu.py::test_sort_rules_match_the_oracle_on_the_device",
                                 "tests/test_configs_gpu.py::test_c4_keyword_leg",
                                 "tests/test_configs_gpu.py::test_c4_keyword_leg_on_the_coherent_corpus",
                                 "tests/test_configs_gpu.py::test_rerank_inside_candidate_universes_on_the_corpus",
                                 "tests/test_configs_gpu.py::test_postings_staged_at_index_open_on_the_coherent_corpus",
                                 "tests/test_configs_gpu.py::test_phrases_on_the_coherent_corpus",
                                 "tests/test_configs_gpu.py::test_word_prefix_databases_on_the_coherent_corpus",
                                 "tests/test_configs_gpu.py::test_synonyms_on_the_coherent_corpus",
                                 "tests/test_configs_gpu.py::test_negative_terms_on_the_coherent_corpus"], "", 12),
    # every bucket that is ranked further moves into the compact space of ITS bucket (MSI_SEARCH_LATE_COMPACT=2: by default
    # only buckets of a search whose universe was too large to compact, on indexes of more than a chunk): the reference's
    # snapshot searches and the index settings against the oracle through Ctx::late_enter / late_leave — the caches put
    # aside and restored, the sub-tree's tasks joined, the rank tables rebuilt per bucket
    "ranked-search-bucket-space": (["tests/test_search_gpu.py"], "reference_snapshot or under_index_settings or negative_words",
                                   90, {"MSI_SEARCH_LATE_COMPACT": "2"}),
    # two emulated devices (tests/emu/hip/hip_runtime.h MSI_EMU_DEVICES; RCCL = tests/emu/rccl_emu.cpp): the N > 1 host
    # paths of msi_group / msi_vs_group — one context, stream and store per device, one caller thread per device
    # (replicate), rows sharded + ONE packed all-gather + device merge, the per-rank form joined from two threads — run
    # at least once before the driver's 8-GPU node does.  The emulation stops when work is enqueued on a stream of
    # another device than the current one (a forgotten DeviceGuard).
    "multi-device": (["tests/test_zz_group_gpu.py"], "not world_of_one", 5, {"MSI_EMU_DEVICES": "2"}),
    # universe compaction forced on (MSI_SEARCH_COMPACT=2: by default it only engages where it pays, on indexes of more
    # than one chunk): the reference's snapshot searches, the index settings x the three strategies against the oracle,
    # the one-document-per-chunk spread (ranges of one bit: the shared-word path of VM_DECODEC), the out-of-slots re-run
    "ranked-search-compact-space": (["tests/test_search_gpu.py", "tests/test_zz_vm_gpu.py"],
                                    "not matches_oracle_on_random_corpora and not starved", 100, {"MSI_SEARCH_COMPACT": "2"}),
    # the decode phase BY RANK (round 6: a compact list's postings probed from the universe's side, one workgroup per 64
    # documents of U0, instead of the wide phase's workgroup per chunk of the full space) forced wherever a compact list has
    # no cache fill to do — the test corpora are 1-3 chunks long, so the default rule (fewer workgroups than chunks) would
    # only take it for universes of <= 128 documents: the vm tests and the corpus tests (array / bitmap containers, staged and
    # first-read postings, phrases, prefix databases; universe and bucket spaces as the search chooses them).  The forced
    # compact space of every search (MSI_SEARCH_COMPACT=2) with it: tools/fuzz_ranked_hostlogic.py --emulated-kernels, 99 k cases
    "ranked-search-by-rank": (["tests/test_search_gpu.py", "tests/test_zz_vm_gpu.py",
                               "tests/test_configs_gpu.py::test_c4_keyword_leg_on_the_coherent_corpus",
                               "tests/test_configs_gpu.py::test_postings_staged_at_index_open_on_the_coherent_corpus",
                               "tests/test_configs_gpu.py::test_phrases_on_the_coherent_corpus",
                               "tests/test_configs_gpu.py::test_word_prefix_databases_on_the_coherent_corpus"],
                              "not matches_oracle_on_random_corpora and not starved", 100,
                              {"MSI_VM_BY_RANK_FORCE": "1", "MSI_VM_BY_RANK_MAX_DOCS": "1000000"}),
    # the reaper (MSI_VM_REAPER=1: rounds handed to a second thread that notices their completion and wakes the searches
    # — an experiment that stays off by default): the hand-over, concurrent searches, shutdown
    "ranked-search-reaper": (["tests/test_zz_vm_gpu.py"], "concurrent_searches or cold_and_warm or universe_of_an_unfiltered or index_views",
                             4, {"MSI_VM_REAPER": "1"}),
}


_running = {}


def _build_once():
    sys.path.insert(0, os.path.join(ROOT, "tests", "emu"))
    import run_emulated
    run_emulated.build()
    run_emulated.build_runner()
    run_emulated.build_rccl()


def _start_all(selected):
    """Every group is its own subprocess; the groups this session runs (`-k` may have selected some) are all started when
    the first one is asked for (the emulated library is built once, before, so that they do not race for it) and run
    side by side — the tier takes as long as its longest group instead of their sum.  What is still running when the
    session ends is stopped."""
    if _running:
        return
    import atexit
    import tempfile
    _build_once()

    def _stop_leftovers():
        for proc, _, _ in _running.values():
            if proc.poll() is None:
                proc.kill()
    atexit.register(_stop_leftovers)
    for group, spec in GROUPS.items():
        if group not in selected:
            continue
        files, expr = spec[:2]
        env = dict(os.environ, **(spec[3] if len(spec) > 3 else {}))
        k = f"not ({NEEDS_TORCH_CUDA})" + (f" and {expr}" if expr else "")
        out = tempfile.TemporaryFile(mode="w+")
        err = tempfile.TemporaryFile(mode="w+")
        proc = subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "emu", "run_emulated.py"), "-q", "-x", "-m", "gpu",
                                 "-p", "no:cacheprovider", "-k", k] + files, cwd=ROOT, stdout=out, stderr=err, text=True, env=env)
        _running[group] = (proc, out, err)


@pytest.mark.parametrize("group", list(GROUPS))
def test_gpu_test_bodies_on_emulated_kernels(group, request):
    at_least = GROUPS[group][2]
    _start_all({it.callspec.params["group"] for it in request.session.items
                if getattr(it, "callspec", None) is not None and "group" in it.callspec.params})
    proc, out, err = _running[group]
    try:
        proc.wait(timeout=1500)
    except subprocess.TimeoutExpired:
        proc.kill()
        raise
    out.seek(0)
    err.seek(0)
    stdout, stderr = out.read(), err.read()
    tail = stdout[-3000:] + stderr[-2000:]
    assert proc.returncode == 0, tail
    m = re.search(r"(\d+) passed", stdout)
    assert m and int(m.group(1)) >= at_least, tail
    assert " failed" not in stdout and " error" not in stdout, tail


# Searches the differential fuzzer (tools/fuzz_ranked_hostlogic.py --emulated-kernels) once got wrong, replayed: (first
# seed argument, the query of that seed, environment).  FUZZ_ONLY runs that one query of the seed's six.
FUZZ_REGRESSIONS = [
    # 17 first-k commands in a phase of 16 (Dev::first_k_later checked its limits one after the other across a task park):
    # the ids of a leaf bucket came back as zeros
    (12003396, "sun delicious the brown sweet", {"MSI_SEARCH_LEVELS_PER_WAIT": "16"}),
    (12003396, "sun delicious the brown sweet", {"MSI_SEARCH_LEVELS_PER_WAIT": "16", "MSI_SEARCH_COMPACT": "2"}),
    # ... and with the default 8 levels per wait (pages of 100 hits: many leaf buckets per round)
    (13004424, "flower sun delicious sweet", {}),
    (14000465, "summer dogs interconnection", {"MSI_SEARCH_COMPACT": "2", "FUZZ_SPREAD": "2200"}),
    (14000984, "sun summer sunflowlr interconnection dessert ", {"MSI_SEARCH_COMPACT": "2", "FUZZ_SPREAD": "2200"}),
]


@pytest.mark.parametrize("case", FUZZ_REGRESSIONS, ids=[f"{c[0] + 1}:{'+'.join(c[2].values()) or 'default'}" for c in FUZZ_REGRESSIONS])
def test_fuzz_regression_seeds(case):
    seed0, query, extra = case
    _build_once()
    env = dict(os.environ, FUZZ_ONLY=query, **extra)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "fuzz_ranked_hostlogic.py"), str(seed0), "1", "--emulated-kernels"],
                         cwd=ROOT, capture_output=True, text=True, timeout=600, env=env)
    assert out.returncode == 0 and "cases 1 bad 0" in out.stdout, out.stdout[-2000:] + out.stderr[-2000:]


def test_the_keyword_leg_tool_sweeps_configurations_in_one_process():
    """tools/kw_leg.py --emulated --sweep: the measurement tool of round 5's A/B (profiles/r5_reaper.log) on a small corpus
    through the emulated kernels — every configuration (MSI_VM_REAPER read per round, callers through rb_set_active_callers)
    on its own segment of fresh queries, one JSON object each, the host CPU profile of the search threads filled in."""
    import json
    env = dict(os.environ, MSI_SEARCH_CPU_PROFILE="1")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "kw_leg.py"), "--emulated", "--docs", "20000", "--words", "8000",
                          "--callers", "4", "--queries", "32", "--segment", "16", "--sweep", "1:2,0:4", "--cache-mb", "64"],
                         cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [json.loads(l) for l in out.stdout.splitlines() if l.startswith("{")]
    assert [(l["MSI_VM_REAPER"], l["callers"]) for l in lines] == [("1", 2), ("0", 4)]
    for l in lines:
        assert l["measured_searches"] == 16 and l["fresh_stream"] and l["queries_per_s"] > 0
        assert l["vm"]["lists_per_query"] > 1 and l["host_cpu_us_per_query"]["search_threads"] > 0
