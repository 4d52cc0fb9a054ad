#!/usr/bin/env python
"""bench.py — milli's query-time scoring path on MI355X, one JSON line per run.

    python bench.py [--config c4|c2|c3|c5|c1] [--gpus N --steps K --warmup W]

BASELINE.json metric: queries/sec + p50 latency, 10 M-doc index (768-d hybrid / 2-typo), 1 -> 8 MI355X.
The default (`--config c4`, the line the driver records) is the configuration the metric is quoted on; it fits
one GPU (30.72 GB of 288 GB), so every rank holds a replica and answers its own query stream:

  c4  one "step" = one batch of 768 hybrid queries per GPU through the hot path, inputs resident in HBM:
        * 768 query vectors -> exact cosine top-20 over the 10 M x 768 f32 store (vs_scan: one HBM sweep per
          msi_vs_max_batch() = 96 queries; candidates rescored with the reference arithmetic + exactness proof);
        * 1536 query words -> typo derivations (1 / 2 typos by char count, 30 % prefix) over the 2 M-term
          dictionary (dict_lookup) on the context's second stream;
        * keyword leg: msi_keyword_search_ranked for every query — all 7 default criteria, detailed scores (what
          hybrid search asks for), 64 caller threads over a 10 M-document synthetic inverted index whose typo
          derivations come from its own 200 k-word dictionary — and the hybrid merge (semanticRatio 0.5);
        * results copied to the host.
      `legs` of the line: each leg on its own (untimed extras), the scan kernel's roofline fraction without the
      keyword lists running beside it, host CPUs used by the keyword leg.
  c2  1 M x 384 f32, cosine top-20, 256 queries per step                       (SURVEY §8 d)
  c3  2 M-term dictionary, 8 192 query words per step (1 / 2 typos, 30 % prefix)              (C3; unit words/s)
  c5  one GPU's shard of config 5: 12.5 M x 1024 bf16 rows, 1 % candidate filter, k = 1000, Words -> Typo
      rerank of each top-1000, 32 queries per step                                                        (C5)
  c1  keyword-only plumbing on a 32 k-document synthetic corpus: the CPU oracle end to end (the reported
      baseline) beside msi_keyword_search_ranked on the same index, hits compared                          (C1)

Every line carries `roofline` (dominant kernel, HIP events on its launch stream, algorithmic bytes; `traffic` =
HBM bytes per launch measured LIVE by a 2-step child of this script under `rocprofv3 --pmc FETCH_SIZE`, gfx950
correction x2 per the microarch guide), `cpu_baseline` (the CPU restatement oracle/msi_cpubase.c, kind "port",
on this box's host threads, bounded sample) and `parity` (an UNTIMED post-run check of the step's own results
against the oracle, oracle/parity.py: part of the cpu_baseline leg, rank 0, N = 1).

N GPUs: one process per GPU (launched by torch.distributed.run); queries are sharded, the index is replicated
("scaling": "weak"); per-rank top-k lists travel in ONE packed all-gather per step, issued by libmsi itself
(msi_group_create_rank + msi_group_allgather: RCCL inside the library; the launcher's process group only carries the
128-byte communicator id).  `--shard rows` (c4) shards
the rows instead (strong scaling): same batch on every rank, all-gather of packed (distance, docid, count) + device
k-way merge.
"""
import argparse
import json
import os
import statistics
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
# The command-list rounds of the keyword leg run on 16 streams; the HIP runtime maps streams onto GPU_MAX_HW_QUEUES hardware
# queues (default 4) and rounds that share a queue serialise: 4 -> 16 queues took the keyword leg from 2.8 k to 6.1 k
# queries/s (profiles/r3_bench_variants.txt).  Must be set before the runtime starts (import torch); a deployment sets it
# in the server's environment.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")


_PHASES = []


_HBM_FREE = []


def note_hbm(torch_mod):
    """Free HBM right now (the whole device: hipMemGetInfo) — the line carries the minimum over the marks of the run."""
    try:
        _HBM_FREE.append(torch_mod.cuda.mem_get_info()[0] / 1e9)
    except Exception:      # noqa: BLE001 - the emulated tier has no device to ask
        pass


def phase(name):
    """Wall-clock marks of the run (the detail object's `phase_seconds`: where the default command's minutes go)."""
    _PHASES.append((name, time.time()))


def phase_seconds():
    out = {}
    for (name, t0), (_, t1) in zip(_PHASES, _PHASES[1:] + [("end", time.time())]):
        out[name] = round(out.get(name, 0.0) + (t1 - t0), 1)
    return out


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", choices=["c1", "c2", "c3", "c4", "c5"], default="c4")
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--rows", type=int, default=None)
    ap.add_argument("--dim", type=int, default=None)
    ap.add_argument("--k", type=int, default=None)
    ap.add_argument("--queries", type=int, default=None, help="queries (c3: words) per step per GPU")
    ap.add_argument("--words-per-query", type=int, default=2)
    ap.add_argument("--dict-words", type=int, default=2_000_000)
    ap.add_argument("--storage", choices=["f32", "bf16"], default=None,
                    help="row storage in HBM (f32 = the reference's; bf16 = BASELINE.json config 5's build-side choice)")
    ap.add_argument("--shard", choices=["queries", "rows"], default="queries")
    ap.add_argument("--no-typo", action="store_true")
    ap.add_argument("--no-rank", action="store_true", help="c4: leave the keyword leg and the hybrid merge out")
    ap.add_argument("--kw-slots", type=int, default=512, help="c4: slots of every caller's docid-set pool (n_docs / 8 bytes each)")
    ap.add_argument("--kw-threads", type=int, default=256, help="c4: caller threads of the keyword leg (one in-flight search each)")
    ap.add_argument("--kw-terms", type=int, default=3, help="c4: words per keyword query")
    ap.add_argument("--kw-dict-words", type=int, default=None, help="c4: vocabulary of the keyword leg's index (default: 2 000 000 "
                                                                   "for the coherent corpus, 200 000 for the hashed index)")
    ap.add_argument("--kw-corpus", choices=["coherent", "hashed"], default="coherent",
                    help="c4: the keyword leg's index — BASELINE's C4 text workload (a coherent corpus: every database derived from the "
                         "same documents, queries taken out of them and misspelled) or round 3's independently hashed postings with "
                         "3-word queries over the 300 most frequent words")
    ap.add_argument("--kw-cache-mb", type=int, default=8192, help="c4: HBM posting cache of the index version")
    ap.add_argument("--kw-stage", type=int, default=1, help="c4, coherent corpus: 1 (default) = the index's word_docids / word_fid_docids / "
                    "word_position_docids / field_id_word_count_docids are staged into the HBM posting cache when the index opens "
                    "(msi_dict_stage_postings: the north_star's 'staged once into HBM'); 0 = every posting reaches the engine "
                    "through the index callbacks on first use, as in rounds 1-5")
    ap.add_argument("--kw-features", type=int, default=1, help="c4, coherent corpus: 1 (default) = after the timed steps, one more keyword "
                    "leg on the same index with the features a real index has — word-prefix databases (one- to three-letter "
                    "prefix queries such as workloads/search/movies.json's \"t\"), synonyms, quoted phrases, negative terms — "
                    "on fresh queries, checked against the oracle: legs.keyword_with_features")
    ap.add_argument("--kw-stream", choices=["fresh", "cycle"], default="fresh",
                    help="c4: the keyword queries of the steps — fresh (default): no query is met twice, the posting cache holds what "
                         "4 x Q primer queries of the same generator left behind; cycle: round 4's stream (the 4 x Q primer queries cycled)")
    ap.add_argument("--legs", choices=["tail", "serial", "overlap"], default="serial",
                    help="c4: how the two legs of a step share the device.  serial (default): the scan streams HBM on its own (0.73 of "
                         "peak), then the keyword leg; overlap: both from the start; tail: the keyword leg first, the scan starts "
                         "when --tail-at of its searches are done.  All three measured within 3 %% of each other (152-157 ms per "
                         "step, profiles/r3_bench_variants.txt): a step is the SUM of its legs — the scan saturates HBM and "
                         "stretches every keyword round beside it by as much as it gains")
    ap.add_argument("--sweep-split", type=int, default=16, help="c4: msi_vs_set_sweep_split for the side-by-side leg (and for the timed "
                    "steps under --legs overlap / tail)")
    ap.add_argument("--no-overlapped-leg", action="store_true", help="c4: skip legs.hybrid_legs_side_by_side")
    ap.add_argument("--kw-gap-probe", action="store_true", help="c4: legs.keyword_gap_probe (the keyword leg behind idle time / behind the vector leg)")
    ap.add_argument("--tail-at", type=float, default=0.8, help="c4, --legs tail: the fraction of the step's keyword searches done when the scan starts")
    ap.add_argument("--no-cpu-baseline", action="store_true", help="also skips the parity check (same leg)")
    ap.add_argument("--no-rows-sharded-extra", action="store_true", help="c4, N > 1: skip the rows-sharded extra object")
    ap.add_argument("--extra-timeout", type=float, default=240.0, help="c4, N > 1: seconds the rows-sharded extra may take before "
                    "the line is printed without it")
    ap.add_argument("--no-also", action="store_true", help="c4: do not run the short C2 / C3 / C5 legs after the C4 line")
    ap.add_argument("--no-pmc", action="store_true", help="do not spawn the rocprofv3 --pmc child for roofline.traffic")
    ap.add_argument("--kw-roofline", action="store_true", help="c4: also the keyword leg's counters (four children on the headline's "
                                                               "corpus: minutes; profiles/ keeps the builder's run)")
    ap.add_argument("--cpu-sample-rows", type=int, default=200_000)
    ap.add_argument("--cpu-sample-words", type=int, default=4096)
    ap.add_argument("--parity-queries", type=int, default=96, help="vector queries checked against the oracle (96 = one whole 6-tile sweep)")
    ap.add_argument("--parity-kw-queries", type=int, default=64, help="c4: keyword searches of the timed step checked against the ranking oracle")
    return ap.parse_args()


# ----------------------------------------------------------------------------------------------- harness

def _emulate_the_device(torch):
    """MSI_BENCH_EMULATED=1 (the CPU tier's two-rank run, tests/test_bench_two_ranks_cpu.py — never a measurement): libmsi is the
    CPU-emulated build of tests/emu (every csrc/*.hip compiled as plain C++: "device" pointers are host pointers), torch
    tensors live on the CPU, the process group is gloo and RCCL is the stand-in of tests/emu/rccl_emu.cpp joined through
    shared memory.  What runs is the HOST side of the N > 1 line — sharding, the exchange through msi_group_*, the merge, the
    line's fields — on sizes the emulation finishes in a minute."""
    sys.path.insert(0, os.path.join(ROOT, "tests", "emu"))
    import run_emulated as E
    import meilisearch_amd as ma
    ma._lib._LIB = E.EmulatedLib(E.build())
    os.environ["MSI_RUNNER_SO"] = E.build_runner()
    os.environ["MSI_RCCL_LIBRARY"] = E.build_rccl()
    os.environ["MSI_RCCL_EMU_SHM"] = "1"

    class _Stream:
        cuda_stream = 0

        def synchronize(self):
            pass
    torch.cuda.set_device = lambda *a, **k: None
    torch.cuda.synchronize = lambda *a, **k: None
    torch.cuda.empty_cache = lambda *a, **k: None
    torch.cuda.current_stream = lambda *a, **k: _Stream()
    return E


class Env:
    def __init__(self, args):
        import torch
        self.torch = torch
        self.rank = int(os.environ.get("RANK", "0"))
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        self.dist = None
        self.emulated = os.environ.get("MSI_BENCH_EMULATED") == "1"
        if self.emulated:
            self.emu = _emulate_the_device(torch)
        if self.world > 1:
            import torch.distributed as dist
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
            if self.emulated:
                dist.init_process_group("gloo")
            else:
                torch.cuda.set_device(self.local_rank)
                dist.init_process_group("nccl", device_id=torch.device("cuda", self.local_rank))
            self.dist = dist
        assert self.world == args.gpus or self.world == 1, f"--gpus {args.gpus} but WORLD_SIZE={self.world}"
        self.dev = torch.device("cpu") if self.emulated else torch.device("cuda", self.local_rank)
        torch.cuda.set_device(self.dev)
        import meilisearch_amd as ma
        self.ma = ma
        self.ctx = ma.Context(self.local_rank)
        self.child = os.environ.get("MSI_BENCH_CHILD") == "1"
        self.check = self.rank == 0 and self.world == 1 and not args.no_cpu_baseline and not self.child

    def sync_all(self):
        self.ctx.synchronize()
        self.torch.cuda.synchronize()
        if self.dist is not None:
            self.dist.barrier()

    def timed(self, step, steps, warmup):
        """W untimed steps, then exactly K steps between barrier + device synchronisation on both sides; the
        slowest rank's time counts."""
        for _ in range(warmup):
            step()
        self.sync_all()
        lat = []
        t0 = time.perf_counter()
        for _ in range(steps):
            s0 = time.perf_counter()
            step()
            lat.append((time.perf_counter() - s0) * 1e3)
        self.sync_all()
        elapsed = time.perf_counter() - t0
        self.per_rank_elapsed = [elapsed]
        if self.dist is not None:
            mine = self.torch.tensor([elapsed], dtype=self.torch.float64, device=self.dev)
            every = self.torch.zeros(self.world, dtype=self.torch.float64, device=self.dev)
            self.dist.all_gather_into_tensor(every, mine)          # every rank's own clock (the line reports them beside the MAX)
            self.per_rank_elapsed = [float(x) for x in every.cpu().tolist()]
            elapsed = max(self.per_rank_elapsed)
        return elapsed, lat

    def gather_scalar(self, x):
        """One float per rank -> the list of them on every rank (N = 1: [x])."""
        if self.dist is None:
            return [float(x)]
        mine = self.torch.tensor([float(x)], dtype=self.torch.float64, device=self.dev)
        every = self.torch.zeros(self.world, dtype=self.torch.float64, device=self.dev)
        self.dist.all_gather_into_tensor(every, mine)
        return [float(v) for v in every.cpu().tolist()]

    def finish(self):
        if self.dist is not None:
            self.dist.barrier()
            self.dist.destroy_process_group()


def pmc_traffic(args, kernel_prefix, must_contain=()):
    """roofline.traffic, measured live: this script again as a 2-step child under `rocprofv3 --pmc FETCH_SIZE`
    (counters in their own pass, no tracing domains), FETCH_SIZE [KB] x 1024 x 2 — on gfx950 the counter tallies
    the 128-byte requests of a wide coalesced stream at 64 bytes (microarch guide, HBM section).  Returns
    (GB per launch | None, source string)."""
    exe = None
    for cand in ("rocprofv3", "/opt/rocm/bin/rocprofv3"):
        try:
            subprocess.run([cand, "--version"], capture_output=True, timeout=60)
            exe = cand
            break
        except Exception:
            continue
    if exe is None:
        return None, "rocprofv3 not found on this box"
    out_dir = tempfile.mkdtemp(prefix="msi_pmc_", dir="/tmp")
    cmd = [exe, "--pmc", "FETCH_SIZE", "--output-format", "csv", "-d", out_dir, "-o", "pmc", "--",
           sys.executable, os.path.abspath(__file__), "--config", args.config, "--steps", "2", "--warmup", "1",
           "--no-cpu-baseline", "--no-pmc", "--kw-threads", "16"]   # (the parent still holds its pools: a child of the
    # same size would not fit beside it; the counter only needs the scan kernel's dispatches)
    for flag, val in (("--rows", args.rows), ("--dim", args.dim), ("--k", args.k), ("--queries", args.queries),
                      ("--storage", args.storage)):
        if val is not None:
            cmd += [flag, str(val)]
    if args.no_typo or args.config == "c4":      # (c4: the child is there for the scan kernel's counters only)
        cmd.append("--no-typo")
    if args.no_rank or args.config == "c4":
        cmd.append("--no-rank")
    env = dict(os.environ, MSI_BENCH_CHILD="1", TMPDIR="/tmp")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    try:
        r = subprocess.run(cmd, cwd="/tmp", env=env, capture_output=True, text=True, timeout=600)
    except Exception as e:                       # noqa: BLE001
        return None, f"rocprofv3 child failed: {e!r}"
    import csv
    import glob
    vals = []
    for path in glob.glob(os.path.join(out_dir, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(path)):
            name = row.get("Kernel_Name", "")
            if kernel_prefix in name and all(m in name for m in must_contain) and row.get("Counter_Name") == "FETCH_SIZE":
                vals.append(float(row["Counter_Value"]))
    if not vals:
        return None, f"rocprofv3 --pmc FETCH_SIZE child produced no rows for {kernel_prefix} (rc {r.returncode})"
    # the full sweeps are the largest launches of that kernel (the sample pass reads a small fraction)
    top = max(vals)
    full = [v for v in vals if v > 0.5 * top]
    gb = sum(full) / len(full) * 1024 * 2 / 1e9
    return round(gb, 3), (f"live: rocprofv3 --pmc FETCH_SIZE child of this run ({len(full)} launches of {kernel_prefix}); "
                          "FETCH_SIZE[KB] x 1024 x 2 (gfx950 wide-load correction, MI355X_MICROARCH.md HBM section)")


def sweep_kernel(stats, n_rows):
    """The dominant kernel of a store's level-0 search and its algorithmic bytes per sweep: the int8 candidate sweep when the
    store keeps the int8 copy of its rows (dpad + 8 bytes per row: the int8 row, its scale, its inverse norm), else the sweep
    over the stored rows (dpad x 4 bytes per f32 row).  -> (display name, kernel-name prefix for the counters, bytes)"""
    tiles = (n_rows + 15) // 16
    if stats.get("i8_bytes_per_tile"):
        return "vs_scan_i8_kernel (main pass over the int8 copy of the rows)", "vs_scan_i8_kernel", tiles * stats["i8_bytes_per_tile"]
    return "vs_scan_kernel (main pass)", "vs_scan_kernel", tiles * stats["bytes_per_tile"]


def scan_roofline(args, env, store, scan_n, scan_ms, algo_bytes, kernel_name, must_contain=(), kernel_prefix="vs_scan_kernel",
                  measured=None):
    scan_avg_ms = scan_ms / max(1, scan_n)
    achieved = algo_bytes / (scan_avg_ms * 1e-3) / 1e9 if scan_n else 0.0
    traffic, src = None, "not measured (--no-pmc, child run, or N > 1)"
    if measured is not None:
        traffic, src = measured
    elif env.rank == 0 and env.world == 1 and not args.no_pmc and not env.child:
        traffic, src = pmc_traffic(args, kernel_prefix, must_contain)
    n_rows = len(store)
    return {"kernel": kernel_name, "bound": "hbm", "achieved": round(achieved, 1), "peak": 8000.0, "unit": "GB/s",
            "bytes_per_row": round(algo_bytes / n_rows, 1) if n_rows else None,
            "bytes_per_row_is": "algorithmic bytes of one launch / rows of the store: dpad + 8 for the int8 sweep (the int8 row, its scale, "
                                "its inverse norm), dpad x 4 for a sweep of f32 rows (SURVEY 8 d), dpad x 2 for bf16 rows; a filtered "
                                "sweep streams only the tiles that hold an allowed row",
            "frac": round(achieved / 8000.0, 4), "traffic": traffic, "traffic_unit": "GB per launch (HBM reads, PMC)",
            "traffic_source": src, "algorithmic_bytes_per_launch": int(algo_bytes),
            "avg_launch_ms": round(scan_avg_ms, 4), "launches_timed": scan_n,
            "timing": "HIP events on the kernel's launch stream, inside the timed region of this run"}


def granted_cpus():
    """CPUs this process may use: the affinity mask, capped by the cgroup CPU quota (cpu.max)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = max(1, min(n, -(-int(quota) // int(period))))
    except (OSError, ValueError):
        pass
    return n


def cpu_thread_counts():
    """The two thread counts every CPU baseline is timed at: what the container's CPU quota grants (oracle/cpubase.py:
    host_threads) and every hardware thread the process may run on.  The better of the two is the reported baseline."""
    from oracle import cpubase
    quota = cpubase.host_threads()
    visible = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    return sorted({quota, visible})


def cpu_vector_baseline(cpu_rows, n, d, k, filter_frac=1.0):
    """-> (seconds per query at the full size [best thread count], threads used, sample rows, {threads: queries/s})"""
    from meilisearch_amd import synth
    from oracle import cpubase
    sample = cpu_rows.shape[0]
    scan = cpubase.CpuVectorScan(cpu_rows, np.arange(sample, dtype=np.uint32))
    q = synth.make_embeddings(16, d, seed=5678)
    by_threads = {}
    for threads in cpu_thread_counts():
        scan.search(q[:2], k, threads=threads)  # warm
        t0 = time.perf_counter()
        scan.search(q, min(k, sample), threads=threads)
        t_vec_sample = (time.perf_counter() - t0) / 16.0
        by_threads[threads] = t_vec_sample * (n * filter_frac / sample)  # exact scan is linear in the rows it visits
    cores = min(by_threads, key=by_threads.get)
    return by_threads[cores], cores, sample, {str(t): round(1.0 / v, 3) for t, v in by_threads.items()}


def cpu_typo_baseline(words, concat, off, n_sample):
    """-> (seconds per word [best thread count], seconds per word on one thread, threads used, {threads: words/s})"""
    from meilisearch_amd import synth
    from oracle import cpubase
    cdict = cpubase.CpuDictionary(concat, off)
    tq = synth.make_typo_queries(words, n_sample, seed=7)
    qb, qoff, qfl = cpubase.pack_queries(tq)
    by_threads = {}
    for threads in cpu_thread_counts():
        cdict.lookup_packed(qb[:], qoff[:65], qfl[:64], threads=threads)  # warm
        t0 = time.perf_counter()
        cdict.lookup_packed(qb, qoff, qfl, threads=threads)
        by_threads[threads] = (time.perf_counter() - t0) / len(tq)
    t0 = time.perf_counter()
    cdict.lookup_packed(qb[:], qoff[:129], qfl[:128], threads=1)
    t_one = (time.perf_counter() - t0) / 128
    cores = min(by_threads, key=by_threads.get)
    return by_threads[cores], t_one, cores, {str(t): round(1.0 / v, 1) for t, v in by_threads.items()}


def cpu_keyword_baseline(n_docs, dict_words, n_terms, universe=0):
    """tools/bin/ranked_bench_cpu as a child: the keyword leg's CPU port on a bounded sample (16 distinct queries of the same
    shape as the step's, two per thread after one warm-up pass that also generates the synthetic index's postings).
    universe > 0: every search ranks inside a candidate universe of that many documents (config 5's rerank of a top-k)."""
    exe = os.path.join(ROOT, "tools", "bin", "ranked_bench_cpu")
    if not os.path.exists(exe):
        return {"note": "tools/bin/ranked_bench_cpu not built (__graft_entry__.build())"}
    threads = granted_cpus()
    env = dict(os.environ, RB_DETAILED="1", RB_DISTINCT_QUERIES="16")
    if universe:
        env["RB_UNIVERSE"] = str(int(universe))
    try:
        r = subprocess.run([exe, str(n_docs), str(dict_words), str(n_terms), "2", str(threads)], env=env, capture_output=True,
                           text=True, timeout=300)
        line = json.loads([x for x in r.stdout.splitlines() if x.startswith("{")][-1])
    except Exception as e:     # noqa: BLE001
        return {"note": f"child run failed: {e!r}"}
    return {"queries_per_s": line["queries_per_s"], "p50_ms": line["p50_ms"], "threads": threads, "kind": "port",
            "sample": f"{2 * threads} searches ({threads} threads x 2) of 16 distinct {n_terms}-word queries over the {n_docs}-document "
                      "synthetic index, default criteria, detailed scores; same host logic as the product over dense host bitsets "
                      "(tests/hostlogic/mock_device.cpp) and the CPU dictionary walk (oracle/msi_cpubase.c)"}


def rocprof_exe():
    for cand in ("rocprofv3", "/opt/rocm/bin/rocprofv3"):
        try:
            subprocess.run([cand, "--version"], capture_output=True, timeout=60)
            return cand
        except Exception:     # noqa: BLE001
            continue
    return None


def pmc_rows(cmd, counters, kernel_substr, env=None, timeout=600):
    """Runs `cmd` under `rocprofv3 --pmc <counters>` (counters in their own pass, no tracing domain) and returns
    {counter: [values of the dispatches whose kernel name holds kernel_substr]} (None when nothing came back)."""
    exe = rocprof_exe()
    if exe is None:
        return None
    import csv
    import glob
    out_dir = tempfile.mkdtemp(prefix="msi_pmc_", dir="/tmp")
    full = [exe, "--pmc"] + list(counters) + ["--output-format", "csv", "-d", out_dir, "-o", "pmc", "--"] + cmd
    e = dict(os.environ if env is None else env, TMPDIR="/tmp")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        e.pop(k, None)
    try:
        subprocess.run(full, cwd="/tmp", env=e, capture_output=True, text=True, timeout=timeout)
    except Exception:     # noqa: BLE001
        return None
    vals = {}
    for path in glob.glob(os.path.join(out_dir, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(path)):
            if kernel_substr in row.get("Kernel_Name", ""):
                vals.setdefault(row.get("Counter_Name"), []).append(float(row["Counter_Value"]))
    import shutil
    shutil.rmtree(out_dir, ignore_errors=True)
    return vals or None


def keyword_roofline(n_docs, kw_threads, measured_qps, dict_words=2_000_000, corpus="coherent"):
    """What bounds the keyword leg (vm_kernel, msi_vm.hip + the host logic of msi_search.hip), from children of this run:
    tools/kw_leg.py — the SAME corpus, query generator and caller threads as the timed step (round 4's children ran the
    round-3 hashed index: VERDICT r4 weak #2) — runs plain with MSI_SEARCH_CPU_PROFILE=1 (host CPU per query by where it
    is spent, lists per query, microseconds per round), then under `rocprofv3 --pmc FETCH_SIZE`, `--pmc WRITE_SIZE` and
    `--pmc TCC_HIT_sum TCC_MISS_sum` (separate passes), the counters summed over the vm_kernel dispatches of the process.
    FETCH_SIZE x 2 is the guide's gfx950 correction for wide coalesced loads; WRITE_SIZE is uncalibrated there and is
    reported as counted.  The object does NOT claim an HBM roofline for this leg: neither HBM nor L2 bandwidth binds it
    (`hbm_frac` says by how much); what binds is the chain of dependent rounds per query and the host CPU that drives them.
    Opt-in (`--kw-roofline`): four children that each build the 10 M-document corpus — minutes, not part of the default run."""
    tool = os.path.join(ROOT, "tools", "kw_leg.py")
    threads = max(1, min(kw_threads, 160))
    distinct, passes = 1536, 1
    cmd = [sys.executable, tool, "--docs", str(n_docs), "--words", str(dict_words), "--callers", str(threads),
           "--queries", str(distinct), "--passes", str(passes), "--corpus", corpus]
    env = dict(os.environ)
    out = {"kernel": "vm_kernel (command lists of msi_keyword_search_ranked)",
           "bound": "latency of dependent rounds + host CPU (not a bandwidth roofline: see hbm_frac)",
           "workload": f"the headline's own: {corpus} corpus, {n_docs} documents, {dict_words}-word vocabulary, {threads} callers, "
                       f"{distinct} distinct queries (tools/kw_leg.py)"}
    try:
        r = subprocess.run(cmd, env=dict(env, MSI_SEARCH_CPU_PROFILE="1"), capture_output=True, text=True, timeout=900)
        line = json.loads([x for x in r.stdout.splitlines() if x.startswith("{")][-1])
    except Exception as e:     # noqa: BLE001
        out["note"] = f"child run failed: {e!r}"
        return out
    out["child"] = line
    host = line.get("host_cpu_us_per_query")
    cpus = granted_cpus()
    if host:
        per_q = (host["search_threads"] + host["combiner_thread"]) * 1e-6
        out["host_cpu_ceiling_queries_per_s_on_the_granted_cpus"] = round(cpus / per_q, 1)
        out["granted_cpus"] = cpus
    searches = line["searches_in_this_process"]     # the counters below are summed over every vm_kernel dispatch of a child
    traffic = {}
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
        rows = pmc_rows(cmd, [counter], "vm_kernel", env=env, timeout=900)
        traffic[counter] = None if not rows or counter not in rows else \
            sum(rows[counter]) / searches * 1024 * (2 if counter == "FETCH_SIZE" else 1)
    tcc = pmc_rows(cmd, ["TCC_HIT_sum", "TCC_MISS_sum"], "vm_kernel", env=env, timeout=900)
    if tcc and tcc.get("TCC_HIT_sum") and tcc.get("TCC_MISS_sum"):
        hit, miss = sum(tcc["TCC_HIT_sum"]), sum(tcc["TCC_MISS_sum"])
        out["l2_counters"] = {"TCC_HIT_per_query": round(hit / searches, 1), "TCC_MISS_per_query": round(miss / searches, 1),
                              "hit_rate": round(hit / max(1.0, hit + miss), 4)}
    out["hbm_traffic_mb_per_query"] = {"reads": None if traffic["FETCH_SIZE"] is None else round(traffic["FETCH_SIZE"] / 1e6, 2),
                                       "writes": None if traffic["WRITE_SIZE"] is None else round(traffic["WRITE_SIZE"] / 1e6, 2),
                                       "source": "rocprofv3 --pmc FETCH_SIZE (x 2, gfx950 wide-load correction) / --pmc WRITE_SIZE children, "
                                                 f"summed over every vm_kernel dispatch, / {searches} searches"}
    moved = (traffic["FETCH_SIZE"] or 0.0) + (traffic["WRITE_SIZE"] or 0.0)
    out["hbm_GBps"] = round(moved * line["queries_per_s"] / 1e9, 1) if moved else None
    out["hbm_frac"] = round(out["hbm_GBps"] / 8000.0, 4) if out["hbm_GBps"] else None
    out["parent_keyword_only_queries_per_s"] = measured_qps
    return out


# --------------------------------------------------------------------------------------------------- C4

def run_c4(args, env):
    torch, ma, ctx, dev = env.torch, env.ma, env.ctx, env.dev
    from meilisearch_amd import synth
    rank, world = env.rank, env.world
    n = args.rows or 10_000_000
    d = args.dim or 768
    k = args.k or 20
    # 768 queries per step: 16 sweeps of the vector store, and 12 searches in a row for each of the 64 keyword caller
    # threads — a step of 96 (one search per thread) measured the latency of the slowest of 96 concurrent searches, not
    # the throughput of the path (profiles/r2_bench_c4_96_queries_per_step.json)
    Q = args.queries or 768
    storage = args.storage or "f32"
    n_total = n
    row_sharded = args.shard == "rows" and world > 1
    r0 = 0
    if row_sharded:
        from meilisearch_amd.distributed import row_range
        r0, r1 = row_range(n_total, rank, world)
        n = r1 - r0          # this rank's shard; docids stay global

    # roofline.traffic comes from a child of this run under `rocprofv3 --pmc` that builds the same store: it runs FIRST, while this
    # process holds nothing on the device (behind 256 caller pools, the store, its rows and the posting cache the child's
    # 70 GB no longer fit: the round's first 256-caller line came back with traffic null)
    pmc_early = None
    if env.rank == 0 and env.world == 1 and not args.no_pmc and not env.child:
        phase("c4: roofline (PMC child)")
        has_i8 = (args.storage or "f32") == "f32" and os.environ.get("MSI_VS_I8", "1") != "0"
        pmc_early = pmc_traffic(args, "vs_scan_i8_kernel" if has_i8 else "vs_scan_kernel", ("false",))
    phase("c4: rows + store upload")
    t_setup = time.time()
    rows_seed = 1234 + (rank if row_sharded else 0)
    rows_t = synth.device_rows(n, d, dev, seed=rows_seed)
    ids_t = torch.arange(n, dtype=torch.int32, device=dev)
    if row_sharded:
        ids_t += r0
    torch.cuda.synchronize()
    store = ma.GpuStore(ctx, d, storage=storage)
    store.upload_device(ids_t, rows_t)
    cpu_rows = None
    if env.check:
        cpu_rows = rows_t[:min(n, args.cpu_sample_rows)].cpu().numpy()
    # the rows leave HBM once the store holds them (30.7 GB at C4's size, beside 256 callers' pools): the full-size parity check
    # draws them again, a chunk at a time, from the same generator (synth.device_rows_chunks)
    ctx.synchronize()
    del rows_t
    rows_t = None
    torch.cuda.synchronize()
    torch.cuda.empty_cache()

    q_t = synth.device_queries(Q, d, dev, seed=5678 + (0 if row_sharded else rank))
    out_ids = torch.zeros((Q, k), dtype=torch.int32, device=dev)
    out_dist = torch.zeros((Q, k), dtype=torch.float32, device=dev)
    out_cnt = torch.zeros(Q, dtype=torch.int32, device=dev)
    inexact = torch.zeros(Q, dtype=torch.int32, device=dev)

    phase("c4: dictionary")
    gdict = None
    n_words_q = 0
    words = concat = off = tq = None
    if not args.no_typo:
        words = synth.make_dictionary(args.dict_words, seed=99)
        concat, off = synth.flatten_words(words)
        gdict = ma.GpuDictionary(ctx, concat=concat, offsets=off)
        tq = synth.make_typo_queries(words, Q * args.words_per_query, seed=7 + rank)
        n_words_q = len(tq)
        qb, qoff, qfl = ma.pack_queries(tq)
        qb_t = torch.from_numpy(qb).to(dev)
        qoff_t = torch.from_numpy(qoff.astype(np.int32)).to(dev)
        qfl_t = torch.from_numpy(qfl).to(dev)
        one_t = torch.zeros((n_words_q, 150), dtype=torch.int32, device=dev)
        two_t = torch.zeros((n_words_q, 50), dtype=torch.int32, device=dev)
        one_c = torch.zeros(n_words_q, dtype=torch.int32, device=dev)
        two_c = torch.zeros(n_words_q, dtype=torch.int32, device=dev)
    # ---- keyword leg: msi_keyword_search_ranked, every rule of the default criteria ------------------------
    # [words, typo, proximity, attributeRank, sort, wordPosition, exactness] over a synthetic inverted index of the same
    # n documents (tools/ranked_bench.cpp: Zipf document frequencies, 3 searchable fields, bucketed positions, word pairs
    # at proximities 1..3; postings handed over as the CboRoaringBitmap bytes milli stores, through the index vtable).
    # Every search derives the typos of its words from the index's own dictionary on the device (micro-batched
    # msi_dict_lookup) and those derivations are what its posting lists are read for.  `kw_threads` native caller
    # threads, one in-flight search each (milli's spawn_blocking threads); their set operations share kernel launches
    # (msi_vm.hip).  The queries are resident before the timed region like every other input: 4 x Q distinct queries
    # cycle through the steps; each was run once in setup so that the SYNTHETIC index has generated its postings
    # (index generation is not what is measured; nothing of a search's results is cached).
    kw = None
    kw_threads = 0
    if not args.no_rank:
        import ctypes as C
        kw_so = os.path.join(ROOT, "tools", "bin", "libmsi_rankedbench.so")
        if env.emulated:
            kw_so = os.environ["MSI_RUNNER_SO"]
        if not os.path.exists(kw_so):      # a tree that was never built: __graft_entry__.build() makes it (hipcc, seconds)
            if rank == 0:
                import __graft_entry__
                __graft_entry__.build()
            if env.dist is not None:
                env.dist.barrier()
        kw_lib = C.CDLL(kw_so)
        kw_lib.rb_create.restype = C.c_void_p
        kw_lib.rb_create.argtypes = [C.c_uint64, C.c_uint32]
        kw_lib.rb_create_corpus.restype = C.c_void_p
        kw_lib.rb_create_corpus.argtypes = [C.c_uint64, C.c_uint32, C.c_uint64]
        kw_lib.rb_run_detailed.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32] + [C.c_void_p] * 6
        kw_lib.rb_attach.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint64]
        kw_lib.rb_prepare_queries.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint64]
        kw_lib.rb_run.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p]
        kw_lib.rb_hybrid_merge.argtypes = [C.c_uint32, C.c_uint32] + [C.c_void_p] * 6 + [C.c_float] + [C.c_void_p] * 4
        kw_lib.rb_dict.restype = C.c_void_p
        kw_lib.rb_dict.argtypes = [C.c_void_p]
        kw_lib.rb_pool.restype = C.c_void_p
        kw_lib.rb_pool.argtypes = [C.c_void_p, C.c_uint32]
        kw_lib.rb_destroy.argtypes = [C.c_void_p]
        kw_lib.rb_start_detailed.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32] + [C.c_void_p] * 6
        kw_lib.rb_done.restype = C.c_uint32
        kw_lib.rb_done.argtypes = [C.c_void_p]
        kw_lib.rb_wait.argtypes = [C.c_void_p]
        kw_lib.rb_last_latencies.restype = C.c_uint32
        kw_lib.rb_last_latencies.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32]
        n_docs_kw = n_total if row_sharded else n
        if args.kw_dict_words is None:
            args.kw_dict_words = 2_000_000 if args.kw_corpus == "coherent" else 200_000
        phase("c4: keyword corpus")
        t_corpus = time.time()
        h = (kw_lib.rb_create_corpus(n_docs_kw, args.kw_dict_words, 42) if args.kw_corpus == "coherent"
             else kw_lib.rb_create(n_docs_kw, args.kw_dict_words))
        corpus_s = time.time() - t_corpus
        # Caller threads of this rank: a waiting search costs no CPU, a search in flight ~0.9-1.0 ms of host CPU per fresh query
        # (round 5; 1.2 ms in round 3, 1.7 ms in round 2) — N ranks share the box's CPUs, so each rank gets its share of callers:
        # 16 per granted CPU (256 on a 16-CPU grant: 11.3 CPUs busy, 9 055 hybrid q/s against 8 284 at 160 callers and 8 530 at
        # 208, keyword p50 at load 17.0 / 11.9 / 14.0 ms: profiles/r5_callers.log; round 3's 10 per CPU dates from 1.2 ms per
        # query, when 256 callers ran into the CPU quota), at least 16
        host_cpus = granted_cpus()
        kw_threads = max(16, min(args.kw_threads, host_cpus * int(os.environ.get("MSI_BENCH_CALLERS_PER_CPU", "16")) // world))
        assert kw_lib.rb_attach(h, ctx.handle, kw_threads, args.kw_slots, args.kw_cache_mb) == 0, "keyword runner: rb_attach failed"
        kw_staged = None
        if args.kw_stage and args.kw_corpus == "coherent" and hasattr(kw_lib, "rb_stage_postings"):
            # index-open: the shim walks the three word databases once and hands their stored values to the engine (here the
            # runner derives them from the corpus' tokens, one pass per word on the granted CPUs); untimed, like an index load
            phase("c4: staging the postings into HBM (index-open)")
            kw_lib.rb_stage_postings.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p]
            sec, cnts = C.c_double(0), (C.c_uint64 * 4)()
            st_ = kw_lib.rb_stage_postings(h, max(4, host_cpus // world), C.byref(sec), cnts)
            assert st_ == 0, f"keyword runner: rb_stage_postings failed ({st_}): {ma._lib.lib().msi_last_error()}"
            kw_staged = {"seconds": round(sec.value, 1), "values": int(cnts[0]), "bodies_in_hbm": int(cnts[1]),
                         "kept_on_host": int(cnts[2]), "stored_bytes_in_hbm": int(cnts[3]),
                         "databases": "word_docids, word_fid_docids, word_position_docids, field_id_word_count_docids (complete); "
                                      "word_pair_proximity_docids stays with the callback + cache"}
        # The keyword stream does NOT repeat (round 4 cycled 4 x Q queries through 20 steps against a posting
        # cache at hit rate 0.98): `kw_prime` primer queries — what the index served before the measurement started — and then
        # Q fresh queries for every warm-up, timed and leg step, all drawn from the same generator (one seeded sequence).
        # Untimed setup: (1) every query runs once so that the SYNTHETIC index derives the databases it reads (the stand-in for
        # what LMDB holds: index generation is not the engine's work); (2) the HBM posting cache is emptied; (3) the primer
        # runs against the cold cache (`keyword_cold_posting_cache_queries_per_s`) and leaves behind what a serving process
        # has in HBM.  The timed steps then meet every query for the first time; the cache hit rate they see is the natural
        # one of the workload (`legs.keyword_posting_cache.hit_rate_timed_steps`).
        kw_prime = 4 * Q
        kw_stream_steps = args.steps + 4                        # the timed steps + the keyword-only leg's three + the by-universe pass
        if args.kw_stream == "cycle" or env.child:
            kw_stream_steps = 0                                 # round 4's stream: the 4 x Q primer queries, cycled
        n_kw_queries = kw_prime + kw_stream_steps * Q
        kw_lib.rb_prepare_queries(h, n_kw_queries, args.kw_terms, 4242 + rank)
        kw = {"lib": kw_lib, "h": h, "ids": np.zeros((Q, k), np.uint32), "n": np.zeros(Q, np.uint32),
              "scores": np.zeros((Q, k), np.float64), "m_ids": np.zeros((Q, k), np.uint32), "m_sem": np.zeros((Q, k), np.uint8),
              "m_cnt": np.zeros(Q, np.uint32), "m_hits": np.zeros(Q, np.uint32), "step": 0, "prime": kw_prime,
              "stream_steps": kw_stream_steps, "n_queries": n_kw_queries, "staged": kw_staged}
        phase("c4: index derivation pass (untimed: the synthetic index derives its databases)")
        t_derive = time.perf_counter()
        kw["stream_distinct"] = kw_stream_steps
        derive_from, n_derive = 0, n_kw_queries
        if world > 1 and kw_stream_steps:
            # N ranks derive their streams on the SAME granted CPUs (the pass costs ~60 ms of CPU per query: 84 s at N = 1 on
            # 16 CPUs — eleven minutes for eight ranks): the primer first, then as many fresh steps as fit a budget at the rate
            # the primer showed (the slowest rank's), at least four; the timed steps cycle through those (the line says so)
            for first in range(0, kw_prime, Q):
                assert kw_lib.rb_run(h, first, Q, k, kw["ids"].ctypes.data, kw["n"].ctypes.data, kw["scores"].ctypes.data) == 0
            t_primer = max(env.gather_scalar(time.perf_counter() - t_derive))
            budget_s = float(os.environ.get("MSI_BENCH_DERIVE_BUDGET_S", "300"))
            fit = int(max(0.0, budget_s - t_primer) / max(1e-3, t_primer / (kw_prime // Q)))
            kw["stream_distinct"] = max(4, min(kw_stream_steps, fit))
            derive_from, n_derive = kw_prime, kw_prime + kw["stream_distinct"] * Q
        for first in range(derive_from, n_derive, Q):
            assert kw_lib.rb_run(h, first, Q, k, kw["ids"].ctypes.data, kw["n"].ctypes.data, kw["scores"].ctypes.data) == 0
        kw["index_derivation_seconds"] = round(time.perf_counter() - t_derive, 1)
        if hasattr(kw_lib, "rb_freeze"):
            # what the index derived becomes a snapshot its callbacks read without a lock, as LMDB's readers do (the runner's
            # reader-writer lock was a sixth of the leg's host CPU on the fresh stream: harness, not engine)
            kw_lib.rb_freeze.argtypes = [C.c_void_p]
            assert kw_lib.rb_freeze(h) == 0
        kw["cold_cache_queries_per_s"] = None
        if not env.child:
            phase("c4: primer on a cold posting cache")
            ma._lib.check(ma._lib.lib().msi_dict_reset_posting_cache(C.c_void_p(kw_lib.rb_dict(h))))
            t0c = time.perf_counter()
            for first in range(0, kw_prime, Q):
                assert kw_lib.rb_run(h, first, Q, k, kw["ids"].ctypes.data, kw["n"].ctypes.data, kw["scores"].ctypes.data) == 0
            kw["cold_cache_queries_per_s"] = round(kw_prime / (time.perf_counter() - t0c), 1)
        kw["corpus_seconds"] = round(corpus_s, 1)
    torch.cuda.synchronize()
    setup_s = time.time() - t_setup

    packed = gathered = None
    if world > 1:
        # ONE packed buffer per rank and step: [Q][k] distances (f32 bits) | [Q][k] docids | [Q] counts, as int32
        packed = torch.zeros(Q * (2 * k + 1), dtype=torch.int32, device=dev)
        gathered = torch.zeros(world * Q * (2 * k + 1), dtype=torch.int32, device=dev)
        m_ids = torch.zeros((Q, k), dtype=torch.int32, device=dev)
        m_dist = torch.zeros((Q, k), dtype=torch.float32, device=dev)
        m_cnt = torch.zeros(Q, dtype=torch.int32, device=dev)

    # the exchange runs through libmsi's own multi-GPU entry points (RCCL inside the library: msi_group_create_rank +
    # msi_group_allgather); rank 0's communicator id reaches the other ranks through the launcher's process group
    group = None
    if world > 1:
        import ctypes as C
        L = ma._lib.lib()
        uid = torch.zeros(128, dtype=torch.uint8, device=dev)
        if rank == 0:
            buf = (C.c_uint8 * 128)()
            ma._lib.check(L.msi_group_unique_id(buf))
            uid.copy_(torch.tensor(list(buf), dtype=torch.uint8))
        env.dist.broadcast(uid, 0)
        buf = (C.c_uint8 * 128)(*uid.cpu().tolist())
        group = C.c_void_p()
        # (under a watchdog: the library's own RCCL instance beside torch's has never run on an 8-GPU node — a communicator
        # that does not come up within two minutes must not cost the whole line; ctypes releases the GIL during the call)
        import threading
        init = {}

        def _init_group():
            try:
                ma._lib.check(L.msi_group_create_rank(ctx.handle, rank, world, buf, C.byref(group)))
                init["ok"] = True
            except Exception as e:   # noqa: BLE001 - the line must still be produced: the launcher's own RCCL group carries the exchange
                init["err"] = e
        th = threading.Thread(target=_init_group, daemon=True)
        th.start()
        th.join(timeout=120.0)
        if not init.get("ok"):
            why = "no answer within 120 s" if th.is_alive() else str(init.get("err"))
            print(f"[bench] rank {rank}: msi_group_create_rank failed ({why}); exchanging through torch.distributed", file=sys.stderr)
            group = None
        ok = torch.tensor([1 if group is not None else 0], dtype=torch.int32, device=dev)
        env.dist.all_reduce(ok, op=env.dist.ReduceOp.MIN)    # every rank takes the same path
        if int(ok.item()) == 0:
            group = None
    exchange_path, rccl_ranks_seen = "none (one GPU)", 1
    if world > 1:
        # which exchange actually runs, and proof that every rank took part: each rank contributes its rank number
        exchange_path = ("RCCL called inside libmsi (msi_group_create_rank + msi_group_allgather)" if group is not None else
                         "torch.distributed all_gather_into_tensor (RCCL through the launcher's process group: "
                         "msi_group_create_rank failed on this box)")
        mine = torch.full((4,), rank, dtype=torch.int32, device=dev)
        seen = torch.full((4 * world,), -1, dtype=torch.int32, device=dev)
        torch.cuda.current_stream().synchronize()
        if group is not None:
            ma._lib.check(L.msi_group_allgather(group, C.c_void_p(mine.data_ptr()), 16, C.c_void_p(seen.data_ptr())))
            ctx.synchronize()
        else:
            env.dist.all_gather_into_tensor(seen, mine)
            torch.cuda.synchronize()
        rccl_ranks_seen = int(torch.unique(seen[seen >= 0]).numel())
        assert rccl_ranks_seen == world, f"the exchange saw {rccl_ranks_seen} of {world} ranks"

    def exchange():
        """The one exchange step: per-rank top-k lists (Q*(2k+1)*4 bytes) in ONE all-gather over xGMI — RCCL called by
        libmsi on the context's stream, in order with the searches that filled the lists."""
        packed[:Q * k].copy_(out_dist.view(torch.int32).reshape(-1))
        packed[Q * k:2 * Q * k].copy_(out_ids.reshape(-1))
        packed[2 * Q * k:].copy_(out_cnt)
        torch.cuda.current_stream().synchronize()      # the packing ran on torch's stream
        if group is not None:
            ma._lib.check(ma._lib.lib().msi_group_allgather(group, C.c_void_p(packed.data_ptr()), packed.numel() * 4,
                                                            C.c_void_p(gathered.data_ptr())))
            ctx.synchronize()
        else:
            env.dist.all_gather_into_tensor(gathered, packed)
            torch.cuda.synchronize()
        g = gathered.view(world, Q * (2 * k + 1))
        return (g[:, Q * k:2 * Q * k].reshape(world, Q, k), g[:, :Q * k].view(torch.float32).reshape(world, Q, k),
                g[:, 2 * Q * k:].reshape(world, Q))

    def kw_first(step=None):
        """First query of keyword step `step` (default: the next one): fresh queries behind the primer while the stream
        lasts, then (untimed extras only) around the primer again."""
        i = kw["step"] if step is None else step
        if i < args.warmup or not kw["stream_steps"]:
            return (i * Q) % kw["prime"]                          # warm-up steps (and the cycled stream): primer queries again
        i -= args.warmup
        if i < kw["stream_steps"]:
            return kw["prime"] + (i % kw["stream_distinct"]) * Q   # (stream_distinct < stream_steps only at N > 1, time-boxed)
        return ((i - kw["stream_steps"]) * Q) % kw["prime"]

    def keyword_run():
        """The keyword leg of this step's Q queries (blocks until the caller threads are done; the vector scan enqueued
        before it keeps the device busy meanwhile)."""
        first = kw_first()
        kw["step"] += 1
        st = kw["lib"].rb_run(kw["h"], first, Q, k, kw["ids"].ctypes.data, kw["n"].ctypes.data, kw["scores"].ctypes.data)
        assert st == 0, "msi_keyword_search_ranked failed"

    def hybrid_merge(res):
        """ScoreWithRatioResult::merge, semanticRatio 0.5, of every query's two lists."""
        v_ids = np.ascontiguousarray(res[0].numpy().view(np.uint32))
        v_dist = np.ascontiguousarray(res[1].numpy())
        v_cnt = np.ascontiguousarray(res[2].numpy().view(np.uint32))
        kw["lib"].rb_hybrid_merge(Q, k, v_ids.ctypes.data, v_dist.ctypes.data, v_cnt.ctypes.data, kw["ids"].ctypes.data,
                                  kw["scores"].ctypes.data, kw["n"].ctypes.data, 0.5, kw["m_ids"].ctypes.data,
                                  kw["m_sem"].ctypes.data, kw["m_cnt"].ctypes.data, kw["m_hits"].ctypes.data)
        return res + ((kw["m_ids"], kw["m_sem"], kw["m_cnt"], kw["m_hits"]),)

    step_parts = {"vector_leg": 0.0, "typo_lookup_enqueue_and_keyword_leg": 0.0, "sync_d2h_merge_exchange": 0.0, "n": 0}

    def step(legs_order=None):
        legs_order = legs_order or args.legs
        if kw is not None and legs_order == "tail":
            # The keyword leg is latency-bound (17 dependent rounds per search, `kw_threads` searches in flight) and ends with
            # a TAIL: the last searches of the step run with most callers already idle.  The vector scan — 8 sweeps that
            # saturate HBM — starts when `--tail-at` of the step's searches are done and streams beside that tail; the
            # keyword rounds in flight at that point are slowed, the bulk of them never sees the scan.
            first = kw_first()
            kw["step"] += 1
            assert kw["lib"].rb_start_detailed(kw["h"], first, Q, k, kw["ids"].ctypes.data, kw["n"].ctypes.data,
                                               kw["scores"].ctypes.data, None, None, None) == 0
            if gdict is not None:
                gdict.lookup_device(qb_t, qoff_t, qfl_t, n_words_q, one_t, one_c, two_t, two_c)
            thr = int(args.tail_at * Q)
            while kw["lib"].rb_done(kw["h"]) < thr:
                time.sleep(0.0003)
            store.search_device(q_t, k, out_ids, out_dist, out_cnt, inexact)  # ceil(Q / max_batch) HBM sweeps
            assert kw["lib"].rb_wait(kw["h"]) == 0, "msi_keyword_search_ranked failed"
        elif kw is not None and legs_order == "overlap":
            # Both legs from the start: the keyword searches are started on their caller threads, the vector leg runs on this
            # thread beside them (msi_vs_search_device returns when every query is answered), then the keyword job is joined.
            first = kw_first()
            kw["step"] += 1
            assert kw["lib"].rb_start_detailed(kw["h"], first, Q, k, kw["ids"].ctypes.data, kw["n"].ctypes.data,
                                               kw["scores"].ctypes.data, None, None, None) == 0
            if gdict is not None:
                gdict.lookup_device(qb_t, qoff_t, qfl_t, n_words_q, one_t, one_c, two_t, two_c)
            store.search_device(q_t, k, out_ids, out_dist, out_cnt, inexact)
            assert kw["lib"].rb_wait(kw["h"]) == 0, "msi_keyword_search_ranked failed"
        else:
            t_a = time.perf_counter()
            store.search_device(q_t, k, out_ids, out_dist, out_cnt, inexact)  # ceil(Q / max_batch) HBM sweeps
            if kw is not None and legs_order == "serial":
                ctx.synchronize()     # the scan streams HBM on its own, then the (latency-bound) keyword lists run
            t_b = time.perf_counter()
            if gdict is not None:     # (VALU-bound, on the context's second stream: beside the keyword rounds, not beside the scan)
                gdict.lookup_device(qb_t, qoff_t, qfl_t, n_words_q, one_t, one_c, two_t, two_c)
            if kw is not None:
                keyword_run()
            step_parts["vector_leg"] += t_b - t_a          # (host clocks around the parts of a serial step: legs.step_parts_ms)
            step_parts["typo_lookup_enqueue_and_keyword_leg"] += time.perf_counter() - t_b
            step_parts["n"] += 1
        t_c = time.perf_counter()
        ctx.synchronize()
        if row_sharded:
            from meilisearch_amd.distributed import merge_topk_device
            g_ids, g_dist, g_cnt = exchange()
            merge_topk_device(ctx, g_ids.contiguous(), g_dist.contiguous(), g_cnt.contiguous(), m_ids, m_dist, m_cnt)
            ctx.synchronize()
            res = (m_ids.cpu(), m_dist.cpu(), m_cnt.cpu())
        else:
            res = (out_ids.cpu(), out_dist.cpu(), out_cnt.cpu())   # what the Rust caller receives
        if gdict is not None:
            res += (one_c.cpu(), two_c.cpu(), one_t.cpu(), two_t.cpu())
        if kw is not None:
            res = hybrid_merge(res)
        if world > 1 and not row_sharded:
            exchange()
        step_parts["sync_d2h_merge_exchange"] += time.perf_counter() - t_c
        return res

    phase("c4: warm-up + timed steps")
    if kw is not None and args.legs != "serial":
        store.set_sweep_split(args.sweep_split)      # the sweeps share the device with the keyword rounds: short workgroups
    for _ in range(args.warmup):
        step()
    ctx.set_profiling(True)
    store.scan_time()
    if gdict is not None:
        gdict.match_time()
    pc_t0 = None
    if kw is not None:
        pc_t0 = (C.c_uint64 * 4)()
        ma._lib.lib().msi_dict_posting_cache_stats(C.c_void_p(kw["lib"].rb_dict(kw["h"])), pc_t0)
    for key in step_parts:
        step_parts[key] = 0
    elapsed, lat = env.timed(step, args.steps, 0)
    timed_parts = dict(step_parts)
    pc_timed = None
    if kw is not None:
        pc_t1 = (C.c_uint64 * 4)()
        ma._lib.lib().msi_dict_posting_cache_stats(C.c_void_p(kw["lib"].rb_dict(kw["h"])), pc_t1)
        pc_timed = (int(pc_t1[0] - pc_t0[0]), int(pc_t1[1] - pc_t0[1]))
    scan_n, scan_ms = store.scan_time()
    match_n, match_ms = gdict.match_time() if gdict is not None else (0, 0.0)
    ctx.set_profiling(False)
    stats = store.stats()
    n_inexact = int(inexact.sum().item())
    phase("c4: legs on their own")
    # the two legs on their own (untimed extras, 3 steps each): what bounds the step
    legs = {}
    if timed_parts.get("n"):
        legs["step_parts_ms"] = {key: round(v / timed_parts["n"] * 1e3, 2) for key, v in timed_parts.items() if key != "n"}
        legs["step_parts_ms"]["is"] = "host clocks around the parts of the timed (serial) steps, mean per step"
    if kw is not None and not env.child:
        ctx.set_profiling(True)
        store.scan_time()
        t0 = time.perf_counter()
        for _ in range(3):
            store.search_device(q_t, k, out_ids, out_dist, out_cnt, inexact)
            ctx.synchronize()
        legs["vector_only_queries_per_s"] = round(3 * Q / (time.perf_counter() - t0), 1)
        alone_n, alone_ms = store.scan_time()
        ctx.set_profiling(False)
        if alone_n:   # the same kernel without the keyword lists beside it (the timed step overlaps the two legs)
            alone_bytes = sweep_kernel(stats, n)[2]
            legs["scan_kernel_alone"] = {"avg_launch_ms": round(alone_ms / alone_n, 4), "launches": alone_n,
                                         "frac_of_8_TBps": round(alone_bytes / (alone_ms / alone_n * 1e-3) / 1e9 / 8000.0, 4)}
        if stats.get("i8_bytes_per_tile"):
            # continuity with rounds 1-4: the same batch through the f32 sweeps (the level the int8 sweep's unproven queries fall
            # to; MSI_VS_FIRST_LEVEL=f32 is read per call), SURVEY 8(d)'s N x d x 4 bytes per sweep
            os.environ["MSI_VS_FIRST_LEVEL"] = "f32"
            try:
                ctx.set_profiling(True)
                store.scan_time()
                t0 = time.perf_counter()
                store.search_device(q_t, k, out_ids, out_dist, out_cnt, inexact)
                ctx.synchronize()
                f32_dt = time.perf_counter() - t0
                f_n, f_ms = store.scan_time()
                ctx.set_profiling(False)
            finally:
                os.environ.pop("MSI_VS_FIRST_LEVEL")
            f32_bytes = ((n + 15) // 16) * stats["bytes_per_tile"]
            if f_n:
                legs["f32_sweep_level"] = {"vector_only_queries_per_s": round(Q / f32_dt, 1), "avg_launch_ms": round(f_ms / f_n, 4),
                                           "launches": f_n, "algorithmic_bytes_per_launch": int(f32_bytes),
                                           "frac_of_8_TBps": round(f32_bytes / (f_ms / f_n * 1e-3) / 1e9 / 8000.0, 4),
                                           "is": "vs_scan_kernel over the f32 rows (rounds 1-4's level 0; now what the int8 sweep's "
                                                 "unproven queries fall to)"}
        import resource
        vs0 = (C.c_uint64 * 6)()
        ma._lib.lib().msi_bits_vm_stats(C.c_void_p(kw["lib"].rb_pool(kw["h"], 0)), vs0)
        vb0, vb1 = (C.c_uint64 * 3)(), (C.c_uint64 * 3)()
        ma._lib.lib().msi_bits_vm_bytes(vb0)
        ru0 = resource.getrusage(resource.RUSAGE_SELF)
        t0 = time.perf_counter()
        for _ in range(3):
            keyword_run()
        dt = time.perf_counter() - t0
        ru1 = resource.getrusage(resource.RUSAGE_SELF)
        if args.kw_gap_probe:
            # why is the keyword leg slower inside the serial step (legs.step_parts_ms) than on its own?  The same three passes
            # with 12 ms of idle host and device in front of each, then with the vector leg in front of each
            def kw_ms():
                t_ = time.perf_counter()
                keyword_run()
                return (time.perf_counter() - t_) * 1e3
            saved_step = kw["step"]
            kw["step"] = max(kw["step"], args.warmup + kw["stream_steps"])    # primer queries again, for all three variants
            b2b, idle, after_vec = [], [], []
            kw_ms()
            for _ in range(3):
                b2b.append(kw_ms())
            for _ in range(3):
                time.sleep(0.012)
                idle.append(kw_ms())
            for _ in range(3):
                store.search_device(q_t, k, out_ids, out_dist, out_cnt, inexact)
                ctx.synchronize()
                after_vec.append(kw_ms())
            kw["step"] = saved_step
            legs["keyword_gap_probe"] = {"fresh_back_to_back_ms": round(dt / 3 * 1e3, 2), "cycled_back_to_back_ms": [round(x, 2) for x in b2b],
                                         "cycled_behind_12_ms_idle_ms": [round(x, 2) for x in idle],
                                         "cycled_behind_the_vector_leg_ms": [round(x, 2) for x in after_vec]}
        legs["keyword_only_queries_per_s"] = round(3 * Q / dt, 1)
        legs["keyword_only_host_cpus_used"] = round((ru1.ru_utime + ru1.ru_stime - ru0.ru_utime - ru0.ru_stime) / dt, 2)
        legs["host_cpus_allowed"] = len(os.sched_getaffinity(0))
        pc = (C.c_uint64 * 4)()
        ma._lib.lib().msi_dict_posting_cache_stats(C.c_void_p(kw["lib"].rb_dict(kw["h"])), pc)
        vs = (C.c_uint64 * 6)()
        ma._lib.lib().msi_bits_vm_stats(C.c_void_p(kw["lib"].rb_pool(kw["h"], 0)), vs)
        legs["keyword_posting_cache"] = {"hits": int(pc[0]), "misses": int(pc[1]), "bytes_used": int(pc[2]),
                                         "hit_rate": round(pc_timed[0] / max(1, pc_timed[0] + pc_timed[1]), 4),
                                         "hit_rate_is": "of the TIMED steps (every query met for the first time; the cache holds what the "
                                                        "primer queries left)" if kw["stream_steps"] else "of the timed steps (cycled queries)",
                                         "hit_rate_since_the_cache_was_emptied": round(pc[0] / max(1, pc[0] + pc[1]), 4)}
        legs["keyword_lists_per_query"] = round((vs[1] - vs0[1]) / (3.0 * Q), 2)
        # the leg's ALGORITHMIC denominator: the stored posting bytes a query's decodes read — what milli's
        # db_cache hands its RoaringBitmaps (db_cache.rs:50-84) — beside what the command lists ask the memory system for (every
        # operand of every command, whole: dense sets of the docid space or of the search's compact space)
        ma._lib.lib().msi_bits_vm_bytes(vb1)
        legs["keyword_algorithmic_bytes_per_query"] = {
            "stored_posting_bytes_read": round((vb1[1] - vb0[1]) / (3.0 * Q), 1),
            "set_operand_bytes_of_the_command_lists": round((vb1[0] - vb0[0]) / (3.0 * Q), 1),
            "operand_bytes_over_posting_bytes": round((vb1[0] - vb0[0]) / max(1.0, float(vb1[1] - vb0[1])), 1),
            "is": "msi_bits_vm_bytes over the keyword-only passes; the ratio is what dense sets cost over the stored containers "
                  "(HBM traffic per query measured by PMC: --kw-roofline)"}
        legs["keyword_host_cpu_ms_per_query"] = round(legs["keyword_only_host_cpus_used"] / max(1e-9, legs["keyword_only_queries_per_s"]) * 1e3, 3)
        legs["keyword_cold_posting_cache_queries_per_s"] = kw.get("cold_cache_queries_per_s")
        if kw.get("staged"):
            ss_ = (C.c_uint64 * 4)()
            ma._lib.lib().msi_dict_staged_stats(C.c_void_p(kw["lib"].rb_dict(kw["h"])), ss_)
            legs["keyword_postings_staged_at_index_open"] = dict(kw["staged"], absent_answers_from_complete_dbs=int(ss_[3]),
                                                                 cold_is="the cache as msi_dict_reset_posting_cache leaves it: "
                                                                         "what was staged, no pair proximity")
        if kw["stream_steps"]:
            # the two ends the fresh stream lies between: the cold cache above, and round 4's stream (queries the engine has met
            # before, here the primer's: posting cache hit rate ~0.98) — what a server converges to on a query mix that repeats
            t0 = time.perf_counter()
            for first in range(0, 3 * Q, Q):
                assert kw["lib"].rb_run(kw["h"], first % kw["prime"], Q, k, kw["ids"].ctypes.data, kw["n"].ctypes.data,
                                        kw["scores"].ctypes.data) == 0
            legs["keyword_cycled_queries_per_s"] = round(3 * Q / (time.perf_counter() - t0), 1)
        # sensitivity to the universe (the documents that match the query at all): one more pass with every search's candidate
        # count and wall time at load, grouped by |universe| / documents
        cand = np.zeros(Q, np.uint64)
        first = kw_first()
        kw["step"] += 1
        assert kw["lib"].rb_run_detailed(kw["h"], first, Q, k, kw["ids"].ctypes.data, kw["n"].ctypes.data, kw["scores"].ctypes.data,
                                         None, None, cand.ctypes.data) == 0
        lat_q = np.zeros(Q, np.float64)
        kw["lib"].rb_last_latencies(kw["h"], lat_q.ctypes.data, Q)
        n_idx = n_total if row_sharded else n
        sweep = {}
        for name, lo_f, hi_f in (("<=0.1%", 0.0, 0.001), ("0.1-1%", 0.001, 0.01), ("1-12.5%", 0.01, 0.125),
                                 ("12.5-50%", 0.125, 0.5), (">50%", 0.5, 1.01)):
            m_ = (cand > lo_f * n_idx) & (cand <= hi_f * n_idx) if lo_f > 0 else (cand <= hi_f * n_idx)
            if m_.any():
                sweep[name] = {"queries": int(m_.sum()), "p50_ms_at_load": round(float(np.median(lat_q[m_])), 3),
                               "mean_ms_at_load": round(float(lat_q[m_].mean()), 3)}
        legs["keyword_by_universe"] = {"buckets": sweep, "is": f"one {Q}-query pass of the keyword leg ({kw_threads} callers): searches grouped "
                                       "by candidates / documents; a universe of <= 1/8 of the index continues in the compact space, above that the sub-tree of a small enough bucket does",
                                       "mean_universe_docs": round(float(cand.mean()), 1)}
        legs["keyword_lists_per_launch_round"] = round((vs[1] - vs0[1]) / max(1, vs[0] - vs0[0]), 2)   # of this leg only
        import ctypes as _C
        cst, lst = (_C.c_uint64 * 3)(), (_C.c_uint64 * 2)()
        ma._lib.lib().msi_search_compaction_stats(cst)
        ma._lib.lib().msi_search_late_compaction_stats(lst)
        legs["keyword_compact_space"] = {
            "searches": int(cst[0]), "continued_in_the_space_of_their_universe": int(cst[1]),
            "sub_trees_moved_into_their_bucket": int(lst[0]), "mean_bucket_docs": round(lst[1] / max(1, lst[0]), 1),
            "is": "process-wide since start: a search whose universe is <= 1/8 of the index continues over the ranks of its "
                  "universe; one whose universe is larger moves the sub-tree of a bucket (<= 1/8 of the index) into the ranks "
                  "of that bucket once nothing else of its bucket sort is alive (msi_search.hip Ctx::late_enter)"}
    # ---- per-QUERY latency (the metric is "queries/sec + p50 latency"; ms_per_step is the latency of a 768-query step) ----
    # N > 1: every rank measured its keyword leg at the same time on the CPUs the ranks share: their sum is the whole job's
    # measured keyword throughput, to be read against the prediction (granted CPUs / host CPU per query)
    if kw is not None and args.legs == "serial" and env.world == 1 and not env.emulated and not env.child and not args.no_overlapped_leg:
        # The same step with its two legs SIDE BY SIDE and the sweep cut into short workgroups (msi_vs_set_sweep_split: the
        # keyword rounds' kernels get in between the sweep's workgroups instead of behind the sweep) — what a server that runs
        # both legs of execute_hybrid concurrently gets.  Not the line's `value`: beside the keyword rounds a sweep takes 3-4x
        # its own time, and the roofline object must describe the sweep, so the timed steps keep the legs one after the other.
        phase("c4: the step with its legs side by side")
        store.set_sweep_split(args.sweep_split)
        try:
            for _ in range(2):
                step("overlap")
            env.sync_all()
            ctx.set_profiling(True)
            store.scan_time()
            n_ov = max(3, min(10, args.steps))
            t0 = time.perf_counter()
            lat_ov = []
            for _ in range(n_ov):
                s0 = time.perf_counter()
                step("overlap")
                lat_ov.append((time.perf_counter() - s0) * 1e3)
            env.sync_all()
            dt_ov = time.perf_counter() - t0
            ov_n, ov_ms = store.scan_time()
            ctx.set_profiling(False)
        finally:
            store.set_sweep_split(1)
        ov_bytes = sweep_kernel(stats, n)[2]
        legs["hybrid_legs_side_by_side"] = {
            "queries_per_s": round(n_ov * Q / dt_ov, 1), "ms_per_step": round(dt_ov / n_ov * 1e3, 3),
            "p50_latency_ms": round(statistics.median(lat_ov), 3), "steps": n_ov, "sweep_split": args.sweep_split,
            "sweep_avg_launch_ms_beside_the_keyword_rounds": round(ov_ms / max(1, ov_n), 4),
            "sweep_frac_of_8_TBps_beside_the_keyword_rounds": round(ov_bytes / max(1e-9, ov_ms / max(1, ov_n) * 1e-3) / 1e9 / 8000.0, 4),
            "is": "the timed step with --legs overlap and msi_vs_set_sweep_split(%d): both legs from the start" % args.sweep_split}
    kw_only_by_rank = env.gather_scalar(legs.get("keyword_only_queries_per_s") or 0.0) if kw is not None else None
    vec_only_by_rank = env.gather_scalar(legs.get("vector_only_queries_per_s") or 0.0) if kw is not None else None
    note_hbm(torch)
    phase("c4: per-query latency")
    latency = None
    if kw is not None and not env.child and rank == 0:
        def p50(xs):
            return round(statistics.median(xs), 3)
        q1 = q_t[:1].contiguous()
        o_i, o_d = out_ids[:1].contiguous(), out_dist[:1].contiguous()
        o_c, o_x = out_cnt[:1].contiguous(), inexact[:1].contiguous()
        vec, kwl, hyb = [], [], []
        one_ids, one_n, one_sc = np.zeros((1, k), np.uint32), np.zeros(1, np.uint32), np.zeros((1, k), np.float64)
        for i in range(40):
            t0 = time.perf_counter()
            store.search_device(q1, k, o_i, o_d, o_c, o_x)          # one query: one sweep of the store for it alone
            ctx.synchronize()
            r_v = (o_i.cpu(), o_d.cpu(), o_c.cpu())
            t1 = time.perf_counter()
            assert kw["lib"].rb_run(kw["h"], i, 1, k, one_ids.ctypes.data, one_n.ctypes.data, one_sc.ctypes.data) == 0
            t2 = time.perf_counter()
            vec.append((t1 - t0) * 1e3)
            kwl.append((t2 - t1) * 1e3)
        for i in range(40):                                          # one hybrid query in flight: both legs, then the merge
            t0 = time.perf_counter()
            store.search_device(q1, k, o_i, o_d, o_c, o_x)          # enqueued; the keyword search runs meanwhile
            assert kw["lib"].rb_run(kw["h"], 100 + i, 1, k, one_ids.ctypes.data, one_n.ctypes.data, one_sc.ctypes.data) == 0
            ctx.synchronize()
            v_ids = np.ascontiguousarray(o_i.cpu().numpy().view(np.uint32))
            v_dist = np.ascontiguousarray(o_d.cpu().numpy())
            v_cnt = np.ascontiguousarray(o_c.cpu().numpy().view(np.uint32))
            kw["lib"].rb_hybrid_merge(1, k, v_ids.ctypes.data, v_dist.ctypes.data, v_cnt.ctypes.data, one_ids.ctypes.data,
                                      one_sc.ctypes.data, one_n.ctypes.data, 0.5, kw["m_ids"].ctypes.data, kw["m_sem"].ctypes.data,
                                      kw["m_cnt"].ctypes.data, kw["m_hits"].ctypes.data)
            hyb.append((time.perf_counter() - t0) * 1e3)
        keyword_run()                                                # a full step of the keyword leg: every search's own wall time
        kw["step"] -= 1
        at_load = np.zeros(Q, np.float64)
        n_lat = kw["lib"].rb_last_latencies(kw["h"], at_load.ctypes.data, Q)
        sweep_ms = scan_ms / max(1, scan_n)
        latency = {
            "vector_p50_ms_b1": p50(vec), "keyword_p50_ms_1_caller": p50(kwl), "hybrid_p50_ms_1_inflight": p50(hyb),
            "keyword_p50_ms_at_load": p50(at_load[:n_lat].tolist()), "keyword_p99_ms_at_load": round(float(np.percentile(at_load[:n_lat], 99)), 3),
            "hybrid_p50_ms_at_load": round(p50(at_load[:n_lat].tolist()) + sweep_ms, 3),
            "at_load_is": f"{kw_threads} keyword callers in flight (wall time of each msi_keyword_search_ranked inside a {Q}-query step) "
                          f"+ one {store.max_batch}-query HBM sweep of the store ({sweep_ms:.2f} ms) for the vector list",
            "per": "query"}
    # ---- N > 1: the north_star's own C4 shape beside the weak-scaling line — rows sharded over the GPUs (1 / N of the store
    # each), the same query batch on every GPU, one packed all-gather + device k-way merge; vector leg only, untimed extra.
    # It runs LAST and under a watchdog: a rank that fails inside it would leave the others in the all-gather, and the
    # weak-scaling line must be printed whatever happens to the extra ----
    def rows_sharded_extra():
        try:
            from meilisearch_amd.distributed import merge_topk_device, row_range
            sr0, sr1 = row_range(n_total, rank, world)
            s_rows = synth.device_rows(sr1 - sr0, d, dev, seed=4321 + rank)
            s_ids = torch.arange(sr0, sr1, dtype=torch.int32, device=dev)
            sh_store = ma.GpuStore(ctx, d, storage=storage)
            sh_store.upload_device(s_ids, s_rows)
            del s_rows
            sq_t = synth.device_queries(Q, d, dev, seed=999)        # the SAME batch on every rank
            s_packed = torch.zeros(Q * (2 * k + 1), dtype=torch.int32, device=dev)
            s_gath = torch.zeros(world * Q * (2 * k + 1), dtype=torch.int32, device=dev)
            sm_ids = torch.zeros((Q, k), dtype=torch.int32, device=dev)
            sm_dist = torch.zeros((Q, k), dtype=torch.float32, device=dev)
            sm_cnt = torch.zeros(Q, dtype=torch.int32, device=dev)

            def sharded_step():
                sh_store.search_device(sq_t, k, out_ids, out_dist, out_cnt, inexact)
                ctx.synchronize()
                s_packed[:Q * k].copy_(out_dist.view(torch.int32).reshape(-1))
                s_packed[Q * k:2 * Q * k].copy_(out_ids.reshape(-1))
                s_packed[2 * Q * k:].copy_(out_cnt)
                torch.cuda.current_stream().synchronize()
                if group is not None:
                    ma._lib.check(ma._lib.lib().msi_group_allgather(group, C.c_void_p(s_packed.data_ptr()), s_packed.numel() * 4,
                                                                    C.c_void_p(s_gath.data_ptr())))
                    ctx.synchronize()
                else:
                    env.dist.all_gather_into_tensor(s_gath, s_packed)
                    torch.cuda.synchronize()
                g = s_gath.view(world, Q * (2 * k + 1))
                merge_topk_device(ctx, g[:, Q * k:2 * Q * k].reshape(world, Q, k).contiguous(),
                                  g[:, :Q * k].view(torch.float32).reshape(world, Q, k).contiguous(),
                                  g[:, 2 * Q * k:].reshape(world, Q).contiguous(), sm_ids, sm_dist, sm_cnt)
                ctx.synchronize()
                return sm_ids.cpu(), sm_dist.cpu(), sm_cnt.cpu()
            s_elapsed, s_lat = env.timed(sharded_step, 10, 2)
            return {
                "what": f"BASELINE config 4 as written: {n_total} docs x {d}-d sharded over {world} GPUs ({sr1 - sr0} rows on rank 0), "
                        f"the same {Q}-query batch on every GPU, ONE packed all-gather of per-shard top-{k} + device k-way merge "
                        "(vector leg only, 10 steps, max over ranks)",
                "queries_per_s": round(Q * 10 / s_elapsed, 1), "ms_per_step": round(s_elapsed / 10 * 1e3, 3),
                "p50_step_ms": round(statistics.median(s_lat), 3), "scaling": "strong", "exchange_path": exchange_path}
        except Exception as e:      # noqa: BLE001 - an extra: the weak-scaling line must still be produced
            return {"error": repr(e)[:300]}

    def guarded_extra(line):
        import threading

        def give_up():
            if line is not None:
                line["rows_sharded"] = {"error": f"no answer within {args.extra_timeout} s (a rank stuck in the exchange); the line itself is complete"}
                print(short_line(line), flush=True)
            os._exit(0)
        dog = threading.Timer(args.extra_timeout, give_up)
        dog.daemon = True
        dog.start()
        try:
            return rows_sharded_extra()
        finally:
            dog.cancel()

    want_extra = world > 1 and not row_sharded and not args.no_rows_sharded_extra
    if rank != 0:
        if want_extra:
            guarded_extra(None)
        return None
    phase("c4: assembling the line")
    total_queries = Q * (1 if row_sharded else world) * args.steps
    kname, kprefix, algo_bytes = sweep_kernel(stats, n)      # one sweep of the level-0 kernel
    out = {
        "metric": "hybrid-search hot path queries/sec (10M-doc index, 768-d cosine top-20 + 2-typo term lookup)",
        "value": round(total_queries / elapsed, 2),
        "unit": "queries/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(elapsed / args.steps * 1e3, 4),
        "p50_latency_ms": round(statistics.median(lat), 4),
        "step_ms": [round(x, 1) for x in lat],
        "higher_is_better": True,
        "scaling": "strong" if row_sharded else "weak",
        "vs_baseline": None,
        "dtype": ("f32 (every distance: the reference's f32 arithmetic on the f32 rows; candidates: int8 MFMA sweep of a quantised copy)"
                  if stats.get("i8_bytes_per_tile") else "f32") if storage == "f32" else "bf16 rows, f32 arithmetic",
        "data": "synthetic (rows N(0,1) seed 1234; dictionary seed 99; query words seed 7; BASELINE.md C4/C3)",
        "config": {
            "workload": f"C4 on one GPU per rank: {n} docs x {d}-d {storage} exact cosine top-{k} "
                        f"+ {args.words_per_query} typo-tolerant words/query over a {args.dict_words}-term dictionary"
                        + ("" if kw is None else
                           (f" + all-rules keyword search over a coherent {n_total if row_sharded else n}-document text corpus (BASELINE.md C4: corpus as C1 "
                            "scaled; queries out of the documents, misspelled) + hybrid merge" if args.kw_corpus == "coherent"
                            else " + all-rules keyword search over the round-3 hashed index (300 frequent words) + hybrid merge")),
            "queries_per_step_per_gpu": Q, "words_per_step_per_gpu": n_words_q,
            "queries_per_hbm_sweep": store.max_batch,
            "scan_math": ("level 0: int8 candidate sweep (v_mfma_i32_16x16x64_i8) over an int8 copy of the normalised rows; what it cannot prove: "
                          if stats.get("i8_bytes_per_tile") else "") +
                         os.environ.get("MSI_VS_SCAN_MATH", "bf16x2") + " candidate scan of the f32 rows (split hi/lo in registers, bf16 query fragments, f32 accumulate)"
                         "; every answer = exact f32 reference rescoring of K' candidates with an exactness proof",
            "sharding": ("rows sharded (%d per GPU of %d), same query batch on every GPU, ONE packed all_gather of "
                         "per-shard top-k (RCCL) + device k-way merge" % (n, n_total)) if row_sharded else
                        "queries sharded, index replicated per GPU, ONE packed all-gather of per-rank top-k per step; exchange path: " + exchange_path,
            "rccl_ranks_seen": rccl_ranks_seen,
            "per_rank_values": [round(Q * args.steps / e, 1) for e in getattr(env, "per_rank_elapsed", [elapsed])],
            "keyword_cap_measured": round(sum(kw_only_by_rank), 1) if kw_only_by_rank else None,
            "keyword_only_queries_per_s_by_rank": [round(x, 1) for x in kw_only_by_rank] if kw_only_by_rank else None,
            "vector_only_queries_per_s_by_rank": [round(x, 1) for x in vec_only_by_rank] if vec_only_by_rank else None,
            "keyword_callers_per_rank": kw_threads if kw is not None else 0, "host_cpus_granted": granted_cpus(),
            # N > 1: all ranks share the box's granted CPUs, and a keyword search costs host CPU whichever GPU runs its sets — the
            # keyword leg of the whole job cannot exceed granted CPUs / host CPU per query (0.66 ms per fresh query at N = 1 with
            # the postings staged, profiles/r6_kw_stage.log: 17.0 k q/s at 12.0 of 16 CPUs; round 5: 1.04 ms), so with 16 granted
            # CPUs the hybrid weak-scaling curve flattens near 1.4 GPUs' worth whatever RCCL does; the vector leg scales
            "keyword_cap_predicted": round(granted_cpus() / 0.66e-3, 0) if kw is not None else None,
            "predicted_cap_is": "granted CPUs / 0.66 ms of host CPU per keyword query on a fresh query stream (round 6, postings staged; legs."
                                "keyword_host_cpu_ms_per_query at N = 1): every rank's searches draw on the same granted CPUs, so the whole "
                                "job's keyword leg cannot exceed it whatever the number of GPUs; keyword_cap_measured = the ranks' keyword-only "
                                "rates, measured at the same time, summed",
            "step_includes": ["vs_scan + select + reference rescoring", "dict_lookup (scan of the first-letter range, "
                              "binary searches for the other first letters, cap logic)", "D2H of results"]
                             + ([] if kw is None else [
                                 "msi_keyword_search_ranked for every query: default criteria [words, typo, proximity, attributeRank, "
                                 "sort, wordPosition, exactness], %s, typo derivations from the index's dictionary (vocabulary %d words) "
                                 "feeding the postings, %d caller threads" % (
                                     "BASELINE's C4 text workload: a coherent %d-document corpus (title 3-6 / overview 20-60 words, "
                                     "Zipf(1.07), every database derived from the same tokens), queries = 1-%d consecutive words of a "
                                     "document with 0-2 edits and a prefix last word + the shapes of workloads/search/movies.json"
                                     % (n_total if row_sharded else n, args.kw_terms) if args.kw_corpus == "coherent" else
                                     "%d words per query over the 300 most frequent words of an independently hashed index" % args.kw_terms,
                                     args.kw_dict_words, kw_threads),
                                 "hybrid merge (semanticRatio 0.5) of the vector list with the keyword list (global scores)"]),
            "step_includes_short": ["vs_scan+select+f32 rescoring+proof", "dict_lookup (typo derivations)", "D2H of results"]
                                   + ([] if kw is None else ["keyword search, 7 criteria, detailed", "hybrid merge (ratio 0.5)"]),
            "step_excludes": ["keyword leg", "hybrid merge"] if kw is None else [],
            "inexact_queries_last_step": n_inexact,
            "keyword_corpus": args.kw_corpus if kw is not None else None,
            "keyword_stream": None if kw is None else (
                f"fresh: every timed step runs {Q} queries the engine meets for the first time ({kw['stream_steps']} x {Q} distinct ones "
                f"behind a {kw['prime']}-query primer)"
                + ("" if kw["stream_distinct"] == kw["stream_steps"] else
                   f"; N > 1: the index derivation pass was time-boxed, {kw['stream_distinct']} distinct steps cycled")
                if kw["stream_steps"] else f"cycle: {kw['prime']} distinct queries, cycled"),
            "keyword_index_derivation_seconds": kw.get("index_derivation_seconds") if kw is not None else None,
            "setup_seconds": round(setup_s, 1),
        },
        "roofline": scan_roofline(args, env, store, scan_n, scan_ms, algo_bytes, kname, must_contain=("false",), kernel_prefix=kprefix,
                                  measured=pmc_early),
        "legs": legs,
        "latency": latency,
        "rows_sharded": None,
        "dict_lookup": {"launches_timed": match_n, "avg_launch_ms": round(match_ms / max(1, match_n), 4),
                        "words_per_launch": n_words_q,
                        "words_per_s_kernel_only": round(n_words_q / (match_ms / max(1, match_n) * 1e-3), 1) if match_n else None},
    }
    if want_extra:
        out["rows_sharded"] = guarded_extra(out)
    if env.check:
        phase("c4: cpu baselines")
        t_vec, cores, sample, vec_by_threads = cpu_vector_baseline(cpu_rows, n, d, k)
        t_word, typo_by_threads, typo_cores = 0.0, None, None
        if gdict is not None:
            t_word, _, typo_cores, typo_by_threads = cpu_typo_baseline(words, concat, off, args.cpu_sample_words)
        # the keyword leg's CPU port (tools/bin/ranked_bench_cpu: the product's host logic over plain-C++ dense bitsets + the CPU
        # dictionary walk — never the product), a bounded sample at the step's own size: all 7 rules, detailed scores
        kw_cpu = None
        if kw is not None:
            kw_cpu = cpu_keyword_baseline(n_total if row_sharded else n, args.kw_dict_words, args.kw_terms)
        t_kw = 1.0 / kw_cpu["queries_per_s"] if kw_cpu and kw_cpu.get("queries_per_s") else 0.0
        per_query = t_vec + args.words_per_query * t_word + t_kw
        out["cpu_baseline"] = {
            "value": round(1.0 / per_query, 3), "unit": "queries/s", "cores": cores, "kind": "port",
            "value_covers": "the whole step: vector leg + typo words + keyword leg (all 7 rules), each at its best thread count, "
                            "run one after the other as the GPU step's legs are",
            "keyword_queries_per_s": kw_cpu.get("queries_per_s") if kw_cpu else None, "keyword": kw_cpu,
            "sample": f"vector: 16 queries x {sample} rows x {d}-d, scaled x{n / sample:.0f} to {n} rows; typo: "
                      f"{args.cpu_sample_words} words over the full {args.dict_words}-term dictionary (threads take queries from a "
                      "shared counter); each leg timed at the container's CPU quota AND at every visible hardware thread, the "
                      "better one kept",
            "vector_queries_per_s": round(1.0 / t_vec, 3), "vector_queries_per_s_by_threads": vec_by_threads,
            "typo_words_per_s": round(1.0 / t_word, 1) if t_word else None, "typo_words_per_s_by_threads": typo_by_threads,
            "typo_threads": typo_cores}
        # ---- untimed parity check of this run's own results (the store of the timed steps, its query batch) ----
        phase("c4: parity, vector + typo")
        from oracle import parity
        nqc = min(args.parity_queries, Q)
        got = store.search(q_t[:nqc].cpu().numpy(), k)     # host entry point: exhaustive reruns included
        chk = parity.TopkChecker(q_t[:nqc].cpu().numpy(), k)
        for c0, c1, rows_chunk in synth.device_rows_chunks(n, d, dev, seed=rows_seed):
            rows = rows_chunk.cpu().numpy()
            del rows_chunk
            if storage == "bf16":
                rows = synth.round_to_bf16(rows)
            chk.add_chunk(np.arange(c0, c1, dtype=np.uint32), rows)
        par = chk.verdict(*got)
        # the device-pointer path of the timed steps returned the same lists
        same = bool((out_ids[:nqc].cpu().numpy().view(np.uint32) == got[0]).all()
                    and (out_dist[:nqc].cpu().numpy().view(np.uint32) == got[1].view(np.uint32)).all())
        par["timed_path_equals_checked_path"] = same
        if gdict is not None:
            nw = min(256, n_words_q)
            gw = gdict.lookup(tq[:nw])
            tp = parity.check_typo_lookup(concat, off, tq[:nw], gw)
            par["typo"] = tp
            par["mismatches"] += tp["mismatches"]
        if not same:
            par["mismatches"] += 1
        if kw is not None:
            # keyword leg, checked against the ORACLE at the step's own size: the first `--parity-kw-queries` queries of the
            # step that was just timed, through oracle/ranking_oracle.py (the restatement the reference's snapshot searches
            # pin) reading the synthetic index's stored posting bytes — docids in order, every hit's score details and the
            # candidate counts (oracle/parity.py: KeywordLegChecker; tests/test_configs_gpu.py::test_c4_keyword_leg)
            phase("c4: parity, keyword vs oracle")
            keyword_run()                               # (the latency legs above ran other queries since the timed steps)
            first = kw_first(kw["step"] - 1)
            nk = min(args.parity_kw_queries, Q)
            kchk = parity.KeywordLegChecker(kw["lib"], kw["h"], n_total if row_sharded else n)
            prod = kchk.run_product(first, nk, k)
            kpar = kchk.verdict(first, nk, k, product=prod)
            # ... and the lists the last keyword_run of this process produced for the same queries are those lists
            same_kw = bool((prod[1] == kw["n"][:nk]).all()) and all(
                prod[0][i, :int(prod[1][i])].tolist() == kw["ids"][i, :int(prod[1][i])].tolist() for i in range(nk))
            kpar["timed_path_equals_checked_path"] = same_kw
            par["mismatches"] += 0 if same_kw else 1
            # second field: the command-list back end against the direct back end (one launch per set operation) on ALL of
            # the step's queries
            phase("c4: parity, lists vs direct back end")
            kw["step"] -= 1
            keyword_run()
            a_ids, a_n, a_sc = kw["ids"].copy(), kw["n"].copy(), kw["scores"].copy()
            kw["step"] -= 1
            os.environ["MSI_SEARCH_VM"] = "0"
            keyword_run()
            os.environ.pop("MSI_SEARCH_VM")
            kbad = 0
            for qi in range(Q):
                m = int(a_n[qi])
                if m != int(kw["n"][qi]) or a_ids[qi, :m].tolist() != kw["ids"][qi, :m].tolist() or \
                        a_sc[qi, :m].tolist() != kw["scores"][qi, :m].tolist():
                    kbad += 1
            kpar["command_lists_vs_direct_back_end"] = {"checked_queries": Q, "mismatches": kbad,
                                                        "what": "MSI_SEARCH_VM=0 against the default: docids and global scores identical"}
            par["keyword"] = kpar
            par["mismatches"] += kpar["mismatches"] + kbad
        out["parity"] = par
    if kw is not None and args.kw_features and args.kw_corpus == "coherent" and not env.child and world == 1:
        out["legs"]["keyword_with_features"] = keyword_features_leg(args, kw, Q, k, phase, n_total if row_sharded else n)
        note_hbm(torch)
        if out.get("parity") is not None and isinstance(out["legs"]["keyword_with_features"].get("parity"), dict):
            out["parity"]["mismatches"] += out["legs"]["keyword_with_features"]["parity"]["mismatches"]
    if kw is not None:
        note_hbm(torch)
        if _HBM_FREE:
            out["config"]["hbm_free_gb_min"] = round(min(_HBM_FREE), 1)     # (with the store, the cache and every caller's pools resident)
        kw_qps = legs.get("keyword_only_queries_per_s") or 0.0
        kw["lib"].rb_destroy(kw["h"])      # the runner's pools go before the children / the other configurations start
        kw = None
        if env.rank == 0 and env.world == 1 and not env.child and not args.no_pmc and args.kw_roofline:
            # the keyword leg's kernel: HBM traffic and L2 counters per query, host CPU by where it is spent (children of this run)
            out["keyword_roofline"] = keyword_roofline(n, kw_threads, kw_qps, args.kw_dict_words, args.kw_corpus)
    return out


def keyword_features_leg(args, kw, Q, k, phase, n_docs):
    """the index shapes of a real index in a timed keyword stream.  The SAME index (runner handle, staged
    postings, caller threads) gets what milli builds beside the word databases — word-prefix databases (prefixes of 1-4 bytes
    that >= 50 dictionary words share: a prefix term whose word is such a key reads them instead of enumerating its
    derivations, compute_derivations.rs:193-205) and synonyms — and a new query stream out of the same documents in which
    every eighth query opens with a quoted phrase, three in 64 end in a one- to three-letter prefix (movies.json's "t"), every
    eighth is a word or pair with synonyms and every eighth excludes a word or a phrase (rb_prepare_queries_ex flags 15).
    Untimed: one pass in which the synthetic index derives the databases these queries read, the posting cache forgets what
    searches left (the staged databases stay), a primer of Q queries.  Timed: 2 x Q queries the engine meets for the first
    time.  Parity: the first `--parity-kw-queries` of them against oracle/ranking_oracle.py."""
    import ctypes as C
    import resource
    import meilisearch_amd as ma
    from oracle import parity
    L, h = kw["lib"], kw["h"]
    lib = ma._lib.lib()
    phase("c4: keyword leg with index features (prefix databases, synonyms, phrases, negatives)")
    L.rb_enable_prefix_dbs.argtypes = [C.c_void_p, C.c_uint32]
    L.rb_enable_synonyms.argtypes = [C.c_void_p]
    L.rb_prepare_queries_ex.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint64, C.c_uint32]
    assert L.rb_enable_prefix_dbs(h, 50) == 0 and L.rb_enable_synonyms(h) == 0
    n_q = 3 * Q
    assert L.rb_prepare_queries_ex(h, n_q, args.kw_terms, 5151, 15) == 0

    def run(first, n):
        assert L.rb_run(h, first, n, k, kw["ids"].ctypes.data, kw["n"].ctypes.data, kw["scores"].ctypes.data) == 0, \
            "msi_keyword_search_ranked failed (features leg)"
    t0 = time.perf_counter()
    for first in range(0, n_q, Q):
        run(first, Q)                              # untimed: the synthetic index derives what these queries read
    derive_s = time.perf_counter() - t0
    if hasattr(L, "rb_freeze"):
        assert L.rb_freeze(h) == 0
    ma._lib.check(lib.msi_dict_reset_posting_cache(C.c_void_p(L.rb_dict(h))))
    run(0, Q)                                      # the primer
    lib.msi_search_cpu_profile_enable(1)
    cp0, cp1 = (C.c_uint64 * 8)(), (C.c_uint64 * 8)()
    pc0, pc1 = (C.c_uint64 * 4)(), (C.c_uint64 * 4)()
    lib.msi_search_cpu_profile(cp0)
    lib.msi_dict_posting_cache_stats(C.c_void_p(L.rb_dict(h)), pc0)
    ru0 = resource.getrusage(resource.RUSAGE_SELF)
    t0 = time.perf_counter()
    run(Q, Q)
    run(2 * Q, Q)
    dt = time.perf_counter() - t0
    ru1 = resource.getrusage(resource.RUSAGE_SELF)
    lib.msi_search_cpu_profile(cp1)
    lib.msi_dict_posting_cache_stats(C.c_void_p(L.rb_dict(h)), pc1)
    lib.msi_search_cpu_profile_enable(0)
    m = max(1.0, float(cp1[0] - cp0[0]))
    d = [(cp1[i] - cp0[i]) / 1e3 / m for i in range(8)]
    res = {"queries_per_s": round(2 * Q / dt, 1), "timed_queries": 2 * Q, "fresh": True,
           "host_cpus_used": round((ru1.ru_utime + ru1.ru_stime - ru0.ru_utime - ru0.ru_stime) / dt, 2),
           "host_cpu_us_per_query": {"search_threads": round(d[1], 1), "index_callbacks": round(d[5], 1),
                                     "index_callbacks_share": round(d[5] / max(1e-9, d[1]), 3),
                                     "typo_derivations": round(d[4], 1), "list_submit_and_wait": round(d[2], 1)},
           "lists_per_query": round((cp1[7] - cp0[7]) / m, 2),
           "posting_cache_hit_rate": round((pc1[0] - pc0[0]) / max(1, (pc1[0] - pc0[0]) + (pc1[1] - pc0[1])), 4),
           "index_derivation_seconds": round(derive_s, 1),
           "stream": "rb_prepare_queries_ex flags 15: phrases, 1-3 letter prefixes over the word-prefix databases (threshold 50), "
                     "synonyms, negative terms; callbacks of the prefix / synonym databases are the synthetic index's own "
                     "(memoised scans: an LMDB read would be microseconds)"}
    nk = min(args.parity_kw_queries, Q)
    if not args.no_cpu_baseline and nk:
        phase("c4: parity, keyword leg with features vs oracle")
        kchk = parity.KeywordLegChecker(L, h, n_docs)
        prod = kchk.run_product(Q, nk, k)
        res["parity"] = kchk.verdict(Q, nk, k, product=prod)
        qs_ = [kchk.index.query(Q + i) for i in range(nk)]
        res["parity"]["queries_with_a_phrase"] = sum(1 for q in qs_ if '"' in q)
        res["parity"]["queries_ending_in_a_short_prefix"] = sum(1 for q in qs_ if q.split() and len(q.split()[-1]) <= 3 and '"' not in q.split()[-1])
    return res


def also_configs(args, env):
    """The other BASELINE configurations inside the default invocation (the driver only runs the default command): short
    runs of C2, C3 and the C5 shard, each with its own roofline, cpu_baseline and parity objects.  Untimed extras of the C4
    line — `value` is not affected.  Each leg frees what it allocated before the next one starts."""
    import copy
    import gc
    out = {}
    for cfg, fn, steps in (("c2", run_c2, 10), ("c3", run_c3, 5), ("c5", run_c5, 5)):
        a = copy.copy(args)
        a.config, a.steps, a.warmup = cfg, steps, (4 if cfg == "c5" else 2)   # (c5's steps cycle through four query sets)
        a.rows = a.dim = a.k = a.queries = a.storage = None
        a.parity_queries = 16
        t0 = time.time()
        phase(f"also: {cfg}")
        try:
            line = fn(a, env)
            keep = {k_: line[k_] for k_ in ("metric", "value", "unit", "ms_per_step", "p50_latency_ms", "dtype", "config", "roofline",
                                             "cpu_baseline", "parity", "dict_roofline", "cache_stream", "densities") if k_ in line}
            keep["steps"], keep["seconds"] = steps, round(time.time() - t0, 1)
            out[cfg] = keep
        except Exception as e:      # noqa: BLE001 - an extra: the C4 line must still be produced
            out[cfg] = {"error": repr(e)[:300]}
        gc.collect()
        env.torch.cuda.empty_cache()
    # the vector leg on clustered rows, at C2's and C4's sizes
    out["clustered"] = {}
    for tag, n, d, Q, steps in (("c2-size", 1_000_000, 384, 256, 5), ("c4-size", 10_000_000, 768, 768, 3)):
        a = copy.copy(args)
        a.steps = steps
        t0 = time.time()
        phase(f"also: clustered {tag}")
        try:
            line = run_clustered(a, env, n, d, Q, tag)
            line["seconds"] = round(time.time() - t0, 1)
            out["clustered"][tag] = line
        except Exception as e:      # noqa: BLE001
            out["clustered"][tag] = {"error": repr(e)[:300]}
        gc.collect()
        env.torch.cuda.empty_cache()
    return out


# --------------------------------------------------------------------------------------------------- C2

def run_c2(args, env):
    torch, ma, ctx, dev = env.torch, env.ma, env.ctx, env.dev
    from meilisearch_amd import synth
    n = args.rows or 1_000_000
    d = args.dim or 384
    k = args.k or 20
    Q = args.queries or 256
    storage = args.storage or "f32"
    rows_t = synth.device_rows(n, d, dev, seed=1234)
    ids_t = torch.arange(n, dtype=torch.int32, device=dev)
    torch.cuda.synchronize()
    store = ma.GpuStore(ctx, d, storage=storage)
    store.upload_device(ids_t, rows_t)
    q_t = synth.device_queries(Q, d, dev, seed=5678 + env.rank)
    out_ids = torch.zeros((Q, k), dtype=torch.int32, device=dev)
    out_dist = torch.zeros((Q, k), dtype=torch.float32, device=dev)
    out_cnt = torch.zeros(Q, dtype=torch.int32, device=dev)
    inexact = torch.zeros(Q, dtype=torch.int32, device=dev)

    def step():
        store.search_device(q_t, k, out_ids, out_dist, out_cnt, inexact)
        ctx.synchronize()
        return out_ids.cpu(), out_dist.cpu(), out_cnt.cpu()

    for _ in range(args.warmup):
        step()
    ctx.set_profiling(True)
    store.scan_time()
    elapsed, lat = env.timed(step, args.steps, 0)
    scan_n, scan_ms = store.scan_time()
    ctx.set_profiling(False)
    if env.rank != 0:
        return None
    kname, kprefix, algo_bytes = sweep_kernel(store.stats(), n)
    out = {
        "metric": "vector k-NN queries/sec (1M docs x 384-d, exact cosine top-20)",
        "value": round(Q * env.world * args.steps / elapsed, 2), "unit": "queries/s",
        "n_gpus": env.world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(elapsed / args.steps * 1e3, 4), "p50_latency_ms": round(statistics.median(lat), 4),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": ("f32 (every distance: the reference's f32 arithmetic on the f32 rows; candidates: int8 MFMA sweep of a quantised copy)"
                  if store.stats().get("i8_bytes_per_tile") else "f32") if storage == "f32" else "bf16 rows, f32 arithmetic",
        "data": "synthetic (rows N(0,1) seed 1234, queries seed 5678; BASELINE.md C2)",
        "config": {"workload": f"C2: {n} docs x {d}-d {storage} exact cosine top-{k}, {Q} queries per step per GPU",
                   "queries_per_hbm_sweep": store.max_batch, "inexact_queries_last_step": int(inexact.sum().item())},
        "roofline": scan_roofline(args, env, store, scan_n, scan_ms, algo_bytes, kname, must_contain=("false",), kernel_prefix=kprefix),
    }
    if env.check:
        cpu_rows = rows_t[:min(n, args.cpu_sample_rows)].cpu().numpy()
        t_vec, cores, sample, by_threads = cpu_vector_baseline(cpu_rows, n, d, k)
        out["cpu_baseline"] = {"value": round(1.0 / t_vec, 3), "unit": "queries/s", "cores": cores, "kind": "port",
                               "sample": f"16 queries x {sample} rows x {d}-d, scaled x{n / sample:.0f}; timed at the CPU quota and "
                                         "at every visible thread, the better kept", "queries_per_s_by_threads": by_threads}
        from oracle import parity
        nqc = min(args.parity_queries, Q)
        qh = q_t[:nqc].cpu().numpy()
        got = store.search(qh, k)
        chk = parity.TopkChecker(qh, k)
        for c0 in range(0, n, 1_000_000):
            c1 = min(n, c0 + 1_000_000)
            rows = rows_t[c0:c1].cpu().numpy()
            chk.add_chunk(np.arange(c0, c1, dtype=np.uint32), synth.round_to_bf16(rows) if storage == "bf16" else rows)
        par = chk.verdict(*got)
        same = bool((out_ids[:nqc].cpu().numpy().view(np.uint32) == got[0]).all())
        par["timed_path_equals_checked_path"] = same
        par["mismatches"] += 0 if same else 1
        out["parity"] = par
    return out


# --------------------------------------------------------------------------------------------------- clustered rows
def run_clustered(args, env, n, d, Q, tag):
    """The vector leg on data shaped like an embedding corpus: 10 000 clusters whose members sit 1e-3 ... 1e-2
    (1 - cos) from their centre, 1 % exact duplicates, queries that are stored rows moved by 5e-3 — thousands of rows inside
    the bf16x2 proof's margin of the k-th neighbour at 10 M rows.  Through the HOST entry point (msi_vs_search): that is where
    an unproven query is re-run (bf16x3 second opinion, then exhaustively) and where the first pass's arithmetic adapts to the
    share of unproven queries (msi_vs.hip: x2_flagged_ema).  Reports how the queries were resolved, queries/s, and the HBM
    roofline fraction of the EFFECTIVE bytes — every tile every pass streamed, over the passes' kernel time and over wall time."""
    torch, ma, ctx, dev = env.torch, env.ma, env.ctx, env.dev
    from meilisearch_amd import synth
    k = 20
    rows_t, _ = synth.device_rows_clustered(n, d, dev, seed=4321)
    ids_t = torch.arange(n, dtype=torch.int32, device=dev)
    torch.cuda.synchronize()
    store = ma.GpuStore(ctx, d, storage="f32")
    store.upload_device(ids_t, rows_t)
    q_t = synth.device_queries_near_rows(rows_t, Q, seed=8765)
    qh = q_t.cpu().numpy()
    for _ in range(2):
        store.search(qh, k)          # warm-up: also lets the first pass's arithmetic settle on this data
    s0 = store.stats()
    ctx.set_profiling(True)
    store.scan_time()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        got = store.search(qh, k)
    wall = time.perf_counter() - t0
    scan_n, scan_ms = store.scan_time()
    ctx.set_profiling(False)
    s1 = store.stats()
    nq = Q * args.steps
    tiles = s1["scan_tiles"] - s0["scan_tiles"]
    tiles8 = s1["i8_scan_tiles"] - s0["i8_scan_tiles"]            # (the part of them that streamed the int8 copy)
    eff_bytes = (tiles - tiles8) * s1["bytes_per_tile"] + tiles8 * s1["i8_bytes_per_tile"]
    x2, x3 = s1["x2_sweeps"] - s0["x2_sweeps"], s1["x3_first_sweeps"] - s0["x3_first_sweeps"]
    second, exh = s1["second_opinion_queries"] - s0["second_opinion_queries"], s1["exhaustive_reruns"] - s0["exhaustive_reruns"]
    one_sweep = ((n + 15) // 16) * s1["bytes_per_tile"]
    out = {
        "metric": f"vector k-NN queries/sec on clustered rows ({tag}: {n} x {d}-d f32, top-{k}), host entry point",
        "value": round(nq / wall, 1), "unit": "queries/s", "steps": args.steps, "queries_per_step": Q,
        "data": "synthetic clustered (synth.device_rows_clustered seed 4321: 10 000 centres, members 1e-3..1e-2 from their centre "
                "(1 - cos, log-uniform), 1 % exact duplicates; queries = stored rows moved by 5e-3, seed 8765)",
        "resolved": {"queries": nq, "int8_sweeps": int(s1["i8_sweeps"] - s0["i8_sweeps"]), "bf16x2_sweeps": int(x2), "bf16x3_first_sweeps": int(x3), "second_opinion_bf16x3_queries": int(second),
                     "exhaustive_queries": int(exh),
                     "fraction_needing_more_than_the_first_pass": round((second + (exh if x3 else 0)) / max(1, nq), 4)},
        "roofline": {"kernel": "vs_scan_i8_kernel + vs_scan_kernel (every pass: samples, int8 sweeps, f32 levels)", "bound": "hbm",
                     "achieved": round(eff_bytes / max(1e-9, scan_ms * 1e-3) / 1e9, 1), "peak": 8000.0, "unit": "GB/s",
                     "frac": round(eff_bytes / max(1e-9, scan_ms * 1e-3) / 1e9 / 8000.0, 4),
                     "effective_bytes": int(eff_bytes), "f32_passes_over_the_store": round((tiles - tiles8) * s1["bytes_per_tile"] / one_sweep, 2),
                     "int8_passes_over_the_copy": round(tiles8 / max(1, (n + 15) // 16), 2),
                     "scan_kernel_ms": round(scan_ms, 3), "launches_timed": scan_n,
                     "end_to_end_GBps": round(eff_bytes / wall / 1e9, 1), "end_to_end_frac": round(eff_bytes / wall / 1e9 / 8000.0, 4),
                     "timing": "HIP events on the scan's launch stream (kernel) / wall clock around msi_vs_search (end to end: H2D of the "
                               "queries, selection, rescoring, proof, D2H included)", "traffic": None},
    }
    if env.check:
        from oracle import parity
        nqc = min(16, Q)
        chk = parity.TopkChecker(qh[:nqc], k)
        for c0 in range(0, n, 1_000_000):
            c1 = min(n, c0 + 1_000_000)
            chk.add_chunk(np.arange(c0, c1, dtype=np.uint32), rows_t[c0:c1].cpu().numpy())
        out["parity"] = chk.verdict(got[0][:nqc], got[1][:nqc], got[2][:nqc])
    return out


# --------------------------------------------------------------------------------------------------- C3

def run_c3(args, env):
    torch, ma, ctx, dev = env.torch, env.ma, env.ctx, env.dev
    from meilisearch_amd import synth
    B = args.queries or 8192
    words = synth.make_dictionary(args.dict_words, seed=99)
    concat, off = synth.flatten_words(words)
    gdict = ma.GpuDictionary(ctx, concat=concat, offsets=off)
    tq = synth.make_typo_queries(words, B, seed=7 + env.rank)
    if os.environ.get("MSI_BENCH_C3_SORTED") == "1":   # experiment: the batch in dictionary order (the same first letters side by side)
        tq = sorted(tq, key=lambda t: t[0].encode("utf-8"))
    qb, qoff, qfl = ma.pack_queries(tq)
    qb_t = torch.from_numpy(qb).to(dev)
    qoff_t = torch.from_numpy(qoff.astype(np.int32)).to(dev)
    qfl_t = torch.from_numpy(qfl).to(dev)
    one_t = torch.zeros((B, 150), dtype=torch.int32, device=dev)
    two_t = torch.zeros((B, 50), dtype=torch.int32, device=dev)
    one_c = torch.zeros(B, dtype=torch.int32, device=dev)
    two_c = torch.zeros(B, dtype=torch.int32, device=dev)

    def step():
        gdict.lookup_device(qb_t, qoff_t, qfl_t, B, one_t, one_c, two_t, two_c)
        ctx.synchronize()
        return one_c.cpu(), two_c.cpu(), one_t.cpu(), two_t.cpu()

    for _ in range(args.warmup):
        step()
    p0 = gdict.stats()["pairs_scanned"]
    step()
    dp_pairs = gdict.stats()["pairs_scanned"] - p0
    ctx.set_profiling(True)
    gdict.match_time()
    elapsed, lat = env.timed(step, args.steps, 0)
    match_n, match_ms = gdict.match_time()
    ctx.set_profiling(False)
    if env.rank != 0:
        return None
    # algorithmic work of one launch: every word of each query's first-letter range passes the 10-byte filter
    # (8-byte signature + 2 bytes of lengths); survivors load their 16-byte slot for the DP
    first = {}
    wb = [w.encode("utf-8") for w in words]
    import bisect
    range_words = 0
    for w, _, _ in tq:
        c0 = w.encode("utf-8")[:len(w[0].encode("utf-8"))]
        if c0 not in first:
            lo = bisect.bisect_left(wb, c0)
            hi = bisect.bisect_left(wb, c0[:-1] + bytes([c0[-1] + 1])) if c0[-1] < 255 else len(wb)
            first[c0] = hi - lo
        range_words += first[c0]
    algo_bytes = range_words * 8 + dp_pairs * 18
    avg_ms = match_ms / max(1, match_n)
    achieved = algo_bytes / (avg_ms * 1e-3) / 1e9 if match_n else 0.0
    # VALU view (SURVEY §8 d: this kernel is integer-VALU bound, the dictionary is cache resident): wave-level
    # instruction counts of the two inner loops, counted in the gfx950 ISA of this build
    # counted in the ISA of this build (hipcc -S of msi_dict.hip, dict_lookup_kernel): the filter step of 64 words is 31 VALU
    # + 2 loads + 1 LDS store; a drain of 64 queued pairs is ~920 VALU to unpack the 16-byte slots into code points and set
    # the band up, then 78-102 VALU per word char (two loop variants; ~9 chars per word on this dictionary)
    I_FILTER, I_DP = 34, 920 + 90 * 9
    valu_instr = range_words / 64.0 * I_FILTER + dp_pairs / 64.0 * I_DP
    valu_peak = 256 * 4 * 2.4e9 / 2    # 1024 SIMDs, one wave64 VALU instruction per 2 cycles
    out = {
        "metric": "typo-tolerant term lookups/sec (2M-term dictionary, 1/2 typos by length, 30% prefix)",
        "value": round(B * env.world * args.steps / elapsed, 2), "unit": "words/s",
        "n_gpus": env.world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(elapsed / args.steps * 1e3, 4), "p50_latency_ms": round(statistics.median(lat), 4),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u8 / u32 code points (integer)",
        "data": "synthetic (dictionary seed 99, query words seed 7; BASELINE.md C3)",
        "config": {"workload": f"C3: {len(words)}-term dictionary, {B} query words per step per GPU, results copied to host",
                   "first_letter_range_words_per_query": round(range_words / B, 1), "dp_pairs_per_query": round(dp_pairs / B, 1),
                   "hits_one": int(one_c.sum().item()), "hits_two": int(two_c.sum().item())},
        # C3 is integer-VALU work over a cache-resident dictionary — its roofline is the VALU issue peak (the
        # object below: `roofline` IS `dict_roofline`, whose `measured` part comes from live SQ counters); the bytes its range
        # scan streams come out of L2 / Infinity Cache, not HBM, and are reported as what they are
        "cache_stream": {"bytes_per_launch": int(algo_bytes), "GBps": round(achieved, 1),
                         "is": "8-byte filter words of every first-letter range + 16-byte slots of the survivors, served by L2 / "
                               "Infinity Cache (the staged dictionary is 52 MB): not HBM traffic, no HBM roofline is claimed",
                         "avg_launch_ms": round(avg_ms, 4), "launches_timed": match_n},
        "dict_roofline": {"kernel": "dict_other_kernel + dict_lookup_kernel", "bound": "valu", "unit": "wave-instructions/s",
                          "achieved": round(valu_instr / (avg_ms * 1e-3), 1) if match_n else 0.0, "peak": valu_peak,
                          "frac": round(valu_instr / (avg_ms * 1e-3) / valu_peak, 4) if match_n else 0.0,
                          "algorithmic_wave_instructions_per_launch": int(valu_instr),
                          "model": f"{I_FILTER} per 64-word filter step + {I_DP} per 64-pair DP drain = 920 set-up + 90 per word char x 9 chars (ISA count, DESIGN §4.3)"},
    }
    if env.check and not args.no_pmc:
        # measured, not modelled: SQ counters of dict_lookup_kernel from a 2-step child of this configuration
        child = [sys.executable, os.path.abspath(__file__), "--config", "c3", "--steps", "2", "--warmup", "1", "--no-cpu-baseline",
                 "--no-pmc", "--queries", str(B), "--dict-words", str(args.dict_words)]
        rows = pmc_rows(child, ["SQ_INSTS_VALU", "SQ_ACTIVE_INST_VALU", "SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_WAVES"],
                        "dict_lookup_kernel", env=dict(os.environ, MSI_BENCH_CHILD="1"))
        if rows and rows.get("SQ_INSTS_VALU"):
            # (two dict_lookup_kernel dispatches per lookup: the bit-parallel one and the one for queries above 64 chars, whose
            # workgroups leave at once on this batch — only the dispatches that did work count)
            top = max(rows["SQ_INSTS_VALU"])
            keep = [i for i, v in enumerate(rows["SQ_INSTS_VALU"]) if v > 0.1 * top]
            mean = {c: sum(v[i] for i in keep) / len(keep) for c, v in rows.items() if v and len(v) == len(rows["SQ_INSTS_VALU"])}
            insts = mean["SQ_INSTS_VALU"]
            out["dict_roofline"]["measured"] = {
                "source": "live: rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES child of this run, "
                          "mean per dict_lookup_kernel dispatch",
                "counters_per_launch": {c: round(v, 1) for c, v in mean.items()},
                "valu_wave_instructions_per_s": round(insts / (avg_ms * 1e-3), 1) if match_n else None,
                "frac_of_valu_issue_peak": round(insts / (avg_ms * 1e-3) / valu_peak, 4) if match_n else None,
                "valu_active_per_wave_cycle": round(mean.get("SQ_ACTIVE_INST_VALU", 0.0) / max(1.0, mean.get("SQ_WAVE_CYCLES", 1.0)), 4),
                "model_over_measured_instructions": round(valu_instr / max(1.0, insts), 3)}
    rf = dict(out["dict_roofline"])
    if rf.get("measured"):      # the measured counters are the roofline's numbers when they exist, the ISA-count model else
        rf["achieved"] = rf["measured"]["valu_wave_instructions_per_s"]
        rf["frac"] = rf["measured"]["frac_of_valu_issue_peak"]
        rf["achieved_is"] = "measured: SQ_INSTS_VALU per launch / the kernels' average launch time"
    out["roofline"] = rf
    if env.check:
        t_all, t_one, cores, by_threads = cpu_typo_baseline(words, concat, off, args.cpu_sample_words)
        out["cpu_baseline"] = {"value": round(1.0 / t_all, 1), "unit": "words/s", "cores": cores, "kind": "port",
                               "sample": f"{args.cpu_sample_words} words over the full dictionary, threads taking queries from a "
                                         "shared counter, timed at the CPU quota and at every visible thread (the better kept); "
                                         "one thread: %.1f words/s" % (1.0 / t_one), "words_per_s_by_threads": by_threads}
        from oracle import parity
        nw = min(512, B)
        got = gdict.lookup(tq[:nw])
        par = parity.check_typo_lookup(concat, off, tq[:nw], got)
        o1, o2 = one_c[:nw].cpu().numpy(), two_c[:nw].cpu().numpy()
        same = all(int(o1[i]) == got[i][0].size and int(o2[i]) == got[i][1].size for i in range(nw))
        par["timed_path_equals_checked_path"] = bool(same)
        par["mismatches"] += 0 if same else 1
        out["parity"] = par
    return out


# --------------------------------------------------------------------------------------------------- C5

def run_c5(args, env):
    torch, ma, ctx, dev = env.torch, env.ma, env.ctx, env.dev
    from meilisearch_amd import ranking as R
    from meilisearch_amd import synth
    n = args.rows or 12_500_000
    d = args.dim or 1024
    k = args.k or 1000
    storage = args.storage or "bf16"
    store = ma.GpuStore(ctx, d, storage)
    rows_t = synth.device_rows(n, d, dev, seed=1234 + env.rank)
    ids_t = torch.arange(n, dtype=torch.int32, device=dev)
    torch.cuda.synchronize()
    store.upload_device(ids_t, rows_t)
    B = args.queries or store.max_batch
    q_t = synth.device_queries(B, d, dev, seed=5678 + env.rank)
    out_ids = torch.zeros((B, k), dtype=torch.int32, device=dev)
    out_dist = torch.zeros((B, k), dtype=torch.float32, device=dev)
    out_cnt = torch.zeros(B, dtype=torch.int32, device=dev)
    inexact = torch.zeros(B, dtype=torch.int32, device=dev)
    nt = 3
    FILTER, POST, UNI = 0, 1, 1 + 3 * nt
    pool = ma.BitsPool(ctx, n, UNI + 5 * B)
    rng = np.random.default_rng(31)
    words64 = (n + 63) // 64
    dens = [(0.30, 0.05, 0.02), (0.10, 0.02, 0.01), (0.02, 0.005, 0.002)]
    terms, slot = [], POST
    for i in range(nt):
        sl = []
        for p in dens[i % 3]:
            bits = rng.random(words64 * 64) < p
            pool.set_from_words(slot, np.packbits(bits, bitorder="little").view(np.uint64))
            sl.append(slot)
            slot += 1
        terms.append((sl[0], sl[1], sl[2], 2 if i % 2 else 1))
    nodes = [(i, i, t[0], t[1], t[2], t[3]) for i, t in enumerate(terms)]
    batch = R.RankBatch(pool, [(nodes, nt, UNI + 5 * i, UNI + 5 * i + 1) for i in range(B)])
    fptr = pool.device_ptr(FILTER)
    # ---- the rerank as BASELINE.md C5 writes it: the FULL default criteria (words, typo, proximity, attributeRank, sort,
    # wordPosition, exactness; detailed scores) over a text index of the same documents, every query ranked inside its own
    # top-1000 — msi_keyword_search_ranked with the candidate set as its universe, one caller thread per query of the batch.
    # The text index is the coherent corpus of tools/ranked_bench.cpp at this shard's size (200 000-word vocabulary).
    import ctypes as C
    kw_so = os.environ["MSI_RUNNER_SO"] if env.emulated else os.path.join(ROOT, "tools", "bin", "libmsi_rankedbench.so")
    kwl = C.CDLL(kw_so)
    kwl.rb_create_corpus.restype = C.c_void_p
    kwl.rb_create_corpus.argtypes = [C.c_uint64, C.c_uint32, C.c_uint64]
    kwl.rb_attach.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint64]
    kwl.rb_prepare_queries.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint64]
    kwl.rb_run.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p]
    kwl.rb_run_universes.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p, C.c_uint32] + [C.c_void_p] * 6
    kwl.rb_destroy.argtypes = [C.c_void_p]
    kwh = kwl.rb_create_corpus(n, 200_000, 43)
    assert kwl.rb_attach(kwh, ctx.handle, B, 256, 1024) == 0, "keyword runner: rb_attach failed"
    kwl.rb_prepare_queries(kwh, 4 * B, 3, 777 + env.rank)
    rr_ids, rr_n, rr_sc = np.zeros((B, 20), np.uint32), np.zeros(B, np.uint32), np.zeros((B, 20), np.float64)
    for first in range(0, 4 * B, B):        # untimed: the index derives the databases these queries read
        assert kwl.rb_run(kwh, first, B, 20, rr_ids.ctypes.data, rr_n.ctypes.data, rr_sc.ctypes.data) == 0
    rr_step = [0]

    def step():
        store.search_device(q_t, k, out_ids, out_dist, out_cnt, inexact, filter_ptr=fptr, filter_nbits=n)
        ctx.synchronize()
        u_ids = np.ascontiguousarray(out_ids.cpu().numpy().view(np.uint32))      # what the Rust caller receives: the candidates
        u_cnt = np.ascontiguousarray(out_cnt.cpu().numpy().view(np.uint32))
        first = (rr_step[0] * B) % (4 * B)
        rr_step[0] += 1
        assert kwl.rb_run_universes(kwh, first, B, 20, u_ids.ctypes.data, u_cnt.ctypes.data, k, rr_ids.ctypes.data, rr_n.ctypes.data,
                                    rr_sc.ctypes.data, None, None, None) == 0
        return rr_ids, rr_n

    def step_fast():
        # rounds 1-3: the Words -> Typo prefix of the criteria as one bit-sliced kernel over synthetic term sets, the top-1000
        # of every query turned into its universe on the device (same stream: no sync, no copy)
        store.search_device(q_t, k, out_ids, out_dist, out_cnt, inexact, filter_ptr=fptr, filter_nbits=n)
        pool.set_from_docid_lists_device(UNI, 5, out_ids, out_cnt)
        return batch.run(R.TERMS_LAST, True, 0, 20)

    def knn_only():
        store.search_device(q_t, k, out_ids, out_dist, out_cnt, inexact, filter_ptr=fptr, filter_nbits=n)
        ctx.synchronize()

    # three filter densities.  The 1 % line is the object's own `value` (the round-2/3 line); all three are
    # under `densities`, each with its roofline, the bytes it streamed against the bytes of the allowed rows, and its parity.
    per_density = {}
    main = None
    # (the rocprofv3 --pmc child of this leg counts the line's own density only: its `traffic` is held against the 1 %
    # launches' algorithmic bytes — with all three in the child the 10 % launches were the ones that got averaged)
    for sel_d in ((0.01,) if env.child else (0.10, 0.01, 0.001)):
        fb = synth.random_bitset_words(n, sel_d, seed=31)
        pool.set_from_words(FILTER, fb)
        for _ in range(args.warmup):
            step()
        ctx.set_profiling(True)
        store.scan_time()
        t0s = store.stats()["scan_tiles"]
        l0s = store.stats()["scan_launches"]
        elapsed, lat = env.timed(step, args.steps, 0)
        scan_n, scan_ms = store.scan_time()
        ctx.set_profiling(False)
        t_k = time.perf_counter()
        for _ in range(args.steps):
            knn_only()
        knn_ms = (time.perf_counter() - t_k) / args.steps * 1e3
        step_fast()
        t_k = time.perf_counter()
        for _ in range(args.steps):
            step_fast()
        ctx.synchronize()
        fast_ms = (time.perf_counter() - t_k) / args.steps * 1e3
        st = store.stats()
        if env.rank != 0:
            continue
        # with a candidate filter a sweep visits ITEMS of 16 rows (msi_vs_filter_stats: counted by the device): the allowed rows
        # of a region compacted, or its tiles that hold an allowed row where that moves fewer bytes per unit of bandwidth
        allowed_tiles = int(np.count_nonzero(fb.view(np.uint16)[: (n + 15) // 16]))
        n_allowed = int(np.unpackbits(fb.view(np.uint8), bitorder="little")[:n].sum())
        fstats = store.filter_stats()
        assert fstats["allowed_rows"] == n_allowed, (fstats, n_allowed)
        algo_bytes = fstats["items"] * st["bytes_per_tile"]
        row_bytes = st["bytes_per_tile"] // 16
        line = {
            "filter_density": sel_d, "allowed_rows": n_allowed,
            "value": round(B * env.world * args.steps / elapsed, 2), "unit": "queries/s",
            "ms_per_step": round(elapsed / args.steps * 1e3, 4), "p50_latency_ms": round(statistics.median(lat), 4),
            "step_ms": [round(x, 1) for x in lat],
            "knn_only_ms_per_step": round(knn_ms, 4),
            "words_typo_fast_path_ms_per_step": round(fast_ms, 4), "words_typo_fast_path_queries_per_s": round(B / (fast_ms * 1e-3), 1),
            "tiles_streamed_per_launch": fstats["items"], "tiles_in_store": (n + 15) // 16,
            "items": dict(fstats, tiles_that_hold_an_allowed_row=allowed_tiles,
                          **{"is": "16-row items a sweep visits: compacted allowed rows (gathered, 64-byte sectors) / whole tiles (streamed)"}),
            "scan_tiles_counted_by_the_library": int((st["scan_tiles"] - t0s) / max(1, st["scan_launches"] - l0s)),
            "bytes_streamed_over_allowed_row_bytes": round(algo_bytes / max(1, n_allowed * row_bytes), 2),
            "scan_share_of_the_step": round((scan_ms / max(1, scan_n)) / (elapsed / args.steps * 1e3), 4),
            "inexact_queries_last_step": int(inexact.sum().item()),
            "roofline": scan_roofline(args, env, store, scan_n, scan_ms, algo_bytes,
                                      "vs_scan_kernel (main pass over the tiles that hold an allowed row)", must_contain=("false",))
            if sel_d == 0.01 else
            {"kernel": "vs_scan_kernel (main pass over the tiles that hold an allowed row)", "bound": "hbm",
             "achieved": round(algo_bytes / max(1e-9, scan_ms / max(1, scan_n) * 1e-3) / 1e9, 1), "peak": 8000.0, "unit": "GB/s",
             "frac": round(algo_bytes / max(1e-9, scan_ms / max(1, scan_n) * 1e-3) / 1e9 / 8000.0, 4), "traffic": None,
             "algorithmic_bytes_per_launch": int(algo_bytes), "avg_launch_ms": round(scan_ms / max(1, scan_n), 4), "launches_timed": scan_n},
        }
        if env.check:
            allowed = np.nonzero(np.unpackbits(fb.view(np.uint8), bitorder="little")[:n])[0]
            al_t = torch.from_numpy(allowed).to(dev)
            sub = synth.round_to_bf16(rows_t[al_t].cpu().numpy()) if storage == "bf16" else rows_t[al_t].cpu().numpy()
            # the CPU port of the WHOLE step at this density: the exact scan over the allowed rows (bounded sample,
            # scaled: linear in the rows it visits) + the all-criteria rerank inside a top-k universe (tools/bin/ranked_bench_cpu
            # with RB_UNIVERSE=k: the product's host logic over host bitsets; measured once, the same for every density)
            t_vec, cores, sample, by_threads = cpu_vector_baseline(sub[:min(args.cpu_sample_rows, 50_000)], allowed.size, d, k)
            if "rerank_cpu" not in per_density:
                per_density["rerank_cpu"] = cpu_keyword_baseline(n, 200_000, 3, universe=k)
            rr = per_density["rerank_cpu"]
            t_rr = 1.0 / rr["queries_per_s"] if rr.get("queries_per_s") else 0.0
            line["cpu_baseline"] = {"value": round(1.0 / (t_vec + t_rr), 3), "unit": "queries/s", "cores": cores, "kind": "port",
                                    "value_covers": "filtered exact scan + all-criteria rerank inside the top-k, one after the other",
                                    "vector_queries_per_s": round(1.0 / t_vec, 3),
                                    "rerank_queries_per_s": rr.get("queries_per_s"), "rerank": rr,
                                    "sample": f"vector: 16 queries x {sample} allowed rows x {d}-d, scaled to the {allowed.size} allowed "
                                              f"rows; rerank: ranked_bench_cpu, universes of {k} documents; each timed at the CPU quota "
                                              "and at every visible thread, the better kept", "queries_per_s_by_threads": by_threads}
            from oracle import parity
            nqc = min(16, B)
            qh = q_t[:nqc].cpu().numpy()
            kk = min(k, allowed.size)
            got = store.search(qh, k, fb, n)
            chk = parity.TopkChecker(qh, k)
            for c0 in range(0, allowed.size, 500_000):
                chk.add_chunk(allowed[c0:c0 + 500_000].astype(np.uint32), sub[c0:c0 + 500_000])
            par = chk.verdict(*got)
            knn_only()
            same = bool((out_ids[:nqc, :kk].cpu().numpy().view(np.uint32) == got[0][:, :kk]).all())
            par["timed_path_equals_checked_path"] = same
            par["mismatches"] += 0 if same else 1
            if sel_d == 0.01:
                # the rerank: 8 queries of the batch inside their own (checked) top-1000 against oracle/ranking_oracle.py
                from oracle import synth_index as SI
                kchk = parity.KeywordLegChecker(SI.runner_lib(), kwh, n)
                uni = (np.ascontiguousarray(got[0][:8]), np.ascontiguousarray(got[2][:8]))
                rpar = kchk.verdict(0, 8, 20, universes=uni)
                par["rerank"] = rpar
                par["mismatches"] += rpar["mismatches"]
            line["parity"] = par
            del sub, al_t
        per_density[f"{sel_d:g}"] = line
        if sel_d == 0.01:
            main = line
    per_density.pop("rerank_cpu", None)
    if env.rank != 0:
        return None
    out = {
        "metric": "filtered vector search + ranking-rule rerank queries/sec (one GPU's 12.5M x 1024 bf16 shard of config 5)",
        "value": main["value"], "unit": "queries/s",
        "n_gpus": env.world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": main["ms_per_step"], "p50_latency_ms": main["p50_latency_ms"],
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "bf16 rows, f32 arithmetic",
        "data": "synthetic (rows N(0,1) seed 1234 rounded to bf16; filter seed 31; BASELINE.md C5)",
        "config": {"workload": f"C5 shard: {n} docs x {d}-d {storage}, candidate filter resident in HBM at 10 % / 1 % / 0.1 % (the line's own "
                               f"numbers: 1 %), exact cosine top-{k}, then the keyword ranking of every query INSIDE its top-{k} with the full "
                               f"default criteria (msi_keyword_search_ranked, candidate set = universe; coherent text corpus of the same {n} "
                               f"documents, 1-3 word queries with typos), top-20 returned; {B} queries per step, one caller per query",
                   "tiles_streamed_per_launch": main["tiles_streamed_per_launch"], "tiles_in_store": (n + 15) // 16,
                   "scan_tiles_counted_by_the_library": main["scan_tiles_counted_by_the_library"],
                   "inexact_queries_last_step": main["inexact_queries_last_step"],
                   "row_granular_note": "round 6: a filtered sweep gathers the allowed rows at 64-byte-sector granularity "
                                        "(bytes_streamed_over_allowed_row_bytes ~1 at every density of this line; rounds 1-5 streamed "
                                        "every 16-row tile that held an allowed row: 8 / 15 / 16 x at 10 / 1 / 0.1 %)"},
        "roofline": main["roofline"],
        "densities": per_density,
    }
    kwl.rb_destroy(kwh)
    if "cpu_baseline" in main:
        out["cpu_baseline"] = main["cpu_baseline"]
    if "parity" in main:
        par = dict(main["parity"])
        par["mismatches"] = sum(v["parity"]["mismatches"] for v in per_density.values() if "parity" in v)
        par["checked_queries"] = sum(v["parity"]["checked_queries"] for v in per_density.values() if "parity" in v)
        par["densities_checked"] = [k_ for k_, v in per_density.items() if "parity" in v]
        out["parity"] = par
    return out


# --------------------------------------------------------------------------------------------------- C1

def run_c1(args, env):
    """Plumbing (SURVEY §8 d, C1): keyword-only search on a 32 k-document synthetic corpus.  The reference's
    movies.json is a remote dataset; the corpus is restated synthetically (Zipf(1.07) over a 60 k-word vocabulary,
    title 3-6 / overview 20-60 words, seed 42) with the workload's four queries shapes ("" placeholder, two words,
    a stop-word-like frequent word, a one-letter prefix) + sampled 1-3 word queries with 0-2 edits; limit 100."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from c1_corpus import run as run_corpus          # tests/c1_corpus.py (test infrastructure: toy indexer + oracle)
    return run_corpus(args, env)


# ------------------------------------------------------------------------------------- the line the driver parses

SHORT_LINE_MAX = 4096        # bytes; the driver keeps the last 8 KB of stdout and parses its last line


def _pick(d, keys):
    return {k_: d[k_] for k_ in keys if isinstance(d, dict) and k_ in d and d[k_] is not None}


def _clip(s, n):
    return s if not isinstance(s, str) or len(s) <= n else s[:n - 2] + ".."


def _roofline_short(r):
    if not isinstance(r, dict):
        return None
    o = _pick(r, ("kernel", "bound", "achieved", "peak", "unit", "frac", "traffic", "avg_launch_ms", "launches_timed",
                  "algorithmic_bytes_per_launch", "bytes_per_row"))
    if "kernel" in o:
        o["kernel"] = _clip(o["kernel"], 48)
    if "traffic" not in o:
        o["traffic"] = None
    return o


def _parity_counts(p):
    """Counts only: what was compared and how many differed."""
    if not isinstance(p, dict):
        return None
    o = _pick(p, ("checked_queries", "checked_words", "mismatches", "timed_path_equals_checked_path"))
    if isinstance(p.get("typo"), dict):
        o["typo_words"] = p["typo"].get("checked_words")
    kwp = p.get("keyword")
    if isinstance(kwp, dict):
        o["keyword_vs_oracle"] = kwp.get("checked_queries")
        o["keyword_hits"] = kwp.get("hits_compared")
        o["keyword_score_details"] = kwp.get("score_details_compared")
        if isinstance(kwp.get("command_lists_vs_direct_back_end"), dict):
            o["keyword_lists_vs_direct"] = kwp["command_lists_vs_direct_back_end"].get("checked_queries")
    if isinstance(p.get("rerank"), dict):
        o["rerank_queries"] = p["rerank"].get("checked_queries")
    return o


def _also_one(line):
    """value / roofline fraction / CPU baseline / mismatches of one side configuration."""
    if not isinstance(line, dict):
        return None
    if "error" in line:
        return {"error": _clip(line["error"], 120)}
    o = _pick(line, ("value", "unit", "ms_per_step"))
    r = line.get("roofline") or {}
    o["frac"], o["bound"] = r.get("frac"), r.get("bound")
    if r.get("traffic") is not None:
        o["traffic_over_algorithmic"] = round(r["traffic"] * 1e9 / max(1, r.get("algorithmic_bytes_per_launch", 1)), 3)
    cb = line.get("cpu_baseline") or {}
    if cb:
        o["cpu"], o["cpu_cores"] = cb.get("value"), cb.get("cores")
    pr = line.get("parity") or {}
    if pr:
        o["checked"] = pr.get("checked_queries", pr.get("checked_words"))
        o["mismatches"] = pr.get("mismatches")
    return o


def short_line(full, detail_path=None):
    """The ONE line the driver records, assembled from the full result object (which goes to `detail_path` and to an earlier
    stdout line): BASELINE.json's metric on its C4 configuration, the dominant kernel's roofline, the CPU baseline, parity
    counts and one entry per side configuration — never more than SHORT_LINE_MAX bytes (tests/test_bench_line_cpu.py)."""
    cfg = full.get("config") or {}
    out = _pick(full, ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "p50_latency_ms"))
    out["higher_is_better"] = full.get("higher_is_better", True)
    out["scaling"] = full.get("scaling", "weak")
    out["vs_baseline"] = full.get("vs_baseline")
    out["dtype"] = full.get("dtype")
    out["data"] = _clip(full.get("data", "synthetic"), 72)
    c = {"workload": _clip(cfg.get("workload", ""), 210)}
    c.update(_pick(cfg, ("queries_per_step_per_gpu", "words_per_step_per_gpu", "queries_per_hbm_sweep", "rccl_ranks_seen",
                         "keyword_callers_per_rank", "host_cpus_granted", "keyword_corpus", "keyword_stream",
                         "inexact_queries_last_step", "per_rank_values", "keyword_cap_predicted", "keyword_cap_measured", "hbm_free_gb_min")))
    if cfg.get("sharding"):
        sh = cfg["sharding"]                      # (the short line keeps what is sharded and the exchange path, not the prose between)
        if "; exchange path: " in sh:
            sh = sh.split(",")[0] + "; exchange: " + sh.split("; exchange path: ", 1)[1]
        c["sharding"] = _clip(sh, 110)
    if isinstance(c.get("keyword_stream"), str):
        c["keyword_stream"] = _clip(c["keyword_stream"], 60)
    c["step_includes"] = [_clip(s, 36) for s in cfg.get("step_includes_short", cfg.get("step_includes", []))]
    c["step_excludes"] = cfg.get("step_excludes", [])
    out["config"] = c
    out["roofline"] = _roofline_short(full.get("roofline"))
    cb = full.get("cpu_baseline")
    if isinstance(cb, dict):
        o = _pick(cb, ("value", "unit", "cores", "kind", "vector_queries_per_s", "typo_words_per_s", "keyword_queries_per_s"))
        o["sample"] = _clip(cb.get("sample", ""), 100)
        out["cpu_baseline"] = o
    if full.get("parity") is not None:
        out["parity"] = _parity_counts(full["parity"])
    legs = full.get("legs") or {}
    if isinstance(legs.get("f32_sweep_level"), dict):
        out.setdefault("roofline", {})
        if isinstance(out["roofline"], dict):
            out["roofline"]["f32_sweep_frac"] = legs["f32_sweep_level"].get("frac_of_8_TBps")
    lg = _pick(legs, ("vector_only_queries_per_s", "keyword_only_queries_per_s", "keyword_only_host_cpus_used",
                      "keyword_cold_posting_cache_queries_per_s", "keyword_cycled_queries_per_s", "keyword_lists_per_query", "keyword_algorithmic_bytes_per_query",
                      "keyword_host_cpu_ms_per_query"))
    if isinstance(lg.get("keyword_algorithmic_bytes_per_query"), dict):
        ab = lg.pop("keyword_algorithmic_bytes_per_query")
        lg["keyword_mb_per_query"] = {"stored_postings_read": round((ab.get("stored_posting_bytes_read") or 0) / 1e6, 1),
                                      "set_operands": round((ab.get("set_operand_bytes_of_the_command_lists") or 0) / 1e6, 1),
                                      "ratio": ab.get("operand_bytes_over_posting_bytes")}
    if isinstance(legs.get("step_parts_ms"), dict):
        sp = legs["step_parts_ms"]
        lg["step_parts_ms"] = [sp.get("vector_leg"), sp.get("typo_lookup_enqueue_and_keyword_leg"), sp.get("sync_d2h_merge_exchange")]
    if isinstance(legs.get("keyword_posting_cache"), dict):
        lg["keyword_posting_cache_hit_rate"] = legs["keyword_posting_cache"].get("hit_rate")
    if isinstance(legs.get("hybrid_legs_side_by_side"), dict):
        sbs = legs["hybrid_legs_side_by_side"]
        lg["legs_side_by_side"] = {"queries_per_s": sbs.get("queries_per_s"), "ms_per_step": sbs.get("ms_per_step"),
                                   "sweep_split": sbs.get("sweep_split"),
                                   "sweep_frac": sbs.get("sweep_frac_of_8_TBps_beside_the_keyword_rounds")}
    if isinstance(legs.get("keyword_with_features"), dict):
        kf = legs["keyword_with_features"]
        lg["keyword_with_features_queries_per_s"] = kf.get("queries_per_s")
        lg["keyword_with_features_callbacks_share"] = (kf.get("host_cpu_us_per_query") or {}).get("index_callbacks_share")
        if isinstance(kf.get("parity"), dict):
            lg["keyword_with_features_parity"] = [kf["parity"].get("checked_queries"), kf["parity"].get("mismatches")]
    if isinstance(legs.get("keyword_postings_staged_at_index_open"), dict):
        lg["keyword_postings_staged_gb"] = round(legs["keyword_postings_staged_at_index_open"].get("stored_bytes_in_hbm", 0) / 1e9, 2)
    if lg:
        out["legs"] = lg
    if isinstance(full.get("latency"), dict):
        out["latency_ms"] = _pick(full["latency"], ("vector_p50_ms_b1", "keyword_p50_ms_1_caller", "hybrid_p50_ms_1_inflight",
                                                     "keyword_p50_ms_at_load", "hybrid_p50_ms_at_load"))
    if isinstance(full.get("rows_sharded"), dict):
        out["rows_sharded"] = _pick(full["rows_sharded"], ("queries_per_s", "ms_per_step", "scaling", "error"))
    also = full.get("also")
    if isinstance(also, dict):
        a = {}
        for name in ("c2", "c3"):
            if name in also:
                a[name] = _also_one(also[name])
        if isinstance(also.get("c5"), dict):
            c5 = also["c5"]
            if "error" in c5:
                a["c5"] = {"error": _clip(c5["error"], 120)}
            else:
                per = {}
                for dens, line in (c5.get("densities") or {}).items():
                    o = _also_one(line)
                    for key in ("unit", "bound", "cpu_cores"):
                        o.pop(key, None)
                    if isinstance(line.get("cpu_baseline"), dict):
                        o["cpu"] = line["cpu_baseline"].get("value")
                    per[dens] = o
                a["c5"] = {"unit": c5.get("unit"), "k": (c5.get("parity") or {}).get("k"), "by_filter_density": per,
                           "rerank_checked": ((c5.get("parity") or {}).get("rerank") or {}).get("checked_queries"),
                           "mismatches": (c5.get("parity") or {}).get("mismatches")}
        if isinstance(also.get("clustered"), dict):
            a["clustered"] = {tag: (_pick(line, ("value", "error")) | {"frac": (line.get("roofline") or {}).get("frac"),
                                                                        "mismatches": (line.get("parity") or {}).get("mismatches")})
                              for tag, line in also["clustered"].items() if isinstance(line, dict)}
        out["also"] = a
    out["seconds"] = full.get("seconds")
    if detail_path:
        out["detail"] = detail_path
    # never longer than the driver reads: drop the optional groups, least important first
    for drop in (None, "latency_ms", "rows_sharded", "data", "legs", "also"):
        if drop is not None:
            out.pop(drop, None)
        line = json.dumps(out, separators=(",", ":"))
        if len(line.encode()) < SHORT_LINE_MAX:
            return line
    out["config"] = {"workload": _clip(cfg.get("workload", ""), 200)}
    return json.dumps(out, separators=(",", ":"))


def emit(full, args, final):
    """Detail first (a file + a stdout line that does not parse as JSON on its own), the short line LAST."""
    detail_path = None
    try:
        d = os.environ.get("MSI_BENCH_DETAIL_DIR") or os.path.join(ROOT, "gpurun_out")
        os.makedirs(d, exist_ok=True)
        detail_path = os.path.join(d, f"bench_detail_{args.config}_n{args.gpus}.json")
        with open(detail_path, "w") as f:
            json.dump(full, f)
        detail_path = os.path.relpath(detail_path, ROOT)
    except OSError:
        detail_path = None
    if final:
        print("BENCH_DETAIL " + json.dumps(full), flush=True)
    print(short_line(full, detail_path), flush=True)


def main():
    t_start = time.time()
    args = parse_args()
    phase("start-up (import torch, context)")
    env = Env(args)
    out = {"c1": run_c1, "c2": run_c2, "c3": run_c3, "c4": run_c4, "c5": run_c5}[args.config](args, env)
    want_also = args.config == "c4" and out is not None and env.check and not args.no_also
    if env.rank == 0 and out is not None and want_also:
        out["seconds"] = round(time.time() - t_start, 1)
        emit(out, args, final=False)       # the headline is on stdout before the side configurations start
    if want_also:
        import gc
        gc.collect()
        env.torch.cuda.empty_cache()
        out["also"] = also_configs(args, env)
    env.finish()
    if env.rank == 0 and out is not None:
        out["seconds"] = round(time.time() - t_start, 1)
        out["phase_seconds"] = phase_seconds()
        emit(out, args, final=True)


if __name__ == "__main__":
    main()
