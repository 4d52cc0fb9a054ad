"""Keyword leg on the coherent corpus, by universe class: throughput of each class ALONE (160 callers) and its latency with one
caller.  Tells which searches the device time of the leg goes to."""
import ctypes as C, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import meilisearch_amd as ma
n = int(os.environ.get("N_DOCS", 10_000_000)); k = 20; NQ = 3072; callers = int(os.environ.get("CALLERS", 160))
ctx = ma.Context(0)
L = C.CDLL(os.path.join(ROOT, "tools", "bin", "libmsi_rankedbench.so"))
L.rb_create_corpus.restype = C.c_void_p; L.rb_create_corpus.argtypes = [C.c_uint64, C.c_uint32, C.c_uint64]
L.rb_attach.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint64]
L.rb_prepare_queries.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint64]
L.rb_run_detailed.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32] + [C.c_void_p] * 6
L.rb_run.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p]
L.rb_permute_queries.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32]
L.rb_last_latencies.restype = C.c_uint32; L.rb_last_latencies.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32]
L.rb_destroy.argtypes = [C.c_void_p]
h = L.rb_create_corpus(n, 2_000_000, 42)
assert L.rb_attach(h, ctx.handle, callers, 512, 8192) == 0
L.rb_prepare_queries(h, NQ, 3, 4242)
ids = np.zeros((NQ, k), np.uint32); cnt = np.zeros(NQ, np.uint32); sc = np.zeros((NQ, k), np.float64); cand = np.zeros(NQ, np.uint64)
assert L.rb_run_detailed(h, 0, NQ, k, ids.ctypes.data, cnt.ctypes.data, sc.ctypes.data, None, None, cand.ctypes.data) == 0
order = np.argsort(cand, kind="stable").astype(np.uint32)
assert L.rb_permute_queries(h, order.ctypes.data, NQ) == 0
cs = cand[order].astype(np.float64) / n
edges = [0, 0.001, 0.01, 0.125, 0.5, 1.01]
names = ["<=0.1%", "0.1-1%", "1-12.5%", "12.5-50%", ">50%"]
def mixed(tag, lo=0, n=None):
    n = NQ if n is None else n
    t0 = time.perf_counter(); assert L.rb_run(h, lo, n, k, ids.ctypes.data, cnt.ctypes.data, sc.ctypes.data) == 0
    print(f"{tag}: {n / (time.perf_counter() - t0):.0f} q/s", flush=True)
if os.environ.get("CHW_SWEEP"):
    for rep in range(int(os.environ.get("REPS", 2))):
        for v in os.environ["CHW_SWEEP"].split(","):
            os.environ["MSI_VM_COMPACT_CHW"] = v
            mixed(f"compact chunk width {v:>7s}")
    sys.exit(0)
if os.environ.get("BURST"):
    # round 5: the burst that collapsed the leg in round 4 — the searches whose universe is 1-12.5 % of the index (48-153-chunk
    # compact lists), in sorted order, all callers alike; FUSE max chunks per list from BURST="24,160"
    lo, hi = int(np.searchsorted(cs, 0.01, "right")), int(np.searchsorted(cs, 0.125, "right"))
    for fm in os.environ["BURST"].split(","):
        os.environ["MSI_VM_FUSE_MAX_CHUNKS"] = fm
        for rep in range(2): mixed(f"burst of {hi - lo} alike searches (universe 1-12.5 %), fuse_max_chunks {fm:>4s}", lo, hi - lo)
    L.rb_destroy(h)
    sys.exit(0)
if os.environ.get("FUSE_SWEEP"):
    # original order first (what bench.py runs), then sorted by universe (bursts of alike searches)
    inv = np.argsort(order, kind="stable").astype(np.uint32)
    for srt in (0, 1):
        if srt == 0: assert L.rb_permute_queries(h, inv.ctypes.data, NQ) == 0
        else: assert L.rb_permute_queries(h, order.ctypes.data, NQ) == 0
        for fm in os.environ["FUSE_SWEEP"].split(","):
            os.environ["MSI_VM_FUSE_MAX_CHUNKS"] = fm
            for rep in range(2): mixed(f"{'sorted  ' if srt else 'unsorted'} fuse_max_chunks {fm:>4s}")
    sys.exit(0)
mixed("all classes mixed")
tot = 0.0
only = os.environ.get("CLASSES")
for i, name in enumerate(names):
    if only and str(i) not in only.split(","): continue
    lo, hi = int(np.searchsorted(cs, edges[i], "right" if i else "left")), int(np.searchsorted(cs, edges[i + 1], "right"))
    if hi <= lo: continue
    m = hi - lo
    reps = int(os.environ.get("REPS", max(1, 1500 // m)))
    m = min(m, int(os.environ.get("MAX_PER_CLASS", m)))
    c0 = os.times(); t0 = time.perf_counter()
    assert L.rb_run(h, lo, m, k, ids.ctypes.data, cnt.ctypes.data, sc.ctypes.data) == 0
    c1 = os.times()
    print(f"{name}: first pass of {m}: {time.perf_counter() - t0:.2f} s wall, {c1[0] + c1[1] - c0[0] - c0[1]:.2f} s CPU", flush=True)
    t0 = time.perf_counter()
    for _ in range(reps): assert L.rb_run(h, lo, m, k, ids.ctypes.data, cnt.ctypes.data, sc.ctypes.data) == 0
    qps = reps * m / (time.perf_counter() - t0)
    one = []
    for j in range(lo, min(hi, lo + 24)):
        t1 = time.perf_counter(); L.rb_run(h, j, 1, k, ids.ctypes.data, cnt.ctypes.data, sc.ctypes.data); one.append((time.perf_counter() - t1) * 1e3)
    tot += m / qps
    print(f"{name:10s} {m:5d} queries ({m / NQ:.2f})  alone {qps:8.0f} q/s  -> {m / qps * 1e3 / NQ * 768:6.1f} ms of a 768-query step   one caller p50 {np.median(one):.2f} ms", flush=True)
print(f"sum of classes run alone: {NQ / tot:.0f} q/s")
st = (C.c_uint64 * 3)(); lt = (C.c_uint64 * 2)()
ma._lib.lib().msi_search_compaction_stats(st); ma._lib.lib().msi_search_late_compaction_stats(lt)
print(f"searches {st[0]}, compacted {st[1]} (mean universe {st[2] / max(1, st[1]):.0f}); sub-trees moved into their bucket's space {lt[0]} (mean bucket {lt[1] / max(1, lt[0]):.0f})")
L.rb_destroy(h)
