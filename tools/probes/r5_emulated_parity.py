"""Round 5, after the last GPU minute: the keyword leg on the coherent corpus at a size of one's choice through the CPU-emulated
kernels (tests/emu), every search checked against the ranking oracle on the same stored bytes.
    python tools/probes/r5_emulated_parity.py <docs> <vocabulary> <queries> [seed]"""
import ctypes as C, os, sys, time
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests/emu")
import run_emulated as E
from meilisearch_amd import _lib
_lib._LIB = E.EmulatedLib(E.build())
os.environ["MSI_RUNNER_SO"] = E.build_runner()
import meilisearch_amd as ma
from oracle import parity, synth_index as SI
n_docs, n_words, n_queries, limit = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), 20
ctx = ma.Context(0)
lib = SI.runner_lib()
lib.rb_attach.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint64]
h = lib.rb_create_corpus(n_docs, n_words, 42)
assert lib.rb_attach(h, ctx.handle, 8, 1024, 2048) == 0
lib.rb_prepare_queries(h, n_queries, 3, int(sys.argv[4]) if len(sys.argv) > 4 else 4242)
chk = parity.KeywordLegChecker(lib, h, n_docs)
t = time.time()
cold = chk.run_product(0, n_queries, limit)
print("product", round(time.time() - t, 1), "s", flush=True)
v = chk.verdict(0, n_queries, limit, product=cold)
print(v)
warm = chk.run_product(0, n_queries, limit)
assert all((a == b).all() for a, b in zip(cold, warm))
cst, lst = (C.c_uint64 * 3)(), (C.c_uint64 * 2)()
_lib.lib().msi_search_compaction_stats(cst); _lib.lib().msi_search_late_compaction_stats(lst)
print("searches", cst[0], "compacted", cst[1], "late", lst[0])
