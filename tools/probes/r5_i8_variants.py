"""The int8 candidate sweep's kernel shapes side by side (MSI_VS_I8_VARIANT, msi_vs.hip launch_scan8_variant): one store,
128-query batches through msi_vs_search_device, kernel time of the main pass from HIP events (msi_vs_scan_time)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
import meilisearch_amd as ma
from meilisearch_amd import synth
n, d, k, Q = int(os.environ.get("N_ROWS", 10_000_000)), int(os.environ.get("DIM", 768)), 20, int(os.environ.get("Q", 768))
dev = torch.device("cuda", 0)
ctx = ma.Context(0)
rows = synth.device_rows(n, d, dev, seed=1234)
ids = torch.arange(n, dtype=torch.int32, device=dev)
st = ma.GpuStore(ctx, d, storage=os.environ.get("STORAGE", "f32"))   # (STORAGE=bf16: round 6, the int8 copy of a bf16 store)
st.upload_device(ids, rows)
del rows
q = synth.device_queries(Q, d, dev, seed=5678)
o_i = torch.zeros((Q, k), dtype=torch.int32, device=dev); o_d = torch.zeros((Q, k), dtype=torch.float32, device=dev)
o_c = torch.zeros(Q, dtype=torch.int32, device=dev); o_x = torch.zeros(Q, dtype=torch.int32, device=dev)
stats = st.stats()
bytes8 = ((n + 15) // 16) * stats["i8_bytes_per_tile"]
ref = None
for v in os.environ.get("VARIANTS", "0,1,2,3,4").split(","):
    os.environ["MSI_VS_I8_VARIANT"] = v
    for _ in range(2):
        st.search_device(q, k, o_i, o_d, o_c, o_x); ctx.synchronize()
    ctx.set_profiling(True); st.scan_time()
    t0 = time.perf_counter()
    for _ in range(3):
        st.search_device(q, k, o_i, o_d, o_c, o_x); ctx.synchronize()
    dt = (time.perf_counter() - t0) / 3
    cnt, ms = st.scan_time(); ctx.set_profiling(False)
    got = (o_i.cpu().numpy().copy(), o_d.cpu().numpy().copy())
    if ref is None: ref = got
    same = bool((got[0] == ref[0]).all() and (got[1].view(np.uint32) == ref[1].view(np.uint32)).all())
    print(f"variant {v}: sweep {ms / cnt:.3f} ms = {bytes8 / (ms / cnt * 1e-3) / 1e12:.2f} TB/s ({bytes8 / (ms / cnt * 1e-3) / 8e12:.3f} of peak), "
          f"{Q} queries in {dt * 1e3:.2f} ms = {Q / dt:.0f} q/s, sweeps {cnt // 3}, reruns {st.stats()['device_rerun_queries']}, same lists {same}", flush=True)
os.environ.pop("MSI_VS_I8_VARIANT")
os.environ["MSI_VS_FIRST_LEVEL"] = "f32"
for _ in range(2):
    st.search_device(q, k, o_i, o_d, o_c, o_x); ctx.synchronize()
ctx.set_profiling(True); st.scan_time()
t0 = time.perf_counter(); st.search_device(q, k, o_i, o_d, o_c, o_x); ctx.synchronize(); dt = time.perf_counter() - t0
cnt, ms = st.scan_time()
b32 = ((n + 15) // 16) * stats["bytes_per_tile"]
got = (o_i.cpu().numpy(), o_d.cpu().numpy())
print(f"f32 level: sweep {ms / cnt:.3f} ms = {b32 / (ms / cnt * 1e-3) / 8e12:.3f} of peak, {Q} queries in {dt * 1e3:.2f} ms = {Q / dt:.0f} q/s, "
      f"same lists {bool((got[0] == ref[0]).all() and (got[1].view(np.uint32) == ref[1].view(np.uint32)).all())}")
