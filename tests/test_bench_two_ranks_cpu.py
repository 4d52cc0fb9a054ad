"""bench.py --gpus 2 end to end, in the CPU tier (N > 1 on RCCL has never run on hardware — one-GPU boxes — so
the first 8-GPU run must not be the first time the N > 1 line is assembled).  Two ranks under torch.distributed.run, each
with the CPU-emulated build of libmsi (tests/emu: every csrc/*.hip as plain C++) as "its GPU", gloo as the launcher's
process group and the RCCL stand-in joined through shared memory (tests/emu/rccl_emu.cpp) behind msi_group_create_rank /
msi_group_allgather — the library's own exchange path, the one bench.py takes on a GPU node.  Sizes are what the emulation
finishes in about a minute; nothing here is a measurement.  Asserted: the line parses, both ranks took part in the exchange
through libmsi (rccl_ranks_seen == 2, exchange path named), per-rank values, the keyword caps, the staged postings, and the
rows-sharded extra (strong scaling: one packed all-gather + device merge) answered like the replicated search."""
import glob
import json
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.timeout(1500)
def test_two_ranks_on_emulated_devices():
    sys.path.insert(0, os.path.join(ROOT, "tests", "emu"))
    import run_emulated as E
    E.build()
    E.build_runner()
    E.build_rccl()          # (before the ranks start: they must not race for the build)
    env = dict(os.environ, MSI_BENCH_EMULATED="1", MSI_EMU_DEVICES="2", OMP_NUM_THREADS="2", MSI_BENCH_CALLERS_PER_CPU="1",
               MSI_BENCH_DERIVE_BUDGET_S="60")
    env.pop("MSI_RUNNER_SO", None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
           "--rows", "20000", "--dim", "64", "--queries", "32", "--dict-words", "4000", "--kw-dict-words", "6000",
           "--kw-threads", "4", "--kw-slots", "256", "--kw-cache-mb", "64", "--no-cpu-baseline", "--no-also", "--no-pmc",
           "--extra-timeout", "600"]
    try:
        r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=1400)
    finally:
        for f in glob.glob("/dev/shm/msi_rccl_emu_*"):
            try:
                os.remove(f)
            except OSError:
                pass
    assert r.returncode == 0, (r.stdout[-3000:], r.stderr[-6000:])
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert lines, r.stdout[-2000:]
    line = json.loads(lines[-1])
    assert len(lines[-1]) <= 4096
    assert line["n_gpus"] == 2 and line["steps"] == 2 and line["warmup"] == 1 and line["scaling"] == "weak"
    assert line["value"] > 0 and line["ms_per_step"] > 0
    cfg = line["config"]
    assert cfg["rccl_ranks_seen"] == 2
    assert len(cfg["per_rank_values"]) == 2 and all(v > 0 for v in cfg["per_rank_values"])
    assert abs(sum(cfg["per_rank_values"]) - line["value"]) <= 0.35 * line["value"]     # (the slowest rank's clock sets `value`)
    assert cfg["keyword_cap_measured"] > 0 and cfg["keyword_cap_predicted"] > 0
    assert "msi_group_allgather" in cfg["sharding"] or "libmsi" in cfg["sharding"], cfg["sharding"]
    assert line["roofline"]["bound"] == "hbm" and "frac" in line["roofline"]      # (the emulation has no clock worth a fraction)
    if "rows_sharded" in line:
        assert "error" not in line["rows_sharded"], line["rows_sharded"]
