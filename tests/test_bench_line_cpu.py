"""bench.py's last stdout line is what the driver records: it must stay short enough to survive the driver's 8 KB tail
(round 4's 26 KB line lost its head: BENCH_r04.json `parsed: null`) and carry the fields BASELINE.json's metric needs."""
import copy
import importlib.util
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    spec = importlib.util.spec_from_file_location("msi_bench_module", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    argv, sys.argv = sys.argv, ["bench.py"]
    try:
        spec.loader.exec_module(mod)
    finally:
        sys.argv = argv
    return mod


def _canned():
    # a complete result object of the default command (round 4's own run, 26 KB with its `also` tree)
    with open(os.path.join(ROOT, "profiles", "r4_bench_c4.json")) as f:
        return json.load(f)


REQUIRED = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "p50_latency_ms", "higher_is_better",
            "scaling", "vs_baseline", "dtype", "config", "roofline", "cpu_baseline", "parity")


def test_line_is_short_and_round_trips():
    b = _bench()
    full = _canned()
    assert len(json.dumps(full)) > 20000          # the canned input is the line that broke the parser
    line = b.short_line(full, "gpurun_out/bench_detail_c4_n1.json")
    assert "\n" not in line
    assert len(line.encode()) < 4096
    got = json.loads(line)
    for key in REQUIRED:
        assert key in got, key
    assert got["value"] == full["value"] and got["ms_per_step"] == full["ms_per_step"]
    assert got["config"]["workload"].startswith("C4 on one GPU per rank")
    assert got["config"]["rccl_ranks_seen"] == 1
    r = got["roofline"]
    assert r["bound"] == "hbm" and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3 and "traffic" in r
    assert got["cpu_baseline"]["kind"] == "port" and got["cpu_baseline"]["cores"] > 0
    assert got["parity"]["mismatches"] == 0 and got["parity"]["checked_queries"] == 96
    also = got["also"]
    assert also["c2"]["frac"] == full["also"]["c2"]["roofline"]["frac"] and also["c2"]["mismatches"] == 0
    assert also["c3"]["unit"] == "words/s" and also["c3"]["cpu"] > 0
    assert set(also["c5"]["by_filter_density"]) == {"0.1", "0.01", "0.001"}


def test_line_stays_short_whatever_the_legs_say():
    b = _bench()
    full = _canned()
    # a hostile input: every string ten times longer, extra densities, error texts
    def grow(o):
        if isinstance(o, dict):
            return {k: grow(v) for k, v in o.items()}
        if isinstance(o, list):
            return [grow(v) for v in o] * 3
        if isinstance(o, str):
            return o * 10
        return o
    big = grow(copy.deepcopy(full))
    big["also"]["c5"]["densities"].update({f"0.{i}": big["also"]["c5"]["densities"]["0.1"] for i in range(2, 9)})
    big["also"]["c2"] = {"error": "x" * 5000}
    line = b.short_line(big)
    assert len(line.encode()) < 4096
    got = json.loads(line)
    for key in REQUIRED:
        assert key in got, key


def test_a_line_without_side_configurations():
    b = _bench()
    full = _canned()
    for k in ("also", "legs", "latency", "keyword_roofline", "parity", "cpu_baseline"):
        full.pop(k, None)
    got = json.loads(b.short_line(full))
    assert got["value"] == full["value"] and "also" not in got and got["roofline"]["frac"] == full["roofline"]["frac"]


def test_round_5_object_fits_too():
    """The round-5 result object (int8 level, per-rank fields, C5 CPU baselines at three densities, phase and step times)."""
    b = _bench()
    with open(os.path.join(ROOT, "profiles", "r5_bench_c4.json")) as f:
        full = json.load(f)
    line = b.short_line(full, "gpurun_out/bench_detail_c4_n1.json")
    assert len(line.encode()) < 4096
    got = json.loads(line)
    for key in REQUIRED:
        assert key in got, key
    assert got["config"]["rccl_ranks_seen"] == 1 and abs(got["config"]["per_rank_values"][0] - full["value"]) < 0.1
    assert got["roofline"]["kernel"].startswith("vs_scan_i8_kernel") and got["roofline"]["traffic"] is not None
    assert set(got["also"]["c5"]["by_filter_density"]) == {"0.1", "0.01", "0.001"}
    assert all(v.get("cpu") for v in got["also"]["c5"]["by_filter_density"].values())      # the CPU port at every density
    assert "step_ms" not in got and "phase_seconds" not in got                             # detail only
