#!/usr/bin/env python
"""Summarise a rocprofv3 --pmc run (…_counter_collection.csv) into per-kernel HBM
traffic, corrected as /opt/skills/guides/MI355X_MICROARCH.md §HBM prescribes for
gfx950: FETCH_SIZE counts 64-B units of 128-B requests for wide (16 B/lane)
coalesced reads, i.e. it reports exactly half of the bytes — double it; units are KB.

    python tools/summarize_pmc.py gpurun_out/pmc_fetch/r1_counter_collection.csv > profiles/r1_pmc_fetch.json
"""
import csv
import json
import sys
from collections import defaultdict


def main(path):
    per = defaultdict(lambda: defaultdict(list))
    for r in csv.DictReader(open(path)):
        name = r["Kernel_Name"]
        if "msi" not in name and "anonymous namespace" not in name:
            continue
        short = name.split("(")[0].replace("void ", "").replace("(anonymous namespace)::", "")
        if "anonymous namespace" in name and "<" in name:
            short = name.split("((anonymous")[0].replace("void ", "").replace("(anonymous namespace)::", "")
        per[short][r["Counter_Name"]].append(float(r["Counter_Value"]))
    out = {}
    for k, ctrs in per.items():
        e = {}
        for c, v in ctrs.items():
            e[c] = {"launches": len(v), "avg": sum(v) / len(v), "max": max(v)}
            if c == "FETCH_SIZE":
                e["hbm_read_bytes_per_launch_max"] = max(v) * 1024 * 2
                e["hbm_read_bytes_per_launch_avg"] = sum(v) / len(v) * 1024 * 2
            if c == "WRITE_SIZE":
                e["hbm_write_bytes_per_launch_avg_uncalibrated"] = sum(v) / len(v) * 1024
        out[k] = e
    json.dump({"source": path, "correction": "FETCH_SIZE[KB] * 1024 * 2 (gfx950 wide-load under-count, microarch guide §HBM)",
               "kernels": out}, sys.stdout, indent=1)


if __name__ == "__main__":
    main(sys.argv[1])
