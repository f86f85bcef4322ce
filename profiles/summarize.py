#!/usr/bin/env python3
"""Condenses the rocprofv3 CSVs written by profiles/collect.sh into one JSON document.

Per kernel: launches and average duration from the --kernel-trace --stats pass; HBM bytes per launch from the
FETCH_SIZE / WRITE_SIZE passes (rocprofv3 reports KiB; FETCH_SIZE is doubled as MI355X_MICROARCH.md prescribes for
gfx950, WRITE_SIZE is taken as is); SQ counters per launch from the SQ pass.
"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict


def short(name):
    name = name.split("(")[0]
    return name.replace("void ", "").strip()


def counters(d):
    acc = defaultdict(lambda: defaultdict(list))
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        per_dispatch = defaultdict(float)
        names = {}
        with open(f, newline="") as fh:
            for r in csv.DictReader(fh):
                key = (r["Dispatch_Id"], r["Counter_Name"])
                per_dispatch[key] += float(r["Counter_Value"])   # rows are per XCD / instance: sum them
                names[r["Dispatch_Id"]] = short(r["Kernel_Name"])
        for (disp, cn), v in per_dispatch.items():
            acc[names[disp]][cn].append(v)
    return {k: {cn: sum(v) / len(v) for cn, v in cs.items()} for k, cs in acc.items()}


def main():
    d = sys.argv[1]
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench                                    # kernel_source_sha(): what bench.py checks before quoting the counters
    out = {"source": d, "kernel_source_sha": bench.kernel_source_sha(),
           "clips_per_gpu": int(sys.argv[2]) if len(sys.argv) > 2 else 10000,
           "clip_seconds": float(sys.argv[3]) if len(sys.argv) > 3 else 10.0,
           "fetch_correction": "FETCH_SIZE x2 (MI355X_MICROARCH.md: gfx950 tallies 128-B requests of wide coalesced reads at "
                               "64 B); calibrated for 16-B-per-lane streams only - kernels that read one dword per lane "
                               "(ll_prepare, the chain kernels' 4- and 8-byte PCM loads) may be over-counted by up to 2x",
           "kernels": {}}
    for f in glob.glob(os.path.join(d, "stats", "**", "*kernel_stats.csv"), recursive=True):
        with open(f, newline="") as fh:
            for r in csv.DictReader(fh):
                out["kernels"][short(r["Name"])] = {"launches": int(r["Calls"]), "avg_ms": float(r["AverageNs"]) / 1e6,
                                                    "min_ms": float(r["MinNs"]) / 1e6, "max_ms": float(r["MaxNs"]) / 1e6,
                                                    "percent": float(r["Percentage"])}
    fetch, write, sq = counters(os.path.join(d, "fetch")), counters(os.path.join(d, "write")), counters(os.path.join(d, "sq"))
    for k, v in counters(os.path.join(d, "sq2")).items():
        sq.setdefault(k, {}).update(v)
    for k in set(fetch) | set(write) | set(sq):
        e = out["kernels"].setdefault(k, {})
        if k in fetch and "FETCH_SIZE" in fetch[k]:
            e["hbm_read_bytes_per_launch"] = fetch[k]["FETCH_SIZE"] * 1024 * 2   # KiB, gfx950 x2 correction
            e["FETCH_SIZE_raw_KiB"] = fetch[k]["FETCH_SIZE"]
        if k in write and "WRITE_SIZE" in write[k]:
            e["hbm_write_bytes_per_launch"] = write[k]["WRITE_SIZE"] * 1024
        if k in sq:
            e["sq_per_launch"] = sq[k]
    json.dump(out, sys.stdout, indent=1, sort_keys=True)
    print()


if __name__ == "__main__":
    main()
