#!/usr/bin/env python3
"""HBM traffic per kernel from two rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE cannot share a pass: TCC counter
budget, /opt/skills/guides/MI355X_MICROARCH.md) of
    rocprofv3 --kernel-trace --pmc FETCH_SIZE -- python bench.py --steps 1 --warmup 0 --inflight 1 --no-cpu-baseline
    rocprofv3 --kernel-trace --pmc WRITE_SIZE -- python bench.py --steps 1 --warmup 0 --inflight 1 --no-cpu-baseline
usage: summarize_pmc.py fetch_results.db write_results.db bases_per_launch [source_sha] > rNN_hbm_traffic.json
gfx950 correction (same guide, HBM section): FETCH_SIZE reports half the bytes of wide coalesced reads -> doubled;
WRITE_SIZE is taken as reported.  Values are per LAUNCH (the last dispatch of each kernel = the timed step)."""
import json
import sqlite3
import sys


def per_kernel(db, counter):
    c = sqlite3.connect(db)
    out = {}
    for name, val, disp in c.execute("select kernel_name, value, dispatch_id from counters_collection where counter_name=? order by dispatch_id", (counter,)):
        short = name.split("(")[0].replace("void ", "")
        out.setdefault(short, []).append(val * 1024.0)  # the counters are in KiB
    return out


fetch, write, bases = per_kernel(sys.argv[1], "FETCH_SIZE"), per_kernel(sys.argv[2], "WRITE_SIZE"), float(sys.argv[3])
res = {"command": "rocprofv3 --kernel-trace --pmc FETCH_SIZE|WRITE_SIZE (separate passes) -- python bench.py --steps 1 --warmup 0 --inflight 1 --no-cpu-baseline",
       "workload_bp_per_launch": bases,
       "source_sha": sys.argv[4] if len(sys.argv) > 4 else None,
       "note": "gfx950: FETCH_SIZE x2 (counts 128-B requests as 64 B); per launch = last dispatch of the kernel in the run (bench.py decodes each batch once untimed before the timed step)",
       "kernels": {}}
for k in sorted(set(fetch) | set(write)):
    f = fetch.get(k, [0.0])[-1]
    w = write.get(k, [0.0])[-1]
    res["kernels"][k] = {"FETCH_SIZE_bytes_raw": f, "FETCH_SIZE_bytes_corrected_x2": 2 * f, "WRITE_SIZE_bytes": w,
                         "traffic_bytes_per_launch": 2 * f + w, "traffic_bytes_per_bp": (2 * f + w) / bases,
                         "dispatches_in_run": len(fetch.get(k, []))}
# the trellis runs in passes (kTrellis<BLK, 0>: every segment, <BLK, 1>: fix-ups, <BLK, 2/3>: continuations): their sum
tr = [v for k, v in res["kernels"].items() if k.startswith("kTrellis")]
res["whole_step_traffic_bytes"] = sum(v["traffic_bytes_per_launch"] for k, v in res["kernels"].items() if k.startswith("k"))
if tr:
    res["kTrellis"] = {"traffic_bytes_per_launch": sum(v["traffic_bytes_per_launch"] for v in tr),
                       "traffic_bytes_per_bp": sum(v["traffic_bytes_per_bp"] for v in tr),
                       "passes": {k: v["traffic_bytes_per_launch"] for k, v in res["kernels"].items() if k.startswith("kTrellis")}}
for k, v in res["kernels"].items():
    if k.startswith("kCand"):
        res["kCand"] = v
json.dump(res, sys.stdout, indent=1)
print()
