#!/bin/bash
# On the GPU box (through gpurun): HBM bytes per kernel from two separate rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE do
# not fit one pass; --pmc is combined with --kernel-trace only, as /opt/skills/guides/MI355X_MICROARCH.md prescribes)
#   usage: profiles/run_pmc.sh <tag> [bench.py arguments...]   -> gpurun_out/<tag>_hbm_traffic.json
set -e
TAG=$1; shift
ROOT=$(cd "$(dirname "$0")/.." && pwd)
# hash of the kernel / host sources the profile was taken with: bench.py only quotes a profile whose hash is that of the tree it runs in
SRC_SHA=$(python "$ROOT/profiles/source_sha.py")
mkdir -p "$ROOT/gpurun_out"
export TMPDIR=/tmp
cd /tmp
for C in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_$C
  rocprofv3 --kernel-trace --pmc $C -d /tmp/pmc_$C -o t1 -- python "$ROOT/bench.py" --no-cpu-baseline --no-e2e --no-product --no-utr --steps 1 --warmup 0 --inflight 1 "$@" > /tmp/pmc_$C.out 2> /tmp/pmc_$C.err || true
done
F=$(find /tmp/pmc_FETCH_SIZE -name '*results.db' | head -1)
W=$(find /tmp/pmc_WRITE_SIZE -name '*results.db' | head -1)
python "$ROOT/profiles/summarize_pmc.py" "$F" "$W" 100000000 "$SRC_SHA" > "$ROOT/gpurun_out/${TAG}_hbm_traffic.json"
python - "$ROOT/gpurun_out/${TAG}_hbm_traffic.json" <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
tot = 0
for k, v in sorted(d["kernels"].items(), key=lambda kv: -kv[1]["traffic_bytes_per_launch"]):
    if v["traffic_bytes_per_launch"] > 1e8:
        print("%-40s %8.1f GB  (%d dispatches)" % (k[:40], v["traffic_bytes_per_launch"] / 1e9, v["dispatches_in_run"]))
        tot += v["traffic_bytes_per_launch"]
print("sum of last dispatches: %.1f GB" % (tot / 1e9))
PY
# ---- the same for the S = 71 decode (dense kernels; the shape of bench.py's `utr` leg) -> gpurun_out/<tag>_utr_hbm_traffic.json
for C in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmcu_$C
  rocprofv3 --kernel-trace --pmc $C -d /tmp/pmcu_$C -o t1 -- python "$ROOT/profiles/dense_stages.py" human 256 160000 > /tmp/pmcu_$C.out 2> /tmp/pmcu_$C.err || true
done
F=$(find /tmp/pmcu_FETCH_SIZE -name '*results.db' | head -1)
W=$(find /tmp/pmcu_WRITE_SIZE -name '*results.db' | head -1)
python "$ROOT/profiles/summarize_pmc.py" "$F" "$W" 40960000 "$SRC_SHA" > "$ROOT/gpurun_out/${TAG}_utr_hbm_traffic.json"
# ---- L2 hit rates of both workloads (one more pass each: the TCC block has four counter slots) -> gpurun_out/<tag>_l2.txt
rm -rf /tmp/pmc_l2a /tmp/pmc_l2b
rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum -d /tmp/pmc_l2a -o t1 -- python "$ROOT/bench.py" --no-cpu-baseline --no-e2e --no-product --no-utr --steps 1 --warmup 0 --inflight 1 > /tmp/pmc_l2a.out 2> /tmp/pmc_l2a.err || true
rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum -d /tmp/pmc_l2b -o t1 -- python "$ROOT/profiles/dense_stages.py" human 256 160000 > /tmp/pmc_l2b.out 2> /tmp/pmc_l2b.err || true
python - "$(find /tmp/pmc_l2a -name '*results.db' | head -1)" "$(find /tmp/pmc_l2b -name '*results.db' | head -1)" "$SRC_SHA" > "$ROOT/gpurun_out/${TAG}_l2.txt" <<'PY'
import sqlite3, sys
print("# L2 (TCC) hit rate per kernel, last dispatch: rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum (47-state bench step, then the S = 71 decode)")
for db in sys.argv[1:3]:
    c = sqlite3.connect(db)
    v = {}
    for name, cn, val, disp in c.execute("select kernel_name, counter_name, value, dispatch_id from counters_collection order by dispatch_id"):
        k = name.split("(")[0].replace("void ", "")
        v.setdefault(k, {}).setdefault(cn, {})
        v[k][cn][disp] = v[k][cn].get(disp, 0.0) + val
    for k in sorted(v, key=lambda k: -sum(max(d.values()) for d in v[k].values())):
        h = v[k].get("TCC_HIT_sum", {0: 0.0}); m = v[k].get("TCC_MISS_sum", {0: 0.0})
        hh, mm = h[max(h)], m[max(m)]
        if hh + mm > 1e6:
            print("%-28s hits %12.4g misses %12.4g hit rate %5.1f%%" % (k[:28], hh, mm, 100 * hh / (hh + mm)))
    print()
print("# source_sha: %s" % sys.argv[3])
PY
cat "$ROOT/gpurun_out/${TAG}_l2.txt"

