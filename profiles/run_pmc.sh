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
