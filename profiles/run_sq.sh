#!/bin/bash
# On the GPU box (through gpurun): SQ counters of every kernel of (a) one bench step of the 47-state decode and (b) the S = 71 decode,
# one rocprofv3 PMC pass each (--pmc with --kernel-trace only) -> gpurun_out/<tag>_sq.txt, <tag>_utr_sq.txt
#   usage: profiles/run_sq.sh <tag>
set -e
TAG=$1
ROOT=$(cd "$(dirname "$0")/.." && pwd)
SRC_SHA=$(python "$ROOT/profiles/source_sha.py")
mkdir -p "$ROOT/gpurun_out"
export TMPDIR=/tmp
cd /tmp
CTRS="SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS"
rm -rf /tmp/sq_a /tmp/sq_b
rocprofv3 --kernel-trace --pmc $CTRS -d /tmp/sq_a -o t1 -- python "$ROOT/bench.py" --no-cpu-baseline --no-e2e --no-product --no-utr --steps 1 --warmup 0 --inflight 1 > /tmp/sq_a.out 2> /tmp/sq_a.err || true
python "$ROOT/profiles/summarize_sq.py" "$(find /tmp/sq_a -name '*results.db' | head -1)" "$SRC_SHA" > "$ROOT/gpurun_out/${TAG}_sq.txt"
rocprofv3 --kernel-trace --pmc $CTRS -d /tmp/sq_b -o t1 -- python "$ROOT/profiles/dense_stages.py" human 256 160000 > /tmp/sq_b.out 2> /tmp/sq_b.err || true
python "$ROOT/profiles/summarize_sq.py" "$(find /tmp/sq_b -name '*results.db' | head -1)" "$SRC_SHA" > "$ROOT/gpurun_out/${TAG}_utr_sq.txt"
cat "$ROOT/gpurun_out/${TAG}_sq.txt" "$ROOT/gpurun_out/${TAG}_utr_sq.txt"
