#!/bin/bash
# On the GPU box (through gpurun): per-kernel time of the S = 71 decode (--species=human --UTR=on, the dense kernels of device/dense.h) on
# the shape of bench.py's `utr` leg -> gpurun_out/<tag>_kernel_stats.txt
#   usage: profiles/run_profile_utr.sh <tag> [contigs [length]]
set -e
TAG=$1; shift
ROOT=$(cd "$(dirname "$0")/.." && pwd)
SRC_SHA=$(python "$ROOT/profiles/source_sha.py")
mkdir -p "$ROOT/gpurun_out"
export TMPDIR=/tmp
OUT=/tmp/prof_$TAG
rm -rf "$OUT"
cd /tmp
NC=${1:-256}; LEN=${2:-160000}
rocprofv3 --kernel-trace --stats -d "$OUT" -o t1 -- python "$ROOT/profiles/dense_stages.py" human $NC $LEN > "$ROOT/gpurun_out/${TAG}_run.txt" 2>&1 || true
DB=$(find "$OUT" -name '*results.db' | head -1)
python "$ROOT/profiles/summarize_rocpd.py" "$DB" > "$ROOT/gpurun_out/${TAG}_kernel_stats.txt"
echo "# command: rocprofv3 --kernel-trace --stats -- python profiles/dense_stages.py human $NC $LEN   (two decodes of one resident batch)" >> "$ROOT/gpurun_out/${TAG}_kernel_stats.txt"
echo "# source_sha: $SRC_SHA" >> "$ROOT/gpurun_out/${TAG}_kernel_stats.txt"
tail -2 "$ROOT/gpurun_out/${TAG}_run.txt" >> "$ROOT/gpurun_out/${TAG}_kernel_stats.txt"
cat "$ROOT/gpurun_out/${TAG}_kernel_stats.txt"
