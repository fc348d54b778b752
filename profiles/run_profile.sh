#!/bin/bash
# On the GPU box (through gpurun): per-kernel time of one bench configuration -> gpurun_out/<tag>_kernel_stats.txt
#   usage: profiles/run_profile.sh <tag> [bench.py arguments...]
# (counters are collected in their own runs, see profiles/run_pmc.sh: --pmc is never combined with other trace domains)
set -e
TAG=$1; shift
ROOT=$(cd "$(dirname "$0")/.." && pwd)
# hash of the kernel / host sources the profile was taken with: bench.py only quotes a profile whose hash is that of the tree it runs in
SRC_SHA=$(python "$ROOT/profiles/source_sha.py")
mkdir -p "$ROOT/gpurun_out"
export TMPDIR=/tmp
OUT=/tmp/prof_$TAG
rm -rf "$OUT"
cd /tmp
rocprofv3 --kernel-trace --stats -d "$OUT" -o t1 -- python "$ROOT/bench.py" --no-cpu-baseline --no-e2e --no-product --no-utr "$@" > "$ROOT/gpurun_out/${TAG}_bench.json" 2> "$ROOT/gpurun_out/${TAG}_bench.err" || true
DB=$(find "$OUT" -name '*results.db' | head -1)
python "$ROOT/profiles/summarize_rocpd.py" "$DB" > "$ROOT/gpurun_out/${TAG}_kernel_stats.txt"
echo "# command: rocprofv3 --kernel-trace --stats -- python bench.py --no-cpu-baseline --no-e2e --no-product --no-utr $*" >> "$ROOT/gpurun_out/${TAG}_kernel_stats.txt"
echo "# (the vgpr= field of the dispatch lines is rocprofv3's: HALF the hardware's count on gfx950.  Registers, spills, LDS and scratch of every kernel as the" >> "$ROOT/gpurun_out/${TAG}_kernel_stats.txt"
echo "#  code objects' ELF notes have them: profiles/${TAG%%_*}_kernel_resources.txt = python profiles/kernel_resources.py build/obj)" >> "$ROOT/gpurun_out/${TAG}_kernel_stats.txt"
echo "# source_sha: $SRC_SHA" >> "$ROOT/gpurun_out/${TAG}_kernel_stats.txt"
cat "$ROOT/gpurun_out/${TAG}_kernel_stats.txt"
tail -1 "$ROOT/gpurun_out/${TAG}_bench.json" | cut -c1-600
