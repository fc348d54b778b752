#!/bin/bash
# On the GPU box (through gpurun): per-kernel time of the FORWARD pass (posterior sampling, --sample) of both kernel families:
# kForward<8> (47 states, human) and kDense<4,1> (71 states, fly at its defaults) on 32 pieces -> gpurun_out/<tag>_forward_kernel_stats.txt
#   usage: profiles/run_profile_forward.sh <tag>
set -e
TAG=$1
ROOT=$(cd "$(dirname "$0")/.." && pwd)
SRC_SHA=$(python "$ROOT/profiles/source_sha.py")
mkdir -p "$ROOT/gpurun_out"
export TMPDIR=/tmp
OUT=/tmp/prof_fwd_$TAG
rm -rf "$OUT"
cd /tmp
cat > /tmp/fwd_probe.py <<PY
import sys, time
sys.path.insert(0, "$ROOT"); sys.path.insert(0, "$ROOT/tests")
import augustus_amd as ax, bench
from helpers import config_path
for species, opts, n, ln in (("human", {}, 32, 1000000), ("fly", {}, 32, 200000)):
    m = ax.Model(config_path(), species, **opts)
    d = ax.Decoder(m, 0)
    b = ax.Batch(d, bench.synth_contigs(n, ln, 4000))
    b.decode(sync=True)
    t0 = time.perf_counter()
    b.forward()
    ax._check(ax.lib().augx_batch_sync(d._h))
    print("%s S=%d: forward pass of %d x %d bp: %.3f s" % (species, m.n_states, n, ln, time.perf_counter() - t0))
    b.close(); d.close()
PY
rocprofv3 --kernel-trace --stats -d "$OUT" -o t1 -- python /tmp/fwd_probe.py > "$ROOT/gpurun_out/${TAG}_forward_run.txt" 2>&1 || true
DB=$(find "$OUT" -name '*results.db' | head -1)
python "$ROOT/profiles/summarize_rocpd.py" "$DB" > "$ROOT/gpurun_out/${TAG}_forward_kernel_stats.txt"
echo "# command: rocprofv3 --kernel-trace --stats -- (Viterbi decode + forward pass: human 32 x 1 Mbp, then fly defaults 32 x 200 kb)" >> "$ROOT/gpurun_out/${TAG}_forward_kernel_stats.txt"
echo "# source_sha: $SRC_SHA" >> "$ROOT/gpurun_out/${TAG}_forward_kernel_stats.txt"
grep "forward pass" "$ROOT/gpurun_out/${TAG}_forward_run.txt" | sed 's/^/# /' >> "$ROOT/gpurun_out/${TAG}_forward_kernel_stats.txt"
cat "$ROOT/gpurun_out/${TAG}_forward_kernel_stats.txt"
