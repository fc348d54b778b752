#!/bin/bash
# On the GPU box (through gpurun): every profile file of a round in one call, each step bounded, with the seconds it took.
#   usage: profiles/run_round_set.sh r06      -> gpurun_out/r06_{one_batch,two_batches,utr,forward}_*, r06_{hbm_traffic,utr_hbm_traffic}.json,
#                                                r06_{l2,sq,utr_sq}.txt, r06_bench.json  (copy what is to be judged into profiles/)
R=$1
ROOT=$(cd "$(dirname "$0")/.." && pwd)
cd "$ROOT"
step() { local t0=$SECONDS; timeout 600 "$@" > /tmp/step.out 2>&1; echo "[$((SECONDS - t0)) s, rc $?] $*"; tail -3 /tmp/step.out | cut -c1-300; }
step profiles/run_pmc.sh $R
step profiles/run_profile.sh ${R}_one_batch --steps 6 --warmup 1 --inflight 1 --no-two-batches
step profiles/run_profile.sh ${R}_two_batches --steps 6 --warmup 1 --inflight 2
step profiles/run_profile_utr.sh ${R}_utr
step profiles/run_profile_forward.sh ${R}
step profiles/run_sq.sh $R
cp "$ROOT"/gpurun_out/${R}_hbm_traffic.json "$ROOT"/gpurun_out/${R}_utr_hbm_traffic.json "$ROOT"/profiles/ 2>/dev/null   # (bench.py quotes the PMC traffic of THIS tree from profiles/)
t0=$SECONDS
timeout 900 python bench.py > gpurun_out/${R}_bench.json 2> gpurun_out/${R}_bench.err
echo "[$((SECONDS - t0)) s, rc $?] python bench.py"
cut -c1-900 gpurun_out/${R}_bench.json
