# developer aid (GPU box): the laps of the executable with posterior sampling, 32 x 1 Mbp of uniform-random DNA
set -e
cd $GRAFT_REPO_ROOT
python - <<'PY'
import sys, os
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import bench
from helpers import write_fasta
seqs = bench.synth_contigs(32, 1000000, 12345)
write_fasta("/tmp/s32.fa", [("c%d" % i, s.decode()) for i, s in enumerate(seqs)])
PY
export AUGUSTUS_CONFIG_PATH=$(python -c "import sys; sys.path.insert(0,'tests'); from helpers import config_path; print(config_path())")
AUGX_TIMING=1 AUGX_DEVICES=1 ./augustus_amd/bin/augustus --species=human --sample=100 --outfile=/tmp/o.gff /tmp/s32.fa 2>&1 | grep 'augx timing'
# ... and fly at its defaults (UTR on: the dense kernels; sample 100; 200 kb pieces) on 8 of the contigs
head -c 8200000 /tmp/s32.fa > /tmp/s8.fa
AUGX_TIMING=1 AUGX_DEVICES=1 ./augustus_amd/bin/augustus --species=fly --outfile=/tmp/o2.gff /tmp/s8.fa 2>&1 | grep 'augx timing' | grep -v 'batch on device'
