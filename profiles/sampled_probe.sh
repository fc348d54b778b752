set -e
cd $GRAFT_REPO_ROOT
python - <<'PY'
import sys, os
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import bench
from helpers import write_fasta
seqs = bench.synth_contigs(8, 1000000, 12345)
write_fasta("/tmp/s8.fa", [("c%d" % i, s.decode()) for i, s in enumerate(seqs)])
PY
export AUGUSTUS_CONFIG_PATH=$(python -c "import sys; sys.path.insert(0,'tests'); from helpers import config_path; print(config_path())")
for i in 1; do AUGX_TIMING=1 AUGX_DEVICES=1 ./augustus_amd/bin/augustus --species=human --sample=100 --outfile=/tmp/o.gff /tmp/s8.fa 2>&1 | grep 'augx timing' ; done
AUGX_TIMING=1 AUGX_DEVICES=1 ./augustus_amd/bin/augustus --species=fly --outfile=/tmp/o2.gff /tmp/s8.fa 2>&1 | grep 'augx timing'
