"""the 23 Mbp contig of bench.py (long_contig_utr) through the executable with the cut finder's debug dump"""
import os, sys, subprocess, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench
from helpers import config_path
n = int(os.environ.get("LEN", "23000000"))
big = bench.synth_contigs(1, n, bench.SEED0 + 77)
fa = "/tmp/long.fa"
with open(fa, "wb") as f:
    f.write(b">long\n")
    s = big[0]
    for k in range(0, len(s), 60):
        f.write(s[k:k + 60] + b"\n")
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
tag = os.environ.get("TAG", "utr")
flags = ["--species=fly", "--UTR=on", "--sample=0", "--softmasking=0"]
env = dict(os.environ, AUGUSTUS_CONFIG_PATH=config_path(), AUGX_TIMING="1", AUGX_CUT_DEBUG=os.path.join(ROOT, "gpurun_out", "cutdbg_%s.txt" % tag))
t0 = time.time()
r = subprocess.run([os.path.join(ROOT, "augustus_amd", "bin", "augustus")] + flags + ["--outfile=/tmp/o.gff", fa], capture_output=True, env=env)
print("rc", r.returncode, "wall %.2f s" % (time.time() - t0))
print("\n".join(l for l in r.stderr.decode().splitlines() if l.startswith("augx timing: ")))
