"""BASELINE config 5 at size: the 1.0 Gbp stand-in (tests/golden/make_golden_long.py: genome_1g_records) through the executable"""
import os, sys, subprocess, time, resource, tarfile, tempfile, threading, hashlib, gzip, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import numpy as np
from helpers import config_path, read_fasta, gff_body
from make_golden_long import genome_1g_records
d = tempfile.mkdtemp()
with tarfile.open(os.path.join(ROOT, "tests", "golden", "big_inputs.tar.gz")) as t:
    t.extractall(d)
only_first = os.environ.get("ONLY_FIRST") == "1"
t0 = time.time()
recs = genome_1g_records(read_fasta(os.path.join(d, "genome.fa"))[0][1], only_first=only_first)
print("records built in %.1f s:" % (time.time() - t0), [(n, len(s)) for n, s in recs], flush=True)
fa = os.path.join(d, "g1.fa")
t0 = time.time()
with open(fa, "wb") as f:
    for nm, sq in recs:
        f.write(b">" + nm.encode() + b"\n")
        arr = np.frombuffer(sq.encode(), dtype=np.uint8)
        k = len(arr) // 60 * 60
        f.write(np.concatenate([arr[:k].reshape(-1, 60), np.full((k // 60, 1), 10, dtype=np.uint8)], axis=1).tobytes())
        if k < len(arr):
            f.write(arr[k:].tobytes() + b"\n")
bases = sum(len(s) for _, s in recs)
nN = sum(s.count("N") for _, s in recs)
del recs
print("FASTA written in %.1f s, %.2f Gbp (%.1f Mbp of N)" % (time.time() - t0, bases / 1e9, nN / 1e6), flush=True)
import torch
for devs in os.environ.get("DEVS", "0").split(";"):
    env = dict(os.environ, AUGUSTUS_CONFIG_PATH=config_path(), AUGX_TIMING="1", AUGX_DEVICES=devs)
    free0 = torch.cuda.mem_get_info(0)[0]
    low = [free0]
    stop = threading.Event()
    def watch():
        while not stop.is_set():
            low[0] = min(low[0], torch.cuda.mem_get_info(0)[0])
            stop.wait(0.05)
    th = threading.Thread(target=watch); th.start()
    out = os.path.join(d, "o.gff")
    t0 = time.time()
    r = subprocess.run([os.path.join(ROOT, "augustus_amd", "bin", "augustus"), "--species=human", "--progress=true", "--outfile=" + out, fa], capture_output=True, env=env)
    dt = time.time() - t0
    stop.set(); th.join()
    err = r.stderr.decode(errors="replace")
    rss = resource.getrusage(resource.RUSAGE_CHILDREN).ru_maxrss / 1e6
    print("AUGX_DEVICES=%s rc %d wall %.2f s = %.1f Mbp/s, HBM high-water %.1f GB, host peak RSS %.1f GB" % (devs, r.returncode, dt, bases / 1e6 / dt, (free0 - low[0]) / 1e9, rss), flush=True)
    laps = [l for l in err.splitlines() if l.startswith("augx timing: ")]
    print("\n".join(laps))
    nb = sum(1 for l in err.splitlines() if l.startswith("augx timing:   batch on device"))
    print("batches:", nb, " cut finder line:", [l for l in err.splitlines() if "cut finder:" in l])
    if r.returncode != 0:
        print(err[-3000:])
    txt = "\n".join([l for l in err.splitlines() if l.startswith("examining piece")] + gff_body(open(out).read())) + "\n"
    print("sha256", hashlib.sha256(txt.encode()).hexdigest(), "lines", txt.count("\n"), "cuts", sum(1 for l in txt.splitlines() if l.startswith("examining piece")))
    if only_first:
        meta = json.load(open(os.path.join(ROOT, "tests", "golden", "golden_long.json"))).get("genome_1g_chr1")
        print("golden:", meta, "parity:", meta and meta["sha256"] == hashlib.sha256(txt.encode()).hexdigest())
    with open(os.path.join(ROOT, "gpurun_out", "genome_1g_timing_%s.txt" % devs.replace(",", "_")), "w") as fh:
        fh.write("\n".join(l for l in err.splitlines() if l.startswith("augx timing")) + "\n")
