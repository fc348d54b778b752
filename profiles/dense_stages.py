#!/usr/bin/env python
"""Developer aid: where the dense trellis kernel (device/dense.h) spends its cycles, stage by stage.
Needs the -DAUGX_PROFILE build (the Makefile's default) and a GPU:  AUGX_PROF=1 python profiles/dense_stages.py [species] [contigs] [len]"""
import os, sys
if "AUGX_PROF" not in os.environ and "--prof" in sys.argv: os.environ["AUGX_PROF"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import augustus_amd as ax
from helpers import config_path, random_dna

argv = [a for a in sys.argv[1:] if not a.startswith("--")]
species = argv[0] if len(argv) > 0 else "human"
nc = int(argv[1]) if len(argv) > 1 else 256
ln = int(argv[2]) if len(argv) > 2 else 40000
m = ax.Model(config_path(), species, UTR="on", sample="0")
d = ax.Decoder(m, 0)
import bench
seqs = bench.synth_contigs(nc, ln, 1000)
b = ax.Batch(d, seqs)
b.decode(sync=True)
b.decode(sync=True)
k = b.kernel_ms()
print("%s S=%d  %d x %d bp: prep %.1f ms, kDense %.1f ms, back-trace %.1f ms -> %.1f Mbp/s in the kernel" %
      (species, m.n_states, nc, ln, k["prep_ms"], k["trellis_ms"], k["backtrace_ms"], nc * ln / k["trellis_ms"] / 1e3))
