#!/usr/bin/env python
"""Developer aid (GPU box): the phases of augx_decode_batch on contigs with GC-content steps (bench.py's gc_steps leg), AUGX_TIMING=1"""
import os, sys, time, ctypes
os.environ["AUGX_TIMING"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import augustus_amd as ax
import bench
from helpers import config_path
m = ax.Model(config_path(), "human")
d = ax.Decoder(m, 0)
iso = bench.synth_isochore_contigs(100, 1000000, bench.SEED0 + 4242)
for rep in range(2):
    t0 = time.perf_counter()
    res = d.decode(iso)
    print("decode of 100 x 1 Mbp with GC steps: %.3f s" % (time.perf_counter() - t0), flush=True)
