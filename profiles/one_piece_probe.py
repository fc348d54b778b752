#!/usr/bin/env python
"""Developer aid (GPU box): what one cut-finder round costs -- a batch of ONE 50 kb piece (fly model, 47 states), decoded as the driver
does it (new batch object every time), wall clock and kernel times."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import augustus_amd as ax
import bench
from helpers import config_path
m = ax.Model(config_path(), "fly", UTR="off", sample="0", softmasking="0")
d = ax.Decoder(m, 0)
seqs = bench.synth_contigs(12, 50001, 77)
for i, s in enumerate(seqs):
    t0 = time.perf_counter()
    b = ax.Batch(d, [s])
    t1 = time.perf_counter()
    b.decode(sync=True)
    t2 = time.perf_counter()
    p = b.paths()
    t3 = time.perf_counter()
    k = b.kernel_ms()
    b.close()
    t4 = time.perf_counter()
    if i >= 2:
        print("create %.2f ms, decode %.2f ms (kernels: prep %.2f, trellis %.2f, back-trace %.2f), paths %.2f ms, close %.2f ms" %
              ((t1 - t0) * 1e3, (t2 - t1) * 1e3, k["prep_ms"], k["trellis_ms"], k["backtrace_ms"], (t3 - t2) * 1e3, (t4 - t3) * 1e3))
