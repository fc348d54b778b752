#!/usr/bin/env python3
"""Where the wavefronts of each kernel spend their cycles, from one rocprofv3 PMC pass of SQ counters (profiles/run_sq.sh):
    SQ_WAVE_CYCLES = SQ_WAIT_ANY (parked: s_waitcnt / barrier) + SQ_WAIT_INST_ANY (issue stall) + SQ_ACTIVE_INST_ANY (issuing),
all in quad-cycles (/opt/skills/guides/MI355X_MICROARCH.md, the SQ row of the counter table); instruction counts beside them.
usage: summarize_sq.py results.db [source_sha] > rNN_<tag>_sq.txt     (values: the LAST dispatch of each kernel, summed over the chip)"""
import sqlite3
import sys

c = sqlite3.connect(sys.argv[1])
vals = {}
for name, cn, val, disp in c.execute("select kernel_name, counter_name, value, dispatch_id from counters_collection order by dispatch_id"):
    short = name.split("(")[0].replace("void ", "")
    vals.setdefault(short, {}).setdefault(cn, {})[disp] = vals.get(short, {}).get(cn, {}).get(disp, 0.0) + val
print("# rocprofv3 --kernel-trace --pmc SQ_* ; last dispatch of each kernel; cycles in quad-cycles summed over all wavefronts")
print("%-28s %10s %14s %8s %8s %8s %12s %12s %12s %12s" % ("kernel", "waves", "wave_cycles", "parked", "stall", "issuing", "VALU", "SALU", "VMEM_RD", "LDS"))
for k in sorted(vals, key=lambda k: -max(vals[k].get("SQ_WAVE_CYCLES", {0: 0}).values())):
    v = {cn: d[max(d)] for cn, d in vals[k].items()}
    wc = v.get("SQ_WAVE_CYCLES", 0.0)
    if wc <= 0:
        continue
    print("%-28s %10.0f %14.4g %7.1f%% %7.1f%% %7.1f%% %12.4g %12.4g %12.4g %12.4g" % (
        k[:28], v.get("SQ_WAVES", 0), wc, 100 * v.get("SQ_WAIT_ANY", 0) / wc, 100 * v.get("SQ_WAIT_INST_ANY", 0) / wc,
        100 * v.get("SQ_ACTIVE_INST_ANY", 0) / wc, v.get("SQ_INSTS_VALU", 0), v.get("SQ_INSTS_SALU", 0), v.get("SQ_INSTS_VMEM_RD", 0), v.get("SQ_INSTS_LDS", 0)))
if len(sys.argv) > 2:
    print("# source_sha: %s" % sys.argv[2])
