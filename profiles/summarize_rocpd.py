#!/usr/bin/env python3
"""Turn a rocprofv3 (rocpd sqlite) result of `rocprofv3 --kernel-trace --stats -- python bench.py ...` into the
plain-text per-kernel summary committed under profiles/.  usage: summarize_rocpd.py results.db > rNN_kernel_stats.txt"""
import sqlite3
import sys

c = sqlite3.connect(sys.argv[1])
print("# rocprofv3 --kernel-trace --stats; durations in ns (the view reports microseconds; view top_kernels of %s)" % sys.argv[1].split("/")[-1])
print("%-78s %6s %16s %16s %8s" % ("kernel", "calls", "total_ns", "avg_ns", "pct"))
for name, calls, total, avg, pct in c.execute("select name,total_calls,total_duration,average,percentage from top_kernels"):
    print("%-78s %6d %16.0f %16.0f %8.3f" % (name[:78], calls, total * 1e3, avg * 1e3, pct))
# resources of every kernel of the decode as dispatched (one line per kernel symbol: grid, workgroup, LDS, scratch, registers)
try:
    seen = set()
    for row in c.execute("select name,grid_x,workgroup_x,lds_size,scratch_size,vgpr_count,accum_vgpr_count,sgpr_count from kernels "
                         "where name like '%kTrellis%' or name like '%kCand%' or name like '%kDense%' or name like '%kForward%' or name like '%kSignals%' "
                         "or name like '%kBacktrace%' or name like '%Scan%' or name like '%kSiteConsts%' or name like '%kUtr%' order by name"):
        key = (row[0], row[2], row[3])
        if key in seen:
            continue
        seen.add(key)
        print("# dispatch %-60s grid_x=%d workgroup_x=%d lds=%d B scratch=%d B/lane vgpr=%d agpr=%d sgpr=%d" % ((row[0].split("(")[0][:60],) + tuple(row[1:])))
except sqlite3.Error as e:
    print("# (kernel resource table not available: %s)" % e)
