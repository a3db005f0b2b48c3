"""Summarise a rocprofv3 --kernel-trace --stats sqlite result (rocpd .db) as a per-kernel table.
usage: python probes/prof_summary.py gpurun_out/prof_x/x_results.db [steps] > profiles/x_kernel_stats.txt"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rows = list(db.execute("select name, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) from kernels group by name order by 3 desc"))
tot = sum(r[2] for r in rows)
print("# rocprofv3 --kernel-trace --stats summary of %s  (%d profiled steps incl. warm-up/instrumented)" % (sys.argv[1].split("/")[-1], steps))
print("# total kernel time %.3f ms  (%.3f ms per step)" % (tot / 1e6, tot / 1e6 / steps))
print("%-64s %7s %12s %10s %10s %10s %6s" % ("kernel", "calls", "total_us", "avg_us", "min_us", "max_us", "%"))
for n, c, t, a, mn, mx in rows:
    short = n.split("(")[0].replace("void ", "")
    if len(short) > 62:
        short = short[:59] + "..."
    print("%-64s %7d %12.1f %10.2f %10.2f %10.2f %6.2f" % (short, c, t / 1e3, a / 1e3, mn / 1e3, mx / 1e3, 100.0 * t / tot))
