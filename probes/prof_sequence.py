"""Ordered kernel sequence of ONE step from a rocprofv3 --kernel-trace sqlite result (serialized run: one stream), with
durations and gaps: usage: python probes/prof_sequence.py x_results.db > sequence.txt.  A step starts at its patchify_kernel."""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
rows = list(db.execute("select name, start, end from kernels order by start"))
starts = [i for i, r in enumerate(rows) if r[0].startswith("patchify_kernel")]
a, b = starts[-2], starts[-1]          # the last complete step
t0 = rows[a][1]
prev_end = t0
print("# kernels %d..%d of %d; step span %.3f ms" % (a, b, len(rows), (rows[b][1] - t0) / 1e6))
print("%5s %9s %8s %7s  %s" % ("#", "start_us", "dur_us", "gap_us", "kernel"))
for i in range(a, b):
    n, s, e = rows[i]
    short = n.split("(")[0].replace("void ", "")[:90]
    print("%5d %9.1f %8.2f %7.2f  %s" % (i - a, (s - t0) / 1e3, (e - s) / 1e3, (s - prev_end) / 1e3, short))
    prev_end = e
