"""Launches per step by origin, from a rocprofv3 --kernel-trace sqlite result of a run whose steps start with patchify_kernel
(any launch mode: graph replays are traced kernel by kernel): hand-written HIP kernels vs ATen / runtime copies.
usage: python probes/aten_count.py x_results.db"""
import collections
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
rows = list(db.execute("select name, start, end from kernels order by start"))
starts = [i for i, r in enumerate(rows) if r[0].startswith("patchify_kernel")]
a, b = starts[-2], starts[-1]
ours, aten = collections.Counter(), collections.Counter()
t_ours = t_aten = 0.0
for n, s, e in rows[a:b]:
    short = n.split("(")[0].replace("void ", "")
    if short.startswith("at::") or "rocclr" in short or short.startswith("Cijk") or "elementwise" in short:
        aten[short[:100]] += 1; t_aten += (e - s) / 1e3
    else:
        ours[short.split("<")[0]] += 1; t_ours += (e - s) / 1e3
print("# one step (the last complete one): %d launches, %d hand-written HIP (%.2f ms), %d ATen / runtime (%.3f ms)"
      % (b - a, sum(ours.values()), t_ours / 1e3, sum(aten.values()), t_aten / 1e3))
for n, k in aten.most_common():
    print("%4d  %s" % (k, n))
