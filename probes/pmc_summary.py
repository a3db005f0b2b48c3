"""Per-kernel average of one PMC counter from a rocprofv3 --pmc run (rocpd sqlite).  Runs ON the GPU box right after
the profiled command (the databases are too large to ship back):
  python probes/pmc_summary.py <results.db> <COUNTER> > gpurun_out/pmc_<counter>.txt"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
want = sys.argv[2]
cur = db.cursor()
cols = [r[1] for r in cur.execute("pragma table_info(counters_collection)")]
print("# columns:", cols)
name_col = "counter_name" if "counter_name" in cols else "name"
q = ("select kernel_name, count(*), avg(value), min(value), max(value), sum(value) from counters_collection "
     "where %s = ? group by kernel_name order by 6 desc" % name_col)
rows = list(cur.execute(q, (want,)))
print("# %s per dispatch (raw counter units as rocprofv3 reports them)" % want)
print("%-64s %7s %14s %14s %14s" % ("kernel", "calls", "avg", "min", "max"))
for n, c, a, mn, mx, tot in rows:
    short = n.split("(")[0].replace("void ", "")[:62]
    print("%-64s %7d %14.1f %14.1f %14.1f" % (short, c, a, mn, mx))
