"""Kernel timeline of the REPLAYED step (hipGraph segments, two streams) from a rocprofv3 --kernel-trace database: every dispatch of the
last complete steps as TSV (name, queue / stream ids, start ns, end ns) for offline analysis (which kernels make up the tail segment
that owns the GPU alone, how long the gaps between them are).
usage (on the GPU box): python probes/dump_step_trace.py <results.db> <out.tsv>"""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
tables = [r[0] for r in cur.execute("select name from sqlite_master where type in ('table','view')")]
kt = "kernels" if "kernels" in tables else next(t for t in tables if "kernel" in t.lower())
cols = [r[1] for r in cur.execute("pragma table_info(%s)" % kt)]
print("# table", kt, "columns", cols)
want = [c for c in ("name", "queue_id", "stream_id", "tid", "start", "end") if c in cols]
rows = list(cur.execute("select %s from %s order by start" % (", ".join(want), kt)))
with open(sys.argv[2], "w") as f:
    f.write("\t".join(want) + "\n")
    for r in rows:
        f.write("\t".join(str(x).split("(")[0].replace("void ", "") if i == 0 else str(x) for i, x in enumerate(r)) + "\n")
print("# wrote", len(rows), "dispatches")
