"""Stream timeline of a rocprofv3 --kernel-trace run of bench.py (rocpd sqlite): per-stream busy time, how much of the
wall time has 1 / 2 / 3+ kernels in flight, and the longest idle gaps of the busiest stream with what ran elsewhere.
Runs ON the GPU box:  python probes/timeline.py <results.db> [first_step_fraction_to_skip]"""
import collections
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
rows = db.execute("select start, end, name, stream_id, queue_id from kernels order by start").fetchall()
t0, t1 = rows[0][0], max(r[1] for r in rows)
# keep the last 45 % of the trace: steady-state timed steps (warm-up and model build come first)
cut = t0 + int((t1 - t0) * 0.55)
rows = [r for r in rows if r[0] >= cut]
t0, t1 = rows[0][0], max(r[1] for r in rows)
span = t1 - t0
print("analysed window %.1f ms, %d kernels" % (span / 1e6, len(rows)))
by = collections.defaultdict(list)
for s, e, n, sid, qid in rows:
    by[(sid, qid)].append((s, e, n))
for k, v in sorted(by.items(), key=lambda kv: -sum(e - s for s, e, _ in kv[1])):
    busy = sum(e - s for s, e, _ in v)
    print("stream %s queue %s: %5d kernels, busy %.1f ms (%.0f %% of window)" % (k[0], k[1], len(v), busy / 1e6, 100.0 * busy / span))
ev = []
for s, e, n, sid, qid in rows:
    ev.append((s, 1)); ev.append((e, -1))
ev.sort()
depth, last, hist = 0, t0, collections.Counter()
for t, d in ev:
    hist[min(depth, 4)] += t - last
    last = t
    depth += d
print("kernels in flight: " + "  ".join("%d%s: %.1f %%" % (k, "+" if k == 4 else "", 100.0 * v / span) for k, v in sorted(hist.items())))
main = max(by.items(), key=lambda kv: sum(e - s for s, e, _ in kv[1]))[1]
gaps = sorted(((main[i + 1][0] - main[i][1], main[i][1], main[i + 1][0], main[i][2], main[i + 1][2]) for i in range(len(main) - 1)), reverse=True)
print("busiest stream: idle %.1f ms in gaps > 20 us (%d gaps), %.1f ms in all gaps" % (
    sum(g[0] for g in gaps if g[0] > 20000) / 1e6, sum(1 for g in gaps if g[0] > 20000), sum(max(g[0], 0) for g in gaps) / 1e6))
for g, a, b, before, after in gaps[:14]:
    other = collections.Counter()
    for s, e, n, sid, qid in rows:
        if s < b and e > a:
            other[n.split("(")[0].replace("void ", "")[:40]] += min(e, b) - max(s, a)
    print("  gap %6.1f us after %-28s before %-28s | meanwhile: %s" % (g / 1e3, before.split("(")[0][-28:], after.split("(")[0][-28:],
                                                                          ", ".join("%s %.0fus" % (k, v / 1e3) for k, v in other.most_common(3))))
