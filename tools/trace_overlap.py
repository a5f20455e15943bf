"""Summarise a rocprofv3 kernel trace (CSV): GPU busy fraction, mean concurrency, and per-kernel
duration in the traced (overlapped) run.  usage: trace_overlap.py trace.csv [skip_fraction]"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
skip = float(sys.argv[2]) if len(sys.argv) > 2 else 0.3
ev = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][:44], r["Queue_Id"]) for r in rows]
ev.sort()
t0, t1 = ev[0][0], max(e[1] for e in ev)
cut = t0 + (t1 - t0) * skip
ev = [e for e in ev if e[0] >= cut]
t0, t1 = ev[0][0], max(e[1] for e in ev)
pts = []
for s, e, _, _ in ev: pts += [(s, 1), (e, -1)]
pts.sort()
busy = 0; area = 0; cur = 0; last = pts[0][0]; hist = collections.Counter()
for t, d in pts:
    if cur > 0: busy += t - last
    area += cur * (t - last); hist[cur] += t - last
    cur += d; last = t
span = t1 - t0
print("span %.2f ms, busy %.1f%%, mean concurrency while busy %.2f" % (span / 1e6, 100 * busy / span, area / max(busy, 1)))
print("time share by #kernels in flight:", {k: round(100 * v / span, 1) for k, v in sorted(hist.items())})
agg = collections.defaultdict(lambda: [0, 0])
for s, e, n, q in ev: agg[n][0] += 1; agg[n][1] += e - s
tot = sum(v[1] for v in agg.values())
print("sum of kernel durations / span = %.2f" % (tot / span))
for n, (c, d) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:16]:
    print("%-46s n=%5d avg=%8.1f us  share=%.1f%%" % (n, c, d / c / 1e3, 100 * d / tot))
print("queues:", collections.Counter(e[3] for e in ev))
