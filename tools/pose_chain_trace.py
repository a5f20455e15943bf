"""Reduce a rocprofv3 kernel trace of the steady loop (tools/host_wait.py) to the life of a pose step on its stream: how long
the kernels of one step take from the first start to the last end, the gaps between them, and how often a stream delivers."""
import csv, collections, statistics, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    r['s'] = int(r['Start_Timestamp']); r['e'] = int(r['End_Timestamp']); r['n'] = r['Kernel_Name'].split('(')[0].replace('void ', '')
byq = collections.defaultdict(list)
for r in rows:
    byq[r['Queue_Id']].append(r)
t0 = min(r['s'] for r in rows)
FIRST = ('k_bf_knn2', 'k_pose_fused')
LAST = ('k_pose_solve', 'k_pose_fused')
chains = []
for q, rs in byq.items():
    rs.sort(key=lambda r: r['s'])
    if not any(r['n'].startswith(LAST) for r in rs):
        continue
    print("queue", q, "dispatches", len(rs), collections.Counter(r['n'] for r in rs).most_common(5))
    cur = None
    for r in rs:
        if r['n'].startswith(FIRST):
            cur = [r]
        elif cur is not None:
            cur.append(r)
        if r['n'].startswith(LAST) and cur:
            chains.append((q, cur)); cur = None
chains.sort(key=lambda c: c[1][0]['s'])
chains = chains[len(chains) // 4:]
dur = [(c[-1]['e'] - c[0]['s']) / 1e3 for _, c in chains]
busy = [sum(r['e'] - r['s'] for r in c) / 1e3 for _, c in chains]
print("steps %d: first start -> last end us: median %.1f mean %.1f max %.1f ; kernels' own time: median %.1f" %
      (len(chains), statistics.median(dur), sum(dur) / len(dur), max(dur), statistics.median(busy)))
perq = collections.defaultdict(list)
for q, c in chains:
    perq[q].append(c)
for q, cs in perq.items():
    gaps = [(b[0]['s'] - a[-1]['e']) / 1e3 for a, b in zip(cs, cs[1:])]
    per = [(b[-1]['e'] - a[-1]['e']) / 1e3 for a, b in zip(cs, cs[1:])]
    print("  queue %s: %d steps, period median %.1f us, idle between steps median %.1f us" % (q, len(cs), statistics.median(per), statistics.median(gaps)))
ends = sorted(c[-1]['e'] for _, c in chains)
g = [(b - a) / 1e3 for a, b in zip(ends, ends[1:])]
print("all streams: a step ends every %.1f us (median), mean %.1f" % (statistics.median(g), sum(g) / len(g)))
# engines: period of the diagonal sweep per queue
for q, rs in sorted(byq.items(), key=lambda kv: int(kv[0])):
    d = [r for r in rs if r['n'].startswith('k_sgbm_diag')]
    if len(d) > 4:
        per = [(b['e'] - a['e']) / 1e3 for a, b in zip(d, d[1:])]
        first = [r for r in rs if r['n'].startswith('k_sgbm_planes')]
        print("  engine queue %s: %d pairs, period median %.0f us" % (q, len(d), statistics.median(per)))
