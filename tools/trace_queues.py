#!/usr/bin/env python3
"""Per-queue busy/gap analysis of one train step from a rocprofv3 --kernel-trace CSV."""
import csv, glob, collections, sys
f = (glob.glob(sys.argv[1] + '/*/*_kernel_trace.csv') + glob.glob(sys.argv[1] + '/*_kernel_trace.csv'))[0]
rows = [r for r in csv.DictReader(open(f)) if 'mvae' in r['Kernel_Name']]
for r in rows:
    r['s'] = int(r['Start_Timestamp']); r['e'] = int(r['End_Timestamp'])
rows.sort(key=lambda r: r['s'])
starts = [r['s'] for r in rows if 'k_set_u64' in r['Kernel_Name']]
a, b = starts[-2], starts[-1]
step = [r for r in rows if a <= r['s'] < b]
print("step span ms %.3f kernels %d" % ((b - a) / 1e6, len(step)))
byq = collections.defaultdict(list)
for r in step:
    byq[r['Queue_Id']].append(r)
for q, rs in byq.items():
    rs.sort(key=lambda r: r['s'])
    busy = sum(r['e'] - r['s'] for r in rs)
    gaps = [rs[i + 1]['s'] - rs[i]['e'] for i in range(len(rs) - 1)]
    pos = [g for g in gaps if g > 0]
    print("queue %s n %d busy %.3f ms span %.3f ms gaps %.3f ms (median gap %.1f us)" % (
        q, len(rs), busy / 1e6, (rs[-1]['e'] - rs[0]['s']) / 1e6, sum(pos) / 1e6,
        sorted(pos)[len(pos) // 2] / 1e3 if pos else 0))
# union busy time over all queues (GPU doing anything)
ev = sorted([(r['s'], 1) for r in step] + [(r['e'], -1) for r in step])
busy, depth, last = 0, 0, None
for t, d in ev:
    if depth > 0: busy += t - last
    depth += d; last = t
print("any-queue busy %.3f ms" % (busy / 1e6))
qmain = max(byq, key=lambda q: sum(r['e'] - r['s'] for r in byq[q]))
agg = collections.defaultdict(lambda: [0, 0])
for r in byq[qmain]:
    k = r['Kernel_Name'].split('(')[0].replace('void ', '').replace('mvae::', '')[:44]
    agg[k][0] += r['e'] - r['s']; agg[k][1] += 1
print("-- busiest queue (scale 0 chain):")
for k, (t, n) in sorted(agg.items(), key=lambda kv: -kv[1][0])[:24]:
    print("  %-46s n=%3d  %.3f ms  avg %.1f us" % (k, n, t / 1e6, t / n / 1e3))
print("-- gaps > 15 us on the busiest queue (kernel before -> after, what ran elsewhere meanwhile):")
rs = byq[qmain]
short = lambda r: r['Kernel_Name'].split('(')[0].replace('void ', '').replace('mvae::', '')[:36]
for i in range(len(rs) - 1):
    g = rs[i + 1]['s'] - rs[i]['e']
    if g > 15000:
        other = [short(r) + "@q%s" % r['Queue_Id'] for r in step if r['Queue_Id'] != qmain and r['s'] < rs[i + 1]['s'] and r['e'] > rs[i]['e']]
        print("  %.1f us at +%.3f ms: %s -> %s | %s" % (g / 1e3, (rs[i]['e'] - a) / 1e6, short(rs[i]), short(rs[i + 1]), ", ".join(other[:6])))
