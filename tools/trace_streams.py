#!/usr/bin/env python3
"""Per-queue (= HIP stream) breakdown of a rocprofv3 --kernel-trace CSV: for the LAST complete step-like window, the busy
time, span and idle time of every queue and the kernels that make up the busiest queue's time (by name).
    python tools/trace_streams.py <dir with *_kernel_trace.csv>"""
import collections, csv, glob, sys
fs = glob.glob(sys.argv[1] + '/*kernel_trace.csv') + glob.glob(sys.argv[1] + '/*/*kernel_trace.csv')
rows = []
for f in fs:
    for r in csv.DictReader(open(f)):
        rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Queue_Id'], r['Kernel_Name']))
rows.sort()
if not rows:
    sys.exit("no kernel trace rows")
# window: the last 3 steps' worth of kernels = from the 3rd-last k_opt_apply to the last one
marks = [i for i, r in enumerate(rows) if 'k_opt_apply' in r[3]]
if len(marks) < 4:
    sys.exit("not enough steps in the trace")
lo, hi = marks[-4] + 1, marks[-1] + 1
win = rows[lo:hi]
nsteps = 3
t0, t1 = win[0][0], max(r[1] for r in win)
print("window: %d kernels, %.3f ms per step" % (len(win), (t1 - t0) / 1e6 / nsteps))
byq = collections.defaultdict(list)
for r in win:
    byq[r[2]].append(r)
for q, rs in sorted(byq.items(), key=lambda kv: -sum(r[1] - r[0] for r in kv[1])):
    busy = sum(r[1] - r[0] for r in rs) / 1e6 / nsteps
    print("queue %s: %4d kernels/step, busy %.3f ms/step" % (q, len(rs) / nsteps, busy))
    agg = collections.defaultdict(lambda: [0, 0.0])
    for r in rs:
        k = r[3].split('(')[0].replace('void ', '').replace('mvae::', '')
        agg[k][0] += 1; agg[k][1] += (r[1] - r[0]) / 1e3
    for k, (n, us) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:28]:
        print("      %-58s n/step %5.1f  us/step %8.1f  avg %7.1f" % (k[:58], n / nsteps, us / nsteps, us / n))
