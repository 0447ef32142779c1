#!/usr/bin/env python3
"""Summarise SQ counters of rocprofv3 --pmc passes per kernel: python tools/pmc_sq.py <dir> [<dir> ...]
Prints, for the kernels with the most wave-cycles, each counter's sum and its ratio to SQ_WAVE_CYCLES / SQ_BUSY_CYCLES."""
import collections, csv, glob, sys
agg = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.Counter()
for d in sys.argv[1:]:
    for f in glob.glob(d + '/*/*_counter_collection.csv') + glob.glob(d + '/*_counter_collection.csv'):
        seen = set()
        for r in csv.DictReader(open(f)):
            k = r['Kernel_Name'].split('(')[0].replace('void ', '')
            if 'mvae' not in k:
                continue
            agg[k][r['Counter_Name']] += float(r['Counter_Value'])
            key = (k, r.get('Dispatch_Id'))
            if key not in seen and d == sys.argv[1]:
                seen.add(key); cnt[k] += 1
names = sorted({c for v in agg.values() for c in v})
for k, v in sorted(agg.items(), key=lambda kv: -kv[1].get('SQ_WAVE_CYCLES', 0))[:int(1e9) if len(sys.argv) > 9 else 14]:
    wc = v.get('SQ_WAVE_CYCLES', 0) or 1
    print("%s  (launches %d)" % (k[:70], cnt[k]))
    print("   " + "  ".join("%s=%.3g(%.2f)" % (n.replace('SQ_', ''), v.get(n, 0), v.get(n, 0) / wc) for n in names))
