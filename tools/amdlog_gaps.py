#!/usr/bin/env python3
"""Where does the host wait?  Reads an AMD_LOG_LEVEL=4 log and prints every jump of the time stamp above a threshold with the
lines around it.   python tools/amdlog_gaps.py LOG [threshold_us] [from_marker]"""
import re, sys
lines = open(sys.argv[1], errors="replace").read().splitlines()
thr = int(sys.argv[2]) if len(sys.argv) > 2 else 30
start = 0
if len(sys.argv) > 3:
    for i, l in enumerate(lines):
        if sys.argv[3] in l: start = i
prev = None
for i in range(start, len(lines)):
    m = re.search(r": (\d+) us:", lines[i])
    if not m: continue
    t = int(m.group(1))
    if prev is not None and t - prev[0] > thr:
        print("---- gap %d us" % (t - prev[0]))
        for j in range(max(start, prev[1] - 2), i + 1): print("   ", lines[j][:220])
    prev = (t, i)
