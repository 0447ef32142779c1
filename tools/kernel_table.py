#!/usr/bin/env python3
"""Print the per-kernel table bench.py wrote (gpurun_out/bench_kernels_<workload>_n<N>.json)."""
import json, sys
path = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/bench_kernels_c32nb_n1.json"
nprof = 3
k = json.load(open(path))
tot = sum(v["ms"] for v in k.values())
print("tagged total ms per step %.3f" % (tot / nprof))
for n, v in sorted(k.items(), key=lambda kv: -kv[1]["ms"]):
    print("%-16s n/step %5.0f  ms/step %8.3f  share %5.1f%%  avg_us %8.1f  GB/s %8.1f  TF %6.2f" % (
        n, v["count"] / nprof, v["ms"] / nprof, 100 * v["share"], v["avg_us"], v["GBps"], v["TFLOPs"]))

# diagnostics (MVAE_PROF_SIZES=1): tags carry "/q<stream>"; sum per launch stream = composition of each scale's chain
if any("/q" in n for n in k):
    per = {}
    for n, v in k.items():
        q = n.rsplit("/q", 1)[1]
        per[q] = per.get(q, 0.0) + v["ms"] / nprof
    print("per stream ms/step:", {q: round(t, 3) for q, t in sorted(per.items())})
