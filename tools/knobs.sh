#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out/r3
run() { echo -n "$1 : "; env $1 python bench.py --workload c256nb --steps 12 --warmup 3 --no-cpu-baseline --no-secondary --no-kernel-profile 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.3f ms  median %.3f hbm %.4f' % (d['ms_per_step'], d['timing']['ms_per_step_median_events'], d['step_roofline']['hbm_frac']))"; }
for c in 256 240 224 208 256; do run "MVAE_BIG_CUS16=$c"; done
