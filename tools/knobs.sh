#!/bin/bash
# environment sweeps of the headline bench (one line per setting); GPU box, repo root
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out/r3
run() { echo -n "$* : "; env "$@" python bench.py --steps 60 --warmup 60 --no-cpu-baseline --no-secondary --no-kernel-profile 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.3f ms  median %.3f' % (d['ms_per_step'], d['timing']['ms_per_step_median_events']))"; }
run X=0
run MVAE_FUSED_CUS8=128
run MVAE_FUSED_CUS8=64
run MVAE_FUSED_CUS16=128
run MVAE_FUSED_CUS16=128 MVAE_FUSED_CUS8=64
run X=1
