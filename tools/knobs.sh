#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out/r3
run() { echo -n "$* : "; env "$@" python bench.py --steps 40 --warmup 8 --no-cpu-baseline --no-secondary --no-kernel-profile 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.3f ms  median %.3f' % (d['ms_per_step'], d['timing']['ms_per_step_median_events']))"; }
run X=0
run MVAE_DUAL_SMALL_TILES=2048 MVAE_DUAL_BPC_SMALL=1
run MVAE_DUAL_SMALL_TILES=8192 MVAE_DUAL_BPC_SMALL=1
run MVAE_DUAL_SMALL_TILES=512 MVAE_DUAL_BPC_SMALL=1
run X=1
