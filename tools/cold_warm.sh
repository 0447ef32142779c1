#!/bin/bash
# is the first bench process on a fresh box slower than later ones (GPU clock / power state), and does a longer warm-up cure it?
cd "${GRAFT_REPO_ROOT:-/root/repo}"
run() { echo -n "$* : "; python bench.py --no-cpu-baseline --no-secondary --no-kernel-profile "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.3f ms  median %.3f min %.3f p90 %.3f' % (d['ms_per_step'], d['timing']['ms_per_step_median_events'], d['timing']['ms_per_step_min_events'], d['timing']['ms_per_step_p90_events']))"; }
run --steps 50 --warmup 10
run --steps 50 --warmup 10
run --steps 50 --warmup 300
run --steps 200 --warmup 10
run --steps 50 --warmup 10
