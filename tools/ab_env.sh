#!/bin/bash
# Interleaved A/B of one environment switch on the headline bench (GPU box, repo root): boxes of the pool differ by +-3 % and a
# process's first seconds run slower, so the two arms alternate in ONE call and every arm is measured REPS times.
#   tools/ab_env.sh MVAE_SPLIT_WGRAD "1 0" 3 [extra bench.py arguments]
cd "${GRAFT_REPO_ROOT:-/root/repo}"
VAR=$1; VALS=$2; REPS=${3:-3}; shift 3
for r in $(seq 1 $REPS); do
  for v in $VALS; do
    env $VAR=$v python bench.py --steps 60 --warmup 20 --no-cpu-baseline --no-secondary --no-kernel-profile "$@" 2>/dev/null | python -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('$VAR=$v  ms/step %.4f  median %.4f  hbm_frac %.4f' % (d['ms_per_step'], d['timing']['ms_per_step_median_events'], d['step_roofline']['hbm_frac']))"
  done
done
