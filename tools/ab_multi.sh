#!/bin/bash
# Interleaved comparison of several "ENV=val ENV2=val" settings on one box:  tools/ab_multi.sh REPS "A=1 B=2" "A=0" -- [bench.py arguments]
cd "${GRAFT_REPO_ROOT:-/root/repo}"
REPS=$1; shift
SETS=()
while [ $# -gt 0 ] && [ "$1" != "--" ]; do SETS+=("$1"); shift; done
shift
for r in $(seq 1 $REPS); do
  for s in "${SETS[@]}"; do
    env $s python bench.py --steps 60 --warmup 20 --no-cpu-baseline --no-secondary --no-kernel-profile "$@" 2>/dev/null | S="$s" python -c "
import sys, json, os
d = json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('%-44s ms/step %.4f  median %.4f  hbm_frac %.4f' % (os.environ['S'], d['ms_per_step'], d['timing']['ms_per_step_median_events'], d['step_roofline']['hbm_frac']))"
  done
done
