#!/bin/bash
# A/B of one environment switch on the headline bench with the per-tag kernel table:  AB_VAR=NAME AB_VALUES="1 0" bash tools/ab_bench.sh
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out/r3
for v in ${AB_VALUES:-1 0}; do
  echo "${AB_VAR}=$v"
  env ${AB_VAR}=$v timeout -k 10 300 python bench.py --steps 40 --warmup 8 --no-cpu-baseline --no-secondary > gpurun_out/r3/ab_bench_$v.json 2> gpurun_out/r3/ab_bench_$v.err || exit 1
  python - <<PY
import json
d=json.loads(open('gpurun_out/r3/ab_bench_$v.json').read().strip().splitlines()[-1])
print('ms/step %.3f median %.3f hbm_frac %.4f'%(d['ms_per_step'], d['timing']['ms_per_step_median_events'], d['step_roofline']['hbm_frac']))
k=json.load(open('gpurun_out/bench_kernels_c32nb_f32_n1.json'))
for t,v in sorted(k.items(), key=lambda kv:-kv[1]['ms'])[:${AB_TOP:-6}]:
    print('  %-34s n %5.1f  ms %7.3f  avg_us %7.1f  GB/s %6.0f'%(t[:34],v['count']/3,v['ms']/3,v['avg_us'],v['GBps']))
PY
done
