#!/bin/bash
# float32 parity subset + headline bench A/B of one environment switch (quick check of a kernel change); GPU box, repo root
#   AB_VAR=MVAE_CONVT_MERGED AB_VALUES="1 0" OUT=gpurun_out/r4 tools/quick_f32.sh
cd "${GRAFT_REPO_ROOT:-/root/repo}"; O=${OUT:-gpurun_out/r4}; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_parity_gpu.py tests/test_deterministic_gpu.py tests/test_split_conv_gpu.py -x -q > $O/quick_f32.log 2>&1
rc=$?; tail -5 $O/quick_f32.log
[ $rc -ne 0 ] && exit $rc
for v in ${AB_VALUES:-1 0}; do
  echo "${AB_VAR:-MVAE_SPLIT_DUAL}=$v"
  env ${AB_VAR:-MVAE_SPLIT_DUAL}=$v timeout -k 10 300 python bench.py --steps 40 --warmup 8 --no-cpu-baseline --no-secondary > $O/quick_bench_$v.json 2> $O/quick_bench_$v.err || exit 1
  python - <<PY
import json
d=json.loads(open('$O/quick_bench_$v.json').read().strip().splitlines()[-1])
print('ms/step %.3f median %.3f hbm_frac %.4f'%(d['ms_per_step'], d['timing']['ms_per_step_median_events'], d['step_roofline']['hbm_frac']))
k=json.load(open('gpurun_out/bench_kernels_c32nb_f32_n1.json'))
for t,v in sorted(k.items(), key=lambda kv:-kv[1]['ms'])[:16]:
    print('  %-34s n %5.1f  ms %7.3f  avg_us %7.1f  GB/s %6.0f'%(t[:34],v['count']/3,v['ms']/3,v['avg_us'],v['GBps']))
PY
done
