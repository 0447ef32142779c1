#!/bin/bash
# Locally, after tools/collect_profiles.sh ran through gpurun: copy / summarise what the judge reads into profiles/.
# (The bench lines profiles/round3_bench_line_*.json come from plain `python bench.py [--workload c256nb]` runs.)
set -e
cd "$(dirname "$0")/.."
O=gpurun_out/r3prof
cp $O/stats_c32/c32_kernel_stats.csv profiles/round3_kernel_stats_c32nb_f32.csv
cp $O/stats_c256/c256_kernel_stats.csv profiles/round3_kernel_stats_c256nb_bf16.csv
python tools/pmc_traffic.py $O/pmc_FETCH_SIZE_c32 $O/pmc_WRITE_SIZE_c32 profiles/round3_pmc_traffic_c32nb.json | head -4
python tools/pmc_traffic.py $O/pmc_FETCH_SIZE_c256 $O/pmc_WRITE_SIZE_c256 profiles/round3_pmc_traffic_c256nb.json | head -4
python - <<'PY'
import csv, glob, json, collections
f = glob.glob('gpurun_out/r3prof/pmc_mfma_c256/*_counter_collection.csv')[0]
agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for r in csv.DictReader(open(f)):
    k = r['Kernel_Name'].split('(')[0].replace('void ', '')
    if 'mvae' not in k:
        continue
    agg[k][r['Counter_Name']] += float(r['Counter_Value'])
    if r['Counter_Name'] == 'GRBM_GUI_ACTIVE':
        n[k] += 1
out = {}
for k, v in agg.items():
    if v.get('SQ_VALU_MFMA_BUSY_CYCLES', 0) <= 0:
        continue
    out[k] = dict(launches=n[k], mfma_busy_cycles=v['SQ_VALU_MFMA_BUSY_CYCLES'], grbm_gui_active=v.get('GRBM_GUI_ACTIVE', 0),
                  mfma_util=v['SQ_VALU_MFMA_BUSY_CYCLES'] / max(v.get('GRBM_GUI_ACTIVE', 1) * 128, 1))
json.dump(out, open('profiles/round3_pmc_mfma_c256nb.json', 'w'), indent=1, sort_keys=True)
PY
