#!/bin/bash
# Locally, after tools/collect_profiles.sh ran through gpurun: copy / summarise what the judge reads into profiles/.
# (The bench lines profiles/round<R>_bench_line_*.json come from plain `python bench.py [--workload c256nb]` runs.)
set -e
cd "$(dirname "$0")/.."
R=${R:-4}
O=gpurun_out/r${R}prof
cp $O/stats_c32/c32_kernel_stats.csv profiles/round${R}_kernel_stats_c32nb_f32.csv
cp $O/stats_c256/c256_kernel_stats.csv profiles/round${R}_kernel_stats_c256nb_bf16.csv
python tools/pmc_traffic.py $O/pmc_FETCH_SIZE_c32 $O/pmc_WRITE_SIZE_c32 profiles/round${R}_pmc_traffic_c32nb.json | head -4
python tools/pmc_traffic.py $O/pmc_FETCH_SIZE_c256 $O/pmc_WRITE_SIZE_c256 profiles/round${R}_pmc_traffic_c256nb.json | head -4
R=$R python - <<'PY'
import csv, glob, json, collections, os
R = os.environ["R"]
for wl in ("c256", "c32"):
    fs = glob.glob('gpurun_out/r%sprof/pmc_mfma_%s/*_counter_collection.csv' % (R, wl)) + glob.glob('gpurun_out/r%sprof/pmc_mfma_%s/*/*_counter_collection.csv' % (R, wl))
    if not fs:
        continue
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
    for r in csv.DictReader(open(fs[0])):
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
        # GRBM_GUI_ACTIVE is summed over the 8 XCDs; an XCD has 32 CUs x 4 SIMDs = 128 matrix pipes
        out[k] = dict(launches=n[k], mfma_busy_cycles=v['SQ_VALU_MFMA_BUSY_CYCLES'], grbm_gui_active=v.get('GRBM_GUI_ACTIVE', 0),
                      mfma_util=v['SQ_VALU_MFMA_BUSY_CYCLES'] / max(v.get('GRBM_GUI_ACTIVE', 1) * 128, 1))
    json.dump(out, open('profiles/round%s_pmc_mfma_%snb.json' % (R, wl), 'w'), indent=1, sort_keys=True)
    top = sorted(out.items(), key=lambda kv: -kv[1]['mfma_busy_cycles'])[:8]
    for k, v in top:
        print(wl, "%-60s mfma_util %.3f" % (k[:60], v['mfma_util']))
PY
