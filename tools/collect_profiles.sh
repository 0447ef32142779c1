#!/bin/bash
# Profile collection of a round (R=4 -> gpurun_out/r4prof) on the GPU box (run through gpurun): rocprofv3 kernel stats of the two
# bench workloads, then separate --pmc passes (FETCH_SIZE / WRITE_SIZE / MFMA busy; no trace flags beside them) per
# MI355X_MICROARCH.md.  Summaries are copied into profiles/ by tools/profiles_summarise.sh afterwards (locally).
set -o pipefail
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
R=${R:-4}
O=gpurun_out/r${R}prof
rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_c32 -o c32 -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary > $O/bench_c32_stats.json 2> $O/bench_c32_stats.err
echo "stats c32 rc=$?"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_c256 -o c256 -- python3 bench.py --workload c256nb --steps 6 --warmup 2 --no-cpu-baseline > $O/bench_c256_stats.json 2> $O/bench_c256_stats.err
echo "stats c256 rc=$?"
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $C --output-format csv -d $O/pmc_${C}_c32 -o p -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-secondary --no-kernel-profile > /dev/null 2>&1
  echo "pmc $C c32 rc=$?"
  rocprofv3 --pmc $C --output-format csv -d $O/pmc_${C}_c256 -o p -- python3 bench.py --workload c256nb --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-profile > /dev/null 2>&1
  echo "pmc $C c256 rc=$?"
done
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_mfma_c256 -o p -- python3 bench.py --workload c256nb --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-profile > /dev/null 2>&1
echo "pmc mfma c256 rc=$?"
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_mfma_c32 -o p -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-secondary --no-kernel-profile > /dev/null 2>&1
echo "pmc mfma c32 rc=$?"
# the trace csv is large: keep the stats / counter summaries only
find $O -name "*kernel_trace.csv" -delete
find $O -name "*.csv" -size +20M -delete
du -sh $O
