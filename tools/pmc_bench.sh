#!/bin/bash
# SQ counters of a bench workload's kernels, two rocprofv3 --pmc passes (no trace flags beside them); GPU box, repo root.
#   OUT=gpurun_out/r4/pmc_c256 tools/pmc_bench.sh --workload c256nb --steps 2 --warmup 1
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=${OUT:-gpurun_out/r4/pmc_bench}; rm -rf $O; mkdir -p $O
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/p1 -o p -- python3 bench.py --no-cpu-baseline --no-secondary --no-kernel-profile "$@" > /dev/null 2>&1
echo "p1 rc=$?"
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VMEM SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD --output-format csv -d $O/p2 -o p -- python3 bench.py --no-cpu-baseline --no-secondary --no-kernel-profile "$@" > /dev/null 2>&1
echo "p2 rc=$?"
find $O -name "*.csv" -size +30M -delete
python3 tools/pmc_sq.py $O/p1 $O/p2 > $O/summary.txt 2>&1
head -60 $O/summary.txt
