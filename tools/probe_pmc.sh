#!/bin/bash
# SQ counters of the 5x5 probe kernels, two passes.  Run through gpurun:  OUT=gpurun_out/r4/pmc_probe tools/probe_pmc.sh
# default program: tools/conv_probe_s.bin 512 (float32 split kernels); bf16:  PROG='./tools/bf16_unit.bin time 64' MATCH='taps|wgrad' tools/probe_pmc.sh
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=${OUT:-gpurun_out/r4/pmc_probe}; rm -rf $O; mkdir -p $O
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/p1 -o p -- ${PROG:-./tools/conv_probe_s.bin 512} > /dev/null 2>&1
echo "p1 rc=$?"
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VMEM SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL --output-format csv -d $O/p2 -o p -- ${PROG:-./tools/conv_probe_s.bin 512} > /dev/null 2>&1
echo "p2 rc=$?"
O=$O MATCH="${MATCH:-conv}" python3 - <<'PY'
import csv, glob, collections, os, re
O = os.environ["O"]
agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for f in glob.glob(O + '/*/*_counter_collection.csv') + glob.glob(O + '/*/*/*_counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'].split('(')[0].replace('void ', '').replace('mvae::', '')
        if not re.search(os.environ['MATCH'], k) or 'mfma' in k: continue
        agg[k][r['Counter_Name']] += float(r['Counter_Value'])
        if r['Counter_Name'] in ('SQ_WAVE_CYCLES',): n[k] += 1
for k, v in agg.items():
    wc = v.get('SQ_WAVE_CYCLES', 1) or 1
    print(k, "launches", n[k])
    print("   " + "  ".join("%s=%.3g(%.3f)" % (c.replace('SQ_', ''), x, x / wc) for c, x in sorted(v.items())))
PY
