#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
rocm-smi --showserial 2>/dev/null | grep -i "serial number:" | head -1
rocm-smi --showpower --showclocks --showtemp 2>/dev/null | grep -iE "power|sclk|Temperature \(Sensor junction" | head -6
run() { timeout -k 10 100 python - <<'PY' 2>&1 | grep -E "self-test|status"
import sys, os
sys.path.insert(0, os.getcwd())
from tests.common import engine_args
from multiscale_variational_autoencoder_amd.engine import Engine
import torch
torch.cuda.init()
eng = Engine(**engine_args("c32nb", 8)).bind(0)
print("status", eng.lib.mvae_split_conv_status())
PY
}
echo "== default"; MVAE_SPLIT_SELFTEST=2 run
for v in 0 1 2 3 4; do MVAE_SPLIT_SELFTEST=3 MVAE_SELFTEST_VICTIM=$v run; done
MVAE_SPLIT_SELFTEST=3 MVAE_SELFTEST_VICTIM=3 MVAE_SELFTEST_CONV=1 run
MVAE_SPLIT_SELFTEST=3 MVAE_SELFTEST_VICTIM=0 MVAE_SELFTEST_VB=64 run
MVAE_SPLIT_SELFTEST=3 MVAE_SELFTEST_VICTIM=4 MVAE_SELFTEST_VB=64 run
