#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out/r3
run() { echo "== $*"; env "$@" python tools/det_spread.py 2>/dev/null | head -4; }
run MVAE_SPLIT_DUAL=1 MVAE_FUSE_DW_CONV0_F32=0
run MVAE_SPLIT_DUAL=0 MVAE_FUSE_DW_CONV0_F32=1
run MVAE_SPLIT_DUAL=1 MVAE_FUSE_DW_CONV0_F32=1 MVAE_STREAMS=0
run MVAE_SPLIT_DUAL=1 MVAE_FUSE_DW_CONV0_F32=1 MVAE_GRAPHS=0
run MVAE_SPLIT_DUAL=0 MVAE_FUSE_DW_CONV0_F32=0 MVAE_SPLIT_CONV=0
