#!/bin/bash
# Timing experiments on k_gemm_dual_s: rebuilds kernels_split.hip with -DMVAE_DUAL_VARIANT=n (parts of the kernel removed;
# results are wrong by construction) and links tools/dual_probe against it.  build: `bash tools/dual_variants.sh build`
# (here, hipcc), run: `bash tools/dual_variants.sh` on the GPU box.
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out/r3
C=multiscale_variational_autoencoder_amd/csrc
if [ "$1" = build ]; then
  OBJS=$(ls $C/*.o | grep -v -e runtime.o -e kernels_split.o)
  for v in ${VARIANTS:-1 2 3 4 5}; do
    hipcc -Xclang -target-feature -Xclang -packed-fp32-ops -Wno-unknown-attributes --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DMVAE_DUAL_VARIANT=$v \
      -x hip -c $C/kernels_split.hip -o /tmp/ks_v$v.o 2>/dev/null || exit 1
    hipcc --offload-arch=gfx950 tools/dual_probe.o /tmp/ks_v$v.o $OBJS -o tools/dual_probe_v$v.bin || exit 1
  done
  exit 0
fi
echo -n "product : "; MVAE_SPLIT_SELFTEST=0 ./tools/dual_probe.bin 512 32
for v in ${VARIANTS:-1 2 3 4 5}; do echo "variant $v : "; MVAE_SPLIT_SELFTEST=0 timeout -k 10 60 ./tools/dual_probe_v$v.bin 512 32 || exit 1; done 2>&1 | tee gpurun_out/r3/dual_variants.log
