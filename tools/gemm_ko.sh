#!/bin/bash
# Timing-only knock-out builds of gemm_f32.hip (GRL_GEMM_KO bits: 1 no in-loop staging, 2 no stage barrier,
# 4 no epilogue, 8 every stage re-reads k block 0): tools/_ko/libgrl_hip_ko<N>.so, selected with GRL_HIP_LIB=...  Results are WRONG by construction.
#   tools/gemm_ko.sh 1 2 3 4 7
set -e
cd "$(dirname "$0")/.."
make -s -C grl_amd/csrc -j8
mkdir -p tools/_ko
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -ffp-contract=off -Xclang -target-feature -Xclang -packed-fp32-ops"
OTHERS=$(ls grl_amd/csrc/*.o | grep -v gemm_f32.o)
for n in "$@"; do
  /opt/rocm/bin/hipcc $FLAGS -DGRL_GEMM_PIPE=0 -DGRL_GEMM_KO=$n -c grl_amd/csrc/gemm_f32.hip -o tools/_ko/gemm_f32_ko$n.o 2>&1 | grep -v "packed-fp32-ops' is not a recognized" || true
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OTHERS tools/_ko/gemm_f32_ko$n.o -o tools/_ko/libgrl_hip_ko$n.so
  echo built tools/_ko/libgrl_hip_ko$n.so
done
