#!/bin/bash
# Same-box A/B of two library builds on the attention micro-benchmark: lib/libldm_hip.so vs lib/libldm_hip_exp.so
set -o pipefail
for r in 1 2; do
  for lib in libldm_hip.so libldm_hip_exp.so; do
    echo -n "$lib: "
    LDM_LIB=$PWD/language-diffusion-model_amd/lib/$lib SHAPE=${SHAPE:-512,512,8,96} timeout -k 10 120 python scripts/attn_kernels_bench.py || exit 1
  done
done
