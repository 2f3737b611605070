#!/bin/bash
# A/B of two builds of the library on ONE box (boxes differ by up to 30 %): lib/libldm_hip_base.so (BASE) vs lib/libldm_hip.so (EXP)
for i in 1 2; do
  echo BASE; LDM_LIB=$PWD/language-diffusion-model_amd/lib/libldm_hip_base.so ONLY=${ONLY:-fwd,dx,dw} timeout -k 10 100 python scripts/bench_gemm.py || exit 1
  echo EXP; ONLY=${ONLY:-fwd,dx,dw} timeout -k 10 100 python scripts/bench_gemm.py || exit 1
done
