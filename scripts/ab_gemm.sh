#!/bin/bash
# A/B of two builds of the library on ONE box (boxes differ by up to 30 %): lib/libldm_hip.so vs lib/libldm_hip_exp.so
for i in 1 2; do
  echo BASE; ONLY=${ONLY:-fwd,dx,dw} timeout -k 10 100 python scripts/bench_gemm.py || exit 1
  echo EXP; LDM_LIB=$PWD/language-diffusion-model_amd/lib/libldm_hip_exp.so ONLY=${ONLY:-fwd,dx,dw} timeout -k 10 100 python scripts/bench_gemm.py || exit 1
done
