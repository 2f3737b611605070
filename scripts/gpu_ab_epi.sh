#!/bin/bash
# Same-box A/B of the reading epilogues: fused SwiGLU-backward GEMM and the residual GEMMs, base vs new library,
# then the whole step for LIBS.
cd $GRAFT_REPO_ROOT
L=$PWD/language-diffusion-model_amd/lib
for i in 1 2; do
  for which in base cur; do
    if [ $which = cur ]; then unset LDM_LIB; else export LDM_LIB=$L/libldm_hip_$which.so; fi
    echo "$which: $(timeout -k 10 100 python3 scripts/bench_swiglu_gemm.py 2>&1 | grep fused)"
    echo "$which: $(RES=1 ONLY=fwd timeout -k 10 100 python3 scripts/bench_gemm.py 2>&1 | grep "^fwd" | awk '{print $4"x"$6":"$7}' | tr '\n' ' ')"
    echo "$which: $(ONLY=dx timeout -k 10 100 python3 scripts/bench_gemm.py 2>&1 | grep "^dx" | awk '{print $4"x"$6":"$7}' | tr '\n' ' ')"
  done
done
unset LDM_LIB
LIBS="${LIBS:-libldm_hip_base.so libldm_hip_v1.so libldm_hip.so}" bash scripts/ab_step_libs.sh
