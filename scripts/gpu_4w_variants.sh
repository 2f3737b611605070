#!/bin/bash
# Store-spreading variants of the four-wave GEMM (libraries built with -DLDM4W_NOW / -DLDM4W_SPS): kernel tests per
# variant, then the step's forward shapes plain and with bias + residual, alternating over the variants.
set -o pipefail
L=language-diffusion-model_amd/lib
VARS=${VARS:-"base n8s2 n16s2 n16s4 n16s1"}
lib_of() { if [ $1 = base ]; then echo $PWD/$L/libldm_hip.so; else echo $PWD/$L/libldm_hip_4w_$1.so; fi; }
for v in $VARS; do
  LDM_LIB=$(lib_of $v) timeout -k 10 300 python -u -m pytest tests/test_gpu_kernels.py -m gpu -q -x --tb=short --timeout 200 --timeout-method thread -p no:cacheprovider -k "four_wave" > gpurun_out/t_4wv_$v.log 2>&1 || { echo "$v: tests failed"; tail -30 gpurun_out/t_4wv_$v.log; exit 1; }
  echo "$v: $(tail -1 gpurun_out/t_4wv_$v.log)"
done
for i in 1 2; do
  for v in $VARS; do
    echo "== $v plain"; LDM_LIB=$(lib_of $v) ONLY=fwd T=262144 REPS=10 timeout -k 10 120 python scripts/bench_gemm.py || exit 1
    echo "== $v bias+residual"; LDM_LIB=$(lib_of $v) RES=1 ONLY=fwd T=262144 REPS=10 timeout -k 10 120 python scripts/bench_gemm.py || exit 1
  done
done
