#!/bin/bash
# GEMM kernel tests against the current library build, then the forward shapes with bias + residual with the
# four-wave kernel off and on (same box, alternating)
set -o pipefail
timeout -k 10 500 python -u -m pytest tests/test_gpu_kernels.py -m gpu -q -x --tb=short --timeout 200 --timeout-method thread -p no:cacheprovider -k "gemm or linear or swiglu or head" > gpurun_out/t_4w.log 2>&1 || { tail -40 gpurun_out/t_4w.log; exit 1; }
tail -2 gpurun_out/t_4w.log
for i in 1 2; do
  echo "8-wave"; LDM_GEMM_4W=0 RES=1 ONLY=fwd T=${T:-262144} REPS=10 timeout -k 10 120 python scripts/bench_gemm.py || exit 1
  echo "4-wave"; LDM_GEMM_4W=1 RES=1 ONLY=fwd T=${T:-262144} REPS=10 timeout -k 10 120 python scripts/bench_gemm.py || exit 1
done
