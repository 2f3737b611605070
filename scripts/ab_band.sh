#!/bin/bash
# m-tiles per band of the GEMM tile walk (LDM_GEMM_BAND) on the step's shapes, one box
for b in 4 8 2 16 4 8; do
  echo "== band $b"
  LDM_GEMM_BAND=$b ONLY=${ONLY:-fwd,dx} T=262144 REPS=10 timeout -k 10 100 python scripts/bench_gemm.py 2>&1 | grep -v amdgpu || exit 1
done
