#!/bin/bash
# Same-box comparison of several library builds on the attention micro-benchmark: LIBS="a.so b.so ..." (names under lib/)
set -o pipefail
for r in 1 2; do
  for lib in $LIBS; do
    echo -n "$lib: "
    LDM_LIB=$PWD/language-diffusion-model_amd/lib/$lib SHAPE=${SHAPE:-512,512,8,96} timeout -k 10 120 python scripts/attn_kernels_bench.py 2>&1 | grep -v amdgpu || exit 1
  done
done
