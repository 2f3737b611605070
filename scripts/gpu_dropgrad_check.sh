#!/bin/bash
# New norm-backward second output: its tests, the other dropout tests, then the dropout-0.1 leg of the bench with the
# hand-off off and on (same box).
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 500 python -u -m pytest tests/test_gpu_kernels.py tests/test_gpu_fullsize.py -x -q --timeout 300 --timeout-method thread \
  -m gpu -k "dropout or rmsnorm" > gpurun_out/dropgrad_tests.log 2>&1 || { tail -30 gpurun_out/dropgrad_tests.log; exit 1; }
tail -3 gpurun_out/dropgrad_tests.log
for i in 1 2; do
  for v in 0 1; do
    LDM_FUSED_DROP_GRAD=$v timeout -k 10 200 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/dropgrad_$v.json 2> gpurun_out/dropgrad_$v.err || { echo "bench $v failed"; tail -5 gpurun_out/dropgrad_$v.err; exit 1; }
    python3 -c "import json;d=json.load(open('gpurun_out/dropgrad_$v.json'));print('handoff=$v', round(d['ms_per_step'],3), 'ms/step; dropout leg', d['dropout01'])"
  done
done
