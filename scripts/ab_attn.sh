#!/bin/bash
# A/B on one box: register-staged attention kernels vs the LDS-DMA ones.
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 300 python -u -m pytest tests/test_gpu_kernels.py -k attention -q --tb=short --timeout 120 --timeout-method thread -p no:cacheprovider > gpurun_out/attn_tests.log 2>&1
rc=$?; tail -5 gpurun_out/attn_tests.log | cut -c1-300
[ $rc -ne 0 ] && exit 1
for v in 0 1 0 1; do
  LDM_ATTN_DMA=$v timeout -k 10 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/ab_attn_$v.json 2> gpurun_out/ab_attn_$v.err || { echo BENCH FAILED; tail -5 gpurun_out/ab_attn_$v.err; exit 1; }
  python -c "
import json; d=json.load(open('gpurun_out/ab_attn_$v.json')); print('DMA=$v', d['ms_per_step'], d['value'])"
done
