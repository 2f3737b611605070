#!/bin/bash
# 2 ranks sharing the one GPU of the box: exercises the bench's multi-rank path (hooks, buckets, barrier).
set -o pipefail
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
for be in nccl gloo; do
  echo "== backend $be"
  LDM_DP_BACKEND=$be LDM_DP_SAME_GPU=1 timeout -k 10 240 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus 2 --steps 3 --warmup 1 --batch 8 > gpurun_out/dp2_$be.json 2> gpurun_out/dp2_$be.err
  echo "rc=$?"; tail -c 600 gpurun_out/dp2_$be.json; tail -5 gpurun_out/dp2_$be.err | cut -c1-300
done
