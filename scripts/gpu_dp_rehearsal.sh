#!/bin/bash
# Multi-rank rehearsal on a one-GPU box: (1) 2 ranks sharing the GPU over gloo (host-staged buckets),
# (2) 1 rank with the bucket/all-reduce machinery forced on over RCCL.  Prints loss and step time.
set -o pipefail
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
LDM_DP_BACKEND=gloo LDM_DP_SAME_GPU=1 timeout -k 10 240 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus 2 --steps 3 --warmup 1 --batch 16 > gpurun_out/dp2_gloo.json 2> gpurun_out/dp2_gloo.err || { echo "gloo x2 FAILED"; tail -5 gpurun_out/dp2_gloo.err; exit 1; }
python3 -c "import json; d=json.loads(open('gpurun_out/dp2_gloo.json').read().strip().splitlines()[-1]); print('gloo x2:', d['n_gpus'], 'ranks, loss', round(d['loss_last_step'],4), 'ms/step', round(d['ms_per_step'],1))"
LDM_DP_FORCE=1 timeout -k 10 240 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29612 bench.py --gpus 1 --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/dp1_nccl.json 2> gpurun_out/dp1_nccl.err || { echo "nccl x1 forced FAILED"; tail -5 gpurun_out/dp1_nccl.err; exit 1; }
python3 -c "import json; d=json.loads(open('gpurun_out/dp1_nccl.json').read().strip().splitlines()[-1]); print('rccl x1 (buckets forced):', 'loss', round(d['loss_last_step'],4), 'ms/step', round(d['ms_per_step'],2), 'buckets', d['grad_buckets'], 'backend', d['dp_backend'])"
timeout -k 10 240 python bench.py --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/dp0.json 2>/dev/null && python3 -c "import json; d=json.load(open('gpurun_out/dp0.json')); print('no DP:', 'loss', round(d['loss_last_step'],4), 'ms/step', round(d['ms_per_step'],2))"
