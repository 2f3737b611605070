#!/bin/bash
# A/B over the whole training step on ONE box between two environment settings of the same library:
#   BASE_ENV="LDM_GEMM_4W=0 LDM_DX_WT=0" NEW_ENV="" bash scripts/ab_step_env.sh
for i in 1 2 3; do
  for which in base new; do
    if [ $which = base ]; then E="$BASE_ENV"; else E="$NEW_ENV"; fi
    env $E timeout -k 10 200 python bench.py --steps 10 --warmup 3 --no-cpu-baseline ${BENCH_ARGS} > gpurun_out/ab_step_$which.json 2> gpurun_out/ab_step_$which.err || { echo "$which failed"; tail -5 gpurun_out/ab_step_$which.err; exit 1; }
    python3 -c "import json;d=json.load(open('gpurun_out/ab_step_$which.json'));print('$which', round(d['ms_per_step'],3), 'ms/step', round(d['value']), 'tok/s', ' '.join('%s %.2f' % (k['class'], k['ms_per_step']) for k in d.get('roofline_by_kernel', [])))"
  done
done
