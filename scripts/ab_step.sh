#!/bin/bash
# A/B of two builds of the library on ONE box over the whole training step:
# lib/libldm_hip_base.so (LDM_LIB) vs lib/libldm_hip.so, alternating, ms/step of each run printed.
for i in 1 2 3; do
  for which in base new; do
    if [ $which = base ]; then export LDM_LIB=$PWD/language-diffusion-model_amd/lib/libldm_hip_base.so; else unset LDM_LIB; fi
    timeout -k 10 200 python bench.py --steps 10 --warmup 3 --no-cpu-baseline ${BENCH_ARGS} > gpurun_out/ab_step_$which.json 2> gpurun_out/ab_step_$which.err || { echo "$which failed"; tail -5 gpurun_out/ab_step_$which.err; exit 1; }
    python3 -c "import json;d=json.load(open('gpurun_out/ab_step_$which.json'));print('$which', round(d['ms_per_step'],3), 'ms/step', round(d['value']), 'tok/s', ' '.join('%s %.2f' % (k['class'], k['ms_per_step']) for k in d.get('roofline_by_kernel', [])))"
  done
done
