#!/bin/bash
# Same-box comparison of several builds of the library over the whole training step: LIBS = names under
# language-diffusion-model_amd/lib/ (without the libldm_hip prefix / .so suffix; "" = the product build).
for i in 1 2; do
  for which in ${LIBS}; do
    if [ "$which" = cur ]; then unset LDM_LIB; else export LDM_LIB=$PWD/language-diffusion-model_amd/lib/libldm_hip_$which.so; fi
    timeout -k 10 200 python bench.py --steps 10 --warmup 3 --no-cpu-baseline ${BENCH_ARGS} > gpurun_out/ab_libs_$which.json 2> gpurun_out/ab_libs_$which.err || { echo "$which failed"; tail -5 gpurun_out/ab_libs_$which.err; exit 1; }
    python3 -c "import json;d=json.load(open('gpurun_out/ab_libs_$which.json'));print('%-8s' % '$which', round(d['ms_per_step'],3), 'ms/step', ' '.join('%s %.2f' % (k['class'], k['ms_per_step']) for k in d.get('roofline_by_kernel', [])))"
  done
done
