#!/bin/bash
# Whole-step comparison of several library builds on one box: LIBS="a.so b.so" (names under lib/), alternating
for i in 1 2 3; do
  for lib in $LIBS; do
    LDM_LIB=$PWD/language-diffusion-model_amd/lib/$lib timeout -k 10 200 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/abl.json 2> gpurun_out/abl.err || { echo "$lib failed"; tail -5 gpurun_out/abl.err; exit 1; }
    python3 -c "import json;d=json.load(open('gpurun_out/abl.json'));print('%-22s' % '$lib', round(d['ms_per_step'],3), ' '.join('%s %.2f' % (k['class'], k['ms_per_step']) for k in d.get('roofline_by_kernel', [])))"
  done
done
