#!/bin/bash
# In-step sweep of the GEMM band height per call site (LDM_GEMM_BAND_MAP), one box; prints ms/step and the GEMM classes
run() {
  LDM_GEMM_BAND_MAP="$1" timeout -k 10 200 python bench.py --steps 8 --warmup 3 --no-cpu-baseline > gpurun_out/abs.json 2> gpurun_out/abs.err || { echo failed; tail -3 gpurun_out/abs.err; exit 1; }
  python3 -c "import json;d=json.load(open('gpurun_out/abs.json'));print('%-28s' % '$1', round(d['ms_per_step'],2), ' '.join('%s %.2f' % (k['class'], k['ms_per_step']) for k in d['roofline_by_kernel'][:5]))"
}
run ""
for site in "4:0:768:0" "3:0:768:0" "1:0:768:0" "1:0:1024:0" "2:0:768:0" "0:0:768:768" "0:0:2304:768" "0:0:2048:768" "0:0:768:30528" "0:0:30528:768"; do
  for b in 2 4 8 16 32; do run "$site=$b"; done
done
run ""
