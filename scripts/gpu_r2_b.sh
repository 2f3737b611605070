#!/bin/bash
# Targeted GPU check: the named tests, then the bench.
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -u -m pytest $TESTS -m gpu -q --tb=short -s --timeout 300 --timeout-method thread -p no:cacheprovider > gpurun_out/tests_b.log 2>&1
rc=$?; echo "pytest rc=$rc"; grep -E "^(FAILED|ERROR)|passed|failed|^E  " gpurun_out/tests_b.log | cut -c1-300 | tail -30
[ $rc -ne 0 ] && exit 1
timeout -k 10 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline $BENCH_ARGS > gpurun_out/bench_b.json 2> gpurun_out/bench_b.err || { echo BENCH FAILED; tail -20 gpurun_out/bench_b.err; exit 1; }
python - <<'PY'
import json
d=json.load(open("gpurun_out/bench_b.json"))
print("ms/step", d["ms_per_step"], "tok/s", d["value"], "gemm frac", d["roofline"]["frac"])
for k in d["roofline_by_kernel"]: print(k["class"], round(k["ms_per_step"],3), round(k["achieved"],1), round(k["frac"],3), k["launches_per_step"])
PY
