#!/bin/bash
# Check: GPU tests , smoke, bench with and without launch events.
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 1000 python -u -m pytest tests -m gpu -q --tb=short -s --timeout 400 --timeout-method thread -p no:cacheprovider > gpurun_out/tests.log 2>&1
rc=$?; echo "pytest rc=$rc"; grep -E "^(FAILED|ERROR)|passed|failed|^\[" gpurun_out/tests.log | cut -c1-300 | tail -40
[ $rc -ne 0 ] && exit 1
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1 || { echo SMOKE FAILED; tail -5 gpurun_out/smoke.log; exit 1; }
tail -3 gpurun_out/smoke.log
timeout -k 10 600 python bench.py --steps 10 --warmup 3 > gpurun_out/bench.json 2> gpurun_out/bench.err || { echo BENCH FAILED; tail -20 gpurun_out/bench.err; exit 1; }
LDM_BENCH_EVENTS=0 timeout -k 10 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/bench_noev.json 2> gpurun_out/bench_noev.err || { echo BENCH2 FAILED; tail -20 gpurun_out/bench_noev.err; exit 1; }
python - <<'PY'
import json
d=json.load(open("gpurun_out/bench.json")); e=json.load(open("gpurun_out/bench_noev.json"))
print("with events", d["ms_per_step"], "without", e["ms_per_step"], d["value"])
print(json.dumps(d["roofline"]))
for k in d["roofline_by_kernel"]: print(k["class"], round(k["ms_per_step"],3), round(k["achieved"],1), round(k["frac"],3), k["launches_per_step"])
print(d.get("cpu_baseline"))
PY
