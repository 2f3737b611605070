#!/bin/bash
# One GPU call: tests, smoke, bench, kernel-trace profile.  Each GPU step has its own time limit.
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -u -m pytest tests -m gpu -q --tb=line --timeout 120 --timeout-method thread -p no:cacheprovider > gpurun_out/tests.log 2>&1
echo "pytest rc=$?"; tail -4 gpurun_out/tests.log | cut -c1-200
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1 || { echo SMOKE FAILED; tail -5 gpurun_out/smoke.log; exit 1; }
tail -3 gpurun_out/smoke.log
timeout -k 10 600 python bench.py --steps 10 --warmup 3 > gpurun_out/bench.json 2> gpurun_out/bench.err || { echo BENCH FAILED; tail -20 gpurun_out/bench.err; exit 1; }
cat gpurun_out/bench.json
