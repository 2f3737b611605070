#!/bin/bash
# One GPU call: tests, smoke, bench, kernel-trace profile.  Each GPU step has its own time limit.
set -o pipefail
mkdir -p gpurun_out/prof
timeout -k 10 600 python -u -m pytest tests -m gpu -q --tb=line --timeout 120 --timeout-method thread -p no:cacheprovider > gpurun_out/tests.log 2>&1
rc=$?; echo "pytest rc=$rc"; grep -E "^(FAILED|ERROR)|passed|failed" gpurun_out/tests.log | cut -c1-300 | tail -15
[ $rc -ne 0 ] && exit 1
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1 || { echo SMOKE FAILED; tail -5 gpurun_out/smoke.log; exit 1; }
tail -3 gpurun_out/smoke.log
timeout -k 10 600 python bench.py --steps 10 --warmup 3 ${BENCH_ARGS} > gpurun_out/bench.json 2> gpurun_out/bench.err || { echo BENCH FAILED; tail -20 gpurun_out/bench.err; exit 1; }
python - <<'PY'
import json
d=json.load(open("gpurun_out/bench.json"))
print({k:d[k] for k in ("value","ms_per_step","mfma_fraction_of_step")}, d["roofline"]["achieved"], d.get("cpu_baseline",{}).get("value"))
PY
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof -o run -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline ${BENCH_ARGS} > gpurun_out/prof_bench.json 2> gpurun_out/prof.err
python - <<'PY'
import csv
rows=list(csv.DictReader(open("gpurun_out/prof/run_kernel_stats.csv")))
tot=sum(float(r["TotalDurationNs"]) for r in rows)
print("total kernel ms per step (7 steps):", tot/7e6)
for r in rows[:22]:
    print("%8.3f ms/step  %5s calls  avg %9.1f us  %s" % (float(r["TotalDurationNs"])/7e6, r["Calls"], float(r["AverageNs"])/1e3, r["Name"][:110]))
PY
