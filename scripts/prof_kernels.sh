#!/bin/bash
# kernel-trace stats of a short bench run; prints the kernels matching $1
set -o pipefail
mkdir -p gpurun_out/prof
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof -o run -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline ${BENCH_ARGS} > gpurun_out/prof_bench.json 2> gpurun_out/prof.err || { tail -5 gpurun_out/prof.err; exit 1; }
python - "$1" <<'PY'
import csv, sys, re
rows=list(csv.DictReader(open("gpurun_out/prof/run_kernel_stats.csv")))
tot=sum(float(r["TotalDurationNs"]) for r in rows)
print("total kernel ms per step:", tot/7e6)
for r in rows:
    if re.search(sys.argv[1], r["Name"]):
        print("%8.3f ms/step  %5s calls  avg %9.1f us  %s" % (float(r["TotalDurationNs"])/7e6, r["Calls"], float(r["AverageNs"])/1e3, r["Name"][:100]))
PY
