#!/bin/bash
# Kernel-trace profile of the bench command; summaries are copied to gpurun_out/prof_*.
set -o pipefail
mkdir -p gpurun_out/prof
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout -k 10 900 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof -o run -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/prof_bench.json 2> gpurun_out/prof.err
echo "rc=$?"
find gpurun_out/prof -name "*stats*" | head
f=$(find gpurun_out/prof -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && head -40 "$f" | cut -c1-220
