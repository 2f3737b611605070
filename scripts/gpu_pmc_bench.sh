#!/bin/bash
# HBM traffic of the dominant kernel (LM-head forward GEMM) from the memory-side counters:
# one rocprofv3 --pmc pass per counter over a short bench run; output gpurun_out/pmc_bench/traffic.json
set -o pipefail
mkdir -p gpurun_out/pmc_bench
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -s KILL 300 rocprofv3 --pmc $c --output-format csv -d gpurun_out/pmc_bench -o $c -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/pmc_bench/$c.log 2>&1 || { echo "pass $c failed"; tail -5 gpurun_out/pmc_bench/$c.log; exit 1; }
done
python3 - <<'PY'
import csv, glob, json
out = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob(f"gpurun_out/pmc_bench/**/{c}_counter_collection.csv", recursive=True)[0]
    rows = [r for r in csv.DictReader(open(f)) if "gemm256_k<true, true, 256, unsigned short, 0>" in r["Kernel_Name"]]
    # the LM-head launches are the ones with the largest counter value of this kernel (one per step)
    vals = sorted((float(r["Counter_Value"]) for r in rows), reverse=True)[:4]
    out[c] = {"per_launch_raw": vals, "n_dispatches": len(rows)}
print(json.dumps(out))
json.dump(out, open("gpurun_out/pmc_bench/traffic.json", "w"))
PY
