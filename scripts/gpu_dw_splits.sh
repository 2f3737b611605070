#!/bin/bash
# Weight-gradient GEMM: time and memory-side fetch bytes against the K-split count (does a split that is
# shared by two XCDs cost its panels twice?).
set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=gpurun_out/dw_splits
mkdir -p $OUT
for sp in 0 8 16 24 32 40 48 56 64; do
  echo "== splits $sp"
  LDM_GEMM_SPLITS=$sp T=262144 ONLY=dw REPS=10 timeout -k 10 120 python3 scripts/bench_gemm.py 2>&1 | tee $OUT/time_$sp.txt
done
for sp in 0 8; do
  LDM_GEMM_SPLITS=$sp T=262144 ONLY=dw REPS=2 timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE -d $OUT/pmc_$sp -o f --output-format csv -- python3 scripts/bench_gemm.py > $OUT/pmc_$sp.log 2>&1
  python3 - <<PY
import csv, glob, collections
f = glob.glob("$OUT/pmc_$sp/**/*counter_collection.csv", recursive=True)
acc = collections.OrderedDict()
for row in csv.DictReader(open(f[0])):
    if "gemm256" not in row["Kernel_Name"]: continue
    k = (row["Grid_Size"], row["Kernel_Name"][:60])
    acc.setdefault(row["Dispatch_Id"], 0.0)
    acc[row["Dispatch_Id"]] += float(row["Counter_Value"])
vals = list(acc.values())
print("splits $sp fetch GB per dispatch (x2 corrected):", [round(v * 2 * 1024 / 1e9, 2) for v in vals])
PY
done
