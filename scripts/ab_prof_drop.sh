#!/bin/bash
# Same-box kernel-trace comparison of two library builds at dropout 0.1 (attention kernels with the mask hash)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for which in base new; do
  if [ $which = base ]; then export LDM_LIB=$PWD/language-diffusion-model_amd/lib/libldm_hip_base.so; else unset LDM_LIB; fi
  mkdir -p gpurun_out/abd_$which
  timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/abd_$which -o run -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --dropout 0.1 > gpurun_out/abd_$which/bench.json 2> gpurun_out/abd_$which/err.txt || { echo "$which failed"; tail -5 gpurun_out/abd_$which/err.txt; exit 1; }
  echo "== $which"
  python3 - <<PY
import csv
rows=list(csv.DictReader(open("gpurun_out/abd_$which/run_kernel_stats.csv")))
print("total", sum(float(r["TotalDurationNs"]) for r in rows)/7e6)
for r in rows:
    if "flash" in r["Name"] or "dropout8" in r["Name"]:
        print("%8.3f ms/step avg %9.1f us  %s" % (float(r["TotalDurationNs"])/7e6, float(r["AverageNs"])/1e3, r["Name"][:90]))
PY
done
