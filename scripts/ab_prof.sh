#!/bin/bash
# Per-kernel comparison of library builds INSIDE the training step on ONE box: one kernel-trace
# profile of the bench per build; prints ms/step per kernel, one column per build.
# AB_LIBS="base new" (default): "new" is lib/libldm_hip.so, any other name X is lib/libldm_hip_X.so.
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
AB_LIBS=${AB_LIBS:-"base new"}
for which in $AB_LIBS; do
  if [ $which = new ]; then unset LDM_LIB; else export LDM_LIB=$PWD/language-diffusion-model_amd/lib/libldm_hip_$which.so; fi
  mkdir -p gpurun_out/abprof_$which
  timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/abprof_$which -o run -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline ${BENCH_ARGS} > gpurun_out/abprof_$which.json 2> gpurun_out/abprof_$which.err || { echo "$which failed"; tail -5 gpurun_out/abprof_$which.err; exit 1; }
done
AB_LIBS="$AB_LIBS" python3 - <<'PY'
import csv, os
names = os.environ["AB_LIBS"].split()
def load(w):
    return {r["Name"]: float(r["TotalDurationNs"]) / 7e6 for r in csv.DictReader(open(f"gpurun_out/abprof_{w}/run_kernel_stats.csv"))}
t = [load(w) for w in names]
print("total   " + "  ".join("%s %.2f" % (w, sum(x.values())) for w, x in zip(names, t)) + "  ms/step")
keys = sorted(set().union(*t), key=lambda k: -max(x.get(k, 0) for x in t))[:24]
for k in keys:
    print(" ".join("%8.3f" % x.get(k, 0) for x in t) + "  " + k[:90])
PY
