#!/bin/bash
# MFMA utilisation of the GEMM and attention kernels inside the training step: one rocprofv3 --pmc
# pass (5 SQ counters + GRBM_GUI_ACTIVE) over a short bench run, summarised per kernel into
# gpurun_out/pmc_mfma/mfma_util.json.
#   util = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs * 256 CUs * 4 SIMDs)
set -o pipefail
mkdir -p gpurun_out/pmc_mfma
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout -s KILL 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/pmc_mfma -o mfma -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline ${BENCH_ARGS} > gpurun_out/pmc_mfma/run.log 2>&1 || { echo "pmc pass failed"; tail -5 gpurun_out/pmc_mfma/run.log; exit 1; }
python3 - <<'PY'
import csv, glob, json, collections, re
f = glob.glob("gpurun_out/pmc_mfma/**/mfma_counter_collection.csv", recursive=True)[0]
agg = collections.defaultdict(lambda: collections.defaultdict(float))
disp = collections.defaultdict(set)
for r in csv.DictReader(open(f)):
    k = re.sub(r"\(anonymous namespace\)::|void ", "", r["Kernel_Name"])
    k = k.split("(")[0]
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
    disp[k].add(r["Dispatch_Id"])
out = []
for k, d in agg.items():
    if not re.search(r"gemm|flash", k):
        continue
    gui = d.get("GRBM_GUI_ACTIVE", 0.0)
    util = d.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / max(gui / 8 * 256 * 4, 1.0)
    wc = max(d.get("SQ_WAVE_CYCLES", 0.0), 1.0)
    out.append({"kernel": k, "dispatches": len(disp[k]), "mfma_util": util,
                "wait_any_per_wave_cycle": d.get("SQ_WAIT_ANY", 0.0) / wc,
                "wait_inst_per_wave_cycle": d.get("SQ_WAIT_INST_ANY", 0.0) / wc,
                "counters": dict(d)})
out.sort(key=lambda e: -e["counters"].get("GRBM_GUI_ACTIVE", 0))
for e in out:
    print(f"{e['kernel'][:70]:70s} n={e['dispatches']:4d} mfma_util={e['mfma_util']:.3f} wait_any={e['wait_any_per_wave_cycle']:.2f}")
json.dump({"command": "rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline",
           "formula": "mfma_util = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 * 256 * 4), summed over the kernel's dispatches",
           "kernels": out}, open("gpurun_out/pmc_mfma/mfma_util.json", "w"), indent=1)
PY
