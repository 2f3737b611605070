#!/bin/bash
# Profiles (named r2_*) of the bench command at its default shape (per-GPU batch 512, seq 512, bf16):
#   1. rocprofv3 --kernel-trace --stats                      -> gpurun_out/r2_prof/kernel_stats.csv
#   2. one --pmc pass each for FETCH_SIZE and WRITE_SIZE       -> gpurun_out/r2_prof/traffic.json  (per kernel, per launch)
#   3. one --pmc pass for MFMA-busy and wait counters          -> gpurun_out/r2_prof/mfma_util.json
# Counter passes run on their own (no trace domains next to --pmc).
set -o pipefail
OUT=gpurun_out/r2_prof
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o run -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline > $OUT/trace_bench.json 2> $OUT/trace.err || { echo "trace failed"; tail -5 $OUT/trace.err; exit 1; }
cp $(find $OUT/trace -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats.csv && echo "kernel stats ok"
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -s KILL 400 rocprofv3 --pmc $c --output-format csv -d $OUT/pmc_$c -o $c -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > $OUT/$c.log 2>&1 || { echo "pass $c failed"; tail -5 $OUT/$c.log; exit 1; }
  echo "pmc $c ok"
done
timeout -s KILL 400 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_mfma -o mfma -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > $OUT/mfma.log 2>&1 || { echo "mfma pass failed"; tail -5 $OUT/mfma.log; exit 1; }
echo "pmc mfma ok"
python3 - <<'PY'
import csv, glob, json, collections, re
OUT = "gpurun_out/r2_prof"
def short(k):
    return re.sub(r"\(anonymous namespace\)::|void ", "", k).split("(")[0]
# ---- traffic
per = collections.defaultdict(lambda: {"fetch_kib_raw": 0.0, "write_kib": 0.0, "dispatches": set()})
for c, key in (("FETCH_SIZE", "fetch_kib_raw"), ("WRITE_SIZE", "write_kib")):
    f = glob.glob(f"{OUT}/pmc_{c}/**/{c}_counter_collection.csv", recursive=True)[0]
    for r in csv.DictReader(open(f)):
        k = short(r["Kernel_Name"])
        per[k][key] += float(r["Counter_Value"])
        per[k]["dispatches"].add(r["Dispatch_Id"])
FETCH_CORR = 2.0   # gfx950 tallies 128-byte read requests at 64 B
kern = []
for k, d in per.items():
    n = len(d["dispatches"])
    kern.append({"kernel": k, "dispatches": n, "fetch_kib_raw": d["fetch_kib_raw"], "write_kib": d["write_kib"],
                 "bytes_per_launch": (d["fetch_kib_raw"] * FETCH_CORR + d["write_kib"]) * 1024.0 / max(n, 1)})
kern.sort(key=lambda e: -(e["fetch_kib_raw"] * FETCH_CORR + e["write_kib"]))
g = [e for e in kern if e["kernel"].startswith("gemm256_k")]
gn = sum(e["dispatches"] for e in g)
gb = sum((e["fetch_kib_raw"] * FETCH_CORR + e["write_kib"]) * 1024.0 for e in g)
json.dump({"config": "default", "dtype": "bf16", "batch": 512, "seq": 512,
           "command": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (one pass each) -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline (scripts/gpu_r2_profile.sh)",
           "unit": "KiB summed over the kernel's dispatches; FETCH_SIZE doubled before use (gfx950 tallies 128-B read requests at 64 B); counters sit on the L2's memory side (Infinity-Cache hits included)",
           "fetch_correction": FETCH_CORR, "gemm_dispatches": gn, "gemm_bytes_per_launch": gb / max(gn, 1),
           "kernels": kern[:40]}, open(f"{OUT}/traffic.json", "w"), indent=1)
print("gemm256_k: dispatches", gn, "bytes per launch %.1f MB" % (gb / max(gn, 1) / 1e6))
for e in kern[:14]:
    print(f"{e['kernel'][:64]:64s} n={e['dispatches']:4d} {e['bytes_per_launch'] / 1e6:9.1f} MB/launch")
# ---- mfma util
f = glob.glob(f"{OUT}/pmc_mfma/**/mfma_counter_collection.csv", recursive=True)[0]
agg = collections.defaultdict(lambda: collections.defaultdict(float))
disp = collections.defaultdict(set)
for r in csv.DictReader(open(f)):
    k = short(r["Kernel_Name"])
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
    disp[k].add(r["Dispatch_Id"])
out = []
for k, d in agg.items():
    if not re.search(r"gemm|flash", k):
        continue
    gui = d.get("GRBM_GUI_ACTIVE", 0.0)
    wc = max(d.get("SQ_WAVE_CYCLES", 0.0), 1.0)
    out.append({"kernel": k, "dispatches": len(disp[k]),
                "mfma_util": d.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / max(gui / 8 * 256 * 4, 1.0),
                "wait_any_per_wave_cycle": d.get("SQ_WAIT_ANY", 0.0) / wc, "counters": dict(d)})
out.sort(key=lambda e: -e["counters"].get("GRBM_GUI_ACTIVE", 0))
for e in out:
    print(f"{e['kernel'][:64]:64s} n={e['dispatches']:4d} mfma_util={e['mfma_util']:.3f} wait_any={e['wait_any_per_wave_cycle']:.2f}")
json.dump({"command": "rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline",
           "formula": "mfma_util = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 * 256 * 4), summed over the kernel's dispatches",
           "kernels": out}, open(f"{OUT}/mfma_util.json", "w"), indent=1)
PY
