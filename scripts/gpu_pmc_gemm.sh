#!/bin/bash
# PMC pass over the GEMM micro-benchmark (forward shapes): SQ wait/busy breakdown and LDS conflicts.
set -o pipefail
mkdir -p gpurun_out/pmc
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
export ONLY=${ONLY:-fwd} REPS=3 T=${T:-32768}
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d gpurun_out/pmc -o sq -- python3 scripts/bench_gemm.py > gpurun_out/pmc/sq.log 2>&1
echo "rc=$?"
timeout -s KILL 120 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum --output-format csv -d gpurun_out/pmc -o tcc -- python3 scripts/bench_gemm.py > gpurun_out/pmc/tcc.log 2>&1
echo "rc=$?"
python3 - <<'PY'
import csv, glob, collections
for tag in ("tcc",):
    f = glob.glob(f"gpurun_out/pmc/**/{tag}_counter_collection.csv", recursive=True)
    if not f: print("no", tag); continue
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
    for r in csv.DictReader(open(f[0])):
        k = r["Kernel_Name"][:60]
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[k] += 1
    for k, d in agg.items():
        if "gemm" not in k: continue
        print(k, " ".join(f"{a}={b:.4g}" for a, b in sorted(d.items())), f"hit_rate={d['TCC_HIT_sum']/max(d['TCC_HIT_sum']+d['TCC_MISS_sum'],1):.3f}")
PY
python3 - <<'PY'
import csv, glob, collections
f = glob.glob("gpurun_out/pmc/**/sq_counter_collection.csv", recursive=True)
if not f:
    print("no counter file", glob.glob("gpurun_out/pmc/**/*", recursive=True)[:10]); raise SystemExit
agg = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.Counter()
for r in csv.DictReader(open(f[0])):
    k = r["Kernel_Name"][:60] + " grid=" + r.get("Grid_Size", "")
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
    cnt[k] += 1
for k, d in agg.items():
    if "gemm" not in k: continue
    wc = d.get("SQ_WAVE_CYCLES", 1)
    print(k)
    print("   " + "  ".join(f"{n}={v:.3g}" for n, v in sorted(d.items())))
    print(f"   wait_any/wave={d.get('SQ_WAIT_ANY',0)/wc:.2f} wait_inst/wave={d.get('SQ_WAIT_INST_ANY',0)/wc:.2f} active/wave={d.get('SQ_ACTIVE_INST_ANY',0)/wc:.2f}"
          f" mfma_busy/busy={d.get('SQ_VALU_MFMA_BUSY_CYCLES',0)/max(d.get('SQ_BUSY_CYCLES',1),1):.3f} lds_conflict/lds_active={d.get('SQ_LDS_BANK_CONFLICT',0)/max(d.get('SQ_LDS_IDX_ACTIVE',1),1):.3f}")
PY
