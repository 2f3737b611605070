#!/bin/bash
# Issue/stall counters of the attention kernels alone (scripts/attn_kernels_bench.py at SHAPE): three
# rocprofv3 --pmc passes of 8 SQ counters each, summarised per kernel into gpurun_out/pmc_attn/summary.json.
set -o pipefail
export SHAPE=${SHAPE:-512,512,8,96}
mkdir -p gpurun_out/pmc_attn
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
P1="SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_ANY"
P2="SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM SQ_INSTS_VALU_TRANS_F32 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"
P3="SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL SQ_VALU_MFMA_COEXEC_CYCLES"
i=0
for P in "$P1" "$P2" "$P3"; do
  i=$((i+1))
  timeout -s KILL 120 rocprofv3 --pmc $P --output-format csv -d gpurun_out/pmc_attn/p$i -o p -- python3 scripts/attn_kernels_bench.py > gpurun_out/pmc_attn/run$i.log 2>&1 || { echo "pass $i failed"; tail -5 gpurun_out/pmc_attn/run$i.log; exit 1; }
done
python3 - <<'PY'
import csv, glob, json, collections, re
agg = collections.defaultdict(lambda: collections.defaultdict(float))
n = collections.defaultdict(set)
for f in glob.glob("gpurun_out/pmc_attn/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = re.sub(r"\(anonymous namespace\)::|void ", "", r["Kernel_Name"]).split("(")[0]
        if "flash" not in k:
            continue
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        n[(k, f)].add(r["Dispatch_Id"])
out = {}
for k, d in agg.items():
    wc = d["SQ_WAVE_CYCLES"]
    out[k] = {c: v for c, v in d.items()}
    out[k]["per_wave_cycle"] = {c: round(v / wc, 4) for c, v in d.items() if c.startswith(("SQ_ACTIVE", "SQ_WAIT", "SQ_VMEM", "SQ_LDS_CMD", "SQ_LDS_DATA"))}
    print(k)
    for c in sorted(d):
        print(f"   {c:34s} {d[c]:16.0f}  /wave_cycle {d[c] / wc:8.4f}")
json.dump(out, open("gpurun_out/pmc_attn/summary.json", "w"), indent=1)
PY
