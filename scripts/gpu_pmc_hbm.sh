#!/bin/bash
# Memory-side traffic of the HBM-bound kernels: one rocprofv3 --pmc pass per counter (FETCH_SIZE,
# WRITE_SIZE) over scripts/hbm_kernels_bench.py, averaged per kernel and set beside the algorithmic
# bytes of gpurun_out/hbm_kernels.json -> gpurun_out/pmc_hbm/hbm_traffic.json.
# FETCH_SIZE is doubled (gfx950 tallies 128-B read requests at 64 B); both counters sit on the L2's
# memory side, so Infinity-Cache hits are included.
set -o pipefail
mkdir -p gpurun_out/pmc_hbm
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -s KILL 200 rocprofv3 --pmc $c --output-format csv -d gpurun_out/pmc_hbm -o $c -- python3 scripts/hbm_kernels_bench.py > gpurun_out/pmc_hbm/$c.log 2>&1 || { echo "pass $c failed"; tail -5 gpurun_out/pmc_hbm/$c.log; exit 1; }
done
python3 - <<'PY'
import csv, glob, json, re, collections
agg = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob(f"gpurun_out/pmc_hbm/**/{c}_counter_collection.csv", recursive=True)[0]
    d = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        k = re.sub(r"\(anonymous namespace\)::|void ", "", r["Kernel_Name"]).split("(")[0]
        d[k].append(float(r["Counter_Value"]))
    agg[c] = d
alg = {e["kernel"]: e for e in json.load(open("gpurun_out/hbm_kernels.json"))["kernels"]}
want = {"rmsnorm_fwd8_k": "rmsnorm_fwd", "rmsnorm_bwd8_k": "rmsnorm_bwd (+residual grad)", "rope_qk8_k": "rope_qk (in place)",
        "swiglu_fwd8_k": "swiglu_fwd", "swiglu_bwd8_k": "swiglu_bwd", "embed_fwd_k": "embed_fwd", "embed_bwd_k": "embed_bwd",
        "ce_fwd_walk8_k": "ce_fwd", "ce_bwd_compact8_k": "ce_bwd_compact", "ce_fwd_grad_k": "ce_fwd_grad (one pass)"}
out = []
for kname, label in want.items():
    fk = [k for k in agg["FETCH_SIZE"] if k.startswith(kname)]
    if not fk or label not in alg:
        continue
    k = fk[0]
    # the timed launches dominate the count; the median is one steady launch
    med = lambda v: sorted(v)[len(v) // 2]
    fetch = 2.0 * med(agg["FETCH_SIZE"][k]) * 1024.0
    write = med(agg["WRITE_SIZE"].get(k, [0.0])) * 1024.0
    a = alg[label]["bytes"]
    out.append({"kernel": k, "algorithmic_bytes": a, "fetch_bytes": fetch, "write_bytes": write, "traffic_over_algorithmic": (fetch + write) / a})
    print(f"{k[:40]:40s} algorithmic {a/1e6:8.1f} MB  fetch {fetch/1e6:8.1f}  write {write/1e6:8.1f}  ratio {(fetch+write)/a:5.2f}")
json.dump({"unit": "bytes per launch (median launch); FETCH_SIZE doubled as gfx950 requires; Infinity-Cache hits included",
           "command": "rocprofv3 --pmc FETCH_SIZE | WRITE_SIZE -- python3 scripts/hbm_kernels_bench.py (scripts/gpu_pmc_hbm.sh)",
           "kernels": out}, open("gpurun_out/pmc_hbm/hbm_traffic.json", "w"), indent=1)
PY
