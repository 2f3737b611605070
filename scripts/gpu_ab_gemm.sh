#!/bin/bash
# GEMM-related kernel tests against the current library build, then the same-box A/B of the step's GEMM shapes
set -o pipefail
timeout -k 10 400 python -u -m pytest tests/test_gpu_kernels.py -m gpu -q -x --tb=short --timeout 200 --timeout-method thread -p no:cacheprovider -k "gemm or linear or swiglu or head" > gpurun_out/t_exp.log 2>&1 || { tail -30 gpurun_out/t_exp.log; exit 1; }
tail -1 gpurun_out/t_exp.log
T=${T:-262144} REPS=10 bash scripts/ab_gemm.sh > gpurun_out/ab_gemm.log 2>&1 || { tail gpurun_out/ab_gemm.log; exit 1; }
python - <<'PY'
import re, collections
cur=None; d=collections.defaultdict(lambda: {"BASE":[], "EXP":[]})
for l in open("gpurun_out/ab_gemm.log"):
    l=l.strip()
    if l in ("BASE","EXP"): cur=l; continue
    m=re.match(r"(\w+)\s+M=\s*(\d+) N=\s*(\d+) K=\s*(\d+)\s+([\d.]+) us", l)
    if m: d[(m[1],m[2],m[3],m[4])][cur].append(float(m[5]))
tb=te=0
for k,v in d.items():
    b=sum(v["BASE"])/len(v["BASE"]); e=sum(v["EXP"])/len(v["EXP"]); 
    if k[2]!="30528" and k[1]!="30528" and k[3]!="30528": tb+=b; te+=e
    print("%-4s M=%7s N=%6s K=%7s base %9.1f exp %9.1f  %+5.1f%%" % (*k,b,e,(e/b-1)*100))
print("layer shapes total: base %.1f exp %.1f %+.1f%%" % (tb,te,(te/tb-1)*100))
PY
