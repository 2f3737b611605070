#!/bin/bash
# Bench records of the other single-GPU configurations (~1B at seq 512 and 4096, the 8B layer shape at 2 layers)
set -o pipefail
O=gpurun_out/cfg
mkdir -p $O
timeout -k 10 300 python bench.py --config 1b --seq 512 --batch 128 --steps 5 --warmup 2 --no-cpu-baseline > $O/1b_s512.json 2> $O/1b_s512.err || { tail -5 $O/1b_s512.err; exit 1; }
timeout -k 10 300 python bench.py --config 1b --seq 4096 --batch 16 --steps 5 --warmup 2 --no-cpu-baseline > $O/1b_s4096.json 2> $O/1b_s4096.err || { tail -5 $O/1b_s4096.err; exit 1; }
timeout -k 10 300 python bench.py --config 8b-2layer --seq 4096 --batch 8 --steps 5 --warmup 2 --no-cpu-baseline > $O/8b2_s4096.json 2> $O/8b2_s4096.err || { tail -5 $O/8b2_s4096.err; exit 1; }
python3 - <<'PY'
import json
for f in ("1b_s512", "1b_s4096", "8b2_s4096"):
    d = json.load(open(f"gpurun_out/cfg/{f}.json"))
    print(f, round(d["ms_per_step"], 1), "ms/step", round(d["value"]), "tok/s", "mfma", round(d["mfma_fraction_of_step"], 3), d["config"]["workload"])
PY
