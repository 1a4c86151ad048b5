#!/bin/bash
set -u
TAG=${1:-r03f}
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/$TAG
mkdir -p "$OUT"
cd "$R"
timeout 600 python tools/eager_gpu_baseline.py 1 --blocks 1 1 > "$OUT/${TAG}_eager_1_1.log" 2>&1
tail -5 "$OUT/${TAG}_eager_1_1.log"
timeout 900 python tools/eager_gpu_baseline.py 4 > "$OUT/${TAG}_eager_gpu_baseline.log" 2>&1
tail -5 "$OUT/${TAG}_eager_gpu_baseline.log"
timeout 900 python bench.py > "$OUT/${TAG}_bench_b4_1024.json.log" 2>&1
tail -1 "$OUT/${TAG}_bench_b4_1024.json.log" | python3 -c "
import sys, json
d = json.loads(sys.stdin.read())
print({k: d[k] for k in ('value', 'ms_per_denoise_step')}, d['roofline']['frac'], d.get('fp8_mlp'), d.get('fp8_all'), d.get('cpu_baseline'))
"
