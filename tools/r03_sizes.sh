#!/bin/bash
# bench lines at the other image sizes / batches (regression check of the round-3 kernels away from the headline shape)
cd ${GRAFT_REPO_ROOT:-$PWD}
for cfg in "512 1" "512 4" "512 16" "768 4" "1024 1" "1024 16" "1536 1"; do
  set -- $cfg
  timeout 600 python bench.py --size $1 --batch $2 --no-cpu-baseline --no-fp8-lines 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('size $1 batch $2:', round(d['value'],3), 'images/s', round(d['ms_per_denoise_step'],2), 'ms/step', round(d.get('model_tflops_per_gpu',0)), 'TF')"
done
