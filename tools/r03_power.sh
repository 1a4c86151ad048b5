#!/bin/bash
# package power while each MFMA shape runs alone (long runs: the reading lags by about a second)
cd ${GRAFT_REPO_ROOT:-$PWD}
mkdir -p gpurun_out/r03p
for v in 0 1; do
  bash tools/clock_watch.sh gpurun_out/r03p/power_v$v.log -- tools/ubench/bin/mfma_power 1 $v 40 | tail -1
  echo "variant $v:"; sed 's/=*//g; s/GPU\[0\]//g; s/\t//g' gpurun_out/r03p/power_v$v.log | cut -c1-140 | awk 'NR%2==0' | head -9
done
