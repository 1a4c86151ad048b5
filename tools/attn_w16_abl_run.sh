#!/bin/bash
# Tools-only: run tools/attn_w16_clock.py against every ablation library built by tools/attn_w16_sweep_build.sh (measurement only).
cd "$(dirname "$0")/.."
echo "== product"; python tools/attn_w16_clock.py 1
for so in x2i_amd/libx2i_hip_w16_L*V*_*.so; do
  v=$(basename $so .so); v=${v#libx2i_hip_}
  echo "== $v"; X2I_LIB_VARIANT=$v timeout 300 python tools/attn_w16_clock.py 1 w16only
done
echo "== product"; python tools/attn_w16_clock.py 1
