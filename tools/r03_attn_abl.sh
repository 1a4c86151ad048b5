#!/bin/bash
# measurement only: rebuild the attention statement with generator switches on the GPU box (hipcc is there) and time them
set -u
R=${GRAFT_REPO_ROOT:-$PWD}
cd "$R"
timeout 300 python -m pytest tests/test_ops_gpu.py -x -q -k "attention" 2>&1 | grep -v amdgpu.ids | tail -3
CFGS=${ABL_CFGS:-none:4:2 none:3:4 none:5:4 novalu:4:4 nosync:4:4 nobar:4:4}
for cfg in $CFGS; do
  IFS=: read abl lead vd <<< "$cfg"
  X2I_ATTN_ABL=$abl X2I_ATTN_LEAD=$lead X2I_ATTN_VDELAY=$vd python x2i_amd/csrc/gen_attn_w4.py > /dev/null
  python -m x2i_amd.build > /dev/null 2>&1
  echo "== $abl lead $lead vdelay $vd"
  [ "$abl" = none ] && timeout 300 python -m pytest tests/test_ops_gpu.py -x -q -k "attention" 2>&1 | grep -v amdgpu.ids | tail -1
  python tools/attn_bench.py 4 2>/dev/null | grep "hand-scheduled\|ping-pong" | tail -3
done
X2I_ATTN_ABL= python x2i_amd/csrc/gen_attn_w4.py > /dev/null
