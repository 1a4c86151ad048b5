#!/bin/bash
# measurement only: rebuild the attention statement with ablations on the GPU box (hipcc is there) and time them
set -u
R=${GRAFT_REPO_ROOT:-$PWD}
cd "$R"
timeout 300 python -m pytest tests/test_ops_gpu.py -x -q -s -k "attention" 2>&1 | grep -v amdgpu.ids | tail -8
for cfg in ${ABL_CFGS:-"none:4 none:3 none:5 novalu:4 nosync:4"}; do
  abl=${cfg%%:*}; lead=${cfg##*:}
  X2I_ATTN_ABL=$abl X2I_ATTN_LEAD=$lead python x2i_amd/csrc/gen_attn_w4.py > /dev/null
  python -m x2i_amd.build > /dev/null 2>&1
  echo "== $abl lead $lead"
  python tools/attn_bench.py 4 2>/dev/null | grep "hand-scheduled" | tail -2
done
X2I_ATTN_ABL= python x2i_amd/csrc/gen_attn_w4.py > /dev/null
