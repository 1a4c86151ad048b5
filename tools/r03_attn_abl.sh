#!/bin/bash
# measurement only: rebuild the attention statement with ablations on the GPU box (hipcc is there) and time them
set -u
R=${GRAFT_REPO_ROOT:-$PWD}
cd "$R"
for cfg in "none 3" "nolgk 3" "none 4" "none 5" "novalu 3"; do
  set -- $cfg; abl=$1; lead=$2
  X2I_ATTN_ABL=$abl X2I_ATTN_LEAD=$lead python x2i_amd/csrc/gen_attn_w4.py > /dev/null
  python -m x2i_amd.build > /dev/null 2>&1
  echo "== $abl lead $lead"
  python tools/attn_bench.py 4 2>/dev/null | grep "hand-scheduled" | tail -2
done
X2I_ATTN_ABL=none python x2i_amd/csrc/gen_attn_w4.py > /dev/null
