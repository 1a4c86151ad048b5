#!/bin/bash
# Tools-only (measurement, wrong results by design): libx2i_hip_w16_<abl>.so for item-boundary ablations of gen_attn_w16.py -- noepi (no epilogue), noprold (no Q loads /
# first K tiles in the prologue), noepi+noprold; tools/attn_item_cost.py runs against them with X2I_LIB_VARIANT=w16_<abl>.  The product .inc is restored afterwards.
set -e
cd "$(dirname "$0")/.."
OBJS=$(ls x2i_amd/_build/*.hip.o | grep -v "\.abl\.o" | grep -v attention_w16)
for abl in "$@"; do
  (cd x2i_amd/csrc && X2I_ATTN_ABL=$abl python gen_attn_w16.py > /dev/null)
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-value -ffp-contract=fast -c x2i_amd/csrc/attention_w16.hip -o /tmp/attention_w16_$abl.o 2>/dev/null
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "x2i_amd/libx2i_hip_w16_${abl//+/_}.so" $OBJS /tmp/attention_w16_$abl.o
  echo "built x2i_amd/libx2i_hip_w16_${abl//+/_}.so"
done
(cd x2i_amd/csrc && X2I_ATTN_ABL= python gen_attn_w16.py > /dev/null)
