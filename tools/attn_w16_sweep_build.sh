#!/bin/bash
# Tools-only: build libx2i_hip_w16_L<lead>V<vdelay>.so for a list of "lead:vdelay" schedule parameters of gen_attn_w16.py (the product .inc is
# restored afterwards); tools/attn_w16_clock.py runs against them with X2I_LIB_VARIANT=w16_L<lead>V<vdelay>.  With X2I_ATTN_ABL=<x> in the
# environment (novalu / nolgk / nosync / nobar: measurement only, wrong results) the library is libx2i_hip_w16_L<lead>V<vdelay>_<x>.so.
set -e
cd "$(dirname "$0")/.."
OBJS=$(ls x2i_amd/_build/*.hip.o | grep -v "\.abl\.o" | grep -v attention_w16)
for lv in "$@"; do
  L=${lv%%:*}; V=${lv##*:}
  SFX=${X2I_ATTN_ABL:+_$X2I_ATTN_ABL}
  (cd x2i_amd/csrc && X2I_ATTN_LEAD=$L X2I_ATTN_VDELAY=$V python gen_attn_w16.py > /dev/null)
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-value -ffp-contract=fast -c x2i_amd/csrc/attention_w16.hip -o /tmp/attention_w16_L${L}V${V}${SFX}.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o x2i_amd/libx2i_hip_w16_L${L}V${V}${SFX}.so $OBJS /tmp/attention_w16_L${L}V${V}${SFX}.o
  echo "built x2i_amd/libx2i_hip_w16_L${L}V${V}${SFX}.so"
done
(cd x2i_amd/csrc && X2I_ATTN_ABL= python gen_attn_w16.py > /dev/null)
