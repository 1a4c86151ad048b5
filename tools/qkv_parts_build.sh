#!/bin/bash
# Tools-only: libx2i_hip_qkv<n>.so = the PRODUCT build with the fused-QKV epilogue of the persistent GEMM compiled with -DX2I_QKV_ABL=<n>
# (gemm256p.hip: 86 no cos / sin loads, 87 no RMS lane reduction, 88 no Q / K stores, 89 no V^T stores, 90 q / k tiles parked only, 91 v tiles
# parked only, 92 no epilogue).  Wrong results by design; tools/qkv_parts.py times them (X2I_LIB_VARIANT=qkv<n>).
set -e
cd "$(dirname "$0")/.."
OBJS=$(ls x2i_amd/_build/*.hip.o | grep -v "\.abl\.o" | grep -v "gemm256p.hip.o")
for n in "$@"; do
  ( /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-value -ffp-contract=fast -DX2I_QKV_ABL=$n -c x2i_amd/csrc/gemm256p.hip -o /tmp/gemm256p_qkv$n.o &&
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o x2i_amd/libx2i_hip_qkv$n.so $OBJS /tmp/gemm256p_qkv$n.o && echo "built x2i_amd/libx2i_hip_qkv$n.so" ) &
done
wait
