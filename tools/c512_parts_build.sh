#!/bin/bash
# Tools-only: libx2i_hip_c512_<n>.so = the product build with csrc/gemm512c.hip compiled with -DX2I_C512_ABL=<n> (1: no epilogue, 2: no global
# stores, 3: the next unit's gather offsets are not computed).  Wrong results by design; timed with X2I_LIB_VARIANT=c512_<n> python tools/conv_probe.py
set -e
cd "$(dirname "$0")/.."
OBJS=$(ls x2i_amd/_build/*.hip.o | grep -v "\.abl\.o" | grep -v "gemm512c.hip.o")
for n in "$@"; do
  ( /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-value -ffp-contract=fast -DX2I_C512_ABL=$n -c x2i_amd/csrc/gemm512c.hip -o /tmp/gemm512c_$n.o &&
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o x2i_amd/libx2i_hip_c512_$n.so $OBJS /tmp/gemm512c_$n.o && echo "built x2i_amd/libx2i_hip_c512_$n.so" ) &
done
wait
