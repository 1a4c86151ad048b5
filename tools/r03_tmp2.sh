cd $GRAFT_REPO_ROOT/x2i_amd/csrc
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DX2I_QKV_NO_REORDER -c gemm256p.hip -o /tmp/g.o -I../../include 2>&1 | grep error
cd $GRAFT_REPO_ROOT
python - <<'PY'
import subprocess,glob,os
# relink the product library with the test object
from x2i_amd import build as B
print("relink")
PY
