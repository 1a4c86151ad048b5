"""Build libx2i_hip.so (gfx950) in-tree with hipcc.  `python -m x2i_amd.build [--force]`.

hipcc cross-compiles without a GPU; the resulting .so is git-ignored but travels to the GPU box with the
working-tree snapshot.  One object per .hip file, compiled in parallel, relinked only when something changed.
"""
import concurrent.futures
import glob
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(HERE, "_build")
LIB = os.path.join(HERE, "libx2i_hip.so")
# measurement-only library for tools/ (ablation kernels that are "wrong results by design", the k-half-unit GEMM form):
# same sources, the files below recompiled with -DX2I_ABLATION.  Never loaded by the product package.
LIB_ABLATE = os.path.join(HERE, "libx2i_hip_ablate.so")
ABLATE_SRCS = ("gemm.hip", "gemm_ablate.hip", "gemm_r2.hip", "gemm256w.hip", "gemm256p.hip", "attention.hip", "attention16.hip", "attention_pp.hip", "attention_w16.hip", "attention_bwd.hip", "c_api.hip")
# A/B kernels that no product path selects (round 6 prune): compiled and linked ONLY into the measurement library -- gemm_r2.hip (the "two
# residents" GEMM, measured 1.6x slower), attention16.hip (compiler-scheduled 16 x 16 x 32 attention, superseded by the generated attention_w16.hip)
ABLATE_ONLY_SRCS = ("gemm_r2.hip", "attention16.hip")
ARCH = "gfx950"
FLAGS = ["--offload-arch=" + ARCH, "-O3", "-std=c++17", "-fPIC", "-Wno-unused-value", "-ffp-contract=fast"]
# per-file extras.  attention_bwd.hip: the SLP vectoriser pairs the element-wise values of two pipeline slots into v_pk_add_f32 / v_pk_mul_f32, which
# moves the even value's work into the odd slot and costs more beside an MFMA than two plain instructions (MI355X_MICROARCH.md, filler price list)
FILE_FLAGS = {"attention_bwd.hip": ["-fno-slp-vectorize"]}


def _hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    raise RuntimeError("hipcc not found")


def _newest_header():
    hs = glob.glob(os.path.join(CSRC, "*.h")) + glob.glob(os.path.join(CSRC, "*.inc")) + glob.glob(os.path.join(HERE, "..", "include", "*.h"))
    return max(os.path.getmtime(h) for h in hs)


def _compile(src, force, ablate=False):
    obj = os.path.join(OBJ, os.path.basename(src) + (".abl.o" if ablate else ".o"))
    if not force and os.path.exists(obj) and os.path.getmtime(obj) > max(os.path.getmtime(src), _newest_header(), os.path.getmtime(os.path.abspath(__file__))):
        return obj, False
    cmd = [_hipcc()] + FLAGS + FILE_FLAGS.get(os.path.basename(src), []) + (["-DX2I_ABLATION"] if ablate else []) + ["-c", src, "-o", obj]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("hipcc failed for %s:\n%s\n%s" % (src, r.stdout, r.stderr))
    return obj, True


def _link(lib, objs, changed, verbose):
    if changed or not os.path.exists(lib):
        cmd = [_hipcc(), "--offload-arch=" + ARCH, "-shared", "-fPIC", "-o", lib] + objs
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("link failed:\n%s\n%s" % (r.stdout, r.stderr))
        if verbose:
            print("built", lib)
    elif verbose:
        print("up to date:", lib)


# third, tiny library: register-resident MFMA streams for bench.py's live `matrix_pipe_alone` figure (tools/ubench/mfma_pipe_lib.hip).
# Measurement only: never linked into, or loaded by, the product library / package.
UBENCH_SRC = os.path.join(HERE, "..", "tools", "ubench", "mfma_pipe_lib.hip")
LIB_UBENCH = os.path.join(HERE, "..", "tools", "ubench", "libx2i_ubench.so")


def build_ubench(force=False, verbose=True):
    deps = [UBENCH_SRC, os.path.join(os.path.dirname(UBENCH_SRC), "mfma_power_kernel.h")]
    if not force and os.path.exists(LIB_UBENCH) and os.path.getmtime(LIB_UBENCH) > max(os.path.getmtime(d) for d in deps):
        return LIB_UBENCH
    cmd = [_hipcc(), "--offload-arch=" + ARCH, "-O3", "-shared", "-fPIC", UBENCH_SRC, "-o", LIB_UBENCH]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("hipcc failed for %s:\n%s\n%s" % (UBENCH_SRC, r.stdout, r.stderr))
    if verbose:
        print("built", os.path.normpath(LIB_UBENCH))
    return LIB_UBENCH


def build(force=False, verbose=True, ablate=True):
    os.makedirs(OBJ, exist_ok=True)
    srcs = sorted(glob.glob(os.path.join(CSRC, "*.hip")))
    jobs = [(s, False) for s in srcs if os.path.basename(s) not in ABLATE_ONLY_SRCS]
    if ablate:
        jobs += [(s, True) for s in srcs if os.path.basename(s) in ABLATE_SRCS]
    with concurrent.futures.ThreadPoolExecutor(max_workers=min(os.cpu_count() or 4, len(jobs))) as ex:
        res = list(ex.map(lambda j: _compile(j[0], force, j[1]), jobs))
    prod = {os.path.basename(j[0]): r for j, r in zip(jobs, res) if not j[1]}
    abl = {os.path.basename(j[0]): r for j, r in zip(jobs, res) if j[1]}
    _link(LIB, [o for o, _ in prod.values()], any(c for _, c in prod.values()), verbose)
    if ablate:
        objs = [(abl.get(n) or prod[n]) for n in prod] + [abl[n] for n in ABLATE_ONLY_SRCS if n in abl]
        _link(LIB_ABLATE, [o for o, _ in objs], any(c for _, c in objs), verbose)
    build_ubench(force, verbose)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
