#!/usr/bin/env python3
"""Generator of the K-loop of the "two residents" bf16 GEMM (gemm_r2.hip) -> gemm_r2_loop.inc.

The form DESIGN section 10 item 1 asked for, turned around: instead of ONE wave per SIMD that keeps two accumulator sets and carries a
generated epilogue stream through its K-loop, TWO independent workgroups live on a CU (two waves per SIMD, <= 256 registers each), each
with a 256 x 128 output tile; while one of them runs its epilogue the other one's K-loop keeps the matrix pipe busy -- the overlap is the
hardware's wave scheduling, and every epilogue of gemm_device.h is reused as it is.  What makes two residents fit:

  * wave tile 64 x 128 (waves stacked along M): 4 x 8 accumulators of v_mfma_f32_16x16x32_bf16 = 128 registers in the accumulator file;
  * the A operand never touches LDS: a wave's 64 rows are its own, so their fragments come STRAIGHT FROM GLOBAL MEMORY into registers
    in MFMA layout (lane l: row 16 i + (l & 15), 16 bytes at k-byte 64 kh + 16 (l >> 4)), two K-tiles ahead, into the registers of the
    fragment that just died (2 slots x 2 k-halves x 4 = 64 registers);
  * the W operand (128 rows, wanted by all four waves) goes through a 4-stage LDS ring of 16 KiB K-tile images (LDS-DMA, three K-tiles
    ahead); its fragments pass through an 8-register-quad ring, each read seven fragments (28 MFMAs) ahead of its first use;
  * MFMA order: k-half, W fragment j, A fragment i (innermost): a W fragment dies after 4 MFMAs, an A fragment lives a k-half.

One barrier and two counted vmcnt waits per K-tile; lgkmcnt waits are counted (7 younger fragment reads stay in flight).  The body is
unrolled four times so that ring stages and register slots are immediates / fixed names.  Wait counts come from simulating the queues.
nk % 4 == 0, nk >= 4.  `python gen_gemm_r2.py` rewrites gemm_r2_loop.inc (committed; a CPU test regenerates and compares).
"""
import os
import sys

STAGE = 16384      # bytes per W K-tile image
NST = 4            # ring stages


def acc(i, j):
    return f"%[c{i * 8 + j}]"


def areg(slot, kh, i):
    return f"%[a{slot}{kh}{i}]"


class Sim:
    """Issue-order queues of the vector-memory and LDS operations (both return in order)."""

    def __init__(self):
        self.vm, self.ds = [], []

    def issue_vm(self, tag):
        self.vm.append(tag)

    def issue_ds(self, tag):
        self.ds.append(tag)

    def _count(self, q, tags):
        idx = [n for n, t in enumerate(q) if t in tags]
        if not idx:
            return None          # issued before the simulated window (the prologue drained it): nothing to wait for
        return len(q) - 1 - max(idx)

    def wait_vm(self, tags):
        n = self._count(self.vm, tags)
        return None if n is None else f"s_waitcnt vmcnt({n})"

    def wait_ds(self, tags):
        n = self._count(self.ds, tags)
        return None if n is None else f"s_waitcnt lgkmcnt({n})"


def body(u, sim, abl=0):
    """K-tile u.  abl (measurement builds only, wrong results): 1 = no A loads after the prologue, 2 = no W DMA, 4 = no W fragment reads."""
    L = []
    slot = u & 1
    for k in range(64):
        n, i = k >> 2, k & 3
        kh, j = n >> 3, n & 7
        if k == 0:
            w = sim.wait_vm({("A", u, 0, ii) for ii in range(4)})
            if w:
                L.append(w)
        if k == 32:
            # A fragments of k-half 1 (issued two bodies ago, behind that body's W pieces for tile u + 1: one wait covers both), then the
            # barrier: every wave's pieces of tile u + 1 have landed, and every wave is done with the stage of tile u - 1
            w = sim.wait_vm({("A", u, 1, ii) for ii in range(4)} | {("W", u + 1, jj) for jj in range(4)})
            if w:
                L.append(w)
            L.append("s_barrier")
        if i == 0:
            w = sim.wait_ds({("F", u, n)})
            if w:
                L.append(w)
        L.append(f"v_mfma_f32_16x16x32_bf16 {acc(i, j)}, %[wr{j}], {areg(slot, kh, i)}, {acc(i, j)}")
        if i == 3 and not (abl & 4):
            # ring slot j is free: fragment n + 8 (the other k-half of column block j; of the NEXT tile when n >= 8)
            n2 = (n + 8) & 15
            u2 = u + ((n + 8) >> 4)
            off = (u2 % NST) * STAGE + (n2 & 7) * 2048 + (n2 >> 3) * 512
            L.append(f"ds_read_b128 %[wr{j}], %[lw]" + (f" offset:{off}" if off else ""))
            sim.issue_ds(("F", u2, n2))
        if 28 <= k < 32 and not (abl & 1):
            L.append(f"buffer_load_dwordx4 {areg(slot, 0, k - 28)}, %[va{k - 28}], %[ra], %[koffa] offen")
            sim.issue_vm(("A", u + 2, 0, k - 28))
        if 60 <= k < 64 and not (abl & 1):
            L.append(f"buffer_load_dwordx4 {areg(slot, 1, k - 60)}, %[va{k - 60}], %[ra], %[koffa] offen offset:64")
            sim.issue_vm(("A", u + 2, 1, k - 60))
        if k in (36, 40, 44, 48) and not (abl & 2):
            jj = (k - 36) >> 2
            L.append(f"s_add_u32 m0, %[dma], {((u + 3) % NST) * STAGE + jj * 4096}")
        if k in (37, 41, 45, 49) and not (abl & 2):
            jj = (k - 37) >> 2
            L.append(f"buffer_load_dwordx4 %[vw{jj}], %[rw], %[koffw] offen lds")
            sim.issue_vm(("W", u + 3, jj))
    L += ["s_add_u32 %[koffa], %[koffa], 128", "s_add_u32 %[koffw], %[koffw], 128"]
    return L


def generate(abl=0):
    P = ["s_nop 4", "s_mov_b32 %[koffw], 0"]
    for t in range(3):
        for jj in range(4):
            P += [f"s_add_u32 m0, %[dma], {t * STAGE + jj * 4096}", "s_nop 0", f"buffer_load_dwordx4 %[vw{jj}], %[rw], %[koffw] offen lds"]
        P.append("s_add_u32 %[koffw], %[koffw], 128")
    P.append("s_mov_b32 %[koffa], 0")
    for t in range(2):
        for kh in range(2):
            for i in range(4):
                P.append(f"buffer_load_dwordx4 {areg(t, kh, i)}, %[va{i}], %[ra], %[koffa] offen" + (" offset:64" if kh else ""))
        P.append("s_add_u32 %[koffa], %[koffa], 128")
    P += ["s_waitcnt vmcnt(0)", "s_barrier"]
    for g in range(8):
        P.append(f"ds_read_b128 %[wr{g}], %[lw]" + (f" offset:{g * 2048}" if g else ""))
    P.append("s_lshr_b32 %[it], %[nk], 2")
    # steady state: simulate bodies 0..11 and take 8..11 (their queues have the steady-state shape); bodies 4..7 must give the same text
    sim = Sim()
    for g in range(8):
        sim.issue_ds(("F", 0, g))
    texts = [body(u, sim, abl) for u in range(12)]
    for u in range(4):
        assert texts[4 + u] == texts[8 + u], f"body {u}: wait counts of two consecutive rounds differ"
    loop = ["1:"]
    for u in range(8, 12):
        loop += texts[u]
    loop += ["s_sub_u32 %[it], %[it], 1", "s_cmp_lg_u32 %[it], 0", "s_cbranch_scc1 1b"]
    # the prefetches of the last bodies ran past K (into registers / stages nobody reads): drain them before LDS is reused
    D = ["s_waitcnt vmcnt(0) lgkmcnt(0)", "s_barrier"]
    # sanity: the first bodies (behind the drained prologue) must never wait for LESS than the steady state does
    return P + loop + D


OPERANDS_DOC = """// operands of X2I_GEMM_R2_LOOP (all named):
//   c0..c31   "+a"  f32x4  accumulators, c[i*8 + j] = rows 16i.., columns 16j.. of the 64 x 128 wave tile
//   wr0..7    "=&v" bf16x8 W fragment ring;  a{slot}{kh}{i} "=&v" bf16x8 A fragments (slot = K-tile parity)
//   va0..3    "v"   byte offset of this lane's 16 bytes of A row 16i + (lane & 15) in K-tile 0, k-half 0 (0x80000000 = out of range -> zeros)
//   vw0..3    "v"   byte offset of this lane's 16 bytes of W piece jj in K-tile 0
//   lw        "v"   LDS byte address of this lane's W fragment read in stage 0, column block 0, k-half 0
//   ra, rw    "s"   buffer descriptors of A / W;   dma "s" LDS byte address of this wave's first W piece in stage 0 (wave * 1024)
//   nk        "s"   number of K-tiles (multiple of 4, >= 4);   koffa, koffw, it "=&s" scratch
"""


def main():
    out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "gemm_r2_loop.inc")
    txt = ["// GENERATED by gen_gemm_r2.py -- do not edit; the schedule lives in the generator.", OPERANDS_DOC]
    for name, abl in (("X2I_GEMM_R2_LOOP", 0), ("X2I_GEMM_R2_LOOP_NOA", 1), ("X2I_GEMM_R2_LOOP_NOW", 2), ("X2I_GEMM_R2_LOOP_NOMEM", 3), ("X2I_GEMM_R2_LOOP_MFMA", 7)):
        L = generate(abl)
        if abl:
            txt.append("#ifdef X2I_ABLATION   // measurement builds only: wrong results by design")
        txt.append(f"// {name}: {len(L)} lines")
        txt.append(f"#define {name} \\")
        txt += [f'  "{l}\\n" \\' for l in L[:-1]]
        txt.append(f'  "{L[-1]}\\n"')
        if abl:
            txt.append("#endif")
        txt.append("")
    accs = ", ".join(f'[c{i * 8 + j}] "+a"(acc[{j >> 2}][{i}][{j & 3}])' for i in range(4) for j in range(8))
    txt.append("// acc[h][i][jj]: rows 16i.., columns 64h + 16jj.. of the 64 x 128 wave tile (two halves in the layout epilogue_store_lds takes)")
    txt.append(f"#define X2I_GEMM_R2_OPS_ACC(acc) {accs}")
    txt.append("#define X2I_GEMM_R2_OPS_ACC_IN(acc) " + accs.replace('"+a"', '"a"'))
    frs = ", ".join(f'[wr{n}] "=&v"(wr[{n}])' for n in range(8)) + ", " + \
        ", ".join(f'[a{s}{kh}{i}] "=&v"(af[{s}][{kh}][{i}])' for s in range(2) for kh in range(2) for i in range(4))
    txt.append(f"#define X2I_GEMM_R2_OPS_FRAG(wr, af) {frs}")
    vo = ", ".join(f'[va{n}] "v"(va[{n}])' for n in range(4)) + ", " + ", ".join(f'[vw{n}] "v"(vw[{n}])' for n in range(4))
    txt.append(f"#define X2I_GEMM_R2_OPS_VOFF(va, vw) {vo}")
    txt.append("")
    data = "\n".join(txt)
    if "--check" in sys.argv:
        cur = open(out).read() if os.path.exists(out) else ""
        sys.exit(0 if cur == data else 1)
    with open(out, "w") as fh:
        fh.write(data)
    print(f"wrote {out}")


if __name__ == "__main__":
    main()
