#!/usr/bin/env python3
"""Generator of the hand-scheduled K-loop of the persistent four-wave e4m3 GEMM (gemm256p.hip, F8 = true) -> gemm256f8_loop.inc.

Same tile (256 x 256, four waves, 128 x 128 wave tiles in the accumulator file), LDS images, DMA pieces and buffer ring as the bf16
kernel (gen_gemm256w.py) -- a K-tile is 128 BYTES of every operand row in both, so everything that moves bytes is shared -- but the
matrix instruction is `v_mfma_f32_16x16x128_f8f6f4` (e4m3 x e4m3, K = 128; the un-scaled encoding = unit block scales, one
instruction word less per MFMA than v_mfma_scale_*): ONE instruction per accumulator and K-tile, 64 per K-tile and wave (32 cycles
each: the same 2048 matrix-pipe cycles as the bf16 kernel's 128 x 16, for twice the K depth).  Its A / B operands are 8-register
tuples = the two 16-byte k-halves of a fragment row (one ds_read_b128 each), so there is no k-half double buffering: the wave holds
the WHOLE fragment set of the current K-tile (W[0..7], A[0..7]: 128 VGPRs, hand-assigned v128..v255 -- an asm operand cannot be split
into the two halves a fragment is read in -- and clobbered) and replaces each fragment in place once its last MFMA has issued.

MFMA order: two COLUMN halves.  k = 32 h + 4 i + jj multiplies A[i] with W[4 h + jj]: W[0..3] are dead behind MFMA 31 and are
re-read half a tile before the next tile needs them; A[i] is dead behind MFMA 35 + 4 i; W[4..7] and A[7] (dead only at the very
end) are read at the HEAD of the next tile, 28+ MFMAs before their first use.  (The first form of this loop ran row-major, k = 8 i + j:
W[j] was dead behind MFMA 56 + j and wanted again 8 MFMAs later, all four waves re-reading 64 KiB of LDS inside 256 cycles: 0.44 ->
0.48 of the 5 PF peak only.)

  gaps 0..4     A[7], W[4..7] of THIS tile (from the current buffer)
  gap 11 / 12   every fragment of this tile is in registers: s_waitcnt lgkmcnt(0), barrier -> the tile's LDS buffer is free
  gaps 13..44   the 16 LDS-DMA pieces of tile t+2 into that buffer, M0 write and piece in separate gaps
  gap 45 / 46   s_waitcnt vmcnt(16) (tile t+1 has landed; the 16 pieces just issued stay in flight), barrier, read addresses flip
  gaps 47..60   W[0..3], A[0..6] of tile t+1, each behind its last use
  next tile     counted lgkmcnt waits in front of MFMA 0 (W[0..3], A[0]), 4 (A[1]), 8 (A[2]); the full wait at gap 11 covers the rest

Two barriers per K-tile (four in the bf16 loop), prefetch distance a whole K-tile (2048 cycles).  Persistent form only: PRO (first
two K-tiles of a workgroup's first unit), MAIN (one unit: PRE = fragments of its first K-tile, then len bodies; the last two fetch
the NEXT unit's first two K-tiles; nothing stays in registers between two statements except the accumulators, so the epilogue has the
whole VGPR file), DRAIN.  Wait counts are derived by simulating the stream; hazards are asserted (check()).
`python gen_gemm256f8.py` rewrites gemm256f8_loop.inc (committed; tests/test_host_cpu.py regenerates and compares).
"""
import os
import sys

NF = 8
FB = 128          # first fragment register: W[j] = v[FB + 8j : +7], A[i] = v[FB + 64 + 8i : +7]


def acc(i, j):
    return f"%[c{i * NF + j}]"


def W(j, half=None):
    b = FB + 8 * j
    return f"v[{b}:{b + 7}]" if half is None else f"v[{b + 4 * half}:{b + 4 * half + 3}]"


def A(i, half=None):
    b = FB + 64 + 8 * i
    return f"v[{b}:{b + 7}]" if half is None else f"v[{b + 4 * half}:{b + 4 * half + 3}]"


def mfma_operands(k):
    """MFMA k of a K-tile -> (i, j): row fragment A[i], column fragment W[j]."""
    h, i, jj = k >> 5, (k >> 2) & 7, k & 3
    return i, 4 * h + jj


def last_use(kind, n):
    return max(k for k in range(64) if mfma_operands(k)[0 if kind == "RA" else 1] == n)


def first_use(kind, n):
    return min(k for k in range(64) if mfma_operands(k)[0 if kind == "RA" else 1] == n)


HEAD = [("RA", 7), ("RW", 4), ("RW", 5), ("RW", 6), ("RW", 7)]          # read at the head of their own tile
TAIL = [("RW", 0), ("RW", 1), ("RW", 2), ("RW", 3)] + [("RA", i) for i in range(7)]   # read at the end of the previous tile


# events: ("RA", i) / ("RW", j): both halves of a fragment
def slots(kind):
    """kind: 'A' normal body, 'B1' / 'B2' the last two K-tiles of a unit (B2 reads no next-tile fragments)."""
    s = {}

    def put(k, *ev):
        s.setdefault(k, []).extend(ev)

    for g, ev in enumerate(HEAD):
        put(g, ev)
    put(5, ("TL",))
    put(11, ("LGK0",))
    put(12, ("BAR",))
    for jj in range(16):
        op, n = ("A", jj) if jj < 8 else ("W", jj - 8)
        put(13 + 2 * jj, ("M0", op, n))
        put(14 + 2 * jj, ("D", op, n))
    put(45, ("XD",))
    if kind != "B2":
        put(45, ("VM",))
        put(46, ("BAR",))
    put(46, ("XA",), ("XW",))
    if kind != "B2":
        for g, ev in enumerate(TAIL[:10]):
            put(47 + g, ev)              # W[0..3] (dead behind MFMA 31), A[0..5] (A[5] behind 55)
        put(60, TAIL[10])                # A[6] behind MFMA 59
    if kind == "A":
        put(61, ("CNT", 0))
        put(62, ("CNT", 1))
    return s


def check(s, kind):
    pos = {}
    for k in sorted(s):
        for n, ev in enumerate(s[k]):
            pos.setdefault(ev, (k, n))
    bars = sorted((k, n) for k in s for n, ev in enumerate(s[k]) if ev == ("BAR",))
    for ev in HEAD:      # fragments of the CURRENT tile read at its head: before their first MFMA (with room for the LDS latency), before
        assert pos[ev][0] + 8 <= first_use(*ev) and pos[ev] < pos[("LGK0",)] < bars[0] and pos[ev] < pos[("XA",)]   # the full wait and the flips
    assert pos[("LGK0",)][0] < first_use("RA", 3), "A[3] and later rely on the full wait (A[0..2] have counted waits)"
    for op in "AW":
        for n in range(8):
            assert bars[0] < pos[("M0", op, n)] and pos[("M0", op, n)][0] < pos[("D", op, n)][0], "M0 write and its piece must sit in different gaps"
            assert pos[("D", op, n)] < pos[("XD",)]
    m0s = sorted((pos[e], e) for e in pos if e[0] in ("M0", "D"))
    for (p0, e0), (p1, e1) in zip(m0s[::2], m0s[1::2]):
        assert e0[0] == "M0" and e1[0] == "D" and e0[1:] == e1[1:], (e0, e1)
    if kind != "B2":
        assert len(bars) == 2 and pos[("VM",)] < bars[1]
        assert max(pos[("D", op, n)] for op in "AW" for n in range(8)) < pos[("VM",)], "the wait count assumes all 16 new pieces are younger"
        for ev in TAIL:
            assert pos[ev][0] >= last_use(*ev), "%s%d overwritten before its last MFMA" % ev
            assert bars[1] < pos[ev] and pos[("XA",)] < pos[ev] and pos[("XW",)] < pos[ev]
    if kind == "A":
        assert all(pos[e] < pos[("CNT", 1)] for e in pos if e[0] in ("TL", "M0", "XD") or e == ("CNT", 0)), "SCC must survive to the branch"


ABL = os.environ.get("X2I_F8_ABL", "")   # measurement builds only (WRONG results): the loop without its pieces (nodma), barriers (nobar),
                                         # fragment reads (nolds) or M0 writes (nom0) -- profiles/r04am_fp8_kloop_ablation.log


def emit(ev, st):
    kind = ev[0]
    mode = st["mode"]
    if (ABL == "nobar" and kind == "BAR") or (ABL == "nom0" and kind == "M0"):
        return []
    if ABL == "nolds" and kind in ("RA", "RW"):
        st["ds"] += [(kind, ev[1], 0), (kind, ev[1], 1)]
        return []
    if ABL == "nodma" and kind == "D":
        st["vm"] += 1
        return []
    if ABL == "nodma" and kind == "VM":
        return []
    if kind in ("RA", "RW"):
        n = ev[1]
        reg, addr = (A, "%[la]") if kind == "RA" else (W, "%[lw]")
        st["ds"] += [(kind, n, 0), (kind, n, 1)]
        return [f"ds_read_b128 {reg(n, 0)}, {addr}" + (f" offset:{n * 2048}" if n else ""),
                f"ds_read_b128 {reg(n, 1)}, {addr} offset:{n * 2048 + 512}"]
    if kind == "TL":
        if mode == "B1":
            return ["s_mov_b32 %[koff], %[nk0b]"]
        if mode == "B2":
            return ["s_add_u32 %[koff], %[nk0b], 128"]
        return []
    if kind == "LGK0":
        st["ds"] = []
        return ["s_waitcnt lgkmcnt(0)"]
    if kind == "BAR":
        return ["s_barrier"]
    if kind == "M0":
        off = (32768 if ev[1] == "W" else 0) + ev[2] * 4096
        return [f"s_add_u32 m0, %[dma], {off}" if off else "s_mov_b32 m0, %[dma]"]
    if kind == "D":
        nxt = mode in ("B1", "B2")
        v = ("n" if nxt else "v") + ev[1].lower() + str(ev[2])
        r = ("nr" if nxt else "r") + ev[1].lower()
        st["vm"] += 1
        return [f"buffer_load_dwordx4 %[{v}], %[{r}], %[koff] offen lds"]
    if kind == "XD":
        return ["s_xor_b32 %[dma], %[dma], 0x10000"] + (["s_add_u32 %[koff], %[koff], 128"] if mode == "A" else [])
    if kind == "VM":
        assert st["vm"] == 16
        return ["s_waitcnt vmcnt(16)"]
    if kind == "XA":
        return ["v_xor_b32 %[la], 0x10000, %[la]"]
    if kind == "XW":
        return ["v_xor_b32 %[lw], 0x10000, %[lw]"]
    if kind == "CNT":
        return [["s_sub_u32 %[it], %[it], 1", "s_cmp_lg_u32 %[it], 0"][ev[1]]]
    raise ValueError(ev)


def need(st, what):
    """s_waitcnt so that the fragment reads named in `what` (list of (kind, n)) have returned (LDS reads return in order; the
    counter has four bits, so a count above 15 becomes the stricter 15)."""
    idx = [i for i, (k, n, _) in enumerate(st["ds"]) if (k, n) in what]
    if not idx:
        return []
    cnt = min(15, len(st["ds"]) - 1 - max(idx))
    return [f"s_waitcnt lgkmcnt({cnt})"]


def body(kind, st, zero=False):
    """64 MFMAs of one K-tile with the events of slots(kind) in the gaps.  st['ds'] carries the fragment reads still outstanding
    from the previous body (or PRE): TAIL, in that order."""
    s = slots(kind)
    check(s, kind)
    st["mode"], st["vm"] = kind, 0
    L = []
    full = False
    for k in range(64):
        i, j = mfma_operands(k)
        if not full:
            want = {0: [("RW", 0), ("RW", 1), ("RW", 2), ("RW", 3), ("RA", 0)], 4: [("RA", 1)], 8: [("RA", 2)]}.get(k, [])
            if want:
                L += need(st, want)
        c = "0" if zero else acc(i, j)
        L.append(f"v_mfma_f32_16x16x128_f8f6f4 {acc(i, j)}, {W(j)}, {A(i)}, {c}")
        for ev in s.get(k, []):
            if ev[0] == "LGK0":
                full = True
            L += emit(ev, st)
    return L


def pre_reads(st):
    L = []
    st["ds"] = []
    for ev in TAIL:
        L += emit(ev, st)
    return L


def generate():
    # PRO: K-tiles k0 / k0 + 1 of the workgroup's first unit -> buffers 0 / 1
    P = ["s_nop 4", "s_mov_b32 %[koff], %[k0b]"]
    for b in range(2):
        for op, base in (("a", 0), ("w", 32768)):
            for jj in range(8):
                P += [f"s_add_u32 m0, %[dma], {b * 65536 + base + jj * 4096}", "s_nop 0", f"buffer_load_dwordx4 %[v{op}{jj}], %[r{op}], %[koff] offen lds"]
        if b == 0:
            P += ["s_add_u32 %[koff], %[k0b], 128"]
    # MAIN
    st = dict(ds=[], vm=0, mode="A")
    PRE = ["s_nop 4", "s_waitcnt vmcnt(16)", "s_barrier"] + pre_reads(st)
    ds_after_pre = list(st["ds"])
    bodies = {}
    tails = {}
    for kind, zero in (("A0", True), ("A", False), ("B1", False), ("B2", False)):
        st["ds"] = list(ds_after_pre)
        bodies[kind] = body("A" if kind == "A0" else kind, st, zero=zero)
        tails[kind] = list(st["ds"])
    # every body that is followed by another one must leave the same outstanding reads as PRE (same order): the waits of the next
    # body were derived from that list
    assert tails["A"] == ds_after_pre == tails["B1"] == tails["A0"], (tails, ds_after_pre)
    assert tails["B2"] == []
    MC = PRE + ["s_add_u32 %[koff], %[k0b], 256", "s_sub_u32 %[it], %[nk], 2", "s_cmp_lg_u32 %[zs], 0", "s_cbranch_scc1 5f",
                "s_cmp_lg_u32 %[it], 0", "s_cbranch_scc0 2f", "s_branch 1f", "5:"]
    MC += bodies["A0"] + ["s_cbranch_scc0 2f", "1:"] + bodies["A"] + ["s_cbranch_scc1 1b", "2:"] + bodies["B1"] + bodies["B2"] + ["s_nop 7", "s_nop 7"]
    D = ["s_waitcnt vmcnt(0)", "s_barrier"]
    return P, MC, D


DOC = """// operands of the e4m3 persistent loop (all named; fragments are NOT operands: v128..v255, clobbered):
//   c0..c63   "+a"  f32x4  accumulators, c[i*8 + j] = rows 16i.., columns 16j.. of the wave tile
//   va0..7, vw0..7 / na0..7, nw0..7  "v"  per-piece byte offsets of this lane's 16 bytes in this / the next unit (0x80000000 = zeros)
//   la, lw    "+v"  LDS byte address of this lane's A / W fragment read in the buffer of the unit's first K-tile
//   ra, rw, nra, nrw  "s"  buffer descriptors;  dma "+s" LDS byte address of this wave's first A piece in that buffer
//   nk "s" K-tiles of the unit (>= 3 from zero, >= 2 continuing); k0b / nk0b "s" byte offset (128 per K-tile) of this / the next unit's
//   first K-tile within a row; zs "s" != 0: start from zero (the first K-tile takes C = 0)
//   koff, it  "=&s" scratch
"""


def main():
    # an ablation stream (X2I_F8_ABL: a loop without its DMA / barriers / LDS reads, WRONG results) never lands in the product file: it goes
    # to gemm256f8_loop_abl_<what>.inc and carries an #error for builds without -DX2I_ABLATION
    out = os.path.join(os.path.dirname(os.path.abspath(__file__)), f"gemm256f8_loop_abl_{ABL}.inc" if ABL else "gemm256f8_loop.inc")
    P, MC, D = generate()
    txt = ["// GENERATED by gen_gemm256f8.py -- do not edit; the schedule table lives in the generator.", DOC]
    if ABL:
        txt += [f"// ABLATION STREAM ({ABL}): wrong results by design, measurement builds only", "#ifndef X2I_ABLATION",
                f'#error "gemm256f8_loop_abl_{ABL}.inc is an ablation stream: measurement builds (-DX2I_ABLATION) only"', "#endif"]
    for name, L in (("X2I_GEMM256F8_PRO", P), ("X2I_GEMM256F8_MAIN", MC), ("X2I_GEMM256F8_DRAIN", D)):
        txt.append(f"// {name}: {len(L)} lines")
        txt.append(f"#define {name} \\")
        txt += [f'  "{l}\\n" \\' for l in L[:-1]]
        txt.append(f'  "{L[-1]}\\n"')
        txt.append("")
    clob = ", ".join(f'"v{r}"' for r in range(FB, 256))
    txt.append(f"#define X2I_GEMM256F8_FRAG_CLOBBERS {clob}")
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
