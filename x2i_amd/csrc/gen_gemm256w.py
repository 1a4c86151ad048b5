#!/usr/bin/env python3
"""Generator of the hand-scheduled K-loop of the 4-wave 256x256x64 bf16 GEMM (gemm256w.hip) -> gemm256w_loop.inc.

The loop is ONE `asm volatile` statement (prologue + loop + drain): hipcc neither re-orders nor re-waits anything inside it, so the
instruction stream below is exactly what the wave issues.  Organisation (one wave per SIMD, wave tile 128 x 128 = 8 x 8 accumulators
of `v_mfma_f32_16x16x32_bf16`, 128 MFMAs per K-tile):

  * two LDS tile buffers (A image 32 KiB | W image 32 KiB each, the full-line image of gemm256.hip); the fragments of a k-half live
    in registers one k-half ahead (four sets of eight: wa/aa = k-half 0, wb/ab = k-half 1);
  * while tile t is multiplied, tile t+1 is landing / has landed in the other buffer and tile t+2 is fetched INTO THE BUFFER OF TILE t,
    operand by operand, as soon as a barrier has shown that every wave holds that operand's fragments of tile t in registers
    (prefetch distance: more than one whole K-tile, with two buffers);
  * LDS-DMA pieces, fragment reads and scalar bookkeeping sit one or two to a gap between MFMAs; waits are counted
    (`vmcnt(N)` = the number of younger pieces that may stay in flight), never drains.

The schedule is data: SLOTS maps "after MFMA k" to the events issued there; wait counts are derived by simulating the stream, and
the register / buffer hazards are asserted (see check()).  `python gen_gemm256w.py` rewrites gemm256w_loop.inc (committed; a CPU
test regenerates it and compares).
"""
import os
import sys

NF = 8  # fragments per operand per k-half (128 rows / 16)


def acc(i, j):
    return f"%[c{i * NF + j}]"


# ------------------------------------------------------------------------------------------------ event vocabulary
#   ("R1W", j) / ("R1A", i): read k-half 1 fragment of the CURRENT tile          -> wb[j] / ab[i]
#   ("R0W", j) / ("R0A", i): read k-half 0 fragment of the NEXT tile             -> wa[j] / aa[i]
#   ("XW",) / ("XA",):       flip the W / A fragment read address to the other buffer
#   ("TL",n):                scalar bookkeeping for the tile to fetch (three SALU steps, n = 0..2)
#   ("M0W", jj) / ("M0A", jj): set M0 for piece jj;  ("DW", jj) / ("DA", jj): issue the LDS-DMA piece
#   ("XD",):                 flip the DMA destination buffer
#   ("LGK", what):           s_waitcnt lgkmcnt(n) so that the reads named by `what` ("W1", "A1", "all") have returned
#   ("VM", what):            s_waitcnt vmcnt(n) so that the NEXT tile's pieces of operand `what` ("W", "A") have landed
#   ("BAR",):                s_barrier
#   ("CNT", n):              loop counter (0: decrement, 1: compare)
def default_slots(variant=0):
    """variant 0 = the product schedule; 3 = one barrier for both operands of the next tile (measurement builds only)."""
    s = {}

    def put(k, *ev):
        s.setdefault(k, []).extend(ev)

    for j in range(8):
        put(2 * j, ("R1W", j))
    put(1, ("TL", 0)); put(3, ("TL", 1)); put(5, ("TL", 2))
    put(15, ("XW",))
    put(18, ("LGK", "W1")); put(19, ("BAR",))
    k = 20
    for jj in range(5):  # W pieces 0..4 interleaved with the A k-half-1 reads
        put(k, ("M0W", jj)); put(k + 1, ("DW", jj)); put(k + 2, ("R1A", jj))
        k += 3
    put(36, ("R1A", 5)); put(38, ("R1A", 6)); put(40, ("R1A", 7)); put(41, ("XA",))
    put(45, ("LGK", "A1")); put(46, ("BAR",))
    put(47, ("M0W", 5)); put(48, ("DW", 5)); put(50, ("M0W", 6)); put(51, ("DW", 6)); put(53, ("M0W", 7)); put(54, ("DW", 7))
    put(56, ("M0A", 0)); put(57, ("DA", 0)); put(59, ("M0A", 1)); put(60, ("DA", 1))
    if variant == 3:
        put(62, ("VM", "A")); put(63, ("BAR",))      # the next tile's A pieces are younger than its W pieces: one wait covers both
        for j in range(8):
            put(64 + 2 * j, ("R0W", j))
        for i in range(8):
            put(80 + 2 * i, ("R0A", i))
        k = 96
        for jj in range(2, 8):
            put(k, ("M0A", jj)); put(k + 1, ("DA", jj))
            k += 3
        put(115, ("XD",))
    else:
        put(62, ("VM", "W")); put(63, ("BAR",))
        for j in range(8):
            put(64 + 2 * j, ("R0W", j))
        k = 80
        for jj in range(2, 7):
            put(k, ("M0A", jj)); put(k + 1, ("DA", jj))
            k += 3
        put(96, ("VM", "A")); put(97, ("BAR",))
        for i in range(8):
            put(98 + 2 * i, ("R0A", i))
        put(114, ("M0A", 7)); put(115, ("DA", 7)); put(116, ("XD",))
    put(121, ("CNT", 0)); put(122, ("CNT", 1))
    put(125, ("LGK", "all"))
    return s


def conv_slots():
    """Implicit-GEMM convolution form of the persistent loop (gemm256c.hip): the A pieces of a K-tile are the gather of ONE filter tap x 64
    input channels, so a piece's address is base(pixel) + a per-K-tile SCALAR offset (tap offset + channel slice: affine in the tap) and the
    only per-lane work is the zero padding -- bit `tap` of the pixel's invalid-tap mask turns the offset out of range:
      ("VA", jj):  tv = (mk[jj] << sh) & 0x80000000 | va[jj]   (sh = 31 - tap: two VALU instructions in a gap in front of piece jj)
      ("ADV", n):  ten SALU steps behind the last A piece that move (koffa, sh, cc, rc) to the next K-tile: koffa += 128, and at the end of a
                   tap's channel slices the tap advances (sh -= 1), at the end of a filter row koffa jumps to the next image row."""
    s = default_slots()
    for k in (121, 122):
        s[k] = [e for e in s[k] if e[0] != "CNT"]
    for jj, k in {0: 52, 1: 55, 2: 79, 3: 82, 4: 85, 5: 88, 6: 91, 7: 113}.items():
        s.setdefault(k, []).append(("VA", jj))
    for n in range(ADV_STEPS):
        s.setdefault(116 + n, []).append(("ADV", n))
    s.setdefault(124, []).append(("CNT", 0))
    s.setdefault(126, []).append(("CNT", 1))
    s[125] = [e for e in s[125] if e[0] != "LGK"]
    s.setdefault(127, []).append(("LGK", "all"))
    return {k: v for k, v in s.items() if v}


# Tile geometries of the persistent statements.  "256": the product tile (256 x 256: waves 2 x 2, 8 A + 8 W pieces per wave and K-tile, LDS buffers
# 64 KiB apart, W image 32 KiB behind the A image).  "512": 512 x 128 for the convolutions with <= 128 output channels (gemm512c.hip) -- the SAME
# 128 x 128 wave tile and MFMA / fragment stream, four waves stacked along M: 16 A pieces + 4 W pieces per wave, A buffers at 0 / 64 KiB, W buffers
# at 128 / 144 KiB (their own DMA base `dmaw`, flip 16 KiB): all 160 KiB of LDS, no epilogue staging (that epilogue leaves straight from registers).
GEOMS = {"256": dict(npa=8, npw=8, a_flip=0x10000, w_flip=0x10000, dmaw=False, w_buf1=65536 + 32768, w_buf0=32768),
         "512": dict(npa=16, npw=4, a_flip=0x10000, w_flip=0x4000, dmaw=True, w_buf1=16384, w_buf0=0)}


def conv512_slots():
    """Schedule of the 512 x 128 convolution form: the fragment reads, waits and barriers where the product schedule has them; 4 W pieces behind
    the first barrier, 16 A pieces (offset made in the gap in front of each: ("VA", jj) shares a gap with ("M0A", jj)) behind the second."""
    s = {}

    def put(k, *ev):
        s.setdefault(k, []).extend(ev)
    for j in range(8):
        put(2 * j, ("R1W", j))
    put(1, ("TL", 0)); put(3, ("TL", 1)); put(5, ("TL", 2))
    put(15, ("XW",))
    put(18, ("LGK", "W1")); put(19, ("BAR",))
    for jj in range(4):
        put(20 + 3 * jj, ("M0W", jj)); put(21 + 3 * jj, ("DW", jj))
    for i, k in enumerate((22, 25, 28, 31, 33, 35, 37, 39)):
        put(k, ("R1A", i))
    put(41, ("XA",))
    put(45, ("LGK", "A1")); put(46, ("BAR",))
    k = 47
    for jj in range(7):          # A pieces 0..6
        put(k, ("VA", jj), ("M0A", jj)); put(k + 1, ("DA", jj))
        k += 2
    put(62, ("VM", "W")); put(63, ("BAR",))
    for j in range(8):
        put(64 + 2 * j, ("R0W", j))
    for n, jj in enumerate(range(7, 11)):     # A pieces 7..10 in the odd gaps between the W fragment reads
        put(65 + 4 * n, ("VA", jj), ("M0A", jj)); put(67 + 4 * n, ("DA", jj))
    k = 80
    for jj in range(11, 16):      # A pieces 11..15
        put(k, ("VA", jj), ("M0A", jj)); put(k + 1, ("DA", jj))
        k += 2
    put(90, ("XD",))
    for n, k in enumerate((91, 92, 93, 94, 95, 99, 101, 103)):
        put(k, ("ADV", n))
    put(96, ("VM", "A")); put(97, ("BAR",))
    for i in range(8):
        put(98 + 2 * i, ("R0A", i))
    put(121, ("CNT", 0)); put(122, ("CNT", 1))
    put(125, ("LGK", "all"))
    return s


# The gather state of the K-tile to fetch -- (koffa: byte offset of its tap + channel slice in the image, koff: byte offset of its 64 weights in a W row,
# sh: 31 - tap index) -- advances through TWO nested counters with per-level increments, so that the K order is the launch's choice:
#   order "kx inner" (the product order since round 6):  K-tile (ky, c, kx): the three taps of a filter row follow each other for one channel
#       slice -- they read the SAME image lines shifted by a pixel, one K-tile apart: the re-read hits the XCD's L2 instead of the fabric (with
#       [ky][kx][c] the same lines come back Cin / 64 K-tiles later, after 32 workgroups' worth of other lines: a miss; for <= 128 output
#       channels that traffic -- 9 x the image at 128 FLOP per byte -- is what bounded the convolution at ~1.0 PFLOP/s);
#   order "c inner": K-tile (ky, kx, c) = the weight layout's own order, every offset advances by 128 bytes (the order of the other kernels:
#       bit-identical to them, option conv_korder = 0).
# state: c0 / c1 = steps left in the inner / middle counter; constants n0s / n1s, dA0 dS0 (every step), dA1 dS1 (added when the inner
# counter wraps), dA2 dW2 dS2 (added when the middle counter wraps as well).
ADV_LINES = ["s_add_u32 %[koffa], %[koffa], %[dA0]",
             "s_add_u32 %[koff], %[koff], %[dA0]",
             "s_add_u32 %[sh], %[sh], %[dS0]",
             "s_sub_u32 %[kc0], %[kc0], 1",
             "s_cmp_eq_u32 %[kc0], 0",
             "s_cselect_b32 %[kc0], %[n0s], %[kc0]",
             "s_cselect_b32 %[msk], -1, 0",
             "s_and_b32 %[tmp], %[msk], %[dA1]",
             "s_add_u32 %[koffa], %[koffa], %[tmp]",
             "s_add_u32 %[koff], %[koff], %[tmp]",
             "s_and_b32 %[tmp], %[msk], %[dS1]",
             "s_add_u32 %[sh], %[sh], %[tmp]",
             "s_and_b32 %[tmp], %[msk], 1",
             "s_sub_u32 %[kc1], %[kc1], %[tmp]",
             "s_cmp_eq_u32 %[kc1], 0",
             "s_cselect_b32 %[kc1], %[n1s], %[kc1]",
             "s_cselect_b32 %[msk], -1, 0",
             "s_and_b32 %[tmp], %[msk], %[dA2]",
             "s_add_u32 %[koffa], %[koffa], %[tmp]",
             "s_and_b32 %[tmp], %[msk], %[dW2]",
             "s_add_u32 %[koff], %[koff], %[tmp]",
             "s_and_b32 %[tmp], %[msk], %[dS2]",
             "s_add_u32 %[sh], %[sh], %[tmp]"]
ADV_STEPS = 8     # three instructions per step (the last one two); the W-row offset shares the image offset's first two increments (dW0 = dA0, dW1 = dA1 in both orders)
CONV_STATE_INIT = ["s_mov_b32 %[koffa], 0", "s_mov_b32 %[koff], 0", "s_mov_b32 %[sh], 31", "s_mov_b32 %[kc0], %[n0s]", "s_mov_b32 %[kc1], %[n1s]"]


def emit_event(ev, st):
    """-> list of asm lines; st = running stream state used to derive the wait counts."""
    kind = ev[0]
    if kind in ("R1W", "R1A", "R0W", "R0A"):
        n = ev[1]
        half = 512 if kind[1] == "1" else 0
        if kind[2] == "W":
            reg, addr = (f"%[wb{n}]" if kind[1] == "1" else f"%[wa{n}]"), "%[lw]"
        else:
            reg, addr = (f"%[ab{n}]" if kind[1] == "1" else f"%[aa{n}]"), "%[la]"
        st["ds"].append(kind + str(n))
        off = n * 2048 + half
        return [f"ds_read_b128 {reg}, {addr}" + (f" offset:{off}" if off else "")]
    geom = st.get("geom", GEOMS["256"])
    if kind == "XW":
        return [f"v_xor_b32 %[lw], {hex(geom['w_flip'])}, %[lw]"]
    if kind == "XA":
        return [f"v_xor_b32 %[la], {hex(geom['a_flip'])}, %[la]"]
    mode = st.get("mode", "W")
    conv = st.get("conv", False)
    if kind == "VA":
        k = ev[1] & 3
        m, v = ("nmk", "na") if mode in ("B1", "B2") else ("mk", "va")
        return [f"v_lshlrev_b32 %[tv{k}], %[sh], %[{m}{ev[1]}]", f"v_and_or_b32 %[tv{k}], %[tv{k}], %[cmsb], %[{v}{ev[1]}]"]
    if kind == "ADV":
        return ADV_LINES[3 * ev[1]:3 * ev[1] + 3]
    if kind == "TL" and conv and mode == "B1" and ev[1] == 0:
        return list(CONV_STATE_INIT)   # the next unit starts at tap 0, channel slice 0
    if kind == "TL":
        if mode == "W":  # one-tile kernel: fetch tile min(t + 2, nk - 1)
            return [["s_add_u32 %[tl], %[tl], 1", "s_min_u32 %[tmp], %[tl], %[nkm1]", "s_lshl_b32 %[koff], %[tmp], 7"][ev[1]]]
        if mode in ("B1", "B2"):  # last two K-tiles of a unit: fetch the first two K-tiles of the NEXT unit (byte offset nk0b of its row)
            if conv:
                return []         # (the convolution forms carry koff in the gather state: reset in B1, advanced by ADV)
            return (["s_mov_b32 %[koff], %[nk0b]"] if mode == "B1" else ["s_add_u32 %[koff], %[nk0b], 128"]) if ev[1] == 0 else []
        return []  # mode "A": koff advances behind the last piece (XD)
    if kind == "M0W":
        if geom["dmaw"]:
            return [f"s_add_u32 m0, %[dmaw], {ev[1] * 4096}" if ev[1] else "s_mov_b32 m0, %[dmaw]"]
        return [f"s_add_u32 m0, %[dma], {32768 + ev[1] * 4096}"]
    if kind == "M0A":
        return [f"s_add_u32 m0, %[dma], {ev[1] * 4096}" if ev[1] else "s_mov_b32 m0, %[dma]"]
    if kind == "DW":
        st["vm"].append(("W", st["iter"], ev[1]))
        v, r = ("nw", "rw" if conv else "nrw") if mode in ("B1", "B2") else ("vw", "rw")     # (the next unit may belong to another problem of a grouped launch; a convolution launch has one problem)
        return [f"buffer_load_dwordx4 %[{v}{ev[1]}], %[{r}], %[koff] offen" + st["aux_w"] + " lds"]
    if kind == "DA":
        st["vm"].append(("A", st["iter"], ev[1]))
        v, r = ("na", "ra" if conv else "nra") if mode in ("B1", "B2") else ("va", "ra")
        if conv:
            return [f"buffer_load_dwordx4 %[tv{ev[1] & 3}], %[{r}], %[koffa] offen" + st["aux_a"] + " lds"]
        return [f"buffer_load_dwordx4 %[{v}{ev[1]}], %[{r}], %[koff] offen" + st["aux_a"] + " lds"]
    if kind == "XD":
        return ([f"s_xor_b32 %[dma], %[dma], {hex(geom['a_flip'])}"] + ([f"s_xor_b32 %[dmaw], %[dmaw], {hex(geom['w_flip'])}"] if geom["dmaw"] else []) +
                (["s_add_u32 %[koff], %[koff], 128"] if mode == "A" and not conv else []))
    if kind == "BAR":
        return ["s_barrier"]
    if kind == "CNT":
        if mode in ("B1", "B2"):
            return []
        return [["s_sub_u32 %[it], %[it], 1", "s_cmp_lg_u32 %[it], 0"][ev[1]]]
    if kind == "LGK":
        want = {"W1": "R1W", "A1": "R1A", "all": ""}[ev[1]]
        last = max((i for i, d in enumerate(st["ds"]) if d.startswith(want)), default=None)
        n = len(st["ds"]) - 1 - last if last is not None else 0
        st["ds_waited"] = max(st.get("ds_waited", 0), (last + 1) if last is not None else 0)
        return [f"s_waitcnt lgkmcnt({n})"]
    if kind == "VM":
        # pieces of operand ev[1] of the NEXT tile were issued one iteration ago (st['iter'] - 1)
        idx = [i for i, (op, it, _) in enumerate(st["vm"]) if op == ev[1] and it == st["iter"] - 1]
        assert idx, "no pieces of the next tile in the simulated stream"
        n = len(st["vm"]) - 1 - max(idx)
        assert n < 64
        st["vm_n"][ev[1]] = n
        return [f"s_waitcnt vmcnt({n})"]
    raise ValueError(ev)


def body(slots, st, zero=False):
    """zero = True: the first K-tile of an output tile -- k-half 0 takes the constant 0 as its C operand (no accumulator clearing)."""
    lines = []
    for k in range(128):
        kh, i, j = k >> 6, (k >> 3) & 7, k & 7
        w = f"%[w{'ab'[kh]}{j}]"
        a = f"%[a{'ab'[kh]}{i}]"
        lines.append(f"v_mfma_f32_16x16x32_bf16 {acc(i, j)}, {w}, {a}, " + ("0" if zero and kh == 0 else acc(i, j)))
        for ev in slots.get(k, []):
            lines += emit_event(ev, st)
    return lines


def check(slots, geom=None):
    """Static hazards of the schedule (register reuse, buffer reuse, wait coverage)."""
    geom = geom or GEOMS["256"]
    npa, npw = geom["npa"], geom["npw"]
    pos = {}
    order = []
    for k in sorted(slots):
        for n, ev in enumerate(slots[k]):
            pos[ev] = (k, n)
            order.append(ev)
    assert sorted(e[1] for e in pos if e[0] == "DW") == list(range(npw)) and sorted(e[1] for e in pos if e[0] == "DA") == list(range(npa))
    for j in range(npw):
        assert pos[("M0W", j)] < pos[("DW", j)] < pos[("XD",)]
    for j in range(npa):
        assert pos[("M0A", j)] < pos[("DA", j)] < pos[("XD",)]
    for j in range(8):
        # wa[j] is read by MFMAs 8i + j (i = 0..7, k-half 0): last use 56 + j;  aa[i] by 8i .. 8i+7
        assert pos[("R0W", j)][0] >= 56 + j, "wa[%d] overwritten before its last use" % j
        assert pos[("R0A", j)][0] >= 8 * j + 7, "aa[%d] overwritten before its last use" % j
        # k-half 1 fragments must be in registers before MFMA 64 + j (W) / 64 + 8i (A): covered by an LGK wait placed earlier
        assert pos[("R1W", j)] < pos[("LGK", "W1")] and pos[("LGK", "W1")][0] < 64
        assert pos[("R1A", j)] < pos[("LGK", "A1")] and pos[("LGK", "A1")][0] < 64
        assert pos[("R1W", j)] < pos[("XW",)] < pos[("R0W", j)]
        assert pos[("R1A", j)] < pos[("XA",)] < pos[("R0A", j)]
    bars = [pos[e] for e in order if e == ("BAR",)]
    # barrier after the W (A) k-half-1 reads have returned, before the first W (A) piece overwrites the current buffer
    barpos = sorted(p for e, p in pos.items() if e[0] == "BAR")
    # (BAR events are identical tuples: recover their positions from the slot table)
    barpos = sorted((k, n) for k in slots for n, ev in enumerate(slots[k]) if ev == ("BAR",))
    assert len(barpos) in (3, 4)
    assert pos[("LGK", "W1")] < barpos[0] < min(pos[("DW", j)] for j in range(npw))
    assert pos[("LGK", "A1")] < barpos[1] < min(pos[("DA", j)] for j in range(npa))
    if len(barpos) == 4:
        assert pos[("VM", "W")] < barpos[2] < min(pos[("R0W", j)] for j in range(8))
        assert pos[("VM", "A")] < barpos[3] < min(pos[("R0A", j)] for j in range(8))
    else:  # one wait on the (younger) A pieces + one barrier in front of both fragment sets
        assert ("VM", "W") not in pos and pos[("VM", "A")] < barpos[2] < min(min(pos[("R0W", j)], pos[("R0A", j)]) for j in range(8))
    # M0 must be written at least one instruction (here: an MFMA) before the piece that uses it, and not be overwritten in between
    m0w = sorted((pos[e], e) for e in pos if e[0] in ("M0W", "M0A", "DW", "DA"))
    for (p0, e0), (p1, e1) in zip(m0w[::2], m0w[1::2]):
        assert e0[0].startswith("M0") and e1[0].startswith("D") and e0[1] == e1[1] and e0[0][2] == e1[0][1], (e0, e1)
        assert p1[0] > p0[0], "M0 write and its piece must be separated by an MFMA"
    # SCC: the compare must be the last SCC-writing scalar instruction of the body
    assert all(pos[e] < pos[("CNT", 1)] for e in pos if e[0] in ("TL", "M0W", "M0A", "XD", "ADV") or e == ("CNT", 0))
    for j in range(npa):
        if ("VA", j) in pos:   # convolution form: a piece's offset is made in an earlier gap, its temporary is free again (piece j - 4 issued)
            assert pos[("VA", j)][0] < pos[("DA", j)][0] and (j < 4 or pos[("DA", j - 4)][0] < pos[("VA", j)][0])
            assert all(pos[("ADV", n)][0] > pos[("DA", npa - 1)][0] for n in range(ADV_STEPS)) and all(pos[("ADV", n)] < pos[("ADV", n + 1)] for n in range(ADV_STEPS - 1))
    assert pos[("LGK", "all")] > max(pos[("R0A", j)] for j in range(8))
    del bars


def generate(aux_a="", aux_w="", variant=0):
    slots = default_slots(variant)
    check(slots)
    st = dict(ds=[], vm=[], iter=0, aux_a=aux_a, aux_w=aux_w, vm_n={})
    L = []
    L.append("s_nop 4")  # scalar operands may be fresh from v_readfirstlane
    L.append("s_sub_u32 %[nkm1], %[nk], 1")
    L.append("s_mov_b32 %[koff], 0")
    # ---- prologue: tile 0 -> buffer 0 (any order), tile 1 (clamped to the last tile) -> buffer 1 in the loop's order (W, then A)
    for jj in range(8):
        L += [f"s_add_u32 m0, %[dma], {jj * 4096}", "s_nop 0", f"buffer_load_dwordx4 %[va{jj}], %[ra], %[koff] offen" + aux_a + " lds"]
    for jj in range(8):
        L += [f"s_add_u32 m0, %[dma], {32768 + jj * 4096}", "s_nop 0", f"buffer_load_dwordx4 %[vw{jj}], %[rw], %[koff] offen" + aux_w + " lds"]
    L += ["s_min_u32 %[tl], %[nkm1], 1", "s_lshl_b32 %[koff], %[tl], 7"]
    st["iter"] = -1
    for jj in range(8):
        L += [f"s_add_u32 m0, %[dma], {65536 + 32768 + jj * 4096}", "s_nop 0", f"buffer_load_dwordx4 %[vw{jj}], %[rw], %[koff] offen" + aux_w + " lds"]
        st["vm"].append(("W", -1, jj))
    for jj in range(8):
        L += [f"s_add_u32 m0, %[dma], {65536 + jj * 4096}", "s_nop 0", f"buffer_load_dwordx4 %[va{jj}], %[ra], %[koff] offen" + aux_a + " lds"]
        st["vm"].append(("A", -1, jj))
    L += ["s_mov_b32 %[tl], 1", "s_mov_b32 %[it], %[nk]"]
    L += ["s_waitcnt vmcnt(16)", "s_barrier"]
    for j in range(8):
        L.append(f"ds_read_b128 %[wa{j}], %[lw]" + (f" offset:{j * 2048}" if j else ""))
    for i in range(8):
        L.append(f"ds_read_b128 %[aa{i}], %[la]" + (f" offset:{i * 2048}" if i else ""))
    L.append("s_waitcnt lgkmcnt(0)")
    # ---- steady state: simulate iteration 0 (behind the prologue) and iteration 1; their wait counts must agree
    st["iter"] = 0
    st["ds"] = []
    b0 = body(slots, st)
    n0 = dict(st["vm_n"])
    st["iter"] = 1
    st["ds"] = []
    b1 = body(slots, st)
    assert b0 == b1 and n0 == st["vm_n"], "the wait counts of the first and of a steady-state iteration differ"
    st["vm_n"].setdefault("W", st["vm_n"]["A"])
    L.append("1:")
    L += b0
    L.append("s_cbranch_scc1 1b")
    L += ["s_waitcnt vmcnt(0)", "s_barrier", "s_nop 7", "s_nop 7"]
    return L, st["vm_n"]


def generate_persistent(aux_a="", aux_w="", conv=False, geom_name="256"):
    """Seamless form for the persistent kernel: PRO (first output tile of a workgroup: K-tiles 0 and 1, first fragments), MAIN (the nk
    K-tiles of one output tile; the last two iterations fetch K-tiles 0 / 1 of the NEXT output tile through the `na` / `nw` offsets and
    leave its first fragments in wa / aa, so the next MAIN starts multiplying at once), DRAIN (after the last tile)."""
    geom = GEOMS[geom_name]
    npa, npw = geom["npa"], geom["npw"]
    slots = (conv512_slots() if geom_name == "512" else conv_slots()) if conv else default_slots()
    check(slots, geom)
    st = dict(ds=[], vm=[], iter=-1, aux_a=aux_a, aux_w=aux_w, vm_n={}, mode="A", conv=conv, geom=geom)
    wbase = "%[dmaw]" if geom["dmaw"] else "%[dma]"

    def w_piece(jj, buf):
        off = (geom["w_buf1"] if buf else geom["w_buf0"]) + jj * 4096
        return [f"s_add_u32 m0, {wbase}, {off}" if off else f"s_mov_b32 m0, {wbase}", "s_nop 0",
                f"buffer_load_dwordx4 %[vw{jj}], %[rw], %[koff] offen" + aux_w + " lds"]

    def a_piece(jj, buf):   # prologue A piece: plain offset, or the convolution gather's masked offset + scalar tap offset
        m0 = [f"s_add_u32 m0, %[dma], {buf * 65536 + jj * 4096}"]
        if conv:   # (the two VALU instructions are the wait state between the M0 write and the piece)
            return m0 + emit_event(("VA", jj), dict(mode="A", geom=geom)) + [f"buffer_load_dwordx4 %[tv{jj & 3}], %[ra], %[koffa] offen" + aux_a + " lds"]
        return m0 + ["s_nop 0", f"buffer_load_dwordx4 %[va{jj}], %[ra], %[koff] offen" + aux_a + " lds"]
    P = ["s_nop 4"] + (list(CONV_STATE_INIT) if conv else ["s_mov_b32 %[koff], %[k0b]"])
    for jj in range(npa):
        P += a_piece(jj, 0)
    for jj in range(npw):
        P += w_piece(jj, 0)
    P += list(ADV_LINES) if conv else ["s_add_u32 %[koff], %[k0b], 128"]
    for jj in range(npw):
        P += w_piece(jj, 1)
        st["vm"].append(("W", -1, jj))
    for jj in range(npa):
        P += a_piece(jj, 1)
        st["vm"].append(("A", -1, jj))
    P += list(ADV_LINES) if conv else []
    P += [f"s_waitcnt vmcnt({npa + npw})", "s_barrier"]
    for j in range(8):
        P.append(f"ds_read_b128 %[wa{j}], %[lw]" + (f" offset:{j * 2048}" if j else ""))
    for i in range(8):
        P.append(f"ds_read_b128 %[aa{i}], %[la]" + (f" offset:{i * 2048}" if i else ""))
    P.append("s_waitcnt lgkmcnt(0)")
    # simulate PRO, A, A, B1, B2, A (next output tile): every body must derive the same wait counts
    bodies, counts = {}, []
    for it, mode in enumerate(["A", "A", "B1", "B2", "A"]):
        st["iter"], st["ds"], st["mode"] = it, [], mode
        b = body(slots, st)
        counts.append(dict(st["vm_n"]))
        if mode in bodies:
            assert bodies[mode] == b
        bodies[mode] = b
    assert all(c == counts[0] for c in counts), counts
    # MAIN: one statement for both starts ("+a" accumulators).  zs != 0: start from zero -- the first iteration is a copy of body A
    # whose k-half 0 takes C = 0, so the incoming accumulator values are never read (len >= 3); zs == 0: continue from the
    # accumulators handed in (len >= 2).  One statement, so that the accumulators keep ONE register assignment around it (two
    # statements on the two sides of a C++ branch made hipcc spill the whole accumulator file at the join).
    st["iter"], st["ds"], st["mode"] = 5, [], "A"
    a0 = body(slots, st, zero=True)
    MC = ["s_nop 4"] + ([] if conv else ["s_add_u32 %[koff], %[k0b], 256"]) + ["s_sub_u32 %[it], %[nk], 2", "s_cmp_lg_u32 %[zs], 0", "s_cbranch_scc1 5f",
          "s_cmp_lg_u32 %[it], 0", "s_cbranch_scc0 2f", "s_branch 1f", "5:"]
    MC += a0 + ["s_cbranch_scc0 2f", "1:"] + bodies["A"] + ["s_cbranch_scc1 1b", "2:"] + bodies["B1"] + bodies["B2"] + ["s_nop 7", "s_nop 7"]
    # The last body has requested the NEXT unit's first fragments (ds_read into the FRAG0 operands).  They must have landed when the
    # statement ends: the compiler is free to copy or spill those registers between two statements (the fused-QKV instantiation does
    # spill), and a copy taken while the LDS data is still in flight carries stale values into the next unit's first MFMAs.
    MC += ["s_waitcnt lgkmcnt(0)"]
    MZ = None
    D = ["s_waitcnt vmcnt(0)", "s_barrier"]
    return P, MC, MZ, D, counts[0]


def generate_partial_io():
    """Stream-K segment hand-off: the 64 accumulator tiles of a wave go to / come from a 256 KiB slab [64 tiles][256 threads][16 B]
    straight from the accumulator file (no VGPR round trip).  Stores are write-through (sc1) and drained before the statement ends
    (the flag that publishes them is stored behind a workgroup barrier); loads are sc1 (served past the CU's L1)."""
    ST = ["s_nop 4", "s_mov_b32 %[so], 0"]
    LD = ["s_nop 4", "s_mov_b32 %[so], 0", "s_cmp_lg_u32 %[fromp], 0", "s_cbranch_scc0 3f"]  # fromp == 0: nothing to load (accumulators undefined)
    for n in range(64):
        ST += [f"buffer_store_dwordx4 %[c{n}], %[vo], %[rs], %[so] offen sc1", "s_add_u32 %[so], %[so], 4096"]
        LD += [f"buffer_load_dwordx4 %[c{n}], %[vo], %[rs], %[so] offen sc1", "s_add_u32 %[so], %[so], 4096"]
    ST += ["s_waitcnt vmcnt(0)"]
    LD += ["s_waitcnt vmcnt(0)", "3:"]
    return ST, LD


OPERANDS_DOC = """// operands of X2I_GEMM256W_LOOP (all named):
//   c0..c63   "+a"  f32x4  accumulators, c[i*8 + j] = rows 16i.., columns 16j.. of the wave tile
//   wa0..7, wb0..7, aa0..7, ab0..7  "=&v" bf16x8 fragment registers (scratch)
//   va0..7, vw0..7  "v"  per-piece byte offsets of this lane's 16 bytes (0x80000000 = out of range -> zeros)
//   la, lw    "+v"  LDS byte address of this lane's A / W fragment read in buffer 0
//   ra, rw    "s"   buffer descriptors of A / W
//   dma       "+s"  LDS byte address of this wave's first A piece in buffer 0 (wave * 1024)
//   nk        "s"   number of K-tiles (>= 1)
//   nkm1, koff, tl, tmp, it  "=&s" scratch
"""


def main():
    out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "gemm256w_loop.inc")
    # _V1.._V3 are A/B variants compiled only into the measurement library (-DX2I_ABLATION): non-temporal A / W pieces, one barrier
    # for both operands of the next tile
    variants = {"X2I_GEMM256W_LOOP": ("", "", 0), "X2I_GEMM256W_LOOP_V1": (" nt", "", 0), "X2I_GEMM256W_LOOP_V2": ("", " nt", 0),
                "X2I_GEMM256W_LOOP_V3": ("", "", 3)}
    txt = ["// GENERATED by gen_gemm256w.py -- do not edit; the schedule table lives in the generator.", OPERANDS_DOC]
    for name, (aa, aw, var) in variants.items():
        L, vm = generate(aa, aw, var)
        txt.append(f"// {name}: {len(L)} lines; vmcnt before the next tile's W / A fragments are read: {vm['W']} / {vm['A']}")
        txt.append(f"#define {name} \\")
        txt += [f'  "{l}\\n" \\' for l in L[:-1]]
        txt.append(f'  "{L[-1]}\\n"')
        txt.append("")
    P, MC, MZ, D, vm = generate_persistent()
    txt.append("// persistent (seamless) form -- additional operands: wa0..7 / aa0..7 are \"+v\" (live between the statements), na0..7 / nw0..7 \"v\" = the NEXT")
    txt.append("// output tile's piece offsets (0x80000000 everywhere behind the last tile); nk >= 2;")
    txt.append("// k0b / nk0b \"s\" = byte offset (128 per K-tile) of this / the next unit's first K-tile within a row (stream-K segments start at K-tile k0)")
    ST, LD = generate_partial_io()
    for name, L in (("X2I_GEMM256P_PRO", P), ("X2I_GEMM256P_MAIN", MC), ("X2I_GEMM256P_DRAIN", D),
                    ("X2I_GEMM256P_STORE_PARTIAL", ST), ("X2I_GEMM256P_LOAD_PARTIAL", LD)):
        txt.append(f"// {name}: {len(L)} lines" + (f"; vmcnt W / A: {vm['W']} / {vm['A']}" if "MAIN" in name else ""))
        txt.append(f"#define {name} \\")
        txt += [f'  "{l}\\n" \\' for l in L[:-1]]
        txt.append(f'  "{L[-1]}\\n"')
        txt.append("")
    # convolution form of the persistent statements (gemm256c.hip)
    P, MC, MZ, D, vmc = generate_persistent(conv=True)
    txt.append("// implicit-GEMM convolution form (gemm256c.hip) -- additional operands: mk0..7 / nmk0..7 \"v\" = invalid-tap masks of this / the next unit's")
    txt.append("// piece rows (bit t = filter tap t lies outside the image), tv0..3 \"=&v\" scratch, c8 \"s\" = 0x80000000, koffa / sh / cc / rc \"+s\" = the gather")
    txt.append("// state of the K-tile to fetch; koff (the W row offset) is part of that state here: \"+s\"; c0 / c1 \"+s\" = steps left in the inner / middle counter;")
    txt.append("// n0s / n1s / dA0..2 / dW0..2 / dS0..2 \"s\" = the launch's K order (see ADV_LINES in the generator); msk / tmp \"=&s\" scratch.  A unit is a whole tile.")
    for name, L in (("X2I_GEMM256C_PRO", P), ("X2I_GEMM256C_MAIN", MC)):
        txt.append(f"// {name}: {len(L)} lines" + (f"; vmcnt W / A: {vmc['W']} / {vmc['A']}" if "MAIN" in name else ""))
        txt.append(f"#define {name} \\")
        txt += [f'  "{l}\\n" \\' for l in L[:-1]]
        txt.append(f'  "{L[-1]}\\n"')
        txt.append("")
    mk = ", ".join(f'[mk{n}] "v"(mk[{n}])' for n in range(NF)) + ", " + ", ".join(f'[nmk{n}] "v"(nmk[{n}])' for n in range(NF))
    txt.append(f"#define X2I_GEMM256C_OPS_MASK(mk, nmk) {mk}")
    tv = ", ".join(f'[tv{n}] "=&v"(tv[{n}])' for n in range(4))
    txt.append(f"#define X2I_GEMM256C_OPS_TMP(tv) {tv}")
    # 512 x 128 convolution form (gemm512c.hip): the same statements on the "512" geometry
    P, MC, MZ, D, vm5 = generate_persistent(conv=True, geom_name="512")
    txt.append("// 512 x 128 implicit-GEMM convolution form (gemm512c.hip: <= 128 output channels; four waves stacked along M, the SAME 128 x 128 wave tile and")
    txt.append("// MFMA / fragment stream) -- operands as the 256 x 256 convolution form with 16 A pieces (va / na / mk / nmk 0..15) and 4 W pieces (vw / nw 0..3)")
    txt.append("// per wave, and dmaw \"+s\" = LDS byte address of this wave's first W piece in buffer 0 (W buffers at 128 / 144 KiB).")
    for name, L in (("X2I_GEMM512C_PRO", P), ("X2I_GEMM512C_MAIN", MC)):
        txt.append(f"// {name}: {len(L)} lines" + (f"; vmcnt W / A: {vm5['W']} / {vm5['A']}" if "MAIN" in name else ""))
        txt.append(f"#define {name} \\")
        txt += [f'  "{l}\\n" \\' for l in L[:-1]]
        txt.append(f'  "{L[-1]}\\n"')
        txt.append("")
    g5 = GEOMS["512"]
    vo5 = ", ".join(f'[va{n}] "v"(va[{n}])' for n in range(g5["npa"])) + ", " + ", ".join(f'[vw{n}] "v"(vw[{n}])' for n in range(g5["npw"]))
    txt.append(f"#define X2I_GEMM512C_OPS_VOFF(va, vw) {vo5}")
    nx5 = ", ".join(f'[na{n}] "v"(na[{n}])' for n in range(g5["npa"])) + ", " + ", ".join(f'[nw{n}] "v"(nw[{n}])' for n in range(g5["npw"]))
    txt.append(f"#define X2I_GEMM512C_OPS_NEXT(na, nw) {nx5}")
    mk5 = ", ".join(f'[mk{n}] "v"(mk[{n}])' for n in range(g5["npa"])) + ", " + ", ".join(f'[nmk{n}] "v"(nmk[{n}])' for n in range(g5["npa"]))
    txt.append(f"#define X2I_GEMM512C_OPS_MASK(mk, nmk) {mk5}")
    # operand lists (the asm statement itself is written out in gemm256w.hip)
    accs = ", ".join(f'[c{i * NF + j}] "+a"(acc[{j >> 2}][{i}][{j & 3}])' for i in range(NF) for j in range(NF))
    txt.append("// acc[h][i][jj]: rows 16i.., columns 64h + 16jj.. of the 128 x 128 wave tile (two halves in the layout the shared epilogues take)")
    txt.append(f"#define X2I_GEMM256W_OPS_ACC(acc) {accs}")
    frs = ", ".join(f'[{nm}{n}] "=&v"(fr[{b * 8 + n}])' for b, nm in enumerate(("wa", "wb", "aa", "ab")) for n in range(NF))
    txt.append(f"#define X2I_GEMM256W_OPS_FRAG(fr) {frs}")
    vo = ", ".join(f'[va{n}] "v"(va[{n}])' for n in range(NF)) + ", " + ", ".join(f'[vw{n}] "v"(vw[{n}])' for n in range(NF))
    txt.append(f"#define X2I_GEMM256W_OPS_VOFF(va, vw) {vo}")
    txt.append("// persistent kernel: acc[h][c][r][jj] = rows 32c + 16r.., columns 64h + 16jj.. (the epilogue leaves in 32-row chunks)")
    for con, tag in (("+a", "IO"), ("=&a", "OUT"), ("a", "IN")):
        accp = ", ".join(f'[c{i * NF + j}] "{con}"(acc[{j >> 2}][{i >> 1}][{i & 1}][{j & 3}])' for i in range(NF) for j in range(NF))
        txt.append(f"#define X2I_GEMM256P_OPS_ACC_{tag}(acc) {accp}")
    for con, tag in (("+v", "IO"), ("=&v", "OUT")):
        f0 = ", ".join(f'[{nm}{n}] "{con}"(fr[{b * 8 + n}])' for b, nm in ((0, "wa"), (2, "aa")) for n in range(NF))
        txt.append(f"#define X2I_GEMM256P_OPS_FRAG0_{tag}(fr) {f0}")
    f1 = ", ".join(f'[{nm}{n}] "=&v"(fr[{b * 8 + n}])' for b, nm in ((1, "wb"), (3, "ab")) for n in range(NF))
    txt.append(f"#define X2I_GEMM256P_OPS_FRAG1(fr) {f1}")
    nx = ", ".join(f'[na{n}] "v"(na[{n}])' for n in range(NF)) + ", " + ", ".join(f'[nw{n}] "v"(nw[{n}])' for n in range(NF))
    txt.append(f"#define X2I_GEMM256P_OPS_NEXT(na, nw) {nx}")
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
