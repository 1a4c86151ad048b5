#!/usr/bin/env python3
"""Generator of the hand-scheduled flash-attention forward on v_mfma_f32_16x16x32_bf16 (attention_w16.hip) -> attn_w16_loop.inc.

gen_attn_w4.py's program (one wave per SIMD, four waves x 64 query rows per workgroup, 64-key tiles, three-tile software pipeline with the
softmax of tile i in the gaps between the MFMAs of O += V(i-1) P(i-1) and S(i+1) = K(i+1) Q~^T - m, fragments read into the accumulator
file, 2-deep K / V^T rings by LDS-DMA, defer-max with the running maximum as the C operand of a tile's first MFMAs) on the OTHER MFMA
shape: the matrix pipe alone sustains 11-14 % more FLOP/s at this part's power cap with 16 x 16 x 32 than with 32 x 32 x 16
(tools/ubench/mfma_power.hip).  What the shape changes:

  * a wave's 64 query rows are four query blocks qb of 16; a tile's 64 keys four key blocks kb; lane (c = lane & 15, g = lane >> 4) holds
    query c of each query block and keys 4g .. 4g+3 of each key block: S[set][kb][qb] = 4 registers;
  * a unit = one fragment (1 KiB) + its FOUR MFMAs (one per query block): ("K", ds, kb): S[kb][qb] += K(kb, ds) Q[qb][ds], d in four steps of 32;
    ("V", sp, db): O[db][qb] += V^T(db, sp) P[sp][qb], eight d-blocks of 16, two key spans of 32 -- 128 MFMAs of 16 cycles per tile;
  * P^T[sp][qb] = bf16 pairs of S[2 sp][qb][0..3], S[2 sp + 1][qb][0..3]: the lane's OWN registers are the eight k-positions 8g .. 8g+7 of the
    PV MFMA's B operand when k-position kk stands for key 16 ((kk >> 2) & 1) + 4 (kk >> 3) + (kk & 3) of the span -- so V^T must arrive with its
    keys in that order within every 32-key span (the contract attention16.hip's VPERM form proved; the caller permutes for now);
  * a query's keys sit in four lanes (g = 0 .. 3): the tile maximum is combined with v_permlane16_swap + v_permlane32_swap, the row sums stay
    per lane until the epilogue; the -m copies are 4 registers per query block (every register of a lane's accumulator is the same query).

Round 6 (stream-K, attention_w16.hip): the program can START at any key tile (%[so0] / %[so20]) from a state left in memory (%[cont]: prologue(cont=True) --
O, -m and the row sums come from the slab %[sr], S(t0) gets C = -m, and the first iteration is an ordinary softmax pass with the rare rescale of the loaded O)
and can END by leaving that state instead of the epilogue (%[hand]: state_store()).  An item cut along the key axis into chained parts is summed in exactly
the order of the undivided item: bit-identical.

`python gen_attn_w16.py` rewrites attn_w16_loop.inc (committed; tests/test_host_cpu.py regenerates and compares).
"""
import os
import sys

LEAD = int(os.environ.get("X2I_ATTN_LEAD", "4"))   # fragment reads are issued this many units (= one fragment, two MFMAs) ahead of their use
RING = 16           # fragment ring slots (4 accumulator registers each)
NEG_BIG = "0xf149f2ca"   # -1.0e30f
ABL = os.environ.get("X2I_ATTN_ABL", "")
if ABL == "none":
    ABL = ""            # measurement only (tools/attn_w16_sweep_build.sh): nobar / nosync / nolgk / novalu / noread / nodma give wrong results
ABLS = set(filter(None, ABL.split("+")))   # several at once: X2I_ATTN_ABL=novalu+noread+nolgk+nosync+nodma = the MFMA stream alone

# ------------------------------------------------------------------------------------------------ register map
_v, _a = 32, 0      # v0..v31 belong to the statement's operands


def valloc(n, align=1):
    global _v
    _v = (_v + align - 1) // align * align
    b = _v
    _v += n
    assert _v <= 256, "out of VGPRs"
    return b


def aalloc(n, align=1):
    global _a
    _a = (_a + align - 1) // align * align
    b = _a
    _a += n
    assert _a <= 256, "out of accumulator registers"
    return b


MPU = 4             # MFMAs per unit (one per query block)
SA = valloc(128, 4)       # S[set][kb][qb][r]: scores MINUS the running row maximum (exp2 domain) -- see NEGM
PF = valloc(32, 4)        # P[sp][qb][w]: ONE set; softmax(i) overwrites span sp behind the last PV MFMA that read P(i-1)[sp]
NEGM = valloc(16, 4)      # NEGM[qb][r]: minus the running row maximum, four copies per query block = the C operand of the first
                          # QK^T MFMA of a key block, so that the scores arrive with the maximum already subtracted
TMP = valloc(12, 4)
LACC = valloc(16, 4)      # L[qb][r]: row sums of p, accumulated BY THE MATRIX PIPE (ones x P^T, one MFMA per span and query block, issued with the
                          # span's first PV unit): the vector unit is the scarce resource of this kernel (twice the MFMA issues per tile), the sums
                          # then cover all 64 keys of a tile in every lane of a query (no exchange in the epilogue), and they scale with O
ONES = valloc(4, 4)       # bf16 pairs of 1.0: the A operand of the row-sum MFMAs
MX, MX2, ALPHA, DELTA = (valloc(4) for _ in range(4))
OA = aalloc(128, 4)       # O[db][qb][r]
QF = aalloc(64, 4)        # Q[qb][ds], pre-multiplied by scale * log2(e)
FR = aalloc(RING * 4, 4)  # fragment ring
QTMP = SA + 64            # the raw Q fragments pass through score set 1 in the prologue


def S(st, kb, qb, r=None):
    b = SA + ((st * 4 + kb) * 4 + qb) * 4
    return f"v[{b}:{b + 3}]" if r is None else f"v{b + r}"


def P(sp, qb, w=None):
    b = PF + (sp * 4 + qb) * 4
    return f"v[{b}:{b + 3}]" if w is None else f"v{b + w}"


def NM(qb, r=None):
    b = NEGM + qb * 4
    return f"v[{b}:{b + 3}]" if r is None else f"v{b + r}"


def O(db, qb, r=None):
    b = OA + (db * 4 + qb) * 4
    return f"a[{b}:{b + 3}]" if r is None else f"a{b + r}"


def Q(qb, ds):
    b = QF + (qb * 4 + ds) * 4
    return f"a[{b}:{b + 3}]"


def F(slot):
    b = FR + (slot % RING) * 4
    return f"a[{b}:{b + 3}]"


def LA(qb, r=None):
    b = LACC + qb * 4
    return f"v[{b}:{b + 3}]" if r is None else f"v{b + r}"


def T(i):
    assert i < 12
    return f"v{TMP + i}"


# ------------------------------------------------------------------------------------------------ softmax of one tile (VALU stream)
def swap_combine(L, op, regs):
    """regs[qb] <- op over the four lanes that hold one query (lane ^ 16, lane ^ 32), through MX2 copies."""
    for swap in ("v_permlane16_swap_b32", "v_permlane32_swap_b32"):
        for qb in range(4):
            L.append(f"v_mov_b32 v{MX2 + qb}, v{regs + qb}")
        L.append("s_nop 1")                                                    # VALU write -> v_permlane*_swap read
        for qb in range(4):
            L.append(f"{swap} v{regs + qb}, v{MX2 + qb}")
        for qb in range(4):
            L.append(f"{op} v{regs + qb}, v{regs + qb}, v{MX2 + qb}")


def softmax_pass1(st, masked, first, uid):
    """Row maxima of score set `st` and the defer-max decision.  The scores already carry -m_run (NEGM was the C operand of their
    first MFMA), so the common case is: max over the tile <= THR, nothing to do.  Otherwise (rare; scalar branch) the rows that grew
    adopt the tile maximum: delta = max(tile max, 0), scores -= delta, NEGM -= delta, alpha = exp2(-delta), and the flag makes the end of
    the iteration scale O and the row sums by alpha.  `first`: the scores are raw (C = 0) and every row adopts its maximum."""
    L = []
    if masked:
        # keys at or behind the sequence end.  Lane holds (tile-relative) key 16 kb + 4 g + r; %[lim] = S - kv0 - 4 g
        L.append(f"v_mov_b32 {T(11)}, {NEG_BIG}")
        for kb in range(4):
            for r in range(4):
                L.append(f"v_cmp_lt_i32 vcc, {kb * 16 + r}, %[lim]")
                for qb in range(4):
                    L.append(f"v_cndmask_b32 {S(st, kb, qb, r)}, {T(11)}, {S(st, kb, qb, r)}, vcc")
    for qb in range(4):
        vals = [S(st, kb, qb, r) for kb in range(4) for r in range(4)]
        t = [T((qb & 1) * 4 + k) for k in range(4)]
        for k in range(4):
            L.append(f"v_max3_f32 {t[k]}, {vals[3 * k]}, {vals[3 * k + 1]}, {vals[3 * k + 2]}")
        L.append(f"v_max3_f32 {t[0]}, {t[0]}, {vals[12]}, {vals[13]}")
        L.append(f"v_max3_f32 {t[1]}, {t[1]}, {vals[14]}, {vals[15]}")
        L.append(f"v_max3_f32 {t[0]}, {t[0]}, {t[1]}, {t[2]}")
        L.append(f"v_max_f32 v{MX + qb}, {t[0]}, {t[3]}")
    # the other 48 keys of a query row live in lanes ^ 16, ^ 32, ^ 48.  The common case needs no exchange: "no row grew by more than THR" is
    # "no LANE saw a score above THR" (the scores carry -m already); only the rare path (and the first tile) combines the four lanes' maxima
    if first:
        swap_combine(L, "v_max_f32", MX)
    subs = [f"v_sub_f32 {S(st, kb, qb, r)}, {S(st, kb, qb, r)}, v{(MX if first else DELTA) + qb}" for kb in range(4) for qb in range(4) for r in range(4)]
    if first:
        for qb in range(4):
            L.append(f"v_sub_f32 {NM(qb, 0)}, 0, v{MX + qb}")
            L += [f"v_mov_b32 {NM(qb, r)}, {NM(qb, 0)}" for r in range(1, 4)]
        L += subs
        L += [f"v_mov_b32 {LA(qb, r)}, 0" for qb in range(4) for r in range(4)]
        return L
    L.append(f"v_max3_f32 {T(0)}, v{MX}, v{MX + 1}, v{MX + 2}")
    L.append(f"v_max_f32 {T(0)}, {T(0)}, v{MX + 3}")
    L.append(f"v_cmp_lt_f32 %[cnd], %[thr], {T(0)}")                           # some row of this wave grew by more than THR = 8 ?
    D = ["s_nop 3", "s_cmp_lg_u64 %[cnd], 0", f"s_cbranch_scc0 .Lkeep{uid}_%="]
    swap_combine(D, "v_max_f32", MX)
    for qb in range(4):
        D.append(f"v_max_f32 v{DELTA + qb}, 0, v{MX + qb}")
    for qb in range(4):
        D.append(f"v_exp_f32 v{ALPHA + qb}, -v{DELTA + qb}")
    for qb in range(4):
        D.append(f"v_sub_f32 {NM(qb, 0)}, {NM(qb, 0)}, v{DELTA + qb}")
        D += [f"v_mov_b32 {NM(qb, r)}, {NM(qb, 0)}" for r in range(1, 4)]
    D += subs
    D += ["s_mov_b32 %[fl], 1", "s_nop 1", f".Lkeep{uid}_%=:"]
    L.append(D)       # a list element stays contiguous (the branch must not skip interleaved MFMAs / reads)
    return L


def softmax_pass2(st):
    """p = exp2(s'); row sums; bf16 pairs -> P fragment (sub-tile u, k-step kt) = registers 8kt .. 8kt+7 of S[u].  The steps of a value
    run as a software pipeline -- exp(k), add(k - 2), cvt of the pair behind (k - 3) -- which also spreads the slower exponentials
    evenly over the MFMA gaps.  A conversion is tagged with its group g: it may only be issued behind the PV MFMAs that read P[g]."""
    seq = []
    for sp in range(2):
        for qb in range(4):
            for k in range(8):
                kb, r = 2 * sp + (k >> 2), k & 3
                seq.append((S(st, kb, qb, r), qb, sp, P(sp, qb, k // 2) if k & 1 else None, S(st, kb, qb, r - 1) if k & 1 else None))
    L = []
    n = len(seq)
    for k in range(n + 3):
        if k < n:
            L.append(f"v_exp_f32 {seq[k][0]}, {seq[k][0]}")
        if 0 <= k - 3 < n and seq[k - 3][3] is not None:
            e = seq[k - 3]
            L.append(("CVT", e[2], f"v_cvt_pk_bf16_f32 {e[3]}, {e[4]}, {e[0]}"))
    return L


def o_rescale(uid):
    """O *= alpha per query row when the flag is up; behind the last PV MFMA of the iteration."""
    L = ["s_cmp_lg_u32 %[fl], 0", f"s_cbranch_scc0 .Lnors{uid}_%=", "s_nop 15", "s_nop 15"]
    for db in range(8):
        for qb in range(4):
            L += [f"v_accvgpr_read_b32 {T(k)}, {O(db, qb, k)}" for k in range(4)]
            L += [f"v_mul_f32 {T(k)}, {T(k)}, v{ALPHA + qb}" for k in range(4)]
            L += [f"v_accvgpr_write_b32 {O(db, qb, k)}, {T(k)}" for k in range(4)]
    L += [f"v_mul_f32 {LA(qb, r)}, {LA(qb, r)}, v{ALPHA + qb}" for qb in range(4) for r in range(4)]
    L += ["s_mov_b32 %[fl], 0", "s_nop 3", f".Lnors{uid}_%=:"]
    return L


# ------------------------------------------------------------------------------------------------ MFMA / LDS stream
def unit_list(has_qk, has_pv):
    """Fragment-sized units of an iteration.  ("K", ds, u): S[u][qb] += K(u, ds) Q[qb][ds]; ("V", g, db) with g = (u, kt):
    O[db][qb] += V^T(db, u, kt) P[u][kt][qb].  The PV units come EARLY (P(i-1) is complete when the iteration starts, and its
    registers are wanted back for P(i)); the QK units late (the first of them reads NEGM, which the defer-max decision may change)."""
    ku = [("K", ds, kb) for ds in range(4) for kb in range(4)] if has_qk else []
    vu = [("V", sp, db) for sp in range(2) for db in range(8)] if has_pv else []
    if not (ku and vu):
        return ku + vu
    out = vu[:8]
    for i in range(8):
        out += [vu[8 + i], ku[i]]
    return out + ku[8:]


def frag_read(unit, slot, par):
    """`par`: ring slot of the tile (K and V^T rings are two slots of 16 KiB: the slot is an immediate, both parities are emitted)."""
    kind, a, b = unit
    if kind == "K":   # d-step ds = a (address register per ds: the swizzle is an XOR), key block kb = b (+ 16 rows x 256 B)
        return f"ds_read_b128 {F(slot)}, %[ka{a}] offset:{b * 4096 + par * 0x4000}"
    return f"ds_read_b128 {F(slot)}, %[va{a}] offset:{b * 2048 + par * 0x4000}"   # span sp = a, d-block db = b (+ 16 rows x 128 B)


def unit_mfmas(unit, slot, s_dst, c_init):
    kind, a, b = unit
    if kind == "K":
        ds, kb = a, b
        return [f"v_mfma_f32_16x16x32_bf16 {S(s_dst, kb, qb)}, {F(slot)}, {Q(qb, ds)}, " +
                ((NM(qb) if c_init else "0") if ds == 0 else S(s_dst, kb, qb)) for qb in range(4)]
    sp, db = a, b
    m = [f"v_mfma_f32_16x16x32_bf16 {O(db, qb)}, {F(slot)}, {P(sp, qb)}, {O(db, qb)}" for qb in range(4)]
    if db == 0:   # row sums of the span: ones x P^T (every register of a lane's result is the sum over the span's 32 keys for query c)
        m += [f"v_mfma_f32_16x16x32_bf16 {LA(qb)}, v[{ONES}:{ONES + 3}], {P(sp, qb)}, {LA(qb)}" for qb in range(4)]
    return m


def n_mfmas(us):
    return sum(len(unit_mfmas(u, 0, 0, True)) for u in us)


def sync_wait():
    """Ring hand-over, part 1: every fragment read of this iteration has been issued (and had time to return) above.  Wait for them
    and for this wave's pieces of the tiles the NEXT iteration reads, barrier; the slots this iteration read are free."""
    if "nosync" in ABLS:
        return []
    return ["s_waitcnt vmcnt(0) lgkmcnt(0)"] + ([] if "nobar" in ABLS else ["s_barrier"])


def sync_dma(par):
    """Part 2, as (M0 write, piece) pairs to be spread between the iteration's last MFMAs: the next K tile (%[so] = its byte offset) and
    the next V^T tile (%[so2]) into ring slot `par`, the one this iteration read.  Unconditional: behind the last tile the pieces read
    rows at or past the sequence end (zero rows / zero fill past the buffer end), are never consumed, and cost three tiles per
    workgroup."""
    pairs = [(f"s_add_u32 m0, %[kdst], {par * 0x4000 + j * 4096}", f"buffer_load_dwordx4 %[kd{j}], %[kr], %[so] offen lds") for j in range(4)]
    pairs += [(f"s_add_u32 m0, %[vdst], {par * 0x4000 + j * 4096}", f"buffer_load_dwordx4 %[vd{j}], %[vr], %[so2] offen lds") for j in range(4)]
    post = ["s_add_u32 %[so], %[so], 0x4000", "s_add_u32 %[so2], %[so2], 128"]   # 64 keys x 256 B per K tile; 64 keys x 2 B per V^T row
    if "nodma" in ABLS:
        pairs = [(m0, "s_nop 0") for m0, _ in pairs]
    return pairs, post


def sync_block(par):
    """The whole hand-over without MFMAs to hide it (prologue only)."""
    pairs, post = sync_dma(par)
    L = sync_wait()
    for m0, ld in pairs:
        L += [m0, "s_nop 0", ld]
    return L + post


_uid = [0]
VALU_DELAY = int(os.environ.get("X2I_ATTN_VDELAY", "6"))      # MFMAs at the head of an iteration that carry no softmax instruction: the scores the softmax reads were written by the
                    # previous iteration's last MFMAs, and nothing interlocks a VALU read against an MFMA still in the pipe


class Stream:
    """Fragment bookkeeping of one unit list: ring slots are dealt to the units in order; reads are issued ahead of their use, and
    the uses wait in PAIRS (the even fragment waits for itself and its successor: one s_waitcnt per four MFMAs)."""

    def __init__(self, us, preread, par):
        self.us, self.par = us, par
        self.q = list(range(min(preread, len(us))))      # outstanding reads (unit indices), oldest first
        self.next = min(preread, len(us))                # next unit to read

    def read_upto_unit(self, L, unit_limit):
        while self.next < min(unit_limit, len(self.us)):
            if "noread" not in ABLS:
                L.append(frag_read(self.us[self.next], self.next, self.par))
            self.q.append(self.next)
            self.next += 1
            assert len(self.q) <= 15

    def wait(self, L, k):
        if k not in self.q:
            return
        assert self.next > k
        tgt = k + 1 if (k % 2 == 0 and (k + 1) in self.q) else k
        pos = self.q.index(tgt)
        if "nolgk" not in ABLS:
            L.append(f"s_waitcnt lgkmcnt({len(self.q) - 1 - pos})")
        self.q = self.q[pos + 1:]

    def mfmas(self, k, s_dst, c_init):
        return unit_mfmas(self.us[k], k, s_dst, c_init)


def prereads(kind, par):
    us = unit_list(*kind)
    st = Stream(us, 0, par)
    L = []
    st.read_upto_unit(L, LEAD)
    return L


def issue_cost(line):
    """Relative issue time of one instruction beside the MFMAs (plain VALU = 3)."""
    op = line.split()[0]
    if op.startswith("v_exp"):
        return 5          # a transcendental issues at about 5/3 of a plain VALU
    if op.endswith(":") or op == "s_nop":
        return 0
    if op.startswith("s_"):
        return 2
    return 3


def iteration(st, has_qk, has_pv, masked, first, next_kinds, force_rescale=False):
    """The iteration with its softmax stream BALANCED against everything else the gaps carry: a dry run without the softmax gives
    the issue time of the reads / waits / DMA pieces behind each MFMA; the VALU stream then fills every gap up to a common level
    (water-filling), so that no gap outlasts its MFMA while others idle."""
    n2 = n_mfmas(unit_list(has_qk, has_pv))
    dry = _iteration(st, has_qk, has_pv, masked, first, next_kinds, None, True, force_rescale)
    others, g = [0] * (n2 + 1), 0
    for line in dry[0]:
        if line.startswith("v_mfma"):
            g += 1
            if g > n2:
                break
        elif g:
            others[g] += issue_cost(line)
        if line.startswith("s_branch") and g == n2:
            break
    total = dry[1]
    elig = [g for g in range(1, n2 + 1) if g > VALU_DELAY]
    targets = None
    if elig and total:
        lo, hi = 0.0, float(total + max(others) + 1)
        for _ in range(60):
            c = (lo + hi) / 2
            if sum(max(0.0, c - others[g]) for g in elig) < total:
                lo = c
            else:
                hi = c
        targets, acc = [0] * (n2 + 1), 0.0
        for g in range(1, n2 + 1):
            if g in elig:
                acc += max(0.0, hi - others[g])
            targets[g] = int(acc + 0.5)
    return _iteration(st, has_qk, has_pv, masked, first, next_kinds, targets, False, force_rescale)[0]


def _iteration(st, has_qk, has_pv, masked, first, next_kinds, targets, dry, force_rescale=False):
    """One pipelined iteration i (st = i & 1): softmax of score set `st`, S(i+1) into set st ^ 1, O += V(i-1) P(i-1).  K(i+1) and
    V(i-1) sit in ring slot st ^ 1.  The first LEAD fragments arrive pre-read (ring slots 0 .. LEAD-1).  The iteration ends with the
    ring hand-over and the pre-reads of the next iteration; `next_kinds` = [(conditional, (has_qk, has_pv), label)]: the first entry
    is taken when %[cnt] == 0."""
    _uid[0] += 1
    uid = _uid[0]
    us = unit_list(has_qk, has_pv)
    n = len(us)
    par = st ^ 1
    p1 = softmax_pass1(st, masked, first, uid)
    p2 = softmax_pass2(st)
    L = []
    if first or n == 0:
        # no MFMA can cover the first tile's maxima: S(1)'s first MFMAs read NEGM, which this pass writes
        L += ["s_nop 15", "s_nop 15"] + [x for e in p1 for x in (e if isinstance(e, list) else [e])] + ["s_nop 1"]
        p1 = []
    va = p1 + p2
    total_cost = sum(3 * min(len(e), 8) if isinstance(e, list) else issue_cost(e[2] if isinstance(e, tuple) else e) for e in va)
    if "novalu" in ABLS or dry:      # (novalu: measurement only: the MFMA / LDS / DMA stream alone)
        va = []
    decide_at = max([k for k, e in enumerate(va) if isinstance(e, list)], default=-1)
    vi = 0

    def text(e):
        return e[2] if isinstance(e, tuple) else e

    def cost(e):
        return 3 * min(len(e), 8) if isinstance(e, list) else issue_cost(text(e))

    def target(mf_):
        if targets is not None:
            return targets[min(mf_, len(targets) - 1)]
        return (mf_ - VALU_DELAY) * total_cost // gaps

    spent = [0]
    pv_left = {g: (8 if has_pv else 0) for g in range(2)}     # V units of span g not yet issued: P[g] may not be overwritten before

    def emit_next():
        nonlocal vi
        e = va[vi]
        if isinstance(e, tuple) and pv_left[e[1]] > 0:
            return False
        spent[0] += cost(e)
        if isinstance(e, list):
            L.extend(e)
        else:
            L.append(text(e))
        vi += 1
        return True

    def fill_to(target_cost):
        while vi < len(va) and spent[0] + cost(va[vi]) <= target_cost:
            if not emit_next():
                break

    split = max(0, n - LEAD)              # units in front of the hand-over
    if has_pv and not has_qk:
        split = n                         # (last tile: every PV unit in front of it -- the conversions of P wait for their group's reads)
    early = max(0, split - 3)             # from this unit on, every remaining read of the iteration is issued at once: the hand-over's
                                          # lgkmcnt(0) then finds them returned instead of exposing one LDS round trip per iteration
    gaps = max(1, n_mfmas(us) - VALU_DELAY)     # the softmax stream runs over the whole iteration, the MFMAs behind the hand-over included
    sm = Stream(us, LEAD, par)
    mf = 0
    for k in range(split):
        sm.read_upto_unit(L, n if k >= early else k + LEAD + 1)
        sm.wait(L, k)
        if us[k][0] == "K" and us[k][1] == 0 and not first:
            while vi <= decide_at:        # the first QK^T MFMAs read NEGM: the defer-max decision must be behind us
                assert emit_next()
        for m in sm.mfmas(k, st ^ 1, True):
            L.append(m)
            mf += 1
            if mf > VALU_DELAY:
                fill_to(target(mf))
        if us[k][0] == "V":
            pv_left[us[k][1]] -= 1
    assert all(v == 0 for v in pv_left.values()), "PV units behind the hand-over"
    sm.read_upto_unit(L, n)
    vi_split, spent_split, mf_split = vi, spent[0], mf
    for ci, (cond, nxt, label) in enumerate(next_kinds):
        vi, spent[0], mf = vi_split, spent_split, mf_split       # (each successor's copy of the tail carries the same rest of the stream)
        if cond:
            L += ["s_cmp_lg_u32 %[cnt], 0", f"s_cbranch_scc1 .Lalt{uid}_%="]
        L += sync_wait()                    # (lgkmcnt(0) inside: every fragment of this iteration is in registers)
        L += prereads(nxt, st)              # the next iteration reads ring slot st
        pairs, post = sync_dma(par)
        tail_m = [m for k in range(split, n) for m in sm.mfmas(k, st ^ 1, True)]
        # pieces between the remaining MFMAs: M0 write, an MFMA (or a nop) in between, the piece
        pi = 0
        L.append(pairs[0][0])
        for mi, m in enumerate(tail_m):
            L.append(m)
            mf += 1
            fill_to(target(mf))
            share = (mi + 1) * len(pairs) // max(1, len(tail_m)) - pi
            for _ in range(share):
                L.append(pairs[pi][1])
                pi += 1
                if pi < len(pairs):
                    L.append(pairs[pi][0])
                    if _ + 1 < share:
                        L.append("s_nop 0")
        while pi < len(pairs):
            L += ["s_nop 0", pairs[pi][1]]
            pi += 1
            if pi < len(pairs):
                L.append(pairs[pi][0])
        L += post
        while vi < len(va):
            assert emit_next(), "a P group is still being read"
        if has_pv or force_rescale:
            L += o_rescale(f"{uid}x{ci}")
        L.append(f"s_branch {label}")
        if cond:
            L.append(f".Lalt{uid}_%=:")
    return L, total_cost


def solo(kind, s_dst, par, preread, c_init=False):
    """A unit list on its own (prologue S(0) with C = 0, tail PV): reads LEAD ahead, no VALU stream."""
    us = unit_list(*kind)
    sm = Stream(us, LEAD if preread else 0, par)
    L = []
    if not preread:
        sm.read_upto_unit(L, LEAD)
    for k in range(len(us)):
        sm.read_upto_unit(L, k + LEAD + 1)
        sm.wait(L, k)
        L += sm.mfmas(k, s_dst, c_init)
    return L


STATE_PIECE = 4096      # bytes per state load / store instruction of a workgroup (256 lanes x 16 B)


def state_io(op, regs):
    """One 16-byte piece per lane of the hand-over state (%[sr] = the slab's descriptor, %[sto] = lane * 16, %[so] = running piece offset)."""
    return [f"buffer_{op}_dwordx4 v[{regs}:{regs + 3}], %[sto], %[sr], %[so] offen sc1", f"s_add_u32 %[so], %[so], {STATE_PIECE}"]   # (write-through stores, loads past the L1: the stream-K slabs' protocol, gen_gemm256w.py)


def prologue(cont=False):
    """cont = False: a work item from its first key tile (%[so0] = 0) or from any tile (the offsets are inputs).  cont = True (stream-K, the closing part of
    an item cut along the key axis): O, -m and the row sums arrive from the slab the opening part left (state layout: 32 pieces of O -- accumulator
    registers 4 k .. 4 k + 3 --, one of -m, one of l), S(t0) starts from C = -m like every later tile's, and the first iteration is an ordinary
    (non-first) softmax pass without a PV product."""
    L = ["s_nop 4"]
    L += [f"v_mov_b32 v{ONES + k}, 0x3f803f80" for k in range(4)]
    for qb in range(4):
        L += [f"v_mov_b32 v{ALPHA + qb}, 1.0"]
    L += ["s_mov_b32 %[fl], 0"]
    # Q fragments (second operand of S^T = K Q^T): lane holds Q[q0 + 16 qb + c][32 ds + 8 g .. + 8]
    PL = []
    for qb in range(4):
        for ds in range(4):
            b = QTMP + (qb * 4 + ds) * 4
            PL.append(f"global_load_dwordx4 v[{b}:{b + 3}], %[qo{qb}], %[qp] offset:{ds * 64}")
    # K(t0) -> slot 0, K(t0 + 1) -> slot 1 (rows behind Spad read as zero)
    PL += ["s_mov_b32 %[so], %[so0]"]
    for j in range(4):
        PL += [f"s_add_u32 m0, %[kdst], {j * 4096}", "s_nop 0", f"buffer_load_dwordx4 %[kd{j}], %[kr], %[so] offen lds"]
    PL += ["s_add_u32 %[so], %[so0], 0x4000"]
    for j in range(4):
        PL += [f"s_add_u32 m0, %[kdst], {0x4000 + j * 4096}", "s_nop 0", f"buffer_load_dwordx4 %[kd{j}], %[kr], %[so] offen lds"]
    if "noprold" in ABLS:      # measurement only (tools/attn_item_parts_build.sh): no Q loads, no first K tiles -- what the prologue's memory latency costs an item
        PL = [x for x in PL if not (x.startswith("global_load") or x.startswith("buffer_load"))]
    if not cont:
        # %[pre] (persistent form): the previous unit's exit already requested this item's Q block (into the LDS area behind the rings, K-tile image per
        # wave) and its first two K tiles -- behind that unit's epilogue, whose final wait covered them
        L += ["s_cmp_lg_u32 %[pre], 0", "s_cbranch_scc1 .Lpre_%="]
    L += PL
    if cont:
        # the slab: O pieces 0 .. 15 through score set 0 (the raw Q fragments sit in set 1), -m and l through the temporaries
        L += ["s_mov_b32 %[so], 0"]
        for k in range(16):
            L += state_io("load", SA + 4 * k)
        L += ["s_mov_b32 %[so], " + str(32 * STATE_PIECE)]
        L += state_io("load", TMP) + state_io("load", TMP + 4)
        L += ["s_waitcnt vmcnt(0)"]
        for qb in range(4):
            L += [f"v_mov_b32 {NM(qb, r)}, {T(qb)}" for r in range(4)]
            L += [f"v_mov_b32 {LA(qb, r)}, {T(4 + qb)}" for r in range(4)]
        L += [f"v_accvgpr_write_b32 a{OA + i}, v{SA + i}" for i in range(64)]
    else:
        # O = 0 under the flight of Q and the first K tiles (in front of the loads it was 512 cycles before the first request left)
        for db in range(8):
            for qb in range(4):
                L += [f"v_accvgpr_write_b32 {O(db, qb, r)}, 0" for r in range(4)]
        L += ["s_waitcnt vmcnt(8)"]
    # under the K flight: Q fragments into the accumulator file.  %[pres] == 0: Q already carries scale * log2 e (x2i_qkv_desc.q_scale);
    # otherwise Q~ = bf16(Q * scale * log2 e) here (a second rounding of Q).  Either way p = exp2(s') with s' = q~ . k - m
    sfx = "c" if cont else ""
    L += ["s_cmp_lg_u32 %[pres], 0", f"s_cbranch_scc1 .Lpres{sfx}_%="]
    L += [f"v_accvgpr_write_b32 a{QF + i}, v{QTMP + i}" for i in range(64)]
    L += [f"s_branch .Lqdone{sfx}_%=", f".Lpres{sfx}_%=:"]
    for i in range(64):
        L += [f"v_lshlrev_b32 {T(0)}, 16, v{QTMP + i}", f"v_and_b32 {T(1)}, 0xffff0000, v{QTMP + i}",
              f"v_mul_f32 {T(0)}, %[sc], {T(0)}", f"v_mul_f32 {T(1)}, %[sc], {T(1)}",
              f"v_cvt_pk_bf16_f32 {T(0)}, {T(0)}, {T(1)}", f"v_accvgpr_write_b32 a{QF + i}, {T(0)}"]
    if not cont:
        L += ["s_branch .Lqdone_%=", ".Lpre_%=:"]
        for db in range(8):
            for qb in range(4):
                L += [f"v_accvgpr_write_b32 {O(db, qb, r)}, 0" for r in range(4)]
        L += ["s_barrier"]                                 # every wave's pieces of the Q image have landed (each waited for its own at the previous exit)
        L += [f"v_add_u32 {T(ds)}, %[qdel], %[ka{ds}]" for ds in range(4)]
        for qb in range(4):
            for ds in range(4):
                b = QF + (qb * 4 + ds) * 4
                L.append(f"ds_read_b128 a[{b}:{b + 3}], {T(ds)} offset:{qb * 4096}")
        L += ["s_waitcnt lgkmcnt(0)"]
    L += [f".Lqdone{sfx}_%=:"]
    if cont:
        # O pieces 16 .. 31 (accumulator registers 64 .. 127) through score set 1, now that the raw Q fragments have left it
        L += ["s_mov_b32 %[so], " + str(16 * STATE_PIECE)]
        for k in range(16):
            L += state_io("load", SA + 64 + 4 * k)
        L += ["s_waitcnt vmcnt(0)"]
        L += [f"v_accvgpr_write_b32 a{OA + 64 + i}, v{SA + 64 + i}" for i in range(64)]
    L += ["s_add_u32 %[so], %[so0], 0x8000", "s_mov_b32 %[so2], %[so20]"]
    L += ["s_waitcnt vmcnt(0)", "s_barrier"]
    # S(t0) = K(t0) Q~^T alone (score set 0; raw for a fresh item: the first softmax subtracts its maxima itself; with the running maximum
    # subtracted, like every other tile's scores, for a continued one)
    L += solo((True, False), 0, 0, False, c_init=cont)
    return L


def next_item_requests():
    """%[nxt] (persistent form): in FRONT of the epilogue, request the next work item's Q block (%[nqr]: 256 rows; as four K-tile images, one per wave's 64
    rows, into the LDS area at %[qdst]) and its first two K tiles (%[nkr], %[nso0]; ring slots 0 / 1): the epilogue's ~3 us of conversions and stores hide
    the ~3.7 us these loads cost an item when its prologue has to wait for them (tools/attn_item_cost.py on the noepi / noprold builds).  The stale
    pieces behind the last tile (sync_dma) must have landed in every wave before a ring slot is written again: wait + barrier first."""
    L = ["s_cmp_lg_u32 %[nxt], 0", "s_cbranch_scc0 .Lnonxt_%=", "s_waitcnt vmcnt(0)", "s_barrier"]
    for t in range(4):
        L += [f"s_mov_b32 %[so], {t * 16384}"]
        for j in range(4):
            L += [f"s_add_u32 m0, %[qdst], {t * 16384 + j * 4096}", "s_nop 0", f"buffer_load_dwordx4 %[kd{j}], %[nqr], %[so] offen lds"]
    L += ["s_mov_b32 %[so], %[nso0]"]
    for j in range(4):
        L += [f"s_add_u32 m0, %[kdst], {j * 4096}", "s_nop 0", f"buffer_load_dwordx4 %[kd{j}], %[nkr], %[so] offen lds"]
    L += ["s_add_u32 %[so], %[nso0], 0x4000"]
    for j in range(4):
        L += [f"s_add_u32 m0, %[kdst], {0x4000 + j * 4096}", "s_nop 0", f"buffer_load_dwordx4 %[kd{j}], %[nkr], %[so] offen lds"]
    L += [".Lnonxt_%=:"]
    return L


def state_store():
    """Stream-K hand-over (the opening part of an item cut along the key axis): the un-normalised O, -m and the row sums into the slab instead of
    the epilogue; the closing part continues from them, so every row is summed in exactly the order of an undivided item."""
    L = ["s_nop 15", "s_nop 15", "s_mov_b32 %[so], 0"]
    for k in range(32):
        L += [f"v_accvgpr_read_b32 v{SA + 4 * k + r}, a{OA + 4 * k + r}" for r in range(4)]
    L += ["s_nop 1"]
    for k in range(32):
        L += state_io("store", SA + 4 * k)
    for qb in range(4):
        L += [f"v_mov_b32 {T(qb)}, {NM(qb, 0)}", f"v_mov_b32 {T(4 + qb)}, {LA(qb, 0)}"]
    L += ["s_nop 1"]
    L += state_io("store", TMP) + state_io("store", TMP + 4)
    L += ["s_waitcnt vmcnt(0)"]
    return L


def epilogue():
    """O[q][d] = O^T[d][q] / l: lane (query c of block qb; g) holds d = 16 db + 4 g + r.  Two d-blocks at a time: v_permlane16_swap gives the
    even-g lanes their odd neighbour's four values of the first block and the odd-g lanes their even neighbour's four of the second, i.e.
    eight consecutive d = one 16-byte store per lane (%[oo] carries the lane's share of the address: + 0 / 32 / 16 / 48 bytes for g = 0..3);
    log2-sum-exp rows on request."""
    L = ["s_nop 15", "s_nop 15"]
    for qb in range(4):   # row sums: every register of L[qb] holds the sum over ALL keys for query c
        L += [f"v_mov_b32 v{MX + qb}, {LA(qb, 0)}"]
    for qb in range(4):
        L += [f"v_rcp_f32 v{ALPHA + qb}, v{MX + qb}"]
    # log2-sum-exp (x2i_attention_lse_bf16): m_run + log2(l) for q < S, +1e30 on the padding rows; lanes with g == 0 store
    L += ["s_cmp_lg_u32 %[lsef], 0", "s_cbranch_scc0 .Lnolse_%="]
    for qb in range(4):
        L += [f"v_log_f32 {T(4 + qb)}, v{MX + qb}"]
    L += ["s_nop 1"]
    for qb in range(4):
        L += [f"v_sub_f32 {T(4 + qb)}, {T(4 + qb)}, {NM(qb, 0)}",
              f"v_add_u32 {T(8)}, {16 * qb}, %[qv]",
              f"v_cmp_gt_i32 vcc, %[sS], {T(8)}",                          # q < S
              f"v_mov_b32 {T(9)}, 0x7149f2ca",                             # 1.0e30f
              f"v_cndmask_b32 {T(4 + qb)}, {T(9)}, {T(4 + qb)}, vcc",
              f"v_cmp_gt_i32 vcc, %[sSp], {T(8)}",                         # q < Spad ...
              f"v_cmp_eq_u32 %[cnd], 0, %[hi]",                            # ... and g == 0
              "s_nop 3", "s_and_b64 vcc, vcc, %[cnd]", "s_and_saveexec_b64 %[exs], vcc",
              f"global_store_dword %[lo], {T(4 + qb)}, %[lp] offset:{64 * qb}",
              "s_mov_b64 exec, %[exs]"]
    L += [".Lnolse_%=:"]
    for qb in range(4):
        L += [f"v_add_u32 v{MX2}, {16 * qb}, %[qv]", f"v_cmp_gt_i32 vcc, %[sS], v{MX2}", "s_nop 3", "s_and_saveexec_b64 %[exs], vcc"]
        if qb:
            L += [f"v_add_u32 %[oo], %[ostep], %[oo]"]
        for dp in range(4):                                                # d-blocks 2 dp (X) and 2 dp + 1 (Y)
            L += [f"v_accvgpr_read_b32 {T(k)}, {O(2 * dp, qb, k)}" for k in range(4)]
            L += [f"v_accvgpr_read_b32 {T(4 + k)}, {O(2 * dp + 1, qb, k)}" for k in range(4)]
            L += [f"v_mul_f32 {T(k)}, {T(k)}, v{ALPHA + qb}" for k in range(8)]
            # A = (T8, T9) <- X pairs, B = (T10, T11) <- Y pairs; swap: odd rows of A <-> even rows of B; the store takes A0 A1 B0 B1
            L += [f"v_cvt_pk_bf16_f32 {T(8)}, {T(0)}, {T(1)}", f"v_cvt_pk_bf16_f32 {T(9)}, {T(2)}, {T(3)}",
                  f"v_cvt_pk_bf16_f32 {T(10)}, {T(4)}, {T(5)}", f"v_cvt_pk_bf16_f32 {T(11)}, {T(6)}, {T(7)}", "s_nop 1",
                  f"v_permlane16_swap_b32 {T(8)}, {T(10)}", f"v_permlane16_swap_b32 {T(9)}, {T(11)}", "s_nop 1",
                  f"global_store_dwordx4 %[oo], v[{TMP + 8}:{TMP + 11}], %[op] offset:{dp * 64}", "s_nop 1"]
        L += ["s_mov_b64 exec, %[exs]"]
    L += ["s_waitcnt vmcnt(0)"]
    return L


def build():
    """Program (nt = number of 64-key tiles; %[cnt] = mid iterations left = nt - 2):
         prologue: Q, K(0), K(1); S(0) = K(0) Q^T (score set 0)
         i = 0       FIRST : softmax(0)                  + S(1)                          (nt > 1)
         i = 1..nt-2 MID   : softmax(i)                  + S(i+1) + O += V(i-1) P(i-1)    (both set parities are emitted)
         i = nt-1    LAST  : softmax(nt-1), masked                + O += V(nt-2) P(nt-2)
         tail              :                                        O += V(nt-1) P(nt-1)
         (nt == 1: ONLY = softmax(0) masked, then the tail)"""
    MIDK, LASTK, TAILK = (True, True), (False, True), (False, True)
    L = ["s_cmp_lg_u32 %[cont], 0", "s_cbranch_scc1 .Lcont_%="]
    L += prologue()
    # hand-over behind S(0): frees K slot 0, fetches K(2) / V(0); the fragment addresses now point at the slots of tile 1
    L += sync_block(0)
    L += ["s_cmp_eq_u32 %[nt], 1", "s_cbranch_scc1 .Lonly_%="]
    L += prereads((True, False), 1)
    # FIRST (score set 0): successor LAST(set 1) when nt == 2 (%[cnt] == 0), else MID(set 1)
    L += iteration(0, True, False, False, True, [(True, LASTK, ".Llast1_%="), (False, MIDK, ".Lmid1_%=")])
    for par in (1, 0):
        o = par ^ 1
        L += [f".Lmid{par}_%=:", "s_sub_u32 %[cnt], %[cnt], 1"]
        L += iteration(par, True, True, False, False, [(True, LASTK, f".Llast{o}_%="), (False, MIDK, f".Lmid{o}_%=")])
    for par in (1, 0):
        L += [f".Llast{par}_%=:"]
        L += iteration(par, False, True, True, False, [(False, TAILK, f".Ltail{par}_%=")])
    # ONLY (nt == 1): softmax(0) masked, then the hand-over that waits for V(0)
    L += [".Lonly_%=:"]
    L += iteration(0, False, False, True, True, [(False, TAILK, ".Ltail0_%=")])
    # CONT (stream-K; nt >= 2, the launcher's cuts keep it so): the closing part of a cut item -- its own prologue, then the first tile as an
    # ordinary softmax pass (the rare rescale of the loaded O included) + S(t0 + 1), and on into the common MID / LAST code
    L += [".Lcont_%=:"] + prologue(cont=True)
    L += sync_block(0)
    L += prereads((True, False), 1)
    L += iteration(0, True, False, False, False, [(True, LASTK, ".Llast1_%="), (False, MIDK, ".Lmid1_%=")], force_rescale=True)
    for par in (1, 0):
        L += [f".Ltail{par}_%=:"] + solo((False, True), 0, par, True) + ["s_branch .Lepi_%="]
    L += [".Lepi_%=:", "s_cmp_lg_u32 %[hand], 0", "s_cbranch_scc1 .Lhand_%="] + next_item_requests() + ([] if "noepi" in ABLS else epilogue()) + ["s_branch .Lend_%=", ".Lhand_%=:"] + state_store() + [".Lend_%=:"]   # (noepi: measurement only)
    return L


def main():
    out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "attn_w16_loop.inc")
    L = build()
    txt = ["// GENERATED by gen_attn_w16.py -- do not edit; register map and schedule live in the generator.",
           f"// register map: scores v{SA}..v{SA + 127}, P fragments v{PF}..v{PF + 31}, -max copies v{NEGM}..v{NEGM + 15}, softmax state up to v{_v - 1};"
           f" O a{OA}..a{OA + 127}, Q fragments a{QF}..a{QF + 63}, fragment ring a{FR}..a{FR + RING * 4 - 1}",
           f"// {len(L)} lines", "#define X2I_ATTN_W16_TEXT \\"]
    txt += [f'  "{l}\\n" \\' for l in L[:-1]]
    txt.append(f'  "{L[-1]}\\n"')
    clob = ", ".join(f'"v{i}"' for i in range(32, 256)) + ", " + ", ".join(f'"a{i}"' for i in range(256))
    txt.append(f"#define X2I_ATTN_W16_CLOBBERS {clob}")
    txt.append("")
    data = "\n".join(txt)
    if "--check" in sys.argv:
        cur = open(out).read() if os.path.exists(out) else ""
        sys.exit(0 if cur == data else 1)
    with open(out, "w") as fh:
        fh.write(data)
    print(f"wrote {out}: {len(L)} lines, v up to {_v - 1}, a up to {_a - 1}")


if __name__ == "__main__":
    main()
