#!/usr/bin/env python3
"""Generator of the hand-scheduled flash-attention forward (attention_w4.hip) -> attn_w4_loop.inc.

Organisation (head_dim 128, bf16, exp2-domain online softmax with defer-max -- the math of attention.hip): ONE wave per SIMD, four
waves per workgroup, 64 query rows per wave (two 32-row blocks qb), 64-key tiles.  Per tile and wave 64 `v_mfma_f32_32x32x16_bf16`
(2048 matrix-pipe cycles): 32 for S^T = K Q^T and 32 for O^T += V^T P^T.  The stream is software-pipelined over THREE tiles so that
the matrix pipe never waits for the vector unit -- iteration i issues the MFMAs of S(i+1) and of the PV product of tile i-1 while the
VALU computes the softmax of tile i in the gaps between them (4-5 other instructions per MFMA):

    MFMA   S(i+1) = K(i+1) Q^T     O += V(i-1) P(i-1)        (independent of the softmax in flight)
    VALU   softmax(i): row max, defer-max decision, p = exp2(s * c - m), row sums, P(i) as bf16 pairs
    LDS    fragments of K(i+1) / V(i-1): one ds_read_b128 per two MFMAs, LEAD units ahead, straight into accumulator registers
    DMA    K(i+3), V(i+1) into the ring slots iteration i has finished reading (2-deep rings, one barrier per iteration)

Register files are laid out by hand (an asm operand cannot be indexed into a 16-register tuple): O accumulators (128), the Q
fragments (64) and the K / V fragment ring (64) live in the ACCUMULATOR file -- MFMA A / B operands may come from there, `ds_read` and
`global_load` can target it -- which leaves the 256 architectural VGPRs for two score sets (128), two P sets (64) and the softmax
state.  The compiler only sees the statement's operands (v0..v31) and a clobber list; prologue, loop and epilogue (normalise, pack,
half-wave exchange, 16-byte stores, log-sum-exp) are all inside the one statement.

`python gen_attn_w4.py` rewrites attn_w4_loop.inc (committed; tests/test_host_cpu.py regenerates and compares).
"""
import os
import sys

LEAD = int(os.environ.get("X2I_ATTN_LEAD", "3"))   # fragment reads are issued this many units (= one fragment, two MFMAs) ahead of their use
RING = 8            # fragment ring slots (4 accumulator registers each)
NEG_BIG = "0xf149f2ca"   # -1.0e30f

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


SA = valloc(128, 16)      # S[set][u][qb][r]
PF = valloc(64, 4)        # P[set][u][kt][qb][w]
TMP = valloc(12, 4)
M_RUN, L_RUN, M_USE, MX, MX2, ALPHA, PSUM = (valloc(2) for _ in range(7))
NEGBIG = valloc(1)
ONES = valloc(4, 4)       # bf16 1.0 x 8: the A operand that makes the matrix pipe produce the row sums of P
OA = aalloc(128, 16)      # O[db][qb][r]
LA = aalloc(32, 16)       # L[qb][r]: row sums of P (every register of a lane holds the sum of ITS query row), scaled like O
QF = aalloc(64, 4)        # Q[qb][ds][w]
FR = aalloc(RING * 4, 4)  # fragment ring


def S(st, u, qb, r=None):
    b = SA + ((st * 2 + u) * 2 + qb) * 16
    return f"v[{b}:{b + 15}]" if r is None else f"v{b + r}"


def P(st, u, kt, qb, w=None):
    b = PF + (((st * 2 + u) * 2 + kt) * 2 + qb) * 4
    return f"v[{b}:{b + 3}]" if w is None else f"v{b + w}"


def O(db, qb, r=None):
    b = OA + (db * 2 + qb) * 16
    return f"a[{b}:{b + 15}]" if r is None else f"a{b + r}"


def LS(qb, r=None):
    b = LA + qb * 16
    return f"a[{b}:{b + 15}]" if r is None else f"a{b + r}"


def Q(qb, ds):
    b = QF + (qb * 8 + ds) * 4
    return f"a[{b}:{b + 3}]"


def F(slot):
    b = FR + (slot % RING) * 4
    return f"a[{b}:{b + 3}]"


def T(i):
    return f"v{TMP + i}"


# ------------------------------------------------------------------------------------------------ softmax of one tile (VALU stream)
def softmax_stream(st, masked):
    """VALU instructions for the softmax of score set `st` (both 32-row query blocks of the wave): P fragments of set `st`, m_run / l_run;
    raises %[fl] when a row maximum grew by more than the defer-max threshold (O is then rescaled at the end of the iteration)."""
    L = []
    if masked:
        # keys at or behind the sequence end.  Lane holds (tile-relative) key u*32 + 16*(r >> 3) + 8*hi + (r & 7); %[lim] = S - kv0 - 8*hi
        for u in range(2):
            for r in range(16):
                kidx = u * 32 + 16 * (r >> 3) + (r & 7)
                L.append(f"v_cmp_lt_i32 vcc, {kidx}, %[lim]")
                for qb in range(2):
                    L.append(f"v_cndmask_b32 {S(st, u, qb, r)}, v{NEGBIG}, {S(st, u, qb, r)}, vcc")
    for qb in range(2):
        vals = [S(st, u, qb, r) for u in range(2) for r in range(16)]
        t = [T(qb * 4 + k) for k in range(4)]
        for k in range(4):
            L.append(f"v_max3_f32 {t[k]}, {vals[3 * k]}, {vals[3 * k + 1]}, {vals[3 * k + 2]}")
        for k in range(4):
            base = 12 + 4 * k
            L.append(f"v_max3_f32 {t[k]}, {t[k]}, {vals[base]}, {vals[base + 1]}")
            L.append(f"v_max3_f32 {t[k]}, {t[k]}, {vals[base + 2]}, {vals[base + 3]}")
        L.append(f"v_max3_f32 {t[0]}, {t[0]}, {vals[28]}, {vals[29]}")
        L.append(f"v_max3_f32 {t[1]}, {t[1]}, {vals[30]}, {vals[31]}")
        L.append(f"v_max3_f32 {t[0]}, {t[0]}, {t[1]}, {t[2]}")
        L.append(f"v_max_f32 v{MX + qb}, {t[0]}, {t[3]}")
        L.append(f"v_mov_b32 v{MX2 + qb}, v{MX + qb}")
    L.append("s_nop 1")                                                       # VALU write -> v_permlane32_swap read
    for qb in range(2):
        L.append(f"v_permlane32_swap_b32 v{MX + qb}, v{MX2 + qb}")             # the other 32 keys of a query row live in lane ^ 32
    for qb in range(2):
        L.append(f"v_max_f32 v{MX + qb}, v{MX + qb}, v{MX2 + qb}")
        L.append(f"v_mul_f32 v{MX + qb}, %[sc], v{MX + qb}")                   # exp2 domain
        L.append(f"v_max_f32 v{MX + qb}, v{M_RUN + qb}, v{MX + qb}")          # candidate new running max
        L.append(f"v_sub_f32 v{MX2 + qb}, v{MX + qb}, v{M_RUN + qb}")
    L.append(f"v_max_f32 v{MX2}, v{MX2}, v{MX2 + 1}")
    L.append(f"v_cmp_lt_f32 %[cnd], %[thr], v{MX2}")                             # some row of this wave grew by more than THR = 8 ?
    L.append(("DECIDE",))
    # p = exp2(s * c - m); bf16 pairs -> P fragment (sub-tile u, k-step kt) = registers 8kt .. 8kt+7 of S[u].  The row sums come from
    # the matrix pipe (ones x P^T beside the PV products).  The three steps of a value are issued as a software pipeline -- fma(k),
    # exp(k - 2), cvt of the pair behind (k - 4) -- so that the quarter-rate exponentials are spread evenly over the MFMA gaps
    # instead of arriving eight in a row (a burst of transcendentals outlasts the gap and starves the matrix pipe)
    seq = []
    for u in range(2):
        for kt in range(2):
            for qb in range(2):
                for k in range(8):
                    seq.append((S(st, u, qb, kt * 8 + k), qb, P(st, u, kt, qb, k // 2) if k & 1 else None, S(st, u, qb, kt * 8 + k - 1) if k & 1 else None))
    n = len(seq)
    for k in range(n + 4):
        if k < n:
            L.append(f"v_fma_f32 {seq[k][0]}, {seq[k][0]}, %[sc], -v{M_USE + seq[k][1]}")
        if 0 <= k - 2 < n:
            L.append(f"v_exp_f32 {seq[k - 2][0]}, {seq[k - 2][0]}")
        if 0 <= k - 4 < n and seq[k - 4][2] is not None:
            L.append(f"v_cvt_pk_bf16_f32 {seq[k - 4][2]}, {seq[k - 4][3]}, {seq[k - 4][0]}")
    return L


def decide(uid):
    """Scalar branch on the defer-max test.  Rare path: adopt the new maxima, alpha = exp2(m_old - m_new), scale l_run, raise the flag."""
    L = ["s_nop 3", "s_cmp_lg_u64 %[cnd], 0", f"s_cbranch_scc0 .Lkeep{uid}_%="]
    for qb in range(2):
        L += [f"v_sub_f32 v{ALPHA + qb}, v{M_RUN + qb}, v{MX + qb}", f"v_mov_b32 v{M_RUN + qb}, v{MX + qb}"]
    for qb in range(2):
        L += [f"v_exp_f32 v{ALPHA + qb}, v{ALPHA + qb}"]
    L += ["s_mov_b32 %[fl], 1", f".Lkeep{uid}_%=:"]
    for qb in range(2):
        L += [f"v_mov_b32 v{M_USE + qb}, v{M_RUN + qb}"]
    return L


def o_rescale(uid):
    """O *= alpha per query row when the flag is up; behind the last PV MFMA of the iteration."""
    L = ["s_cmp_lg_u32 %[fl], 0", f"s_cbranch_scc0 .Lnors{uid}_%=", "s_nop 15", "s_nop 15"]
    for db in range(4):
        for qb in range(2):
            for r0 in range(0, 16, 4):
                L += [f"v_accvgpr_read_b32 {T(k)}, {O(db, qb, r0 + k)}" for k in range(4)]
                L += [f"v_mul_f32 {T(k)}, {T(k)}, v{ALPHA + qb}" for k in range(4)]
                L += [f"v_accvgpr_write_b32 {O(db, qb, r0 + k)}, {T(k)}" for k in range(4)]
    for qb in range(2):
        for r0 in range(0, 16, 4):
            L += [f"v_accvgpr_read_b32 {T(k)}, {LS(qb, r0 + k)}" for k in range(4)]
            L += [f"v_mul_f32 {T(k)}, {T(k)}, v{ALPHA + qb}" for k in range(4)]
            L += [f"v_accvgpr_write_b32 {LS(qb, r0 + k)}, {T(k)}" for k in range(4)]
    L += ["s_mov_b32 %[fl], 0", "s_nop 3", f".Lnors{uid}_%=:"]
    return L


# ------------------------------------------------------------------------------------------------ MFMA / LDS stream
def unit_list(has_qk, has_pv):
    """Fragment-sized units of an iteration, K and V units alternating.  ("K", ds, u): S[u][qb] += K(u, ds) Q[qb][ds];
    ("V", g, db) with g = (u, kt): O[db][qb] += V^T(db, u, kt) P[u][kt][qb]."""
    ku = [("K", ds, u) for ds in range(8) for u in range(2)] if has_qk else []
    vu = [("V", g, db) for g in range(4) for db in range(4)] if has_pv else []
    out = []
    for i in range(16):
        if ku:
            out.append(ku[i])
        if vu:
            out.append(vu[i])
            if i % 4 == 3:
                out.append(("L", i // 4, 0))   # L[qb] += ones x P^T(u, kt): no fragment to read
    return out


def frag_read(unit, slot):
    kind, a, b = unit
    assert kind != "L"
    if kind == "K":   # d-step ds = a (address register per ds: the swizzle is an XOR), sub-tile u = b (+ 32 rows x 256 B)
        return f"ds_read_b128 {F(slot)}, %[ka{a}]" + (f" offset:{b * 8192}" if b else "")
    return f"ds_read_b128 {F(slot)}, %[va{a}]" + (f" offset:{b * 4096}" if b else "")   # (u, kt) = a, d-block db = b (+ 32 rows x 128 B)


def unit_mfmas(unit, slot, s_dst, p_src):
    kind, a, b = unit
    if kind == "K":
        ds, u = a, b
        return [f"v_mfma_f32_32x32x16_bf16 {S(s_dst, u, qb)}, {F(slot)}, {Q(qb, ds)}, " + ("0" if ds == 0 else S(s_dst, u, qb)) for qb in range(2)]
    if kind == "L":
        g = a
        return [f"v_mfma_f32_32x32x16_bf16 {LS(qb)}, v[{ONES}:{ONES + 3}], {P(p_src, g >> 1, g & 1, qb)}, {LS(qb)}" for qb in range(2)]
    g, db = a, b
    return [f"v_mfma_f32_32x32x16_bf16 {O(db, qb)}, {F(slot)}, {P(p_src, g >> 1, g & 1, qb)}, {O(db, qb)}" for qb in range(2)]


def sync_wait():
    """Ring hand-over, part 1: every fragment read of this iteration has been issued (and had time to return) above.  Wait for them
    and for this wave's pieces of the tiles the NEXT iteration reads, barrier; the slots this iteration read are free: flip the
    fragment addresses to the other slots."""
    L = ["s_waitcnt vmcnt(0) lgkmcnt(0)", "s_barrier"]
    if os.environ.get("X2I_ATTN_ABL") == "nobar":      # measurement only (wrong results): what the per-tile barrier costs
        L = ["s_waitcnt vmcnt(0) lgkmcnt(0)"]
    if os.environ.get("X2I_ATTN_ABL") == "nosync":     # measurement only: neither the waits nor the barrier
        L = []
    L += [f"v_xor_b32 %[ka{ds}], 0x4000, %[ka{ds}]" for ds in range(8)]
    L += [f"v_xor_b32 %[va{g}], 0x4000, %[va{g}]" for g in range(4)]
    return L


def sync_dma():
    """Part 2, as (M0 write, piece) pairs to be spread between the iteration's last MFMAs: K(tile %[tk]) / V(tile %[tv]) into the freed
    slots.  Unconditional: behind the last tile the pieces read rows at or past the sequence end (zero rows / zero fill past the
    buffer end), are never consumed, and cost three tiles per workgroup."""
    pre = ["s_lshl_b32 %[so], %[tk], 14", "s_lshl_b32 %[so2], %[tv], 7"]     # 64 keys x 256 B per K tile; 64 keys x 2 B per V^T row
    pairs = [(f"s_add_u32 m0, %[kdst], {j * 4096}", f"buffer_load_dwordx4 %[kd{j}], %[kr], %[so] offen lds") for j in range(4)]
    pairs += [(f"s_add_u32 m0, %[vdst], {j * 4096}", f"buffer_load_dwordx4 %[vd{j}], %[vr], %[so2] offen lds") for j in range(4)]
    post = ["s_xor_b32 %[kdst], %[kdst], 0x4000", "s_xor_b32 %[vdst], %[vdst], 0x4000", "s_add_u32 %[tk], %[tk], 1", "s_add_u32 %[tv], %[tv], 1"]
    return pre, pairs, post


def sync_block(uid):
    """The whole hand-over without MFMAs to hide it (prologue only)."""
    pre, pairs, post = sync_dma()
    L = sync_wait() + pre
    for m0, ld in pairs:
        L += [m0, "s_nop 0", ld]
    return L + post


_uid = [0]
VALU_DELAY = 4      # MFMAs at the head of an iteration that carry no softmax instruction: the scores the softmax reads were written by the
                    # previous iteration's last MFMAs, and nothing interlocks a VALU read against an MFMA still in the pipe


class Stream:
    """Fragment bookkeeping of one unit list: ring slots are dealt to the units that read a fragment, in order; reads are issued
    ahead of their use and a use waits with lgkmcnt(number of younger reads)."""

    def __init__(self, us, preread):
        self.us = us
        self.fidx = []
        f = 0
        for u in us:
            self.fidx.append(f if u[0] != "L" else None)
            f += u[0] != "L"
        self.nfrag = f
        self.q = [i for i in range(min(preread, f))]     # outstanding reads (fragment indices), oldest first
        self.next = min(preread, f)                      # next fragment index to read
        self.funits = [k for k, u in enumerate(us) if u[0] != "L"]

    def read_upto_unit(self, L, unit_limit):
        """issue the reads of every fragment unit with unit index < unit_limit"""
        while self.next < self.nfrag and self.funits[self.next] < unit_limit:
            L.append(frag_read(self.us[self.funits[self.next]], self.next))
            self.q.append(self.next)
            self.next += 1

    def wait(self, L, k):
        f = self.fidx[k]
        if f is not None and f in self.q:
            pos = self.q.index(f)
            if os.environ.get("X2I_ATTN_ABL") != "nolgk":   # (measurement only: no waits on the fragment reads)
                L.append(f"s_waitcnt lgkmcnt({len(self.q) - 1 - pos})")
            self.q = self.q[pos + 1:]

    def mfmas(self, k, s_dst, p_src):
        return unit_mfmas(self.us[k], self.fidx[k] if self.fidx[k] is not None else 0, s_dst, p_src)


def prereads(kind):
    us = unit_list(*kind)
    st = Stream(us, 0)
    L = []
    st.read_upto_unit(L, st.funits[LEAD - 1] + 1 if st.nfrag >= LEAD else len(us))
    return L


def iteration(st, has_qk, has_pv, softmax, masked, next_kinds):
    """One pipelined iteration: softmax of score set `st`, S(next) into set st ^ 1, O += V P with P of set st ^ 1.  The first LEAD
    fragments arrive pre-read (ring slots 0 .. LEAD-1).  The iteration ends with the ring hand-over and the pre-reads of the next
    iteration; `next_kinds` = [(conditional, (has_qk, has_pv), label)]: the first entry is taken when %[cnt] == 0."""
    _uid[0] += 1
    uid = _uid[0]
    us = unit_list(has_qk, has_pv)
    n = len(us)
    va = []   # elements: one instruction, or a list that must stay contiguous (a scalar branch and the code it jumps over)
    for ins in (softmax_stream(st, masked) if softmax else []):
        va.append(decide(uid) if isinstance(ins, tuple) else ins)
    if os.environ.get("X2I_ATTN_ABL") == "novalu":      # measurement only: the MFMA / LDS / DMA stream alone
        va = []
    L = []
    vi = 0

    def cost(e):
        if isinstance(e, list):
            return len(e)
        return 4 if e.startswith("v_exp") else 1     # a transcendental occupies the VALU about four times as long

    total_cost = sum(cost(e) for e in va)
    spent = [0]

    def fill_to(target_cost):
        nonlocal vi
        while vi < len(va) and spent[0] + cost(va[vi]) <= target_cost:
            spent[0] += cost(va[vi])
            if isinstance(va[vi], list):
                L.extend(va[vi])
            else:
                L.append(va[vi])
            vi += 1

    def fill(count):
        fill_to(10 ** 9) if count >= len(va) else None
    split = max(0, n - LEAD)              # units in front of the hand-over
    early = max(0, split - 3)             # from this unit on, every remaining read of the iteration is issued at once: the hand-over's
                                          # lgkmcnt(0) then finds them returned instead of exposing one LDS round trip per iteration
    gaps = max(1, 2 * split - VALU_DELAY)
    sm = Stream(us, LEAD)
    mf = 0
    for k in range(split):
        sm.read_upto_unit(L, n if k >= early else k + LEAD + 1)
        sm.wait(L, k)
        for m in sm.mfmas(k, st ^ 1, st ^ 1):
            L.append(m)
            mf += 1
            if mf > VALU_DELAY:
                g = mf - VALU_DELAY          # spread the VALU stream evenly by issue time: g / G of its cost is out after gap g
                fill_to(g * total_cost // gaps)
    if n == 0:
        L += ["s_nop 15", "s_nop 15"]       # (no MFMA cover at all: the scores come from the prologue's last MFMAs)
    fill(len(va))
    sm.read_upto_unit(L, n)
    assert vi == len(va) and sm.next == sm.nfrag
    for ci, (cond, nxt, label) in enumerate(next_kinds):
        if cond:
            L += ["s_cmp_lg_u32 %[cnt], 0", f"s_cbranch_scc1 .Lalt{uid}_%="]
        L += sync_wait()                    # (lgkmcnt(0) inside: every fragment of this iteration is in registers)
        L += prereads(nxt)
        pre, pairs, post = sync_dma()
        L += pre
        tail_m = [m for k in range(split, n) for m in sm.mfmas(k, st ^ 1, st ^ 1)]
        # pieces between the remaining MFMAs: M0 write, an MFMA (or a nop) in between, the piece
        pi = 0
        if pairs:
            L.append(pairs[0][0])
        for mi, m in enumerate(tail_m):
            L.append(m)
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
        if has_pv:
            L += o_rescale(f"{uid}x{ci}")
        else:
            L += ["s_mov_b32 %[fl], 0"]     # O is still zero (first tile): nothing to rescale, and alpha(0) = 0 must never reach it
        L.append(f"s_branch {label}")
        if cond:
            L.append(f".Lalt{uid}_%=:")
    return L


def solo(kind, s_dst, p_src, preread):
    """A unit list on its own (prologue S(0), tail PV): reads LEAD ahead, no VALU stream."""
    us = unit_list(*kind)
    sm = Stream(us, LEAD if preread else 0)
    L = []
    if not preread:
        sm.read_upto_unit(L, sm.funits[LEAD - 1] + 1)
    for k in range(len(us)):
        sm.read_upto_unit(L, k + LEAD + 1)
        sm.wait(L, k)
        L += sm.mfmas(k, s_dst, p_src)
    return L


def prologue():
    L = ["s_nop 4", f"v_mov_b32 v{NEGBIG}, {NEG_BIG}"]
    for qb in range(2):
        L += [f"v_mov_b32 v{M_RUN + qb}, {NEG_BIG}", f"v_mov_b32 v{L_RUN + qb}, 0", f"v_mov_b32 v{M_USE + qb}, {NEG_BIG}",
              f"v_mov_b32 v{ALPHA + qb}, 1.0"]
    L += ["s_mov_b32 %[fl], 0"]
    for db in range(4):
        for qb in range(2):
            L += [f"v_accvgpr_write_b32 {O(db, qb, r)}, 0" for r in range(16)]
    for qb in range(2):
        L += [f"v_accvgpr_write_b32 {LS(qb, r)}, 0" for r in range(16)]
    L += [f"v_mov_b32 v{ONES + k}, 0x3f803f80" for k in range(4)]
    # Q fragments (B operand of S^T = K Q^T): lane holds Q[q0 + 32 qb + li][16 ds + 8 hi .. + 8], loaded straight into the accumulator file
    for qb in range(2):
        for ds in range(8):
            L.append(f"global_load_dwordx4 {Q(qb, ds)}, %[qo{qb}], %[qp] offset:{ds * 32}")
    # K(0) -> slot 0, K(1) -> slot 1 (rows behind Spad read as zero)
    L += ["s_mov_b32 %[so], 0"]
    for j in range(4):
        L += [f"s_add_u32 m0, %[kdst], {j * 4096}", "s_nop 0", f"buffer_load_dwordx4 %[kd{j}], %[kr], %[so] offen lds"]
    L += ["s_mov_b32 %[so], 0x4000"]
    for j in range(4):
        L += [f"s_add_u32 m0, %[kdst], {0x4000 + j * 4096}", "s_nop 0", f"buffer_load_dwordx4 %[kd{j}], %[kr], %[so] offen lds"]
    L += ["s_mov_b32 %[tk], 2", "s_mov_b32 %[tv], 0", "s_waitcnt vmcnt(0)", "s_barrier"]
    # S(0) = K(0) Q^T alone (score set 0)
    L += solo((True, False), 0, 0, False)
    return L


def epilogue():
    """O[q][d] = O^T[d][q] / l: lane (q = li, hi) holds d = 32 db + 8 (r >> 2) + 4 hi + (r & 3); a half-wave exchange turns two 8-byte
    fragments of neighbouring d-groups into one 16-byte store (as attention.hip); log2-sum-exp rows on request."""
    L = ["s_nop 15", "s_nop 15"]
    for qb in range(2):   # row sums: every L register of a lane holds the sum over ALL keys of its query row (both half-waves fed the MFMA)
        L += [f"v_accvgpr_read_b32 v{L_RUN + qb}, {LS(qb, 0)}"]
    for qb in range(2):
        L += [f"v_rcp_f32 v{PSUM + qb}, v{L_RUN + qb}"]
    # log2-sum-exp (x2i_attention_lse_bf16): m_run + log2(l) for q < S, +1e30 on the padding rows; lanes of the low half store
    L += ["s_cmp_lg_u32 %[lsef], 0", "s_cbranch_scc0 .Lnolse_%="]
    for qb in range(2):
        L += [f"v_log_f32 {T(4 + qb)}, v{L_RUN + qb}"]
    L += ["s_nop 1"]
    for qb in range(2):
        L += [f"v_add_f32 {T(4 + qb)}, {T(4 + qb)}, v{M_RUN + qb}",
              f"v_add_u32 {T(6)}, {32 * qb}, %[qv]",
              f"v_cmp_gt_i32 vcc, %[sS], {T(6)}",                          # q < S
              f"v_mov_b32 {T(7)}, 0x7149f2ca",                             # 1.0e30f
              f"v_cndmask_b32 {T(4 + qb)}, {T(7)}, {T(4 + qb)}, vcc",
              f"v_cmp_gt_i32 vcc, %[sSp], {T(6)}",                         # q < Spad ...
              f"v_cmp_eq_u32 %[cnd], 0, %[hi]",                            # ... and low half
              "s_nop 3", "s_and_b64 vcc, vcc, %[cnd]", "s_and_saveexec_b64 %[exs], vcc",
              f"global_store_dword %[lo], {T(4 + qb)}, %[lp] offset:{128 * qb}",
              "s_mov_b64 exec, %[exs]"]
    L += [".Lnolse_%=:"]
    for qb in range(2):
        L += [f"v_add_u32 {T(8)}, {32 * qb}, %[qv]", f"v_cmp_gt_i32 vcc, %[sS], {T(8)}", "s_nop 3", "s_and_saveexec_b64 %[exs], vcc"]
        if qb == 1:
            L += [f"v_add_u32 %[oo], %[ostep], %[oo]"]
        for db in range(4):
            for g in (0, 2):
                L += [f"v_accvgpr_read_b32 {T(k)}, {O(db, qb, 4 * g + k)}" for k in range(8)]
                L += [f"v_mul_f32 {T(k)}, {T(k)}, v{PSUM + qb}" for k in range(8)]
                L += [f"v_cvt_pk_bf16_f32 {T(8)}, {T(0)}, {T(1)}", f"v_cvt_pk_bf16_f32 {T(9)}, {T(2)}, {T(3)}",
                      f"v_cvt_pk_bf16_f32 {T(10)}, {T(4)}, {T(5)}", f"v_cvt_pk_bf16_f32 {T(11)}, {T(6)}, {T(7)}", "s_nop 1",
                      f"v_permlane32_swap_b32 {T(8)}, {T(10)}", f"v_permlane32_swap_b32 {T(9)}, {T(11)}", "s_nop 1",
                      f"global_store_dwordx4 %[oo], v[{TMP + 8}:{TMP + 11}], %[op] offset:{db * 64 + g * 16}", "s_nop 1"]
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
    L = prologue()
    MIDK, LASTK, TAILK = (True, True), (False, True), (False, True)
    # hand-over behind S(0): frees K slot 0, fetches K(2) / V(0); the fragment addresses now point at the slots of tile 1
    L += sync_block("p0")
    L += ["s_cmp_eq_u32 %[nt], 1", "s_cbranch_scc1 .Lonly_%="]
    L += prereads((True, False))
    # FIRST (score set 0): successor LAST(set 1) when nt == 2 (%[cnt] == 0), else MID(set 1)
    L += iteration(0, True, False, True, False, [(True, LASTK, ".Llast1_%="), (False, MIDK, ".Lmid1_%=")])
    for par in (1, 0):
        o = par ^ 1
        L += [f".Lmid{par}_%=:", "s_sub_u32 %[cnt], %[cnt], 1"]
        L += iteration(par, True, True, True, False, [(True, LASTK, f".Llast{o}_%="), (False, MIDK, f".Lmid{o}_%=")])
    for par in (1, 0):
        L += [f".Llast{par}_%=:"]
        L += iteration(par, False, True, True, True, [(False, TAILK, f".Ltail{par}_%=")])
    # ONLY (nt == 1): softmax(0) masked, then the hand-over that waits for V(0)
    L += [".Lonly_%=:"]
    L += iteration(0, False, False, True, True, [(False, TAILK, ".Ltail0_%=")])
    for par in (1, 0):
        L += [f".Ltail{par}_%=:"] + solo((False, True), 0, par, True) + ["s_branch .Lepi_%="]
    L += [".Lepi_%=:"] + epilogue()
    return L


def main():
    out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "attn_w4_loop.inc")
    L = build()
    txt = ["// GENERATED by gen_attn_w4.py -- do not edit; register map and schedule live in the generator.",
           f"// register map: scores v{SA}..v{SA + 127}, P fragments v{PF}..v{PF + 63}, softmax state up to v{_v - 1};"
           f" O a{OA}..a{OA + 127}, Q fragments a{QF}..a{QF + 63}, fragment ring a{FR}..a{FR + RING * 4 - 1}",
           f"// {len(L)} lines", "#define X2I_ATTN_W4_TEXT \\"]
    txt += [f'  "{l}\\n" \\' for l in L[:-1]]
    txt.append(f'  "{L[-1]}\\n"')
    clob = ", ".join(f'"v{i}"' for i in range(32, 256)) + ", " + ", ".join(f'"a{i}"' for i in range(256))
    txt.append(f"#define X2I_ATTN_W4_CLOBBERS {clob}")
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
