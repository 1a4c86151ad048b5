// Flash attention forward, hand-scheduled, on v_mfma_f32_16x16x32_bf16: attention_w4.hip's organisation (ONE wave per SIMD, four waves x 64
// query rows per workgroup, the whole kernel one generated asm statement: gen_attn_w16.py -> attn_w16_loop.inc) on the MFMA shape the matrix
// pipe sustains 11-14 % faster at this part's power cap.  The sampling path's attention kernel
// (x2i_attention_vp_bf16 / x2i_attention_vp_ws_bf16; attn_variant = 12 forces it for tools and tests): V^T must arrive with its keys permuted
// within every 32-key span (position kk holds key 16 ((kk >> 2) & 1) + 4 (kk >> 3) + (kk & 3)) -- the fused QKV epilogues write that order
// (x2i_qkv_desc.vt_perm); bf16 output only.  Stands behind F.scaled_dot_product_attention of FluxAttnProcessor2_0
// (lightcontrol/lightcontrol_flux.py:92-95,173-177).  This file computes the per-lane addresses and, in the stream-K form, the unit lists.
#include "x2i_common.h"
#include "x2i_kernels.h"
#include "attn_w16_loop.inc"
#include <algorithm>

namespace {

constexpr int SKA_SLAB_BYTES = 34 * 4096;   // hand-over state of a cut item: 32 pieces of O + one of -m + one of l, 256 lanes x 16 B each (gen_attn_w16.py)
constexpr int SKA_FLAG0 = 768;              // workspace flags [768, 1024): K tiles of last-round item E accumulated so far (the GEMMs use [0, 257) and [512, 768))
constexpr int SKA_ERR_SLOT = 256;           // the workspace's give-up marker (csrc/gemm_device.h SK_ERR_SLOT)
constexpr int SKA_MIN_TILES = 4;            // no part shorter than this: a cut that close to an item edge moves onto it
constexpr int SKA_G = 256;                  // workgroups of a stream-K launch (= CUs of the part)
constexpr int SKA_UNIT_TILES = 11;          // what starting + finishing a unit costs, in key tiles (prologue, pipeline fill / drain, epilogue or hand-over); from the cut sweep of tools/attn_sk_bench.py --cut (profiles/r06ze_*)

// SK = false: one work item (256 queries of one head) per workgroup, from its first key tile to its last -- the kernel of round 5.
// SK = true (stream-K, round 6): one workgroup per CU.  nitems = R G + r: every workgroup takes R whole items (item j G + w in round j, the
// order a plain launch is dispatched in) and a share of the LAST round's r items: each of them is cut along the KEY axis at tile c into an opening
// part [0, c) and a closing part [c, nt), CHAINED -- the closing part continues from the un-normalised O, the running maximum and the row sums the
// opening part leaves in the caller's workspace, so every query row is summed in exactly the order of an undivided item: bit-identical results,
// whatever the cut (it depends on the batch; the results do not).  Per XCD (workgroup w runs on XCD w & 7; its last-round items are r / 8
// consecutive ones, i.e. one or two heads): the first r / 8 workgroups run an opening part in front of their last whole item; the others run their
// whole items and then up to m closing parts -- long after the opening parts were published (the wait is bounded all the same and leaves the
// workspace's give-up marker instead of a hung GPU).  All opening parts of an XCD stream the same key tiles of the same heads at the same time, all
// closing parts likewise, and only its last whole-item round runs in two phase groups: the K / V^T tiles stay shared through the XCD's L2.  (A flat deal of
// the last round's tiles over the workgroups -- every workgroup another cut -- lost that sharing and ran SLOWER than whole items.)  c balances the two
// kinds of workgroup with the cost of starting and finishing a unit counted in: c + u = m (nt - c + u).
template <bool SK>
__global__ __launch_bounds__(256) void attn_w16_kernel(const bf16_t* __restrict__ Q, const bf16_t* __restrict__ K, const bf16_t* __restrict__ VT,
                                                       bf16_t* __restrict__ O, int H, int S, int Spad, int ldo, long long o_bs, float scale_log2,
                                                       int nbatch, float* __restrict__ lse, int prescale, int nitems, char* __restrict__ slabs,
                                                       unsigned* __restrict__ flags, int sk_c) {
  extern __shared__ __attribute__((aligned(16))) char smem[];  // K ring [2][16 KiB] | V^T ring [2][16 KiB] | persistent form: the next item's Q block [4 waves][16 KiB]
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int c = lane & 15, g = lane >> 4;
  const int nqt = nitems / (H * nbatch);
  const int nt_full = (S + 63) / 64;
  auto uni = [](int v) { return __builtin_amdgcn_readfirstlane(v); };

  // LDS-DMA source offsets: the images and swizzles of attention_w4.hip (K [64 keys][256 B], chunk ^ (row & 15); V^T [128 d][128 B], chunk ^ ((row >> 1) & 7))
  uint32_t kd[4], vd[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int p = j * 256 + tid;
    {
      const int row = p >> 4, cphys = p & 15;
      kd[j] = (uint32_t)(row * 128 + ((cphys ^ (row & 15)) << 3)) * 2u;
    }
    {
      const int row = p >> 3, cphys = p & 7;
      vd[j] = (uint32_t)(row * Spad + ((cphys ^ ((row >> 1) & 7)) << 3)) * 2u;
    }
  }
  // fragment read addresses (ring slot 0).  K fragment (key block kb, d-step ds): row 16 kb + c, chunk (4 ds + g) ^ c (+ 4096 kb);
  // V^T fragment (d-block db, span sp): row 16 db + c, chunk (4 sp + g) ^ ((c >> 1) & 7) (+ 2048 db)
  const uint32_t sbase = (uint32_t)(uintptr_t)smem;
  uint32_t ka[4], va[2];
#pragma unroll
  for (int ds = 0; ds < 4; ++ds) ka[ds] = sbase + c * 256 + (((ds * 4 + g) ^ c) << 4);
#pragma unroll
  for (int sp = 0; sp < 2; ++sp) va[sp] = sbase + 32768 + c * 128 + (((sp * 4 + g) ^ ((c >> 1) & 7)) << 4);
  const uint32_t kdst = __builtin_amdgcn_readfirstlane(sbase + wave * 1024);
  const uint32_t vdst = __builtin_amdgcn_readfirstlane(sbase + 32768 + wave * 1024);
  const uint32_t lsef = __builtin_amdgcn_readfirstlane((uint32_t)((uintptr_t)lse >> 32) | (uint32_t)(uintptr_t)lse);
  const uint32_t ostep = (uint32_t)ldo * 32u;  // 16 rows of bf16
  const float thr = 8.0f;
  const uint32_t sto = (uint32_t)tid * 16u;

  // ---- this workgroup's units: (item id, first key tile, tiles, slab); SK = false: one whole item
  struct Unit { int id, k0, len, slab; };
  const int G = gridDim.x, w = blockIdx.x;
  int R = 0, n_part = 0, part_e0 = 0, part_step = 0, part_big = 0, r8 = 1;
  if constexpr (SK) {
    R = nitems / G;
    const int r = nitems - R * G;
    if (sk_c > 0) {                                     // the last round's items cut along the key axis (r % 8 == 0, 0 < r8 < 32: launcher)
      r8 = r >> 3;
      const int x = w & 7, i = w >> 3, others = (G >> 3) - r8;
      if (i < r8) { part_big = 1; n_part = 1; part_e0 = x * r8 + i; }
      else { const int k = i - r8; n_part = k < r8 ? (r8 - 1 - k) / others + 1 : 0; part_e0 = x * r8 + k; part_step = others; }
    } else {                                            // whole items only (no workspace, or nothing worth cutting): item R G + w closes the list
      n_part = w < r ? 1 : 0;
    }
  }
  const int n_units = SK ? R + n_part : 1;
  auto unit_at = [&](int ui) -> Unit {
    Unit u = {(int)blockIdx.x, 0, nt_full, -1};
    if constexpr (SK) {
      if (ui >= n_units) return Unit{-1, 0, 0, -1};
      // an opening part runs in front of the workgroup's LAST whole item: published before any closing part is reached (those follow R whole
      // items), and only that one round of the XCD runs in two phase groups (in front of all R: -2 % at ten rounds instead of +3 %)
      const int pos_big = R > 0 ? R - 1 : 0;
      const int j = ui - ((part_big && ui > pos_big) ? 1 : 0);                   // whole item of round j ...
      u = Unit{w + j * G, 0, nt_full, -1};
      if (sk_c > 0) {
        if (part_big && ui == pos_big) u = Unit{R * G + (part_e0 % r8) * 8 + part_e0 / r8, 0, sk_c, part_e0};
        else if (!part_big && ui >= R) {                  // ... or closing part ui - R
          const int E = part_e0 + (ui - R) * part_step;
          u = Unit{R * G + (E % r8) * 8 + E / r8, sk_c, nt_full - sk_c, E};
        }
      }
      u = Unit{uni(u.id), uni(u.k0), uni(u.len), uni(u.slab)};
    }
    return u;
  };
  // (b, h, first query) of a work item id: the XCD-contiguous order of every attention kernel here
  auto place = [&](int id, int& b, int& h, int& qt) {
    int bid = id;
    const int T = nitems, q = T >> 3, r = T & 7, xcd = bid & 7, idx = bid >> 3;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    qt = bid % nqt; h = (bid / nqt) % H; b = bid / (nqt * H);
  };
  auto mk_rsrc = [&](const void* ptr, uint32_t bytes) {   // (workgroup-uniform by construction; readfirstlane makes it provable)
    const unsigned long long a = (unsigned long long)(uintptr_t)ptr;
    const unsigned long long au = ((unsigned long long)(unsigned)uni((int)(a >> 32)) << 32) | (unsigned)uni((int)(a & 0xffffffffu));
    return __builtin_amdgcn_make_buffer_rsrc((void*)(uintptr_t)au, 0, (uint32_t)uni((int)bytes), 0x00020000);
  };
  const uint32_t qdst = __builtin_amdgcn_readfirstlane(sbase + 65536 + wave * 1024);    // the next item's Q block: LDS behind the rings (persistent form)
  const uint32_t qdel = __builtin_amdgcn_readfirstlane(65536 + wave * 16384);           // ... read back per wave through the K fragment addresses
  uint32_t pre = 0;
  for (int ui = 0; ui < n_units; ++ui) {
    const Unit u = unit_at(ui);
    int b, h, qt;
    place(u.id, b, h, qt);
    const int q0 = qt * 256 + wave * 64;
    const long long bh = (long long)b * H + h;
    const bf16_t* Qh = Q + bh * Spad * 128;
    const bf16_t* Kh = K + bh * Spad * 128;
    const bf16_t* Vh = VT + bh * 128 * Spad;
    __amdgpu_buffer_rsrc_t k_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)Kh, 0, (uint32_t)Spad * 256u, 0x00020000);
    __amdgpu_buffer_rsrc_t v_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)Vh, 0, (uint32_t)Spad * 256u, 0x00020000);
    const int q = q0 + c;   // query of block 0; block qb: + 16 qb
    uint32_t qo[4];
#pragma unroll
    for (int qb = 0; qb < 4; ++qb) qo[qb] = (uint32_t)(min(q + 16 * qb, Spad - 1) * 128 + g * 8) * 2u;   // rows at or behind S are never stored: clamp the read
    // 16-byte store of a d-block pair: even g -> first block at d = 4 g, odd g -> second block at d = 16 + 4 (g - 1)
    uint32_t oo = (uint32_t)(((long long)q * ldo) * 2) + ((g & 1) ? 32u + 8u * (uint32_t)(g - 1) : 8u * (uint32_t)g);
    const uint32_t lo_ = (uint32_t)q * 4u;
    const char* Ob = (const char*)O + ((long long)b * o_bs + h * 128) * 2;
    const float* Lb = lse ? lse + bh * Spad : nullptr;
    const int nt = u.len;
    // (integer arithmetic, not comparisons: hipcc materialises an i1 in a VGPR, which cannot feed an "s" operand)
    const uint32_t cont = (uint32_t)uni(min(u.k0, 1)), hand = (uint32_t)uni(min(nt_full - (u.k0 + u.len), 1));   // hand = 0: this part holds the item's last key tile
    const int closes = 1 - (int)hand;
    const int lim = closes ? S - (nt_full - 1) * 64 - 4 * g : 0x10000;   // keys at or behind S are masked in the item's last tile only
    uint32_t cnt = (uint32_t)(nt > 2 ? nt - 2 : 0);
    const uint32_t so0 = (uint32_t)u.k0 * 0x4000u, so20 = (uint32_t)u.k0 * 128u;
    __amdgpu_buffer_rsrc_t s_rsrc = k_rsrc;   // (a valid descriptor when no slab is read or written)
    if constexpr (SK) {
      if (u.slab >= 0) {
        const unsigned long long a = (unsigned long long)(uintptr_t)slabs + (unsigned long long)u.slab * SKA_SLAB_BYTES;
        const unsigned long long au = ((unsigned long long)(unsigned)uni((int)(a >> 32)) << 32) | (unsigned)uni((int)(a & 0xffffffffu));
        s_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)(uintptr_t)au, 0, (uint32_t)SKA_SLAB_BYTES, 0x00020000);
      }
      if (cont) {   // wait until the predecessor has published key tiles [0, k0) of this item
        if (tid == 0) {
          unsigned* flag = flags + SKA_FLAG0 + u.slab;
          int spins = 0;
          while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != (unsigned)u.k0) {
            __builtin_amdgcn_s_sleep(32);
            if (++spins > (1 << 22)) {
              __hip_atomic_store(flags + SKA_ERR_SLOT, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
              break;
            }
          }
          __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        }
        __syncthreads();
      }
    }
    // the next unit's first loads, requested in front of this unit's epilogue (persistent form; a fresh item behind a unit that ends in an epilogue,
    // Q already scaled: the kernel's own scaling needs Q in registers)
    uint32_t nxt = 0, nso0 = 0;
    __amdgpu_buffer_rsrc_t nq_rsrc = k_rsrc, nk_rsrc = k_rsrc;
    if constexpr (SK) {
      const Unit un = unit_at(ui + 1);
      if (un.id >= 0 && un.k0 == 0 && hand == 0 && prescale == 0) {
        int nb, nh, nqt_;
        place(un.id, nb, nh, nqt_);
        const long long nbh = (long long)nb * H + nh;
        const int rows = min(256, Spad - nqt_ * 256);
        nq_rsrc = mk_rsrc(Q + (nbh * Spad + (long long)nqt_ * 256) * 128, (uint32_t)rows * 256u);
        nk_rsrc = mk_rsrc(K + nbh * Spad * 128, (uint32_t)Spad * 256u);
        nxt = 1;
      }
    }
    uint32_t s_so, s_so2, s_fl;
    unsigned long long s_cnd, s_exs;
    asm volatile(X2I_ATTN_W16_TEXT
                 : [oo] "+v"(oo), [cnt] "+s"(cnt), [so] "=&s"(s_so), [so2] "=&s"(s_so2), [fl] "=&s"(s_fl), [cnd] "=&s"(s_cnd), [exs] "=&s"(s_exs)
                 : [ka0] "v"(ka[0]), [ka1] "v"(ka[1]), [ka2] "v"(ka[2]), [ka3] "v"(ka[3]), [va0] "v"(va[0]), [va1] "v"(va[1]), [kd0] "v"(kd[0]),
                   [kd1] "v"(kd[1]), [kd2] "v"(kd[2]), [kd3] "v"(kd[3]), [vd0] "v"(vd[0]), [vd1] "v"(vd[1]), [vd2] "v"(vd[2]), [vd3] "v"(vd[3]),
                   [kdst] "s"(kdst), [vdst] "s"(vdst), [qo0] "v"(qo[0]), [qo1] "v"(qo[1]), [qo2] "v"(qo[2]), [qo3] "v"(qo[3]), [lo] "v"(lo_), [qv] "v"(q),
                   [lim] "v"(lim), [hi] "v"(g), [kr] "s"(k_rsrc), [vr] "s"(v_rsrc), [qp] "s"(Qh), [op] "s"(Ob), [lp] "s"(Lb), [sc] "s"(scale_log2),
                   [sS] "s"(S), [sSp] "s"(Spad), [nt] "s"(nt), [ostep] "s"(ostep), [lsef] "s"(lsef), [thr] "s"(thr), [pres] "s"(prescale),
                   [so0] "s"(so0), [so20] "s"(so20), [cont] "s"(cont), [hand] "s"(hand), [sr] "s"(s_rsrc), [sto] "v"(sto), [pre] "s"(pre), [nxt] "s"(nxt),
                   [nqr] "s"(nq_rsrc), [nkr] "s"(nk_rsrc), [nso0] "s"(nso0), [qdst] "s"(qdst), [qdel] "s"(qdel)
                 : "memory", "vcc", "scc", "m0", X2I_ATTN_W16_CLOBBERS);
    pre = nxt;
    if constexpr (SK) {
      __syncthreads();   // every wave is done with the K / V^T rings (and, for a hand-over, has drained its slab stores) before the next unit
      if (tid == 0 && u.slab >= 0) {
        // opening / middle part: publish the tiles accumulated so far; closing part of a cut item: the flag returns to zero (workspace invariant)
        if (hand) __hip_atomic_store(flags + SKA_FLAG0 + u.slab, (unsigned)(u.k0 + u.len), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else if (cont) __hip_atomic_store(flags + SKA_FLAG0 + u.slab, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
  }
}

}  // namespace

// the last round's items cut along the key axis: possible (a last round of a multiple of 8 items, key sequences long enough) and a workspace was handed over?
static bool w16_streamk(int nitems, int nt, int cus, void* ws, long long ws_bytes) {
  if (!ws || nt < 4 * SKA_MIN_TILES) return false;
  const int r = nitems % cus;
  if (r == 0 || (r & 7)) return false;
  return ws_bytes >= 4096 + (long long)r * SKA_SLAB_BYTES && !(((uintptr_t)ws) & 255);
}

// X2I_ERR_STATE: shape / alignment not served (the caller falls back to the other forms)
int x2i_launch_attention_w16(const void* Q, const void* K, const void* VT, void* O, int B, int H, int S, int Spad, int ldo, long long o_bs,
                             float scale_log2, int prescale, hipStream_t stream, float* lse, void* workspace, long long workspace_bytes) {
  if ((((uintptr_t)O) & 15) || (ldo & 7) || (o_bs & 7) || (long long)S * ldo * 2 >= 0x7f000000LL) return X2I_ERR_STATE;
  const int nitems = ((S + 255) / 256) * H * B;
  const int cus = x2i_num_cus();
  // persistent form (one workgroup per CU walks its items, each unit's exit requests the next one's first loads; option attn_streamk): launches of more
  // than one round on the 256-CU part; with the caller's workspace the last round's items are cut along the key axis as well
  if (!lse && x2i_options().attn_streamk && cus == SKA_G && nitems > cus) {
    const int shm = 65536 + 65536;
    const int rc = x2i_ensure_dynamic_smem((const void*)attn_w16_kernel<true>, shm);
    if (rc) return rc;
    int c = 0;
    if (w16_streamk(nitems, (S + 63) / 64, cus, workspace, workspace_bytes)) {
      // the cut: m = closing parts per closing workgroup, c from c + u = m (nt - c + u) (see the kernel's header)
      const int nt = (S + 63) / 64, r8 = (nitems % cus) >> 3, others = (cus >> 3) - r8;
      const int m = (r8 + others - 1) / others;
      c = (m * nt + (m - 1) * SKA_UNIT_TILES + (m + 1) / 2) / (m + 1);
#ifdef X2I_ABLATION
      if (x2i_options().attn_ablate >= 100) c = x2i_options().attn_ablate - 100;   // measurement library only (tools/attn_sk_bench.py --cut): the cut tile
#endif
      c = std::min(std::max(c, SKA_MIN_TILES), nt - SKA_MIN_TILES);
      // worth it?  An almost full last round leaves few closing workgroups with many short closing parts each (r / 8 = 31: one workgroup per XCD with 31 of
      // them), and every part pays a unit's fixed cost: cut only when the longer kind of workgroup finishes well before a whole item would
      const int t_parts = std::max(c + SKA_UNIT_TILES, m * (nt - c + SKA_UNIT_TILES)), t_whole = nt + SKA_UNIT_TILES;
      if (t_parts * 100 > t_whole * 92) c = 0;
    }
    hipLaunchKernelGGL(attn_w16_kernel<true>, dim3(cus), dim3(256), shm, stream, (const bf16_t*)Q, (const bf16_t*)K, (const bf16_t*)VT, (bf16_t*)O, H, S, Spad,
                       ldo, o_bs, scale_log2, B, lse, prescale, nitems, c ? (char*)workspace + 4096 : (char*)nullptr, c ? (unsigned*)workspace : (unsigned*)nullptr, c);
    return x2i_check_launch("attention (w16, persistent)");
  }
  const int rc = x2i_ensure_dynamic_smem((const void*)attn_w16_kernel<false>, 65536);
  if (rc) return rc;
  hipLaunchKernelGGL(attn_w16_kernel<false>, dim3(nitems), dim3(256), 65536, stream, (const bf16_t*)Q, (const bf16_t*)K, (const bf16_t*)VT, (bf16_t*)O, H, S, Spad, ldo,
                     o_bs, scale_log2, B, lse, prescale, nitems, (char*)nullptr, (unsigned*)nullptr, 0);
  return x2i_check_launch("attention (w16)");
}
