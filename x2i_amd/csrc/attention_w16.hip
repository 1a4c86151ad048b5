// Flash attention forward, hand-scheduled, on v_mfma_f32_16x16x32_bf16: attention_w4.hip's organisation (ONE wave per SIMD, four waves x 64
// query rows per workgroup, the whole kernel one generated asm statement: gen_attn_w16.py -> attn_w16_loop.inc) on the MFMA shape the matrix
// pipe sustains 11-14 % faster at this part's power cap.  A/B kernel (attn_variant = 12): V^T must arrive
// with its keys permuted within every 32-key span (position kk holds key 16 ((kk >> 2) & 1) + 4 (kk >> 3) + (kk & 3)) -- the caller's
// job until the fused QKV epilogue writes that order; bf16 output only.  Stands behind F.scaled_dot_product_attention of
// FluxAttnProcessor2_0 (lightcontrol/lightcontrol_flux.py:92-95,173-177).  This file only computes the per-lane addresses.
#include "x2i_common.h"
#include "x2i_kernels.h"
#include "attn_w16_loop.inc"

namespace {

__global__ __launch_bounds__(256) void attn_w16_kernel(const bf16_t* __restrict__ Q, const bf16_t* __restrict__ K, const bf16_t* __restrict__ VT,
                                                       bf16_t* __restrict__ O, int H, int S, int Spad, int ldo, long long o_bs, float scale_log2,
                                                       int nbatch, float* __restrict__ lse, int prescale) {
  extern __shared__ __attribute__((aligned(16))) char smem[];  // K ring [2][16 KiB] | V^T ring [2][16 KiB]
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int c = lane & 15, g = lane >> 4;
  const int nqt = gridDim.x / (H * nbatch);
  int bid = blockIdx.x;
  {
    const int T = gridDim.x, q = T >> 3, r = T & 7, xcd = bid & 7, idx = bid >> 3;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  const int qt = bid % nqt, h = (bid / nqt) % H, b = bid / (nqt * H);
  const int q0 = qt * 256 + wave * 64;
  const long long bh = (long long)b * H + h;
  const bf16_t* Qh = Q + bh * Spad * 128;
  const bf16_t* Kh = K + bh * Spad * 128;
  const bf16_t* Vh = VT + bh * 128 * Spad;
  __amdgpu_buffer_rsrc_t k_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)Kh, 0, (uint32_t)Spad * 256u, 0x00020000);
  __amdgpu_buffer_rsrc_t v_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)Vh, 0, (uint32_t)Spad * 256u, 0x00020000);

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

  const int q = q0 + c;   // query of block 0; block qb: + 16 qb
  uint32_t qo[4];
#pragma unroll
  for (int qb = 0; qb < 4; ++qb) qo[qb] = (uint32_t)(min(q + 16 * qb, Spad - 1) * 128 + g * 8) * 2u;   // rows at or behind S are never stored: clamp the read
  // 16-byte store of a d-block pair: even g -> first block at d = 4 g, odd g -> second block at d = 16 + 4 (g - 1)
  uint32_t oo = (uint32_t)(((long long)q * ldo) * 2) + ((g & 1) ? 32u + 8u * (uint32_t)(g - 1) : 8u * (uint32_t)g);
  const uint32_t lo = (uint32_t)q * 4u;
  const char* Ob = (const char*)O + ((long long)b * o_bs + h * 128) * 2;
  const float* Lb = lse ? lse + bh * Spad : nullptr;
  const int nt = (S + 63) / 64;
  const int lim = S - (nt - 1) * 64 - 4 * g;
  uint32_t cnt = (uint32_t)(nt > 2 ? nt - 2 : 0);
  const uint32_t ostep = (uint32_t)ldo * 32u;  // 16 rows of bf16
  const uint32_t lsef = __builtin_amdgcn_readfirstlane((uint32_t)((uintptr_t)lse >> 32) | (uint32_t)(uintptr_t)lse);
  const float thr = 8.0f;
  uint32_t s_so, s_so2, s_fl;
  unsigned long long s_cnd, s_exs;
  asm volatile(X2I_ATTN_W16_TEXT
               : [oo] "+v"(oo), [cnt] "+s"(cnt), [so] "=&s"(s_so), [so2] "=&s"(s_so2), [fl] "=&s"(s_fl), [cnd] "=&s"(s_cnd), [exs] "=&s"(s_exs)
               : [ka0] "v"(ka[0]), [ka1] "v"(ka[1]), [ka2] "v"(ka[2]), [ka3] "v"(ka[3]), [va0] "v"(va[0]), [va1] "v"(va[1]), [kd0] "v"(kd[0]),
                 [kd1] "v"(kd[1]), [kd2] "v"(kd[2]), [kd3] "v"(kd[3]), [vd0] "v"(vd[0]), [vd1] "v"(vd[1]), [vd2] "v"(vd[2]), [vd3] "v"(vd[3]),
                 [kdst] "s"(kdst), [vdst] "s"(vdst), [qo0] "v"(qo[0]), [qo1] "v"(qo[1]), [qo2] "v"(qo[2]), [qo3] "v"(qo[3]), [lo] "v"(lo), [qv] "v"(q),
                 [lim] "v"(lim), [hi] "v"(g), [kr] "s"(k_rsrc), [vr] "s"(v_rsrc), [qp] "s"(Qh), [op] "s"(Ob), [lp] "s"(Lb), [sc] "s"(scale_log2),
                 [sS] "s"(S), [sSp] "s"(Spad), [nt] "s"(nt), [ostep] "s"(ostep), [lsef] "s"(lsef), [thr] "s"(thr), [pres] "s"(prescale)
               : "memory", "vcc", "scc", "m0", X2I_ATTN_W16_CLOBBERS);
}

}  // namespace

// X2I_ERR_STATE: shape / alignment not served (the caller falls back to the other forms)
int x2i_launch_attention_w16(const void* Q, const void* K, const void* VT, void* O, int B, int H, int S, int Spad, int ldo, long long o_bs,
                             float scale_log2, int prescale, hipStream_t stream, float* lse) {
  if ((((uintptr_t)O) & 15) || (ldo & 7) || (o_bs & 7) || (long long)S * ldo * 2 >= 0x7f000000LL) return X2I_ERR_STATE;
  const int rc = x2i_ensure_dynamic_smem((const void*)attn_w16_kernel, 65536);
  if (rc) return rc;
  dim3 grid(((S + 255) / 256) * H * B);
  hipLaunchKernelGGL(attn_w16_kernel, grid, dim3(256), 65536, stream, (const bf16_t*)Q, (const bf16_t*)K, (const bf16_t*)VT, (bf16_t*)O, H, S, Spad, ldo,
                     o_bs, scale_log2, B, lse, prescale);
  return x2i_check_launch("attention (w16)");
}
