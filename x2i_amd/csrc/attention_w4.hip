// Flash attention forward, hand-scheduled: ONE wave per SIMD, four waves x 64 query rows per workgroup, the K-tile loop as a single
// generated asm statement (gen_attn_w4.py -> attn_w4_loop.inc: three-tile software pipeline, softmax VALU between the MFMAs of the
// neighbouring tiles, fragments read into the accumulator file, 2-deep K / V^T rings by LDS-DMA).  Same math, LDS images and swizzles as
// attention.hip (swapped QK^T with v_mfma_f32_32x32x16_bf16, key-order permutation, exp2-domain online softmax with defer-max);
// stands behind F.scaled_dot_product_attention of FluxAttnProcessor2_0 (lightcontrol/lightcontrol_flux.py:92-95,173-177).
// This file only computes the per-lane addresses the statement consumes.  Launcher: x2i_launch_attention (attention.hip).
#include "x2i_common.h"
#include "x2i_kernels.h"
#include "attn_w4_loop.inc"

namespace {

__global__ __launch_bounds__(256) void attn_w4_kernel(const bf16_t* __restrict__ Q, const bf16_t* __restrict__ K, const bf16_t* __restrict__ VT,
                                                      bf16_t* __restrict__ O, int H, int S, int Spad, int ldo, long long o_bs, float scale_log2,
                                                      int nbatch, float* __restrict__ lse, int prescale, int out8, float oinv) {
  extern __shared__ __attribute__((aligned(16))) char smem[];  // K ring [2][16 KiB] | V^T ring [2][16 KiB]
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int hi = lane >> 5, li = lane & 31;
  // XCD-aware block order, as attention.hip: an XCD walks a contiguous range of (batch, head, q-tile) triples
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

  // LDS-DMA source offsets of this thread's 16-byte chunks (bytes from the head's base; the LDS image is linear, the swizzle sits on
  // the source): piece j covers chunks [256 j + 64 wave, + 64)
  uint32_t kd[4], vd[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int p = j * 256 + tid;
    {  // K tile: row = key (256 B = 16 chunks); physical chunk c holds logical chunk c ^ (row & 15)
      const int row = p >> 4, cphys = p & 15;
      kd[j] = (uint32_t)(row * 128 + ((cphys ^ (row & 15)) << 3)) * 2u;
    }
    {  // V^T tile: row = d (128 B = 8 chunks); physical chunk c holds logical chunk c ^ ((row >> 1) & 7)
      const int row = p >> 3, cphys = p & 7;
      vd[j] = (uint32_t)(row * Spad + ((cphys ^ ((row >> 1) & 7)) << 3)) * 2u;
    }
  }
  // fragment read addresses (LDS bytes, ring slot 0): K fragment (sub-tile u, d-step ds): row u*32 + kvmap(li), chunk (2 ds + hi) ^ swz;
  // V^T fragment (d-block db, (u, kt) = g): row db*32 + li, chunk (2 g + hi) ^ swz
  const int kvm = (li & 0x13) | ((li & 4) << 1) | ((li & 8) >> 1);  // swap bits 2 and 3 (P^T becomes the PV B operand directly)
  const uint32_t sbase = (uint32_t)(uintptr_t)smem;
  uint32_t ka[8], va[4];
#pragma unroll
  for (int ds = 0; ds < 8; ++ds) ka[ds] = sbase + kvm * 256 + (((ds * 2 + hi) ^ (kvm & 15)) << 4);
#pragma unroll
  for (int g = 0; g < 4; ++g) va[g] = sbase + 32768 + li * 128 + (((2 * g + hi) ^ ((li >> 1) & 7)) << 4);
  const uint32_t kdst = __builtin_amdgcn_readfirstlane(sbase + wave * 1024);
  const uint32_t vdst = __builtin_amdgcn_readfirstlane(sbase + 32768 + wave * 1024);

  const int q = q0 + li;
  const uint32_t qo0 = (uint32_t)(min(q, Spad - 1) * 128 + hi * 8) * 2u;        // rows at or behind S are never stored: clamp the read
  const uint32_t qo1 = (uint32_t)(min(q + 32, Spad - 1) * 128 + hi * 8) * 2u;
  // bf16: 16 bytes per lane and d-group pair (ldo in elements); e4m3: 16 bytes per lane and 32-wide d block (ldo in bytes)
  uint32_t oo = out8 ? (uint32_t)((long long)q * ldo + hi * 16) : (uint32_t)(((long long)q * ldo + hi * 8) * 2);
  const uint32_t lo = (uint32_t)q * 4u;
  const char* Ob = (const char*)O + ((long long)b * o_bs + h * 128) * (out8 ? 1 : 2);
  const float* Lb = lse ? lse + bh * Spad : nullptr;
  const int nt = (S + 63) / 64;
  const int lim = S - (nt - 1) * 64 - 8 * hi;
  uint32_t cnt = (uint32_t)(nt > 2 ? nt - 2 : 0);
  const uint32_t ostep = (uint32_t)ldo * (out8 ? 32u : 64u);  // 32 rows
  // (integer arithmetic, not a comparison: an i1 would be materialised in a VGPR and cannot feed an "s" operand)
  const uint32_t lsef = __builtin_amdgcn_readfirstlane((uint32_t)((uintptr_t)lse >> 32) | (uint32_t)(uintptr_t)lse);
  const float thr = 8.0f;  // defer-max threshold, exp2 domain (as attention.hip)
  uint32_t s_so, s_so2, s_fl;
  unsigned long long s_cnd, s_exs;
  asm volatile(X2I_ATTN_W4_TEXT
               : [oo] "+v"(oo), [cnt] "+s"(cnt), [so] "=&s"(s_so), [so2] "=&s"(s_so2),
                 [fl] "=&s"(s_fl), [cnd] "=&s"(s_cnd), [exs] "=&s"(s_exs)
               : [ka0] "v"(ka[0]), [ka1] "v"(ka[1]), [ka2] "v"(ka[2]), [ka3] "v"(ka[3]), [ka4] "v"(ka[4]), [ka5] "v"(ka[5]), [ka6] "v"(ka[6]),
                 [ka7] "v"(ka[7]), [va0] "v"(va[0]), [va1] "v"(va[1]), [va2] "v"(va[2]), [va3] "v"(va[3]), [kd0] "v"(kd[0]), [kd1] "v"(kd[1]), [kd2] "v"(kd[2]), [kd3] "v"(kd[3]), [vd0] "v"(vd[0]), [vd1] "v"(vd[1]), [vd2] "v"(vd[2]),
                 [vd3] "v"(vd[3]), [kdst] "s"(kdst), [vdst] "s"(vdst), [qo0] "v"(qo0), [qo1] "v"(qo1), [lo] "v"(lo), [qv] "v"(q), [lim] "v"(lim), [hi] "v"(hi), [kr] "s"(k_rsrc),
                 [vr] "s"(v_rsrc), [qp] "s"(Qh), [op] "s"(Ob), [lp] "s"(Lb), [sc] "s"(scale_log2), [sS] "s"(S), [sSp] "s"(Spad), [nt] "s"(nt),
                 [ostep] "s"(ostep), [lsef] "s"(lsef), [thr] "s"(thr), [pres] "s"(prescale), [o8] "s"(out8), [oinv] "s"(oinv), [n448] "s"(-448.0f)
               : "memory", "vcc", "scc", "m0", X2I_ATTN_W4_CLOBBERS);
}

}  // namespace

// X2I_ERR_STATE: shape / alignment not served by this kernel (the caller falls back to the other forms)
int x2i_launch_attention_w4(const void* Q, const void* K, const void* VT, void* O, int B, int H, int S, int Spad, int ldo, long long o_bs,
                            float scale_log2, int prescale, hipStream_t stream, float* lse, int out8, float oinv) {
  // 16-byte row stores; 32-bit store offsets.  (ldo / o_bs: elements for bf16, bytes for e4m3)
  if ((((uintptr_t)O) & 15) || (ldo & (out8 ? 15 : 7)) || (o_bs & (out8 ? 15 : 7)) || (long long)S * ldo * (out8 ? 1 : 2) >= 0x7f000000LL)
    return X2I_ERR_STATE;
  const int rc = x2i_ensure_dynamic_smem((const void*)attn_w4_kernel, 65536);
  if (rc) return rc;
  dim3 grid(((S + 255) / 256) * H * B);
  hipLaunchKernelGGL(attn_w4_kernel, grid, dim3(256), 65536, stream, (const bf16_t*)Q, (const bf16_t*)K, (const bf16_t*)VT, (bf16_t*)O, H, S,
                     Spad, ldo, o_bs, scale_log2, B, lse, prescale, out8, oinv);
  return x2i_check_launch("attention");
}
