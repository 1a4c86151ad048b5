// bf16 MFMA GEMM with fused epilogues for the FLUX DiT / projector linears: operator contract, kernel choice and launch.
//
//   C[z][m][n] = epi( sum_k A[z][m][k] * W[n][k] )         (nn.Linear layout: W is [N,K], K contiguous)
//   v = acc + bias[n];  v = act(v);  if (res) v = res[z][m][n] + (gate ? gate[z][n] : 1) * v
//
// Replaces every nn.Linear on the hot path (reference: lightcontrol/lightcontrol_flux.py:64,66,256,257,282
// and the diffusers Attention/FeedForward linears built at :69-80,:135-153; utils/proj.py:18-25).
//
// Kernels (CDNA4, hand-written): gemm256.hip -- 256x256x64 tiles, 8 waves, full-line LDS-DMA staging (large linears and
// >= 256-channel implicit-GEMM convs); gemm128.hip -- 128x128x64 tiles, 4 waves (small / text-stream launches, peeled tails,
// narrower convs); the generic one-thread-per-output kernel below only for tiny or unaligned problems (bounded, see
// NAIVE_MAX_FLOP).  Shared device code: gemm_device.h.  A/B switches come from x2i_options() (resolved once, c_api.hip).
#include "gemm_device.h"

using namespace x2i_gemm;

namespace {

// Correct-for-any-shape fallback (K not a multiple of 64, unaligned leading dims): one thread per output, rows tiled over
// grid.y/z so that any M is launchable.  Only ever used for tiny problems (reduced-width tests, 4-channel VAE conv_in).
__global__ void gemm_naive_kernel(GemmP p, int batch) {
  const int n = blockIdx.x * blockDim.x + threadIdx.x;
  const long long mz = (long long)blockIdx.z * gridDim.y + blockIdx.y;  // flattened (z, m)
  const int z = (int)(mz / p.M), m = (int)(mz % p.M);
  if (n >= p.N || z >= batch) return;
  const bf16_t* a = p.A + (long long)z * p.a_bs + (long long)m * p.lda;
  const bf16_t* w = p.W + (long long)(z / p.wdiv) * p.w_bs + (long long)n * p.ldw;
  float acc = 0.f;
  for (int k = 0; k < p.K; ++k) acc = fmaf(bf16_to_f32(a[k]), bf16_to_f32(w[k]), acc);
  float v = acc + (p.bias ? bf16_to_f32(p.bias[(long long)(z / p.wdiv) * p.bias_gs + n]) : 0.f) + (p.bias2 ? p.bias2[(long long)z * p.bias2_bs + n] : 0.f);
  v = apply_act(v, p.act);
  if (p.res) {
    const float g = p.gate ? p.gate[(long long)z * p.gate_bs + n] : 1.f;
    v = fmaf(g, v, bf16_to_f32(p.res[(long long)z * p.r_bs + (long long)m * p.ldr + n]));  // one rounding, as in the MFMA kernels
  }
  const long long coff = (long long)z * p.c_bs + (long long)m * p.ldc + n;
  if (p.out_f32) ((float*)p.C)[coff] = v;
  else ((bf16_t*)p.C)[coff] = f32_to_bf16(v);
  if (p.C2) p.C2[coff] = f32_to_bf16(apply_act(v, p.act2));
}

// The generic kernel is 100-1000x slower than the MFMA kernels: refuse (with a message naming the cause) instead of silently
// running a large problem through it.
constexpr double NAIVE_MAX_FLOP = 5.0e10;

}  // namespace

int x2i_gemm_sk_max_tiles() { return SK_MAX_TILES; }
int x2i_gemm_sk_slabs() { return SK_SLABS; }
long long x2i_gemm_sk_slab_bytes() { return SK_SLAB_BYTES; }

static int launch_gemm_impl(const x2i_gemm_args* a, const x2i_conv_desc* cd, const x2i_qkv_desc* qd, hipStream_t stream);
static int check_qkv_desc(const x2i_gemm_args* a, const x2i_qkv_desc* qd, const char* who);

int x2i_launch_gemm(const x2i_gemm_args* a, hipStream_t stream) { return launch_gemm_impl(a, nullptr, nullptr, stream); }
int x2i_launch_gemm_conv(const x2i_gemm_args* a, const x2i_conv_desc* cd, hipStream_t stream) {
  return launch_gemm_impl(a, cd, nullptr, stream);
}
int x2i_launch_gemm_qkv(const x2i_gemm_args* a, const x2i_qkv_desc* qd, hipStream_t stream) {
  if (!a || !qd) return x2i_set_error(X2I_ERR_ARG, "gemm_qkv: null pointer");
  if (const int rc = check_qkv_desc(a, qd, "gemm_qkv")) return rc;
  return launch_gemm_impl(a, nullptr, qd, stream);
}

static void fill_gemm_p(const x2i_gemm_args* a, const x2i_qkv_desc* qd, GemmP& p) {
  p.A = (const bf16_t*)a->A; p.a_bs = a->a_batch_stride; p.lda = a->lda;
  p.W = (const bf16_t*)a->W; p.ldw = a->ldw; p.w_bs = a->w_batch_stride;
  p.wdiv = a->w_group > 0 ? a->w_group : 1; p.bias_gs = a->w_group > 0 ? a->N : 0;
  p.bias = (const bf16_t*)a->bias;
  p.C = a->C; p.c_bs = a->c_batch_stride; p.ldc = a->ldc;
  p.C2 = (bf16_t*)a->C2; p.act2 = a->act2;
  p.gate = a->gate; p.gate_bs = a->gate_batch_stride;
  p.res = (const bf16_t*)a->res; p.r_bs = a->res_batch_stride; p.ldr = a->ldr;
  p.bias2 = a->bias2; p.bias2_bs = a->bias2_batch_stride;
  p.M = a->M; p.N = a->N; p.K = a->K; p.act = a->act; p.out_f32 = a->out_f32;
  p.cH = p.cW = p.cCin = p.cOW = p.cKW = p.cStride = p.cPad = p.cUp = p.cPadW = p.cRowPitch = p.cKorder = 0; p.cMom = nullptr; p.cMomBlocks = 0;
  p.gm = 4;
  p.q_on = 0; p.q_H = p.q_Spad = p.q_tok_off = p.q_rpb = p.q_row0 = p.q_vperm = 0; p.q_eps = 0.f; p.q_qs = 1.f;
  p.q_nq = p.q_nk = nullptr; p.q_cos = p.q_sin = nullptr; p.q_Q = p.q_K = p.q_VT = nullptr;
  p.f_sa = p.f_sw = nullptr; p.f_sa_bs = 0; p.f_alpha = p.f_oinv = 1.f; p.f_out8 = 0;
  p.nbatch = 1; p.sk_on = 0; p.sk_slabs = nullptr; p.sk_flags = nullptr; p.fx_v0 = 0;
  if (qd) {
    p.q_on = 1; p.q_H = qd->H; p.q_Spad = qd->Spad; p.q_tok_off = qd->tok_off; p.q_rpb = qd->rows_per_sample; p.q_eps = qd->eps; p.q_qs = qd->q_scale == 0.f ? 1.f : qd->q_scale;
    p.q_nq = (const bf16_t*)qd->norm_q; p.q_nk = (const bf16_t*)qd->norm_k; p.q_cos = qd->cos; p.q_sin = qd->sin;
    p.q_Q = (bf16_t*)qd->Q; p.q_K = (bf16_t*)qd->K; p.q_VT = (bf16_t*)qd->VT; p.q_vperm = qd->vt_perm ? 1 : 0;
    p.ldc = a->N; p.c_bs = 0;  // C is never written
  }
}

// patch shape per XCD (measured, profiles/r01g_gm_sweep.log): few tile columns and a deep K -> one tile row at a time, so the XCD's
// concurrent tiles share each A panel and it leaves HBM once; otherwise near-square patches (6 x 5.3) keep the L2 traffic per K-step
// lowest; very wide N prefers two rows
static int pick_gm(int tn, int K) {
  const X2IOptions& opt = x2i_options();
  if (opt.gemm_gm > 0) return opt.gemm_gm;
  if (tn <= 16) return K >= 8192 ? 1 : 4;
  if (tn <= 64) return 6;
  return 2;
}

// operands and epilogue of a plain (or fused-QKV) GEMM as the persistent kernel needs them
static bool persistent_ok(const x2i_gemm_args* a, const x2i_qkv_desc* qd) {
  const X2IOptions& opt = x2i_options();
  const bool res = a->res != nullptr;
  const bool fast = (a->K % BK == 0) && (a->lda % 8 == 0) && (a->ldw % 8 == 0) && (((uintptr_t)a->A & 15) == 0) && (((uintptr_t)a->W & 15) == 0) &&
                    ((a->a_batch_stride & 7) == 0) && ((long long)a->M * a->lda * 2 < 0x7f000000LL) && ((long long)a->N * a->ldw * 2 < 0x7f000000LL);
  return fast && opt.gemm_persist && opt.gemm_w4 == 1 && a->K >= 3 * BK && !a->w_batch_stride && (a->N & 7) == 0 && (a->ldc & 7) == 0 &&
         (qd || ((a->c_batch_stride & 7) == 0 && ((((uintptr_t)a->C) | ((uintptr_t)a->C2)) & 15) == 0 && (long long)a->M * a->ldc * 2 < 0x7f000000LL)) &&
         ((long long)(a->batch - 1) * a->a_batch_stride + (long long)a->M * a->lda) * 2 < 0x7f000000LL &&
         (!res || ((a->ldr & 7) == 0 && (a->res_batch_stride & 7) == 0 && (((uintptr_t)a->res) & 15) == 0 && (long long)a->M * a->ldr * 2 < 0x7f000000LL));
}

// stream-K decision for `tiles` 256^2 tiles of nkt K-tiles on `cus` CUs (see launch_gemm_impl); fetches the workspace
static bool streamk_for(const x2i_gemm_args* a, long long tiles, int nkt, int cus, float** slabs, unsigned** flags, int* rc) {
  const X2IOptions& opt = x2i_options();
  *rc = X2I_OK;
  if (!(opt.gemm_streamk && opt.gemm_tile == 0 && tiles > cus && cus <= SK_MAX_TILES && nkt >= 16)) return false;
  const long long r = tiles % cus, S = tiles / cus;
  if (!(r > 0 && r * nkt / cus + 6 <= nkt && r * nkt >= 6 && cus <= r * (S + 1))) return false;
  return x2i_streamk_workspace(a, slabs, flags, rc);   // the caller's workspace (include/x2i.h); none: whole tiles / peeled tail
}

// Parallel split with fix-up (gemm256p.hip, FX): for launches that cannot fill the chip with whole 256^2 tiles -- fewer tiles per batch
// item than CUs -- and whose K is deep enough to cut (>= 96 K-tiles): the gated-residual linears of a small batch (single-block
// proj_out, K = 15360; ff.net.2 / ff_context.net.2, K = 12288; infer/inference_qwenvl.py:233-237 samples at batch 1).  The decision and
// the cuts depend on ONE batch item's shape only, never on the batch: a sample's result is the same whatever rides with it.
// a1: the second problem of a grouped launch (or null); *v0 = workgroups per XCD (of cus / 8) that share problem 0's tiles.
static bool fx_for(const x2i_gemm_args* a0, const x2i_gemm_args* a1, int cus, int* v0, float** slabs, unsigned** flags, int* rc) {
  const X2IOptions& opt = x2i_options();
  *rc = X2I_OK;
  if (!(opt.gemm_fx && opt.gemm_streamk && opt.gemm_tile == 0 && cus <= SK_MAX_TILES && a0->workspace)) return false;
  const int nk = a0->K / BK;
  if (nk < opt.gemm_fx_nk || !a0->res || a0->act != X2I_ACT_NONE || a0->C2 || a0->out_f32) return false;
  if (a1 && (a1->batch != a0->batch || !a1->res || a1->act != X2I_ACT_NONE || a1->C2 || a1->out_f32)) return false;
  const long long T0 = (long long)((a0->M + BM2 - 1) / BM2) * ((a0->N + BN2 - 1) / BN2);
  const long long T1 = a1 ? (long long)((a1->M + BM2 - 1) / BM2) * ((a1->N + BN2 - 1) / BN2) : 0;
  const long long Ts = T0 + T1;
  if (Ts >= cus || Ts * 5 < cus) return false;                 // whole tiles fill the chip / too few tiles: more than five parts per tile
  // gemm_fx = 2 (default): only items with at most HALF a round of tiles (a 512^2 sample: 72; measured x1.34-1.51 at batch 1, x0.81-0.96 at batch
  // 2 / 4, profiles/r05g_*); a 1024^2 sample (216 tiles) keeps whole tiles, where the split measures x0.92-1.03.  Still by the item's shape alone.
  if (opt.gemm_fx == 2 && Ts * 2 > cus) return false;
  // the split is per XCD (an XCD's cus / 8 workgroups share the K-tile space of the item tiles the tile order gives that XCD)
  if ((cus & 7) || (T0 & 7) || (T1 & 7)) return false;
  const int gx = cus >> 3;
  int va = gx;
  if (a1) {
    va = (int)((gx * T0 + Ts / 2) / Ts);
    if (va < 1 || va >= gx) return false;
  }
  const int vb = gx - va;
  auto share_ok = [&](long long T, int v) { return (T >> 3) <= v && (T >> 3) * 6 >= v; };   // one finisher per tile; at most six parts per tile
  if (!share_ok(T0, va) || (a1 && !share_ok(T1, vb))) return false;
  if (!x2i_streamk_workspace(a0, slabs, flags, rc)) return false;
  *v0 = va;
  return true;
}

// Two GEMMs of the same kind in ONE persistent launch (x2i_gemm_pair_bf16 / x2i_gemm_qkv_pair_bf16): problem 1's tiles follow problem
// 0's in the tile list.  Same results as two launches (each output tile is computed exactly as before); taken when both problems
// are served by the persistent kernel, have the same K and the same epilogue kind; otherwise the two launches are issued one
// after the other.
int x2i_launch_gemm_pair(const x2i_gemm_args* a0, const x2i_qkv_desc* q0, const x2i_gemm_args* a1, const x2i_qkv_desc* q1, hipStream_t stream) {
  if (!a0 || !a1 || ((q0 != nullptr) != (q1 != nullptr))) return x2i_set_error(X2I_ERR_ARG, "gemm_pair: null pointer / mixed kinds");
  X2IOptions& opt = x2i_options();
  auto plain = [](const x2i_gemm_args* a) { return a->A && a->W && a->M > 0 && a->N >= 256 && a->K > 0 && a->batch > 0 && !a->out_f32 && !a->bias2 && a->M >= 256; };
  bool ok = opt.gemm_pair && opt.gemm_tile == 0 && plain(a0) && plain(a1) && a0->K == a1->K && a0->act == a1->act &&
            ((a0->res != nullptr) == (a1->res != nullptr)) && ((a0->C2 != nullptr) == (a1->C2 != nullptr)) && (q0 || (a0->C && a1->C)) && persistent_ok(a0, q0) && persistent_ok(a1, q1);
  if (ok && q0) ok = check_qkv_desc(a0, q0, "gemm_qkv_pair") == X2I_OK && check_qkv_desc(a1, q1, "gemm_qkv_pair") == X2I_OK && (q0->H * 128) % BN2 == 0 && (q1->H * 128) % BN2 == 0;
  if (ok && a0->gate && !a0->res) ok = false;
  kern2_t kern = ok ? pick_gemm256p_pair(a0->act, a0->res != nullptr, q0 != nullptr, a0->C2 != nullptr) : nullptr;
  if (!kern) {
    const int rc = q0 ? x2i_launch_gemm_qkv(a0, q0, stream) : x2i_launch_gemm(a0, stream);
    if (rc) return rc;
    return q1 ? x2i_launch_gemm_qkv(a1, q1, stream) : x2i_launch_gemm(a1, stream);
  }
  GemmP2 pp;
  const x2i_gemm_args* as[2] = {a0, a1};
  const x2i_qkv_desc* qs[2] = {q0, q1};
  long long tiles = 0;
  for (int i = 0; i < 2; ++i) {
    GemmP& p = pp.p[i];
    fill_gemm_p(as[i], qs[i], p);
    p.tilesM = (as[i]->M + BM2 - 1) / BM2; p.tilesN = (as[i]->N + BN2 - 1) / BN2;
    p.gm = pick_gm(p.tilesN, as[i]->K);
    p.nbatch = as[i]->batch;
    tiles += (long long)p.tilesM * p.tilesN * p.nbatch;
  }
  const int cus = x2i_num_cus();
  float* sk_slabs = nullptr;
  unsigned* sk_flags = nullptr;
  int rc = X2I_OK, v0 = 0;
  if (!q0 && fx_for(a0, a1, cus, &v0, &sk_slabs, &sk_flags, &rc)) {   // small batch: the two problems' tiles cut along K over all CUs
    kern2_t kfx = pick_gemm256p_pair_fx();
    rc = x2i_ensure_dynamic_smem((const void*)kfx, SMEM2P_BYTES);
    if (rc) return rc;
    pp.p[0].fx_v0 = v0;
    for (int i = 0; i < 2; ++i) pp.p[i].sk_slabs = sk_slabs, pp.p[i].sk_flags = sk_flags;   // (the epilogue of either problem reads the slabs)
    hipLaunchKernelGGL(kfx, dim3((unsigned)cus), dim3(256), SMEM2P_BYTES, stream, pp);
    opt.last_gemm_tile = 3256;  // (read-back for tests: grouped launch, parallel split with fix-up)
    return x2i_check_launch("gemm_pair (fx)");
  }
  if (rc) return rc;
  rc = x2i_ensure_dynamic_smem((const void*)kern, SMEM2P_BYTES);
  if (rc) return rc;
  const bool sk = streamk_for(a0, tiles, a0->K / BK, cus, &sk_slabs, &sk_flags, &rc);
  if (rc) return rc;
  pp.p[0].sk_on = sk ? 1 : 0; pp.p[0].sk_slabs = sk_slabs; pp.p[0].sk_flags = sk_flags;
  hipLaunchKernelGGL(kern, dim3((unsigned)(tiles < cus ? tiles : cus)), dim3(256), SMEM2P_BYTES, stream, pp);
  opt.last_gemm_tile = 2256;  // (read-back for tests: the grouped launch was taken)
  return x2i_check_launch("gemm_pair");
}

// ---- channel moments from the conv epilogue (x2i_conv_desc.moments): the kernels write per-row-block sums (gemm_device.h: mom_flush), two small
// passes add them in a fixed order: MOM_SLABS slabs of row blocks, then the slabs (+ the caller's running moments when accumulating)
constexpr int MOM_SLABS = 64;
// Sum of rows [r0, r1) of a row-major f32 matrix with n2 <= 1024 columns (n2 % 4 == 0), by one 256-thread block: n2 / 4 lanes of 16 bytes per
// row, 256 / lanes row groups side by side (four loads in flight each), the groups added through LDS in a fixed order: deterministic.  The sum
// of columns 4l .. 4l+3 is returned to thread l < n2 / 4.  (The first form -- one thread per column walking all its rows -- ran at the latency
// of one load per row: 27 + 17 us per convolution, as much as the statistics pass it replaces.)
__device__ __forceinline__ f32x4_t mom_block_sum(const float* __restrict__ src, int r0, int r1, int n2, f32x4_t* red) {
  const int lanes = n2 >> 2, groups = 256 / lanes;
  const int l = threadIdx.x % lanes, g = threadIdx.x / lanes;
  f32x4_t a0 = {0.f, 0.f, 0.f, 0.f}, a1 = a0, a2 = a0, a3 = a0;
  if (g < groups) {
    const float* p = src + 4 * l;
    int rb = r0 + g;
    for (; rb + 3 * groups < r1; rb += 4 * groups) {
      a0 += *(const f32x4_t*)(p + (long long)rb * n2);
      a1 += *(const f32x4_t*)(p + (long long)(rb + groups) * n2);
      a2 += *(const f32x4_t*)(p + (long long)(rb + 2 * groups) * n2);
      a3 += *(const f32x4_t*)(p + (long long)(rb + 3 * groups) * n2);
    }
    for (; rb < r1; rb += groups) a0 += *(const f32x4_t*)(p + (long long)rb * n2);
  }
  f32x4_t v = (a0 + a1) + (a2 + a3);
  red[threadIdx.x] = v;
  __syncthreads();
  if (g == 0)
    for (int gg = 1; gg < groups; ++gg) v += red[gg * lanes + l];
  return v;
}
// partial sums f32 [batch][blocks][n2] (n2 = N / 2: (sum, sum of squares) per channel quad) -> tmp f32 [batch][MOM_SLABS][n2]
__global__ __launch_bounds__(256) void conv_moments_slabs_kernel(const float* __restrict__ part, float* __restrict__ tmp, int blocks, int n2) {
  __shared__ f32x4_t red[256];
  const int z = blockIdx.y, s = blockIdx.x;
  const int per = (blocks + MOM_SLABS - 1) / MOM_SLABS;
  const int b0 = min(blocks, s * per), b1 = min(blocks, b0 + per);
  const f32x4_t v = mom_block_sum(part + (long long)z * blocks * n2, b0, b1, n2, red);
  if ((int)threadIdx.x < (n2 >> 2)) *(f32x4_t*)(tmp + ((long long)z * MOM_SLABS + s) * n2 + 4 * threadIdx.x) = v;
}
// tmp -> moments f32 [batch][N][2]: the quad's sums at its first channel, zeros at the other three
// (`rows` = MOM_SLABS behind the slab kernel, or the row blocks themselves when there are few of them: one launch instead of two)
__global__ __launch_bounds__(256) void conv_moments_finish_kernel(const float* __restrict__ tmp, float* __restrict__ mom, int rows, int n2, int accumulate) {
  __shared__ f32x4_t red[256];
  const int z = blockIdx.x;
  const f32x4_t v = mom_block_sum(tmp + (long long)z * rows * n2, 0, rows, n2, red);
  if ((int)threadIdx.x < (n2 >> 2)) {   // this thread: quads 2 l and 2 l + 1 = channels 8 l and 8 l + 4
    float* dst = mom + (long long)z * n2 * 4 + (long long)threadIdx.x * 16;
    if (accumulate) {
      dst[0] += v[0]; dst[1] += v[1]; dst[8] += v[2]; dst[9] += v[3];
    } else {
      *(f32x4_t*)dst = (f32x4_t){v[0], v[1], 0.f, 0.f};
      *(f32x4_t*)(dst + 4) = (f32x4_t){0.f, 0.f, 0.f, 0.f};
      *(f32x4_t*)(dst + 8) = (f32x4_t){v[2], v[3], 0.f, 0.f};
      *(f32x4_t*)(dst + 12) = (f32x4_t){0.f, 0.f, 0.f, 0.f};
    }
  }
}
long long x2i_conv_moments_scratch(int M, int N, int batch) {   // row blocks of 64 rows at most two per 128-row tile (the finer of the two kernels) + the slab sums
  return (long long)batch * (((long long)(M + 127) / 128 * 2) + MOM_SLABS) * (N / 2);
}

static int pad_w_of(const x2i_conv_desc* cd) { return cd->pad_w_p1 <= 0 ? cd->pad : cd->pad_w_p1 - 1; }
// behind a convolution launch: the per-row-block partial moments (`blocks` row blocks per batch item) -> x2i_conv_desc.moments
static int conv_moments_tail(const x2i_gemm_args* a, const x2i_conv_desc* cd, hipStream_t stream, int blocks) {
  int rc = x2i_check_launch("conv");
  if (rc || !cd->moments) return rc;
  const int n2 = a->N / 2;
  if (blocks <= 128) {   // few row blocks (small images / small batches live here): the finishing block adds them itself, in the same fixed tree
    hipLaunchKernelGGL(conv_moments_finish_kernel, dim3(a->batch), dim3(256), 0, stream, (const float*)cd->moments_scratch, cd->moments, blocks, n2,
                       cd->moments_accumulate ? 1 : 0);
    return x2i_check_launch("conv_moments_finish");
  }
  float* tmp = cd->moments_scratch + (long long)a->batch * blocks * n2;
  hipLaunchKernelGGL(conv_moments_slabs_kernel, dim3(MOM_SLABS, a->batch), dim3(256), 0, stream, (const float*)cd->moments_scratch, tmp, blocks, n2);
  rc = x2i_check_launch("conv_moments_slabs");
  if (rc) return rc;
  hipLaunchKernelGGL(conv_moments_finish_kernel, dim3(a->batch), dim3(256), 0, stream, (const float*)tmp, cd->moments, MOM_SLABS, n2, cd->moments_accumulate ? 1 : 0);
  return x2i_check_launch("conv_moments_finish");
}

// One launch of a persistent convolution kernel (gemm256c.hip / gemm512c.hip) per CHUNK of batch items whose images fit the kernel's single 2 GB
// input descriptor (a 1024^2 x 256-channel image is 512 MiB: four of them do not fit, two launches of two do); every chunk's tile list fills the
// chip on its own.  Pointers of `pm` advance by whole batch items; the moments of a chunk's items are finished behind its launch.
static int launch_conv_chunks(kern_t kc, int smem, const GemmP& p0, const x2i_gemm_args* a, const x2i_conv_desc* cd, hipStream_t stream, int mom_blocks) {
  const long long img2 = (long long)cd->H * cd->W * cd->Cin * 2, bias_b = ((long long)cd->pad * cd->W + pad_w_of(cd)) * cd->Cin * 2;
  const long long room = 0x7f000000LL - bias_b - img2;
  if (room < 0) return X2I_ERR_STATE;
  long long per = a->a_batch_stride > 0 ? room / (a->a_batch_stride * 2) + 1 : a->batch;
  if (per > a->batch) per = a->batch;
  // grouped weights: a chunk holds whole groups, or (when a group does not fit) a part of ONE group
  const int wg = p0.wdiv > 1 ? p0.wdiv : 0;
  if (wg && per < a->batch) {
    if (per >= wg) per -= per % wg;
    else while (wg % per) --per;
  }
  const int cus = x2i_num_cus();
  for (int z0 = 0; z0 < a->batch; z0 += (int)per) {
    const int nb = (int)(a->batch - z0 < per ? a->batch - z0 : per);
    GemmP pm = p0;
    pm.nbatch = nb;
    pm.A = p0.A + (long long)z0 * p0.a_bs;
    pm.C = (void*)((bf16_t*)p0.C + (long long)z0 * p0.c_bs);
    if (p0.res) pm.res = p0.res + (long long)z0 * p0.r_bs;
    if (p0.bias2) pm.bias2 = p0.bias2 + (long long)z0 * p0.bias2_bs;
    if (wg) {
      pm.W = p0.W + (long long)(z0 / wg) * p0.w_bs;
      if (p0.bias) pm.bias = p0.bias + (long long)(z0 / wg) * p0.bias_gs;
      if (per < wg) pm.wdiv = nb;   // (all items of this chunk belong to one group)
    }
    const long long tiles = (long long)pm.tilesM * pm.tilesN * nb;
    hipLaunchKernelGGL(kc, dim3((unsigned)(tiles < cus ? tiles : cus)), dim3(256), smem, stream, pm);
    if (cd->moments) {
      x2i_gemm_args ac = *a;
      x2i_conv_desc cc = *cd;
      ac.batch = nb;
      cc.moments = cd->moments + (long long)z0 * a->N * 2;
      const int rc = conv_moments_tail(&ac, &cc, stream, mom_blocks);
      if (rc) return rc;
    }
  }
  return x2i_check_launch("conv");
}

static int launch_gemm_impl(const x2i_gemm_args* a, const x2i_conv_desc* cd, const x2i_qkv_desc* qd, hipStream_t stream) {
  if (!a || !a->A || !a->W || (!a->C && !qd)) return x2i_set_error(X2I_ERR_ARG, "gemm: null pointer");
  const bool conv = cd != nullptr;
  if (a->M <= 0 || a->N <= 0 || a->K <= 0 || a->batch <= 0) return x2i_set_error(X2I_ERR_SHAPE, "gemm: bad shape M=%d N=%d K=%d batch=%d", a->M, a->N, a->K, a->batch);
  if (a->gate && !a->res) return x2i_set_error(X2I_ERR_ARG, "gemm: gate without residual");
  if (a->w_group < 0 || (a->w_group > 0 && (!a->w_batch_stride || qd || a->C2 || a->out_f32)))
    return x2i_set_error(X2I_ERR_ARG, "gemm: w_group=%d needs w_batch_stride != 0 and goes with the plain bf16 outputs only (no fused QKV / C2 / f32 output)", a->w_group);
  GemmP p;
  fill_gemm_p(a, qd, p);
  if (conv) {
    if (a->gate) return x2i_set_error(X2I_ERR_ARG, "conv: a gated residual is not part of the convolution epilogues (residual: plain add)");
    if (cd->Cin % 64 || cd->H <= 0 || cd->W <= 0 || cd->KH <= 0 || cd->KW <= 0 || cd->stride <= 0)
      return x2i_set_error(X2I_ERR_SHAPE, "conv: Cin must be a multiple of 64 (Cin=%d)", cd->Cin);
    if (cd->up < 0 || cd->up > 2) return x2i_set_error(X2I_ERR_ARG, "conv: up must be 0, 1 (x2 along H and W) or 2 (x2 along H only), got %d", cd->up);
    const int up_h = cd->up ? 1 : 0, up_w = cd->up == 1 ? 1 : 0, up = up_h | (up_w << 1);
    const int pad_w = cd->pad_w_p1 <= 0 ? cd->pad : cd->pad_w_p1 - 1;
    if (cd->KH * cd->KW > 32) return x2i_set_error(X2I_ERR_SHAPE, "conv: at most 32 filter taps (KH=%d KW=%d)", cd->KH, cd->KW);
    int OH = ((cd->H << up_h) + 2 * cd->pad - cd->KH) / cd->stride + 1;
    if (cd->out_h) {
      if (cd->out_h < 0 || (cd->out_h - 1) * cd->stride - cd->pad >= (cd->H << up_h)) return x2i_set_error(X2I_ERR_SHAPE, "conv: out_h=%d lies outside the input (H=%d pad=%d)", cd->out_h, cd->H, cd->pad);
      OH = cd->out_h;
    }
    int OW = ((cd->W << up_w) + 2 * pad_w - cd->KW) / cd->stride + 1;
    if (cd->out_w) {   // fewer output columns than the symmetric padding gives: the caller's right-hand padding is smaller than pad_w (or larger: zero fill)
      if (cd->out_w < 0 || (cd->out_w - 1) * cd->stride - pad_w >= (cd->W << up_w)) return x2i_set_error(X2I_ERR_SHAPE, "conv: out_w=%d lies outside the input (W=%d pad_w=%d)", cd->out_w, cd->W, pad_w);
      OW = cd->out_w;
    }
    if (a->M != OH * OW || a->K != cd->KH * cd->KW * cd->Cin)
      return x2i_set_error(X2I_ERR_SHAPE, "conv: M=%d K=%d do not match OH*OW=%d, KH*KW*Cin=%d", a->M, a->K, OH * OW, cd->KH * cd->KW * cd->Cin);
    if ((long long)cd->H * cd->W * cd->Cin * 2 >= 0x7f000000LL) return x2i_set_error(X2I_ERR_SHAPE, "conv: image too large");
    p.cH = cd->H; p.cW = cd->W; p.cCin = cd->Cin; p.cOW = OW; p.cKW = cd->KW; p.cStride = cd->stride; p.cPad = cd->pad; p.cPadW = pad_w; p.cUp = up;
    if (cd->moments) {
      if (!cd->moments_scratch) return x2i_set_error(X2I_ERR_ARG, "conv: moments without moments_scratch (x2i_conv_moments_scratch_floats)");
      if (a->N > 2048) return x2i_set_error(X2I_ERR_SHAPE, "conv: moments serve N <= 2048 (N=%d)", a->N);
      // (exactly the kernels' own condition for their whole-line epilogue, where the partial sums are written: with a residual whose ldr is no
      // multiple of 4 they would take the element-wise epilogue and the reduce kernels would sum uninitialised scratch)
      if ((a->N & 7) || (a->ldc & 7) || (a->c_batch_stride & 7) || (((uintptr_t)a->C) & 15) || a->out_f32 || a->C2 || (((uintptr_t)cd->moments_scratch) & 15) ||
          (a->res && (a->ldr & 3)))
        return x2i_set_error(X2I_ERR_ALIGN, "conv: moments need the whole-line bf16 epilogue (N, ldc, c_batch_stride multiples of 8, ldr a multiple of 4, 16-byte aligned C and scratch, no f32 / second output)");
      p.cMom = cd->moments_scratch;
    }
    if (cd->out_row_pitch) {
      if (cd->out_row_pitch < (long long)(OW - 1) * a->ldc + a->N || (cd->out_row_pitch & 7) || a->res || a->C2 || a->out_f32)
        return x2i_set_error(X2I_ERR_SHAPE, "conv: out_row_pitch=%d must hold a row of %d pixels at ldc=%d, be a multiple of 8, and goes with the plain bf16 epilogue only", cd->out_row_pitch, OW, (int)a->ldc);
      p.cRowPitch = cd->out_row_pitch;
    }
  }
  p.tilesM = (a->M + BM - 1) / BM; p.tilesN = (a->N + BN - 1) / BN;
  const bool fast = (a->K % BK == 0) && (conv || a->lda % 8 == 0) && (a->ldw % 8 == 0) && (((uintptr_t)a->A & 15) == 0) &&
                    (((uintptr_t)a->W & 15) == 0) && ((a->a_batch_stride & 7) == 0) &&
                    (conv || (long long)a->M * a->lda * 2 < 0x7f000000LL) && ((long long)a->N * a->ldw * 2 < 0x7f000000LL);
  const bool res = p.res != nullptr, c2 = p.C2 != nullptr, f32 = p.out_f32 != 0;
  X2IOptions& opt = x2i_options();
  kern_t kern = pick_gemm128(p.act, res, f32, c2, conv);
  kern_t kern2 = pick_gemm256l(p.act, res, f32, c2, conv);
  int threads2 = 512;
  if (!conv && opt.gemm_w4) {  // plain GEMMs: the 4-wave kernel with the hand-scheduled K-loop (bit-identical results)
    if (kern_t kw = pick_gemm256w(p.act, res, f32, c2)) kern2 = kw, threads2 = 256;
  }
#ifdef X2I_ABLATION
  if (!conv && opt.gemm_w4 > 1 && p.act == X2I_ACT_NONE && !res && !f32 && !c2) {  // A/B schedules of the 4-wave K-loop
    if (kern_t kv = pick_gemm256w_var(opt.gemm_w4 - 1)) kern2 = kv, threads2 = 256;
  }
  // measurement-only library: the k-half-unit form (gemm_lform = 0) and its ablation variants replace the product kernel
  if (!conv && (opt.gemm_ablate || !opt.gemm_lform)) kern2 = pick_gemm256u(p.act, res, f32, c2, opt.gemm_ablate), threads2 = 512;
#endif
  const int force = opt.gemm_tile;  // 0 = automatic, 128 / 256 = A/B override
  const long long tiles256 = (long long)((a->M + BM2 - 1) / BM2) * ((a->N + BN2 - 1) / BN2) * a->batch;
  // Tile choice (re-measured with the full-line staging kernel, B = 1, 2, 4): the 256^2 kernel wins from about half a round
  // of tiles upwards (1.0-1.38 PF against 0.8-1.0 PF for 128^2 tiles), also when its last round is partly filled; only
  // launches with very few tiles or few rows per batch item (text stream) fill the GPU better with 128^2 tiles.
  const long long min256 = opt.gemm_min256;
  // batched launches with few rows per item (text stream, 512 rows per sample) keep 128^2 tiles below three full rounds
  bool use256 = !conv && a->N >= 256 && (tiles256 >= 768 ? a->M >= 256 : (tiles256 >= min256 && a->M >= 1024));
  // ... except WIDE outputs (N >= 8192: ff_context.net.0 when it is not grouped with the image stream, i.e. in the e4m3 configurations):
  // from 192 tiles up the persistent 256^2 kernel with its stream-K remainder wins (202 -> 145 us at batch 4, 90 -> 70 us at batch 2,
  // profiles/r04o_text_gemm_probe.log); the deep-K, narrow-N text launches (ff_context.net.2, to_add_out) stay on 128^2 tiles.  Both kernels
  // sum in the same order (bit-identical), so this choice may depend on the batch.
  if (!conv && !use256 && a->N >= 8192 && a->M >= 512 && tiles256 >= 192 && opt.gemm_w4 == 1 && opt.gemm_persist) use256 = true;
  // convolutions with >= 256 output channels: the full-line kernel's implicit-GEMM form (option conv256 = 0: 128^2 tiles, A/B)
  bool conv256 = false;
  if (conv && a->N >= 256 && a->N % 8 == 0 && tiles256 >= min256 && a->M >= 1024 && opt.conv256) conv256 = use256 = true;
  // ... on the persistent four-wave core (gemm256c.hip: the linear kernels' hand-scheduled K-loop with the gather as a scalar tap offset + a
  // padding mask per piece row) when the gather is affine in the tap (no fused upsampling), ONE image fits the 2 GB descriptor (batches that do not are launched in chunks)
  // and the epilogue is one it instantiates; bit-identical to the eight-wave form (option conv_w4 = 0)
  // The choice of THESE kernels depends on the batch ITEM's shape only (not on the tile count of the whole batch, which gates the eight-wave
  // form above): in their product K order they sum in another order than the other convolution kernels, and a sample's bits must not depend on
  // the batch it rides in (tests/test_fullscale_parity_gpu.py: batch independence of the LightControl step)
  // ... and only for items with at least 64 tiles (a quarter of the chip per item: the VAE's and ControlNeXt's convolutions at 1024^2); smaller
  // images keep the older kernels, whose smaller tiles fill the chip better there (tests lower the threshold through gemm_min256)
  // grouped weights ride the persistent kernels when all groups' weights fit ONE descriptor (the kernels address them through a per-item offset)
  const bool wgrp_ok = !a->w_batch_stride || (a->w_group > 0 && (a->w_batch_stride & 7) == 0 &&
                                              ((long long)((a->batch - 1) / a->w_group) * a->w_batch_stride + (long long)a->N * a->ldw) * 2 < 0x7f000000LL);
  const long long conv_item_thr = opt.gemm_min256 < 64 ? opt.gemm_min256 : 64;
  const bool conv_item256 = conv && a->N >= 256 && a->N % 8 == 0 && a->M >= 1024 && opt.conv256 &&
                            (long long)((a->M + BM2 - 1) / BM2) * ((a->N + BN2 - 1) / BN2) >= conv_item_thr;
  if (conv_item256 && opt.conv_w4 && fast && cd->up == 0 && !f32 && !c2 && wgrp_ok && a->K >= 3 * BK && (a->ldc & 7) == 0 && (a->c_batch_stride & 7) == 0 &&
      (((uintptr_t)a->C) & 15) == 0 && (long long)a->M < (1LL << 24) &&
      (long long)cd->H * cd->W * cd->Cin * 2 + ((long long)cd->pad * cd->W + pad_w_of(cd)) * cd->Cin * 2 < 0x7f000000LL &&
      (long long)a->M * a->ldc * 2 < 0x7f000000LL && (!cd->out_row_pitch || (long long)(a->M / p.cOW) * cd->out_row_pitch * 2 < 0x7f000000LL) &&
      (!res || ((a->ldr & 7) == 0 && (a->res_batch_stride & 7) == 0 && (((uintptr_t)a->res) & 15) == 0 && (long long)a->M * a->ldr * 2 < 0x7f000000LL))) {
    if (kern_t kc = pick_gemm256c(p.act, res)) {
      int rc = x2i_ensure_dynamic_smem((const void*)kc, SMEM2P_BYTES);
      if (rc) return rc;
      GemmP pm = p;
      pm.tilesM = (a->M + BM2 - 1) / BM2; pm.tilesN = (a->N + BN2 - 1) / BN2;
      pm.gm = pick_gm(pm.tilesN, a->K);
      pm.cMomBlocks = pm.tilesM * 2;
      pm.cKorder = opt.conv_korder ? 1 : 0;
      opt.last_gemm_tile = 5256;   // (read-back for tests: the persistent four-wave convolution kernel)
      return launch_conv_chunks(kc, SMEM2P_BYTES, pm, a, cd, stream, pm.tilesM * 2);
    }
  }
  // ... and with at most 128 output channels (the VAE's last up block, ControlNeXt's 128-wide convolutions): 512 x 128 tiles on the same core
  // (gemm512c.hip: the same wave tile and K-loop, epilogue straight from registers); outputs bit-identical to the 128^2 kernel's
  if (conv && opt.conv_w4 && opt.gemm_tile == 0 && fast && cd->up == 0 && !cd->out_row_pitch && !f32 && !c2 && wgrp_ok && a->K >= 3 * BK && a->N > 64 && a->N <= 128 &&
      (a->N & 7) == 0 && (a->ldc & 7) == 0 && (a->c_batch_stride & 7) == 0 && (((uintptr_t)a->C) & 15) == 0 && a->M >= 2048 && (long long)a->M < (1LL << 24) &&
      (long long)((a->M + 511) / 512) >= conv_item_thr &&
      (long long)cd->H * cd->W * cd->Cin * 2 + ((long long)cd->pad * cd->W + pad_w_of(cd)) * cd->Cin * 2 < 0x7f000000LL &&
      (long long)a->M * a->ldc * 2 < 0x7f000000LL &&
      (!res || ((a->ldr & 3) == 0 && (a->res_batch_stride & 3) == 0 && (((uintptr_t)a->res) & 7) == 0 && (long long)a->M * a->ldr * 2 < 0x7f000000LL))) {
    if (kern_t kc = pick_gemm512c(p.act, res)) {
      int rc = x2i_ensure_dynamic_smem((const void*)kc, SMEM5C_BYTES);
      if (rc) return rc;
      GemmP pm = p;
      pm.tilesM = (a->M + 511) / 512; pm.tilesN = (a->N + 127) / 128;
      pm.cMomBlocks = pm.tilesM * 4;     // (128-row wave tiles, four per 512-row tile)
      pm.cKorder = opt.conv_korder ? 1 : 0;
      opt.last_gemm_tile = 5512;   // (read-back for tests: the persistent 512 x 128 convolution kernel)
      return launch_conv_chunks(kc, SMEM5C_BYTES, pm, a, cd, stream, pm.tilesM * 4);
    }
  }
  // small batches: a launch whose batch item has fewer 256^2 tiles than the chip has CUs, with a deep K, is cut along K over all CUs
  // (parallel split with fix-up, fx_for above); decided by the item's shape alone
  if (!conv && !qd && fast && kern2 && opt.gemm_w4 == 1 && persistent_ok(a, nullptr)) {
    const int cus = x2i_num_cus();
    float* fslabs = nullptr;
    unsigned* fflags = nullptr;
    int frc = X2I_OK, v0 = 0;
    if (fx_for(a, nullptr, cus, &v0, &fslabs, &fflags, &frc)) {
      kern_t kfx = pick_gemm256p_fx();
      int rc = x2i_ensure_dynamic_smem((const void*)kfx, SMEM2P_BYTES);
      if (rc) return rc;
      GemmP pm = p;
      pm.tilesM = (a->M + BM2 - 1) / BM2; pm.tilesN = (a->N + BN2 - 1) / BN2;
      pm.gm = pick_gm(pm.tilesN, a->K);
      pm.nbatch = a->batch;
      pm.fx_v0 = v0; pm.sk_slabs = fslabs; pm.sk_flags = fflags;
      hipLaunchKernelGGL(kfx, dim3((unsigned)cus), dim3(256), SMEM2P_BYTES, stream, pm);
      opt.last_gemm_tile = 3256;
      return x2i_check_launch("gemm (fx)");
    }
    if (frc) return frc;
  }
#ifdef X2I_ABLATION   // (measurement library only since round 6: measured 1.57-1.62x slower than the persistent kernel)
  // A/B (option gemm_r2): the "two residents" form -- 256 x 128 tiles, two workgroups per CU, epilogues hidden behind the other
  // workgroup's K-loop; whole K-tiles in groups of four
  if (opt.gemm_r2 && !conv && !qd && fast && !f32 && a->K % (4 * BK) == 0 && a->M >= 256 && a->N >= 128 && !a->w_batch_stride) {
    const int var = opt.gemm_ablate;
    if (kern_t kr = pick_gemm_r2(p.act, res, f32, c2, var)) {
      const int rc = x2i_ensure_dynamic_smem((const void*)kr, SMEM_R2_BYTES);
      if (rc) return rc;
      GemmP pm = p;
      pm.tilesM = (a->M + 255) / 256; pm.tilesN = (a->N + 127) / 128;
      pm.gm = opt.gemm_gm > 0 ? opt.gemm_gm : 4;
      hipLaunchKernelGGL(kr, dim3(pm.tilesM * pm.tilesN, a->batch), dim3(256), SMEM_R2_BYTES, stream, pm);
      opt.last_gemm_tile = 4256;
      return x2i_check_launch("gemm (r2)");
    }
  }
#endif
  if (force == 128) use256 = false;
  if (force == 256 && !conv) use256 = true;
  if (conv && !conv256) use256 = false;
  if (!kern2) use256 = false;
  if (qd && (qd->H * 128) % BN2) use256 = false;  // a 256-column tile must not straddle the q / k / v sections
  if (qd && !(fast && kern))
    return x2i_set_error(X2I_ERR_SHAPE, "gemm_qkv: the fused epilogue needs K %% 64 == 0 (K=%d), lda/ldw %% 8 == 0 (lda=%d ldw=%d), 16-byte aligned "
                         "A/W and operands below 2 GB", a->K, a->lda, a->ldw);
  if (fast && kern && use256) {
    int rc = x2i_ensure_dynamic_smem((const void*)kern2, SMEM2_BYTES);
    if (rc) return rc;
    // Tile-quantisation fix: with one 256x256 workgroup per CU the launch runs in rounds of 256 tiles; a last round that
    // is less than ~60% full wastes the machine (e.g. M=4x4608, N=3072: 864 tiles = 3.375 rounds).  Peel the trailing
    // rows of every batch item off into a second launch of the 128x128 kernel (2 workgroups per CU, 4x smaller tiles)
    // that fills in behind the last full round.
    const int tm_all = (a->M + BM2 - 1) / BM2, tn = (a->N + BN2 - 1) / BN2;
    const long long per_row = (long long)tn * a->batch;
    const long long full_rounds = tiles256 / 256, rem = tiles256 % 256;
    // persistent form: plain and batched GEMMs whose epilogue leaves as whole bf16 lines (the fused QKV epilogue and f32 outputs stay
    // on the one-tile-per-workgroup kernel); needs >= 3 K-tiles (the first one starts the accumulators, the last two prefetch the
    // next unit)
    kern_t kernp = nullptr;
    if (threads2 == 256 && opt.gemm_persist && opt.gemm_w4 == 1 && a->K >= 3 * BK && !a->w_batch_stride && (a->N & 7) == 0 && (a->ldc & 7) == 0 &&
        (qd || ((a->c_batch_stride & 7) == 0 && ((((uintptr_t)a->C) | ((uintptr_t)a->C2)) & 15) == 0 && (long long)a->M * a->ldc * 2 < 0x7f000000LL)) &&
        ((long long)(a->batch - 1) * a->a_batch_stride + (long long)a->M * a->lda) * 2 < 0x7f000000LL &&
        (!res || ((a->ldr & 7) == 0 && (a->res_batch_stride & 7) == 0 && (((uintptr_t)a->res) & 15) == 0 && (long long)a->M * a->ldr * 2 < 0x7f000000LL)))
      kernp = qd ? pick_gemm256p_qkv() : pick_gemm256p(p.act, res, f32, c2);
    // stream-K: with at least one full round in front, the tiles of a partly filled last round are cut along K and dealt out over all
    // workgroups (chained partial accumulators: the same summation order, bit-identical results) instead of being peeled into a
    // 128^2 launch.  Shares stay below one tile (share + 6 <= nk) and segments at or above 6 K-tiles.
    const int cus = x2i_num_cus();
    const int nkt = a->K / BK;
    bool sk = false;
    float* sk_slabs = nullptr;
    unsigned* sk_flags = nullptr;
    // ... and only when the segments of a tile can run at different places of the workgroups' tile lists (segments per tile
    // = cus / r <= whole tiles per workgroup + 1): with fewer whole tiles the chain of hand-offs serialises (measured: M = 2048,
    // N = 9216, 288 tiles -> 8-segment chains behind ONE whole tile ran at half the speed of the peeled form)
    if (kernp) {
      sk = streamk_for(a, tiles256, nkt, cus, &sk_slabs, &sk_flags, &rc);
      if (rc) return rc;
    }
    int tm_main = tm_all;
    // (re-measured in round 2 with a sweep of last-round fill levels: peeling pays up to a 3/8-full last round at any depth, and for a half-full one
    // only behind >= 8 full rounds; a fuller last round is faster left in the one launch.  Either way the results are bit-identical.)
    if (!sk && force == 0 && !conv && full_rounds >= 1 && rem > 0 && (rem <= 96 || (rem <= 128 && full_rounds >= 8)) && opt.gemm_split_tail) {
      const long long tm_fit = (full_rounds * 256) / per_row;
      if (tm_fit >= 1 && tm_fit < tm_all) tm_main = (int)tm_fit;
    }
    GemmP pm = p;
    pm.gm = pick_gm(tn, a->K);
    pm.M = (tm_main < tm_all) ? tm_main * BM2 : a->M;
    pm.tilesM = tm_main; pm.tilesN = tn;
    pm.cMomBlocks = tm_main * 2;   // (conv: 128-row wave tiles, two per 256-row tile)
    if (kernp) {
      rc = x2i_ensure_dynamic_smem((const void*)kernp, SMEM2P_BYTES);
      if (rc) return rc;
      pm.nbatch = a->batch;  // one tile list over all batch items
      pm.sk_on = sk ? 1 : 0; pm.sk_slabs = sk_slabs; pm.sk_flags = sk_flags;
      const long long tiles = (long long)pm.tilesM * pm.tilesN * a->batch;
      hipLaunchKernelGGL(kernp, dim3((unsigned)(tiles < cus ? tiles : cus)), dim3(256), SMEM2P_BYTES, stream, pm);
    } else {
      hipLaunchKernelGGL(kern2, dim3(pm.tilesM * pm.tilesN, a->batch), dim3(threads2), SMEM2_BYTES, stream, pm);
    }
    opt.last_gemm_tile = 256 + (tm_main < tm_all ? 1000 : 0);
    if (tm_main < tm_all) {
      rc = x2i_ensure_dynamic_smem((const void*)kern, 4 * TILE_BYTES);
      if (rc) return rc;
      GemmP pt = p;
      const long long r0 = (long long)tm_main * BM2;
      pt.A = p.A + r0 * p.lda;
      pt.C = p.out_f32 ? (void*)((float*)p.C + r0 * p.ldc) : (void*)((bf16_t*)p.C + r0 * p.ldc);
      if (p.C2) pt.C2 = p.C2 + r0 * p.ldc;
      if (p.res) pt.res = p.res + r0 * p.ldr;
      pt.M = a->M - (int)r0;
      pt.q_row0 = (int)r0;
      pt.tilesM = (pt.M + BM - 1) / BM; pt.tilesN = (a->N + BN - 1) / BN;
      hipLaunchKernelGGL(kern, dim3(pt.tilesM * pt.tilesN, a->batch), dim3(256), 4 * TILE_BYTES, stream, pt);
    }
  } else if (fast && kern) {
    const int rc = x2i_ensure_dynamic_smem((const void*)kern, 4 * TILE_BYTES);
    if (rc) return rc;
    dim3 grid(p.tilesM * p.tilesN, a->batch);
    p.cMomBlocks = p.tilesM * 2;    // (conv: 64-row wave tiles, two per 128-row tile)
    hipLaunchKernelGGL(kern, grid, dim3(256), 4 * TILE_BYTES, stream, p);
    opt.last_gemm_tile = 128;
  } else if (conv) {
    return x2i_set_error(X2I_ERR_SHAPE, "conv: unsupported epilogue/alignment combination");
  } else {
    // generic kernel: tiny / unaligned problems only
    const double flop = 2.0 * a->M * (double)a->N * a->K * a->batch;
    if (flop > NAIVE_MAX_FLOP)
      return x2i_set_error(X2I_ERR_SHAPE,
                           "gemm: M=%d N=%d K=%d batch=%d has no MFMA path (%s) and is too large for the generic kernel", a->M, a->N,
                           a->K, a->batch,
                           !kern ? "epilogue combination act+residual / act+f32 / act+C2 / residual+f32 is not instantiated"
                                 : "needs K %% 64 == 0, lda/ldw %% 8 == 0, 16-byte aligned A/W, operands below 2 GB");
    const long long rows = (long long)a->M * a->batch;
    const int gy = (int)(rows < 32768 ? rows : 32768), gz = (int)((rows + gy - 1) / gy);
    dim3 grid((a->N + 127) / 128, gy, gz);
    hipLaunchKernelGGL(gemm_naive_kernel, grid, dim3(128), 0, stream, p, a->batch);
    opt.last_gemm_tile = 0;
  }
  if (conv && cd->moments) return conv_moments_tail(a, cd, stream, (opt.last_gemm_tile == 128 ? (a->M + BM - 1) / BM : (a->M + BM2 - 1) / BM2) * 2);
  return x2i_check_launch("gemm");
}


// ---------------------------------------------------------------------------------------------------------------------
// fp8 (e4m3) operands: x2i_gemm_fp8.  One kernel family (256^2 tiles, gemm256_fp8.hip); shapes it does not serve are refused
// with a message -- the host keeps those GEMMs on the bf16 path (there is no silent slow fallback).
static int check_qkv_desc(const x2i_gemm_args* a, const x2i_qkv_desc* qd, const char* who) {
  if (!qd->norm_q || !qd->norm_k || !qd->cos || !qd->Q || !qd->K || !qd->VT)   // (sin == NULL: `cos` is the pair-form table, include/x2i.h)
    return x2i_set_error(X2I_ERR_ARG, "%s: null pointer in descriptor", who);
  if (qd->H <= 0 || a->N != 3 * qd->H * 128) return x2i_set_error(X2I_ERR_SHAPE, "%s: N=%d must be 3*H*128 (H=%d)", who, a->N, qd->H);
  if (qd->Spad % 128 || qd->rows_per_sample <= 0 || qd->tok_off < 0 || qd->tok_off + qd->rows_per_sample > qd->Spad)
    return x2i_set_error(X2I_ERR_SHAPE, "%s: bad token geometry (tok_off=%d rows_per_sample=%d Spad=%d)", who, qd->tok_off,
                         qd->rows_per_sample, qd->Spad);
#ifdef X2I_ABLATION
  const bool extra = false;  // (measurement library: bias2 carries the timestamp buffer of tools/gemm_unit_timeline.py)
#else
  const bool extra = a->bias2 != nullptr;
#endif
  if (a->act || a->res || a->gate || a->C2 || a->out_f32 || extra)
    return x2i_set_error(X2I_ERR_ARG, "%s: only the plain bias epilogue can be fused", who);
  if ((((uintptr_t)qd->Q | (uintptr_t)qd->K | (uintptr_t)qd->VT | (uintptr_t)qd->norm_q | (uintptr_t)qd->norm_k | (uintptr_t)qd->cos |
        (uintptr_t)qd->sin) & 15) != 0)
    return x2i_set_error(X2I_ERR_ALIGN, "%s: descriptor pointers must be 16-byte aligned", who);
  return X2I_OK;
}

static int launch_gemm_fp8_impl(const x2i_gemm_args* a, const x2i_fp8_desc* f, const x2i_qkv_desc* qd, hipStream_t stream);
int x2i_launch_gemm_fp8(const x2i_gemm_args* a, const x2i_fp8_desc* f, hipStream_t stream) {
  return launch_gemm_fp8_impl(a, f, nullptr, stream);
}
int x2i_launch_gemm_qkv_fp8(const x2i_gemm_args* a, const x2i_fp8_desc* f, const x2i_qkv_desc* qd, hipStream_t stream) {
  if (!a || !f || !qd) return x2i_set_error(X2I_ERR_ARG, "gemm_qkv_fp8: null pointer");
  if (const int rc = check_qkv_desc(a, qd, "gemm_qkv_fp8")) return rc;
  if (f->out_fp8) return x2i_set_error(X2I_ERR_ARG, "gemm_qkv_fp8: Q / K / V^T are written as bf16");
  if ((qd->H * 128) % BN2) return x2i_set_error(X2I_ERR_SHAPE, "gemm_qkv_fp8: H*128 must be a multiple of 256 (H=%d)", qd->H);
  return launch_gemm_fp8_impl(a, f, qd, stream);
}

static int launch_gemm_fp8_impl(const x2i_gemm_args* a, const x2i_fp8_desc* f, const x2i_qkv_desc* qd, hipStream_t stream) {
  if (!a || !f || !a->A || !a->W || (!a->C && !qd)) return x2i_set_error(X2I_ERR_ARG, "gemm_fp8: null pointer");
  if (a->M <= 0 || a->N <= 0 || a->K <= 0 || a->batch <= 0) return x2i_set_error(X2I_ERR_SHAPE, "gemm_fp8: bad shape M=%d N=%d K=%d batch=%d", a->M, a->N, a->K, a->batch);
  if (a->gate && !a->res) return x2i_set_error(X2I_ERR_ARG, "gemm_fp8: gate without residual");
  if (a->C2 || a->out_f32 || a->w_batch_stride || a->w_group) return x2i_set_error(X2I_ERR_ARG, "gemm_fp8: C2 / f32 output / per-batch W are not supported");
  if (a->K % 128 || a->lda % 16 || a->ldw % 16 || (a->a_batch_stride & 15) || (((uintptr_t)a->A | (uintptr_t)a->W) & 15))
    return x2i_set_error(X2I_ERR_ALIGN, "gemm_fp8: needs K %% 128 == 0 (K=%d), lda / ldw / a_batch_stride %% 16 == 0, 16-byte aligned A / W", a->K);
  if ((long long)a->M * a->lda >= 0x7f000000LL || (long long)a->N * a->ldw >= 0x7f000000LL)
    return x2i_set_error(X2I_ERR_SHAPE, "gemm_fp8: operand larger than 2 GB per batch item");
  const bool out8 = f->out_fp8 != 0, res = a->res != nullptr;
  const int nal = out8 ? 15 : 7;
  if ((a->N & nal) || (!qd && ((a->ldc & nal) || (a->c_batch_stride & nal) || (((uintptr_t)a->C) & 15))) || (res && ((a->ldr & 7) || (a->res_batch_stride & 7))))
    return x2i_set_error(X2I_ERR_ALIGN, "gemm_fp8: N, ldc and c_batch_stride must be multiples of %d, C 16-byte aligned", nal + 1);
  if (out8 && (res || a->gate)) return x2i_set_error(X2I_ERR_ARG, "gemm_fp8: e4m3 output has no residual form");
  kern_t kern = pick_gemm256_fp8(a->act, res, out8);
  if (!kern) return x2i_set_error(X2I_ERR_ARG, "gemm_fp8: epilogue (act=%d%s%s) is not instantiated", a->act, res ? ", residual" : "", out8 ? ", e4m3 out" : "");
  GemmP p;
  p.A = (const bf16_t*)a->A; p.a_bs = a->a_batch_stride; p.lda = a->lda;
  p.W = (const bf16_t*)a->W; p.ldw = a->ldw; p.w_bs = 0; p.wdiv = 1; p.bias_gs = 0;
  p.bias = (const bf16_t*)a->bias;
  p.C = a->C; p.c_bs = a->c_batch_stride; p.ldc = a->ldc;
  p.C2 = nullptr; p.act2 = 0;
  p.gate = a->gate; p.gate_bs = a->gate_batch_stride;
  p.res = (const bf16_t*)a->res; p.r_bs = a->res_batch_stride; p.ldr = a->ldr;
  p.bias2 = a->bias2; p.bias2_bs = a->bias2_batch_stride;
  p.M = a->M; p.N = a->N; p.K = a->K; p.act = a->act; p.out_f32 = 0;
  p.cH = p.cW = p.cCin = p.cOW = p.cKW = p.cStride = p.cPad = p.cUp = p.cPadW = p.cRowPitch = p.cKorder = 0; p.cMom = nullptr; p.cMomBlocks = 0;
  p.q_on = 0; p.q_H = p.q_Spad = p.q_tok_off = p.q_rpb = p.q_row0 = p.q_vperm = 0; p.q_eps = 0.f; p.q_qs = 1.f;
  p.q_nq = p.q_nk = nullptr; p.q_cos = p.q_sin = nullptr; p.q_Q = p.q_K = p.q_VT = nullptr;
  if (qd) {
    p.q_on = 1; p.q_H = qd->H; p.q_Spad = qd->Spad; p.q_tok_off = qd->tok_off; p.q_rpb = qd->rows_per_sample; p.q_eps = qd->eps; p.q_qs = qd->q_scale == 0.f ? 1.f : qd->q_scale;
    p.q_nq = (const bf16_t*)qd->norm_q; p.q_nk = (const bf16_t*)qd->norm_k; p.q_cos = qd->cos; p.q_sin = qd->sin;
    p.q_Q = (bf16_t*)qd->Q; p.q_K = (bf16_t*)qd->K; p.q_VT = (bf16_t*)qd->VT; p.q_vperm = qd->vt_perm ? 1 : 0;
    p.ldc = a->N; p.c_bs = 0;  // C is never written
  }
  p.f_sa = f->a_scale; p.f_sa_bs = f->a_scale_batch_stride; p.f_sw = f->w_scale; p.f_alpha = f->alpha; p.f_oinv = f->out_inv_scale;
  p.f_out8 = out8 ? 1 : 0;
  p.nbatch = 1; p.sk_on = 0; p.sk_slabs = nullptr; p.sk_flags = nullptr; p.fx_v0 = 0;
#ifdef X2I_ABLATION
  p.act2 = a->act2;   // (measurement library: the unit-timeline hooks of gemm256p.hip, tools/gemm_unit_timeline.py --fp8)
#endif
  const int tm = (a->M + BM2 - 1) / BM2, tn = (a->N + BN2 - 1) / BN2;
  p.tilesM = tm; p.tilesN = tn;
  const X2IOptions& opt = x2i_options();
  if (opt.gemm_gm > 0) p.gm = opt.gemm_gm;
  else if (tn <= 16) p.gm = a->K >= 16384 ? 1 : 4;
  else if (tn <= 64) p.gm = 6;
  else p.gm = 2;
  // persistent four-wave form (gemm256p.hip with the K-loop of gen_gemm256f8.py): one workgroup per CU walks the tile list of all
  // batch items, the last partly filled round is cut along K and chained through the caller's workspace -- everything the bf16
  // launches do.  Same accumulation order and epilogue arithmetic as the one-tile kernel below: bit-identical (tested).
  X2IOptions& wopt = x2i_options();
  const int ob = out8 ? 1 : 2, oal = out8 ? 15 : 7;
  kern_t kernp = nullptr;
  if (wopt.gemm_persist && wopt.gemm_fp8_persist && wopt.gemm_tile == 0 && a->K >= 3 * 128 && (((uintptr_t)f->w_scale) & 15) == 0 &&
      ((long long)(a->batch - 1) * a->a_batch_stride + (long long)a->M * a->lda) < 0x7f000000LL &&
      (qd || ((a->ldc & oal) == 0 && (a->c_batch_stride & oal) == 0 && (long long)a->M * a->ldc * ob < 0x7f000000LL)) &&
      (!res || ((((uintptr_t)a->res) & 15) == 0 && (long long)a->M * a->ldr * 2 < 0x7f000000LL)) &&
      (!qd || (long long)a->batch * a->M <= 0x7fffffffLL))
    kernp = pick_gemm256p_fp8(a->act, res, out8, qd != nullptr);
  if (kernp) {
    int rc = x2i_ensure_dynamic_smem((const void*)kernp, SMEM2P_BYTES);
    if (rc) return rc;
    const int cus = x2i_num_cus();
    const long long tiles = (long long)tm * tn * a->batch;
    float* sk_slabs = nullptr;
    unsigned* sk_flags = nullptr;
    const bool sk = streamk_for(a, tiles, a->K / 128, cus, &sk_slabs, &sk_flags, &rc);
    if (rc) return rc;
    p.nbatch = a->batch;
    p.sk_on = sk ? 1 : 0; p.sk_slabs = sk_slabs; p.sk_flags = sk_flags;
    hipLaunchKernelGGL(kernp, dim3((unsigned)(tiles < cus ? tiles : cus)), dim3(256), SMEM2P_BYTES, stream, p);
    wopt.last_gemm_tile = sk ? 9256 : 8256;   // (read-back for tests: persistent e4m3 launch, with / without stream-K)
    return x2i_check_launch("gemm_fp8 (persistent)");
  }
  const int rc = x2i_ensure_dynamic_smem((const void*)kern, SMEM2_BYTES);
  if (rc) return rc;
  hipLaunchKernelGGL(kern, dim3(tm * tn, a->batch), dim3(512), SMEM2_BYTES, stream, p);
  wopt.last_gemm_tile = 7256;
  return x2i_check_launch("gemm_fp8");
}
