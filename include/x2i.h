/* x2i.h -- C ABI of libx2i_hip.so: the MI355X (gfx950) implementation of the X2I sampling hot path.
 *
 * The reference (OPPO-Mente-Lab/X2I) is pure Python and has NO FFI layer; the boundary it exposes for this
 * path is a set of Python call signatures.  Each entry point below names the reference interface it stands
 * behind (file:line in /root/reference); the ctypes binding that puts it behind the reference's call sites is
 * x2i_amd/_lib.py, and INTEGRATION.md shows the stub a reference maintainer would add.
 *
 * Conventions
 *   - plain pointers + sizes only; every pointer is a DEVICE pointer owned by the caller (PyTorch allocates
 *     inputs, outputs and workspace); the library borrows them for the duration of the call
 *   - `stream` is a hipStream_t passed as void*; work is enqueued, never synchronised (hipGraph-capturable)
 *   - bf16 tensors are raw uint16 storage; "f32" means IEEE float
 *   - every function returns 0 on success, a negative X2I_ERR_* code otherwise; x2i_last_error() returns a
 *     thread-local message.  Nothing exits or throws across the boundary.
 *   - one process per GPU.  The library owns NO device memory: there are no model handles, no weight packing step and no
 *     library-owned workspace ("PyTorch allocates and owns every buffer ... workspace via a
 *     workspace_size query") -- weights are consumed in the reference's own nn.Linear [N,K] layout, fused only by
 *     row-concatenation on the host side, and every scratch buffer is a caller-owned argument sized by a query:
 *     x2i_groupnorm_scratch_floats, x2i_streamk_workspace_bytes (x2i_gemm_args.workspace, x2i_attention_vp_ws_bf16).  Process-wide state, all of it
 *     mutex-protected: the option table below and a per-kernel "dynamic LDS size already raised" cache.  (The measurement library additionally keeps one
 *     side stream with two events per device for the two-stream A/B form of the attention backward.)
 *   - ABI version 5 (x2i_abi_version; 5 adds x2i_attention_vp_ws_bf16 -- no struct changed; 4 appended `w_group` to x2i_gemm_args, 0 = what version 3 did, and added the *_grouped entry points).  Since version 1: x2i_gemm_args grew `workspace` / `workspace_bytes`, x2i_qkv_desc `q_scale`
 *     and x2i_conv_desc a ninth field (version 2); version 3 re-defines that field as `pad_w_p1` (0 = same padding as `pad`, so that a
 *     zero-initialised descriptor means what it meant in version 1), appends `out_w`, `out_h`, `out_row_pitch` (0 = computed / dense) and the `moments` fields (NULL = off) to it, gives `up` the value 2, and appends `vt_perm` to x2i_qkv_desc (0 = the old layout).  A caller built against another version must not load this
 *     library (x2i_amd/_lib.py checks).
 */
#ifndef X2I_H
#define X2I_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define X2I_ABI_VERSION 5

#define X2I_OK 0
#define X2I_ERR_ARG (-1)
#define X2I_ERR_SHAPE (-2)
#define X2I_ERR_ALIGN (-3)
#define X2I_ERR_HIP (-4)
#define X2I_ERR_STATE (-5)

/* activation codes */
#define X2I_ACT_NONE_ 0
#define X2I_ACT_GELU_TANH_ 1 /* nn.GELU(approximate="tanh"): lightcontrol_flux.py:65, FeedForward "gelu-approximate" */
#define X2I_ACT_GELU_ERF_ 2  /* nn.GELU(): utils/proj.py:19,23 */
#define X2I_ACT_SILU_ 3
#define X2I_ACT_RELU_ 4

typedef void* x2i_stream_t;

int x2i_abi_version(void);
const char* x2i_last_error(void);

/* A/B and tuning switches.  Defaults are the product configuration; each option is initialised ONCE (first use) from the
 * environment variable X2I_<NAME> and afterwards only changes through x2i_set_option -- nothing on the launch path reads
 * the environment.  Names: "gemm_tile" (0 auto | 128 | 256), "gemm_min256", "gemm_gm" (0 auto), "gemm_split_tail" (1),
 * "gemm_w4" (1: 4-wave hand-scheduled 256^2 kernel; 0: the 8-wave form), "gemm_persist" (1: one workgroup per CU walks the output
 * tiles), "gemm_fp8_persist" (1: x2i_gemm_fp8 / x2i_gemm_qkv_fp8 take the persistent four-wave form too; 0: the one-tile e4m3 kernel, bit-identical),
 * "gemm_fx_nk" (96: the K split of "gemm_fx" applies from this many 64-wide K-tiles, i.e. K >= 6144; 48 adds the K = 3072 gated-residual launches:
 *  +0.5 % at 512^2 batch 1, measured, not the default),
 * "gemm_fx" (DEFAULT 2: as 1, but only for items with at most HALF a round of 256^2 tiles -- a 512^2 sample's 72; 0: never; 1: a gated-residual launch whose batch ITEM has fewer 256^2 output tiles than the chip has CUs, with K >= 6144, is cut along K
 * over all CUs -- every part summed from zero in parallel, the parts of a tile added in a fixed order by the workgroup that holds the last
 * one; needs the workspace.  Decided and cut by the item's shape alone, so a sample's result does not depend on its batch; deterministic;
 * NOT bit-identical to the whole-tile kernels (another association of the K sum, same tolerance).  0: whole tiles), "gemm_streamk" (1: the persistent kernel cuts the tiles of the last, partly filled round along K and chains the segments
 * through the CALLER's workspace, x2i_gemm_args.workspace -- bit-identical to the one-tile kernel; 0, or no workspace: the peeled
 * 128^2 tail launch), "gemm_pair" (1: x2i_gemm_pair_bf16 / x2i_gemm_qkv_pair_bf16 group their two problems into one launch when they can),
 * "attn_w16" (1: x2i_attention_prefers_vt_perm may answer 1 -- the sampling path then uses the 16 x 16 x 32 attention kernel; 0: never; 2: at any size -- tests),
 * "attn_bwd_overlap" (1: the dQ and the dK / dV pass of x2i_attention_bwd_bf16 run as ONE launch, dQ blocks in front, so that their partly filled last
 * rounds fill each other; 0: one after the other; bit-identical),
 * "attn_streamk" (1: the 16 x 16 x 32 attention kernel runs persistent -- one workgroup per CU, each unit's exit requests the next item's first loads -- for
 * launches of more than one round, and x2i_attention_vp_ws_bf16 cuts the items of a partly filled last round along the key axis, chained through the
 * workspace; 0: one workgroup per item; bit-identical),
 * "train_rows_wg" (1: x2i_ln_mod_bwd_bf16 / x2i_gate_bwd_bf16 run a workgroup per row group with a thread per eight columns; 0: a wave per row --
 * same values up to the summation order of the row statistics),
 * "conv256" (1), "conv_w4" (1: convolutions with >= 256 output channels take the persistent four-wave kernel with the hand-scheduled K-loop,
 * csrc/gemm256c.hip; 0: the eight-wave one-tile form -- bit-identical), "attn_variant" (0 auto; 4 = 4-wave kernel, 8 = the 8-wave ping-pong
 * kernel, 9 = the hand-scheduled one-wave-per-SIMD kernel for any scale, 12 = the hand-scheduled kernel on 16x16x32 MFMAs (V^T span-permuted by the
 * caller -- tools and tests); the A/B forms 1..3, 5..7, 10, 11 exist only in the measurement library since round 6 and mean "automatic" here),
 * "conv5_variant" (0),
 * "fp8" (0; 2 = x2i_ln_modulate_fp8 keeps its per-row kernel at D = 3072, bit-identical A/B); "last_gemm_tile" is a read-back for tests: the tile edge of the kernel the
 * last GEMM / conv launch took (256, 128, 0 = generic kernel; +1000 = a peeled 128^2 tail launch followed).  Unknown names
 * return X2I_ERR_ARG.  Every setting selects between
 * kernels with identical results (bit-identical where the tests say so); the measurement-only kernels ("wrong results by
 * design" ablations) are NOT in this library -- they are compiled only into libx2i_hip_ablate.so (-DX2I_ABLATION), where
 * x2i_is_ablation_build() returns 1 and the extra options "gemm_lform", "gemm_ablate", "attn_ablate", "gemm_r2" (the "two residents" GEMM form,
 * csrc/gemm_r2.hip: measured 1.6x slower) exist together with the A/B attention forms (csrc/attention16.hip, the ping-pong schedules 0 / 1). */
int x2i_set_option(const char* name, int64_t value);
int x2i_get_option(const char* name, int64_t* value);
int x2i_is_ablation_build(void);

/* ---------------------------------------------------------------------------------------------------------
 * nn.Linear with fused epilogue.   C[z] = epi(A[z] W^T)
 *   v = acc + bias[n]; v = act(v); if (res) v = res[z][m][n] + (gate ? gate[z][n] : 1) * v; C = v; C2 = act2(v)
 * Stands behind: every nn.Linear on the path -- lightcontrol_flux.py:64,66,256,257,282; diffusers Attention
 * to_q/k/v/add_*_proj/to_out/to_add_out and FeedForward (ctor args lightcontrol_flux.py:69-80,135-153);
 * utils/proj.py:18-25.  The gate/residual form is `hidden + gate.unsqueeze(1) * linear(...)`
 * (lightcontrol_flux.py:98-100,180-181,186-189,193-200).
 * A: bf16 [M,K] (row stride lda, batch stride a_batch_stride elements); W: bf16 [N,K] (row stride ldw);
 * C: bf16 (or f32 when out_f32) row stride ldc; C2 optional bf16 (same strides as C); gate: f32 [batch][N].
 * res may alias C.  Fast path needs K % 64 == 0 and 16-byte aligned rows; anything else takes a slow
 * generic kernel. */
typedef struct x2i_gemm_args {
  const void* A; int64_t a_batch_stride; int32_t lda;
  const void* W; int32_t ldw;
  const void* bias;
  void* C; int64_t c_batch_stride; int32_t ldc;
  void* C2; int32_t act2;
  const float* gate; int64_t gate_batch_stride;
  const void* res; int64_t res_batch_stride; int32_t ldr;
  const float* bias2; int64_t bias2_batch_stride; /* optional f32 [batch][N] added before act (time-embedding bias) */
  int64_t w_batch_stride;                            /* 0 = one W for every batch item (nn.Linear); else W[z] = W + z*stride (q k^T) */
  int32_t M, N, K, batch;
  int32_t act; int32_t out_f32;
  void* workspace;                                   /* optional caller-owned stream-K workspace (below); NULL: none */
  int64_t workspace_bytes;
  int32_t w_group;                                   /* > 0 (needs w_batch_stride != 0): GROUPED weights, see below; 0: off */
  int32_t reserved1;
} x2i_gemm_args;
/* Grouped weights (w_group > 0): w_group consecutive batch items share one W and one bias,
 *     W[z] = W + (z / w_group) * w_batch_stride,   bias[z] = bias + (z / w_group) * N.
 * The 19 ControlNeXt models of a LightControl step (lightcontrol_flux.py:504-507: control_nets[i] behind block i) x B samples ride in one launch
 * this way (x2i_amd/lightcontrol.py: ControlNeXtBank).  Served by x2i_gemm_bf16 and x2i_conv2d_nhwc_bf16 with bf16 outputs (no C2 / f32 output;
 * the fused-QKV and fp8 entry points refuse it).  An item's results are bit-identical to those of a launch of that item alone with its W and bias. */
int x2i_gemm_bf16(const x2i_gemm_args* args, x2i_stream_t stream);

/* Stream-K workspace of the persistent GEMM kernel (csrc/gemm256p.hip).  A launch whose last round of 256 x 256 output tiles is
 * partly filled cuts those tiles along K into segments that are chained from workgroup to workgroup: a segment parks its fp32
 * accumulators in the workspace and the next one continues them (same summation order as an undivided tile: bit-identical
 * results).  The workspace is CALLER-OWNED device memory of at least x2i_streamk_workspace_bytes() bytes, 256-byte aligned,
 * ZERO-FILLED ONCE before its first use (the kernels re-arm it themselves), passed in x2i_gemm_args.workspace by every entry point
 * that takes x2i_gemm_args (for the *_pair_* entry points: args0's).  One workspace per stream (and per captured graph), never
 * shared -- and two workspace-carrying launches must not OVERLAP on one device at all: a launch with chained segments assumes its
 * workgroups (one per CU) are co-resident; two of them on two streams can starve each other's predecessors until the bounded spin
 * gives up (marker below).  The caller orders such launches (x2i_amd/ops.py: a stream that takes over waits for the previous one;
 * graphs captured with a workspace are replayed on one stream).  Without a workspace
 * (NULL / too small = X2I_ERR_ARG) such launches peel the partly filled round into a second launch of the 128^2 kernel: same
 * results, a few percent slower.
 * x2i_streamk_workspace_status: a SYNCHRONISING read of the workspace's give-up marker (a chained segment waits for its
 * predecessor with a bounded spin; if it ever gives up -- never observed -- the marker is set and the results of that launch are
 * undefined).  Returns 0 = clean, 1 = a segment gave up, negative X2I_ERR_* on a HIP error.  Callers check it after a batch of
 * launches and treat 1 as a hard error (x2i_amd: FluxPipeline, harness and bench.py raise). */
int64_t x2i_streamk_workspace_bytes(void);
int x2i_streamk_workspace_status(const void* workspace, int64_t workspace_bytes);

/* ---------------------------------------------------------------------------------------------------------
 * fp8 path (BASELINE north_star: "MFMA bf16/fp8 for the QKV/out-proj and MLP GEMMs"; the reference has no fp8 code -- the
 * linears this stands behind are the same as x2i_gemm_bf16's, first of all FeedForward ff.net.0 / ff.net.2 and the single
 * blocks' proj_mlp / proj_out, lightcontrol_flux.py:64-66,150-153).  Operands are OCP e4m3fn bytes; the product is
 *     C[z][m][n] = epi( a_scale[z][m] * w_scale[n] * alpha * sum_k A8[z][m][k] * W8[n][k] )
 * with the epilogue of x2i_gemm_bf16 (bias, GELU-tanh, gate * v + residual; bf16 output) or, with out_fp8, an e4m3 output
 * sat(epi * out_inv_scale) that is the next GEMM's A operand.  `args` is read as for x2i_gemm_bf16 with A / W pointing at e4m3
 * bytes and lda / ldw / K counted in elements (= bytes); ldc counts output elements.  Served shapes: K % 128 == 0, lda / ldw %
 * 16 == 0, N % 8 == 0 (N % 16 for out_fp8); anything else returns X2I_ERR_ALIGN / X2I_ERR_SHAPE -- callers keep such linears on
 * the bf16 entry point.  bf16 remains the default arithmetic of the path; fp8 is opt-in per model (x2i_amd: fp8= argument). */
typedef struct x2i_fp8_desc {
  const float* a_scale;          /* f32 [batch][M] per-row dequantisation scale of A (NULL: 1) */
  int64_t a_scale_batch_stride;
  const float* w_scale;          /* f32 [N] per-output-channel dequantisation scale of W (NULL: 1) */
  float alpha;                   /* scalar factor on the product (static per-tensor activation scale), normally 1 */
  int32_t out_fp8;               /* 1: C is e4m3 */
  float out_inv_scale;           /* e4m3 output = sat(value * out_inv_scale); the consumer passes alpha = 1 / out_inv_scale */
} x2i_fp8_desc;
int x2i_gemm_fp8(const x2i_gemm_args* args, const x2i_fp8_desc* fp8, x2i_stream_t stream);

/* Row-wise e4m3 quantisation of a bf16 matrix x [rows][cols] (row stride ldx elements): scale[r] = amax(|x[r]|) / 448 (1 when the
 * row is zero), y[r][c] = e4m3_rne(x[r][c] / scale[r]).  Used once per weight (per-output-channel scales) and for activations
 * that have no producing kernel to fuse into.  scale == NULL: static quantisation y = sat(x * static_inv_scale).  cols % 8 == 0. */
int x2i_quantize_rows_fp8(const void* x, int64_t rows, int32_t cols, int64_t ldx, void* y, int64_t ldy, float* scale,
                          float static_inv_scale, x2i_stream_t stream);

/* x2i_ln_modulate_bf16 with an additional e4m3 output: Y8[b][s][:] = e4m3(y / row_scale[b*S + s]), row_scale = amax(|y|) / 448,
 * where y is the modulated LayerNorm output (the A operand of the following fp8 GEMM).  Y (bf16) may be NULL when no bf16
 * consumer exists (the double blocks' feed-forward norm); Y8 row stride ldy8 bytes. */
int x2i_ln_modulate_fp8(const void* X, int64_t x_bs, int32_t ldx, void* Y, int64_t y_bs, int32_t ldy, void* Y8, int64_t y8_bs,
                        int32_t ldy8, float* row_scale, int32_t B, int32_t S, int32_t D, int32_t S0, const float* shift0,
                        const float* scale0, const float* shift1, const float* scale1, int64_t mod_bs, float eps,
                        x2i_stream_t stream);

/* ---------------------------------------------------------------------------------------------------------
 * nn.Conv2d as an implicit GEMM on NHWC bf16 activations (ControlNeXt hint encoder, lightcontrol_flux.py:593-668,
 * diffusers ResnetBlock2D / Downsample2D convs).  `args` is the GEMM view: A = input [batch][H][W][Cin] (a_batch_stride
 * = H*W*Cin; lda unused), W = weight repacked to [Cout][KH][KW][Cin] (ldw = KH*KW*Cin), M = OH*OW per batch,
 * N = Cout, K = KH*KW*Cin, C = output [batch][OH*OW][Cout] (NHWC) with the usual epilogue (bias, bias2 = per-sample
 * time-embedding term, ReLU, residual add).  Cin must be a multiple of 64; zero padding comes from the buffer
 * descriptor's out-of-range rule.  The 3-channel stem conv has its own entry point below. */
typedef struct x2i_conv_desc {
  int32_t H, W, Cin, KH, KW, stride, pad;
  int32_t up; /* 1: F.interpolate(scale_factor=2, mode="nearest") fused in front of the conv (diffusers Upsample2D); 2: the same along H
               * only (rows doubled, columns as they are) -- for the column-phase form of an upsampling convolution: output columns 2x + px
               * of Upsample2D's 3 x 3 conv see source columns {x - 1, x} (px = 0) or {x, x + 1} (px = 1) only, so each phase is a 3 x 2 conv
               * with the coinciding taps' weights added, 6/9 of the multiply-adds (x2i_amd/vae.py) */
  int32_t pad_w_p1; /* 0 (a zero-initialised descriptor): `pad` pads both dimensions, nn.Conv2d(padding=int); otherwise the padding
                     * along W is pad_w_p1 - 1 and `pad` is the padding along H (nn.Conv2d(padding=(pad, pad_w_p1 - 1))) */
  int32_t out_w;    /* 0: the output has (W' + 2 pad_w - KW) / stride + 1 columns (W' = W, or 2 W with up = 1).  Otherwise: exactly out_w columns --
                     * left padding pad_w, right padding whatever out_w implies (zero fill): asymmetric padding, M = OH * out_w */
  int32_t out_h;    /* the same for the rows: 0 = computed; otherwise out_h output rows (top padding `pad`, bottom padding implied) */
  int32_t out_row_pitch; /* 0: output row oy of a batch item starts at C + oy * OW * ldc (dense).  Otherwise at C + oy * out_row_pitch (elements,
                          * multiple of 8): with ldc = 2 Cout and a pitch of two full rows, the four (row, column) phases of an upsampling
                          * convolution interleave into one NHWC tensor.  Plain bf16 epilogue only (no residual / second output / f32) */
  int32_t moments_accumulate; /* 1: add this launch's moments to `moments` instead of overwriting them (the phases of one tensor) */
  int32_t reserved0;          /* 0 */
  float* moments;         /* NULL, or f32 [batch][N][2]: (sum, sum of squares) of the bf16 outputs over the M pixels of each batch item, per channel
                           * QUAD: entry [c] with c % 4 == 0 holds the sums over channels c .. c+3, the other three entries are zero -- what
                           * x2i_groupnorm_nhwc_from_moments_bf16 needs when its groups are whole quads and it has no per-channel pre_add (group
                           * sums are sums of quads).  Written by the epilogue (per-row-block sums, added in a fixed order by two small kernels:
                           * deterministic), so the GroupNorm behind the conv needs no statistics pass.  Needs the whole-line bf16 epilogue (N,
                           * ldc, c_batch_stride multiples of 8, 16-byte aligned C) */
  float* moments_scratch; /* caller-owned, x2i_conv_moments_scratch_floats(M, N, batch) floats, 16-byte aligned (required with `moments`) */
} x2i_conv_desc;
int x2i_conv2d_nhwc_bf16(const x2i_gemm_args* args, const x2i_conv_desc* conv, x2i_stream_t stream);
int64_t x2i_conv_moments_scratch_floats(int32_t M, int32_t N, int32_t batch);

/* Conv2d(Cin -> Cout <= 4, k=3, stride=1, pad=1) on NHWC bf16: the VAE decoder's conv_out (diffusers Decoder.conv_out, 128 -> 3 channels at the
 * image resolution; infer/inference_qwenvl.py:213-214).  x [B][H][W][Cin], Cin in {32, 64, 96, 128}; w bf16 [Cout][3][3][Cin] (ky,kx,ci);
 * bias bf16 [Cout] or NULL; y [B][H][W][ldy] with ldy % 4 == 0: channels 0 .. 3 of every pixel are written (those behind Cout as zeros),
 * the rest of a pixel is left alone.  The output channels ride in the ROWS of the 16 x 16 x 32 MFMA, 16 neighbouring pixels in its columns
 * (csrc/conv_narrow.hip) -- as an implicit GEMM three channels would pay for a 128-column tile. */
int x2i_conv3x3_narrow_bf16(const void* x, const void* w, const void* bias, void* y, int32_t B, int32_t H, int32_t W, int32_t Cin, int32_t Cout,
                            int32_t ldy, x2i_stream_t stream);

/* Conv2d(3 -> Cout, k=3, stride=2, pad=1) on an NHWC bf16 image (lightcontrol_flux.py:594); w f32 [Cout][3][3][3]
 * (ky,kx,ci), bias f32 [Cout]; y NHWC bf16 [B][H/2][W/2][Cout], Cout % 16 == 0 and <= 64. */
int x2i_conv_stem_bf16(const void* x, const float* w, const float* bias, void* y, int32_t B, int32_t H, int32_t W,
                       int32_t Cout, x2i_stream_t stream);

/* nn.GroupNorm(G, C, eps) on NHWC bf16 [B][HW][C] with fused epilogue: y = act(GN(x + pre_add[b][c]) * w + b) + post_add
 * (ControlNeXt embedding GN+ReLU :595-602; ResnetBlock2D norm1/norm2 + SiLU with the time-embedding term added before
 * norm2; mid block conv->ReLU->GN ... + x :632-653,744).  pre_add: f32 [B][C] or NULL; post_add: bf16 like x or NULL.
 * `partial` is caller-owned scratch of x2i_groupnorm_scratch_floats(B, G) floats. */
int64_t x2i_groupnorm_scratch_floats(int32_t B, int32_t G);
int x2i_groupnorm_nhwc_bf16(const void* x, void* y, int32_t B, int64_t HW, int32_t C, int32_t G, const void* weight,
                            const void* bias, float eps, int32_t act, const float* pre_add, const void* post_add,
                            float* partial, x2i_stream_t stream);

/* The same GroupNorm with its statistics taken from cached per-channel moments: x2i_groupnorm_moments_f32 writes moments f32
 * [B][C][2] = (sum over pixels of x, of x^2) of an NHWC bf16 tensor (scratch: x2i_groupnorm_moments_scratch_floats(B, C) floats), and
 * x2i_groupnorm_nhwc_from_moments_bf16 normalises x + pre_add with the group statistics derived from them (sum (x + v) = S1 + HW v,
 * sum (x + v)^2 = S2 + 2 v S1 + HW v^2) -- no statistics pass over x.  For ResnetBlock2D's norm2 in the ControlNeXt branch: its input is
 * conv1(...) + time_emb_proj(silu(temb)) (diffusers ResnetBlock2D; lightcontrol_flux.py:620-640, forward :741), and conv1's output depends
 * on the hint only, so its moments are taken once per hint and the per-step statistics cost nothing.  `partial` as above. */
int64_t x2i_groupnorm_moments_scratch_floats(int32_t B, int32_t C);
int x2i_groupnorm_moments_f32(const void* x, int32_t B, int64_t HW, int32_t C, float* moments, float* scratch, x2i_stream_t stream);
int x2i_groupnorm_nhwc_from_moments_bf16(const void* x, void* y, int32_t B, int64_t HW, int32_t C, int32_t G, const void* weight,
                                         const void* bias, float eps, int32_t act, const float* moments, const float* pre_add,
                                         const void* post_add, float* partial, x2i_stream_t stream);

/* GROUPED affine parameters: weight / bias are [ceil(B / w_group)][C] and item b uses row b / w_group (w_group = 0: one [C] pair for all items =
 * the two entry points above).  The batch of a ControlNeXt bank is (model, sample)-major: the 19 models of a step (lightcontrol_flux.py:504-507)
 * normalise in one launch, each with its own nn.GroupNorm parameters.  Everything else as above; an item's results do not depend on the batch. */
int x2i_groupnorm_nhwc_grouped_bf16(const void* x, void* y, int32_t B, int64_t HW, int32_t C, int32_t G, const void* weight, const void* bias,
                                    int32_t w_group, float eps, int32_t act, const float* pre_add, const void* post_add, float* partial,
                                    x2i_stream_t stream);
int x2i_groupnorm_nhwc_from_moments_grouped_bf16(const void* x, void* y, int32_t B, int64_t HW, int32_t C, int32_t G, const void* weight,
                                                 const void* bias, int32_t w_group, float eps, int32_t act, const float* moments,
                                                 const float* pre_add, const void* post_add, float* partial, x2i_stream_t stream);

/* ---------------------------------------------------------------------------------------------------------
 * F.scaled_dot_product_attention(q, k, v, dropout_p=0, is_causal=False) for head_dim 128 (diffusers
 * FluxAttnProcessor2_0; reference call sites lightcontrol_flux.py:92-95,173-177).
 * Q,K: bf16 [B,H,Spad,128]; VT: bf16 [B,H,128,Spad] (V pre-transposed by x2i_qkv_split); rows/cols >= S are
 * zero padding (Spad % 128 == 0).  O: bf16 token-major, O[b][s][h*128 + d] with row stride ldo and batch
 * stride o_batch_stride (elements) -- i.e. `transpose(1,2).reshape(B,-1,H*128)` is free.
 * `scale` multiplies the scores (1/sqrt(128) for the reference's call).  scale == ln 2 declares that Q already carries
 * softmax_scale * log2(e) (x2i_qkv_desc.q_scale): the exp2-domain multiplier is then exactly 1, and launches of >= 256 workgroups
 * take the hand-scheduled kernel whose softmax has no multiply (csrc/attention_w4.hip); same result within the stated tolerance. */
int x2i_attention_bf16(const void* Q, const void* K, const void* VT, void* O, int32_t B, int32_t H, int32_t S,
                       int32_t Spad, int32_t ldo, int64_t o_batch_stride, float scale, x2i_stream_t stream);
/* The same attention for a V^T written "span-permuted" (x2i_qkv_desc.vt_perm = 1): the hand-scheduled kernel on the 16 x 16 x 32 MFMA shape
 * (csrc/attention_w16.hip), which the matrix pipe sustains at a ~11 % higher clock at this part's power cap than the 32 x 32 x 16 shape of the
 * other kernels.  x2i_attention_prefers_vt_perm(H, S, scale) says (1 / 0) whether a caller should ask its QKV producers for that layout:
 * scale == ln 2 (Q carries the scale), sequences long enough that one sample's 256-row workgroups fill half the chip -- by H and S, never by
 * the batch -- and option "attn_w16" (default 1).  A call the kernel does not serve is an error (X2I_ERR_SHAPE), never a silent fallback:
 * no other kernel reads that layout. */
int x2i_attention_vp_bf16(const void* Q, const void* K, const void* VT, void* O, int32_t B, int32_t H, int32_t S,
                          int32_t Spad, int32_t ldo, int64_t o_batch_stride, float scale, x2i_stream_t stream);
int x2i_attention_prefers_vt_perm(int32_t H, int32_t S, float scale);
/* x2i_attention_vp_bf16 with the caller's stream-K workspace (the one x2i_gemm_args.workspace names: x2i_streamk_workspace_bytes(), 256-byte
 * aligned, flags zero between launches; NULL = none): when B * H * ceil(S / 256) work items leave a partly filled last round of one-workgroup-per-CU
 * blocks (1024^2: 1.7 / 3.4 / 6.75 rounds at batch 1 / 2 / 4), that round's items are cut along the KEY axis over all CUs and the parts of an item are
 * CHAINED through the workspace -- the closing part continues from the un-normalised output, running maximum and row sums of the opening part, so
 * every row is summed in the order of an undivided item: bit-identical to x2i_attention_vp_bf16 whatever the cuts (they depend on the batch; a
 * sample's result does not).  Same co-residency assumption and give-up marker as the GEMMs' chained segments.  Option "attn_streamk" (default 1).
 * Stands behind the same call (lightcontrol_flux.py:92-95,173-177). */
int x2i_attention_vp_ws_bf16(const void* Q, const void* K, const void* VT, void* O, int32_t B, int32_t H, int32_t S,
                             int32_t Spad, int32_t ldo, int64_t o_batch_stride, float scale, void* workspace, int64_t workspace_bytes,
                             x2i_stream_t stream);

/* The same attention with an e4m3 output O8[b][s][h*128 + d] = sat(o * out_inv_scale) (ldo / o_batch_stride in bytes, multiples
 * of 8): the A operand of an fp8 projection (single blocks' proj_out in the fp8 configuration), no bf16 round trip. */
int x2i_attention_e4m3out(const void* Q, const void* K, const void* VT, void* O8, int32_t B, int32_t H, int32_t S, int32_t Spad,
                          int32_t ldo, int64_t o_batch_stride, float scale, float out_inv_scale, x2i_stream_t stream);

/* RMSNorm(q,k) + RoPE + head split + V transpose, from fused QKV rows to attention layout.
 * Stands behind FluxAttnProcessor2_0's view/transpose, norm_q/norm_k/norm_added_q/norm_added_k (RMSNorm, eps
 * 1e-6), torch.cat([txt, img], dim=2) and apply_rotary_emb.
 * Joint token s of batch b comes from qkv0 row b*S0+s if s < S0, else qkv1 row b*(S-S0)+(s-S0); a row is
 * [q(H*128) | k(H*128) | v(H*128)] with stride ld.  nq0/nk0 are the RMSNorm weights for source 0
 * (norm_added_q/k), nq1/nk1 for source 1 (norm_q/k).  cos/sin: f32 [S,128]. */
int x2i_qkv_split_bf16(const void* qkv0, const void* qkv1, int32_t ld0, int32_t ld1, int32_t B, int32_t S, int32_t S0,
                       int32_t H, const void* nq0, const void* nk0, const void* nq1, const void* nk1,
                       const float* cos, const float* sin, void* Q, void* K, void* VT, int32_t Spad, float eps,
                       x2i_stream_t stream);

/* The QKV projection with x2i_qkv_split_bf16 fused into its epilogue: C = A W^T + bias is never written; each output
 * tile (128 or 256 columns = one or two heads of the q, k or v section) goes straight from the accumulators through
 * LDS to the attention layout -- RMSNorm(q/k) * weight + RoPE into Q/K [B,H,Spad,128], V transposed into VT
 * [B,H,128,Spad] (same operators and reference lines as x2i_qkv_split_bf16; the values normalised are the bf16-rounded
 * linear outputs, as in the reference's bf16 run).  args: M/N/K/batch/A/W/bias as for x2i_gemm_bf16 with N == 3*H*128,
 * act == 0, no residual / gate / C2 / f32 output; args->C is ignored.  Row m of batch item z of this GEMM is joint token
 * s = tok_off + m % rows_per_sample of sample b = z + m / rows_per_sample. */
typedef struct x2i_qkv_desc {
  const void* norm_q;  /* bf16 [128] RMSNorm weights for the q / k heads of these rows */
  const void* norm_k;
  const float* cos;    /* f32 [S,128] interleaved-pair RoPE tables over joint positions (cos[s][2k] == cos[s][2k+1]: FluxPosEmbed's repeat_interleave) */
  const float* sin;    /* ... or NULL: `cos` then points at the PAIR-form table f32 [S,64,2] = (cos, sin) of dim pair k -- the same values in
                        * half the bytes; the persistent kernel's q / k epilogue fetches it two half chunks ahead (same results, bit for bit) */
  void* Q;             /* bf16 [B,H,Spad,128] */
  void* K;
  void* VT;            /* bf16 [B,H,128,Spad] */
  int32_t H, Spad, tok_off, rows_per_sample;
  float eps;
  float q_scale;       /* 0 or 1: none.  Otherwise Q is written as q * q_scale, the factor applied in f32 in front of the one bf16
                        * rounding: a caller that passes softmax_scale * log2(e) here and scale = ln 2 to x2i_attention_* gets the same
                        * attention with the score multiply gone from the kernels' inner loops (and none of the double rounding a
                        * separate rescale of the bf16 Q would cost) */
  int32_t vt_perm;     /* 0: V^T rows hold the tokens in sequence order.  1: "span-permuted" -- within every 32-token span position kk holds token
                        * 16 ((kk >> 2) & 1) + 4 (kk >> 3) + (kk & 3): the order x2i_attention_vp_bf16 (the 16 x 16 x 32 MFMA attention kernel)
                        * reads.  Every producer of one attention call's V^T must use the same value (x2i_attention_prefers_vt_perm) */
} x2i_qkv_desc;
int x2i_gemm_qkv_bf16(const x2i_gemm_args* args, const x2i_qkv_desc* qkv, x2i_stream_t stream);
/* Two GEMMs of the same kind -- the image-stream and the text-stream linear of a double-stream block (lightcontrol_flux.py:173-200:
 * to_q|k|v / add_q|k|v_proj, to_out[0] / to_add_out, ff / ff_context): same layer type, different weights, 8 : 1 in rows -- as ONE
 * launch of the persistent kernel: problem 1's output tiles ride in the rounds of problem 0 instead of under-filling a launch of
 * their own.  Every output tile is computed exactly as by x2i_gemm_bf16 / x2i_gemm_qkv_bf16 (bit-identical).  Grouped when both
 * problems have the same K, activation and residual kind and are served by the persistent kernel; otherwise (and with option
 * "gemm_pair" = 0) the two launches are issued one after the other, so the call is always valid where the two calls are. */
int x2i_gemm_pair_bf16(const x2i_gemm_args* args0, const x2i_gemm_args* args1, x2i_stream_t stream);
int x2i_gemm_qkv_pair_bf16(const x2i_gemm_args* args0, const x2i_qkv_desc* qkv0, const x2i_gemm_args* args1, const x2i_qkv_desc* qkv1,
                           x2i_stream_t stream);
/* The same fused projection on e4m3 operands (x2i_fp8_desc as for x2i_gemm_fp8, out_fp8 = 0): A is the e4m3 LayerNorm output with
 * its row scales, W the quantised to_q|to_k|to_v weight.  The accumulators are dequantised before the shared RMSNorm / RoPE
 * epilogue, so Q / K / V^T are bf16 as above.  Needs H*128 % 256 == 0 and the x2i_gemm_fp8 alignment rules. */
int x2i_gemm_qkv_fp8(const x2i_gemm_args* args, const x2i_fp8_desc* fp8, const x2i_qkv_desc* qkv, x2i_stream_t stream);

/* LayerNorm(elementwise_affine=False, eps) * (1 + scale[b]) + shift[b]   (AdaLayerNormZero / ZeroSingle /
 * Continuous and `norm2(x) * (1 + scale_mlp[:, None]) + shift_mlp[:, None]`, lightcontrol_flux.py:166-170,
 * 183-184,196-197,89,542).  X,Y: bf16 [B][S][D] with row strides ldx/ldy and batch strides (elements).  Rows
 * s < S0 use (shift0, scale0), the others (shift1, scale1); each is f32 [B][D] with batch stride mod_bs. */
int x2i_ln_modulate_bf16(const void* X, int64_t x_bs, int32_t ldx, void* Y, int64_t y_bs, int32_t ldy, int32_t B,
                         int32_t S, int32_t D, int32_t S0, const float* shift0, const float* scale0,
                         const float* shift1, const float* scale1, int64_t mod_bs, float eps, x2i_stream_t stream);

/* nn.LayerNorm(D, eps) with affine weight/bias (utils/proj.py:17,29; model_internvl/proj.py norm0/norm1). */
int x2i_ln_affine_bf16(const void* X, void* Y, int64_t rows, int32_t D, const void* weight, const void* bias,
                       float eps, x2i_stream_t stream);

/* Y[b][n] (+)= act_out( bias[n] + sum_k W[n][k] * act_in(X[b][k]) ), M = B <= 64 rows: the HBM-bound
 * "skinny" linears -- AdaLayerNorm* modulation (SiLU in), time/text/guidance embedders (diffusers
 * CombinedTimestep*Embeddings; lightcontrol_flux.py:249-254,452-456), ControlNeXt time embedding.
 * X: f32 or bf16 [B,K]; W bf16 [N,K]; bias bf16 or NULL; Y f32 [B,N] (row stride ldy). */
int x2i_skinny_linear(const void* X, int32_t x_is_bf16, const void* W, const void* bias, float* Y, int32_t ldy,
                      int32_t B, int32_t N, int32_t K, int32_t act_in, int32_t act_out, int32_t accumulate,
                      x2i_stream_t stream);
/* `groups` independent skinny linears in one launch (the time embeddings of a ControlNeXt bank): group g has W[g] [N][K] and bias[g] [N]
 * (contiguous), reads X + g * x_group_stride elements (0: one X [B][K] for every group) and writes rows g*B .. g*B + B-1 of Y.  B <= 8. */
int x2i_skinny_linear_grouped(const void* X, int32_t x_is_bf16, int64_t x_group_stride, const void* W, const void* bias, float* Y, int32_t ldy,
                              int32_t groups, int32_t B, int32_t N, int32_t K, int32_t act_in, int32_t act_out, int32_t accumulate,
                              x2i_stream_t stream);

/* diffusers Timesteps(dim, flip_sin_to_cos=True, downscale_freq_shift=0): out[b] = [cos(t f_k) | sin(t f_k)] */
int x2i_timestep_sinusoid(const float* t, float* out, int32_t B, int32_t dim, int32_t round_bf16, x2i_stream_t stream);

/* FluxPosEmbed (diffusers 0.31; used at lightcontrol_flux.py:247,472): rotary tables for position
 * ids f32 [S,3] over three axes of (even) widths d0,d1,d2: column c of axis a holds cos / sin of ids[s][a] *
 * theta^(-2k/d_a), k = pair index, each value repeated for the two elements of its pair.  Frequencies and angles are
 * evaluated in float64 like the reference; outputs f32 [S, d0+d1+d2].  Step-invariant: called once per prompt. */
int x2i_rope_table_f32(const float* ids, int32_t S, int32_t d0, int32_t d1, int32_t d2, float theta, float* cos, float* sin,
                       x2i_stream_t stream);

/* X[b][s][:] = bf16(X + gate[b][:] * T[b][s][:]): `hidden_states + gate.unsqueeze(1) * attn_output` (lightcontrol_flux.py:180-181,
 * 193-194) as a separate pass.  The sampling path never needs it (the gate / residual ride in the projection's epilogue); it exists
 * for the attention-distillation capture (train/train_qwenvl.py:206-214: forward hooks on every block's `attn`), where the
 * projected attention outputs have to exist as tensors of their own.  Strides in elements; gate f32 [B][D] with batch stride gate_bs. */
int x2i_gated_residual_bf16(void* X, int64_t x_bs, int32_t ldx, const void* T, int64_t t_bs, int32_t ldt, const float* gate,
                            int64_t gate_bs, int32_t B, int32_t S, int32_t D, x2i_stream_t stream);

/* FlowMatchEulerDiscreteScheduler.step: x = bf16(f32(x) + dt[0] * f32(eps)); dt is a DEVICE scalar so the
 * call can be captured in a hipGraph. */
int x2i_euler_step_bf16(void* x, const void* eps, int64_t n, const float* dt, x2i_stream_t stream);

/* Projector layer fusion (utils/proj.py:62-72).  x: bf16 [B,C,S,H] -> y: bf16 [B,S,H]
 *   conv5x5:   Conv2d(C->1, k=5, pad=2) over the (S,H) plane (:68-69); w f32 [C,5,5], bias f32 [1]
 *              The f32 taps are rounded to bf16 inside the kernel (packed bf16 tap pairs feed v_dot2c_f32_bf16; the reference's
 *              conv runs in bf16 under autocast, so its weights are bf16 values already); accumulation and bias are fp32.
 *   layer_mean: (cha_scale * x).mean(1) (:66-67) or plain mean (:70-71) when scale == NULL; scale f32 [C] */
int x2i_proj_conv5x5_bf16(const void* x, const float* w, const float* bias, void* y, int32_t B, int32_t C, int32_t S,
                          int32_t H, x2i_stream_t stream);
int x2i_proj_layer_mean_bf16(const void* x, const float* scale, void* y, int32_t B, int32_t C, int64_t plane,
                             x2i_stream_t stream);
/* conv5x5 on the matrix cores (the form x2i_amd uses; same result up to fp32 summation order): the taps are first expanded, once
 * per weight, into banded-Toeplitz MFMA fragments -- table: bf16 [C][5][64][8] (C * 5 KiB, 16-byte aligned, caller-owned) -- and the
 * convolution then runs as v_mfma_f32_16x16x32_bf16 over 32-column windows of 16 input rows (csrc/proj_conv_mfma.hip).  H % 8 == 0,
 * y 8-byte aligned; any C. */
int x2i_proj_conv5x5_pack(const float* w, void* table, int32_t C, x2i_stream_t stream);
int x2i_proj_conv5x5_packed_bf16(const void* x, const void* table, const float* bias, void* y, int32_t B, int32_t C, int32_t S,
                                 int32_t H, x2i_stream_t stream);
/* torch.mean(x1, 1): x f32 [B,S,N] -> y f32 [B,N] (utils/proj.py:32) */
int x2i_seq_mean_f32(const float* x, float* y, int32_t B, int32_t S, int32_t N, x2i_stream_t stream);

/* In-place row softmax, bf16 [rows][cols]: x <- softmax(scale * x) with fp32 statistics (the single-head mid-block attention
 * of the VAE decoder, diffusers Attention/AttnProcessor2_0 on [B, HW, 512]). cols % 8 == 0. */
int x2i_softmax_rows_bf16(void* x, int64_t rows, int32_t cols, float scale, x2i_stream_t stream);

/* dtype casts used at the boundary */
int x2i_cast_f32_to_bf16(const float* x, void* y, int64_t n, x2i_stream_t stream);
int x2i_cast_bf16_to_f32(const void* x, float* y, int64_t n, x2i_stream_t stream);

/* ---------------------------------------------------------------------------------------------------------
 * N4 -- backward pass of the attention-distillation step (train/train_qwenvl.py:556-654: `loss.backward()` with only the
 * projector trainable).  The transformer is frozen, so its backward is an ACTIVATION-gradient chain: every matrix product is
 * x2i_gemm_bf16 on a transposed operand (x2i_transpose_bf16), and the entry points below are the row kernels around them
 * (csrc/train.hip).  Column sums (d scale / d shift / d gate / d weight / d bias) are two-stage: a kernel writes per-wave partial
 * rows into a caller-provided f32 scratch, x2i_reduce_rows_f32 finishes -- no atomics, results independent of scheduling.
 * All tensors bf16 unless typed float; rows must be 16-byte aligned, column counts multiples of 8 (<= 4096 for the row kernels). */
/* out[z][c][r] = in[z][r][c] */
int x2i_transpose_bf16(const void* in, int64_t in_batch_stride, int64_t ld_in, void* out, int64_t out_batch_stride, int64_t ld_out,
                       int32_t batch, int32_t R, int32_t C, x2i_stream_t stream);
/* x [nz][Rt][ld]: P = softmax(scale * x) over the Cv valid columns of the Rv valid rows, in place; padding up to (Rt, Ct) becomes 0 */
int x2i_softmax_pad_bf16(void* x, int64_t ld, int32_t nz, int32_t Rt, int32_t Rv, int32_t Ct, int32_t Cv, float scale, x2i_stream_t stream);
/* dP <- scale * P * (dP - rowsum(P * dP)), same padding rules */
int x2i_softmax_bwd_bf16(const void* P, void* dP, int64_t ld, int32_t nz, int32_t Rt, int32_t Rv, int32_t Ct, int32_t Cv, float scale,
                         x2i_stream_t stream);
/* backward of y = LayerNorm_noaffine(x) * mult + shift over S rows per sample: dXout = (dXin ? dXin : 0) + dx;
 * mult_is_scale: mult = 1 + mult_ptr[b][col] (AdaLN scale, f32, batch stride mult_bs) else mult = mult_ptr[col] (affine weight as f32).
 * partial: f32 [B][ceil(S / rows_per_wave)][2][D]: [0] = sum dy * xhat (-> d scale / d weight), [1] = sum dy (-> d shift / d bias) */
int x2i_ln_mod_bwd_bf16(const void* X, int64_t x_bs, int32_t ldx, const void* dY, int64_t dy_bs, int32_t ldy, const float* mult, int64_t mult_bs,
                        int32_t mult_is_scale, const void* dXin, void* dXout, int64_t dx_bs, int32_t lddx, int32_t B, int32_t S, int32_t D,
                        int32_t rows_per_wave, float* partial, float eps, x2i_stream_t stream);
/* backward of x + gate[b][:] * t: dT = gate * dX (+ G, a gradient injected at t; may be NULL); partial f32 [B][ceil(S/R)][D] = sum dX * T
 * (-> d gate).  gate == NULL: dT = dX (+ G), no partials. */
int x2i_gate_bwd_bf16(const void* dX, int64_t dx_bs, int32_t lddx, const void* T, int64_t t_bs, int32_t ldt, const float* gate, int64_t gate_bs,
                      const void* G, int64_t g_bs, int32_t ldg, void* dT, int64_t dt_bs, int32_t lddt, int32_t B, int32_t S, int32_t D,
                      int32_t rows_per_wave, float* partial, x2i_stream_t stream);
/* out[z][i] = (accumulate ? out[z][i] : 0) + alpha * sum_{p < np} in[z * in_z_stride + p * in_p_stride + i] */
int x2i_reduce_rows_f32(const float* in, int64_t in_z_stride, int32_t np, int64_t in_p_stride, float* out, int64_t out_z_stride, int32_t nz,
                        int32_t len, int32_t accumulate, float alpha, x2i_stream_t stream);
/* dA <- dA * act'(pre) (X2I_ACT_GELU_TANH / GELU_ERF / SILU); bf16 matrices with row strides, or contiguous f32 (is_f32) */
int x2i_act_bwd(void* dA, int64_t ldd, const void* pre, int64_t ldp, int64_t rows, int32_t cols, int32_t act, int32_t is_f32, x2i_stream_t stream);
/* backward of x2i_qkv_split_bf16: dQ / dK / dV bf16 [B,H,Spad,128] (dV row-major per head) -> d(q|k|v) rows, from the saved pre-norm rows */
int x2i_qkv_split_bwd_bf16(const void* qkv0, const void* qkv1, int32_t ld0, int32_t ld1, void* d0, void* d1, int32_t ldd0, int32_t ldd1, int32_t B,
                           int32_t S, int32_t S0, int32_t H, const void* nq0, const void* nk0, const void* nq1, const void* nk1, const float* cosp,
                           const float* sinp, const void* dQ, const void* dK, const void* dV, int32_t Spad, float eps, x2i_stream_t stream);
/* dx[b][:] = sum_n dy[b][n] * W[n][:] (B <= 8): partial f32 [ceil(N / chunk)][B][K], finished by x2i_reduce_rows_f32 */
int x2i_skinny_linear_bwd(const float* dy, int64_t dy_bs, const void* W, int32_t ldw, float* partial, int32_t B, int32_t N, int32_t K,
                          int32_t chunk, x2i_stream_t stream);
/* distillation loss rows (:58-61, :613-634): row_loss[r] = KL(softmax(normalize(student_r)/T) vs softmax(normalize(teacher_r)/T)) as
 * F.kl_div(q.log(), p) sums it; grad (may be NULL) = loss_scale * d row_loss / d student */
int x2i_kd_loss_bf16(const void* teacher, int64_t ldt, const void* student, int64_t lds, void* grad, int64_t ldg, float* row_loss, int64_t rows,
                     int32_t D, float temperature, float loss_scale, x2i_stream_t stream);
/* g[0..n) <- 0 when *term is NaN / Inf (the reference skips non-finite per-block loss terms, :617-620); no host synchronisation */
int x2i_zero_if_nonfinite_bf16(void* g, int64_t n, const float* term, x2i_stream_t stream);

/* Fused (flash-style) attention backward, head_dim 128 (csrc/attention_bwd.hip): dQ, dK, dV bf16 [B,H,Spad,128] from
 *   Q, K, V, dO  bf16 [B,H,Spad,128] (row-major per head, zero beyond S)      QT, KT, dOT  bf16 [B,H,128,Spad] (their transposes)
 *   D  f32 [B,H,Spad] = rowsum(dO * O) (x2i_attention_bwd_prep_bf16 from the token-major dO / O)     lse2  f32 [B,H,Spad] scratch
 * Three launches: log2-sum-exp statistics into lse2 (skipped when have_lse != 0: lse2 then is an input), dQ (persistent query blocks),
 * dK / dV (persistent key blocks); no atomics. */
int x2i_attention_bwd_bf16(const void* Q, const void* K, const void* V, const void* QT, const void* KT, const void* dO, const void* dOT, float* lse2,
                           const float* D, void* dQ, void* dK, void* dV, int32_t B, int32_t H, int32_t S, int32_t Spad, float scale,
                           int32_t have_lse, x2i_stream_t stream);
/* x2i_attention_bf16 that also writes lse2 f32 [B,H,Spad] = log2 sum_j exp(scale * s_qj) (+big on the padding rows): handed to
 * x2i_attention_bwd_bf16 with have_lse = 1 it saves the statistics pass of the backward. */
int x2i_attention_lse_bf16(const void* Q, const void* K, const void* VT, void* O, float* lse2, int32_t B, int32_t H, int32_t S, int32_t Spad,
                           int32_t ldo, int64_t o_batch_stride, float scale, x2i_stream_t stream);
int x2i_attention_bwd_prep_bf16(const void* dO, int64_t do_bs, int32_t lddo, const void* O, int64_t o_bs, int32_t ldo, float* D, int32_t B, int32_t H,
                                int32_t S, int32_t Spad, x2i_stream_t stream);
/* -- the trainable side (projector): weight gradients of the layer fusion, gradient clipping, AdamW.  Linear-layer weight gradients are
 * x2i_gemm_bf16 launches on transposed operands (dW = dY^T X). */
/* conv5x5 (utils/proj.py:50,68-69) weight gradient: partial f32 [C][B][ceil(S/16)][25], dw[c] = sum over [B][chunk] (x2i_reduce_rows_f32) */
int x2i_proj_conv5x5_wgrad(const void* x, const void* dy, float* partial, int32_t B, int32_t C, int32_t S, int32_t H, x2i_stream_t stream);
/* partial f32 [C][B][nchunk] = sum_i dy[b][i] * x[b][c][i] over chunks of the plane (cha_scale gradient, utils/proj.py:66-67) */
int x2i_plane_dot_bf16(const void* x, const void* dy, float* partial, int32_t B, int32_t C, int64_t plane, int32_t nchunk, x2i_stream_t stream);
/* partial[blk] = sum x (squares = 0) or sum x^2 (squares = 1) over a grid-strided slice of n elements (f32 or bf16) */
int x2i_sum_partials(const void* x, int32_t is_bf16, int64_t n, int32_t squares, float* partial, int32_t nblocks, x2i_stream_t stream);
/* out[0] = min(1, max_norm / (sqrt(*sumsq) + 1e-6)) (torch.nn.utils.clip_grad_norm_, train/train_qwenvl.py:628), out[1] = the norm */
int x2i_clip_coef_f32(const float* sumsq, float max_norm, float* out, x2i_stream_t stream);
/* AdamW step (torch.optim.AdamW semantics, :447-459) on bf16 parameters, f32 gradients (scaled by *grad_coef when non-NULL) and f32 moments */
int x2i_adamw_bf16(void* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2, float eps, float weight_decay,
                   float bias_correction1, float bias_correction2, const float* grad_coef, x2i_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* X2I_H */
