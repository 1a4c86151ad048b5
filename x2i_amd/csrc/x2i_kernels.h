// Internal launcher declarations (host side). Each launcher enqueues on `stream` and returns an X2I_* code.
#pragma once
#include <hip/hip_runtime.h>
#include "../../include/x2i.h"

int x2i_launch_gemm(const x2i_gemm_args* a, hipStream_t stream);
int x2i_launch_gemm_conv(const x2i_gemm_args* a, const x2i_conv_desc* cd, hipStream_t stream);
int x2i_launch_gemm_qkv(const x2i_gemm_args* a, const x2i_qkv_desc* qd, hipStream_t stream);
int x2i_launch_gemm_pair(const x2i_gemm_args* a0, const x2i_qkv_desc* q0, const x2i_gemm_args* a1, const x2i_qkv_desc* q1, hipStream_t stream);
int x2i_launch_gemm_fp8(const x2i_gemm_args* a, const x2i_fp8_desc* f, hipStream_t stream);
int x2i_launch_gemm_qkv_fp8(const x2i_gemm_args* a, const x2i_fp8_desc* f, const x2i_qkv_desc* qd, hipStream_t stream);
int x2i_launch_quantize_rows_fp8(const void* x, long long rows, int cols, long long ldx, void* y, long long ldy, float* scale,
                                 float static_inv_scale, hipStream_t stream);
int x2i_launch_ln_modulate_fp8(const void* X, long long x_bs, int ldx, void* Y, long long y_bs, int ldy, void* Y8, long long y8_bs,
                               int ldy8, float* row_scale, int B, int S, int D, int S0, const float* shift0, const float* scale0,
                               const float* shift1, const float* scale1, long long mod_bs, float eps, hipStream_t stream);
int x2i_launch_conv3x3_narrow(const void* x, const void* w, const void* bias, void* y, int B, int H, int W, int Cin, int Cout, int ldy,
                              hipStream_t stream);   // conv_narrow.hip
int x2i_launch_conv_stem(const void* x, const float* w, const float* bias, void* y, int B, int H, int W, int Cout,
                         hipStream_t stream);
long long x2i_groupnorm_scratch(int B, int G);
int x2i_launch_groupnorm(const void* x, void* y, int B, long long HW, int C, int G, const void* w, const void* b, float eps,
                         int act, const float* pre_add, const void* post_add, float* partial, hipStream_t stream, int w_group = 0);
long long x2i_groupnorm_moments_scratch(int B, int C);
int x2i_launch_groupnorm_moments(const void* x, int B, long long HW, int C, float* moments, float* scratch, hipStream_t stream);
int x2i_launch_groupnorm_from_moments(const void* x, void* y, int B, long long HW, int C, int G, const void* w, const void* b, float eps, int act,
                                      const float* moments, const float* pre_add, const void* post_add, float* partial, hipStream_t stream,
                                      int w_group = 0);
int x2i_launch_attention(const void* Q, const void* K, const void* VT, void* O, int B, int H, int S, int Spad, int ldo,
                         long long o_bs, float scale, hipStream_t stream, int out8 = 0, float oinv = 1.f, float* lse = nullptr);
int x2i_launch_attention_w16(const void* Q, const void* K, const void* VT, void* O, int B, int H, int S, int Spad, int ldo, long long o_bs,
                             float scale_log2, int prescale, hipStream_t stream, float* lse, void* workspace = nullptr, long long workspace_bytes = 0);   // hand-scheduled, 16x16x32 (attention_w16.hip); workspace: the caller's stream-K workspace (or none)
int x2i_launch_attention_16(const void* Q, const void* K, const void* VT, void* O, int B, int H, int S, int Spad, int ldo, long long o_bs,
                            float scale_log2, hipStream_t stream, float* lse, int vperm);   // A/B: 16x16x32 MFMA shape (attention16.hip, attn_variant = 10 / 11)
int x2i_launch_attention_pp(const void* Q, const void* K, const void* VT, void* O, int B, int H, int S, int Spad, int ldo,
                            long long o_bs, float scale_log2, hipStream_t stream, int out8, float oinv, int thr, float* lse);
// hand-scheduled one-wave-per-SIMD form (attention_w4.hip, generated K-tile loop); X2I_ERR_STATE = not served (alignment)
int x2i_launch_attention_w4(const void* Q, const void* K, const void* VT, void* O, int B, int H, int S, int Spad, int ldo, long long o_bs,
                            float scale_log2, int prescale, hipStream_t stream, float* lse, int out8, float oinv);
int x2i_launch_qkv_split(const void* qkv0, const void* qkv1, int ld0, int ld1, int B, int S, int S0, int H,
                         const void* nq0, const void* nk0, const void* nq1, const void* nk1, const float* cosp,
                         const float* sinp, void* Q, void* K, void* VT, int Spad, float eps, hipStream_t stream);
int x2i_launch_ln_modulate(const void* X, long long x_bs, int ldx, void* Y, long long y_bs, int ldy, int B, int S, int D,
                           int S0, const float* shift0, const float* scale0, const float* shift1, const float* scale1,
                           long long mod_bs, float eps, hipStream_t stream);
int x2i_launch_ln_affine(const void* X, void* Y, long long rows, int D, const void* w, const void* b, float eps,
                         hipStream_t stream);
int x2i_launch_skinny_linear(const void* X, int x_is_bf16, const void* W, const void* bias, float* Y, int ldy, int B,
                             int N, int K, int act_in, int act_out, int accumulate, hipStream_t stream);
int x2i_launch_skinny_linear_grouped(const void* X, int x_is_bf16, long long x_gs, const void* W, const void* bias, float* Y, int ldy, int G, int B,
                                     int N, int K, int act_in, int act_out, int accumulate, hipStream_t stream);
int x2i_launch_timestep_sinusoid(const float* t, float* out, int B, int dim, int round_bf16, hipStream_t stream);
int x2i_launch_rope_table(const float* ids, int S, int d0, int d1, int d2, float theta, float* cosp, float* sinp, hipStream_t stream);
int x2i_launch_gated_residual(void* X, long long x_bs, int ldx, const void* T, long long t_bs, int ldt, const float* gate, long long g_bs,
                              int B, int S, int D, hipStream_t stream);
int x2i_launch_euler_step(void* x, const void* eps, long long n, const float* dt, hipStream_t stream);
int x2i_launch_proj_conv5x5(const void* x, const float* w, const float* bias, void* y, int B, int C, int S, int H,
                            hipStream_t stream);
int x2i_launch_proj_conv5x5_pack(const float* w, void* table, int C, hipStream_t stream);
int x2i_launch_proj_conv5x5_packed(const void* x, const void* table, const float* bias, void* y, int B, int C, int S, int H,
                                   hipStream_t stream);
int x2i_launch_layer_mean(const void* x, const float* scale, void* y, int B, int C, long long plane, hipStream_t stream);
int x2i_launch_seq_mean(const float* x, float* y, int B, int S, int N, hipStream_t stream);
int x2i_launch_softmax_rows(void* x, long long rows, int cols, float scale, hipStream_t stream);
int x2i_launch_cast_f32_bf16(const float* x, void* y, long long n, hipStream_t stream);
int x2i_launch_cast_bf16_f32(const void* x, float* y, long long n, hipStream_t stream);

/* ---- backward kernels of the attention-distillation step (train.hip) */
int x2i_launch_transpose(const void* in, long long in_bs, long long ld_in, void* out, long long out_bs, long long ld_out, int batch, int R, int C,
                         hipStream_t stream);
int x2i_launch_softmax_pad(void* x, long long ld, int nz, int Rt, int Rv, int Ct, int Cv, float scale, hipStream_t stream);
int x2i_launch_softmax_bwd(const void* P, void* dP, long long ld, int nz, int Rt, int Rv, int Ct, int Cv, float scale, hipStream_t stream);
int x2i_launch_ln_mod_bwd(const void* X, long long x_bs, int ldx, const void* dY, long long dy_bs, int ldy, const float* m, long long m_bs,
                          int mult_is_scale, const void* dXin, void* dXout, long long dx_bs, int lddx, int B, int S, int D, int R,
                          float* partial, float eps, hipStream_t stream);
int x2i_launch_gate_bwd(const void* dX, long long dx_bs, int lddx, const void* T, long long t_bs, int ldt, const float* gate, long long g_bs,
                        const void* G, long long gg_bs, int ldg, void* dT, long long dt_bs, int lddt, int B, int S, int D, int R,
                        float* partial, hipStream_t stream);
int x2i_launch_reduce_rows(const float* in, long long in_zs, int np, long long in_ps, float* out, long long out_zs, int nz, int len,
                           int accumulate, float alpha, hipStream_t stream);
int x2i_launch_act_bwd(void* dA, long long ldd, const void* pre, long long ldp, long long rows, int cols, int act, int is_f32,
                       hipStream_t stream);
int x2i_launch_qkv_split_bwd(const void* qkv0, const void* qkv1, int ld0, int ld1, void* d0, void* d1, int ldd0, int ldd1, int B, int S, int S0,
                             int H, const void* nq0, const void* nk0, const void* nq1, const void* nk1, const float* cosp, const float* sinp,
                             const void* dQ, const void* dK, const void* dV, int Spad, float eps, hipStream_t stream);
int x2i_launch_skinny_bwd(const float* dy, long long dy_bs, const void* W, int ldw, float* partial, int B, int N, int K, int chunk,
                          hipStream_t stream);
int x2i_launch_kd_loss(const void* teacher, long long ldt, const void* student, long long lds, void* grad, long long ldg, float* row_loss,
                       long long rows, int D, float temperature, float loss_scale, hipStream_t stream);
int x2i_launch_zero_if_nonfinite(void* g, long long n, const float* term, hipStream_t stream);
int x2i_launch_conv5x5_wgrad(const void* x, const void* dy, float* partial, int B, int C, int S, int H, hipStream_t stream);
int x2i_launch_plane_dot(const void* x, const void* dy, float* partial, int B, int C, long long plane, int nchunk, hipStream_t stream);
int x2i_launch_sum(const void* x, int is_bf16, long long n, int mode, float* partial, int nblocks, hipStream_t stream);
int x2i_launch_clip_coef(const float* sumsq, float max_norm, float* out, hipStream_t stream);
int x2i_launch_adamw(void* p, const float* g, float* m, float* v, long long n, float lr, float b1, float b2, float eps, float wd, float bc1,
                     float bc2, const float* coef, hipStream_t stream);
int x2i_launch_attention_bwd(const void* Q, const void* K, const void* V, const void* QT, const void* KT, const void* dOh, const void* dOT,
                             float* L2, const float* Dv, void* dQ, void* dK, void* dV, int B, int H, int S, int Spad, float scale,
                             int have_lse, hipStream_t stream);
int x2i_launch_attention_bwd_prep(const void* dO, long long do_bs, int lddo, const void* O, long long o_bs, int ldo, float* Dv, int B, int H,
                                  int S, int Spad, hipStream_t stream);
