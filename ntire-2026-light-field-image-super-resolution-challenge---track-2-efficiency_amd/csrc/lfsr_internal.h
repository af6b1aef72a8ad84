// Internal (non-ABI) entry points shared between the translation units of liblfsr_hip.so.
#pragma once
#include "lfsr_common.h"

// gather modes (values match gemm_gather_kernel.h)
enum { LFSR_IN_SAME = 0, LFSR_IN_CONV3 = 1, LFSR_IN_ANG = 2, LFSR_IN_EPIH = 3, LFSR_IN_EPIV = 4,
       LFSR_IN_CHK_H = 5, LFSR_IN_CHK_V = 6, LFSR_IN_LINE_H = 7, LFSR_IN_LINE_V = 8 };
enum { LFSR_OUT_SAME = 0, LFSR_OUT_VIEWS = 1, LFSR_OUT_EPIH = 2, LFSR_OUT_EPIV = 3 };

#if defined(__HIPCC__)
// The exact three-term bf16 split of one PAIR of fp32 values (x = x0 + x1 + x2 by truncation), planes in MFMA operand order (element 0 in the low half).
//   p0 = (top 16 bits of a1, top 16 bits of a0), r = a - trunc(a) (exact), p1 likewise from r, p2 from q = r - trunc(r).
// LFSR_SPLIT_DOT2 (default 1): the residuals as ONE v_dot2c_f32_bf16 each on the plane just packed -- r0 = a0 + (-1.0) * p0.lo + 0 * p0.hi, r1 = a1 + 0 * p0.lo
// + (-1.0) * p0.hi: every partial sum is exactly representable, so the instruction's internal order and rounding do not matter (tools/lab/dot2_split.hip: bit-equal
// to the and / sub form on 2^26 random bit patterns over every exponent, denormals included) -- 3.5 VALU per element instead of 5.5.  The builtin (not asm) so that
// the compiler pads the DOT-write -> VALU-read hazard (3 wait states, which inline asm does not get: found as spurious mismatches in the first lab run).
// The selector constants go through an SGPR the compiler cannot see into: given the literal 0x0000BF80 it encodes the INLINE constant -1.0, which the hardware reads
// as the f32 pattern 0xBF800000 for this operand -- the other half (lab: r0 came out as a0 - trunc(a1)).
#ifndef LFSR_SPLIT_DOT2
#define LFSR_SPLIT_DOT2 1
#endif
__device__ __forceinline__ void lfsr_split_pair(float a0, float a1, unsigned& p0, unsigned& p1, unsigned& p2) {
  typedef __bf16 lfsr_bf16x2 __attribute__((ext_vector_type(2)));
  p0 = __builtin_amdgcn_perm(__float_as_uint(a1), __float_as_uint(a0), 0x07060302u);
#if LFSR_SPLIT_DOT2
  unsigned klo, khi;
  asm("s_mov_b32 %0, 0xbf80" : "=s"(klo));
  asm("s_mov_b32 %0, 0xbf800000" : "=s"(khi));
  const float r0 = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(lfsr_bf16x2, p0), __builtin_bit_cast(lfsr_bf16x2, klo), a0, false);
  const float r1 = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(lfsr_bf16x2, p0), __builtin_bit_cast(lfsr_bf16x2, khi), a1, false);
  p1 = __builtin_amdgcn_perm(__float_as_uint(r1), __float_as_uint(r0), 0x07060302u);
  const float q0 = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(lfsr_bf16x2, p1), __builtin_bit_cast(lfsr_bf16x2, klo), r0, false);
  const float q1 = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(lfsr_bf16x2, p1), __builtin_bit_cast(lfsr_bf16x2, khi), r1, false);
#else
  const float r0 = a0 - __uint_as_float(__float_as_uint(a0) & 0xffff0000u), r1 = a1 - __uint_as_float(__float_as_uint(a1) & 0xffff0000u);
  p1 = __builtin_amdgcn_perm(__float_as_uint(r1), __float_as_uint(r0), 0x07060302u);
  const float q0 = r0 - __uint_as_float(__float_as_uint(r0) & 0xffff0000u), q1 = r1 - __uint_as_float(__float_as_uint(r1) & 0xffff0000u);
#endif
  p2 = __builtin_amdgcn_perm(__float_as_uint(q1), __float_as_uint(q0), 0x07060302u);
}
#endif

#ifdef LFSR_CONV_DIAG
// diagnostic builds only (tools/build_diag.sh): buffer that receives the conv kernels' in-kernel s_memtime stamps; its own argument, never an
// operand slot (round 1 passed it as R2, which selected the two-residual kernel variant on a null-based descriptor: DESIGN.md, incident note)
extern float* g_lfsr_diag_buf;
extern "C" int lfsr_diag_set_buffer(float* buf);
#endif

// op_profile.cpp: operator-level timing hook (RAII).  Off (default): one relaxed atomic load.  On (lfsr_op_profile(1)): a hipEvent pair on `st` around
// the scope, aggregated per (op, a, b) by lfsr_op_profile_read.  `op` must be a string literal.
struct LfsrOpTimer {
  LfsrOpTimer(const char* op, int a, int b, hipStream_t st);
  ~LfsrOpTimer();
  LfsrOpTimer(const LfsrOpTimer&) = delete;
  LfsrOpTimer& operator=(const LfsrOpTimer&) = delete;
 private:
  int slot_, gen_;
  hipStream_t st_;
};

// conv3x3_halo.hip
int lfsr_conv3x3_halo_launch(const float* x, int x_stride, int x_choff, const float* w_packed, float* y, int y_stride, int y_choff,
                             const float* r1, int r1_stride, int r1_choff, const float* r2, int r2_stride, int r2_choff,
                             const float* mk, int mk_stride, int mk_choff, float mk_slope,
                             int n_img, int h, int w, float slope, hipStream_t st);
int lfsr_conv3x3_halo_tail_launch(const float* x, int x_stride, int x_choff, const float* w_packed, float* y, int y_stride, int y_choff,
                                  const float* r1, int r1_stride, int r1_choff, const float* r2, int r2_stride, int r2_choff,
                                  const float* mk, int mk_stride, int mk_choff, float mk_slope,
                                  int n_img, int h, int w, float slope, int tile_begin, int tile_count, hipStream_t st);
// conv3x3_wino.hip: Winograd F(2x2,3x3) form of the same op.  A packed 3x3 64->64 weight is the direct pack [9][64][64]
// (LFSR_CONV3_DIRECT_FLOATS) immediately followed by its Winograd-domain pack (LFSR_CONV3_WINO_FLOATS), see lfsr_pack_wino.
#define LFSR_CONV3_DIRECT_FLOATS (9 * 64 * 64)
// The Winograd part is the F(2x2,3x3) pack (LFSR_CONV3_WINO2_FLOATS) followed by the F(4x4,3x3) pack of conv3x3_wino4.hip.
#define LFSR_CONV3_WINO2_FLOATS (16 * 64 * 64)
#define LFSR_CONV3_WINO4_FLOATS (36 * 64 * 64)
#define LFSR_CONV3_WINO_FLOATS (LFSR_CONV3_WINO2_FLOATS + LFSR_CONV3_WINO4_FLOATS)
int lfsr_pack_wino(const float* direct_packed, float* out, hipStream_t st);   // every Winograd-domain copy (the operator-level pack)
// ... or only the copies in `mask` (LFSR_W_WINO2 | LFSR_W_WINO4).  The model runtimes repack every weight each
// training step and write only what the selected 3x3 kernel reads: lfsr_conv3_variant_mask() = the copies LFSR_CONV3X3 selects (default: wino4).
// The selection is read when weights are packed AND when a conv is launched: set it before loading a model.
enum { LFSR_W_WINO2 = 1, LFSR_W_WINO4 = 2, LFSR_W_ALL = 3 };
int lfsr_conv3_variant_mask();          // union of the copies the forward (LFSR_CONV3X3) and the data-gradient (LFSR_DGRAD3) selections read
const char* lfsr_conv3_fwd_sel();
const char* lfsr_conv3_dgrad_sel();
// Batched repack (training: every weight is repacked every step; one launch per pack KIND with a device-side descriptor table instead of one 4-us
// launch per weight and layout).  kind 0: lfsr_pack_conv_weight's direct pack (perm 0 / 1), 1: lfsr_pack_weight_T (flip = taps reversed), 2: lfsr_pack_weight_chunkT.
struct LfsrPackDesc { const float* src; float* dst; float* dst2; int kind, O, C, T, Npad, perm, ch, flip; };
int lfsr_pack_generic_batch(const LfsrPackDesc* table_dev, int n, hipStream_t st);                    // kinds 0..2 (pack_batch.hip)
int lfsr_pack_conv3_raw_wino4_batch(const LfsrPackDesc* table_dev, int n, hipStream_t st);            // src -> dst (direct) + dst2 (F(4x4) copy); flip = transposed form
int lfsr_pack_epi_wino_batch(const LfsrPackDesc* table_dev, int n, hipStream_t st);                   // src = the direct pack, dst = its F(2,5) copy
int lfsr_pack_conv3_raw_wino4(const float* w_raw, float* direct_out, float* wino4_out, int transposed, hipStream_t st);   // direct + F(4x4) copies in one launch
int lfsr_pack_wino_m(const float* direct_packed, float* out, int mask, hipStream_t st);
int lfsr_pack_conv_weight_m(const float* w, float* packed, int O, int C, int taps, int perm, int ch, int mask, void* stream);
int lfsr_pack_weight_T_m(const float* w, float* out, int O, int C, int T, int flip, int mask, hipStream_t st);
int lfsr_pack_wino4(const float* direct_packed, float* out, hipStream_t st);
// conv3x3_wino4.hip: F(4x4,3x3) form; LFSR_E_ARG = geometry not covered (operands of 1 GiB and more)
int lfsr_conv3x3_wino4_launch(const float* x, int x_stride, int x_choff, const float* w_wino4, float* y, int y_stride, int y_choff,
                              const float* r1, int r1_stride, int r1_choff, const float* r2, int r2_stride, int r2_choff,
                              const float* mk, int mk_stride, int mk_choff, float mk_slope,
                              int n_img, int h, int w, float slope, hipStream_t st);
int lfsr_conv3x3_wino_launch(const float* x, int x_stride, int x_choff, const float* w_wino, const float* w_direct, float* y, int y_stride, int y_choff,
                             const float* r1, int r1_stride, int r1_choff, const float* r2, int r2_stride, int r2_choff,
                             const float* mk, int mk_stride, int mk_choff, float mk_slope,
                             int n_img, int h, int w, float slope, const char* sel, hipStream_t st);
// ang_fused.hip: the AngConv branch (conv AxA stride A 64->16, 1x1 16->16AA, PixelShuffle(A)) in one launch
bool lfsr_ang_fused_ok(int A);
int lfsr_ang_fused_launch(const float* x, int x_stride, int x_choff, const float* w1_packed, const float* w2_packed, float* t, float* y,
                          int y_stride, int y_choff, int B, int A, int h, int w, float slope, hipStream_t st);
// epi_fused.hip: Winograd F(2,5) pack of an EPIConv.0 weight (O = 32, C = 64, taps = 25 = 5 x 5): LFSR_EPI_WINO_FLOATS after the direct pack
#define LFSR_EPI_WINO_FLOATS (5 * 6 * 32 * 64)
// epi_b3.hip: the three bf16 planes of the EPI branch's weights in the order k_epi_b3 stages them: after the F(2,5) copy of EPIConv.0 (O = 32, C = 64, taps = 25)
// and after the direct pack of EPIConv.2 (O = 160, C = 32, taps = 1), both 16-B aligned
#define LFSR_EPI_B3_W1_FLOATS (25 * 32 * 64 * 3 / 2)
#define LFSR_EPI_B3_W2_FLOATS (160 * 32 * 3 / 2)
bool lfsr_epi_use_b3();     // (epi_fused.hip) LFSR_EPI unset: the three-term bf16 kernel at angRes 5
int lfsr_pack_epi_b3(const float* direct_packed, float* out, int kind, hipStream_t st);              // kind 0: EPIConv.0, 1: EPIConv.2
int lfsr_pack_epi_b3_batch(const LfsrPackDesc* table_dev, int n, hipStream_t st);                     // src = the direct pack, dst = its planes, kind as above
int lfsr_epi_b3_launch(const float* x, int x_stride, int x_choff, const float* w1_planes, const float* w2_planes, float* y, int y_stride,
                       int choffH, int choffV, float* t_h, float* t_v, int B, int A, int h, int w, int which, float slope, hipStream_t st);
int lfsr_pack_epi_wino(const float* w1_direct_packed, float* out, hipStream_t st);
// epi_fused.hip  (t_h / t_v: optional (B*A*h*w, 32) buffers receiving the post-LeakyReLU stage-1 activations for backward)
bool lfsr_epi_fused_ok(int A, int h, int w);
int lfsr_epi_fused_launch(const float* x, int x_stride, int x_choff, const float* w1_packed, const float* w2_packed, float* y, int y_stride,
                          int choffH, int choffV, float* t_h, float* t_v, int B, int A, int h, int w, int which, float slope, hipStream_t st);

// gemm_gather.hip: two-launch gather-GEMM EPI path; tmp (B*A*h*w, 32) receives the stage-1 activations
extern "C" int lfsr_epiconv_gather(const float* x, int x_stride, int x_choff, const float* w1_packed, const float* w2_packed,
                        float* tmp, float* y, int y_stride, int y_choff, int B, int A, int h, int w, int vertical, float slope, hipStream_t st);

// bwd_ops.hip: one entry for every backward gather-GEMM (dgrad) instantiation
struct LfsrGemm {
  int in_mode, out_mode, cin;
  const float* X; int x_stride, x_choff;
  const float* Wp;
  float* Y; int y_stride, y_choff;
  const float* R1; int r1_stride, r1_choff;        // added after the mask (may alias Y: in-place accumulate)
  const float* Mk; int mk_stride, mk_choff; float mk_slope;
  int M, N, A, h, w, ntaps, CH;
};
int lfsr_bwd_gemm(const LfsrGemm& g, hipStream_t st);
int lfsr_conv3x3_bwd_data_r2(const float* dy, int dy_stride, const float* wT_packed, float* dx, const float* r1, const float* r2, int n_img, int h, int w, hipStream_t st);
int lfsr_conv3x3_bwd_data(const float* dy, int dy_stride, int dy_choff, const float* wT_packed, float* dx, int dx_stride, int dx_choff,
                          const float* r1, int r1_stride, int r1_choff, const float* mk, int mk_stride, int mk_choff, float mk_slope,
                          int n_img, int h, int w, hipStream_t st);
int lfsr_head_bwd_data(const float* dout, const float* wf, float* df, float* g16, int B, int A, int h, int w, int s, hipStream_t st);
int lfsr_colsum(const float* g, int M, int N, float* partial, int* nblk_out, hipStream_t st);
int lfsr_head_fold_bwd(const float* dWf, const float* colsum_partial, int nblk, const float* w0, const float* b0, const float* w2,
                       float* dw0, float* db0, float* dw2, int s, hipStream_t st);
int lfsr_init_gather9(const float* x, float* xg, int B, int A, int h, int w, hipStream_t st);
int lfsr_add_inplace(float* a, const float* b, long long n, hipStream_t st);   // a += b
int lfsr_pack_weight_chunkT(const float* w, float* out, int O, int C, int ch, int perm, hipStream_t st);

// rowgemm.hip: persistent row-streaming GEMM with LDS-resident weights; LFSR_E_ARG = shape not covered (use the gather-GEMM)
int lfsr_rowgemm_dgrad144_launch(const float* dy, int dy_stride, int dy_choff, const float* wT_packed, const float* mk, int mk_stride, int mk_choff, float mk_slope,
                                 float* dx, int dx_stride, int dx_choff, long long M, hipStream_t st);
int lfsr_rowgemm_ln_launch(const float* x, int x_stride, int x_choff, int K, const float* w_packed, const float* ln_g, const float* ln_b, float ln_eps, int ln_cols,
                           const float* pe, int pe_stride, int pe_rows, int pe_div, float* y, int y_stride, int y_choff,
                           float* y2, int y2_stride, int y2_choff, int split_n, long long M, int N, hipStream_t st);
int lfsr_rowgemm_b3_ln_launch(const float* x, int x_stride, int x_choff, int K, const float* w_packed, const float* ln_g, const float* ln_b, float ln_eps, int ln_cols,
                              const float* pe, int pe_stride, int pe_rows, int pe_div, float* y, int y_stride, int y_choff,
                              float* y2, int y2_stride, int y2_choff, int split_n, long long M, int N, hipStream_t st);
// lnlin_b3.hip: LayerNorm + q | k | v projection with the weights in registers and the token rows through LDS (K = 128, N = 384); LFSR_E_ARG = shape not covered
int lfsr_lnlin_b3_launch(const float* x, int x_stride, int x_choff, int K, const float* w_packed, const float* ln_g, const float* ln_b, float ln_eps, int ln_cols,
                         const float* pe, int pe_stride, int pe_rows, int pe_div, float* y, int y_stride, int y_choff,
                         float* y2, int y2_stride, int y2_choff, int split_n, long long M, int N, hipStream_t st);
int lfsr_rowgemm_b3_dgrad_launch(const float* dy, int dy_stride, int dy_choff, const float* wT_packed, const float* mk, int mk_stride, int mk_choff, float mk_slope,
                                 float* dx, int dx_stride, int dx_choff, long long M, int N, hipStream_t st);
int lfsr_rowgemm_b3_launch(const float* x, int x_stride, int x_choff, int K, const float* w_packed, const float* res, int res_stride, int res_choff,
                           float* y, int y_stride, int y_choff, long long M, int N, float slope, hipStream_t st);
int lfsr_rowgemm_launch(const float* x, int x_stride, int x_choff, int K, const float* w_packed, const float* bias,
                        const float* res, int res_stride, int res_choff, float* y, int y_stride, int y_choff, long long M, int N, float slope, hipStream_t st);

// ffn_fused.hip: feed-forward block with the LayerNorm in front of it formed in registers (ln_g / ln_b null: x is already normalised)
int lfsr_ffn_ln_launch(const float* x, int x_stride, int x_choff, const float* ln_g, const float* ln_b, float ln_eps, const float* w1_packed, const float* w2_packed,
                       const float* res, int res_stride, int res_choff, float* y, int y_stride, int y_choff,
                       long long M, int K1, int H, int N2, float slope, hipStream_t st, const void* wsplit = nullptr);   // wsplit: the weights' pre-split image (lfsr_ffn_b3_presplit), optional

int lfsr_ffn_b3_launch(const float* x, int x_stride, int x_choff, const float* ln_g, const float* ln_b, float ln_eps, const float* w1_packed, const float* w2_packed,
                       const float* res, int res_stride, int res_choff, float* y, int y_stride, int y_choff,
                       long long M, int K1, int H, int N2, float slope, hipStream_t st, const void* wsplit = nullptr);
size_t lfsr_ffn_b3_presplit_bytes(int K1, int H, int N2);
int lfsr_ffn_b3_presplit(const float* w1_packed, const float* w2_packed, int K1, int H, int N2, void* out, hipStream_t st);

// attn_mfma.hip: EPI attention on MFMA; LFSR_E_ARG = geometry not covered
int lfsr_epi_attn_mfma_launch(const float* q, int q_stride, int q_choff, const float* k, int k_stride, int k_choff, const float* v, int v_stride, int v_choff,
                              float* o, int o_stride, int o_choff, int nheads, int ns0, int ns1, int ns2, long long bs0, long long bs1, long long bs2,
                              int n1, int n2, long long st1, long long st2, int l1, int r1, int l2, int r2, int clip2, hipStream_t st);
// win_attn_mfma.hip: 5 x 5 spatial window attention (LFT) on the matrix pipe; LFSR_E_ARG = geometry not covered
int lfsr_win_attn_mfma_launch(const float* q, int q_stride, int q_choff, const float* k, int k_stride, int k_choff, const float* v, int v_stride, int v_choff,
                              float* o, int o_stride, int o_choff, int nheads, int ns0, int ns1, int ns2, long long bs0, long long bs1, long long bs2,
                              int n1, int n2, long long st1, long long st2, int l1, int r1, int l2, int r2, int clip2, hipStream_t st);

// wgrad.hip
int lfsr_wgrad_splits(int M, int ntaps, int K);
size_t lfsr_wgrad_partial_floats(int M, int ntaps, int N, int K);
int lfsr_wgrad_launch(int gmode, int xmode, const float* G, int g_stride, int g_choff, const float* X, int x_stride, int x_choff,
                      float* P, int M, int N, int K, int A, int h, int w, int ntaps, hipStream_t st);
// halo-tile 3x3 conv weight gradient: partials P [lfsr_wgrad_conv3_blocks()][9][64][64], reduce with nsplit = that count
int lfsr_wgrad_conv3_blocks(int n_img, int h, int w);
int lfsr_wgrad_pw144_blocks(int M);
int lfsr_wgrad_pw144_launch(const float* G, int g_stride, int g_choff, const float* X, int x_stride, int x_choff, float* P, int M, hipStream_t st);
int lfsr_wgrad_epi0_blocks(int B, int A, int h, int w, int vert);
int lfsr_ang0_dgrad_launch(const float* dA16, const float* w_direct, float* dx, int dx_stride, int dx_choff, int B, int A, int h, int w, hipStream_t st);
int lfsr_epi0_dgrad_launch(const float* dE, const float* w_direct, float* dx, int dx_stride, int dx_choff, int B, int A, int h, int w, int vert, hipStream_t st);
int lfsr_wgrad_epi0_launch(const float* dE, const float* dE_v, const float* X, int x_stride, int x_choff, float* P, int B, int A, int h, int w, int vert, hipStream_t st);
int lfsr_wgrad_conv3_launch(const float* G, int g_stride, int g_choff, const float* X, int x_stride, int x_choff, float* P,
                            int n_img, int h, int w, hipStream_t st);
// c_valid < C: only the first c_valid input channels are written, with row length c_valid (init_conv's 9 taps)
int lfsr_wgrad_reduce(const float* P, int nsplit, const float* P2, int nsplit2, float* dW, int O, int C, int T, int perm, int ch,
                      int accumulate, int c_valid, int chunk_mode, hipStream_t st);
// chunk_mode = 1: the T 'taps' of the partials are the chunks of a (1-D) pixel shuffle: row = perm ? n*T + t : t*ch + n (n < ch), dW (T*ch, C)
int lfsr_pack_weight_T(const float* w, float* out, int O, int C, int T, int flip, hipStream_t st);

// branch_bwd.cpp: backward of the angular / epipolar branches (DistgSSR.py:84-97,108) on packed weights; C-ABI wrappers lfsr_angconv_bwd / lfsr_epiconv_hv_bwd
size_t lfsr_branch_bwd_partial_floats(int B, int A, int h, int w);
int lfsr_ang_branch_bwd(const float* dcat, int dc_stride, int dc_choff, const float* xin, const float* a16, const float* w0_packed, const float* w0T_packed,
                        const float* w2T_packed, float* dx, float* dw0, float* dw2, float* dA16, float* P, int B, int A, int h, int w, float slope, hipStream_t st);
int lfsr_epi_branch_bwd(const float* dcat, int dc_stride, int choff_h, int choff_v, const float* xin, const float* eh, const float* ev,
                        const float* w0_packed, const float* w0T_packed, const float* w2T_packed, float* dx, float* dw0, float* dw2,
                        float* dEh, float* dEv, float* const P[4], int B, int A, int h, int w, float slope, hipStream_t st);
// ... the same in parts, for callers that run independent parts on two streams (lfsr_distgssr_backward): p1 = everything but the read-modify-write of dx
int lfsr_ang_branch_bwd_p1(const float* dcat, int dc_stride, int dc_choff, const float* xin, const float* a16, const float* w2T_packed, float* dw0, float* dw2,
                           float* dA16, float* P, int B, int A, int h, int w, float slope, hipStream_t st);
int lfsr_ang_branch_bwd_p2(const float* dA16, const float* w0_packed, const float* w0T_packed, float* dx, int B, int A, int h, int w, hipStream_t st);
int lfsr_epi_branch_bwd_p1(const float* dcat, int dc_stride, int choff_h, int choff_v, const float* eh, const float* ev, const float* w2T_packed, float* dw2,
                           float* dEh, float* dEv, float* const P[4], int B, int A, int h, int w, float slope, hipStream_t st);
int lfsr_epi_branch_bwd_p2d(const float* dEh, const float* dEv, const float* w0_packed, const float* w0T_packed, float* dx, int B, int A, int h, int w, hipStream_t st);
int lfsr_epi_branch_bwd_p2w(const float* dEh, const float* dEv, const float* xin, float* dw0, float* const P[4], int B, int A, int h, int w, hipStream_t st);
