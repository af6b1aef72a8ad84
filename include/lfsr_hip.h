/*
 * lfsr_hip.h -- C ABI of liblfsr_hip.so: the MI355X (gfx950) light-field SR hot path.
 *
 * Drop-in boundary for the reference's arithmetic layer (stock PyTorch ops called from
 * model/SR/{DistgSSR,EPIT,LFT,LF_InterNet}.py and utils/utils.py of the BasicLFSR fork).  Every entry
 * point cites the reference interface it replaces (file:line relative to the reference root).
 *
 * Conventions
 *   - plain pointers and sizes only; all data pointers are DEVICE pointers borrowed from the caller
 *     (never freed or retained past the call, except packed weights / workspaces the caller hands
 *     to a model context and keeps alive itself);
 *   - outputs are pre-allocated by the caller;
 *   - every call is asynchronous on `stream` (a hipStream_t passed as void*; NULL = default stream);
 *   - return value: 0 on success, LFSR_E_ARG (-1) for a bad argument, LFSR_E_WS (-2) for a too-small
 *     workspace, -(1000 + hipError_t) for a HIP runtime failure.  No exceptions cross the ABI;
 *   - re-entrant; the only process-wide state is per-device caches of idempotent function attributes / CU counts (atomic flags) and the
 *     lazily resolved RCCL entry points.
 *
 * Tensor layouts
 *   NCHW "SAI mosaic"  (B,C,A*h,A*w)  element [b,c,u*h+y,v*w+x]      -- what the reference passes around
 *   NCHW "MacPI"       (B,C,h*A,w*A)  element [b,c,y*A+u,x*A+v]      -- DistgSSR / LF_InterNet interior
 *   VCL  "view-major channel-last" [B][A*A][h][w][C] fp32             -- this library's interior layout:
 *        one pixel's C channels are contiguous (256 B for C=64), so spatial, angular and epipolar
 *        gathers are all coalesced and SAI<->MacPI rearranges disappear from the network interior.
 *        A VCL operand is described by (ptr, stride, choff): channel c of pixel p lives at
 *        ptr[p*stride + choff + c] (lets a branch write straight into a slice of a concat buffer).
 */
#ifndef LFSR_HIP_H
#define LFSR_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LFSR_OK 0
#define LFSR_E_ARG (-1)
#define LFSR_E_WS (-2)

/* library / build identification: returns a static NUL-terminated string ("lfsr_hip gfx950 ...") */
const char* lfsr_version(void);

/* ------------------------------------------------------------------------------------------------
 * a1-a7: integer-indexing primitives, bit-exact, reference NCHW layouts.  elem_bytes is 2 or 4.
 * ---------------------------------------------------------------------------------------------- */

/* SAI2MacPI, model/SR/DistgSSR.py:145-155 (dup LF_InterNet.py:155-165).  in/out (B,C,A*h,A*w). */
int lfsr_sai2macpi(const void* in, void* out, int B, int C, int A, int h, int w, int elem_bytes, void* stream);
/* MacPI2SAI, model/SR/DistgSSR.py:134-142 (dup LF_InterNet.py:144-152). */
int lfsr_macpi2sai(const void* in, void* out, int B, int C, int A, int h, int w, int elem_bytes, void* stream);
/* nn.PixelShuffle(r) as called at DistgSSR.py:26,89; EPIT.py:46; LFT.py:54; LF_InterNet.py:51,114,132.
 * in (B,C*r*r,H,W) -> out (B,C,H*r,W*r). */
int lfsr_pixel_shuffle2d(const void* in, void* out, int B, int C, int r, int H, int W, int elem_bytes, void* stream);
/* PixelShuffle1D, model/SR/DistgSSR.py:114-131 (factor-major).  in (B,f*C,H,W) -> out (B,C,H,W*f). */
int lfsr_pixel_shuffle1d(const void* in, void* out, int B, int C, int f, int H, int W, int elem_bytes, void* stream);
/* ImageExtend, utils/utils.py:137-149.  in (N,h,w) -> out (N,h+top+bottom,w+left+right), symmetric. */
int lfsr_image_extend(const void* in, void* out, int N, int h, int w, int top, int bottom, int left, int right,
                      int elem_bytes, void* stream);
/* LFdivide, utils/utils.py:152-166.  in (A*h0,A*w0) -> out (numU,numV,A*P,A*P);
 * numU=(h0+2*bdr-1)/S, numV=(w0+2*bdr-1)/S, bdr=(P-S)/2 (returned through num_u/num_v if non-NULL;
 * with out == NULL only the counts are computed). */
int lfsr_lf_divide(const void* in, void* out, int A, int h0, int w0, int P, int S, int elem_bytes,
                   int* num_u, int* num_v, void* stream);
/* LFintegrate, utils/utils.py:169-178.  in (numU,numV,A*pz,A*pz) -> out (A,A,h,w). */
int lfsr_lf_integrate(const void* in, void* out, int A, int numU, int numV, int pz, int stride, int h, int w,
                      int elem_bytes, void* stream);

/* LFintegrate factored for patch-sharded ranks (utils/utils.py:169-178 keeps only the centre stride x stride of every view of every SR patch):
 * crop: in (count,A*pz,A*pz) SR patches -> tiles (count,A,A,stride,stride), run by the rank that computed the patches (a quarter of the bytes then
 * crosses xGMI); place: the tiles of patches [first, first+count) of the row-major (numU,numV) list -> out (A,A,h,w), pixels beyond (h,w) dropped.
 * place(crop(x)) over all patches == lfsr_lf_integrate(x), bit for bit. */
int lfsr_lf_crop_tiles(const void* in, void* tiles, int A, int count, int pz, int stride, int elem_bytes, void* stream);
int lfsr_lf_place_tiles(const void* tiles, void* out, int A, int numU, int numV, int first, int count, int stride, int h, int w,
                        int elem_bytes, void* stream);

/* NCHW <-> VCL (fp32).  layout: 0 = SAI mosaic, 1 = MacPI.  VCL side described by (stride, choff). */
int lfsr_nchw_to_vcl(const float* in, float* out, int out_stride, int out_choff, int B, int C, int A, int h, int w,
                     int layout, void* stream);
int lfsr_vcl_to_nchw(const float* in, int in_stride, int in_choff, float* out, int B, int C, int A, int h, int w,
                     int layout, void* stream);

/* ------------------------------------------------------------------------------------------------
 * d1-d9: DistgSSR operator classes on VCL tensors (fp32, MFMA v_mfma_f32_32x32x2_f32 = exact fp32).
 * Weights are passed PACKED (see lfsr_pack_*): [tap][Npad][Cin], k contiguous, Npad = N rounded up to 32.
 * ---------------------------------------------------------------------------------------------- */

/* Pack a PyTorch conv weight (O,C,kh,kw) (device, fp32) into [kh*kw][Npad][C].
 * perm = 0: n' = n.   perm = 1 (PixelShuffle(A) feeding VCL views, DistgSSR.py:87-89): the reference's
 * output channel c*r2 + q becomes n' = q*ch + c (r2 = O/ch views, ch channels per view).
 * A 3x3 64->64 weight (O = C = 64, taps = 9, perm = 0) is followed by its Winograd-domain copies U = G g G^t
 * (computed in fp64, rounded once): F(2x2,3x3), 16 x 64 x 64 floats in the fragment order of conv3x3_wino.hip, then
 * F(4x4,3x3), 36 x 64 x 64 floats in the fragment order of conv3x3_wino4.hip (the default kernel);
 * lfsr_packed_weight_floats includes both and lfsr_conv3x3_fwd expects them there. */
int lfsr_pack_conv_weight(const float* w, float* packed, int O, int C, int taps, int perm, int ch, void* stream);
size_t lfsr_packed_weight_floats(int O, int C, int taps);

/* per-view 3x3 conv, zero pad 1 (== the MacPI conv "k3, dilation A, padding A" of DistgSSR.py:22,47,64,
 * 79-83,101; == Conv3d(1,3,3) of EPIT.py:24-32,136-142 / LFT.py:36-46), Cin=Cout=64:
 *   y = act(conv(x)) [+ r1] [+ r2],  act = LeakyReLU(slope) if slope != 1.0f.
 * n_img = B*A*A images of h x w. */
int lfsr_conv3x3_fwd(const float* x, int x_stride, int x_choff, const float* w_packed,
                     float* y, int y_stride, int y_choff,
                     const float* r1, int r1_stride, int r1_choff, const float* r2, int r2_stride, int r2_choff,
                     int n_img, int h, int w, float slope, void* stream);

/* pointwise (1x1) conv: y[p, choff + n] = act(sum_k x[p,k] * W[n,k] (+bias[n])), Cin in {16,32,64,144},
 * any N (DistgSSR.py:99 fuse.0).  M pixels. */
int lfsr_pointwise_fwd(const float* x, int x_stride, int x_choff, int cin, const float* w_packed, const float* bias,
                       float* y, int y_stride, int y_choff, int M, int N, float slope, void* stream);

/* AngConv, DistgSSR.py:84-90: t = lrelu(conv AxA stride A 64->16); y = lrelu(1x1 16->16*A*A) scattered
 * by PixelShuffle(A) to the A*A views.  tmp: (B*h*w*16) floats. w1 packed perm 0, w2 packed perm 1 (ch 16). */
int lfsr_angconv_fwd(const float* x, int x_stride, int x_choff, const float* w1_packed, const float* w2_packed,
                     float* tmp, float* y, int y_stride, int y_choff, int B, int A, int h, int w, float slope, void* stream);

/* EPIConv, DistgSSR.py:91-97 (horizontal: vertical = 0) and its transposed application DistgSSR.py:108
 * (vertical = 1, same weights): t = lrelu(conv 1xA^2 stride A pad A(A-1)/2 64->32);
 * y = lrelu(1x1 32->32*A) scattered by PixelShuffle1D(A).  tmp: (B*A*h*w*32) floats; on return it holds the pass's stage-1 activation t, rows ordered
 * (b*A+u, y, x) / (b*A+v, y, x) -- what lfsr_epiconv_hv_bwd takes as e_h / e_v. */
int lfsr_epiconv_fwd(const float* x, int x_stride, int x_choff, const float* w1_packed, const float* w2_packed,
                     float* tmp, float* y, int y_stride, int y_choff, int B, int A, int h, int w, int vertical,
                     float slope, void* stream);

/* Both EPI passes of a DisentgBlock (DistgSSR.py:107-108) in one launch: horizontal result to channel slice
 * choff_h, vertical (transposed application, same weights) to choff_v of y.  Uses the fused LDS-tile kernel
 * when (A odd, A <= 5, h,w <= 32), else two gather-GEMM launches per pass through tmp (may be NULL if fused). */
int lfsr_epiconv_hv_fwd(const float* x, int x_stride, int x_choff, const float* w1_packed, const float* w2_packed,
                        float* tmp, float* y, int y_stride, int choff_h, int choff_v, int B, int A, int h, int w,
                        float slope, void* stream);

/* init_conv, DistgSSR.py:22,32 fused with SAI2MacPI (DistgSSR.py:31): x (B,1,A*h,A*w) SAI mosaic NCHW,
 * w (64,1,3,3) raw PyTorch layout -> y VCL 64 channels. */
int lfsr_initconv_fwd(const float* x, const float* w, float* y, int y_stride, int y_choff, int B, int A, int h, int wd,
                      void* stream);

/* upsample head, DistgSSR.py:24-27,35 fused with MacPI2SAI (:34), PixelShuffle(s) and the bilinear skip
 * (:30): out (B,1,A*h*s,A*w*s) = PS_s(W'f + b') + bilinear_s(x_lr), with the linear chain
 * 1x1(64->64 s^2, bias) -> PixelShuffle -> 1x1(64->1) folded to W' (s^2 x 64), b' (s^2) by lfsr_fold_head. */
int lfsr_fold_head(const float* w0, const float* b0, const float* w2, float* wf, float* bf, int C, int s, void* stream);
int lfsr_upsample_head_fwd(const float* f, int f_stride, int f_choff, const float* wf, const float* bf,
                           const float* x_lr, float* out, int B, int A, int h, int w, int s, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Whole-model driver: DistgSSR forward (get_model.forward, DistgSSR.py:29-36).
 * ---------------------------------------------------------------------------------------------- */
typedef struct lfsr_distgssr lfsr_distgssr;

int lfsr_distgssr_create(lfsr_distgssr** ctx, int A, int scale, int n_group, int n_block, int channels);
void lfsr_distgssr_destroy(lfsr_distgssr* ctx);
/* bytes of device memory the packed weights need; caller allocates and hands over with set_packed */
size_t lfsr_distgssr_packed_bytes(const lfsr_distgssr* ctx);
int lfsr_distgssr_set_packed(lfsr_distgssr* ctx, void* packed, size_t bytes);
/* feed one state_dict entry (reference key names), raw PyTorch layout, device fp32.
 * Returns LFSR_E_ARG for an unknown key or a wrong element count. */
int lfsr_distgssr_load_param(lfsr_distgssr* ctx, const char* key, const float* data, size_t numel, void* stream);
/* after all params are loaded: folds the upsample head; returns LFSR_E_ARG if a parameter is missing */
int lfsr_distgssr_finalize(lfsr_distgssr* ctx, void* stream);
/* optional, before a round of lfsr_distgssr_load_param calls (a training step's repack): the packs are recorded instead of launched one by one and
 * lfsr_distgssr_finalize launches them, one kernel per pack kind (the descriptor table lives in a small device buffer owned by the context and is
 * re-uploaded only when a parameter's address changed).  The data pointers passed to load_param must stay valid until finalize's work has run. */
int lfsr_distgssr_begin_batched_load(lfsr_distgssr* ctx);
size_t lfsr_distgssr_workspace_bytes(const lfsr_distgssr* ctx, int B, int h, int w);
/* x (B,1,A*h,A*w) -> out (B,1,A*h*s,A*w*s), both NCHW SAI mosaics, fp32 */
int lfsr_distgssr_forward(lfsr_distgssr* ctx, const float* x, float* out, int B, int h, int w,
                          void* workspace, size_t workspace_bytes, void* stream);
/* parity variant: additionally writes interior activations as NCHW MacPI (B,64,h*A,w*A) tensors into
 * taps[i] (device pointers, NULL = skip): 0 = init_conv output, 1 = block(0,0) output, 2 = group 0 output,
 * 3 = disentg output; and block(0,0)'s concat buffer (B,144,h*A,w*A) into taps[4]. */
int lfsr_distgssr_forward_taps(lfsr_distgssr* ctx, const float* x, float* out, int B, int h, int w,
                               void* workspace, size_t workspace_bytes, float* const* taps, void* stream);

/* ---- training (config "DistgSSR x4 training, data-parallel"): forward that keeps the activations, and backward ----
 * Gradients are written into ONE flat fp32 bucket in state_dict order (lfsr_distgssr_param_offset gives each
 * parameter's span): exactly the buffer a data-parallel job hands to a single RCCL all-reduce.
 * The workspace must be the same memory for forward_train and the backward that follows it. */
size_t lfsr_distgssr_num_params(const lfsr_distgssr* ctx);
int lfsr_distgssr_param_offset(const lfsr_distgssr* ctx, const char* key, size_t* offset, size_t* numel);
size_t lfsr_distgssr_train_workspace_bytes(const lfsr_distgssr* ctx, int B, int h, int w);
int lfsr_distgssr_forward_train(lfsr_distgssr* ctx, const float* x, float* out, int B, int h, int w,
                                void* workspace, size_t workspace_bytes, void* stream);
/* parity aid: offset (in floats, into the training workspace) and size of an activation forward_train saved for the backward.  which: 0 SpaConv.0
 * output (VCL, 64 ch), 1 the concat buffer (VCL, 144 ch), 2 AngConv.0 output (rows (b,y,x), 16 ch), 3 / 4 EPIConv.0 output of the horizontal /
 * vertical pass (rows (b*A+u,y,x) / (b*A+v,y,x), 32 ch), 5 fuse.0 output (VCL, 64), 6 block output (VCL, 64); index = group * n_block + block.
 * All are post-LeakyReLU values: their signs are the LeakyReLU' masks the data gradients apply (DistgSSR.py:79-101). */
int lfsr_distgssr_train_saved(const lfsr_distgssr* ctx, int B, int h, int w, int which, int index, size_t* offset_floats, size_t* numel);
/* dout (B,1,A*h*s,A*w*s) = dLoss/dOut; grads: n_grads == lfsr_distgssr_num_params(ctx) floats, overwritten */
int lfsr_distgssr_backward(lfsr_distgssr* ctx, const float* x, const float* dout, int B, int h, int w,
                           void* workspace, size_t workspace_bytes, float* grads, size_t n_grads, void* stream);

/* ---- operator-level backward entry points (SURVEY 8b export list: conv3x3 dgrad / wgrad, pointwise bwd, upsample_head bwd).
 * They replace what autograd derives from nn.Conv2d in the reference's training step (train.py:256-264, fp32). ---- */
/* transposed pack used by the data gradients: w (O,C,kh,kw) raw PyTorch layout -> [tap'][Cpad][O] (3x3: taps flipped; O = C = 64, taps = 9
 * is followed by the Winograd-domain copies, as lfsr_pack_conv_weight does) */
size_t lfsr_packed_weight_tr_floats(int O, int C, int taps);
int lfsr_pack_conv_weight_tr(const float* w, float* packed_T, int O, int C, int taps, void* stream);
/* dx = (conv3x3^T(dy)) * LeakyReLU'(act) + r1: `act` = the saved output of the LeakyReLU(act_slope) in front of this conv's input
 * (NULL: none), r1 = a gradient arriving over a skip connection (NULL: none).  64 -> 64, per-view zero pad 1. */
int lfsr_conv3x3_dgrad(const float* dy, int dy_stride, int dy_choff, const float* wT_packed, float* dx, int dx_stride, int dx_choff,
                       const float* r1, int r1_stride, int r1_choff, const float* act, int act_stride, int act_choff, float act_slope,
                       int n_img, int h, int w, void* stream);
/* dw (64,64,3,3) raw PyTorch layout [= or += if accumulate] sum_pixels dy (x) shifted x; workspace: per-block partial slabs, reduced in a
 * second deterministic pass (no float atomics) */
size_t lfsr_conv3x3_wgrad_workspace_floats(int n_img, int h, int w);
int lfsr_conv3x3_wgrad(const float* dy, int dy_stride, int dy_choff, const float* x, int x_stride, int x_choff, float* dw,
                       float* workspace, size_t workspace_floats, int n_img, int h, int w, int accumulate, void* stream);
/* 1x1 conv y = x W^T, W (cout = 64, cin): dx = (dy W) * LeakyReLU'(act);  dw (cout, cin) = dy^T x  (cout <= 64) */
int lfsr_pointwise_dgrad(const float* dy, int dy_stride, int dy_choff, int cout, const float* wT_packed, float* dx, int dx_stride, int dx_choff, int cin,
                         const float* act, int act_stride, int act_choff, float act_slope, long long M, void* stream);
size_t lfsr_pointwise_wgrad_workspace_floats(long long M, int cout, int cin);
int lfsr_pointwise_wgrad(const float* dy, int dy_stride, int dy_choff, int cout, const float* x, int x_stride, int x_choff, int cin, float* dw,
                         float* workspace, size_t workspace_floats, long long M, int accumulate, void* stream);
/* data gradient of lfsr_upsample_head_fwd w.r.t. f: dout (B,1,A*h*s,A*w*s) -> df (pixels, 64) VCL; g16 (pixels, 16) scratch that receives the
 * un-shuffled output gradient (the operand of the folded matrix' weight gradient) */
int lfsr_upsample_head_dgrad(const float* dout, const float* wf, float* df, float* g16, int B, int A, int h, int w, int s, void* stream);

/* ---- the exchange step of data-parallel training: in-place sum-all-reduce of n fp32 values over RCCL (xGMI) -----------
 * The reference has no gradient exchange (single process; option.py:28 `--local_rank` is vestigial).  `comm` is an ncclComm_t created by
 * lfsr_comm_init (or by the caller with RCCL directly); unique_id = the 128-byte ncclUniqueId obtained on rank 0 and shipped to the others by
 * the host.  RCCL is resolved at run time (dlopen): lfsr_comm_available() is 0 on a host without it, and the calls then return LFSR_E_ARG.
 * Return: 0, LFSR_E_ARG, or -(2000 + ncclResult_t). */
int lfsr_comm_available(void);
int lfsr_comm_unique_id(void* id128);
int lfsr_comm_init(void** comm, int world, int rank, const void* id128);
int lfsr_comm_destroy(void* comm);
int lfsr_allreduce(void* grads, size_t n, void* comm, void* stream);

/* Per-operator-class timing with hipEvents recorded on the launch stream around every launch of the
 * forward (measurement aid for bench.py's roofline line; off by default).
 * classes: 0 conv3x3, 1 angconv, 2 epiconv, 3 pointwise (fuse.0), 4 init_conv, 5 upsample head. */
#define LFSR_DISTG_NCLASS 6
int lfsr_distgssr_profile(lfsr_distgssr* ctx, int enable);   /* 0 off, 1 every class, 2 class 0 (3x3 conv) only, 3 every 4th launch of class 0; drops recorded events */
/* waits for the recorded events, returns summed milliseconds and launch counts per class, then resets */
int lfsr_distgssr_profile_read(lfsr_distgssr* ctx, double* ms, long long* launches);

/* ------------------------------------------------------------------------------------------------
 * e1-e7 / l1-l5: transformer operator classes (EPIT.py:74-128, LFT.py:133-246) on VCL rows (tokens == pixels).
 * ---------------------------------------------------------------------------------------------- */
/* nn.LayerNorm(C) over M rows of (x [+ pe[(row / pe_div) % pe_rows]]); C in {64,128} (EPIT.py:78,83; LFT.py:142,150,211,215) */
int lfsr_layernorm_fwd(const float* x, int x_stride, int x_choff, const float* pe, int pe_stride, long long pe_rows, long long pe_div,
                       const float* gamma, const float* beta, float* y, int y_stride, int y_choff, long long M, int C,
                       float eps, void* stream);
/* nn.Linear (no transposes needed: weight (N,K) packed by lfsr_pack_conv_weight(O=N,C=K,taps=1)); K in {64,128,256};
 * y = act(x W^T + bias) + res;  slope 1 = identity, 0 = ReLU, else LeakyReLU.
 * Arithmetic: fp32 in, fp32 out, fp32 accumulation.  The bias-free K = 64 / 128 case (and lfsr_ffn_*, lfsr_linear_ln_fwd, lfsr_up_tail_fwd's 1x1 conv) runs on the bf16
 * MFMA pipe with every fp32 operand split EXACTLY into three bf16 terms (six products; error against fp64 no larger than the fp32-MFMA kernels': tools/b3_accuracy.py);
 * the environment selectors LFSR_ROWGEMM=f32, LFSR_FFN=f32, LFSR_UPTAIL=v2 choose the fp32-MFMA kernels. */
int lfsr_linear_fwd(const float* x, int x_stride, int x_choff, int cin, const float* w_packed, const float* bias,
                    const float* res, int res_stride, int res_choff, float* y, int y_stride, int y_choff,
                    long long M, int N, float slope, void* stream);
/* Fused feed-forward block of the transformers (EPIT.py:84-90,126; LFT.py:151-156,202,216-221,243):
 *   y = res + W2 . act(W1 . x)      x: (M, K1) LayerNorm'd tokens, W1 (H, K1), W2 (N2, H) packed as for lfsr_linear_fwd, no biases;
 * the (M, H) hidden activations stay on chip.  (K1, N2) in {(128,128), (64,64)}, H a multiple of 32; slope 0 = ReLU. */
int lfsr_ffn_fwd(const float* x, int x_stride, int x_choff, const float* w1_packed, const float* w2_packed,
                 const float* res, int res_stride, int res_choff, float* y, int y_stride, int y_choff,
                 long long M, int K1, int H, int N2, float slope, void* stream);
/* The same block with its leading LayerNorm (feed_forward.0: EPIT.py:85, LFT.py:152,217) formed in registers:
 *   y = res + W2 . act(W1 . LayerNorm(x))      x: (M, K1) RAW tokens, gamma / beta (K1), eps as nn.LayerNorm */
int lfsr_ffn_ln_fwd(const float* x, int x_stride, int x_choff, const float* gamma, const float* beta, float eps,
                    const float* w1_packed, const float* w2_packed, const float* res, int res_stride, int res_choff,
                    float* y, int y_stride, int y_choff, long long M, int K1, int H, int N2, float slope, void* stream);
/* LayerNorm + attention in-projection in one launch (EPIT.py:113-121; LFT.py:190-197,236-241): weight rows n < ln_cols (q | k) see
 * LayerNorm(x [+ pe[(row / pe_div) % pe_rows]]), rows n >= ln_cols (v) see x; columns n >= split_n go to y2 (column n - split_n).
 * K in {64,128}; N, ln_cols, split_n multiples of 64.  Same bits as lfsr_layernorm_fwd + lfsr_linear_fwd. */
int lfsr_linear_ln_fwd(const float* x, int x_stride, int x_choff, int K, const float* w_packed, const float* gamma, const float* beta,
                       float eps, int ln_cols, const float* pe, int pe_stride, int pe_rows, int pe_div,
                       float* y, int y_stride, int y_choff, float* y2, int y2_stride, int y2_choff, int split_n,
                       long long M, int N, void* stream);
/* nn.MultiheadAttention core with the reference's additive window mask evaluated as a predicate (EPIT.py:93-122,
 * LFT.py:161-199,238-241): o = softmax(q k^T / sqrt(hd) + mask) v per head; hd in {8,16}.
 * Sequences (s0,s1,s2) start at pixel s0*bs0+s1*bs1+s2*bs2; token (t1,t2) sits at + t1*st1 + t2*st2; token (t1,t2)
 * attends keys [t1-l1, t1+r1) x [t2-l2, min(t2+r2, clip2 or n2)) clipped to the grid. */
int lfsr_window_attn_fwd(const float* q, int q_stride, int q_choff, const float* k, int k_stride, int k_choff,
                         const float* v, int v_stride, int v_choff, float* o, int o_stride, int o_choff, int nheads, int hd,
                         int ns0, int ns1, int ns2, long long bs0, long long bs1, long long bs2,
                         int n1, int n2, long long st1, long long st2, int l1, int r1, int l2, int r2, int clip2, void* stream);
/* per-view 3x3 conv 64 -> N for any N (gather-GEMM): LFT's unfold(3x3) + Linear(576 -> 128) token embedding (LFT.py:176-182) */
int lfsr_conv3x3_n_fwd(const float* x, int x_stride, int x_choff, const float* w_packed, float* y, int y_stride, int y_choff,
                       int n_img, int h, int w, int N, float slope, void* stream);
/* PositionEncoding.forward (LFT.py:106-130): spa_pe (h*w, C), ang_pe (A*A, C) */
int lfsr_lft_position_fwd(float* spa_pe, float* ang_pe, int A, int h, int w, int C, void* stream);
/* up-sampling tail shared by EPIT (EPIT.py:44-49) and LFT (LFT.py:52-57):
 * 1x1 64->64 s^2 (no bias) + PixelShuffle(s) into the channel-last HR mosaic (B, A*h*s, A*w*s, 64) [w packed perm 1, ch 64];
 * then LeakyReLU(slope) -> 3x3 conv 64->1 (zero pad 1 over the whole mosaic) + per-view bicubic skip of x_lr. */
int lfsr_upsample_ps_fwd(const float* f, int f_stride, int f_choff, const float* w_packed, float* hr, int B, int A, int h, int w,
                         int s, void* stream);
/* both steps fused (s in {2,4}): the (B,64,A h s,A w s) intermediate never exists.  w0_packed as for lfsr_upsample_ps_fwd. */
int lfsr_up_tail_fwd(const float* f, int f_stride, int f_choff, const float* w0_packed, const float* w3, const float* x_lr, float* out,
                     int B, int A, int h, int w, int s, float slope, void* stream);
int lfsr_hr_tail_fwd(const float* hr, const float* w3, const float* x_lr, float* out, int B, int A, int h, int w, int s,
                     float slope, void* stream);

/* Whole-model driver: EPIT forward (get_model.forward, EPIT.py:51-71).  Same life cycle as lfsr_distgssr_*. */
typedef struct lfsr_epit lfsr_epit;
int lfsr_epit_create(lfsr_epit** ctx, int A, int scale, int n_block, int channels);
void lfsr_epit_destroy(lfsr_epit* ctx);
size_t lfsr_epit_packed_bytes(const lfsr_epit* ctx);
int lfsr_epit_set_packed(lfsr_epit* ctx, void* packed, size_t bytes);
int lfsr_epit_load_param(lfsr_epit* ctx, const char* key, const float* data, size_t numel, void* stream);
int lfsr_epit_finalize(lfsr_epit* ctx, void* stream);
size_t lfsr_epit_workspace_bytes(const lfsr_epit* ctx, int B, int h, int w);
int lfsr_epit_forward(lfsr_epit* ctx, const float* x, float* out, int B, int h, int w, void* workspace, size_t workspace_bytes,
                      void* stream);

/* Whole-model driver: LFT forward (get_model.forward, LFT.py:67-98). */
typedef struct lfsr_lft lfsr_lft;
int lfsr_lft_create(lfsr_lft** ctx, int A, int scale, int n_layer, int channels);
void lfsr_lft_destroy(lfsr_lft* ctx);
size_t lfsr_lft_packed_bytes(const lfsr_lft* ctx);
int lfsr_lft_set_packed(lfsr_lft* ctx, void* packed, size_t bytes);
int lfsr_lft_load_param(lfsr_lft* ctx, const char* key, const float* data, size_t numel, void* stream);
int lfsr_lft_finalize(lfsr_lft* ctx, void* stream);
size_t lfsr_lft_workspace_bytes(const lfsr_lft* ctx, int B, int h, int w);
int lfsr_lft_forward(lfsr_lft* ctx, const float* x, float* out, int B, int h, int w, void* workspace, size_t workspace_bytes,
                     void* stream);

/* ---- "next" rows: the steps either side of the hot path in the training loop, on the device ----
 * N2: cal_metrics (utils/utils.py:91-134): per-view PSNR (and SSIM, skimage semantics with gaussian_weights=True) of two
 * (B,1,A*H,A*W) SAI mosaics, fp64 accumulation; psnr/ssim: B*A*A doubles (ssim may be NULL; SSIM needs views >= 11x11). */
int lfsr_view_metrics(const float* label, const float* out, double* psnr, double* ssim, int B, int A, int H, int W, void* stream);
/* N3: MaskedAngularPretraining.forward, utils/masked_pretraining.py:85-139: y = x with the views flagged in mask[A*A]
 * (device bytes) filled with `fill` ('zero' / 'mean' strategies; the reference applies one mask to the whole batch). */
int lfsr_mask_views(const float* x, float* y, const unsigned char* mask, float fill, int B, int C, int A, int h, int w, void* stream);
/* the same with one fill value per view, fill[A*A] device floats (mask_value 'mean' with several masked views: each view is filled with
 * its OWN mean over batch, channels and pixels, masked_pretraining.py:121-123) */
int lfsr_mask_views_fill(const float* x, float* y, const unsigned char* mask, const float* fill, int B, int C, int A, int h, int w, void* stream);

/* N4, output tail of test() (train.py:329-341, inference.py:205-216; utils/utils.py:191-204 ycbcr2rgb):
 * out[(u, v, y, x, c)] = uint8(clip(M255 . (Y, Cb, Cr) - offset, 0, 1) * 255), evaluated in fp64 like the reference's numpy path.
 * y (A*h, A*w) and cbcr (2, A*h, A*w) fp32 SAI mosaics; minv255 = inv(M) * 255 (row-major 3x3) and offset = inv(M) . (16,128,128),
 * both computed by the caller exactly as the reference does; out: (A, A, h, w, 3) uint8. */
int lfsr_ycbcr2rgb_views(const float* y, const float* cbcr, unsigned char* out, int A, int h, int w, const double* minv255,
                         const double* offset, void* stream);

/* Whole-model driver: LF_InterNet forward (get_model.forward, LF_InterNet.py:33-41); n_groups = n_layers = 4 upstream. */
typedef struct lfsr_internet lfsr_internet;
int lfsr_internet_create(lfsr_internet** ctx, int A, int scale, int n_groups, int n_layers);
void lfsr_internet_destroy(lfsr_internet* ctx);
size_t lfsr_internet_packed_bytes(const lfsr_internet* ctx);
int lfsr_internet_set_packed(lfsr_internet* ctx, void* packed, size_t bytes);
int lfsr_internet_load_param(lfsr_internet* ctx, const char* key, const float* data, size_t numel, void* stream);
int lfsr_internet_finalize(lfsr_internet* ctx, void* stream);
size_t lfsr_internet_workspace_bytes(const lfsr_internet* ctx, int B, int h, int w);
int lfsr_internet_forward(lfsr_internet* ctx, const float* x, float* out, int B, int h, int w, void* workspace,
                          size_t workspace_bytes, void* stream);

/* ---- backward of the disentangling branches as operators (disentg_branches_bwd; reference layers DistgSSR.py:84-97,108, differentiated by
 * autograd in train.py:256-264).  Raw PyTorch-layout weights in (the entry points pack what their kernels read into the workspace), raw-layout weight
 * gradients out (overwritten); dx (pixels, 64) VCL is ACCUMULATED into (the block input receives gradients from four branches).
 * AngConv: y = PixelShuffle_A(lrelu(W2 . a16)), a16 = lrelu(W0 (*) x).  dy: dLoss/dy, 16 channels at dy_choff of a VCL buffer of row stride dy_stride;
 * y: the branch output as lfsr_angconv_fwd wrote it (its signs are the LeakyReLU' mask of stage 2), or NULL when dy already is the gradient at the stage-2
 * pre-activation (what lfsr_pointwise_dgrad with act = the concat buffer leaves); x: the block input (pixels, 64) VCL; a16: the stage-1 activation
 * lfsr_angconv_fwd left in its `tmp` argument ((B h w), 16); w0 (16,64,A,A), w2 (16 A^2,16,1,1). */
size_t lfsr_angconv_bwd_workspace_floats(int B, int A, int h, int w);
int lfsr_angconv_bwd(const float* dy, int dy_stride, int dy_choff, const float* y, int y_stride, int y_choff, const float* x, const float* a16,
                     const float* w0, const float* w2, float* dx, float* dw0, float* dw2, float* workspace, size_t workspace_floats,
                     int B, int A, int h, int w, float slope, void* stream);
/* EPIConv on the tensor and on its transpose (shared weights; both passes' weight gradients summed): dy holds dLoss/dy_h at choff_h and dLoss/dy_v at
 * choff_v (32 channels each); y likewise the two outputs (or NULL, as above); e_h / e_v: the stage-1 activations lfsr_epiconv_fwd(vertical = 0 / 1) left
 * in `tmp` ((B A h w), 32); w0 (32,64,1,A^2), w2 (32 A,32,1,1).  Odd angRes only (symmetric padding of the EPI line). */
size_t lfsr_epiconv_hv_bwd_workspace_floats(int B, int A, int h, int w);
int lfsr_epiconv_hv_bwd(const float* dy, int dy_stride, int choff_h, int choff_v, const float* y, int y_stride, int y_choff_h, int y_choff_v,
                        const float* x, const float* e_h, const float* e_v, const float* w0, const float* w2,
                        float* dx, float* dw0, float* dw2, float* workspace, size_t workspace_floats, int B, int A, int h, int w, float slope, void* stream);

/* ---- arithmetic of the GEMMs that exist in two forms (DistgSSR's EPI branch and fuse.0; the transformers' linears, fused FFN and up-sampling tail) ----
 * LFSR_ARITH_DEFAULT: fp32 operands carried EXACTLY as three bf16 terms on the bf16 MFMA pipe, six products, fp32 accumulation (error against fp64 not above the
 * fp32-MFMA kernels': tests/test_gpu_b3_accuracy.py); LFSR_ARITH_F32: every GEMM on fp32 MFMA.  Process-wide, read at every launch; the 3x3 convs, the angular
 * branch and the attention kernels compute on fp32 MFMA either way.  The reference computes all of these layers with stock fp32 torch ops. */
#define LFSR_ARITH_DEFAULT 0
#define LFSR_ARITH_F32 1
int lfsr_set_arithmetic(int mode);
int lfsr_get_arithmetic(void);

/* ---- operator-level timing hooks (measurement aid; the reference times whole forwards only: check_efficiency_official.py:306-330) ----
 * lfsr_op_profile(1): from now on every instrumented operator entry point brackets its launches with a hipEvent pair on its launch stream (and any
 * earlier records are dropped); lfsr_op_profile(0): off (the default; a hook then costs one atomic load).  lfsr_op_profile_read waits for the recorded
 * events, writes one text line "op a b total_ms launches" per (operator, tag a, tag b) into buf (NUL-terminated, truncated to cap), clears the records and
 * returns the bytes the full text needs, or a negative LFSR_E_* / HIP code.  Tags: conv3x3 / conv3x3_dgrad / conv3x3_wgrad (n_img, h*w), linear (K, N),
 * window_attn (head dim, tokens per sequence), ffn (K, hidden), up_tail (scale, 0), epiconv / angconv / pointwise (cin, N). */
int lfsr_op_profile(int enable);
long long lfsr_op_profile_read(char* buf, size_t cap);

#ifdef __cplusplus
}
#endif
#endif /* LFSR_HIP_H */
