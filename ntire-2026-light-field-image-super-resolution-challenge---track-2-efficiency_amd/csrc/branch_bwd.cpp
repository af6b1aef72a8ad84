// Backward of DistgSSR's angular and epipolar branches as operators (disentg_branches_bwd; reference layers DistgSSR.py:84-97,108; what
// autograd derives for them in train.py:256-264, fp32).  lfsr_distgssr_backward calls the two internal functions; the C-ABI entry points
// lfsr_angconv_bwd / lfsr_epiconv_hv_bwd wrap them for callers that hold raw PyTorch-layout weights (they pack what the kernels read into the
// caller's workspace first).
#include <stdlib.h>

#include "lfsr_internal.h"

namespace {
inline size_t al64(size_t v) { return (v + 63) & ~(size_t)63; }
inline int pad32(int v) { return (v + 31) / 32 * 32; }

// out[row][c] = dy[row][c] * LeakyReLU'(y[row][c]) for C channels (a multiple of 4) of strided VCL operands: the gradient at a branch's stage-2 pre-activation
__global__ __launch_bounds__(256) void k_mask_lrelu(const float* __restrict__ dy, int dy_stride, int dy_choff, const float* __restrict__ y, int y_stride, int y_choff,
                                                   float* __restrict__ out, int out_stride, int out_choff, long long rows, int C, float slope) {
  const int c4 = C >> 2;
  const long long total = rows * c4;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const long long r = i / c4;
    const int c = (int)(i - r * c4) * 4;
    const float4 g = *reinterpret_cast<const float4*>(dy + r * dy_stride + dy_choff + c);
    const float4 a = *reinterpret_cast<const float4*>(y + r * y_stride + y_choff + c);
    *reinterpret_cast<float4*>(out + r * out_stride + out_choff + c) =
        make_float4(a.x > 0.f ? g.x : g.x * slope, a.y > 0.f ? g.y : g.y * slope, a.z > 0.f ? g.z : g.z * slope, a.w > 0.f ? g.w : g.w * slope);
  }
}

int mask_lrelu(const float* dy, int dy_stride, int dy_choff, const float* y, int y_stride, int y_choff, float* out, int out_stride, int out_choff, long long rows, int C,
               float slope, hipStream_t st) {
  if ((dy_stride | dy_choff | y_stride | y_choff | out_stride | out_choff | C) & 3 || (((uintptr_t)dy | (uintptr_t)y | (uintptr_t)out) & 15)) return LFSR_E_ARG;
  unsigned grid = lfsr_blocks(rows * (C >> 2), 256);
  if (grid > 8192) grid = 8192;
  hipLaunchKernelGGL(k_mask_lrelu, dim3(grid), dim3(256), 0, st, dy, dy_stride, dy_choff, y, y_stride, y_choff, out, out_stride, out_choff, rows, C, slope);
  LFSR_CHECK_LAUNCH();
  return LFSR_OK;
}
}  // namespace

// AngConv: y = PixelShuffle_A(lrelu(W2 . a16)), a16 = lrelu(W0 (*) x) (A x A, stride A).  dcat: dLoss/dy inside a VCL buffer (16 channels at dc_choff).
// dx += ; dw0 (16,64,A,A), dw2 (16 A^2,16,1,1) overwritten.  dA16: (B h w, 16) scratch; P: partial-slab scratch of >= lfsr_branch_bwd_partial_floats.
// Two parts so that a caller may run part 1 (everything up to dA16 and both weight gradients) on another stream than part 2 (the read-modify-write of dx).
int lfsr_ang_branch_bwd_p1(const float* dcat, int dc_stride, int dc_choff, const float* xin, const float* a16, const float* w2T_packed, float* dw0, float* dw2,
                           float* dA16, float* P, int B, int A, int h, int w, float slope, hipStream_t st) {
  const int AA = A * A, nlr = B * h * w;
  int rc;
#define RC(call) do { rc = (call); if (rc) return rc; } while (0)
  RC(lfsr_wgrad_launch(LFSR_IN_ANG, LFSR_IN_SAME, dcat, dc_stride, dc_choff, a16, 16, 0, P, nlr, 16, 16, A, h, w, AA, st));
  RC(lfsr_wgrad_reduce(P, lfsr_wgrad_splits(nlr, AA, 16), nullptr, 0, dw2, 16 * AA, 16, AA, 1, 16, 0, 0, 1, st));
  {
    LfsrGemm q{};
    q.in_mode = LFSR_IN_ANG; q.out_mode = LFSR_OUT_SAME; q.cin = 16; q.X = dcat; q.x_stride = dc_stride; q.x_choff = dc_choff; q.Wp = w2T_packed;
    q.Y = dA16; q.y_stride = 16; q.Mk = a16; q.mk_stride = 16; q.mk_slope = slope;
    q.M = nlr; q.N = 16; q.A = A; q.h = h; q.w = w; q.ntaps = AA; q.CH = 16;
    RC(lfsr_bwd_gemm(q, st));
  }
  RC(lfsr_wgrad_launch(LFSR_IN_SAME, LFSR_IN_ANG, dA16, 16, 0, xin, 64, 0, P, nlr, 16, 64, A, h, w, AA, st));
  RC(lfsr_wgrad_reduce(P, lfsr_wgrad_splits(nlr, AA, 64), nullptr, 0, dw0, 16, 64, AA, 0, 0, 0, 0, 0, st));
#undef RC
  return LFSR_OK;
}

int lfsr_ang_branch_bwd_p2(const float* dA16, const float* w0_packed, const float* w0T_packed, float* dx, int B, int A, int h, int w, hipStream_t st) {
  const int AA = A * A, nlr = B * h * w;
  LfsrGemm q{};
  q.in_mode = LFSR_IN_SAME; q.out_mode = LFSR_OUT_VIEWS; q.cin = 16; q.X = dA16; q.x_stride = 16; q.Wp = w0T_packed;
  q.Y = dx; q.y_stride = 64; q.R1 = dx; q.r1_stride = 64;
  q.M = nlr; q.N = AA * 64; q.A = A; q.h = h; q.w = w; q.ntaps = 1; q.CH = 64;
  int rc4 = lfsr_ang0_dgrad_launch(dA16, w0_packed, dx, 64, 0, B, A, h, w, st);   // streaming read-modify-write form; else the gather-GEMM
  if (rc4 == LFSR_E_ARG) rc4 = lfsr_bwd_gemm(q, st);
  return rc4;
}

int lfsr_ang_branch_bwd(const float* dcat, int dc_stride, int dc_choff, const float* xin, const float* a16, const float* w0_packed, const float* w0T_packed,
                        const float* w2T_packed, float* dx, float* dw0, float* dw2, float* dA16, float* P, int B, int A, int h, int w, float slope, hipStream_t st) {
  const int rc = lfsr_ang_branch_bwd_p1(dcat, dc_stride, dc_choff, xin, a16, w2T_packed, dw0, dw2, dA16, P, B, A, h, w, slope, st);
  return rc ? rc : lfsr_ang_branch_bwd_p2(dA16, w0_packed, w0T_packed, dx, B, A, h, w, st);
}

// EPIConv on the tensor (horizontal) and on its transpose (vertical), shared weights: y_h / y_v = PixelShuffle1D_A(lrelu(W2 . e)), e = lrelu(W0 (*) x)
// (1 x A^2, stride A along the EPI line).  dcat: dLoss/dy_h at choff_h, dLoss/dy_v at choff_v (32 channels each) of one VCL buffer.
// dx += ; dw0 (32,64,1,A^2), dw2 (32 A,32,1,1) overwritten (both passes summed).  dEh, dEv: (B A h w, 32) scratch; P[4]: partial-slab scratch.
// Three parts: p1 = stage 2 of both passes (dEh, dEv, dw2), p2d = the read-modify-write of dx, p2w = EPIConv.0's weight gradient (reads dEh, dEv and x only).
int lfsr_epi_branch_bwd_p1(const float* dcat, int dc_stride, int choff_h, int choff_v, const float* eh, const float* ev, const float* w2T_packed, float* dw2,
                           float* dEh, float* dEv, float* const P[4], int B, int A, int h, int w, float slope, hipStream_t st) {
  const int nepi = B * A * h * w;
  int rc;
#define RC(call) do { rc = (call); if (rc) return rc; } while (0)
  for (int vert = 0; vert < 2; ++vert) {
    const float* E = vert ? ev : eh;
    float* dE = vert ? dEv : dEh;
    const int choff = vert ? choff_v : choff_h;
    float* Pa = P[vert ? 2 : 0];   // EPIConv.2 partials
    RC(lfsr_wgrad_launch(vert ? LFSR_IN_CHK_V : LFSR_IN_CHK_H, LFSR_IN_SAME, dcat, dc_stride, choff, E, 32, 0, Pa, nepi, 32, 32, A, h, w, A, st));
    LfsrGemm q{};
    q.in_mode = vert ? LFSR_IN_CHK_V : LFSR_IN_CHK_H; q.out_mode = LFSR_OUT_SAME; q.cin = 32; q.X = dcat; q.x_stride = dc_stride; q.x_choff = choff;
    q.Wp = w2T_packed; q.Y = dE; q.y_stride = 32; q.Mk = E; q.mk_stride = 32; q.mk_slope = slope;
    q.M = nepi; q.N = 32; q.A = A; q.h = h; q.w = w; q.ntaps = A; q.CH = 32;
    RC(lfsr_bwd_gemm(q, st));
  }
  RC(lfsr_wgrad_reduce(P[0], lfsr_wgrad_splits(nepi, A, 32), P[2], lfsr_wgrad_splits(nepi, A, 32), dw2, 32 * A, 32, A, 0, 32, 0, 0, 1, st));
#undef RC
  return LFSR_OK;
}

int lfsr_epi_branch_bwd_p2d(const float* dEh, const float* dEv, const float* w0_packed, const float* w0T_packed, float* dx, int B, int A, int h, int w, hipStream_t st) {
  const int nepi = B * A * h * w;
  for (int vert = 0; vert < 2; ++vert) {
    const float* dE = vert ? dEv : dEh;
    LfsrGemm r{};
    r.in_mode = vert ? LFSR_IN_LINE_V : LFSR_IN_LINE_H; r.out_mode = vert ? LFSR_OUT_EPIV : LFSR_OUT_EPIH; r.cin = 32; r.X = dE; r.x_stride = 32;
    r.Wp = w0T_packed; r.Y = dx; r.y_stride = 64; r.R1 = dx; r.r1_stride = 64;
    r.M = nepi; r.N = A * 64; r.A = A; r.h = h; r.w = w; r.ntaps = A; r.CH = 64;
    // EPIConv.0 data gradient (accumulates into dx): EPI-line kernel where it applies, else the gather-GEMM
    int rc3 = lfsr_epi0_dgrad_launch(dE, w0_packed, dx, 64, 0, B, A, h, w, vert, st);
    if (rc3 == LFSR_E_ARG) rc3 = lfsr_bwd_gemm(r, st);
    if (rc3) return rc3;
  }
  return LFSR_OK;
}

int lfsr_epi_branch_bwd_p2w(const float* dEh, const float* dEv, const float* xin, float* dw0, float* const P[4], int B, int A, int h, int w, hipStream_t st) {
  const int AA = A * A, nepi = B * A * h * w;
  const long long npix = (long long)B * AA * h * w;
  int rc;
#define RC(call) do { rc = (call); if (rc) return rc; } while (0)
  // the two passes share EPIConv.0's weights: where the EPI-line kernel applies, ONE weight-gradient launch covers both (one slab per block instead of
  // two sets); else the gather form per pass
  const char* wsel = lfsr_sel("LFSR_WGRAD_EPI");
  const bool epi_merged = lfsr_wgrad_epi0_blocks(B, A, h, w, 2) > 0 && A == 5 && h <= 32 && w <= 32 && npix * 64 * 4 < (1LL << 31) && !(wsel && wsel[0] == 'g');
  int epi_slabs[2] = {0, 0};
  if (epi_merged) {
    RC(lfsr_wgrad_epi0_launch(dEh, dEv, xin, 64, 0, P[1], B, A, h, w, 2, st));
    epi_slabs[0] = lfsr_wgrad_epi0_blocks(B, A, h, w, 2);
  } else {
    for (int vert = 0; vert < 2; ++vert) {
      RC(lfsr_wgrad_launch(LFSR_IN_SAME, vert ? LFSR_IN_EPIV : LFSR_IN_EPIH, vert ? dEv : dEh, 32, 0, xin, 64, 0, P[vert ? 3 : 1], nepi, 32, 64, A, h, w, AA, st));
      epi_slabs[vert] = lfsr_wgrad_splits(nepi, AA, 64);
    }
  }
  RC(lfsr_wgrad_reduce(P[1], epi_slabs[0], epi_slabs[1] ? P[3] : nullptr, epi_slabs[1], dw0, 32, 64, AA, 0, 0, 0, 0, 0, st));
#undef RC
  return LFSR_OK;
}

int lfsr_epi_branch_bwd(const float* dcat, int dc_stride, int choff_h, int choff_v, const float* xin, const float* eh, const float* ev,
                        const float* w0_packed, const float* w0T_packed, const float* w2T_packed, float* dx, float* dw0, float* dw2,
                        float* dEh, float* dEv, float* const P[4], int B, int A, int h, int w, float slope, hipStream_t st) {
  int rc = lfsr_epi_branch_bwd_p1(dcat, dc_stride, choff_h, choff_v, eh, ev, w2T_packed, dw2, dEh, dEv, P, B, A, h, w, slope, st);
  if (!rc) rc = lfsr_epi_branch_bwd_p2d(dEh, dEv, w0_packed, w0T_packed, dx, B, A, h, w, st);
  if (!rc) rc = lfsr_epi_branch_bwd_p2w(dEh, dEv, xin, dw0, P, B, A, h, w, st);
  return rc;
}

size_t lfsr_branch_bwd_partial_floats(int B, int A, int h, int w) {
  const int AA = A * A;
  const int nlr = B * h * w, nepi = B * A * h * w;
  size_t m = 0;
  auto up = [&](size_t v) { if (v > m) m = v; };
  up((size_t)256 * AA * 32 * 64);   // EPI-line weight gradient: one slab per block
  up(lfsr_wgrad_partial_floats(nlr, AA, 16, 64));
  up(lfsr_wgrad_partial_floats(nlr, AA, 16, 16));
  up(lfsr_wgrad_partial_floats(nepi, AA, 32, 64));
  up(lfsr_wgrad_partial_floats(nepi, A, 32, 32));
  return m;
}

namespace {
struct AngWs { float *w0p, *w0T, *w2T, *dA16, *P, *dym; size_t total; };
void ang_layout(int B, int A, int h, int w, float* base, AngWs& t) {
  const int AA = A * A;
  size_t o = 0;
  auto take = [&](size_t f) { float* p = base ? base + o : nullptr; o += al64(f); return p; };
  t.w0p = take(lfsr_packed_weight_floats(16, 64, AA));
  t.w0T = take((size_t)AA * pad32(64) * 16);
  t.w2T = take((size_t)AA * pad32(16) * 16);
  t.dA16 = take((size_t)B * h * w * 16);
  t.P = take(lfsr_branch_bwd_partial_floats(B, A, h, w));
  t.dym = take((size_t)B * AA * h * w * 16);
  t.total = o;
}
struct EpiWs { float *w0p, *w0T, *w2T, *dEh, *dEv, *P[4], *dym; size_t total; };
void epi_layout(int B, int A, int h, int w, float* base, EpiWs& t) {
  const int AA = A * A;
  size_t o = 0;
  auto take = [&](size_t f) { float* p = base ? base + o : nullptr; o += al64(f); return p; };
  t.w0p = take(lfsr_packed_weight_floats(32, 64, AA));
  t.w0T = take((size_t)AA * pad32(64) * 32);
  t.w2T = take((size_t)A * pad32(32) * 32);
  t.dEh = take((size_t)B * A * h * w * 32);
  t.dEv = take((size_t)B * A * h * w * 32);
  for (int i = 0; i < 4; ++i) t.P[i] = take(lfsr_branch_bwd_partial_floats(B, A, h, w));
  t.dym = take((size_t)B * AA * h * w * 64);
  t.total = o;
}
}  // namespace

extern "C" {

size_t lfsr_angconv_bwd_workspace_floats(int B, int A, int h, int w) {
  if (B <= 0 || A <= 0 || A > 15 || h <= 0 || w <= 0) return 0;
  AngWs t;
  ang_layout(B, A, h, w, nullptr, t);
  return t.total;
}

int lfsr_angconv_bwd(const float* dy, int dy_stride, int dy_choff, const float* y, int y_stride, int y_choff, const float* x, const float* a16, const float* w0, const float* w2,
                     float* dx, float* dw0, float* dw2, float* workspace, size_t workspace_floats, int B, int A, int h, int w, float slope, void* stream) {
  if (!dy || !x || !a16 || !w0 || !w2 || !dx || !dw0 || !dw2 || !workspace || B <= 0 || A <= 0 || A > 15 || !(A & 1) || h <= 0 || w <= 0) return LFSR_E_ARG;
  if (dy_stride < dy_choff + 16 || (y && y_stride < y_choff + 16) || ((uintptr_t)workspace & 15)) return LFSR_E_ARG;
  if ((long long)B * A * A * h * w >= (1LL << 31) / 144) return LFSR_E_ARG;
  AngWs t;
  ang_layout(B, A, h, w, workspace, t);
  if (workspace_floats < t.total) return LFSR_E_WS;
  hipStream_t st = lfsr_stream(stream);
  const int AA = A * A;
  int rc = lfsr_pack_conv_weight_m(w0, t.w0p, 16, 64, AA, 0, 0, 0, stream);
  if (!rc) rc = lfsr_pack_weight_T(w0, t.w0T, 16, 64, AA, 0, st);
  if (!rc) rc = lfsr_pack_weight_chunkT(w2, t.w2T, 16 * AA, 16, 16, 1, st);
  if (rc) return rc;
  if (y) {   // dLoss/dy -> the gradient at the stage-2 pre-activation (inside the model the fuse.0 data gradient's LeakyReLU' epilogue has done this already)
    rc = mask_lrelu(dy, dy_stride, dy_choff, y, y_stride, y_choff, t.dym, 16, 0, (long long)B * AA * h * w, 16, slope, st);
    if (rc) return rc;
    dy = t.dym; dy_stride = 16; dy_choff = 0;
  }
  return lfsr_ang_branch_bwd(dy, dy_stride, dy_choff, x, a16, t.w0p, t.w0T, t.w2T, dx, dw0, dw2, t.dA16, t.P, B, A, h, w, slope, st);
}

size_t lfsr_epiconv_hv_bwd_workspace_floats(int B, int A, int h, int w) {
  if (B <= 0 || A <= 0 || A > 15 || h <= 0 || w <= 0) return 0;
  EpiWs t;
  epi_layout(B, A, h, w, nullptr, t);
  return t.total;
}

int lfsr_epiconv_hv_bwd(const float* dy, int dy_stride, int choff_h, int choff_v, const float* y, int y_stride, int y_choff_h, int y_choff_v,
                        const float* x, const float* e_h, const float* e_v, const float* w0, const float* w2,
                        float* dx, float* dw0, float* dw2, float* workspace, size_t workspace_floats, int B, int A, int h, int w, float slope, void* stream) {
  if (!dy || !x || !e_h || !e_v || !w0 || !w2 || !dx || !dw0 || !dw2 || !workspace || B <= 0 || A <= 0 || A > 15 || !(A & 1) || h <= 0 || w <= 0) return LFSR_E_ARG;
  if (dy_stride < choff_h + 32 || dy_stride < choff_v + 32 || (y && (y_stride < y_choff_h + 32 || y_stride < y_choff_v + 32)) || ((uintptr_t)workspace & 15)) return LFSR_E_ARG;
  if ((long long)B * A * A * h * w >= (1LL << 31) / 144) return LFSR_E_ARG;
  EpiWs t;
  epi_layout(B, A, h, w, workspace, t);
  if (workspace_floats < t.total) return LFSR_E_WS;
  hipStream_t st = lfsr_stream(stream);
  const int AA = A * A;
  int rc = lfsr_pack_conv_weight_m(w0, t.w0p, 32, 64, AA, 0, 0, 0, stream);
  if (!rc) rc = lfsr_pack_weight_T(w0, t.w0T, 32, 64, AA, 0, st);
  if (!rc) rc = lfsr_pack_weight_chunkT(w2, t.w2T, 32 * A, 32, 32, 0, st);
  if (rc) return rc;
  if (y) {
    const long long npix = (long long)B * AA * h * w;
    rc = mask_lrelu(dy, dy_stride, choff_h, y, y_stride, y_choff_h, t.dym, 64, 0, npix, 32, slope, st);
    if (!rc) rc = mask_lrelu(dy, dy_stride, choff_v, y, y_stride, y_choff_v, t.dym, 64, 32, npix, 32, slope, st);
    if (rc) return rc;
    dy = t.dym; dy_stride = 64; choff_h = 0; choff_v = 32;
  }
  return lfsr_epi_branch_bwd(dy, dy_stride, choff_h, choff_v, x, e_h, e_v, t.w0p, t.w0T, t.w2T, dx, dw0, dw2, t.dEh, t.dEv, t.P, B, A, h, w, slope, st);
}

}  // extern "C"
