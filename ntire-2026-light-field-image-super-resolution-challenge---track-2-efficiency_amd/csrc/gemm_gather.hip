// Gather-GEMM on fp32 MFMA (v_mfma_f32_32x32x2_f32, exact fp32) for every contraction of DistgSSR
// (model/SR/DistgSSR.py:78-111) on the view-major channel-last (VCL) layout.
//
// Every conv of the network is   Y[m, n] = act( sum_tap sum_k  X[src(m, tap), k] * W[tap][n][k] )
// where a "row" m is an output position, src(m, tap) is a pixel index (or "zero padding") given by the
// operator class, and X[pixel, :] is one contiguous channel vector.  So one kernel serves them all:
//
//   IN_SAME   1x1 conv (fuse.0 :99, AngConv.2 :87, EPIConv.2 :94)          src = m
//   IN_CONV3  per-view 3x3 zero-pad-1 (== MacPI "k3 dil A pad A" :22,47,64,79-83,101)
//   IN_ANG    AngConv.0 (:85) AxA stride A on MacPI: rows (b,y,x), taps = the A*A views
//   IN_EPIH   EPIConv.0 (:92) 1xA^2 stride (1,A) pad A(A-1)/2 on MacPI: rows (b,u,y,x)
//   IN_EPIV   the same conv applied to the transposed MacPI (:108): rows (b,v,y,x)
//
// and the PixelShuffle / PixelShuffle1D / concat that follow each branch are folded into where the
// epilogue stores (OUT_* modes), so no rearrange or cat ever touches HBM.
//
// Tiling: 256 threads = 4 waves; block tile BM=128 rows x BN=32*NT cols; wave w owns rows 32w..32w+31
// and all NT 32-wide column tiles (NT x 16 accumulator VGPRs).  K is walked tap by tap in 64-float
// stages: each stage's A slab (128 x 64) and W slab (BN x 64) are fetched with 16-B loads into
// registers while the previous stage computes, then written to a padded LDS image (row stride 68
// floats: ds_read_b128 conflict-free).  Each lane reads 4 consecutive k with one ds_read_b128; lane half
// h takes k = 8j+4h..8j+4h+3, which is a K permutation applied to A and W alike, so 4 MFMAs consume it.
// LDS: (128 + BN) * 272 B  = 52 KB at NT=2  ->  3 blocks/CU, which is what hides the two barriers per
// stage.  MFMA-bound by design: 128 x 64 x 64 x 2 flop per stage against 48 KB of L2->LDS traffic.
#include <stdlib.h>

#include "gemm_gather_kernel.h"

#include "lfsr_internal.h"

namespace {

// ---- weight packing --------------------------------------------------------------------------------
// in (O, C, T) -> out [T][Npad][C];  perm 1: reference channel c*r2 + q  ->  n' = q*ch + c
__global__ __launch_bounds__(256) void k_pack_weight(const float* __restrict__ w, float* __restrict__ out, int O, int C, int T, int Npad, int perm, int ch) {
  const long long total = (long long)T * Npad * C;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    int c = (int)(i % C);
    long long t2 = i / C;
    int n = (int)(t2 % Npad);
    int t = (int)(t2 / Npad);
    float v = 0.f;
    if (n < O) {
      int nref = n;
      if (perm == 1) { int r2 = O / ch; int q = n / ch, cc = n - q * ch; nref = cc * r2 + q; }
      v = w[((long long)nref * C + c) * T + t];
    }
    out[i] = v;
  }
}

// ---- init_conv (Cin = 1) fused with SAI2MacPI -----------------------------------------------------
// 16 threads per output pixel, 4 output channels each -> one 256-B coalesced store per pixel.
// IDX: the type the (b, view, y, x) decode runs in: unsigned when the launch has < 2^31 work items (one 32-bit division sequence per axis instead of a 64-bit one:
// the round-3 kernel spent more instructions on its four 64-bit div / mod than on its 36 FMAs), long long otherwise.  Addresses stay 64-bit.
template <typename IDX>
__global__ __launch_bounds__(256) void k_initconv(const float* __restrict__ x, const float* __restrict__ w, float* __restrict__ y, int y_stride, int y_choff,
                                                  int B, int A, int h, int wd) {
  const long long npix = (long long)B * A * A * h * wd;
  const int Wm = A * wd, Hm = A * h;
  // a thread keeps ITS four channels for every pixel it visits (the grid stride is a multiple of 16), so their 36 weights live in registers: no LDS read per FMA
  // (round 3: 75 -> see DESIGN; the first form re-read the weights from LDS for every pixel)
  const int c4 = (int)(threadIdx.x & 15) * 4;
  float wr[4][9];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int k = 0; k < 9; ++k) wr[i][k] = w[(c4 + i) * 9 + k];
  for (IDX g = (IDX)blockIdx.x * 256 + threadIdx.x; g < (IDX)(npix * 16); g += (IDX)gridDim.x * 256) {
    const IDX pixi = g >> 4;
    const long long pix = (long long)pixi;
    int xx = (int)(pixi % (IDX)wd);
    IDX t = pixi / (IDX)wd;
    int yy = (int)(t % (IDX)h);
    t /= (IDX)h;
    int view = (int)(t % (IDX)(A * A));
    int b = (int)(t / (IDX)(A * A));
    int u = view / A, v = view - u * A;
    const float* img = x + (long long)b * Hm * Wm + (long long)(u * h) * Wm + v * wd;  // this view's top-left in the mosaic
    float xv[9];
#pragma unroll
    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) {
        int sy = yy + ky - 1, sx = xx + kx - 1;
        xv[ky * 3 + kx] = (sy >= 0 && sy < h && sx >= 0 && sx < wd) ? img[(long long)sy * Wm + sx] : 0.f;
      }
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
#pragma unroll
    for (int k = 0; k < 9; ++k) {
      a0 = fmaf(xv[k], wr[0][k], a0);
      a1 = fmaf(xv[k], wr[1][k], a1);
      a2 = fmaf(xv[k], wr[2][k], a2);
      a3 = fmaf(xv[k], wr[3][k], a3);
    }
    *reinterpret_cast<float4*>(y + pix * y_stride + y_choff + c4) = make_float4(a0, a1, a2, a3);
  }
}

// ---- upsample head ----------------------------------------------------------------------------------
// fold: wf[ij][k] = sum_c w2[c] * w0[c*s2+ij][k];  bf[ij] = sum_c w2[c] * b0[c*s2+ij]   (double accumulation)
__global__ void k_fold_head(const float* __restrict__ w0, const float* __restrict__ b0, const float* __restrict__ w2, float* __restrict__ wf, float* __restrict__ bf, int C, int s2) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < s2 * C) {
    int ij = i / C, k = i - ij * C;
    double a = 0.0;
    for (int c = 0; c < C; ++c) a += (double)w2[c] * (double)w0[((long long)c * s2 + ij) * C + k];
    wf[i] = (float)a;
  }
  if (i < s2) {
    double a = 0.0;
    for (int c = 0; c < C; ++c) a += (double)w2[c] * (double)(b0 ? b0[c * s2 + i] : 0.f);
    bf[i] = (float)a;
  }
}

// s = 4, w % 4 == 0: SIXTEEN LANES PER LR PIXEL, one output each.  A wave stages 16 consecutive pixels (4 KB, one coalesced 16-B load per lane and group of four) in a
// wave-private LDS tile; lane (pixel p, output o = 4 i + j) then reads its pixel's 64 channels as sixteen ds_read_b128 (the 16 lanes of a pixel read the same address: a
// broadcast) against the 64 weights of ITS output row, which it keeps in registers -- the same fma chain over k = 0..63 as the one-thread-per-(pixel, sub-row) form below,
// so the same bits -- adds bias and the bilinear skip and stores one dword (16 B contiguous per pixel and sub-row).  The older form reads 16 B per lane at a 256-B lane
// stride, every pixel four times (105 us for 265 MB); a first 16-lane form that split the CHANNELS over the lanes and summed 16 x 16 partials through LDS was LDS-bound (95 us).
__global__ __launch_bounds__(256) void k_head4_lanes(const float* __restrict__ f, int f_stride, int f_choff, const float* __restrict__ wf, const float* __restrict__ bf,
                                                     const float* __restrict__ xlr, float* __restrict__ out, int B, int A, int h, int w) {
  constexpr int U = 4;                                                       // groups of four pixels per wave and pass
  __shared__ __attribute__((aligned(16))) float st[4][U * 4 * 64];          // [wave][group][pixel][channel]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, pq = lane >> 4, l = lane & 15;
  const int Hm = A * h, Wm = A * w, Ho = Hm * 4, Wo = Wm * 4;
  const unsigned npix = (unsigned)B * A * A * h * w;
  float4 wreg[16];        // the 64 weights of this lane's output row o = l
#pragma unroll
  for (int k = 0; k < 16; ++k) wreg[k] = *reinterpret_cast<const float4*>(wf + l * 64 + 4 * k);
  const float bias = bf[l];
  float* tile = st[wave];
  const int i = l >> 2, j = l & 3;
  for (unsigned base0 = (blockIdx.x * 4u + wave) * (4u * U); base0 < npix; base0 += gridDim.x * (16u * U)) {
    float4 fv[U];
#pragma unroll
    for (int g = 0; g < U; ++g) {
      const unsigned pix = base0 + 4u * g + pq;
      fv[g] = pix < npix ? *reinterpret_cast<const float4*>(f + (size_t)pix * f_stride + f_choff + 4 * l) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int g = 0; g < U; ++g) *reinterpret_cast<float4*>(tile + (g * 4 + pq) * 64 + 4 * l) = fv[g];
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int g = 0; g < U; ++g) {
      const unsigned base = base0 + 4u * g;           // four consecutive pixels (w % 4 == 0: one image row)
      if (base >= npix) break;
      const float* fp = tile + (g * 4 + pq) * 64;
      float acc = 0.f;
#pragma unroll
      for (int k = 0; k < 16; ++k) {
        const float4 q = *reinterpret_cast<const float4*>(fp + 4 * k);
        acc = fmaf(q.x, wreg[k].x, acc); acc = fmaf(q.y, wreg[k].y, acc); acc = fmaf(q.z, wreg[k].z, acc); acc = fmaf(q.w, wreg[k].w, acc);
      }
      // (b, view, y, x) of the group's first pixel: wave-uniform
      const unsigned x0 = base % (unsigned)w; unsigned t = base / (unsigned)w;
      const unsigned y = t % (unsigned)h; t /= (unsigned)h;
      const unsigned view = t % (unsigned)(A * A), b = t / (unsigned)(A * A);
      const int u = (int)view / A, v = (int)view - u * A, x = (int)x0 + pq;
      // bilinear skip on the whole mosaic (F.interpolate align_corners=False, DistgSSR.py:30)
      const int Y = (u * h + (int)y) * 4 + i, X = (v * w + x) * 4 + j;
      float sy = fmaxf(((float)Y + 0.5f) * 0.25f - 0.5f, 0.f);
      int y0 = (int)sy; if (y0 > Hm - 1) y0 = Hm - 1;
      const int y1 = y0 + 1 < Hm ? y0 + 1 : Hm - 1;
      const float ly1 = sy - (float)y0, ly0 = 1.f - ly1;
      float sx = fmaxf(((float)X + 0.5f) * 0.25f - 0.5f, 0.f);
      int xx0 = (int)sx; if (xx0 > Wm - 1) xx0 = Wm - 1;
      const int xx1 = xx0 + 1 < Wm ? xx0 + 1 : Wm - 1;
      const float lx1 = sx - (float)xx0, lx0 = 1.f - lx1;
      const float* xb = xlr + (size_t)b * Hm * Wm;
      const float up = ly0 * (lx0 * xb[(size_t)y0 * Wm + xx0] + lx1 * xb[(size_t)y0 * Wm + xx1]) + ly1 * (lx0 * xb[(size_t)y1 * Wm + xx0] + lx1 * xb[(size_t)y1 * Wm + xx1]);
      out[((size_t)b * Ho + Y) * Wo + X] = acc + bias + up;
    }
    __builtin_amdgcn_wave_barrier();
  }
}

// one thread per (LR pixel, sub-row i): s outputs along j, stored contiguously in the HR mosaic
template <int S, typename IDX>      // IDX: as in k_initconv
__global__ __launch_bounds__(256) void k_head(const float* __restrict__ f, int f_stride, int f_choff, const float* __restrict__ wf, const float* __restrict__ bf,
                                             const float* __restrict__ xlr, float* __restrict__ out, int B, int A, int h, int w) {
  __shared__ float sw[S * S * 64];
  __shared__ float sb[S * S];
  for (int i = threadIdx.x; i < S * S * 64; i += 256) sw[i] = wf[i];
  if (threadIdx.x < S * S) sb[threadIdx.x] = bf[threadIdx.x];
  __syncthreads();
  const int Hm = A * h, Wm = A * w, Ho = Hm * S, Wo = Wm * S;
  const long long total = (long long)B * A * A * h * S * w;   // (b, view, y, i, x)
  const float rs = 1.0f / (float)S;
  for (IDX g = (IDX)blockIdx.x * 256 + threadIdx.x; g < (IDX)total; g += (IDX)gridDim.x * 256) {
    int x = (int)(g % (IDX)w);
    IDX t = g / (IDX)w;
    int i = (int)(t % (IDX)S);
    t /= (IDX)S;
    int y = (int)(t % (IDX)h);
    t /= (IDX)h;
    int view = (int)(t % (IDX)(A * A));
    int b = (int)(t / (IDX)(A * A));
    int u = view / A, v = view - u * A;
    long long pix = (((long long)b * A * A + view) * h + y) * w + x;
    const float4* fp = reinterpret_cast<const float4*>(f + pix * f_stride + f_choff);
    float acc[S];
#pragma unroll
    for (int j = 0; j < S; ++j) acc[j] = 0.f;
#pragma unroll 4
    for (int k4 = 0; k4 < 16; ++k4) {
      float4 fv = fp[k4];
#pragma unroll
      for (int j = 0; j < S; ++j) {
        const float* wr = sw + (i * S + j) * 64 + k4 * 4;
        acc[j] = fmaf(fv.x, wr[0], acc[j]);
        acc[j] = fmaf(fv.y, wr[1], acc[j]);
        acc[j] = fmaf(fv.z, wr[2], acc[j]);
        acc[j] = fmaf(fv.w, wr[3], acc[j]);
      }
    }
    // bilinear skip on the whole mosaic (F.interpolate align_corners=False, DistgSSR.py:30)
    const int Y = (u * h + y) * S + i;
    float sy = fmaxf(((float)Y + 0.5f) * rs - 0.5f, 0.f);
    int y0 = (int)sy; if (y0 > Hm - 1) y0 = Hm - 1;
    int y1 = y0 + 1 < Hm ? y0 + 1 : Hm - 1;
    float ly1 = sy - (float)y0, ly0 = 1.f - ly1;
    const float* xb = xlr + (long long)b * Hm * Wm;
    float* ob = out + ((long long)b * Ho + Y) * Wo;
#pragma unroll
    for (int j = 0; j < S; ++j) {
      const int X = (v * w + x) * S + j;
      float sx = fmaxf(((float)X + 0.5f) * rs - 0.5f, 0.f);
      int x0 = (int)sx; if (x0 > Wm - 1) x0 = Wm - 1;
      int x1 = x0 + 1 < Wm ? x0 + 1 : Wm - 1;
      float lx1 = sx - (float)x0, lx0 = 1.f - lx1;
      float up = ly0 * (lx0 * xb[(long long)y0 * Wm + x0] + lx1 * xb[(long long)y0 * Wm + x1]) +
                 ly1 * (lx0 * xb[(long long)y1 * Wm + x0] + lx1 * xb[(long long)y1 * Wm + x1]);
      ob[X] = acc[j] + sb[i * S + j] + up;
    }
  }
}

}  // namespace

extern "C" {

size_t lfsr_packed_weight_floats(int O, int C, int taps) {
  size_t f = (size_t)taps * (size_t)npad32(O) * (size_t)C;
  if (O == 64 && C == 64 && taps == 9) f += LFSR_CONV3_WINO_FLOATS;   // 3x3 64->64: the Winograd-domain copy follows the direct pack
  if (O == 32 && C == 64 && taps == 25) f += LFSR_EPI_WINO_FLOATS + LFSR_EPI_B3_W1_FLOATS;    // EPIConv.0 at angRes 5: its F(2,5) copy and its three bf16 planes (epi_b3.hip) follow the direct pack
  if (O == 160 && C == 32 && taps == 1) f += LFSR_EPI_B3_W2_FLOATS;   // EPIConv.2 at angRes 5: its three bf16 planes follow the direct pack
  return f;
}

int lfsr_pack_conv_weight(const float* w, float* packed, int O, int C, int taps, int perm, int ch, void* stream) {
  return lfsr_pack_conv_weight_m(w, packed, O, C, taps, perm, ch, LFSR_W_ALL, stream);
}

}  // extern "C"

int lfsr_pack_conv_weight_m(const float* w, float* packed, int O, int C, int taps, int perm, int ch, int mask, void* stream) {
  if (!w || !packed || O <= 0 || C <= 0 || taps <= 0 || (perm != 0 && perm != 1)) return LFSR_E_ARG;
  if (perm == 1 && (ch <= 0 || O % ch != 0)) return LFSR_E_ARG;
  if (O == 64 && C == 64 && taps == 9 && perm == 0 && mask == LFSR_W_WINO4)   // the runtimes' lean repack: one launch per weight
    return lfsr_pack_conv3_raw_wino4(w, packed, packed + LFSR_CONV3_DIRECT_FLOATS + LFSR_CONV3_WINO2_FLOATS, 0, lfsr_stream(stream));
  long long total = (long long)taps * npad32(O) * C;
  hipLaunchKernelGGL(k_pack_weight, dim3(lfsr_blocks(total, 256)), dim3(256), 0, lfsr_stream(stream), w, packed, O, C, taps, npad32(O), perm, ch);
  LFSR_CHECK_LAUNCH();
  if (O == 64 && C == 64 && taps == 9 && perm == 0) return lfsr_pack_wino_m(packed, packed + LFSR_CONV3_DIRECT_FLOATS, mask, lfsr_stream(stream));
  if (O == 32 && C == 64 && taps == 25 && perm == 0) {
    const int rc = lfsr_pack_epi_wino(packed, packed + 25 * 32 * 64, lfsr_stream(stream));
    return rc ? rc : lfsr_pack_epi_b3(packed, packed + 25 * 32 * 64 + LFSR_EPI_WINO_FLOATS, 0, lfsr_stream(stream));
  }
  if (O == 160 && C == 32 && taps == 1 && perm == 0) return lfsr_pack_epi_b3(packed, packed + 160 * 32, 1, lfsr_stream(stream));
  return LFSR_OK;
}

extern "C" {

int lfsr_conv3x3_fwd(const float* x, int x_stride, int x_choff, const float* w_packed, float* y, int y_stride, int y_choff,
                     const float* r1, int r1_stride, int r1_choff, const float* r2, int r2_stride, int r2_choff,
                     int n_img, int h, int w, float slope, void* stream) {
  LfsrOpTimer op_t("conv3x3", n_img, h * w, lfsr_stream(stream));
  if (!x || !w_packed || !y || n_img <= 0 || h <= 0 || w <= 0) return LFSR_E_ARG;
  if (x_stride < x_choff + 64 || y_stride < y_choff + 64 || (x_stride | x_choff) & 3) return LFSR_E_ARG;
  if ((long long)n_img * h * w >= (1LL << 31) / 4) return LFSR_E_ARG;
  {
    // the tile kernels need 16-B aligned channel vectors on every operand; LFSR_CONV3X3 = halo | gather forces the direct
    // 9-tap halo kernel / the v1 gather-GEMM (A/B runs)
    const char* sel = lfsr_conv3_fwd_sel();
    const bool force_v1 = sel && sel[0] == 'g';
    const bool force_halo = sel && sel[0] == 'h';
    const bool al = !((y_stride | y_choff) & 3) && (!r1 || !((r1_stride | r1_choff) & 3)) && (!r2 || !((r2_stride | r2_choff) & 3));
    if (al && !force_v1 && !force_halo) {
      const int rc = lfsr_conv3x3_wino_launch(x, x_stride, x_choff, w_packed + LFSR_CONV3_DIRECT_FLOATS, w_packed, y, y_stride, y_choff, r1, r1_stride, r1_choff,
                                              r2, r2_stride, r2_choff, nullptr, 0, 0, 1.0f, n_img, h, w, slope, sel, lfsr_stream(stream));
      if (rc != LFSR_E_ARG) return rc;   // (E_ARG: a geometry the Winograd launchers do not cover -> the direct kernel)
    }
    if (al && !force_v1)
      return lfsr_conv3x3_halo_launch(x, x_stride, x_choff, w_packed, y, y_stride, y_choff, r1, r1_stride, r1_choff, r2, r2_stride, r2_choff,
                                      nullptr, 0, 0, 1.0f, n_img, h, w, slope, lfsr_stream(stream));
  }
  GemmArgs p{};
  p.X = x; p.x_stride = x_stride; p.x_choff = x_choff; p.Wp = w_packed; p.bias = nullptr;
  p.Y = y; p.y_stride = y_stride; p.y_choff = y_choff;
  p.R1 = r1; p.r1_stride = r1_stride; p.r1_choff = r1_choff; p.R2 = r2; p.r2_stride = r2_stride; p.r2_choff = r2_choff;
  p.M = n_img * h * w; p.N = 64; p.Npad = 64; p.A = 1; p.AA = 1; p.H = h; p.W = w; p.ntaps = 9; p.CH = 64; p.slope = slope;
  return launch_gemm<IN_CONV3, OUT_SAME, 64, 2>(p, lfsr_stream(stream));
}

int lfsr_pointwise_fwd(const float* x, int x_stride, int x_choff, int cin, const float* w_packed, const float* bias,
                       float* y, int y_stride, int y_choff, int M, int N, float slope, void* stream) {
  LfsrOpTimer op_t("pointwise", cin, N, lfsr_stream(stream));
  if (!x || !w_packed || !y || M <= 0 || N <= 0) return LFSR_E_ARG;
  if (x_stride < x_choff + cin || y_stride < y_choff + N || (x_stride | x_choff) & 3) return LFSR_E_ARG;
  GemmArgs p{};
  p.X = x; p.x_stride = x_stride; p.x_choff = x_choff; p.Wp = w_packed; p.bias = bias;
  p.Y = y; p.y_stride = y_stride; p.y_choff = y_choff;
  p.M = M; p.N = N; p.Npad = npad32(N); p.A = 1; p.AA = 1; p.H = 1; p.W = 1; p.ntaps = 1; p.CH = N; p.slope = slope;
  hipStream_t st = lfsr_stream(stream);
  if (M >= 2048 && !lfsr_sel("LFSR_NO_ROWGEMM")) {   // streaming kernel for the shapes it covers (fuse.0: 144 -> 64)
    int rc = lfsr_rowgemm_launch(x, x_stride, x_choff, cin, w_packed, bias, nullptr, 0, 0, y, y_stride, y_choff, M, N, slope, st);
    if (rc != LFSR_E_ARG) return rc;
  }
  const bool two = (p.Npad % 64) == 0;
  switch (cin) {
    case 16: return launch_gemm<IN_SAME, OUT_SAME, 16, 1>(p, st);
    case 32: return launch_gemm<IN_SAME, OUT_SAME, 32, 1>(p, st);
    case 64: return two ? launch_gemm<IN_SAME, OUT_SAME, 64, 2>(p, st) : launch_gemm<IN_SAME, OUT_SAME, 64, 1>(p, st);
    case 144: return two ? launch_gemm<IN_SAME, OUT_SAME, 144, 2>(p, st) : launch_gemm<IN_SAME, OUT_SAME, 144, 1>(p, st);
    default: return LFSR_E_ARG;
  }
}

int lfsr_angconv_fwd(const float* x, int x_stride, int x_choff, const float* w1_packed, const float* w2_packed,
                     float* tmp, float* y, int y_stride, int y_choff, int B, int A, int h, int w, float slope, void* stream) {
  LfsrOpTimer op_t("angconv", B, h * w, lfsr_stream(stream));
  if (!x || !w1_packed || !w2_packed || !tmp || !y || B <= 0 || A <= 0 || h <= 0 || w <= 0) return LFSR_E_ARG;
  if (x_stride < x_choff + 64 || y_stride < y_choff + 16 || (x_stride | x_choff) & 3) return LFSR_E_ARG;
  hipStream_t st = lfsr_stream(stream);
  {
    const char* sel = lfsr_sel("LFSR_ANG");   // LFSR_ANG=gather forces the two-launch gather-GEMM path (A/B runs)
    if (!(sel && sel[0] == 'g') && lfsr_ang_fused_ok(A) && !((y_stride | y_choff) & 3))
      return lfsr_ang_fused_launch(x, x_stride, x_choff, w1_packed, w2_packed, tmp, y, y_stride, y_choff, B, A, h, w, slope, st);
  }
  GemmArgs p{};
  p.X = x; p.x_stride = x_stride; p.x_choff = x_choff; p.Wp = w1_packed;
  p.Y = tmp; p.y_stride = 16; p.y_choff = 0;
  p.M = B * h * w; p.N = 16; p.Npad = 32; p.A = A; p.AA = A * A; p.H = h; p.W = w; p.ntaps = A * A; p.CH = 16; p.slope = slope;
  int rc = launch_gemm<IN_ANG, OUT_SAME, 64, 1>(p, st);
  if (rc) return rc;
  GemmArgs q{};
  q.X = tmp; q.x_stride = 16; q.x_choff = 0; q.Wp = w2_packed;
  q.Y = y; q.y_stride = y_stride; q.y_choff = y_choff;
  q.M = B * h * w; q.N = 16 * A * A; q.Npad = npad32(q.N); q.A = A; q.AA = A * A; q.H = h; q.W = w; q.ntaps = 1; q.CH = 16; q.slope = slope;
  return launch_gemm<IN_SAME, OUT_VIEWS, 16, 1>(q, st);
}

int lfsr_epiconv_gather(const float* x, int x_stride, int x_choff, const float* w1_packed, const float* w2_packed,
                          float* tmp, float* y, int y_stride, int y_choff, int B, int A, int h, int w, int vertical, float slope, hipStream_t st) {
  GemmArgs p{};
  p.X = x; p.x_stride = x_stride; p.x_choff = x_choff; p.Wp = w1_packed;
  p.Y = tmp; p.y_stride = 32; p.y_choff = 0;
  p.M = B * A * h * w; p.N = 32; p.Npad = 32; p.A = A; p.AA = A * A; p.H = h; p.W = w; p.ntaps = A * A; p.CH = 32; p.slope = slope;
  int rc = vertical ? launch_gemm<IN_EPIV, OUT_SAME, 64, 1>(p, st) : launch_gemm<IN_EPIH, OUT_SAME, 64, 1>(p, st);
  if (rc) return rc;
  GemmArgs q{};
  q.X = tmp; q.x_stride = 32; q.x_choff = 0; q.Wp = w2_packed;
  q.Y = y; q.y_stride = y_stride; q.y_choff = y_choff;
  q.M = B * A * h * w; q.N = 32 * A; q.Npad = npad32(q.N); q.A = A; q.AA = A * A; q.H = h; q.W = w; q.ntaps = 1; q.CH = 32; q.slope = slope;
  return vertical ? launch_gemm<IN_SAME, OUT_EPIV, 32, 1>(q, st) : launch_gemm<IN_SAME, OUT_EPIH, 32, 1>(q, st);
}

static bool epi_use_fused(int A, int h, int w) {
  const char* sel = lfsr_sel("LFSR_EPI");   // LFSR_EPI=gather forces the two-launch gather-GEMM path (A/B runs)
  return !(sel && sel[0] == 'g') && lfsr_epi_fused_ok(A, h, w);
}

int lfsr_epiconv_fwd(const float* x, int x_stride, int x_choff, const float* w1_packed, const float* w2_packed,
                     float* tmp, float* y, int y_stride, int y_choff, int B, int A, int h, int w, int vertical, float slope, void* stream) {
  LfsrOpTimer op_t("epiconv", B, h * w, lfsr_stream(stream));
  if (!x || !w1_packed || !w2_packed || !y || B <= 0 || A <= 0 || h <= 0 || w <= 0) return LFSR_E_ARG;
  if (x_stride < x_choff + 64 || y_stride < y_choff + 32 || (x_stride | x_choff) & 3) return LFSR_E_ARG;
  if (epi_use_fused(A, h, w) && !((y_stride | y_choff) & 3) && (long long)B * A * A * h * w * x_stride * 4 < (1LL << 31))   // (16-B output vectors, 32-bit offsets)
    return lfsr_epi_fused_launch(x, x_stride, x_choff, w1_packed, w2_packed, y, y_stride, y_choff, y_choff, vertical ? nullptr : tmp, vertical ? tmp : nullptr, B, A, h, w,
                                 vertical ? 2 : 1, slope, lfsr_stream(stream));     // tmp (may be NULL here): receives the pass's stage-1 activation, as the gather path leaves it
  if (!tmp) return LFSR_E_ARG;
  return lfsr_epiconv_gather(x, x_stride, x_choff, w1_packed, w2_packed, tmp, y, y_stride, y_choff, B, A, h, w, vertical, slope, lfsr_stream(stream));
}

int lfsr_epiconv_hv_fwd(const float* x, int x_stride, int x_choff, const float* w1_packed, const float* w2_packed,
                        float* tmp, float* y, int y_stride, int choff_h, int choff_v, int B, int A, int h, int w, float slope, void* stream) {
  LfsrOpTimer op_t("epiconv_hv", B, h * w, lfsr_stream(stream));
  if (!x || !w1_packed || !w2_packed || !y || B <= 0 || A <= 0 || h <= 0 || w <= 0) return LFSR_E_ARG;
  if (x_stride < x_choff + 64 || y_stride < choff_h + 32 || y_stride < choff_v + 32 || (x_stride | x_choff) & 3) return LFSR_E_ARG;
  if (epi_use_fused(A, h, w) && !((y_stride | choff_h | choff_v) & 3) && (long long)B * A * A * h * w * x_stride * 4 < (1LL << 31))
    return lfsr_epi_fused_launch(x, x_stride, x_choff, w1_packed, w2_packed, y, y_stride, choff_h, choff_v, nullptr, nullptr, B, A, h, w, 3, slope, lfsr_stream(stream));
  if (!tmp) return LFSR_E_ARG;
  int rc = lfsr_epiconv_gather(x, x_stride, x_choff, w1_packed, w2_packed, tmp, y, y_stride, choff_h, B, A, h, w, 0, slope, lfsr_stream(stream));
  if (rc) return rc;
  return lfsr_epiconv_gather(x, x_stride, x_choff, w1_packed, w2_packed, tmp, y, y_stride, choff_v, B, A, h, w, 1, slope, lfsr_stream(stream));
}

int lfsr_initconv_fwd(const float* x, const float* w, float* y, int y_stride, int y_choff, int B, int A, int h, int wd, void* stream) {
  if (!x || !w || !y || B <= 0 || A <= 0 || h <= 0 || wd <= 0 || y_stride < y_choff + 64 || (y_stride | y_choff) & 3) return LFSR_E_ARG;
  long long total = (long long)B * A * A * h * wd * 16;
  unsigned grid = lfsr_blocks(total, 256);
  if (grid > 256u * 16) grid = 256u * 16;
  if (total < (1LL << 31)) hipLaunchKernelGGL(k_initconv<unsigned>, dim3(grid), dim3(256), 0, lfsr_stream(stream), x, w, y, y_stride, y_choff, B, A, h, wd);
  else hipLaunchKernelGGL(k_initconv<long long>, dim3(grid), dim3(256), 0, lfsr_stream(stream), x, w, y, y_stride, y_choff, B, A, h, wd);
  LFSR_CHECK_LAUNCH();
  return LFSR_OK;
}

int lfsr_fold_head(const float* w0, const float* b0, const float* w2, float* wf, float* bf, int C, int s, void* stream) {
  if (!w0 || !w2 || !wf || !bf || C != 64 || s <= 0) return LFSR_E_ARG;
  int n = s * s * C;
  hipLaunchKernelGGL(k_fold_head, dim3((n + 255) / 256), dim3(256), 0, lfsr_stream(stream), w0, b0, w2, wf, bf, C, s * s);
  LFSR_CHECK_LAUNCH();
  return LFSR_OK;
}

int lfsr_upsample_head_fwd(const float* f, int f_stride, int f_choff, const float* wf, const float* bf, const float* x_lr, float* out,
                           int B, int A, int h, int w, int s, void* stream) {
  if (!f || !wf || !bf || !x_lr || !out || B <= 0 || A <= 0 || h <= 0 || w <= 0) return LFSR_E_ARG;
  if (f_stride < f_choff + 64 || (f_stride | f_choff) & 3) return LFSR_E_ARG;
  long long total = (long long)B * A * A * h * s * w;
  unsigned grid = lfsr_blocks(total, 256);
  if (grid > 256u * 16) grid = 256u * 16;
  hipStream_t st = lfsr_stream(stream);
  const bool i32 = total < (1LL << 31);
  if (s == 4 && w % 4 == 0 && i32 && !(((uintptr_t)f | (uintptr_t)wf) & 15)) {      // the x4 model: sixteen lanes per pixel
    const long long npix = (long long)B * A * A * h * w;
    unsigned g4 = lfsr_blocks(npix, 64);
    if (g4 > 256u * 8) g4 = 256u * 8;
    hipLaunchKernelGGL(k_head4_lanes, dim3(g4), dim3(256), 0, st, f, f_stride, f_choff, wf, bf, x_lr, out, B, A, h, w);
    LFSR_CHECK_LAUNCH();
    return LFSR_OK;
  }
#define HEAD_GO(SS) do { if (i32) hipLaunchKernelGGL((k_head<SS, unsigned>), dim3(grid), dim3(256), 0, st, f, f_stride, f_choff, wf, bf, x_lr, out, B, A, h, w); \
                         else hipLaunchKernelGGL((k_head<SS, long long>), dim3(grid), dim3(256), 0, st, f, f_stride, f_choff, wf, bf, x_lr, out, B, A, h, w); } while (0)
  switch (s) {
    case 2: HEAD_GO(2); break;
    case 3: HEAD_GO(3); break;
    case 4: HEAD_GO(4); break;
    default: return LFSR_E_ARG;
  }
#undef HEAD_GO
  LFSR_CHECK_LAUNCH();
  return LFSR_OK;
}

}  // extern "C"
