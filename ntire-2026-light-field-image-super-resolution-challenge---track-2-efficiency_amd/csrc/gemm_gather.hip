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

#include "lfsr_common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

// conv3x3_halo.hip
int lfsr_conv3x3_halo_launch(const float* x, int x_stride, int x_choff, const float* w_packed, float* y, int y_stride, int y_choff,
                             const float* r1, int r1_stride, int r1_choff, const float* r2, int r2_stride, int r2_choff,
                             int n_img, int h, int w, float slope, hipStream_t st);

// epi_fused.hip
bool lfsr_epi_fused_ok(int A, int h, int w);
int lfsr_epi_fused_launch(const float* x, int x_stride, int x_choff, const float* w1_packed, const float* w2_packed, float* y, int y_stride,
                          int choffH, int choffV, int B, int A, int h, int w, int which, float slope, hipStream_t st);

namespace {

enum { IN_SAME = 0, IN_CONV3 = 1, IN_ANG = 2, IN_EPIH = 3, IN_EPIV = 4 };
enum { OUT_SAME = 0, OUT_VIEWS = 1, OUT_EPIH = 2, OUT_EPIV = 3 };

struct GemmArgs {
  const float* X; int x_stride; int x_choff;
  const float* Wp;                 // [ntaps][Npad][CIN]
  const float* bias;               // [>=N] or null
  float* Y; int y_stride; int y_choff;
  const float* R1; int r1_stride; int r1_choff;
  const float* R2; int r2_stride; int r2_choff;
  int M, N, Npad;
  int A, AA, H, W;                 // angular res, A*A, view height/width
  int ntaps;
  int CH;                          // channels per destination pixel for OUT_VIEWS/OUT_EPI* (N = chunks*CH)
  float slope;                     // LeakyReLU slope; 1.0f = identity
  int nblk_m;                      // number of row blocks (for the XCD remap)
};

constexpr int BM = 128;
constexpr int LDS_ROW = 68;  // floats; 272 B = 17 x 16 B -> 16 consecutive rows hit 16 distinct 16-B slots

template <int IN>
struct RowInfo {  // what a loading thread keeps per A-slab row it owns
  int base;       // tap-independent part of the source pixel index; -1 = row beyond M
  int y, x;
};

template <int IN>
__device__ __forceinline__ RowInfo<IN> decode_row(int m, const GemmArgs& p) {
  RowInfo<IN> r;
  r.y = 0; r.x = 0;
  if (m >= p.M) { r.base = -1; return r; }
  if (IN == IN_SAME) { r.base = m; return r; }
  const int HW = p.H * p.W;
  int x = m % p.W;
  int t = m / p.W;
  int y = t % p.H;
  int q = t / p.H;                       // IN_CONV3: image (b*AA+view); IN_ANG: b; IN_EPIH: b*A+u; IN_EPIV: b*A+v
  r.y = y; r.x = x;
  if (IN == IN_CONV3) r.base = m;
  if (IN == IN_ANG) r.base = q * p.AA * HW + y * p.W + x;
  if (IN == IN_EPIH) r.base = q * p.A * HW + y * p.W;              // + v'*HW + x'
  if (IN == IN_EPIV) { int b = q / p.A, v = q - b * p.A; r.base = (b * p.AA + v) * HW + x; }  // + u'*A*HW + y'*W
  return r;
}

// source pixel for (row, tap) or -1 (zero padding / out of range)
template <int IN>
__device__ __forceinline__ int src_pixel(const RowInfo<IN>& r, int tap, const GemmArgs& p) {
  if (r.base < 0) return -1;
  if (IN == IN_SAME) return r.base;
  if (IN == IN_CONV3) {
    int dy = tap / 3 - 1, dx = tap - (tap / 3) * 3 - 1;
    int yy = r.y + dy, xx = r.x + dx;
    if (yy < 0 || yy >= p.H || xx < 0 || xx >= p.W) return -1;
    return r.base + dy * p.W + dx;
  }
  const int HW = p.H * p.W;
  if (IN == IN_ANG) return r.base + tap * HW;
  const int pad = p.A * (p.A - 1) / 2;
  if (IN == IN_EPIH) {
    int col = p.A * r.x + tap - pad;
    if (col < 0 || col >= p.W * p.A) return -1;
    int xs = col / p.A, vs = col - xs * p.A;
    return r.base + vs * HW + xs;
  }
  {  // IN_EPIV
    int row = p.A * r.y + tap - pad;
    if (row < 0 || row >= p.H * p.A) return -1;
    int ys = row / p.A, us = row - ys * p.A;
    return r.base + us * p.A * HW + ys * p.W;
  }
}

// destination pixel for (row m, chunk)
template <int OUT>
__device__ __forceinline__ long long dst_pixel(int m, int chunk, const GemmArgs& p) {
  if (OUT == OUT_SAME) return m;
  const int HW = p.H * p.W;
  int yx = m % HW;
  int q = m / HW;
  if (OUT == OUT_VIEWS) return ((long long)q * p.AA + chunk) * HW + yx;          // q = b, chunk = view
  if (OUT == OUT_EPIH) return ((long long)q * p.A + chunk) * HW + yx;            // q = b*A+u, chunk = v
  int b = q / p.A, v = q - b * p.A;                                               // OUT_EPIV: q = b*A+v, chunk = u
  return (((long long)b * p.A + chunk) * p.A + v) * HW + yx;
}

template <int IN, int OUT, int CIN, int NT>
__global__ __launch_bounds__(256) void k_gemm_gather(GemmArgs p) {
  constexpr int BN = 32 * NT;
  constexpr int NCH = (CIN + 63) / 64;               // 64-float stages per tap
  constexpr int LASTW = CIN - 64 * (NCH - 1);        // width of the last stage of a tap (multiple of 8)
  constexpr int AROWS = BM / 16;                     // A-slab rows per loading thread (8)
  constexpr int BROWS = BN / 16;                     // W-slab rows per loading thread
  static_assert(CIN % 8 == 0 && LASTW % 8 == 0, "K granularity is 8");
  __shared__ __attribute__((aligned(16))) float smem[(BM + BN) * LDS_ROW];
  float* sA = smem;
  float* sB = smem + BM * LDS_ROW;

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int c16 = tid & 15, r0 = tid >> 4;

  // XCD-aware row-block remap: blocks b and b+8 share an XCD (round-robin dispatch), so give each XCD a
  // contiguous run of row blocks -> neighbouring blocks' halo rows and the weights hit the same L2.
  int bid = blockIdx.x;
  {
    const int nb = p.nblk_m, q = nb >> 3, rem = nb & 7, xcd = bid & 7, idx = bid >> 3;
    bid = (xcd < rem ? xcd * (q + 1) : rem * (q + 1) + (xcd - rem) * q) + idx;
  }
  const int m0 = bid * BM;
  const int n0 = blockIdx.y * BN;

  RowInfo<IN> rows[AROWS];
#pragma unroll
  for (int i = 0; i < AROWS; ++i) rows[i] = decode_row<IN>(m0 + r0 + 16 * i, p);

  float4 ra[AROWS], rb[BROWS];
  const int nstages = p.ntaps * NCH;

  auto prefetch = [&](int s) {
    const int tap = s / NCH, ch = s - tap * NCH;
    const int kw = (ch == NCH - 1) ? LASTW : 64;
    const int koff = ch * 64 + c16 * 4;
    const bool kin = c16 * 4 < kw;
#pragma unroll
    for (int i = 0; i < AROWS; ++i) {
      int sp = src_pixel<IN>(rows[i], tap, p);
      ra[i] = (sp >= 0 && kin) ? *reinterpret_cast<const float4*>(p.X + (long long)sp * p.x_stride + p.x_choff + koff)
                               : make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int i = 0; i < BROWS; ++i) {
      int n = n0 + r0 + 16 * i;   // < Npad by construction of the grid
      rb[i] = kin ? *reinterpret_cast<const float4*>(p.Wp + ((long long)tap * p.Npad + n) * CIN + koff)
                  : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  };

  f32x16 acc[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

  const int half = lane >> 5, l31 = lane & 31;
  const float* aRow = sA + (wave * 32 + l31) * LDS_ROW + 4 * half;
  const float* bRow = sB + l31 * LDS_ROW + 4 * half;

  prefetch(0);
  for (int s = 0; s < nstages; ++s) {
    if (s > 0) __syncthreads();  // everyone has finished reading the previous stage's LDS image
#pragma unroll
    for (int i = 0; i < AROWS; ++i) *reinterpret_cast<float4*>(sA + (r0 + 16 * i) * LDS_ROW + c16 * 4) = ra[i];
#pragma unroll
    for (int i = 0; i < BROWS; ++i) *reinterpret_cast<float4*>(sB + (r0 + 16 * i) * LDS_ROW + c16 * 4) = rb[i];
    __syncthreads();
    if (s + 1 < nstages) prefetch(s + 1);  // global loads fly while this stage computes

    const int ch = s % NCH;
    const int ng = ((ch == NCH - 1) ? LASTW : 64) / 8;
#pragma unroll 2
    for (int j = 0; j < ng; ++j) {
      float4 a = *reinterpret_cast<const float4*>(aRow + 8 * j);
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        float4 b = *reinterpret_cast<const float4*>(bRow + t * 32 * LDS_ROW + 8 * j);
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b.x, acc[t], 0, 0, 0);
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b.y, acc[t], 0, 0, 0);
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b.z, acc[t], 0, 0, 0);
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b.w, acc[t], 0, 0, 0);
      }
    }
  }

  // epilogue: C/D layout of 32x32 MFMA: col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int m = m0 + wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
    if (m >= p.M) continue;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const int n = n0 + t * 32 + l31;
      if (n >= p.N) continue;
      float v = acc[t][r];
      if (p.bias) v += p.bias[n];
      v = v >= 0.f ? v : v * p.slope;
      int chunk = 0, c = n;
      if (OUT != OUT_SAME) { chunk = n / p.CH; c = n - chunk * p.CH; }
      const long long dp = dst_pixel<OUT>(m, chunk, p);
      if (p.R1) v += p.R1[dp * p.r1_stride + p.r1_choff + c];
      if (p.R2) v += p.R2[dp * p.r2_stride + p.r2_choff + c];
      p.Y[dp * p.y_stride + p.y_choff + c] = v;
    }
  }
}

template <int IN, int OUT, int CIN, int NT>
int launch_gemm(GemmArgs p, hipStream_t st) {
  if (p.M <= 0) return LFSR_OK;
  if (p.Npad % (32 * NT) != 0 && NT != 1) return LFSR_E_ARG;
  p.nblk_m = (p.M + BM - 1) / BM;
  dim3 grid((unsigned)p.nblk_m, (unsigned)((p.Npad + 32 * NT - 1) / (32 * NT)));
  hipLaunchKernelGGL((k_gemm_gather<IN, OUT, CIN, NT>), grid, dim3(256), 0, st, p);
  LFSR_CHECK_LAUNCH();
  return LFSR_OK;
}

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
__global__ __launch_bounds__(256) void k_initconv(const float* __restrict__ x, const float* __restrict__ w, float* __restrict__ y, int y_stride, int y_choff,
                                                  int B, int A, int h, int wd) {
  __shared__ float sw[64 * 9];
  for (int i = threadIdx.x; i < 64 * 9; i += 256) sw[i] = w[i];
  __syncthreads();
  const long long npix = (long long)B * A * A * h * wd;
  const int Wm = A * wd, Hm = A * h;
  for (long long g = (long long)blockIdx.x * 256 + threadIdx.x; g < npix * 16; g += (long long)gridDim.x * 256) {
    long long pix = g >> 4;
    int c4 = (int)(g & 15) * 4;
    int xx = (int)(pix % wd);
    long long t = pix / wd;
    int yy = (int)(t % h);
    t /= h;
    int view = (int)(t % (A * A));
    int b = (int)(t / (A * A));
    int u = view / A, v = view - u * A;
    const float* img = x + (long long)b * Hm * Wm + (long long)(u * h) * Wm + v * wd;  // this view's top-left in the mosaic
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
#pragma unroll
    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) {
        int sy = yy + ky - 1, sx = xx + kx - 1;
        float xv = (sy >= 0 && sy < h && sx >= 0 && sx < wd) ? img[(long long)sy * Wm + sx] : 0.f;
        int k = ky * 3 + kx;
        a0 = fmaf(xv, sw[(c4 + 0) * 9 + k], a0);
        a1 = fmaf(xv, sw[(c4 + 1) * 9 + k], a1);
        a2 = fmaf(xv, sw[(c4 + 2) * 9 + k], a2);
        a3 = fmaf(xv, sw[(c4 + 3) * 9 + k], a3);
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

// one thread per (LR pixel, sub-row i): s outputs along j, stored contiguously in the HR mosaic
template <int S>
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
  for (long long g = (long long)blockIdx.x * 256 + threadIdx.x; g < total; g += (long long)gridDim.x * 256) {
    int x = (int)(g % w);
    long long t = g / w;
    int i = (int)(t % S);
    t /= S;
    int y = (int)(t % h);
    t /= h;
    int view = (int)(t % (A * A));
    int b = (int)(t / (A * A));
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

inline int npad32(int n) { return (n + 31) / 32 * 32; }

}  // namespace

extern "C" {

size_t lfsr_packed_weight_floats(int O, int C, int taps) { return (size_t)taps * (size_t)npad32(O) * (size_t)C; }

int lfsr_pack_conv_weight(const float* w, float* packed, int O, int C, int taps, int perm, int ch, void* stream) {
  if (!w || !packed || O <= 0 || C <= 0 || taps <= 0 || (perm != 0 && perm != 1)) return LFSR_E_ARG;
  if (perm == 1 && (ch <= 0 || O % ch != 0)) return LFSR_E_ARG;
  long long total = (long long)taps * npad32(O) * C;
  hipLaunchKernelGGL(k_pack_weight, dim3(lfsr_blocks(total, 256)), dim3(256), 0, lfsr_stream(stream), w, packed, O, C, taps, npad32(O), perm, ch);
  LFSR_CHECK_LAUNCH();
  return LFSR_OK;
}

int lfsr_conv3x3_fwd(const float* x, int x_stride, int x_choff, const float* w_packed, float* y, int y_stride, int y_choff,
                     const float* r1, int r1_stride, int r1_choff, const float* r2, int r2_stride, int r2_choff,
                     int n_img, int h, int w, float slope, void* stream) {
  if (!x || !w_packed || !y || n_img <= 0 || h <= 0 || w <= 0) return LFSR_E_ARG;
  if (x_stride < x_choff + 64 || y_stride < y_choff + 64 || (x_stride | x_choff) & 3) return LFSR_E_ARG;
  if ((long long)n_img * h * w >= (1LL << 31) / 4) return LFSR_E_ARG;
  {
    // v2 halo-tile kernel needs 16-B aligned channel vectors on every operand; LFSR_CONV3X3=gather forces v1 (A/B runs)
    const char* sel = getenv("LFSR_CONV3X3");
    const bool force_v1 = sel && sel[0] == 'g';
    const bool al = !((y_stride | y_choff) & 3) && (!r1 || !((r1_stride | r1_choff) & 3)) && (!r2 || !((r2_stride | r2_choff) & 3));
    if (al && !force_v1)
      return lfsr_conv3x3_halo_launch(x, x_stride, x_choff, w_packed, y, y_stride, y_choff, r1, r1_stride, r1_choff, r2, r2_stride, r2_choff,
                                      n_img, h, w, slope, lfsr_stream(stream));
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
  if (!x || !w_packed || !y || M <= 0 || N <= 0) return LFSR_E_ARG;
  if (x_stride < x_choff + cin || y_stride < y_choff + N || (x_stride | x_choff) & 3) return LFSR_E_ARG;
  GemmArgs p{};
  p.X = x; p.x_stride = x_stride; p.x_choff = x_choff; p.Wp = w_packed; p.bias = bias;
  p.Y = y; p.y_stride = y_stride; p.y_choff = y_choff;
  p.M = M; p.N = N; p.Npad = npad32(N); p.A = 1; p.AA = 1; p.H = 1; p.W = 1; p.ntaps = 1; p.CH = N; p.slope = slope;
  hipStream_t st = lfsr_stream(stream);
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
  if (!x || !w1_packed || !w2_packed || !tmp || !y || B <= 0 || A <= 0 || h <= 0 || w <= 0) return LFSR_E_ARG;
  if (x_stride < x_choff + 64 || y_stride < y_choff + 16 || (x_stride | x_choff) & 3) return LFSR_E_ARG;
  hipStream_t st = lfsr_stream(stream);
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

static int epiconv_gather(const float* x, int x_stride, int x_choff, const float* w1_packed, const float* w2_packed,
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
  const char* sel = getenv("LFSR_EPI");   // LFSR_EPI=gather forces the two-launch gather-GEMM path (A/B runs)
  return !(sel && sel[0] == 'g') && lfsr_epi_fused_ok(A, h, w);
}

int lfsr_epiconv_fwd(const float* x, int x_stride, int x_choff, const float* w1_packed, const float* w2_packed,
                     float* tmp, float* y, int y_stride, int y_choff, int B, int A, int h, int w, int vertical, float slope, void* stream) {
  if (!x || !w1_packed || !w2_packed || !y || B <= 0 || A <= 0 || h <= 0 || w <= 0) return LFSR_E_ARG;
  if (x_stride < x_choff + 64 || y_stride < y_choff + 32 || (x_stride | x_choff) & 3) return LFSR_E_ARG;
  if (epi_use_fused(A, h, w))
    return lfsr_epi_fused_launch(x, x_stride, x_choff, w1_packed, w2_packed, y, y_stride, y_choff, y_choff, B, A, h, w, vertical ? 2 : 1, slope, lfsr_stream(stream));
  if (!tmp) return LFSR_E_ARG;
  return epiconv_gather(x, x_stride, x_choff, w1_packed, w2_packed, tmp, y, y_stride, y_choff, B, A, h, w, vertical, slope, lfsr_stream(stream));
}

int lfsr_epiconv_hv_fwd(const float* x, int x_stride, int x_choff, const float* w1_packed, const float* w2_packed,
                        float* tmp, float* y, int y_stride, int choff_h, int choff_v, int B, int A, int h, int w, float slope, void* stream) {
  if (!x || !w1_packed || !w2_packed || !y || B <= 0 || A <= 0 || h <= 0 || w <= 0) return LFSR_E_ARG;
  if (x_stride < x_choff + 64 || y_stride < choff_h + 32 || y_stride < choff_v + 32 || (x_stride | x_choff) & 3) return LFSR_E_ARG;
  if (epi_use_fused(A, h, w))
    return lfsr_epi_fused_launch(x, x_stride, x_choff, w1_packed, w2_packed, y, y_stride, choff_h, choff_v, B, A, h, w, 3, slope, lfsr_stream(stream));
  if (!tmp) return LFSR_E_ARG;
  int rc = epiconv_gather(x, x_stride, x_choff, w1_packed, w2_packed, tmp, y, y_stride, choff_h, B, A, h, w, 0, slope, lfsr_stream(stream));
  if (rc) return rc;
  return epiconv_gather(x, x_stride, x_choff, w1_packed, w2_packed, tmp, y, y_stride, choff_v, B, A, h, w, 1, slope, lfsr_stream(stream));
}

int lfsr_initconv_fwd(const float* x, const float* w, float* y, int y_stride, int y_choff, int B, int A, int h, int wd, void* stream) {
  if (!x || !w || !y || B <= 0 || A <= 0 || h <= 0 || wd <= 0 || y_stride < y_choff + 64 || (y_stride | y_choff) & 3) return LFSR_E_ARG;
  long long total = (long long)B * A * A * h * wd * 16;
  unsigned grid = lfsr_blocks(total, 256);
  if (grid > 256u * 16) grid = 256u * 16;
  hipLaunchKernelGGL(k_initconv, dim3(grid), dim3(256), 0, lfsr_stream(stream), x, w, y, y_stride, y_choff, B, A, h, wd);
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
  switch (s) {
    case 2: hipLaunchKernelGGL((k_head<2>), dim3(grid), dim3(256), 0, st, f, f_stride, f_choff, wf, bf, x_lr, out, B, A, h, w); break;
    case 3: hipLaunchKernelGGL((k_head<3>), dim3(grid), dim3(256), 0, st, f, f_stride, f_choff, wf, bf, x_lr, out, B, A, h, w); break;
    case 4: hipLaunchKernelGGL((k_head<4>), dim3(grid), dim3(256), 0, st, f, f_stride, f_choff, wf, bf, x_lr, out, B, A, h, w); break;
    default: return LFSR_E_ARG;
  }
  LFSR_CHECK_LAUNCH();
  return LFSR_OK;
}

}  // extern "C"
