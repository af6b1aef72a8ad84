// Shared by gemm_gather.hip (forward wrappers) and bwd_ops.hip (backward wrappers): the gather-GEMM kernel
// template, its row/source/destination index maps and the launcher.  See gemm_gather.hip for the design notes.
#pragma once
#include "lfsr_common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

namespace {

enum { IN_SAME = 0, IN_CONV3 = 1, IN_ANG = 2, IN_EPIH = 3, IN_EPIV = 4,
       // backward-only gathers (the transposes of the OUT_* scatters and of the EPI line conv)
       IN_CHK_H = 5,    // rows (b*A+u, y, x), tap = v : pixel (b,u,tap,y,x)      (transpose of OUT_EPIH)
       IN_CHK_V = 6,    // rows (b*A+v, y, x), tap = u : pixel (b,tap,v,y,x)      (transpose of OUT_EPIV)
       IN_LINE_H = 7,   // rows (.., y, x), tap = dxi : row m + pad - dxi if 0 <= x + pad - dxi < W
       IN_LINE_V = 8 }; // rows (.., y, x), tap = dyi : row m + (pad - dyi) W if 0 <= y + pad - dyi < H
enum { OUT_SAME = 0, OUT_VIEWS = 1, OUT_EPIH = 2, OUT_EPIV = 3,
       OUT_PS_HR = 4 };  // rows = VCL pixels, chunk = i*S+j : pixel ((u*h+y)*S+i, (v*w+x)*S+j) of the channel-last HR mosaic (PixelShuffle(S) + MacPI2SAI-free)

struct GemmArgs {
  const float* X; int x_stride; int x_choff;
  const float* Wp;                 // [ntaps][Npad][CIN]
  const float* bias;               // [>=N] or null
  float* Y; int y_stride; int y_choff;
  const float* R1; int r1_stride; int r1_choff;
  const float* R2; int r2_stride; int r2_choff;
  const float* Mk; int mk_stride; int mk_choff; float mk_slope;   // backward: v *= (Mk > 0 ? 1 : mk_slope)  (LeakyReLU')
  int M, N, Npad;
  int A, AA, H, W;                 // angular res, A*A, view height/width
  int ntaps;
  int CH;                          // channels per destination pixel for OUT_VIEWS/OUT_EPI* (N = chunks*CH)
  float slope;                     // LeakyReLU slope; 1.0f = identity, 0.0f = ReLU
  int S;                           // OUT_PS_HR: upscale factor
  int nblk_m;                      // number of row blocks (for the XCD remap)
  int addr32;                      // set by launch_gemm: the gathered tensor spans < 2 GiB -> descriptor loads with 32-bit offsets
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
  if (IN == IN_CHK_H) r.base = q * p.A * HW + y * p.W + x;                                    // + tap*HW
  if (IN == IN_CHK_V) { int b = q / p.A, v = q - b * p.A; r.base = (b * p.AA + v) * HW + y * p.W + x; }  // + tap*A*HW
  if (IN == IN_LINE_H || IN == IN_LINE_V) r.base = m;
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
  if (IN == IN_ANG || IN == IN_CHK_H) return r.base + tap * HW;
  if (IN == IN_CHK_V) return r.base + tap * p.A * HW;
  if (IN == IN_LINE_H) { int xs = r.x + (p.A - 1) / 2 - tap; return (xs < 0 || xs >= p.W) ? -1 : r.base + (p.A - 1) / 2 - tap; }
  if (IN == IN_LINE_V) { int ys = r.y + (p.A - 1) / 2 - tap; return (ys < 0 || ys >= p.H) ? -1 : r.base + ((p.A - 1) / 2 - tap) * p.W; }
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
  if (OUT == OUT_PS_HR) {                                                         // q = b*AA + view
    int y = yx / p.W, x = yx - y * p.W, b = q / p.AA, view = q - b * p.AA, u = view / p.A, v = view - u * p.A;
    int i = chunk / p.S, j = chunk - i * p.S;
    return ((long long)b * p.A * p.H * p.S + (long long)(u * p.H + y) * p.S + i) * ((long long)p.A * p.W * p.S) + (long long)(v * p.W + x) * p.S + j;
  }
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

  // stage loads: with 32-bit addressing (launch_gemm checks the span) through a buffer descriptor -- padding / out-of-range rows
  // become out-of-range offsets that read as zero, no per-row branch and no 64-bit address arithmetic between the barrier and
  // the MFMAs; the 64-bit form remains for larger tensors
  typedef float f32x4q __attribute__((ext_vector_type(4)));
  const __amdgpu_buffer_rsrc_t rsX = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.X), 0, (int)0x80000000u, 0x00020000);
  auto prefetch = [&](int s) {
    const int tap = s / NCH, ch = s - tap * NCH;
    const int kw = (ch == NCH - 1) ? LASTW : 64;
    const int koff = ch * 64 + c16 * 4;
    const bool kin = c16 * 4 < kw;
    if (p.addr32) {
#pragma unroll
      for (int i = 0; i < AROWS; ++i) {
        const int sp = src_pixel<IN>(rows[i], tap, p);
        const f32x4q v = __builtin_bit_cast(f32x4q, __builtin_amdgcn_raw_buffer_load_b128(rsX, (sp >= 0 && kin) ? (sp * p.x_stride + p.x_choff + koff) * 4 : (int)0x80000000u, 0, 0));
        ra[i] = make_float4(v.x, v.y, v.z, v.w);
      }
    } else {
#pragma unroll
      for (int i = 0; i < AROWS; ++i) {
        int sp = src_pixel<IN>(rows[i], tap, p);
        ra[i] = (sp >= 0 && kin) ? *reinterpret_cast<const float4*>(p.X + (long long)sp * p.x_stride + p.x_choff + koff)
                                 : make_float4(0.f, 0.f, 0.f, 0.f);
      }
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

  // epilogue: C/D layout of 32x32 MFMA: col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5).
  // Per column tile: destination pixels first, then every optional operand load (mask, residuals) of the 16 elements back to
  // back, then the arithmetic and the stores -- written as load-use-store per element, each load was waited for in turn
  // (16 serial memory round trips per tile and wave)
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    const int n = n0 + t * 32 + l31;
    const bool ncol = n < p.N;
    int chunk = 0, c = n;
    if (OUT != OUT_SAME && ncol) { chunk = n / p.CH; c = n - chunk * p.CH; }
    const float bs = (p.bias && ncol) ? p.bias[n] : 0.f;
    long long dp[16];
    float mk[16], r1[16], r2[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int m = m0 + wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
      dp[r] = (ncol && m < p.M) ? dst_pixel<OUT>(m, chunk, p) : -1;
    }
    if (p.Mk) {
#pragma unroll
      for (int r = 0; r < 16; ++r) mk[r] = dp[r] >= 0 ? p.Mk[dp[r] * p.mk_stride + p.mk_choff + c] : 1.f;
    }
    if (p.R1) {
#pragma unroll
      for (int r = 0; r < 16; ++r) r1[r] = dp[r] >= 0 ? p.R1[dp[r] * p.r1_stride + p.r1_choff + c] : 0.f;
    }
    if (p.R2) {
#pragma unroll
      for (int r = 0; r < 16; ++r) r2[r] = dp[r] >= 0 ? p.R2[dp[r] * p.r2_stride + p.r2_choff + c] : 0.f;
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      if (dp[r] < 0) continue;
      float v = acc[t][r] + bs;
      v = v >= 0.f ? v : v * p.slope;
      if (p.Mk) v *= mk[r] > 0.f ? 1.f : p.mk_slope;
      if (p.R1) v += r1[r];
      if (p.R2) v += r2[r];
      p.Y[dp[r] * p.y_stride + p.y_choff + c] = v;
    }
  }
}

template <int IN, int OUT, int CIN, int NT>
int launch_gemm(GemmArgs p, hipStream_t st) {
  if (p.M <= 0) return LFSR_OK;
  if (p.Npad % (32 * NT) != 0 && NT != 1) return LFSR_E_ARG;
  p.nblk_m = (p.M + BM - 1) / BM;
  p.addr32 = (long long)p.M * (p.AA > 0 ? p.AA : 1) * p.x_stride * 4 < (1LL << 31);   // every gather reads at most M * A^2 pixels
  dim3 grid((unsigned)p.nblk_m, (unsigned)((p.Npad + 32 * NT - 1) / (32 * NT)));
  hipLaunchKernelGGL((k_gemm_gather<IN, OUT, CIN, NT>), grid, dim3(256), 0, st, p);
  LFSR_CHECK_LAUNCH();
  return LFSR_OK;
}


inline int npad32(int n) { return (n + 31) / 32 * 32; }

}  // namespace
