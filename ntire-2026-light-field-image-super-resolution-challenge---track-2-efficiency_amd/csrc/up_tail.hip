// Fused up-sampling tail of EPIT / LFT (EPIT.py:44-49,70; LFT.py:52-57,94-96), fp32:
//   out = conv3x3_{64->1}( LeakyReLU( PixelShuffle_s( conv1x1_{64 -> 64 s^2}(F) ) ) )  +  per-view bicubic_s(x_lr)
// over the whole HR SAI mosaic (zero pad 1, crossing view borders exactly as the reference does).
// The two-kernel form materialises the (B, 64, A h s, A w s) tensor -- 105 MB per patch, written once and read 9x from L1 --
// and ran at 0.7 TB/s.  Here it never exists: a 512-thread block owns a 4 x 32 tile of LR mosaic pixels (+1 halo, 204 px);
// for each chunk of CC = 64/s^2 feature channels it (1) runs the 1x1 conv for the 204 px x 64 columns (CC channels x s^2
// sub-positions) on fp32 MFMA from the LDS-resident F tile, (2) LeakyReLU's the result into LDS, (3) lets every thread add
// the 9-tap x CC-channel contribution to its 4 HR outputs (the HR neighbour (Y+dy, X+dx) is sub-position ((Y+dy)%s, (X+dx)%s)
// of LR pixel ((Y+dy)/s, (X+dx)/s)).  Halo pixels outside the mosaic hold F = 0, hence U = 0 = the conv's zero padding
// (there is no bias).  Weight chunks are prefetched through registers into a double-buffered LDS slab.
#include "gemm_gather_kernel.h"
#include "lfsr_internal.h"

namespace {

constexpr int UT_Y = 4, UT_X = 32;
constexpr int UT_PIX = (UT_Y + 2) * (UT_X + 2);   // 204
constexpr int UT_ROWS = 224;                      // padded to 7 MFMA row tiles

struct UpTailArgs {
  const float* F; int f_stride; int f_choff;     // VCL features
  const float* W0p;                              // packed [64 s^2][64], row n' = ij*64 + c  (lfsr_pack_conv_weight perm 1, ch 64)
  const float* W3;                               // (1,64,3,3)
  const float* Xlr;                              // (B,1,A*h,A*w)
  float* Out;                                    // (B,1,A*h*s,A*w*s)
  int B, A, h, w;
  int tiles_y, tiles_x;
  float slope;
};

__device__ __forceinline__ void ut_cubic(float t, float c[4]) {
  const float a = -0.75f;
  float x1 = t + 1.f, x2 = t, x3 = 1.f - t, x4 = 2.f - t;
  c[0] = ((a * x1 - 5.f * a) * x1 + 8.f * a) * x1 - 4.f * a;
  c[1] = ((a + 2.f) * x2 - (a + 3.f)) * x2 * x2 + 1.f;
  c[2] = ((a + 2.f) * x3 - (a + 3.f)) * x3 * x3 + 1.f;
  c[3] = ((a * x4 - 5.f * a) * x4 + 8.f * a) * x4 - 4.f * a;
}

template <int S>
__global__ __launch_bounds__(512) void k_up_tail(UpTailArgs p) {
  constexpr int S2 = S * S, CC = 64 / S2, NCH = 64 / CC;   // channels per chunk, number of chunks
  extern __shared__ __attribute__((aligned(16))) float smu[];
  float* sF = smu;                                   // [UT_ROWS][LDS_ROW]
  float* sU = sF + UT_ROWS * LDS_ROW;                // [UT_ROWS][LDS_ROW]   U chunk, column n = dc*S2 + ij
  float* sW = sU + UT_ROWS * LDS_ROW;                // [2][64][LDS_ROW]
  float* sW3 = sW + 2 * 64 * LDS_ROW;                // [64][9]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int c16 = tid & 15, r16 = tid >> 4;
  const int half = lane >> 5, l31 = lane & 31;
  const int Hm = p.A * p.h, Wm = p.A * p.w, HW = p.h * p.w;
  int t = blockIdx.x;
  const int tx = t % p.tiles_x; t /= p.tiles_x;
  const int ty = t % p.tiles_y;
  const int b = t / p.tiles_y;
  const int Y0 = ty * UT_Y, X0 = tx * UT_X;          // LR mosaic origin of the tile

  // ---- stage the F halo tile (zero outside the mosaic), w3, and weight chunk 0 ------------------------------------------
  for (int i = tid; i < 64 * 9; i += 512) sW3[i] = p.W3[i];
  {   // 7 float4 per thread, all loads issued before the first LDS store (a load-store loop waits for each load in turn)
    float4 fv[7];
#pragma unroll
    for (int q = 0; q < 7; ++q) {
      const int idx = tid + 512 * q, px = idx >> 4, ch = idx & 15;
      fv[q] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (px < UT_PIX) {
        int ly = px / (UT_X + 2), lx = px - ly * (UT_X + 2);
        int Ym = Y0 + ly - 1, Xm = X0 + lx - 1;
        if (Ym >= 0 && Ym < Hm && Xm >= 0 && Xm < Wm) {
          int u = Ym / p.h, y = Ym - u * p.h, vv = Xm / p.w, x = Xm - vv * p.w;
          long long pix = ((long long)b * p.A * p.A + u * p.A + vv) * HW + (long long)y * p.w + x;
          fv[q] = *reinterpret_cast<const float4*>(p.F + pix * p.f_stride + p.f_choff + ch * 4);
        }
      }
    }
#pragma unroll
    for (int q = 0; q < 7; ++q) {
      const int idx = tid + 512 * q;
      *reinterpret_cast<float4*>(sF + (idx >> 4) * LDS_ROW + (idx & 15) * 4) = fv[q];
    }
  }
  // weight chunk cc: LDS row (dc*S2 + ij) <- packed row (ij*64 + cc*CC + dc)
  auto wrow = [&](int cc, int r) -> const float* { int dc = r / S2, ij = r - dc * S2; return p.W0p + ((long long)(ij * 64 + cc * CC + dc)) * 64; };
  float4 rw[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) rw[i] = *reinterpret_cast<const float4*>(wrow(0, r16 + 32 * i) + c16 * 4);
#pragma unroll
  for (int i = 0; i < 2; ++i) *reinterpret_cast<float4*>(sW + (r16 + 32 * i) * LDS_ROW + c16 * 4) = rw[i];
  __syncthreads();

  // this thread's 4 HR outputs: o = tid + 512 i  -> local HR (Yl, Xl) in the tile's (UT_Y*S) x (UT_X*S) window
  float oacc[4] = {0.f, 0.f, 0.f, 0.f};
  constexpr int HRW = UT_X * S;
  constexpr int NOUT = UT_Y * S * HRW;            // HR outputs of the tile (2048 for s = 4, 512 for s = 2)

  for (int cc = 0; cc < NCH; ++cc) {
    const float* sWc = sW + (cc & 1) * 64 * LDS_ROW;
    if (cc + 1 < NCH) {
#pragma unroll
      for (int i = 0; i < 2; ++i) rw[i] = *reinterpret_cast<const float4*>(wrow(cc + 1, r16 + 32 * i) + c16 * 4);
    }
    // ---- (1) U[px][n] = sum_k F[px][k] W[n][k] : 7 row tiles x 2 column tiles over 8 waves -----------------------------------
    for (int item = wave; item < 14; item += 8) {
      const int rt = item >> 1, ct = item & 1;
      f32x16 acc;
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = 0.f;
      const float* aRow = sF + (rt * 32 + l31) * LDS_ROW + 4 * half;
      const float* bRow = sWc + (ct * 32 + l31) * LDS_ROW + 4 * half;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        float4 a = *reinterpret_cast<const float4*>(aRow + 8 * j);
        float4 bq = *reinterpret_cast<const float4*>(bRow + 8 * j);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, bq.x, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, bq.y, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, bq.z, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, bq.w, acc, 0, 0, 0);
      }
      // (2) LeakyReLU -> sU ; D layout: column n = lane&31, row = (r&3) + 8*(r>>2) + 4*half
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        int px = rt * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
        float v = acc[r];
        sU[px * LDS_ROW + ct * 32 + l31] = v >= 0.f ? v : v * p.slope;
      }
    }
    if (cc + 1 < NCH) {
      float* sWn = sW + ((cc + 1) & 1) * 64 * LDS_ROW;
#pragma unroll
      for (int i = 0; i < 2; ++i) *reinterpret_cast<float4*>(sWn + (r16 + 32 * i) * LDS_ROW + c16 * 4) = rw[i];
    }
    __syncthreads();
    // ---- (3) 3x3 HR conv contribution of this chunk's CC channels ---------------------------------------------------------------
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int o = tid + 512 * i;
      if (o >= NOUT) break;
      const int Yl = o / HRW, Xl = o - Yl * HRW;
      float a = oacc[i];
#pragma unroll
      for (int ky = 0; ky < 3; ++ky) {
        const int Yh = Yl + ky - 1 + S;               // +S: halo row offset in HR units (never negative)
        const int ly = Yh / S, sy = Yh - ly * S;
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
          const int Xh = Xl + kx - 1 + S;
          const int lx = Xh / S, sx = Xh - lx * S;
          const float* up = sU + (ly * (UT_X + 2) + lx) * LDS_ROW + sy * S + sx;
          const float* wp = sW3 + (cc * CC) * 9 + ky * 3 + kx;
#pragma unroll
          for (int dc = 0; dc < CC; ++dc) a = fmaf(up[dc * S2], wp[dc * 9], a);
        }
      }
      oacc[i] = a;
    }
    __syncthreads();
  }

  // ---- + per-view bicubic skip, store -------------------------------------------------------------------------------------------
  const int Hs = Hm * S, Ws = Wm * S;
  const float rs = 1.0f / (float)S;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int o = tid + 512 * i;
    if (o >= NOUT) break;
    const int Yl = o / HRW, Xl = o - Yl * HRW;
    const int Y = Y0 * S + Yl, X = X0 * S + Xl;
    if (Y >= Hs || X >= Ws) continue;
    const int u = Y / (p.h * S), yl = Y - u * p.h * S, v = X / (p.w * S), xl = X - v * p.w * S;
    const float sy = ((float)yl + 0.5f) * rs - 0.5f, sx = ((float)xl + 0.5f) * rs - 0.5f;
    const float fy = floorf(sy), fx = floorf(sx);
    float cy[4], cx[4];
    ut_cubic(sy - fy, cy);
    ut_cubic(sx - fx, cx);
    const float* img = p.Xlr + (long long)b * Hm * Wm + (long long)(u * p.h) * Wm + v * p.w;
    float up = 0.f;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      int iy = min(max((int)fy - 1 + q, 0), p.h - 1);
      float rowv = 0.f;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        int ix = min(max((int)fx - 1 + j, 0), p.w - 1);
        rowv = fmaf(cx[j], img[(long long)iy * Wm + ix], rowv);
      }
      up = fmaf(cy[q], rowv, up);
    }
    p.Out[((long long)b * Hs + Y) * Ws + X] = oacc[i] + up;
  }
}

// output stage shared by the second and third form: sum over the four channel-group lanes of the pixel, + per-view bicubic skip, store
template <int S>
__device__ __forceinline__ void ut_tail_store(const UpTailArgs& p, float (&oacc)[S * S], int dcg, int b, int Ylr, int Xlr, int Hm, int Wm) {
  constexpr int S2 = S * S;
  // ---- sum over the four channel-group lanes of the pixel; lane dcg then owns S2 / 4 outputs: S = 4: sub-row dcg (4 consecutive HR columns); S = 2: sub-position dcg
#pragma unroll
  for (int i = 0; i < S2; ++i) {
    oacc[i] += __shfl_xor(oacc[i], 1);
    oacc[i] += __shfl_xor(oacc[i], 2);
  }
  const int Hs = Hm * S, Ws = Wm * S;
  const float rs = 1.0f / (float)S;
  if (Ylr >= Hm || Xlr >= Wm) return;
  constexpr int NO = S2 / 4;
  float res[NO];
  const int sy_own = S == 4 ? dcg : dcg >> 1;
#pragma unroll
  for (int j = 0; j < NO; ++j) {
    const int sx = S == 4 ? j : (dcg & 1);
    float sel = 0.f;
#pragma unroll
    for (int i = 0; i < S2; ++i) sel = (i == sy_own * S + sx) ? oacc[i] : sel;     // (register select: oacc is indexed by a lane-dependent value)
    const int Y = Ylr * S + sy_own, X = Xlr * S + sx;
    const int u = Y / (p.h * S), yl = Y - u * p.h * S, v = X / (p.w * S), xl = X - v * p.w * S;
    const float sy = ((float)yl + 0.5f) * rs - 0.5f, sxf = ((float)xl + 0.5f) * rs - 0.5f;
    const float fy = floorf(sy), fx = floorf(sxf);
    float cy[4], cx[4];
    ut_cubic(sy - fy, cy);
    ut_cubic(sxf - fx, cx);
    const float* img = p.Xlr + (long long)b * Hm * Wm + (long long)(u * p.h) * Wm + v * p.w;
    float up = 0.f;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      int iy = min(max((int)fy - 1 + q, 0), p.h - 1);
      float rowv = 0.f;
#pragma unroll
      for (int jx = 0; jx < 4; ++jx) {
        int ix = min(max((int)fx - 1 + jx, 0), p.w - 1);
        rowv = fmaf(cx[jx], img[(long long)iy * Wm + ix], rowv);
      }
      up = fmaf(cy[q], rowv, up);
    }
    res[j] = sel + up;
  }
  float* op = p.Out + ((long long)b * Hs + (Ylr * S + sy_own)) * Ws + (long long)Xlr * S;
  if constexpr (S == 4) *reinterpret_cast<float4*>(op) = make_float4(res[0], res[1], res[2], res[3]);
  else op[dcg & 1] = res[0];
}

// ---- second form (LFSR_UPTAIL=v2) ---------------------------------------------------------------------------------------------------------------------
// Same tile and chunking, two changes.  (1) The 1x1 conv runs on 16 x 16 x 4 MFMAs: 13 row tiles (208 >= 204 px) x 4 column tiles = 52 items,
// 13 per SIMD (the 32 x 32 form had 14 items: 4 / 4 / 3 / 3 per SIMD, and 224 rows).  (2) The 3x3 HR conv is gathered per LR pixel instead of per HR
// output: thread (LR pixel, channel group) loads the (S+2) x (S+2) patch of U around its pixel once per channel -- S-wide rows as one 16-B / 8-B LDS
// read -- and feeds all S^2 outputs of the pixel from registers (S = 4: 6 ds_read_b128 + 12 ds_read_b32 per channel against 144 ds_read_b32); the four
// channel-group lanes of a pixel keep partial sums over their own channels for the whole kernel and are added once at the end.
constexpr int UT2_ROWS = 208;

template <int S>
__global__ __launch_bounds__(512) void k_up_tail2(UpTailArgs p) {
  constexpr int S2 = S * S, CC = 64 / S2, NCH = 64 / CC;   // channels per chunk, number of chunks
  constexpr int CPT = CC / 4;                              // channels per thread and chunk (1 for S = 4, 4 for S = 2)
  constexpr int PW = UT_X + 2;                             // halo tile width (34)
  typedef float f32x4m __attribute__((ext_vector_type(4)));
  extern __shared__ __attribute__((aligned(16))) float smu[];
  float* sF = smu;                                   // [UT2_ROWS][LDS_ROW]
  float* sU = sF + UT2_ROWS * LDS_ROW;               // [UT2_ROWS][LDS_ROW]   U chunk, column n = dc*S2 + ij
  float* sW = sU + UT2_ROWS * LDS_ROW;               // [2][64][LDS_ROW]
  float* sW3 = sW + 2 * 64 * LDS_ROW;                // [64][9]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int c16 = tid & 15, r16 = tid >> 4;
  const int l15 = lane & 15, g = lane >> 4;
  const int Hm = p.A * p.h, Wm = p.A * p.w, HW = p.h * p.w;
  int t = blockIdx.x;
  const int tx = t % p.tiles_x; t /= p.tiles_x;
  const int ty = t % p.tiles_y;
  const int b = t / p.tiles_y;
  const int Y0 = ty * UT_Y, X0 = tx * UT_X;          // LR mosaic origin of the tile

  for (int i = tid; i < 64 * 9; i += 512) sW3[i] = p.W3[i];
  {   // F halo tile, zero outside the mosaic and in the padding rows: 208 x 16 float4 = 6.5 per thread, all loads before the first LDS store
    float4 fv[7];
#pragma unroll
    for (int q = 0; q < 7; ++q) {
      const int idx = tid + 512 * q, px = idx >> 4, ch = idx & 15;
      fv[q] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (px < UT_PIX) {
        int ly = px / PW, lx = px - ly * PW;
        int Ym = Y0 + ly - 1, Xm = X0 + lx - 1;
        if (Ym >= 0 && Ym < Hm && Xm >= 0 && Xm < Wm) {
          int u = Ym / p.h, y = Ym - u * p.h, vv = Xm / p.w, x = Xm - vv * p.w;
          long long pix = ((long long)b * p.A * p.A + u * p.A + vv) * HW + (long long)y * p.w + x;
          fv[q] = *reinterpret_cast<const float4*>(p.F + pix * p.f_stride + p.f_choff + ch * 4);
        }
      }
    }
#pragma unroll
    for (int q = 0; q < 7; ++q) {
      const int idx = tid + 512 * q;
      if ((idx >> 4) < UT2_ROWS) *reinterpret_cast<float4*>(sF + (idx >> 4) * LDS_ROW + (idx & 15) * 4) = fv[q];
    }
  }
  auto wrow = [&](int cc, int r) -> const float* { int dc = r / S2, ij = r - dc * S2; return p.W0p + ((long long)(ij * 64 + cc * CC + dc)) * 64; };
  float4 rw[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) rw[i] = *reinterpret_cast<const float4*>(wrow(0, r16 + 32 * i) + c16 * 4);
#pragma unroll
  for (int i = 0; i < 2; ++i) *reinterpret_cast<float4*>(sW + (r16 + 32 * i) * LDS_ROW + c16 * 4) = rw[i];
  __syncthreads();

  // phase-3 role: LR pixel pp of the 4 x 32 tile, channel group dcg (channels dcg*CPT .. +CPT-1 of every chunk)
  const int dcg = tid & 3, pp = tid >> 2;
  const int ply = 1 + pp / UT_X, plx = 1 + pp % UT_X;           // halo-tile coordinates of the pixel
  const float* uC = sU + (ply * PW + plx) * LDS_ROW;            // its U row; neighbours at +-LDS_ROW (columns) and +-PW*LDS_ROW (rows)
  float oacc[S2];
#pragma unroll
  for (int i = 0; i < S2; ++i) oacc[i] = 0.f;

  for (int cc = 0; cc < NCH; ++cc) {
    const float* sWc = sW + (cc & 1) * 64 * LDS_ROW;
    if (cc + 1 < NCH) {
#pragma unroll
      for (int i = 0; i < 2; ++i) rw[i] = *reinterpret_cast<const float4*>(wrow(cc + 1, r16 + 32 * i) + c16 * 4);
    }
    // ---- (1) U[px][n] = sum_k F[px][k] W[n][k]: item = (row tile of 16 px, column tile of 16); lane (l15, g) feeds k = 16 jj + 4 g + e of step (jj, e)
    for (int item = wave; item < 52; item += 8) {
      const int rt = item >> 2, ct = item & 3;
      f32x4m acc = {0.f, 0.f, 0.f, 0.f};
      const float* aRow = sF + (rt * 16 + l15) * LDS_ROW + 4 * g;
      const float* bRow = sWc + (ct * 16 + l15) * LDS_ROW + 4 * g;
#pragma unroll
      for (int jj = 0; jj < 4; ++jj) {
        const float4 a = *reinterpret_cast<const float4*>(aRow + 16 * jj);
        const float4 bq = *reinterpret_cast<const float4*>(bRow + 16 * jj);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, bq.x, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, bq.y, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, bq.z, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, bq.w, acc, 0, 0, 0);
      }
      // (2) LeakyReLU -> sU ; D layout: row (pixel) = 4 g + r, column = l15
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float v = acc[r];
        sU[(rt * 16 + 4 * g + r) * LDS_ROW + ct * 16 + l15] = v >= 0.f ? v : v * p.slope;
      }
    }
    if (cc + 1 < NCH) {
      float* sWn = sW + ((cc + 1) & 1) * 64 * LDS_ROW;
#pragma unroll
      for (int i = 0; i < 2; ++i) *reinterpret_cast<float4*>(sWn + (r16 + 32 * i) * LDS_ROW + c16 * 4) = rw[i];
    }
    __syncthreads();
    // ---- (3) 3x3 HR conv of this chunk's channels: patch P[py][px], HR offset (py - 1, px - 1) from the pixel's first sub-position ---------------
#pragma unroll
    for (int e = 0; e < CPT; ++e) {
      const int dc = dcg * CPT + e;
      const float* uc = uC + dc * S2;
      float P[S + 2][S + 2];
#pragma unroll
      for (int py = 0; py < S + 2; ++py) {
        const int dy = py == 0 ? -1 : py == S + 1 ? 1 : 0, sy = py == 0 ? S - 1 : py == S + 1 ? 0 : py - 1;
        const float* ur = uc + dy * PW * LDS_ROW + sy * S;
        if constexpr (S == 4) {
          // the two halo columns come as the neighbours' whole 16-B quads: as dword reads (element 3 of the left pixel's quad, element 0 of the right one's) the 32 lanes
          // of a group hit only the 8 banks = 3 (or 0) mod 4 -- four-way conflicts on 12 of a thread's 18 reads per channel (SQ_LDS_BANK_CONFLICT 44 % of this kernel's
          // LDS-active cycles, LDS busy half of the launch: profiles/r04_logs/pmc_lds_*); the 16-B reads are conflict-free for the hardware's lane groups
          const float4 mL = *reinterpret_cast<const float4*>(ur - LDS_ROW);
          const float4 m = *reinterpret_cast<const float4*>(ur);
          const float4 mR = *reinterpret_cast<const float4*>(ur + LDS_ROW);
          P[py][0] = mL.w; P[py][1] = m.x; P[py][2] = m.y; P[py][3] = m.z; P[py][4] = m.w; P[py][5] = mR.x;
        } else {
          P[py][0] = ur[-LDS_ROW + S - 1];
          P[py][S + 1] = ur[LDS_ROW];
          const float2 m = *reinterpret_cast<const float2*>(ur);
          P[py][1] = m.x; P[py][2] = m.y;
        }
      }
      float wv[9];
#pragma unroll
      for (int q = 0; q < 9; ++q) wv[q] = sW3[(cc * CC + dc) * 9 + q];
#pragma unroll
      for (int sy = 0; sy < S; ++sy)
#pragma unroll
        for (int sx = 0; sx < S; ++sx) {
          float a = oacc[sy * S + sx];
#pragma unroll
          for (int ky = 0; ky < 3; ++ky)
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) a = fmaf(P[sy + ky][sx + kx], wv[ky * 3 + kx], a);
          oacc[sy * S + sx] = a;
        }
    }
    __syncthreads();
  }

  ut_tail_store<S>(p, oacc, dcg, b, Y0 + ply - 1, X0 + plx - 1, Hm, Wm);
}

// ---- third form (default): the 1x1 conv on the bf16 MFMA pipe with exact three-term operands -------------------------------------------------------
// As k_up_tail2, but U = F W^T runs as v_mfma_f32_16x16x32_bf16 on fp32 operands split by truncation into three bf16 terms (six products of order <= 2,
// fp32 accumulation; see rowgemm_b3.hip).  The feature tile never enters LDS: a wave owns up to four (16-pixel row tile, 32-column half) units for all chunks
// and keeps the three planes of its row tiles' features in registers (B... A-operand order, loaded straight from global memory with the halo / zero logic);
// the weight chunks are split into planes while they are staged (double-buffered).  26 units over 8 waves = 7 / 7 / 6 / 6 per SIMD, 12 MFMAs of 17 cycles per
// unit and K step against 16 of 32.
typedef unsigned u32x4u __attribute__((ext_vector_type(4)));
__device__ __forceinline__ unsigned ut_hi_pair(unsigned hi_src, unsigned lo_src) { return __builtin_amdgcn_perm(hi_src, lo_src, 0x07060302u); }
__device__ __forceinline__ float ut_residual(float a) { return a - __uint_as_float(__float_as_uint(a) & 0xffff0000u); }
__device__ __forceinline__ void ut_split8(const float4 lo, const float4 hi, u32x4u& p0, u32x4u& p1, u32x4u& p2) {
  const float a[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
#pragma unroll
  for (int j = 0; j < 4; ++j) { unsigned t0, t1, t2; lfsr_split_pair(a[2 * j], a[2 * j + 1], t0, t1, t2); p0[j] = t0; p1[j] = t1; p2[j] = t2; }
}
typedef float f32x4u __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void ut_mfma(f32x4u& c, const u32x4u a, const u32x4u b) {
  asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b));
}

template <int S>
__global__ __launch_bounds__(512) void k_up_tail3(UpTailArgs p) {
  constexpr int S2 = S * S, CC = 64 / S2, NCH = 64 / CC;
  constexpr int CPT = CC / 4;
  constexpr int PW = UT_X + 2;
  constexpr int WPL = 64 * 64;                        // bf16 per weight plane: [K step s][k-group g][row n][8] -- the conflict-free image of rowgemm_b3.hip
  extern __shared__ __attribute__((aligned(16))) float smu[];
  float* sU = smu;                                    // [UT2_ROWS][LDS_ROW]   U chunk, column n = dc*S2 + ij
  float* sW3 = sU + UT2_ROWS * LDS_ROW;               // [64][9]
  unsigned short* sWb = reinterpret_cast<unsigned short*>(sW3 + 64 * 9);   // [2][3 planes][8 (K step, k-group)][64 rows][8]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l15 = lane & 15, g = lane >> 4;
  const int Hm = p.A * p.h, Wm = p.A * p.w, HW = p.h * p.w;
  int t = blockIdx.x;
  const int tx = t % p.tiles_x; t /= p.tiles_x;
  const int ty = t % p.tiles_y;
  const int b = t / p.tiles_y;
  const int Y0 = ty * UT_Y, X0 = tx * UT_X;

  for (int i = tid; i < 64 * 9; i += 512) sW3[i] = p.W3[i];
  // weight chunk cc: LDS row (dc*S2 + ij) <- packed row (ij*64 + cc*CC + dc); thread = (row r = tid >> 3, eight consecutive k: tid & 7), split while staged
  const int wr = tid >> 3, wk = tid & 7;
  auto wsrc = [&](int cc) -> const float* { int dc = wr / S2, ij = wr - dc * S2; return p.W0p + ((long long)(ij * 64 + cc * CC + dc)) * 64 + wk * 8; };
  float4 rw[2];
  auto fetch_w = [&](int cc) { const float* s0 = wsrc(cc); rw[0] = *reinterpret_cast<const float4*>(s0); rw[1] = *reinterpret_cast<const float4*>(s0 + 4); };
  auto store_w = [&](int bufi) {
    u32x4u p0, p1, p2;
    ut_split8(rw[0], rw[1], p0, p1, p2);
    unsigned short* d = sWb + bufi * 3 * WPL + (wk * 64 + wr) * 8;       // wk = 4 s + g
    *reinterpret_cast<u32x4u*>(d) = p0; *reinterpret_cast<u32x4u*>(d + WPL) = p1; *reinterpret_cast<u32x4u*>(d + 2 * WPL) = p2;
  };
  fetch_w(0);

  // this wave's units u = wave + 8 i (i < 4, u < 26): row tile u >> 1, column half u & 1; the feature planes of its row tiles, B-operand order: lane (pixel l15, k-group g)
  u32x4u f0[4][2], f1[4][2], f2[4][2];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int u = wave + 8 * i;
    const int px = (u >> 1) * 16 + l15;
    float4 v[2][2];
#pragma unroll
    for (int s = 0; s < 2; ++s) { v[s][0] = make_float4(0.f, 0.f, 0.f, 0.f); v[s][1] = v[s][0]; }
    if (u < 26 && px < UT_PIX) {
      const int ly = px / PW, lx = px - ly * PW;
      const int Ym = Y0 + ly - 1, Xm = X0 + lx - 1;
      if (Ym >= 0 && Ym < Hm && Xm >= 0 && Xm < Wm) {
        const int uu = Ym / p.h, y = Ym - uu * p.h, vv = Xm / p.w, x = Xm - vv * p.w;
        const long long pix = ((long long)b * p.A * p.A + uu * p.A + vv) * HW + (long long)y * p.w + x;
        const float* src = p.F + pix * p.f_stride + p.f_choff + 8 * g;
#pragma unroll
        for (int s = 0; s < 2; ++s) { v[s][0] = *reinterpret_cast<const float4*>(src + 32 * s); v[s][1] = *reinterpret_cast<const float4*>(src + 32 * s + 4); }
      }
    }
#pragma unroll
    for (int s = 0; s < 2; ++s) ut_split8(v[s][0], v[s][1], f0[i][s], f1[i][s], f2[i][s]);
#pragma unroll
    for (int s = 0; s < 2; ++s) asm volatile("s_nop 4" : "+v"(f0[i][s]), "+v"(f1[i][s]), "+v"(f2[i][s]));   // VALU write -> asm MFMA read: wait states by hand, tied to the planes
  }
  store_w(0);
  __syncthreads();

  const int dcg = tid & 3, pp = tid >> 2;
  const int ply = 1 + pp / UT_X, plx = 1 + pp % UT_X;
  const float* uC = sU + (ply * PW + plx) * LDS_ROW;
  float oacc[S2];
#pragma unroll
  for (int i = 0; i < S2; ++i) oacc[i] = 0.f;

  for (int cc = 0; cc < NCH; ++cc) {
    const unsigned short* sWc = sWb + (cc & 1) * 3 * WPL;
    if (cc + 1 < NCH) fetch_w(cc + 1);
    // ---- (1) U[px][n]: D[weight row n][pixel]: lane (pixel l15, g) receives columns n = 16 ct + 4 g + r of its pixel
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int u = wave + 8 * i;
      if (u < 26) {
        const int rt = u >> 1, ch = u & 1;
#pragma unroll
        for (int c2 = 0; c2 < 2; ++c2) {
          const int ct = 2 * ch + c2;
          f32x4u acc = {0.f, 0.f, 0.f, 0.f};
          asm volatile("s_nop 1" : "+v"(acc));
          const unsigned short* wq = sWc + (g * 64 + ct * 16 + l15) * 8;
#pragma unroll
          for (int s = 0; s < 2; ++s) {
            const u32x4u w0 = *reinterpret_cast<const u32x4u*>(wq + 4 * 64 * 8 * s);
            const u32x4u w1 = *reinterpret_cast<const u32x4u*>(wq + 4 * 64 * 8 * s + WPL);
            const u32x4u w2 = *reinterpret_cast<const u32x4u*>(wq + 4 * 64 * 8 * s + 2 * WPL);
            ut_mfma(acc, w2, f0[i][s]); ut_mfma(acc, w0, f2[i][s]); ut_mfma(acc, w1, f1[i][s]);
            ut_mfma(acc, w1, f0[i][s]); ut_mfma(acc, w0, f1[i][s]); ut_mfma(acc, w0, f0[i][s]);
          }
          asm volatile("s_nop 15\n\ts_nop 7" : "+v"(acc));
          // (2) LeakyReLU -> sU: four consecutive columns of one pixel: one 16-B store
          float4 o;
          o.x = acc[0] >= 0.f ? acc[0] : acc[0] * p.slope; o.y = acc[1] >= 0.f ? acc[1] : acc[1] * p.slope;
          o.z = acc[2] >= 0.f ? acc[2] : acc[2] * p.slope; o.w = acc[3] >= 0.f ? acc[3] : acc[3] * p.slope;
          *reinterpret_cast<float4*>(sU + (rt * 16 + l15) * LDS_ROW + ct * 16 + 4 * g) = o;
        }
      }
    }
    if (cc + 1 < NCH) store_w((cc + 1) & 1);
    __syncthreads();
    // ---- (3) 3x3 HR conv of this chunk's channels, as in k_up_tail2 ------------------------------------------------------------------------------
#pragma unroll
    for (int e = 0; e < CPT; ++e) {
      const int dc = dcg * CPT + e;
      const float* uc = uC + dc * S2;
      float P[S + 2][S + 2];
#pragma unroll
      for (int py = 0; py < S + 2; ++py) {
        const int dy = py == 0 ? -1 : py == S + 1 ? 1 : 0, sy = py == 0 ? S - 1 : py == S + 1 ? 0 : py - 1;
        const float* ur = uc + dy * PW * LDS_ROW + sy * S;
        if constexpr (S == 4) {
          // the two halo columns come as the neighbours' whole 16-B quads: as dword reads (element 3 of the left pixel's quad, element 0 of the right one's) the 32 lanes
          // of a group hit only the 8 banks = 3 (or 0) mod 4 -- four-way conflicts on 12 of a thread's 18 reads per channel (SQ_LDS_BANK_CONFLICT 44 % of this kernel's
          // LDS-active cycles, LDS busy half of the launch: profiles/r04_logs/pmc_lds_*); the 16-B reads are conflict-free for the hardware's lane groups
          const float4 mL = *reinterpret_cast<const float4*>(ur - LDS_ROW);
          const float4 m = *reinterpret_cast<const float4*>(ur);
          const float4 mR = *reinterpret_cast<const float4*>(ur + LDS_ROW);
          P[py][0] = mL.w; P[py][1] = m.x; P[py][2] = m.y; P[py][3] = m.z; P[py][4] = m.w; P[py][5] = mR.x;
        } else {
          P[py][0] = ur[-LDS_ROW + S - 1];
          P[py][S + 1] = ur[LDS_ROW];
          const float2 m = *reinterpret_cast<const float2*>(ur);
          P[py][1] = m.x; P[py][2] = m.y;
        }
      }
      float wv[9];
#pragma unroll
      for (int q = 0; q < 9; ++q) wv[q] = sW3[(cc * CC + dc) * 9 + q];
#pragma unroll
      for (int sy = 0; sy < S; ++sy)
#pragma unroll
        for (int sx = 0; sx < S; ++sx) {
          float a = oacc[sy * S + sx];
#pragma unroll
          for (int ky = 0; ky < 3; ++ky)
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) a = fmaf(P[sy + ky][sx + kx], wv[ky * 3 + kx], a);
          oacc[sy * S + sx] = a;
        }
    }
    __syncthreads();
  }
  ut_tail_store<S>(p, oacc, dcg, b, Y0 + ply - 1, X0 + plx - 1, Hm, Wm);
}

// ---- fourth form (default at s = 4, round 4): the third form with its two phases OVERLAPPED -----------------------------------------------------------------
// k_up_tail3 runs, per chunk of four channels, the 1x1 conv on the matrix pipe (all waves) -> barrier -> the gathered 3x3 HR conv on the VALU (all waves) -> barrier:
// the matrix pipe is busy 27 % of the launch, the VALU phase 40 %, one after the other.  Here the U chunk is double-buffered and one barrier interval holds the VALU
// phase of chunk cc AND the matrix phase of chunk cc + 1; waves 0..3 take the matrix phase first, waves 4..7 the VALU phase first (wave w and w + 4 share a SIMD), so on
// every SIMD one wave's bf16 MFMAs run beside the other wave's FMAs.  One barrier per chunk instead of two.  To fit two U buffers beside the double-buffered weight
// planes the U rows lose their 16-B padding (64 floats): the sixteen 16-B pieces of row r sit at piece ^ (r & 15), which keeps the matrix phase's 16-B stores and the
// gather's 16-B reads (own pixel and both neighbours) conflict-free for the lane groups the hardware serves together; a thread's 18 read offsets are constants of the
// launch.  Same arithmetic in the same order as k_up_tail3: bit-equal.
constexpr int UR4 = 64;

__global__ __launch_bounds__(512) void k_up_tail4(UpTailArgs p) {
  constexpr int S = 4, S2 = 16, CC = 4, NCH = 16;
  constexpr int PW = UT_X + 2;
  constexpr int WPL = 64 * 64;
  extern __shared__ __attribute__((aligned(16))) float smu[];
  float* sU0 = smu;                                   // [2][UT2_ROWS][UR4]
  float* sW3 = smu + 2 * UT2_ROWS * UR4;              // [64][9]
  unsigned short* sWb = reinterpret_cast<unsigned short*>(sW3 + 64 * 9);   // [2][3 planes][8 (K step, k-group)][64 rows][8]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l15 = lane & 15, g = lane >> 4;
  const int Hm = p.A * p.h, Wm = p.A * p.w, HW = p.h * p.w;
  int t = blockIdx.x;
  const int tx = t % p.tiles_x; t /= p.tiles_x;
  const int ty = t % p.tiles_y;
  const int b = t / p.tiles_y;
  const int Y0 = ty * UT_Y, X0 = tx * UT_X;

  for (int i = tid; i < 64 * 9; i += 512) sW3[i] = p.W3[i];
  const int wr = tid >> 3, wk = tid & 7;
  auto wsrc = [&](int cc) -> const float* { int dc = wr / S2, ij = wr - dc * S2; return p.W0p + ((long long)(ij * 64 + cc * CC + dc)) * 64 + wk * 8; };
  float4 rw[2];
  auto fetch_w = [&](int cc) { const float* s0 = wsrc(cc); rw[0] = *reinterpret_cast<const float4*>(s0); rw[1] = *reinterpret_cast<const float4*>(s0 + 4); };
  auto store_w = [&](int bufi) {
    u32x4u p0, p1, p2;
    ut_split8(rw[0], rw[1], p0, p1, p2);
    unsigned short* d = sWb + bufi * 3 * WPL + (wk * 64 + wr) * 8;
    *reinterpret_cast<u32x4u*>(d) = p0; *reinterpret_cast<u32x4u*>(d + WPL) = p1; *reinterpret_cast<u32x4u*>(d + 2 * WPL) = p2;
  };
  fetch_w(0);

  u32x4u f0[4][2], f1[4][2], f2[4][2];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int u = wave + 8 * i;
    const int px = (u >> 1) * 16 + l15;
    float4 v[2][2];
#pragma unroll
    for (int s = 0; s < 2; ++s) { v[s][0] = make_float4(0.f, 0.f, 0.f, 0.f); v[s][1] = v[s][0]; }
    if (u < 26 && px < UT_PIX) {
      const int ly = px / PW, lx = px - ly * PW;
      const int Ym = Y0 + ly - 1, Xm = X0 + lx - 1;
      if (Ym >= 0 && Ym < Hm && Xm >= 0 && Xm < Wm) {
        const int uu = Ym / p.h, y = Ym - uu * p.h, vv = Xm / p.w, x = Xm - vv * p.w;
        const long long pix = ((long long)b * p.A * p.A + uu * p.A + vv) * HW + (long long)y * p.w + x;
        const float* src = p.F + pix * p.f_stride + p.f_choff + 8 * g;
#pragma unroll
        for (int s = 0; s < 2; ++s) { v[s][0] = *reinterpret_cast<const float4*>(src + 32 * s); v[s][1] = *reinterpret_cast<const float4*>(src + 32 * s + 4); }
      }
    }
#pragma unroll
    for (int s = 0; s < 2; ++s) ut_split8(v[s][0], v[s][1], f0[i][s], f1[i][s], f2[i][s]);
#pragma unroll
    for (int s = 0; s < 2; ++s) asm volatile("s_nop 4" : "+v"(f0[i][s]), "+v"(f1[i][s]), "+v"(f2[i][s]));
  }
  store_w(0);
  __syncthreads();

  // ---- the gather's constants: thread (LR pixel pp, channel dcg of the chunk); row r of U = halo pixel (ply + dy) * PW + plx + dx; piece = dcg * 4 + sy
  const int dcg = tid & 3, pp = tid >> 2;
  const int ply = 1 + pp / UT_X, plx = 1 + pp % UT_X;
  int offL[S + 2], offC[S + 2], offR[S + 2];
#pragma unroll
  for (int py = 0; py < S + 2; ++py) {
    const int dy = py == 0 ? -1 : py == S + 1 ? 1 : 0, sy = py == 0 ? S - 1 : py == S + 1 ? 0 : py - 1;
    const int rc = (ply + dy) * PW + plx, piece = dcg * 4 + sy;
    offL[py] = (rc - 1) * UR4 + ((piece ^ ((rc - 1) & 15)) << 2);
    offC[py] = rc * UR4 + ((piece ^ (rc & 15)) << 2);
    offR[py] = (rc + 1) * UR4 + ((piece ^ ((rc + 1) & 15)) << 2);
  }
  float oacc[S2];
#pragma unroll
  for (int i = 0; i < S2; ++i) oacc[i] = 0.f;

  // (1) U[px][n] of chunk cc -> sUb: D[weight row n][pixel]: lane (pixel l15, g) receives columns n = 16 ct + 4 g + r of its pixel; LeakyReLU; one 16-B store
  auto phase_mfma = [&](int cc, float* sUb) {
    const unsigned short* sWc = sWb + (cc & 1) * 3 * WPL;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int u = wave + 8 * i;
      if (u < 26) {
        const int rt = u >> 1, ch = u & 1;
        // the unit's two column tiles as two interleaved accumulator chains: consecutive MFMAs never accumulate into the same registers, and one wait covers both (a wave
        // is alone on its SIMD's matrix pipe in this phase -- its partner is in the gather -- so its own chains are all that fills the pipe: 425 -> 402 us on EPIT's launch;
        // four chains -- two units interleaved -- measured no further gain)
        f32x4u acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = acc0;
        asm volatile("s_nop 1" : "+v"(acc0), "+v"(acc1));
        const unsigned short* wq0 = sWc + (g * 64 + (2 * ch) * 16 + l15) * 8;
        const unsigned short* wq1 = wq0 + 16 * 8;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
          const u32x4u a0 = *reinterpret_cast<const u32x4u*>(wq0 + 4 * 64 * 8 * s);
          const u32x4u a1 = *reinterpret_cast<const u32x4u*>(wq0 + 4 * 64 * 8 * s + WPL);
          const u32x4u a2 = *reinterpret_cast<const u32x4u*>(wq0 + 4 * 64 * 8 * s + 2 * WPL);
          const u32x4u b0 = *reinterpret_cast<const u32x4u*>(wq1 + 4 * 64 * 8 * s);
          const u32x4u b1 = *reinterpret_cast<const u32x4u*>(wq1 + 4 * 64 * 8 * s + WPL);
          const u32x4u b2 = *reinterpret_cast<const u32x4u*>(wq1 + 4 * 64 * 8 * s + 2 * WPL);
          ut_mfma(acc0, a2, f0[i][s]); ut_mfma(acc1, b2, f0[i][s]);
          ut_mfma(acc0, a0, f2[i][s]); ut_mfma(acc1, b0, f2[i][s]);
          ut_mfma(acc0, a1, f1[i][s]); ut_mfma(acc1, b1, f1[i][s]);
          ut_mfma(acc0, a1, f0[i][s]); ut_mfma(acc1, b1, f0[i][s]);
          ut_mfma(acc0, a0, f1[i][s]); ut_mfma(acc1, b0, f1[i][s]);
          ut_mfma(acc0, a0, f0[i][s]); ut_mfma(acc1, b0, f0[i][s]);
        }
        asm volatile("s_nop 15\n\ts_nop 7" : "+v"(acc0), "+v"(acc1));
#pragma unroll
        for (int c2 = 0; c2 < 2; ++c2) {
          const f32x4u acc = c2 ? acc1 : acc0;
          const int ct = 2 * ch + c2;
          float4 o;
          o.x = acc[0] >= 0.f ? acc[0] : acc[0] * p.slope; o.y = acc[1] >= 0.f ? acc[1] : acc[1] * p.slope;
          o.z = acc[2] >= 0.f ? acc[2] : acc[2] * p.slope; o.w = acc[3] >= 0.f ? acc[3] : acc[3] * p.slope;
          *reinterpret_cast<float4*>(sUb + (rt * 16 + l15) * UR4 + ((((ct * 4 + g) ^ l15) & 15) << 2)) = o;      // row & 15 == l15
        }
      }
    }
  };
  // (3) the 3x3 HR conv of chunk cc's channel dcg, gathered per LR pixel from sUb
  auto phase_valu = [&](int cc, const float* sUb) {
    float P[S + 2][S + 2];
#pragma unroll
    for (int py = 0; py < S + 2; ++py) {
      const float4 mL = *reinterpret_cast<const float4*>(sUb + offL[py]);
      const float4 m = *reinterpret_cast<const float4*>(sUb + offC[py]);
      const float4 mR = *reinterpret_cast<const float4*>(sUb + offR[py]);
      P[py][0] = mL.w; P[py][1] = m.x; P[py][2] = m.y; P[py][3] = m.z; P[py][4] = m.w; P[py][5] = mR.x;
    }
    float wv[9];
#pragma unroll
    for (int q = 0; q < 9; ++q) wv[q] = sW3[(cc * CC + dcg) * 9 + q];
#pragma unroll
    for (int sy = 0; sy < S; ++sy)
#pragma unroll
      for (int sx = 0; sx < S; ++sx) {
        float a = oacc[sy * S + sx];
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
          for (int kx = 0; kx < 3; ++kx) a = fmaf(P[sy + ky][sx + kx], wv[ky * 3 + kx], a);
        oacc[sy * S + sx] = a;
      }
  };

  // prologue interval: the matrix phase of chunk 0; the weights of chunk 1
  fetch_w(1);
  phase_mfma(0, sU0);
  store_w(1);
  __syncthreads();
  const bool mfma_first = wave < 4;
#pragma unroll 1
  for (int cc = 0; cc < NCH; ++cc) {
    float* const sUc = sU0 + (cc & 1) * UT2_ROWS * UR4;
    float* const sUn = sU0 + ((cc + 1) & 1) * UT2_ROWS * UR4;
    if (cc + 2 < NCH) fetch_w(cc + 2);
    if (mfma_first) {
      if (cc + 1 < NCH) phase_mfma(cc + 1, sUn);
      phase_valu(cc, sUc);
    } else {
      phase_valu(cc, sUc);
      if (cc + 1 < NCH) phase_mfma(cc + 1, sUn);
    }
    if (cc + 2 < NCH) store_w(cc & 1);       // (the buffer of chunk cc's weights: last read in the previous interval)
    __syncthreads();
  }
  ut_tail_store<S>(p, oacc, dcg, b, Y0 + ply - 1, X0 + plx - 1, Hm, Wm);
}

}  // namespace

extern "C" int lfsr_up_tail_fwd(const float* f, int f_stride, int f_choff, const float* w0_packed, const float* w3, const float* x_lr, float* out,
                                int B, int A, int h, int w, int s, float slope, void* stream) {
  LfsrOpTimer op_t("up_tail", s, B, lfsr_stream(stream));
  if (!f || !w0_packed || !w3 || !x_lr || !out || B <= 0 || A <= 0 || h <= 0 || w <= 0 || (s != 2 && s != 4)) return LFSR_E_ARG;
  if (f_stride < f_choff + 64 || (f_stride | f_choff) & 3) return LFSR_E_ARG;
  const int smem = (2 * UT_ROWS * LDS_ROW + 2 * 64 * LDS_ROW + 64 * 9) * 4;
  static std::atomic<bool> attr_set[64];
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return LFSR_E_ARG;
  if (!attr_set[dev]) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_up_tail<2>), hipFuncAttributeMaxDynamicSharedMemorySize, smem);
    if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_up_tail<4>), hipFuncAttributeMaxDynamicSharedMemorySize, smem);
    if (e != hipSuccess) return LFSR_HIP_ERR(e);
    attr_set[dev] = true;
  }
  UpTailArgs p{};
  p.F = f; p.f_stride = f_stride; p.f_choff = f_choff; p.W0p = w0_packed; p.W3 = w3; p.Xlr = x_lr; p.Out = out;
  p.B = B; p.A = A; p.h = h; p.w = w; p.slope = slope;
  p.tiles_y = (A * h + UT_Y - 1) / UT_Y; p.tiles_x = (A * w + UT_X - 1) / UT_X;
  long long grid = (long long)B * p.tiles_y * p.tiles_x;
  if (grid > 0x7fffffffLL) return LFSR_E_ARG;
  const char* usel = lfsr_sel("LFSR_UPTAIL");         // "v1" / "v2": the first / second (fp32-MFMA) forms (A/B runs)
  if (!(usel && usel[0] == 'v') && !lfsr_arith_f32()) {       // (LFSR_UPTAIL=3: handled below)
    const int smem3 = (UT2_ROWS * LDS_ROW + 64 * 9) * 4 + 2 * 3 * 64 * 64 * 2;
    static std::atomic<bool> attr3_set[64];
    if (!attr3_set[dev]) {
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_up_tail3<2>), hipFuncAttributeMaxDynamicSharedMemorySize, smem3);
      if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_up_tail3<4>), hipFuncAttributeMaxDynamicSharedMemorySize, smem3);
      if (e != hipSuccess) return LFSR_HIP_ERR(e);
      attr3_set[dev] = true;
    }
    if (((uintptr_t)f | (uintptr_t)w0_packed) & 15) return LFSR_E_ARG;
    const bool v3 = usel && usel[0] == '3';             // LFSR_UPTAIL=3 (lab): the third form at s = 4 too
    if (s == 4 && !v3) {
      const int smem4 = (2 * UT2_ROWS * UR4 + 64 * 9) * 4 + 2 * 3 * 64 * 64 * 2;      // 157952
      static std::atomic<bool> attr4_set[64];
      if (!attr4_set[dev]) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_up_tail4), hipFuncAttributeMaxDynamicSharedMemorySize, smem4);
        if (e != hipSuccess) return LFSR_HIP_ERR(e);
        attr4_set[dev] = true;
      }
      hipLaunchKernelGGL(k_up_tail4, dim3((unsigned)grid), dim3(512), smem4, lfsr_stream(stream), p);
      LFSR_CHECK_LAUNCH();
      return LFSR_OK;
    }
    if (s == 2) hipLaunchKernelGGL((k_up_tail3<2>), dim3((unsigned)grid), dim3(512), smem3, lfsr_stream(stream), p);
    else hipLaunchKernelGGL((k_up_tail3<4>), dim3((unsigned)grid), dim3(512), smem3, lfsr_stream(stream), p);
    LFSR_CHECK_LAUNCH();
    return LFSR_OK;
  }
  if (!(usel && usel[1] == '1')) {
    const int smem2 = (2 * UT2_ROWS * LDS_ROW + 2 * 64 * LDS_ROW + 64 * 9) * 4;
    static std::atomic<bool> attr2_set[64];
    if (!attr2_set[dev]) {
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_up_tail2<2>), hipFuncAttributeMaxDynamicSharedMemorySize, smem2);
      if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_up_tail2<4>), hipFuncAttributeMaxDynamicSharedMemorySize, smem2);
      if (e != hipSuccess) return LFSR_HIP_ERR(e);
      attr2_set[dev] = true;
    }
    if (s == 2) hipLaunchKernelGGL((k_up_tail2<2>), dim3((unsigned)grid), dim3(512), smem2, lfsr_stream(stream), p);
    else hipLaunchKernelGGL((k_up_tail2<4>), dim3((unsigned)grid), dim3(512), smem2, lfsr_stream(stream), p);
    LFSR_CHECK_LAUNCH();
    return LFSR_OK;
  }
  if (s == 2) hipLaunchKernelGGL((k_up_tail<2>), dim3((unsigned)grid), dim3(512), smem, lfsr_stream(stream), p);
  else hipLaunchKernelGGL((k_up_tail<4>), dim3((unsigned)grid), dim3(512), smem, lfsr_stream(stream), p);
  LFSR_CHECK_LAUNCH();
  return LFSR_OK;
}
