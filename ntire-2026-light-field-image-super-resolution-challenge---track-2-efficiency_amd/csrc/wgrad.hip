// Weight gradients of the gather-GEMM operator family (backward of model/SR/DistgSSR.py:78-111, i.e. what
// autograd computes for every nn.Conv2d weight at train.py:264), fp32 MFMA:
//
//     dW[tap][n][k] = sum_m  G[gsrc(m,tap)][n] * X[xsrc(m,tap)][k]
//
// G = gradient rows (N <= 64 channels), X = saved forward input rows (K channels), both gathered with the same
// index maps as the forward (gemm_gather_kernel.h).  The reduction runs over ALL rows m, so the grid is
// (row splits, taps, 64-wide k tiles): each 256-thread block reduces its row range for one tap into a 64x64
// accumulator tile (4 waves = 2x2 quadrants of 32x32, "m" is the MFMA's K dimension, two rows per
// v_mfma_f32_32x32x2_f32), and writes a PARTIAL [Npad][K] slab.  A second, deterministic pass sums the slabs in
// split order and scatters into the PyTorch (O, C, kh, kw) layout (no float atomics: bitwise reproducible).
#include "gemm_gather_kernel.h"

namespace {

constexpr int WG_ROWS = 64;   // rows staged per step

struct WgradArgs {
  const float* G; int g_stride; int g_choff;
  const float* X; int x_stride; int x_choff;
  float* P;                 // partials [nsplit][ntaps][Npad][K]
  int M, N, Npad, K;
  int A, AA, H, W, ntaps;
  int rows_per_split, nsplit;
};

template <int GIN, int XIN>
__global__ __launch_bounds__(256) void k_wgrad(WgradArgs p) {
  __shared__ __attribute__((aligned(16))) float sG[WG_ROWS * LDS_ROW];
  __shared__ __attribute__((aligned(16))) float sX[WG_ROWS * LDS_ROW];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int c16 = tid & 15, r0 = tid >> 4;          // 16 threads per row, 16 rows per pass, 4 passes
  const int split = blockIdx.x, tap = blockIdx.y, k0 = blockIdx.z * 64;
  const int nq = wave >> 1, kq = wave & 1;
  const int half = lane >> 5, l31 = lane & 31;

  GemmArgs ga{};   // index maps only
  ga.M = p.M; ga.A = p.A; ga.AA = p.AA; ga.H = p.H; ga.W = p.W;

  const int m_begin = split * p.rows_per_split;
  const int m_end = min(p.M, m_begin + p.rows_per_split);

  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;

  float4 rg[4], rx[4];
  auto prefetch = [&](int m0) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      int m = m0 + r0 + 16 * i;
      rg[i] = make_float4(0.f, 0.f, 0.f, 0.f);
      rx[i] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (m < m_end) {
        if (c16 * 4 < p.Npad && c16 * 4 < p.N) {   // N is a multiple of 4 (16/32/64)
          RowInfo<GIN> ri = decode_row<GIN>(m, ga);
          int sp = src_pixel<GIN>(ri, tap, ga);
          if (sp >= 0) rg[i] = *reinterpret_cast<const float4*>(p.G + (long long)sp * p.g_stride + p.g_choff + c16 * 4);
        }
        if (k0 + c16 * 4 < p.K) {
          RowInfo<XIN> ri = decode_row<XIN>(m, ga);
          int sp = src_pixel<XIN>(ri, tap, ga);
          if (sp >= 0) rx[i] = *reinterpret_cast<const float4*>(p.X + (long long)sp * p.x_stride + p.x_choff + k0 + c16 * 4);
        }
      }
    }
  };

  const bool active = nq * 32 < p.Npad && k0 + kq * 32 < p.K;
  prefetch(m_begin);
  for (int m0 = m_begin; m0 < m_end; m0 += WG_ROWS) {
    if (m0 > m_begin) __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      *reinterpret_cast<float4*>(sG + (r0 + 16 * i) * LDS_ROW + c16 * 4) = rg[i];
      *reinterpret_cast<float4*>(sX + (r0 + 16 * i) * LDS_ROW + c16 * 4) = rx[i];
    }
    __syncthreads();
    if (m0 + WG_ROWS < m_end) prefetch(m0 + WG_ROWS);
    if (active) {
      const float* gp = sG + half * LDS_ROW + nq * 32 + l31;
      const float* xp = sX + half * LDS_ROW + kq * 32 + l31;
#pragma unroll 8
      for (int mm = 0; mm < WG_ROWS; mm += 2)
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(gp[mm * LDS_ROW], xp[mm * LDS_ROW], acc, 0, 0, 0);
    }
  }
  if (active) {
    // D[row = n][col = k]: col = lane&31, row = (r&3) + 8*(r>>2) + 4*half
    float* out = p.P + (((long long)split * p.ntaps + tap) * p.Npad) * p.K;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      int n = nq * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
      int k = k0 + kq * 32 + l31;
      if (n < p.Npad && k < p.K) out[(long long)n * p.K + k] = acc[r];
    }
  }
}


// ---- 3x3 conv weight gradient, halo-tile form ------------------------------------------------------------------------------
// dW[tap][n][k] = sum over pixels p of dY[p][n] * X[p + shift(tap)][k] (zero outside the view).  The gather form above re-reads
// dY and the shifted X for every tap (9x the traffic).  Here a persistent 512-thread block walks 4-row x 32-column tiles: the
// dY tile and the X halo are staged in LDS once and all 9 taps read X at shifted addresses.  The pixel index is the MFMA's K
// dimension (two pixels per v_mfma_f32_32x32x2_f32).  Wave w owns the (n,k) quadrant w&3 for taps 0-4 (w<4) or 5-8 (w>=4):
// 5 x 16 accumulator registers that persist across ALL tiles of the block, so each block writes ONE partial slab at the end.
constexpr int WT_R = 4, WT_C = 32;
constexpr int WX_PIX = (WT_R + 2) * (WT_C + 2);   // 204
constexpr int WG_PIX = WT_R * WT_C;               // 128

struct Wgrad3Args {
  const float* G; int g_stride; int g_choff;
  const float* X; int x_stride; int x_choff;
  float* P;               // [gridDim.x][9][64][64]
  int g_bytes, x_bytes;   // true byte spans (descriptor extents)
  int n_img, H, W, tiles_y, tiles_x;
};

__global__ __launch_bounds__(512) void k_wgrad_conv3_halo(Wgrad3Args p) {
  extern __shared__ __attribute__((aligned(16))) float smw[];
  float* sX = smw;                         // [WX_PIX][LDS_ROW]
  float* sG = smw + WX_PIX * LDS_ROW;      // [WG_PIX][LDS_ROW]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int c16 = tid & 15, r16 = tid >> 4;
  const int half = lane >> 5, l31 = lane & 31;
  const int q = wave & 3, nq = q >> 1, kq = q & 1, t0 = (wave >> 2) ? 5 : 0, ntap = (wave >> 2) ? 4 : 5;
  const int ntiles = p.n_img * p.tiles_y * p.tiles_x;

  f32x16 acc[5];
#pragma unroll
  for (int t = 0; t < 5; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

  // tile loads through buffer descriptors with 32-bit offsets (the launcher checks the spans): image borders and ragged edges
  // are out-of-range offsets that read as zero instead of per-pixel branches with 64-bit addresses
  typedef float f32x4w __attribute__((ext_vector_type(4)));
  constexpr int WOOB = (int)0x80000000u;
  const __amdgpu_buffer_rsrc_t rsX = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.X), 0, p.x_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsG = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.G), 0, p.g_bytes, 0x00020000);
  float4 hx[7], hg[4];
  auto load_tile = [&](int tile) {
    int tx = tile % p.tiles_x; int qq = tile / p.tiles_x;
    int ty = qq % p.tiles_y; int img = qq / p.tiles_y;
    const int y0 = ty * WT_R, x0 = tx * WT_C;
    const int ib = img * p.H * p.W;
#pragma unroll
    for (int i = 0; i < 7; ++i) {
      const int pix = r16 + 32 * i;
      const int r = pix / (WT_C + 2), c = pix - r * (WT_C + 2);
      const int yy = y0 + r - 1, xx = x0 + c - 1;
      const bool ok = pix < WX_PIX && (unsigned)yy < (unsigned)p.H && (unsigned)xx < (unsigned)p.W;
      const f32x4w v = __builtin_bit_cast(f32x4w, __builtin_amdgcn_raw_buffer_load_b128(rsX, ok ? ((ib + yy * p.W + xx) * p.x_stride + p.x_choff + c16 * 4) * 4 : WOOB, 0, 0));
      hx[i] = make_float4(v.x, v.y, v.z, v.w);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int pix = r16 + 32 * i;
      const int r = pix / WT_C, c = pix - r * WT_C;
      const int yy = y0 + r, xx = x0 + c;
      const bool ok = yy < p.H && xx < p.W;
      const f32x4w v = __builtin_bit_cast(f32x4w, __builtin_amdgcn_raw_buffer_load_b128(rsG, ok ? ((ib + yy * p.W + xx) * p.g_stride + p.g_choff + c16 * 4) * 4 : WOOB, 0, 0));
      hg[i] = make_float4(v.x, v.y, v.z, v.w);
    }
  };

  int tile = blockIdx.x;
  if (tile < ntiles) load_tile(tile);
  for (; tile < ntiles; tile += gridDim.x) {
#pragma unroll
    for (int i = 0; i < 7; ++i) {
      int pix = r16 + 32 * i;
      if (pix < WX_PIX) *reinterpret_cast<float4*>(sX + pix * LDS_ROW + c16 * 4) = hx[i];
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) *reinterpret_cast<float4*>(sG + (r16 + 32 * i) * LDS_ROW + c16 * 4) = hg[i];
    __syncthreads();
    if (tile + (int)gridDim.x < ntiles) load_tile(tile + gridDim.x);     // next tile flies under this tile's MFMAs
    const float* gp = sG + nq * 32 + l31;
    const float* xp = sX + kq * 32 + l31;
#pragma unroll 4
    for (int s = 0; s < WG_PIX / 2; ++s) {
      const int pp = 2 * s + half, r = pp / WT_C, c = pp - r * WT_C;
      const float a = gp[pp * LDS_ROW];
      const float* xc = xp + ((r + 1) * (WT_C + 2) + (c + 1)) * LDS_ROW;
#pragma unroll
      for (int t = 0; t < 5; ++t) {
        if (t < ntap) {
          const int tap = t0 + t, dy = tap / 3 - 1, dx = tap - (tap / 3) * 3 - 1;
          acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, xc[(dy * (WT_C + 2) + dx) * LDS_ROW], acc[t], 0, 0, 0);
        }
      }
    }
    __syncthreads();
  }
  // D[row = n][col = k]: col = lane&31, row = (r&3) + 8*(r>>2) + 4*half
  float* out = p.P + (long long)blockIdx.x * 9 * 64 * 64;
#pragma unroll
  for (int t = 0; t < 5; ++t) {
    if (t < ntap) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        int n = nq * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
        out[((long long)(t0 + t) * 64 + n) * 64 + kq * 32 + l31] = acc[t][r];
      }
    }
  }
}

// ---- 3x3 conv weight gradient in Winograd form: the adjoint of F(2x2, 3x3) -------------------------------------------------------
//   forward   Y = At [ (G g Gt) . (Bt d B) ] A            per 2x2 output tile, d = its 4x4 input patch
//   adjoint   dU[a][b] += (A dY At)[a][b] . (Bt d B)[a][b]  summed over tiles and images,   dW = Gt dU G
// 16 products per 2x2 outputs and channel pair instead of 36: 2.25x fewer MFMA flops than the direct form above (coefficients 0, +-1, +-1/2:
// exact transforms, well-conditioned).  The 16 position-GEMMs dU[p][n][k] += sum_tiles dY'[p][tile][n] V[p][tile][k] have the TILE index as
// the MFMA K dimension (v_mfma_f32_16x16x4_f32: four tiles per instruction).  A persistent 512-thread block walks 2-row x 32-column pixel
// tiles (16 Winograd tiles = 4 K steps); wave (a, nh) owns transform row a (positions (a, 0..3)) of the n half nh for all 64 k:
// 4 x 2 x 4 accumulators of 4 registers = 128, alive across ALL tiles of the block.  The transformed operands never touch LDS: lane
// (channel l15, tile g of the K step) forms its own A / B operand values from the raw pixels staged in LDS (double-buffered, one barrier
// per tile): dY' from 2-4 raw values, V from 8, a few adds each -- 38 ds_read_b32 + 40 VALU per 32 MFMAs.  Epilogue: Gt dU G, the b
// contraction in registers, the a contraction across the four waves of an n half through LDS -> ONE [9][64][64] partial slab per block,
// summed by k_wgrad_reduce exactly as the direct kernel's.
typedef float f32x4 __attribute__((ext_vector_type(4)));
#ifndef W2_UNROLL
#define W2_UNROLL 1     // K steps of a tile unrolled together (operand preparation of one step beside the MFMAs of another)
#endif
constexpr int W2ROW = 72;                        // floats per staged pixel: two pixels = 144 = 16 banks mod 64 -> the four tiles of a K step read disjoint banks
constexpr int W2XPIX = 4 * 34, W2GPIX = 2 * 32;
constexpr int W2BUF = (W2XPIX + W2GPIX) * W2ROW;  // floats per buffer

template <int TA>   // the wave's transform row a
__device__ __forceinline__ void wgrad_wino_body(const Wgrad3Args& p, float* smw, f32x4 (&acc)[4][2][4], const int nh) {
  const int tid = threadIdx.x, lane = tid & 63;
  const int l15 = lane & 15, g = lane >> 4;
  const int ntiles = p.n_img * p.tiles_y * p.tiles_x;
  typedef float f32x4w __attribute__((ext_vector_type(4)));
  constexpr int WOOB = (int)0x80000000u;
  const __amdgpu_buffer_rsrc_t rsX = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.X), 0, p.x_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsG = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.G), 0, p.g_bytes, 0x00020000);
  float4 hx[5], hg[2];
  auto load_tile = [&](int tile) {
    int tx = tile % p.tiles_x; int qq = tile / p.tiles_x;
    int ty = qq % p.tiles_y; int img = qq / p.tiles_y;
    const int y0 = ty * 2, x0 = tx * 32;
    const int ib = img * p.H * p.W;
#pragma unroll
    for (int i = 0; i < 5; ++i) {
      const int idx = tid + 512 * i, pix = idx >> 4, c16 = idx & 15;
      const int r = pix / 34, c = pix - r * 34;
      const int yy = y0 + r - 1, xx = x0 + c - 1;
      const bool ok = pix < W2XPIX && (unsigned)yy < (unsigned)p.H && (unsigned)xx < (unsigned)p.W;
      const f32x4w v = __builtin_bit_cast(f32x4w, __builtin_amdgcn_raw_buffer_load_b128(rsX, ok ? ((ib + yy * p.W + xx) * p.x_stride + p.x_choff + c16 * 4) * 4 : WOOB, 0, 0));
      hx[i] = make_float4(v.x, v.y, v.z, v.w);
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int idx = tid + 512 * i, pix = idx >> 4, c16 = idx & 15;
      const int r = pix >> 5, c = pix & 31;
      const int yy = y0 + r, xx = x0 + c;
      const bool ok = yy < p.H && xx < p.W;
      const f32x4w v = __builtin_bit_cast(f32x4w, __builtin_amdgcn_raw_buffer_load_b128(rsG, ok ? ((ib + yy * p.W + xx) * p.g_stride + p.g_choff + c16 * 4) * 4 : WOOB, 0, 0));
      hg[i] = make_float4(v.x, v.y, v.z, v.w);
    }
  };
  int tile = blockIdx.x, it = 0;
  if (tile < ntiles) load_tile(tile);
  for (; tile < ntiles; tile += gridDim.x, ++it) {
    float* const sX = smw + (it & 1) * W2BUF;
    float* const sG = sX + W2XPIX * W2ROW;
#pragma unroll
    for (int i = 0; i < 5; ++i) {
      const int idx = tid + 512 * i, pix = idx >> 4, c16 = idx & 15;
      if (pix < W2XPIX) *reinterpret_cast<float4*>(sX + pix * W2ROW + c16 * 4) = hx[i];
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int idx = tid + 512 * i, pix = idx >> 4, c16 = idx & 15;
      *reinterpret_cast<float4*>(sG + pix * W2ROW + c16 * 4) = hg[i];
    }
    __syncthreads();   // (the only barrier of a tile: the other buffer was last read before the previous tile's barrier)
    if (tile + (int)gridDim.x < ntiles) load_tile(tile + gridDim.x);     // next tile flies under this tile's MFMAs
    // K steps software-pipelined: the raw LDS reads of step s + 1 are issued before the MFMAs of step s (their latency runs under ~1000 MFMA cycles)
    float rawg[2][2][4], rawa[2][4][4], rawb[2][4][4];
    auto load_raw = [&](int s, int buf) {
      const int cx = 2 * (4 * s + g);           // first pixel column of this lane's Winograd tile (= its first halo column)
#pragma unroll
      for (int nb = 0; nb < 2; ++nb) {
        const float* gp = sG + cx * W2ROW + (2 * nh + nb) * 16 + l15;
        if (TA == 0) { rawg[buf][nb][0] = gp[0]; rawg[buf][nb][1] = gp[W2ROW]; }
        else if (TA == 3) { rawg[buf][nb][0] = gp[32 * W2ROW]; rawg[buf][nb][1] = gp[33 * W2ROW]; }
        else { rawg[buf][nb][0] = gp[0]; rawg[buf][nb][1] = gp[W2ROW]; rawg[buf][nb][2] = gp[32 * W2ROW]; rawg[buf][nb][3] = gp[33 * W2ROW]; }
      }
      constexpr int RA = TA == 0 ? 0 : 1, RB = TA == 0 ? 2 : TA == 3 ? 3 : 2;   // the two patch rows transform row a combines
#pragma unroll
      for (int kb = 0; kb < 4; ++kb) {
        const float* xa = sX + (RA * 34 + cx) * W2ROW + kb * 16 + l15;
        const float* xb = sX + (RB * 34 + cx) * W2ROW + kb * 16 + l15;
#pragma unroll
        for (int j = 0; j < 4; ++j) { rawa[buf][kb][j] = xa[j * W2ROW]; rawb[buf][kb][j] = xb[j * W2ROW]; }
      }
    };
    load_raw(0, 0);
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const int buf = s & 1;
      // A operands: dY'[a][b] of channels n = (2 nh + nb) 16 + l15
      float av[2][4];
#pragma unroll
      for (int nb = 0; nb < 2; ++nb) {
        float r0, r1;
        if (TA == 0) { r0 = rawg[buf][nb][0]; r1 = rawg[buf][nb][1]; }
        else if (TA == 3) { r0 = -rawg[buf][nb][0]; r1 = -rawg[buf][nb][1]; }
        else {
          r0 = TA == 1 ? rawg[buf][nb][0] + rawg[buf][nb][2] : rawg[buf][nb][0] - rawg[buf][nb][2];
          r1 = TA == 1 ? rawg[buf][nb][1] + rawg[buf][nb][3] : rawg[buf][nb][1] - rawg[buf][nb][3];
        }
        av[nb][0] = r0; av[nb][1] = r0 + r1; av[nb][2] = r0 - r1; av[nb][3] = -r1;
      }
      // B operands: V[a][b] of channels k = kb 16 + l15
      float bv[4][4];
#pragma unroll
      for (int kb = 0; kb < 4; ++kb) {
        float t[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float u = rawa[buf][kb][j], v = rawb[buf][kb][j];
          t[j] = TA == 0 ? u - v : TA == 1 ? u + v : TA == 2 ? v - u : u - v;   // a = 0: d0 - d2; 1: d1 + d2; 2: d2 - d1; 3: d1 - d3
        }
        bv[kb][0] = t[0] - t[2]; bv[kb][1] = t[1] + t[2]; bv[kb][2] = t[2] - t[1]; bv[kb][3] = t[1] - t[3];
      }
      if (s < 3) load_raw(s + 1, buf ^ 1);
      __builtin_amdgcn_sched_barrier(0);        // (keeps the next step's reads in front of this step's MFMAs)
#pragma unroll
      for (int b = 0; b < 4; ++b)
#pragma unroll
        for (int nb = 0; nb < 2; ++nb)
#pragma unroll
          for (int kb = 0; kb < 4; ++kb)
            acc[b][nb][kb] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[nb][b], bv[kb][b], acc[b][nb][kb], 0, 0, 0);
    }
  }
}

__global__ __launch_bounds__(512) void k_wgrad_conv3_wino(Wgrad3Args p) {
  extern __shared__ __attribute__((aligned(16))) float smw[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l15 = lane & 15, g = lane >> 4;
  const int a = __builtin_amdgcn_readfirstlane(wave & 3), nh = __builtin_amdgcn_readfirstlane(wave >> 2);
  f32x4 acc[4][2][4];
#pragma unroll
  for (int b = 0; b < 4; ++b)
#pragma unroll
    for (int nb = 0; nb < 2; ++nb)
#pragma unroll
      for (int kb = 0; kb < 4; ++kb) acc[b][nb][kb] = f32x4{0.f, 0.f, 0.f, 0.f};
  if (a == 0) wgrad_wino_body<0>(p, smw, acc, nh);
  else if (a == 1) wgrad_wino_body<1>(p, smw, acc, nh);
  else if (a == 2) wgrad_wino_body<2>(p, smw, acc, nh);
  else wgrad_wino_body<3>(p, smw, acc, nh);
  // ---- dW = Gt dU G.  b contraction in registers: e[j] = sum_b G[b][j] dU[a][b]  (G = [1 0 0; 1/2 1/2 1/2; 1/2 -1/2 1/2; 0 0 1])
  __syncthreads();                                   // the staging buffers are dead: reuse them for the exchange
  float* out = p.P + (long long)blockIdx.x * 9 * 64 * 64;
  float* ex = smw;                                   // [wave 8][8 blocks][4 regs][64 lanes] = 64 KB per round
#pragma unroll 1
  for (int j = 0; j < 3; ++j) {
#pragma unroll
    for (int nb = 0; nb < 2; ++nb)
#pragma unroll
      for (int kb = 0; kb < 4; ++kb) {
        const f32x4 u0 = acc[0][nb][kb], u1 = acc[1][nb][kb], u2 = acc[2][nb][kb], u3 = acc[3][nb][kb];
        f32x4 e;
        if (j == 0) e = u0 + 0.5f * (u1 + u2);
        else if (j == 1) e = 0.5f * (u1 - u2);
        else e = 0.5f * (u1 + u2) + u3;
        *reinterpret_cast<f32x4*>(ex + (((wave * 8 + nb * 4 + kb) * 64 + lane) << 2)) = e;
      }
    __syncthreads();
    // a contraction: dW[i][j] = sum_a G[a][i] e_a[j]; 3 x 2 n halves x 8 blocks x 256 values = 12288 float4-free scalars per round over 512 threads
    for (int idx = tid; idx < 3 * 2 * 8 * 64; idx += 512) {
      const int ln = idx & 63, blk = (idx >> 6) & 7, nh2 = (idx >> 9) & 1, i = idx >> 10;
      const f32x4 e0 = *reinterpret_cast<const f32x4*>(ex + ((((nh2 * 4 + 0) * 8 + blk) * 64 + ln) << 2));
      const f32x4 e1 = *reinterpret_cast<const f32x4*>(ex + ((((nh2 * 4 + 1) * 8 + blk) * 64 + ln) << 2));
      const f32x4 e2 = *reinterpret_cast<const f32x4*>(ex + ((((nh2 * 4 + 2) * 8 + blk) * 64 + ln) << 2));
      const f32x4 e3 = *reinterpret_cast<const f32x4*>(ex + ((((nh2 * 4 + 3) * 8 + blk) * 64 + ln) << 2));
      f32x4 d;
      if (i == 0) d = e0 + 0.5f * (e1 + e2);
      else if (i == 1) d = 0.5f * (e1 - e2);
      else d = 0.5f * (e1 + e2) + e3;
      // D layout of the 16x16 block (nb, kb): lane ln holds column k = ln & 15, rows n = 4 (ln >> 4) + r
      const int nb = blk >> 2, kb = blk & 3;
      const int k = kb * 16 + (ln & 15), n0 = (2 * nh2 + nb) * 16 + 4 * (ln >> 4);
#pragma unroll
      for (int r = 0; r < 4; ++r) out[((long long)(3 * i + j) * 64 + n0 + r) * 64 + k] = d[r];
    }
    __syncthreads();
  }
  (void)l15; (void)g;
}

// ---- EPIConv.0 weight gradient, EPI-line form ---------------------------------------------------------------------------------------
// dW[tap = A dx + v'][n][c] = sum over the EPI lines (b, u, y) [vertical: (b, v, x)] and their pixels x of dE[line, x][n] * X[view v', x + dx - pad][c]
// (DistgSSR.py:91-97: a 1 x A^2 conv, stride A, over the MacPI = a 5-tap 1-D conv along the line over A views x 64 channels).  The gather form
// (k_wgrad<IN_SAME, IN_EPIH/V>) runs one block per tap and re-reads dE and the gathered rows for each of the A^2 taps.  Here a persistent
// 512-thread block walks EPI lines (<= 32 pixels): the line's dE (32 x 32) and its A x 36 x 64 input slab (two pixels of zero halo) are
// staged in LDS once (double-buffered, one barrier per line) and all A^2 taps read the slab at shifted addresses.  The PIXEL index is the MFMA K
// dimension (v_mfma_f32_16x16x4_f32); wave (nh, cq) owns the 16 x 16 block (n half, channel quarter) of all 25 taps: 100 accumulator registers alive
// across ALL lines of the block, one [A^2][32][64] partial slab per block at the end.  Pixel row strides 80 / 48 floats: the four pixels of a K
// step read disjoint banks.
constexpr int WE_SI = 80, WE_SE = 48, WE_PX = 36;

struct WgradEpiArgs {
  const float* G2;           // (vert == 2: both passes in one launch) the vertical pass's dE; lines [0, nH) are horizontal (G), the rest vertical (G2)
  const float* G;            // dE rows ((q h + y) w + x), 32 channels
  const float* X; int x_stride; int x_choff;
  float* P;                  // [gridDim.x][A*A][32][64]
  int g_bytes, x_bytes;
  int B, A, H, W, vert;
};

template <int A>
__global__ __launch_bounds__(512) void k_wgrad_epi0_lines(WgradEpiArgs p) {
  extern __shared__ __attribute__((aligned(16))) float sme[];
  constexpr int BUF = A * WE_PX * WE_SI + 32 * WE_SE;   // floats per buffer
  constexpr int NIN = (A * WE_PX * 16 + 511) / 512;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l15 = lane & 15, g = lane >> 4;
  const int nh = wave >> 2, cq = wave & 3;
  const int HW = p.H * p.W;
  const int nH = p.vert == 1 ? 0 : p.B * A * p.H, nV = p.vert == 0 ? 0 : p.B * A * p.W;   // horizontal lines (b, u, y), vertical lines (b, v, x)
  const int nlines = nH + nV;
  typedef float f32x4w __attribute__((ext_vector_type(4)));
  constexpr int WOOB = (int)0x80000000u;
  const __amdgpu_buffer_rsrc_t rsX = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.X), 0, p.x_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsG = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.G), 0, p.g_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsG2 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.vert == 2 ? p.G2 : p.G), 0, p.g_bytes, 0x00020000);
  f32x4 acc[A * A];
#pragma unroll
  for (int t = 0; t < A * A; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
  float4 hin[NIN], he;
  auto load_line = [&](int line) {
    const bool vert = line >= nH;                     // wave-uniform
    const int ln = vert ? line - nH : line;
    const int len = vert ? p.H : p.W, across = vert ? p.W : p.H;
    const int vstride = vert ? A * HW : HW, pstride = vert ? p.W : 1;
    const int q = ln / across, o = ln - q * across;
    int base, mbase;   // first pixel of the line in X (view 0); first dE row of the line
    if (!vert) { base = q * A * HW + o * p.W; mbase = (q * p.H + o) * p.W; }
    else { const int b = q / A, v = q - b * A; base = (b * A * A + v) * HW + o; mbase = q * HW + o; }
#pragma unroll
    for (int i = 0; i < NIN; ++i) {
      const int idx = tid + 512 * i, c16 = idx & 15, pv = idx >> 4;
      const int vv = pv / WE_PX, t = pv - vv * WE_PX - 2;
      const bool ok = pv < A * WE_PX && t >= 0 && t < len;
      const f32x4w v = __builtin_bit_cast(f32x4w, __builtin_amdgcn_raw_buffer_load_b128(rsX, ok ? ((base + vv * vstride + t * pstride) * p.x_stride + p.x_choff + c16 * 4) * 4 : WOOB, 0, 0));
      hin[i] = make_float4(v.x, v.y, v.z, v.w);
    }
    {
      const int px = tid >> 3, c8 = tid & 7;
      const bool ok = tid < 256 && px < len;
      const int off = ok ? ((mbase + px * pstride) * 32 + c8 * 4) * 4 : WOOB;
      const f32x4w v = vert ? __builtin_bit_cast(f32x4w, __builtin_amdgcn_raw_buffer_load_b128(rsG2, off, 0, 0))
                            : __builtin_bit_cast(f32x4w, __builtin_amdgcn_raw_buffer_load_b128(rsG, off, 0, 0));
      he = make_float4(v.x, v.y, v.z, v.w);
    }
  };
  int line = blockIdx.x, it = 0;
  if (line < nlines) load_line(line);
  for (; line < nlines; line += gridDim.x, ++it) {
    float* const sIn = sme + (it & 1) * BUF;
    float* const sE = sIn + A * WE_PX * WE_SI;
#pragma unroll
    for (int i = 0; i < NIN; ++i) {
      const int idx = tid + 512 * i, c16 = idx & 15, pv = idx >> 4;
      if (pv < A * WE_PX) *reinterpret_cast<float4*>(sIn + pv * WE_SI + c16 * 4) = hin[i];
    }
    if (tid < 256) *reinterpret_cast<float4*>(sE + (tid >> 3) * WE_SE + (tid & 7) * 4) = he;
    __syncthreads();   // (the only barrier of a line: the other buffer was last read before the previous line's barrier)
    if (line + (int)gridDim.x < nlines) load_line(line + gridDim.x);
    const float* ep = sE + g * WE_SE + nh * 16 + l15;
    const float* ip = sIn + g * WE_SI + cq * 16 + l15;
#pragma unroll 2
    for (int s = 0; s < 8; ++s) {
      const float a = ep[4 * s * WE_SE];
#pragma unroll
      for (int dx = 0; dx < A; ++dx)
#pragma unroll
        for (int vv = 0; vv < A; ++vv)
          acc[dx * A + vv] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, ip[(vv * WE_PX + 4 * s + dx) * WE_SI], acc[dx * A + vv], 0, 0, 0);
    }
  }
  // D of block (nh, cq): lane holds column c = cq 16 + l15, rows n = nh 16 + 4 g + r
  float* out = p.P + (long long)blockIdx.x * A * A * 32 * 64;
#pragma unroll
  for (int t = 0; t < A * A; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) out[((long long)t * 32 + nh * 16 + 4 * g + r) * 64 + cq * 16 + l15] = acc[t][r];
}

// ---- 1x1 weight gradient of fuse.0 (64 outputs, K = 144 inputs), streaming form --------------------------------------------------------------
// dW[n][k] = sum_m dY[m][n] X[m][k].  The generic kernel above splits K into 64-wide slices (grid.z) and re-reads dY for each: 3 x 52 MB + 118 MB
// at the training geometry.  Here a persistent 512-thread block walks 64-row chunks (double-buffered in LDS, one barrier per chunk) and reads each
// row of dY and X once; wave (nq, kq) owns the 32 x 32 block (n half, k tile kq of 0..3) and waves 0..3 also one 16 x 16 block of the ragged tail
// k = 128..143 -- all accumulators alive across the block's chunks, one [Npad 64][K 144] partial slab per block at the end.
constexpr int PW_ROWS = 64, PW_GS = 68, PW_XS = 148;   // LDS row strides (floats): 68 / 148 = 4 mod 32 banks -> the two pixel halves of an MFMA read disjoint banks

struct WgradPwArgs {
  const float* G; int g_stride; int g_choff;
  const float* X; int x_stride; int x_choff;
  float* P;        // [gridDim.x][64][144]
  int M;
};

__global__ __launch_bounds__(512) void k_wgrad_pw144(WgradPwArgs p) {
  extern __shared__ __attribute__((aligned(16))) float smp[];
  constexpr int BUF = PW_ROWS * (PW_GS + PW_XS);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int nq = wave >> 2, kq = wave & 3;
  const int half = lane >> 5, l31 = lane & 31, l15 = lane & 15, g4 = lane >> 4;
  const int nchunks = (p.M + PW_ROWS - 1) / PW_ROWS;
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  f32x4 acct = {0.f, 0.f, 0.f, 0.f};     // waves 0..3: block (n = 16 wave .. +15, k = 128 .. 143)
  // staging slots: dY 64 rows x 16 float4, X 64 rows x 36 float4 = 3328 float4 over 512 threads: 2 + 5 (the last partly)
  float4 rg[2], rx[5];
  auto prefetch = [&](int chunk) {
    const long long m0 = (long long)chunk * PW_ROWS;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int idx = tid + 512 * i, r = idx >> 4, c = idx & 15;
      const long long m = m0 + r;
      rg[i] = m < p.M ? *reinterpret_cast<const float4*>(p.G + m * p.g_stride + p.g_choff + c * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int i = 0; i < 5; ++i) {
      const int idx = tid + 512 * i, r = idx / 36, c = idx - r * 36;
      const long long m = m0 + r;
      rx[i] = (idx < PW_ROWS * 36 && m < p.M) ? *reinterpret_cast<const float4*>(p.X + m * p.x_stride + p.x_choff + c * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  };
  int chunk = blockIdx.x, it = 0;
  if (chunk < nchunks) prefetch(chunk);
  for (; chunk < nchunks; chunk += gridDim.x, ++it) {
    float* const sG = smp + (it & 1) * BUF;
    float* const sX = sG + PW_ROWS * PW_GS;
#pragma unroll
    for (int i = 0; i < 2; ++i) { const int idx = tid + 512 * i; *reinterpret_cast<float4*>(sG + (idx >> 4) * PW_GS + (idx & 15) * 4) = rg[i]; }
#pragma unroll
    for (int i = 0; i < 5; ++i) { const int idx = tid + 512 * i, r = idx / 36, c = idx - r * 36; if (idx < PW_ROWS * 36) *reinterpret_cast<float4*>(sX + r * PW_XS + c * 4) = rx[i]; }
    __syncthreads();   // (the only barrier of a chunk: the other buffer was last read before the previous chunk's barrier)
    if (chunk + (int)gridDim.x < nchunks) prefetch(chunk + gridDim.x);
    const float* gp = sG + half * PW_GS + nq * 32 + l31;
    const float* xp = sX + half * PW_XS + kq * 32 + l31;
#pragma unroll 8
    for (int mm = 0; mm < PW_ROWS; mm += 2)
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(gp[mm * PW_GS], xp[mm * PW_XS], acc, 0, 0, 0);
    if (wave < 4) {
      const float* gt = sG + g4 * PW_GS + wave * 16 + l15;
      const float* xt = sX + g4 * PW_XS + 128 + l15;
#pragma unroll 4
      for (int mm = 0; mm < PW_ROWS; mm += 4)
        acct = __builtin_amdgcn_mfma_f32_16x16x4f32(gt[mm * PW_GS], xt[mm * PW_XS], acct, 0, 0, 0);
    }
  }
  float* out = p.P + (long long)blockIdx.x * 64 * 144;
#pragma unroll
  for (int r = 0; r < 16; ++r) out[(nq * 32 + (r & 3) + 8 * (r >> 2) + 4 * half) * 144 + kq * 32 + l31] = acc[r];
  if (wave < 4) {
#pragma unroll
    for (int r = 0; r < 4; ++r) out[(wave * 16 + 4 * g4 + r) * 144 + 128 + l15] = acct[r];
  }
}

// sum partial slabs in split order and scatter to the PyTorch layout (O, C, T): inverse of k_pack_weight.
// P2 (optional) is a second partial set with the same geometry (the vertical EPI pass shares its weights).
__global__ __launch_bounds__(512) void k_wgrad_reduce(const float* __restrict__ P, int nsplit, const float* __restrict__ P2, int nsplit2,
                                                      float* __restrict__ dW, int O, int C, int T, int Npad, int perm, int ch, int accumulate, int c_valid, int chunk_mode) {
  // block = 64 elements x 8 split groups: group y sums slabs y, y+8, ... (fixed order), then the 8 group sums are added in
  // fixed order through LDS -> bitwise reproducible, and 8x the memory-level parallelism of one thread per element
  __shared__ float red[8][64];
  const long long total = (long long)T * Npad * C;
  const long long i = (long long)blockIdx.x * 64 + threadIdx.x;
  const int gy = threadIdx.y;
  float s = 0.f;
  if (i < total) {
    for (int sp = gy; sp < nsplit; sp += 8) s += P[(long long)sp * total + i];
    for (int sp = gy; sp < nsplit2; sp += 8) s += P2[(long long)sp * total + i];
  }
  red[gy][threadIdx.x] = s;
  __syncthreads();
  if (gy != 0 || i >= total) return;
  s = red[0][threadIdx.x];
#pragma unroll
  for (int g = 1; g < 8; ++g) s += red[g][threadIdx.x];
  int c = (int)(i % C);
  long long t2 = i / C;
  int n = (int)(t2 % Npad);
  int t = (int)(t2 / Npad);
  if (c >= c_valid) return;
  if (chunk_mode ? n >= ch : n >= O) return;
  int nref = n;
  if (perm == 1) { int r2 = O / ch; int q = n / ch, cc = n - q * ch; nref = cc * r2 + q; }
  long long o = ((long long)nref * c_valid + c) * T + t;
  if (chunk_mode) o = (long long)(perm ? n * T + t : t * ch + n) * C + c;
  dW[o] = accumulate ? dW[o] + s : s;
}

// transposed pack for dgrad: out[t'][k(Cpad rows)][n] = w[n][k][t]  with t' = flip ? T-1-t : t
//   (O,C,T) -> [T][Cpad32][O]: the dgrad GEMM contracts over the forward's output channels.
__global__ __launch_bounds__(256) void k_pack_weight_T(const float* __restrict__ w, float* __restrict__ out, int O, int C, int T, int Cpad, int flip) {
  const long long total = (long long)T * Cpad * O;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    int n = (int)(i % O);
    long long t2 = i / O;
    int k = (int)(t2 % Cpad);
    int tp = (int)(t2 / Cpad);
    int t = flip ? T - 1 - tp : tp;
    out[i] = k < C ? w[((long long)n * C + k) * T + t] : 0.f;
  }
}

template <int GIN, int XIN>
int launch_wgrad(WgradArgs p, hipStream_t st) {
  dim3 grid((unsigned)p.nsplit, (unsigned)p.ntaps, (unsigned)((p.K + 63) / 64));
  hipLaunchKernelGGL((k_wgrad<GIN, XIN>), grid, dim3(256), 0, st, p);
  LFSR_CHECK_LAUNCH();
  return LFSR_OK;
}

}  // namespace

// ---- internal API (lfsr_internal.h) -------------------------------------------------------------------------
#include "lfsr_internal.h"

int lfsr_wgrad_splits(int M, int ntaps, int K) {
  // aim at ~1500 blocks, >= 256 rows per block
  long long ktiles = (K + 63) / 64;
  long long want = 1536 / ((long long)ntaps * ktiles);
  if (want < 1) want = 1;
  long long rows = (M + want - 1) / want;
  if (rows < 256) rows = 256;
  rows = (rows + WG_ROWS - 1) / WG_ROWS * WG_ROWS;
  return (int)((M + rows - 1) / rows);
}

size_t lfsr_wgrad_partial_floats(int M, int ntaps, int N, int K) {
  return (size_t)lfsr_wgrad_splits(M, ntaps, K) * ntaps * npad32(N) * K;
}

int lfsr_wgrad_launch(int gmode, int xmode, const float* G, int g_stride, int g_choff, const float* X, int x_stride, int x_choff,
                      float* P, int M, int N, int K, int A, int h, int w, int ntaps, hipStream_t st) {
  LfsrOpTimer op_t("wgrad_generic", gmode * 16 + xmode, N * 1000 + K, st);
  if (!G || !X || !P || M <= 0 || N <= 0 || N > 64 || (N & 3) || K <= 0 || (K & 3) || ntaps <= 0) return LFSR_E_ARG;
  if ((g_stride | g_choff | x_stride | x_choff) & 3) return LFSR_E_ARG;
  WgradArgs p{};
  p.G = G; p.g_stride = g_stride; p.g_choff = g_choff; p.X = X; p.x_stride = x_stride; p.x_choff = x_choff; p.P = P;
  p.M = M; p.N = N; p.Npad = npad32(N); p.K = K; p.A = A; p.AA = A * A; p.H = h; p.W = w; p.ntaps = ntaps;
  p.nsplit = lfsr_wgrad_splits(M, ntaps, K);
  long long rows = ((long long)M + p.nsplit - 1) / p.nsplit;
  p.rows_per_split = (int)((rows + WG_ROWS - 1) / WG_ROWS * WG_ROWS);
#define WG(GM, XM) if (gmode == GM && xmode == XM) return launch_wgrad<GM, XM>(p, st);
  WG(IN_SAME, IN_CONV3)   // conv3x3
  WG(IN_SAME, IN_SAME)    // 1x1
  WG(IN_SAME, IN_ANG)     // AngConv.0
  WG(IN_ANG, IN_SAME)     // AngConv.2   (G gathered from the A*A views)
  WG(IN_SAME, IN_EPIH)    // EPIConv.0 horizontal
  WG(IN_SAME, IN_EPIV)    // EPIConv.0 vertical
  WG(IN_CHK_H, IN_SAME)   // EPIConv.2 horizontal
  WG(IN_CHK_V, IN_SAME)   // EPIConv.2 vertical
#undef WG
  return LFSR_E_ARG;
}


// LFSR_WGRAD3=direct keeps the direct-form kernel (A/B runs); the Winograd form is the default
static bool wgrad3_wino() { const char* s = lfsr_sel("LFSR_WGRAD3"); return !(s && s[0] == 'd'); }

int lfsr_wgrad_conv3_blocks(int n_img, int h, int w) {
  const int tr = wgrad3_wino() ? 2 : WT_R;
  long long tiles = (long long)n_img * ((h + tr - 1) / tr) * ((w + WT_C - 1) / WT_C);
  return (int)(tiles < 256 ? tiles : 256);
}

int lfsr_wgrad_conv3_launch(const float* G, int g_stride, int g_choff, const float* X, int x_stride, int x_choff, float* P,
                            int n_img, int h, int w, hipStream_t st) {
  LfsrOpTimer op_t("conv3x3_wgrad", n_img, h * w, st);
  if (!G || !X || !P || n_img <= 0 || h <= 0 || w <= 0 || ((g_stride | g_choff | x_stride | x_choff) & 3)) return LFSR_E_ARG;
  if ((long long)n_img * h * w * (x_stride > g_stride ? x_stride : g_stride) * 4 >= (1LL << 31)) return LFSR_E_ARG;   // 32-bit byte offsets
  static std::atomic<bool> attr_set[64];
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return LFSR_E_ARG;
  const int smem = (WX_PIX + WG_PIX) * LDS_ROW * 4;
  constexpr int smem_w = 2 * W2BUF * 4;
  static_assert(smem_w >= 8 * 8 * 64 * 4 * 4, "the epilogue's exchange fits the staging buffers");
  if (!attr_set[dev]) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_wgrad_conv3_halo), hipFuncAttributeMaxDynamicSharedMemorySize, smem);
    if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_wgrad_conv3_wino), hipFuncAttributeMaxDynamicSharedMemorySize, smem_w);
    if (e != hipSuccess) return LFSR_HIP_ERR(e);
    attr_set[dev] = true;
  }
  if (wgrad3_wino()) {
    Wgrad3Args q{};
    q.G = G; q.g_stride = g_stride; q.g_choff = g_choff; q.X = X; q.x_stride = x_stride; q.x_choff = x_choff; q.P = P;
    q.g_bytes = (int)((long long)n_img * h * w * g_stride * 4); q.x_bytes = (int)((long long)n_img * h * w * x_stride * 4);
    q.n_img = n_img; q.H = h; q.W = w; q.tiles_y = (h + 1) / 2; q.tiles_x = (w + WT_C - 1) / WT_C;
    hipLaunchKernelGGL(k_wgrad_conv3_wino, dim3((unsigned)lfsr_wgrad_conv3_blocks(n_img, h, w)), dim3(512), smem_w, st, q);
    LFSR_CHECK_LAUNCH();
    return LFSR_OK;
  }
  Wgrad3Args p{};
  p.G = G; p.g_stride = g_stride; p.g_choff = g_choff; p.X = X; p.x_stride = x_stride; p.x_choff = x_choff; p.P = P;
  p.g_bytes = (int)((long long)n_img * h * w * g_stride * 4); p.x_bytes = (int)((long long)n_img * h * w * x_stride * 4);
  p.n_img = n_img; p.H = h; p.W = w; p.tiles_y = (h + WT_R - 1) / WT_R; p.tiles_x = (w + WT_C - 1) / WT_C;
  hipLaunchKernelGGL(k_wgrad_conv3_halo, dim3((unsigned)lfsr_wgrad_conv3_blocks(n_img, h, w)), dim3(512), smem, st, p);
  LFSR_CHECK_LAUNCH();
  return LFSR_OK;
}

int lfsr_wgrad_reduce(const float* P, int nsplit, const float* P2, int nsplit2, float* dW, int O, int C, int T, int perm, int ch,
                      int accumulate, int c_valid, int chunk_mode, hipStream_t st) {
  LfsrOpTimer op_t("wgrad_reduce", O, C * T, st);
  if (c_valid <= 0 || c_valid > C) c_valid = C;
  if (!P || !dW || O <= 0 || C <= 0 || T <= 0) return LFSR_E_ARG;
  const int Npad = chunk_mode ? npad32(ch) : npad32(O);
  long long total = (long long)T * Npad * C;
  hipLaunchKernelGGL(k_wgrad_reduce, dim3(lfsr_blocks(total, 64)), dim3(64, 8), 0, st, P, nsplit, P2, P2 ? nsplit2 : 0, dW, O, C, T, Npad, perm, ch, accumulate, c_valid, chunk_mode);
  LFSR_CHECK_LAUNCH();
  return LFSR_OK;
}

int lfsr_pack_weight_T(const float* w, float* out, int O, int C, int T, int flip, hipStream_t st) {
  return lfsr_pack_weight_T_m(w, out, O, C, T, flip, LFSR_W_ALL, st);
}

int lfsr_pack_weight_T_m(const float* w, float* out, int O, int C, int T, int flip, int mask, hipStream_t st) {
  if (!w || !out || O <= 0 || C <= 0 || T <= 0) return LFSR_E_ARG;
  if (O == 64 && C == 64 && T == 9 && flip == 1 && mask == LFSR_W_WINO4)   // the runtimes' lean repack: one launch per weight
    return lfsr_pack_conv3_raw_wino4(w, out, out + LFSR_CONV3_DIRECT_FLOATS + LFSR_CONV3_WINO2_FLOATS, 1, st);
  long long total = (long long)T * npad32(C) * O;
  unsigned grid = lfsr_blocks(total, 256);
  if (grid > 4096) grid = 4096;
  hipLaunchKernelGGL(k_pack_weight_T, dim3(grid), dim3(256), 0, st, w, out, O, C, T, npad32(C), flip);
  LFSR_CHECK_LAUNCH();
  if (O == 64 && C == 64 && T == 9 && flip == 1) return lfsr_pack_wino_m(out, out + LFSR_CONV3_DIRECT_FLOATS, mask, st);   // dgrad runs the Winograd kernel too
  return LFSR_OK;
}


// EPIConv.0 weight gradient in EPI-line form (A = 5, lines of <= 32 pixels); LFSR_E_ARG = not covered (the caller keeps the gather form).
// LFSR_WGRAD_EPI=gather forces the gather form (A/B runs).
// vert: 0 horizontal pass (dE), 1 vertical pass (dE), 2 both in one launch (dE = horizontal, dE_v = vertical): one slab set for the shared weights
int lfsr_wgrad_epi0_blocks(int B, int A, int h, int w, int vert) {
  const long long nl = (long long)B * A * (vert == 2 ? h + w : vert ? w : h);
  return (int)(nl < 256 ? nl : 256);
}

int lfsr_wgrad_epi0_launch(const float* dE, const float* dE_v, const float* X, int x_stride, int x_choff, float* P, int B, int A, int h, int w, int vert, hipStream_t st) {
  LfsrOpTimer op_t("epi0_wgrad", B, h * w, st);
  if (!dE || !X || !P || B <= 0 || h <= 0 || w <= 0 || ((x_stride | x_choff) & 3) || vert < 0 || vert > 2 || (vert == 2 && !dE_v)) return LFSR_E_ARG;
  const char* sel = lfsr_sel("LFSR_WGRAD_EPI");
  if (A != 5 || (vert != 0 && h > 32) || (vert != 1 && w > 32) || (sel && sel[0] == 'g')) return LFSR_E_ARG;
  if ((long long)B * A * A * h * w * x_stride * 4 >= (1LL << 31)) return LFSR_E_ARG;
  static std::atomic<bool> attr_set[64];
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return LFSR_E_ARG;
  constexpr int smem = 2 * (5 * WE_PX * WE_SI + 32 * WE_SE) * 4;
  if (!attr_set[dev]) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_wgrad_epi0_lines<5>), hipFuncAttributeMaxDynamicSharedMemorySize, smem);
    if (e != hipSuccess) return LFSR_HIP_ERR(e);
    attr_set[dev] = true;
  }
  WgradEpiArgs p{};
  p.G = dE; p.G2 = dE_v; p.X = X; p.x_stride = x_stride; p.x_choff = x_choff; p.P = P;
  p.g_bytes = (int)((long long)B * A * h * w * 32 * 4); p.x_bytes = (int)((long long)B * A * A * h * w * x_stride * 4);
  p.B = B; p.A = A; p.H = h; p.W = w; p.vert = vert;
  hipLaunchKernelGGL(k_wgrad_epi0_lines<5>, dim3((unsigned)lfsr_wgrad_epi0_blocks(B, A, h, w, vert)), dim3(512), smem, st, p);
  LFSR_CHECK_LAUNCH();
  return LFSR_OK;
}


// fuse.0-shaped 1x1 weight gradient (N = 64, K = 144): one slab per block; LFSR_E_ARG = not covered.  LFSR_WGRAD_PW=gather keeps the generic kernel.
int lfsr_wgrad_pw144_blocks(int M) {
  const int nchunks = (M + PW_ROWS - 1) / PW_ROWS;
  return nchunks < 256 ? nchunks : 256;
}

int lfsr_wgrad_pw144_launch(const float* G, int g_stride, int g_choff, const float* X, int x_stride, int x_choff, float* P, int M, hipStream_t st) {
  LfsrOpTimer op_t("pw144_wgrad", M, 0, st);
  if (!G || !X || !P || M <= 0 || ((g_stride | g_choff | x_stride | x_choff) & 3) || g_stride < g_choff + 64 || x_stride < x_choff + 144) return LFSR_E_ARG;
  const char* sel = lfsr_sel("LFSR_WGRAD_PW");
  if (sel && sel[0] == 'g') return LFSR_E_ARG;
  static std::atomic<bool> attr_set[64];
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return LFSR_E_ARG;
  constexpr int smem = 2 * PW_ROWS * (PW_GS + PW_XS) * 4;
  if (!attr_set[dev]) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_wgrad_pw144), hipFuncAttributeMaxDynamicSharedMemorySize, smem);
    if (e != hipSuccess) return LFSR_HIP_ERR(e);
    attr_set[dev] = true;
  }
  WgradPwArgs p{};
  p.G = G; p.g_stride = g_stride; p.g_choff = g_choff; p.X = X; p.x_stride = x_stride; p.x_choff = x_choff; p.P = P; p.M = M;
  hipLaunchKernelGGL(k_wgrad_pw144, dim3((unsigned)lfsr_wgrad_pw144_blocks(M)), dim3(512), smem, st, p);
  LFSR_CHECK_LAUNCH();
  return LFSR_OK;
}
