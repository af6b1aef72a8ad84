// Fused EPIConv branch (model/SR/DistgSSR.py:91-97 and its transposed application :108) on VCL, fp32 MFMA:
//   t = lrelu(conv 1xA^2, stride (1,A), pad A(A-1)/2, 64->32);  y = lrelu(1x1 32->32A);  PixelShuffle1D(A)
// for BOTH the horizontal pass (EPI lines along x, weights applied to the MacPI tensor) and the vertical pass
// (lines along y, the reference transposes the tensor and reuses the same weights) in ONE launch, writing each
// pass straight into its 32-channel slice of the 144-channel concat buffer.
//
// In VCL the 1xA^2 conv is a 1-D conv along the EPI line with A taps (dxi) over an (A views x 64)-channel
// input: source of tap k = A*dxi + v' for output position t is view v', position t + dxi - (A-1)/2 (odd A).
// One 512-thread block = 8 EPI lines (one per wave: 32 positions = the 32 A-rows of a 32x32x2 MFMA, N = 32).
// K is walked view by view ("stage" v'): the 8 x (32+A-1) input vectors of that view and the A weight slabs
// (k = v', A+v', ...) are staged in LDS once and reused by the A taps at shifted addresses (5x reuse), with the
// next stage prefetched into registers while the current one computes.  After the last stage the 32x32 result
// tile is LeakyReLU'd, transposed through LDS into MFMA A-operand order and multiplied by the 1x1 weights
// (20 KB, read straight from L1/L2), then scattered by chunk (= PixelShuffle1D) to the A views.
#include <stdlib.h>

#include "lfsr_internal.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

namespace {

constexpr int LINES = 8, LROW = 68, TROW = 36;

struct EpiArgs {
  const float* X; int x_stride; int x_choff; int x_bytes;   // x_bytes: the operand's true byte span (descriptor extent)
  const float* W1;    // [A*A][32][64]  (tap k, n, c)
  const float* W1u;   // (A = 5, Winograd form) [5 v'][6 p][32 n][64 c]: U = G g over the five spatial taps of view v' (lfsr_pack_epi_wino)
  const float* W2;    // [32A][32]      (n = chunk*32 + c, k): the 1x1 weights, row-major as packed by lfsr_pack_conv_weight
  float* Y; int y_stride; int choffH; int choffV;
  float* TH; float* TV;   // optional (lines*len, 32): post-LeakyReLU stage-1 activations saved for backward
  int B, A, H, W;
  int tilesH, tilesV;   // tiles per pass; blockIdx.x < tilesH -> horizontal
  int xcd_swz;          // 1: XCD-aware block order (needs gridDim.x % 8 == 0)
  int tpiH, tpiV;       // > 0: blocks are ordered ITEM by item (tpiH horizontal tiles of batch item b, then its tpiV vertical tiles), so an item's
                        // second pass reads its 25 view images while they are still in the last-level caches; 0: all horizontal tiles, then all vertical ones
  float slope;
};

// AT = 5: the BASELINE angular resolution, loops over views / taps fully unrolled; AT = 0: A read at run time (A = 1, 3)
template <int AT>
__global__ __launch_bounds__(512) void k_epi_fused(EpiArgs p) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int A = AT ? AT : p.A, PP = 32 + A - 1, pad = (A - 1) / 2;
  float* sA = smem;                              // [LINES][PP][LROW]
  float* sW = smem + LINES * PP * LROW;          // [A][32][LROW]
  int* sLine = reinterpret_cast<int*>(sW + A * 32 * LROW);   // [LINES] source/dest base pixel or -1
  float* sW2 = reinterpret_cast<float*>(sLine + LINES + 4);   // [32A][TROW]: the 1x1 weights, staged once per block (16-B aligned)

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int c16 = tid & 15, r16 = tid >> 4;
  const int half = lane >> 5, l31 = lane & 31;
  const int HW = p.H * p.W;

  bool vert; int tile;
  if (p.tpiH > 0) {
    // consecutive workgroups go to consecutive XCDs (8 per chip): with xcd_swz the blocks of one XCD walk a CONTIGUOUS range of items, so both passes of an
    // item share one L2 instead of every XCD reading every item twice
    const int bid = p.xcd_swz ? (int)(blockIdx.x & 7) * (int)(gridDim.x >> 3) + (int)(blockIdx.x >> 3) : (int)blockIdx.x;
    const int per = p.tpiH + p.tpiV, item = bid / per, r = bid - item * per;
    vert = r >= p.tpiH;
    tile = vert ? item * p.tpiV + (r - p.tpiH) : item * p.tpiH + r;
  } else {
    vert = (int)blockIdx.x >= p.tilesH;
    tile = vert ? blockIdx.x - p.tilesH : blockIdx.x;
  }
  const int len = vert ? p.H : p.W;                 // positions along the line
  const int across = vert ? p.W : p.H;              // lines per (b, u|v) group
  const int nlines = p.B * A * across;
  const int vstride = vert ? A * HW : HW;           // pixel stride between the A source views / dest chunks
  const int pstride = vert ? p.W : 1;               // pixel stride along the line
  const int choff = vert ? p.choffV : p.choffH;

  if (tid < LINES) {
    int ln = tile * LINES + tid;
    int base = -1;
    if (ln < nlines) {
      int q = ln / across, o = ln - q * across;     // horizontal: q = b*A+u, o = y;  vertical: q = b*A+v, o = x
      if (!vert) base = q * A * HW + o * p.W;
      else { int b = q / A, v = q - b * A; base = (b * A * A + v) * HW + o; }
    }
    sLine[tid] = base;
  }
  {   // 1x1 weights -> LDS (rows padded to TROW floats): all loads of a thread issued before its first store
    float4 wv[3];
#pragma unroll
    for (int q = 0; q < 3; ++q) {
      const int idx = tid + 512 * q;
      wv[q] = idx < A * 32 * 8 ? reinterpret_cast<const float4*>(p.W2)[idx] : make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int q = 0; q < 3; ++q) {
      const int idx = tid + 512 * q;
      if (idx < A * 32 * 8) *reinterpret_cast<float4*>(sW2 + (idx >> 3) * TROW + (idx & 7) * 4) = wv[q];
    }
  }
  __syncthreads();

  const int nvec = LINES * PP;                      // input vectors per stage
  float4 ra[9], rw[5];                              // A <= 5 fast path sizes (checked by the launcher)
  // Stage loads through buffer descriptors: the per-thread byte offsets of its 9 input vectors and 5 weight rows are worked out
  // ONCE per block (line bases from LDS, bounds -> out-of-range offsets that read as zero); a stage only adds the view's
  // wave-uniform offset.  As per-stage branches with 64-bit addresses this code sat un-overlapped in front of every stage's MFMAs.
  constexpr int EOOB = (int)0x80000000u;
  const __amdgpu_buffer_rsrc_t rsX = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.X), 0, p.x_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsW = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.W1), 0, A * A * 32 * 64 * 4, 0x00020000);
  int offA[9], offW[5];
#pragma unroll
  for (int i = 0; i < 9; ++i) {
    const int vec = r16 + 32 * i, l = vec < nvec ? vec / PP : 0, t = vec - l * PP - pad;
    const int base = sLine[l];
    offA[i] = (vec < nvec && base >= 0 && t >= 0 && t < len) ? ((base + t * pstride) * p.x_stride + p.x_choff + c16 * 4) * 4 : EOOB;
  }
#pragma unroll
  for (int i = 0; i < 5; ++i) {
    const int row = r16 + 32 * i, dxi = row >> 5, n = row & 31;   // (dxi, n)
    offW[i] = row < A * 32 ? ((A * dxi * 32 + n) * 64 + c16 * 4) * 4 : EOOB;
  }
  typedef float f32x4e __attribute__((ext_vector_type(4)));
  auto prefetch = [&](int vv) {
    const int sA4 = vv * vstride * p.x_stride * 4, sW4 = vv * 32 * 64 * 4;   // wave-uniform
#pragma unroll
    for (int i = 0; i < 9; ++i) {
      const f32x4e v = __builtin_bit_cast(f32x4e, __builtin_amdgcn_raw_buffer_load_b128(rsX, offA[i], sA4, 0));
      ra[i] = make_float4(v.x, v.y, v.z, v.w);
    }
#pragma unroll
    for (int i = 0; i < 5; ++i) {
      const f32x4e v = __builtin_bit_cast(f32x4e, __builtin_amdgcn_raw_buffer_load_b128(rsW, offW[i], sW4, 0));
      rw[i] = make_float4(v.x, v.y, v.z, v.w);
    }
  };

  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;

  const float* aBase = sA + (wave * PP + l31) * LROW + 4 * half;
  const float* bBase = sW + l31 * LROW + 4 * half;

  prefetch(0);
#pragma unroll
  for (int vv = 0; vv < (AT ? AT : 5); ++vv) {
    if (vv >= A) break;
    if (vv > 0) __syncthreads();                    // previous stage fully consumed
#pragma unroll
    for (int i = 0; i < 9; ++i) {
      int vec = r16 + 32 * i;
      if (vec < nvec) *reinterpret_cast<float4*>(sA + vec * LROW + c16 * 4) = ra[i];
    }
#pragma unroll
    for (int i = 0; i < 5; ++i) {
      int row = r16 + 32 * i;
      if (row < A * 32) *reinterpret_cast<float4*>(sW + row * LROW + c16 * 4) = rw[i];
    }
    __syncthreads();
    if (vv + 1 < A) prefetch(vv + 1);               // flies under this stage's MFMAs
#pragma unroll
    for (int dxi = 0; dxi < (AT ? AT : 5); ++dxi) {
      if (dxi >= A) break;
      const float* aT = aBase + dxi * LROW;
      const float* bT = bBase + dxi * 32 * LROW;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        float4 a = *reinterpret_cast<const float4*>(aT + 8 * j);
        float4 b = *reinterpret_cast<const float4*>(bT + 8 * j);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b.x, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b.y, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b.z, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b.w, acc, 0, 0, 0);
      }
    }
  }
  __syncthreads();                                   // stage area is dead: reuse it for the transposition

  // ---- stage 2: t = lrelu(acc) [pos][32] -> LDS (row stride 36) -> A-operand fragments --------------------
  float* sT = sA + wave * 32 * TROW;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    int pos = (r & 3) + 8 * (r >> 2) + 4 * half;
    float v = acc[r];
    v = v >= 0.f ? v : v * p.slope;
    sT[pos * TROW + l31] = v;
    float* tsave = vert ? p.TV : p.TH;
    // rows of the saved matrix are ordered like the gather-GEMM's stage-1 rows: (b*A+u, y, x) / (b*A+v, y, x)
    if (tsave && sLine[wave] >= 0 && pos < len) {
      int ln = tile * LINES + wave;
      int q = ln / across, o = ln - q * across;
      long long row = vert ? ((long long)q * p.H + pos) * p.W + o : ((long long)q * p.H + o) * p.W + pos;
      tsave[row * 32 + l31] = v;
    }
  }
  __builtin_amdgcn_wave_barrier();
  float4 fa[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) fa[j] = *reinterpret_cast<const float4*>(sT + l31 * TROW + 8 * j + 4 * half);

  const int base = sLine[wave];
  for (int nt = 0; nt < A; ++nt) {                  // chunk nt = destination view along the EPI's angular axis
    f32x16 o;
#pragma unroll
    for (int r = 0; r < 16; ++r) o[r] = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float4 b = *reinterpret_cast<const float4*>(sW2 + (nt * 32 + l31) * TROW + 8 * j + 4 * half);
      o = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[j].x, b.x, o, 0, 0, 0);
      o = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[j].y, b.y, o, 0, 0, 0);
      o = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[j].z, b.z, o, 0, 0, 0);
      o = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[j].w, b.w, o, 0, 0, 0);
    }
    // transpose the 32x32 tile through the wave-private LDS tile so that a lane stores 16 B (4 instructions of 8 positions x 128 B
    // per chunk instead of 16 dword stores)
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int pos = (r & 3) + 8 * (r >> 2) + 4 * half;
      float v = o[r];
      v = v >= 0.f ? v : v * p.slope;
      sT[pos * TROW + l31] = v;
    }
    __builtin_amdgcn_wave_barrier();
    if (base >= 0) {
      const long long dview = (long long)base + (long long)nt * vstride;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int pos = (lane >> 3) + 8 * q, c4 = (lane & 7) * 4;
        if (pos < len)
          *reinterpret_cast<float4*>(p.Y + (dview + (long long)pos * pstride) * p.y_stride + choff + c4) = *reinterpret_cast<const float4*>(sT + pos * TROW + c4);
      }
    }
  }
}


// ---- A = 5, stage 1 in Winograd F(2,5) form ------------------------------------------------------------------------------------------------
// Stage 1 is a 5-tap 1-D conv along the EPI line over 5 x 64 = 320 input channels: y[t] = sum_{v', dx, c} W[v', dx][n][c] x[v'][t + dx - 2][c].
// F(2,5) on the points 0, +-1, +-2, inf (the interpolation points -- and therefore Bt -- of the F(4,3) transform of conv3x3_wino4.hip):
//   Y = At [ sum_{v', c} (G g)[p] . (Bt d)[p] ],   d = the 6 inputs of an output pair, 6 products per 2 outputs instead of 10.
// One wave = one EPI line = 16 output pairs = the 16 A-rows of v_mfma_f32_16x16x4_f32; N = 32 = two N tiles.  The input transform runs in
// the MFMA wave itself, on the operand registers: lane (pair j, g) reads its six raw vectors x[2 j + i][16 s + 4 g ..+3] (ds_read_b128),
// forms Bt d for the four channels at once (packed math: 24 instructions) and has the A operands of 6 p x 4 steps x 2 N tiles = 48 MFMAs;
// U comes from LDS ([p][n][c] rows of one view per stage).  48 accumulator registers; At in registers; then stage 2 exactly as k_epi_fused.
// fp32 round-off: the same transform family as the 3x3 convs' F(4x4,3x3), in one dimension only (tools / tests: test_epiconv*).
typedef float f32x4w __attribute__((ext_vector_type(4)));

typedef float f32x2w __attribute__((ext_vector_type(2)));
// the fma sequence of conv3x3_wino4.hip's bt6 on channel pairs (v_pk_fma_f32 / v_pk_add_f32)
__device__ __forceinline__ void bt6v(f32x2w& d0, f32x2w& d1, f32x2w& d2, f32x2w& d3, f32x2w& d4, f32x2w& d5) {
  const f32x2w m4 = {-4.f, -4.f}, p4 = {4.f, 4.f}, m5 = {-5.f, -5.f}, p2 = {2.f, 2.f}, m2 = {-2.f, -2.f};
  const f32x2w a = __builtin_elementwise_fma(m4, d2, d4), b = __builtin_elementwise_fma(m4, d1, d3);
  const f32x2w c = d4 - d2, e = d3 - d1;
  const f32x2w t0 = __builtin_elementwise_fma(p4, d0, __builtin_elementwise_fma(m5, d2, d4));
  const f32x2w t5 = __builtin_elementwise_fma(p4, d1, __builtin_elementwise_fma(m5, d3, d5));
  d0 = t0; d1 = a + b; d2 = a - b; d3 = __builtin_elementwise_fma(p2, e, c); d4 = __builtin_elementwise_fma(m2, e, c); d5 = t5;
}

__global__ __launch_bounds__(512) void k_epi_wino5(EpiArgs p) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int A = 5, PP = 36, pad = 2;
  float* sA = smem;                              // [LINES][PP][LROW]
  float* sU = smem + LINES * PP * LROW;          // [6][32][LROW]
  int* sLine = reinterpret_cast<int*>(sU + 6 * 32 * LROW);
  float* sW2 = reinterpret_cast<float*>(sLine + LINES + 4);   // [32A][TROW]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int c16 = tid & 15, r16 = tid >> 4;
  const int half = lane >> 5, l31 = lane & 31, l15 = lane & 15, g = lane >> 4;
  const int HW = p.H * p.W;

  bool vert; int tile;
  if (p.tpiH > 0) {
    // consecutive workgroups go to consecutive XCDs (8 per chip): with xcd_swz the blocks of one XCD walk a CONTIGUOUS range of items, so both passes of an
    // item share one L2 instead of every XCD reading every item twice
    const int bid = p.xcd_swz ? (int)(blockIdx.x & 7) * (int)(gridDim.x >> 3) + (int)(blockIdx.x >> 3) : (int)blockIdx.x;
    const int per = p.tpiH + p.tpiV, item = bid / per, r = bid - item * per;
    vert = r >= p.tpiH;
    tile = vert ? item * p.tpiV + (r - p.tpiH) : item * p.tpiH + r;
  } else {
    vert = (int)blockIdx.x >= p.tilesH;
    tile = vert ? blockIdx.x - p.tilesH : blockIdx.x;
  }
  const int len = vert ? p.H : p.W;
  const int across = vert ? p.W : p.H;
  const int nlines = p.B * A * across;
  const int vstride = vert ? A * HW : HW;
  const int pstride = vert ? p.W : 1;
  const int choff = vert ? p.choffV : p.choffH;

  auto line_base = [&](int l) -> int {               // first pixel of EPI line l of this tile, -1 beyond the tensor
    const int ln = tile * LINES + l;
    if (ln >= nlines) return -1;
    const int q = ln / across, o = ln - q * across;
    if (!vert) return q * A * HW + o * p.W;
    const int b = q / A, v = q - b * A;
    return (b * A * A + v) * HW + o;
  };
  if (tid < LINES) sLine[tid] = line_base(tid);

  // stage-0 operands are requested FIRST (line bases in registers, not through LDS), so their HBM latency runs under the staging of the
  // 1x1 weights below: a block's prologue is otherwise serial (1280 blocks = 5 rounds per launch)
  constexpr int nvec = LINES * PP;
  constexpr int EOOB = (int)0x80000000u;
  const __amdgpu_buffer_rsrc_t rsX = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.X), 0, p.x_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsU = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.W1u), 0, A * 6 * 32 * 64 * 4, 0x00020000);
  int offA[9];
#pragma unroll
  for (int i = 0; i < 9; ++i) {
    const int vec = r16 + 32 * i, l = vec < nvec ? vec / PP : 0, t = vec - l * PP - pad;
    const int base = line_base(l);
    offA[i] = (vec < nvec && base >= 0 && t >= 0 && t < len) ? ((base + t * pstride) * p.x_stride + p.x_choff + c16 * 4) * 4 : EOOB;
  }
  const int offU = (r16 * 64 + c16 * 4) * 4;   // row (p = i, n = r16) of a view's U: + i * 32 * 256 bytes
  f32x4w ra[9], ru[6];
  auto prefetch = [&](int vv) {
    const int sA4 = vv * vstride * p.x_stride * 4, sU4 = vv * 6 * 32 * 64 * 4;   // wave-uniform
#pragma unroll
    for (int i = 0; i < 9; ++i) ra[i] = __builtin_bit_cast(f32x4w, __builtin_amdgcn_raw_buffer_load_b128(rsX, offA[i], sA4, 0));
#pragma unroll
    for (int i = 0; i < 6; ++i) ru[i] = __builtin_bit_cast(f32x4w, __builtin_amdgcn_raw_buffer_load_b128(rsU, offU + i * 32 * 256, sU4, 0));
  };
  prefetch(0);
  {
    float4 wv[3];
#pragma unroll
    for (int q = 0; q < 3; ++q) {
      const int idx = tid + 512 * q;
      wv[q] = idx < A * 32 * 8 ? reinterpret_cast<const float4*>(p.W2)[idx] : make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int q = 0; q < 3; ++q) {
      const int idx = tid + 512 * q;
      if (idx < A * 32 * 8) *reinterpret_cast<float4*>(sW2 + (idx >> 3) * TROW + (idx & 7) * 4) = wv[q];
    }
  }

  f32x4w acc[6][2];
#pragma unroll
  for (int q = 0; q < 6; ++q) { acc[q][0] = f32x4w{0.f, 0.f, 0.f, 0.f}; acc[q][1] = acc[q][0]; }

  const float* aBase = sA + (wave * PP + 2 * l15) * LROW + 4 * g;   // raw vector i of this lane's pair: + i * LROW (+ 16 s)
  // U rows are 256 B (no padding) with the 16-B chunk index XORed by the row's low four bits: ds_read_b128 serves a wave in four 16-lane groups, each holding every
  // l15 once with two values of g ({0-3, 12-15, 20-27}, ...: MI355X_MICROARCH.md, LDS table); in the padded layout (row stride 272 B, chunk g at + 16 g B) rows r, g and
  // r - 1, g + 1 shared a slot (SQ_LDS_BANK_CONFLICT 31 % of the LDS-active cycles); chunk ^ l15 gives each group 16 distinct slots, and the 8-lane write groups too.
  // Measured on the headline step, three runs each in one call: 1746 -> 1748 patches/s, i.e. inside the noise -- these reads are not on the kernel's critical path
  const float* uBase = sU + l15 * 64;                                // U row (p, 16 nt + n): + (p * 32 + 16 nt) * 64, chunk ((4 s + g) ^ l15)
  const int ux_lo = (g ^ l15) & 3, ux_hi = l15 & 12;

#pragma unroll 1
  for (int vv = 0; vv < A; ++vv) {
    if (vv > 0) __syncthreads();                    // previous stage fully consumed
#pragma unroll
    for (int i = 0; i < 9; ++i) {
      const int vec = r16 + 32 * i;
      if (vec < nvec) *reinterpret_cast<f32x4w*>(sA + vec * LROW + c16 * 4) = ra[i];
    }
#pragma unroll
    for (int i = 0; i < 6; ++i) *reinterpret_cast<f32x4w*>(sU + (i * 32 + r16) * 64 + ((c16 ^ (r16 & 15)) << 2)) = ru[i];
    __syncthreads();
    if (vv + 1 < A) prefetch(vv + 1);               // flies under this stage's MFMAs
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      f32x2w dl[6], dh[6];
#pragma unroll
      for (int i = 0; i < 6; ++i) {
        const f32x4w t = *reinterpret_cast<const f32x4w*>(aBase + i * LROW + 16 * s);
        dl[i] = __builtin_shufflevector(t, t, 0, 1);
        dh[i] = __builtin_shufflevector(t, t, 2, 3);
      }
      bt6v(dl[0], dl[1], dl[2], dl[3], dl[4], dl[5]);
      bt6v(dh[0], dh[1], dh[2], dh[3], dh[4], dh[5]);
#pragma unroll
      for (int q = 0; q < 6; ++q) {
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
          const f32x4w ub = *reinterpret_cast<const f32x4w*>(uBase + (q * 32 + 16 * nt) * 64 + ((((4 * s) ^ ux_hi) | ux_lo) << 2));
          acc[q][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(dl[q].x, ub.x, acc[q][nt], 0, 0, 0);
          acc[q][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(dl[q].y, ub.y, acc[q][nt], 0, 0, 0);
          acc[q][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(dh[q].x, ub.z, acc[q][nt], 0, 0, 0);
          acc[q][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(dh[q].y, ub.w, acc[q][nt], 0, 0, 0);
        }
      }
    }
  }
  __syncthreads();                                   // stage area is dead: reuse it for the transposition

  // ---- At: y[2 j] = M0 + M1 + M2 + M3 + M4,  y[2 j + 1] = M1 - M2 + 2 (M3 - M4) + M5;  t = lrelu(y) [pos][32] -> LDS (row stride 36) ----
  float* sT = sA + wave * 32 * TROW;
  {
    float* tsave = vert ? p.TV : p.TH;
    const bool save = tsave && sLine[wave] >= 0;
    long long rowbase = 0; int rowstep = 0;
    if (save) {
      int ln = tile * LINES + wave;
      int q = ln / across, o = ln - q * across;
      // rows of the saved matrix are ordered like the gather-GEMM's stage-1 rows: (b*A+u, y, x) / (b*A+v, y, x)
      rowbase = vert ? (long long)q * p.H * p.W + o : ((long long)q * p.H + o) * p.W;
      rowstep = vert ? p.W : 1;
    }
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
      const f32x4w s12 = acc[1][nt] + acc[2][nt], d12 = acc[1][nt] - acc[2][nt], s34 = acc[3][nt] + acc[4][nt], d34 = acc[3][nt] - acc[4][nt];
      const f32x4w y0 = (acc[0][nt] + s12) + s34;
      const f32x4w y1 = (d12 + 2.f * d34) + acc[5][nt];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
#pragma unroll
        for (int a = 0; a < 2; ++a) {
          const int pos = 2 * (4 * g + r) + a;
          float v = a ? y1[r] : y0[r];
          v = v >= 0.f ? v : v * p.slope;
          sT[pos * TROW + 16 * nt + l15] = v;
          if (save && pos < len) tsave[(rowbase + (long long)pos * rowstep) * 32 + 16 * nt + l15] = v;
        }
      }
    }
  }
  __builtin_amdgcn_wave_barrier();
  float4 fa[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) fa[j] = *reinterpret_cast<const float4*>(sT + l31 * TROW + 8 * j + 4 * half);

  const int base = sLine[wave];
  for (int nt = 0; nt < A; ++nt) {                  // chunk nt = destination view along the EPI's angular axis
    f32x16 o;
#pragma unroll
    for (int r = 0; r < 16; ++r) o[r] = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float4 b = *reinterpret_cast<const float4*>(sW2 + (nt * 32 + l31) * TROW + 8 * j + 4 * half);
      o = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[j].x, b.x, o, 0, 0, 0);
      o = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[j].y, b.y, o, 0, 0, 0);
      o = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[j].z, b.z, o, 0, 0, 0);
      o = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[j].w, b.w, o, 0, 0, 0);
    }
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int pos = (r & 3) + 8 * (r >> 2) + 4 * half;
      float v = o[r];
      v = v >= 0.f ? v : v * p.slope;
      sT[pos * TROW + l31] = v;
    }
    __builtin_amdgcn_wave_barrier();
    if (base >= 0) {
      const long long dview = (long long)base + (long long)nt * vstride;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int pos = (lane >> 3) + 8 * q, c4 = (lane & 7) * 4;
        if (pos < len)
          *reinterpret_cast<float4*>(p.Y + (dview + (long long)pos * pstride) * p.y_stride + choff + c4) = *reinterpret_cast<const float4*>(sT + pos * TROW + c4);
      }
    }
  }
}

// U[v'][p][n][c] = sum_dx G[p][dx] W1[tap A dx + v'][n][c]  (fp64, rounded once), G of F(2,5) on the points 0, +-1, +-2, inf
__global__ __launch_bounds__(256) void k_pack_epi_wino(const float* __restrict__ w1, float* __restrict__ out) {
  const int i = blockIdx.x * 256 + threadIdx.x;   // (v', n, c)
  if (i >= 5 * 32 * 64) return;
  const int c = i & 63, n = (i >> 6) & 31, v = i >> 11;
  double gk[5];
#pragma unroll
  for (int dx = 0; dx < 5; ++dx) gk[dx] = (double)w1[((5 * dx + v) * 32 + n) * 64 + c];
  const double G[6][5] = {{1.0 / 4, 0, 0, 0, 0}, {-1.0 / 6, -1.0 / 6, -1.0 / 6, -1.0 / 6, -1.0 / 6}, {-1.0 / 6, 1.0 / 6, -1.0 / 6, 1.0 / 6, -1.0 / 6},
                          {1.0 / 24, 1.0 / 12, 1.0 / 6, 1.0 / 3, 2.0 / 3}, {1.0 / 24, -1.0 / 12, 1.0 / 6, -1.0 / 3, 2.0 / 3}, {0, 0, 0, 0, 1.0}};
#pragma unroll
  for (int q = 0; q < 6; ++q) {
    double u = 0.0;
#pragma unroll
    for (int dx = 0; dx < 5; ++dx) u += G[q][dx] * gk[dx];
    out[((v * 6 + q) * 32 + n) * 64 + c] = (float)u;
  }
}

__global__ __launch_bounds__(256) void k_pack_epi_wino_batch(const LfsrPackDesc* __restrict__ tab) {
  const LfsrPackDesc d = tab[blockIdx.y];
  const float* __restrict__ w1 = d.src;
  float* __restrict__ out = d.dst;
  const int i = blockIdx.x * 256 + threadIdx.x;   // (v', n, c)
  if (i >= 5 * 32 * 64) return;
  const int c = i & 63, n = (i >> 6) & 31, v = i >> 11;
  double gk[5];
#pragma unroll
  for (int dx = 0; dx < 5; ++dx) gk[dx] = (double)w1[((5 * dx + v) * 32 + n) * 64 + c];
  const double G[6][5] = {{1.0 / 4, 0, 0, 0, 0}, {-1.0 / 6, -1.0 / 6, -1.0 / 6, -1.0 / 6, -1.0 / 6}, {-1.0 / 6, 1.0 / 6, -1.0 / 6, 1.0 / 6, -1.0 / 6},
                          {1.0 / 24, 1.0 / 12, 1.0 / 6, 1.0 / 3, 2.0 / 3}, {1.0 / 24, -1.0 / 12, 1.0 / 6, -1.0 / 3, 2.0 / 3}, {0, 0, 0, 0, 1.0}};
#pragma unroll
  for (int q = 0; q < 6; ++q) {
    double u = 0.0;
#pragma unroll
    for (int dx = 0; dx < 5; ++dx) u += G[q][dx] * gk[dx];
    out[((v * 6 + q) * 32 + n) * 64 + c] = (float)u;
  }
}

}  // namespace

int lfsr_pack_epi_wino_batch(const LfsrPackDesc* table_dev, int n, hipStream_t st) {
  if (!table_dev || n <= 0) return n == 0 ? LFSR_OK : LFSR_E_ARG;
  hipLaunchKernelGGL(k_pack_epi_wino_batch, dim3(40, (unsigned)n), dim3(256), 0, st, table_dev);
  LFSR_CHECK_LAUNCH();
  return LFSR_OK;
}

int lfsr_pack_epi_wino(const float* w1_direct_packed, float* out, hipStream_t st) {
  if (!w1_direct_packed || !out) return LFSR_E_ARG;
  hipLaunchKernelGGL(k_pack_epi_wino, dim3(40), dim3(256), 0, st, w1_direct_packed, out);
  LFSR_CHECK_LAUNCH();
  return LFSR_OK;
}

bool lfsr_epi_use_b3() {
  if (lfsr_arith_f32()) return false;
  const char* esel = lfsr_sel("LFSR_EPI");
  return !(esel && (esel[0] == 'w' || esel[0] == 'd' || esel[0] == 'f' || esel[0] == 'g'));     // wino | direct | f32 | gather select an fp32-MFMA form
}

static size_t epi_wino_smem() { return (size_t)(LINES * 36 * LROW + 6 * 32 * LROW) * 4 + (LINES + 4) * 4 + (size_t)5 * 32 * TROW * 4; }

size_t lfsr_epi_fused_smem(int A) {
  return (size_t)(LINES * (32 + A - 1) * LROW + A * 32 * LROW) * 4 + (LINES + 4) * 4 + (size_t)A * 32 * TROW * 4;
}

bool lfsr_epi_fused_ok(int A, int h, int w) {
  return (A & 1) && A >= 1 && A <= 5 && h <= 32 && w <= 32 && lfsr_epi_fused_smem(A) <= 160 * 1024;
}

int lfsr_epi_fused_launch(const float* x, int x_stride, int x_choff, const float* w1_packed, const float* w2_packed, float* y, int y_stride,
                          int choffH, int choffV, float* t_h, float* t_v, int B, int A, int h, int w, int which, float slope, hipStream_t st) {
  LfsrOpTimer op_t("epi_fused", B, h * w, st);
  // which: 1 = horizontal only, 2 = vertical only, 3 = both
  if (!lfsr_epi_fused_ok(A, h, w)) return LFSR_E_ARG;
  if (A == 5 && lfsr_epi_use_b3()) {   // default at angRes 5: epi_b3.hip (exact three-term bf16 operands on the bf16 MFMA pipe); LFSR_EPI=wino | direct: this file's fp32-MFMA kernels
    const int rc = lfsr_epi_b3_launch(x, x_stride, x_choff, w1_packed + 25 * 32 * 64 + LFSR_EPI_WINO_FLOATS, w2_packed + 160 * 32, y, y_stride, choffH, choffV,
                                      t_h, t_v, B, A, h, w, which, slope, st);
    if (rc != LFSR_E_ARG) return rc;
  }
  if ((long long)B * A * A * h * w * x_stride * 4 >= (1LL << 31)) return LFSR_E_ARG;   // 32-bit byte offsets into x
  static std::atomic<bool> attr_set[64];
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return LFSR_E_ARG;
  const int smem = (int)lfsr_epi_fused_smem(5);
  if (!attr_set[dev]) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_epi_fused<5>), hipFuncAttributeMaxDynamicSharedMemorySize, smem);
    if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_epi_fused<0>), hipFuncAttributeMaxDynamicSharedMemorySize, smem);
    if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_epi_wino5), hipFuncAttributeMaxDynamicSharedMemorySize, (int)epi_wino_smem());
    if (e != hipSuccess) return LFSR_HIP_ERR(e);
    attr_set[dev] = true;
  }
  EpiArgs p{};
  p.X = x; p.x_stride = x_stride; p.x_choff = x_choff; p.W1 = w1_packed; p.W2 = w2_packed;
  p.Y = y; p.y_stride = y_stride; p.choffH = choffH; p.choffV = choffV; p.TH = t_h; p.TV = t_v;
  p.B = B; p.A = A; p.H = h; p.W = w; p.slope = slope;
  p.x_bytes = (int)((long long)B * A * A * h * w * x_stride * 4);
  p.tilesH = (which & 1) ? (B * A * h + LINES - 1) / LINES : 0;
  p.tilesV = (which & 2) ? (B * A * w + LINES - 1) / LINES : 0;
  int grid = p.tilesH + p.tilesV;
  if (grid <= 0) return LFSR_E_ARG;
  // item-major block order when both passes run and an item's lines fill whole tiles (LFSR_EPI_ORDER=pass keeps pass-major order: A/B runs)
  {
    const char* osel = lfsr_sel("LFSR_EPI_ORDER");
    if (which == 3 && (A * h) % LINES == 0 && (A * w) % LINES == 0 && !(osel && osel[0] == 'p')) { p.tpiH = A * h / LINES; p.tpiV = A * w / LINES; p.xcd_swz = (grid % 8 == 0) && !(osel && osel[0] == 'i'); }
  }
  // A = 5: stage 1 in Winograd F(2,5) form (the pack appended to the direct one by lfsr_pack_conv_weight); LFSR_EPI=direct keeps the direct form (A/B runs)
  const char* esel = lfsr_sel("LFSR_EPI");
  p.W1u = w1_packed + 25 * 32 * 64;
  if (A == 5 && !(esel && esel[0] == 'd')) hipLaunchKernelGGL(k_epi_wino5, dim3((unsigned)grid), dim3(512), epi_wino_smem(), st, p);
  else if (A == 5) hipLaunchKernelGGL(k_epi_fused<5>, dim3((unsigned)grid), dim3(512), lfsr_epi_fused_smem(A), st, p);
  else hipLaunchKernelGGL(k_epi_fused<0>, dim3((unsigned)grid), dim3(512), lfsr_epi_fused_smem(A), st, p);
  LFSR_CHECK_LAUNCH();
  return LFSR_OK;
}
