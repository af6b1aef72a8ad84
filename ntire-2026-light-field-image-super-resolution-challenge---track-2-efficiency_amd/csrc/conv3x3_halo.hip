// Per-view 3x3 conv, zero pad 1, 64 -> 64 channels, VCL layout, fp32 MFMA -- the dominant kernel
// (53 launches = 77 % of DistgSSR's FLOPs; also EPIT/LFT's Conv3d(1,3,3)).
// Reference: the MacPI convs "k3, dilation A, padding A" of model/SR/DistgSSR.py:22,47,64,79-83,101.
//
// "halo tile", persistent: one 512-thread block (8 waves) per CU walks 8-row x 32-column tiles of view images.  A tile's
// (8+2) x (32+2) input halo is staged into LDS once (zero-filled outside the image = the conv's padding)
// and all 9 taps read it at shifted addresses, so activations cross L2->LDS ~1.3x instead of 9x, and the
// only per-tap traffic is the 16 KB weight slab, double-buffered in LDS behind ONE barrier per tap.
// Wave w owns image row w of the tile (32 pixels = the 32 A-rows of a 32x32x2 MFMA) x all 64 output
// channels (2 column tiles, 32 accumulator registers).  K order inside a tap is permuted identically for
// A and W (lane half h takes k = 8j+4h..+3 from one ds_read_b128), products and accumulation are exact fp32.
// LDS: 340 x 272 B (halo, rows padded to 68 floats -> conflict-free ds_read_b128) + 2 x 64 x 272 B = 124 KB,
// one block per CU, 2 waves per SIMD.  The epilogue transposes the accumulators through the (then dead)
// halo region so that stores/residual loads are 16 B per lane, 256 B contiguous per pixel.
#include <stdlib.h>

#include "lfsr_internal.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

#ifdef LFSR_CONV_DIAG
// diagnostic build only: wave 0 accumulates s_memtime deltas per segment, written to the buffer passed as R2
#define STAMP(k) do { if (wave == 0) { long long t_ = clock64(); seg[k] += t_ - tprev; tprev = t_; } } while (0)
#else
#define STAMP(k) do { } while (0)
#endif

namespace {

constexpr int TR = 8, TC = 32, LROW = 68;
constexpr int HALO_PIX = (TR + 2) * (TC + 2);                 // 340
constexpr int SA_FLOATS = HALO_PIX * LROW;                    // 23120
constexpr int SB_FLOATS = 64 * LROW;                          // per buffer
constexpr int SMEM_BYTES = (SA_FLOATS + 3 * SB_FLOATS) * 4;   // 144704: halo + 3 weight buffers

struct ConvArgs {
  const float* X; int x_stride; int x_choff;
  const float* Wp;  // [9][64][64]  (tap, n, k)
  float* Y; int y_stride; int y_choff;
  const float* R1; int r1_stride; int r1_choff;
  const float* R2; int r2_stride; int r2_choff;
  const float* Mk; int mk_stride; int mk_choff; float mk_slope;   // backward: out *= (Mk > 0 ? 1 : mk_slope), before the residual adds
  int n_img, H, W, tiles_y, tiles_x;
  int tile_begin, tile_count;   // tiles [tile_begin, tile_begin + tile_count) belong to this launch
  float slope;
  float* dbg;                   // (LFSR_CONV_DIAG builds: stamp buffer; else unused)
};

// MASK = false: forward / plain dgrad (optional residuals R1, R2).  MASK = true: dgrad through a LeakyReLU, the operand
// prefetched at tap 8 is the saved activation Mk instead of R1 (the two never occur together on the hot path).
// NHALF = true: tail launch.  When the tile count is not a multiple of the CU count the last round would leave CUs idle for a
// whole tile time; the leftover tiles are instead given to TWO blocks each, block (2t + nh) computing output channels
// [32 nh, 32 nh + 32) of tile t (half the MFMAs, half the time), one tile per block.
template <bool MASK, bool NHALF>
__global__ __launch_bounds__(512) void k_conv3x3_halo(ConvArgs p) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* sA = smem;
  float* sB = smem + SA_FLOATS;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int c16 = tid & 15, r16 = tid >> 4;
  const int half = lane >> 5, l31 = lane & 31;
  const int tiles_per_img = p.tiles_y * p.tiles_x;
  const int ntiles = p.tile_begin + p.tile_count;   // end of this launch's tile range
  const int nh = NHALF ? (int)(blockIdx.x & 1) : 0;
  (void)tiles_per_img;

  // per-thread halo slots: slot i covers (pixel, 16-B chunk) = (tid + 512 i) >> 4, tid & 15
  int hoff[11];   // offset inside the image (pixels) relative to the tile origin, or INT_MIN if slot unused
#pragma unroll
  for (int i = 0; i < 11; ++i) {
    int pix = (tid + i * 512) >> 4;
    hoff[i] = pix < HALO_PIX ? pix : -1;
  }

  auto tile_origin = [&](int t, int& img, int& y0, int& x0) {
    int tx = t % p.tiles_x; int q = t / p.tiles_x;
    int ty = q % p.tiles_y; img = q / p.tiles_y;
    y0 = ty * TR; x0 = tx * TC;
  };
  auto halo_load = [&](int i, int img, int y0, int x0) -> float4 {
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    int pix = hoff[i];
    if (pix >= 0) {
      int r = pix / (TC + 2), c = pix - r * (TC + 2);
      int yy = y0 + r - 1, xx = x0 + c - 1;
      if (yy >= 0 && yy < p.H && xx >= 0 && xx < p.W)
        v = *reinterpret_cast<const float4*>(p.X + ((long long)img * p.H * p.W + (long long)yy * p.W + xx) * p.x_stride + p.x_choff + c16 * 4);
    }
    return v;
  };

  const float* aBase = sA + ((wave + 1) * (TC + 2) + (l31 + 1)) * LROW + 4 * half;   // tap (0,0) position
  const float* bBase = sB + (nh * 32 + l31) * LROW + 4 * half;
  float* sO = sA + wave * 32 * LROW;   // epilogue transposition region (wave-private, inside the dead halo)

#ifdef LFSR_CONV_DIAG
  long long seg[6] = {0, 0, 0, 0, 0, 0};
  long long tprev = clock64();
  float* dbgbuf = p.dbg;
#endif
  int tile = p.tile_begin + (NHALF ? (int)(blockIdx.x >> 1) : (int)blockIdx.x);
  int img, y0, x0;
  tile_origin(tile, img, y0, x0);
  float4 hv[11];
#pragma unroll
  for (int i = 0; i < 11; ++i) hv[i] = halo_load(i, img, y0, x0);
  // weight stream: tap t's slab lives in LDS buffer t%3 (9%3==0, so the stream is periodic across tiles);
  // slab t+1 is written at the START of tap t from registers filled during tap t-1, and made visible by ONE
  // barrier placed in the MIDDLE of tap t's MFMA stream (fragments already in flight -> the pipe does not drain).
  {
    float4 w0 = *reinterpret_cast<const float4*>(p.Wp + r16 * 64 + c16 * 4);
    float4 w1 = *reinterpret_cast<const float4*>(p.Wp + (r16 + 32) * 64 + c16 * 4);
    *reinterpret_cast<float4*>(sB + r16 * LROW + c16 * 4) = w0;
    *reinterpret_cast<float4*>(sB + (r16 + 32) * LROW + c16 * 4) = w1;
  }
  float4 nb0 = *reinterpret_cast<const float4*>(p.Wp + 64 * 64 + r16 * 64 + c16 * 4);          // slab 1
  float4 nb1 = *reinterpret_cast<const float4*>(p.Wp + 64 * 64 + (r16 + 32) * 64 + c16 * 4);

  while (true) {
    // ---- registers -> LDS: this tile's halo and tap 0's weights ------------------------------------
#pragma unroll
    for (int i = 0; i < 11; ++i)
      if (hoff[i] >= 0) *reinterpret_cast<float4*>(sA + hoff[i] * LROW + c16 * 4) = hv[i];
    // LDS-only barrier: the previous tile's output stores stay in flight (a __syncthreads would drain vmcnt)
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    STAMP(4);   // halo LDS write + barrier

    const int next = NHALF ? ntiles : tile + (int)gridDim.x;
    const bool has_next = next < ntiles;
    int nimg = 0, ny0 = 0, nx0 = 0;
    if (has_next) tile_origin(next, nimg, ny0, nx0);

    f32x16 acc0, acc1;
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; }
    float4 res[8];
    float4 fa, fb0, fb1;
    const int yy = y0 + wave;
    const long long row_base = (long long)img * p.H * p.W + (long long)yy * p.W + x0;

#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      // slab tap+1 (loaded during the previous tap) -> LDS buffer (tap+1)%3, last read two taps ago;
      // then start fetching slab tap+2
      {
        float* bN = sB + ((tap + 1) % 3) * SB_FLOATS;
        *reinterpret_cast<float4*>(bN + r16 * LROW + c16 * 4) = nb0;
        *reinterpret_cast<float4*>(bN + (r16 + 32) * LROW + c16 * 4) = nb1;
        const float* wsrc = p.Wp + (long long)((tap + 2) % 9) * 64 * 64;
        nb0 = *reinterpret_cast<const float4*>(wsrc + r16 * 64 + c16 * 4);
        nb1 = *reinterpret_cast<const float4*>(wsrc + (r16 + 32) * 64 + c16 * 4);
      }
      // the next tile's halo is prefetched into registers over taps 4..7, spreading the HBM reads over the
      // compute phase (all CUs run in phase, so a burst at the tile seam would be HBM-bound and exposed)
      if (has_next) {
        if (tap == 4) { hv[0] = halo_load(0, nimg, ny0, nx0); hv[1] = halo_load(1, nimg, ny0, nx0); hv[2] = halo_load(2, nimg, ny0, nx0); }
        if (tap == 5) { hv[3] = halo_load(3, nimg, ny0, nx0); hv[4] = halo_load(4, nimg, ny0, nx0); hv[5] = halo_load(5, nimg, ny0, nx0); }
        if (tap == 6) { hv[6] = halo_load(6, nimg, ny0, nx0); hv[7] = halo_load(7, nimg, ny0, nx0); hv[8] = halo_load(8, nimg, ny0, nx0); }
        if (tap == 7) { hv[9] = halo_load(9, nimg, ny0, nx0); hv[10] = halo_load(10, nimg, ny0, nx0); }
      }
      if (tap == 8 && (MASK ? p.Mk != nullptr : p.R1 != nullptr) && yy < p.H) {   // residual (backward: LeakyReLU' mask) operand
        const float* src = MASK ? p.Mk : p.R1;
        const int sst = MASK ? p.mk_stride : p.r1_stride, sco = MASK ? p.mk_choff : p.r1_choff;
#pragma unroll
        for (int i = 0; i < (NHALF ? 4 : 8); ++i) {
          const int pc = NHALF ? (lane >> 3) + 8 * i : (lane >> 4) + 4 * i;
          const int ch = NHALF ? nh * 8 + (lane & 7) : (lane & 15);
          res[i] = (x0 + pc < p.W) ? *reinterpret_cast<const float4*>(src + (row_base + pc) * sst + sco + ch * 4)
                                   : make_float4(0.f, 0.f, 0.f, 0.f);
        }
      }
      // MFMA stream, software-pipelined by hand: the fragments of step (tap, j+1) are requested from LDS
      // BEFORE the 8 MFMAs of step (tap, j) issue, so the two co-resident waves of a SIMD never both sit
      // in an LDS wait with the matrix pipe empty (they run in lock-step after every barrier).
      const int dy = tap / 3 - 1, dx = tap % 3 - 1;
      const float* aT = aBase + (dy * (TC + 2) + dx) * LROW;
      const float* bT = bBase + (tap % 3) * SB_FLOATS;
      if (tap == 0) {
        fa = *reinterpret_cast<const float4*>(aT);
        fb0 = *reinterpret_cast<const float4*>(bT);
        if (!NHALF) fb1 = *reinterpret_cast<const float4*>(bT + 32 * LROW);
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        if (j == 4) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");   // publishes slab tap+1
        float4 na, nb0f, nb1f;
        if (j < 7) {
          na = *reinterpret_cast<const float4*>(aT + 8 * (j + 1));
          nb0f = *reinterpret_cast<const float4*>(bT + 8 * (j + 1));
          if (!NHALF) nb1f = *reinterpret_cast<const float4*>(bT + 32 * LROW + 8 * (j + 1));
        } else if (tap < 8) {   // first fragments of the next tap (its slab was published by this tap's barrier)
          const int ndy = (tap + 1) / 3 - 1, ndx = (tap + 1) % 3 - 1;
          const float* naT = aBase + (ndy * (TC + 2) + ndx) * LROW;
          const float* nbT = bBase + ((tap + 1) % 3) * SB_FLOATS;
          na = *reinterpret_cast<const float4*>(naT);
          nb0f = *reinterpret_cast<const float4*>(nbT);
          if (!NHALF) nb1f = *reinterpret_cast<const float4*>(nbT + 32 * LROW);
        }
        if (NHALF) {
          __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
          acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(fa.x, fb0.x, acc0, 0, 0, 0);
          acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(fa.y, fb0.y, acc0, 0, 0, 0);
          acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(fa.z, fb0.z, acc0, 0, 0, 0);
          acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(fa.w, fb0.w, acc0, 0, 0, 0);
          __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
        } else {
          __builtin_amdgcn_sched_group_barrier(0x100, 3, 0);   // 3 DS reads first ...
          acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(fa.x, fb0.x, acc0, 0, 0, 0);
          acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(fa.x, fb1.x, acc1, 0, 0, 0);
          acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(fa.y, fb0.y, acc0, 0, 0, 0);
          acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(fa.y, fb1.y, acc1, 0, 0, 0);
          acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(fa.z, fb0.z, acc0, 0, 0, 0);
          acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(fa.z, fb1.z, acc1, 0, 0, 0);
          acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(fa.w, fb0.w, acc0, 0, 0, 0);
          acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(fa.w, fb1.w, acc1, 0, 0, 0);
          __builtin_amdgcn_sched_group_barrier(0x008, 8, 0);   // ... then the 8 MFMAs of this step
        }
        if (j < 7 || tap < 8) { fa = na; fb0 = nb0f; if (!NHALF) fb1 = nb1f; }
      }
    }
    STAMP(0);   // 9 taps
    __syncthreads();   // all waves are done reading the halo (and tap 8's slab) before the epilogue reuses the region
    STAMP(1);   // seam barrier wait

    // ---- epilogue: accumulators -> LDS [pixel][channel] -> 16-B stores (256 B contiguous per pixel) ----
    // C/D layout: channel n = lane&31 (+32 for acc1), pixel column = (reg&3) + 8*(reg>>2) + 4*half
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      int pc = (r & 3) + 8 * (r >> 2) + 4 * half;
      float v0 = acc0[r], v1 = acc1[r];
      v0 = v0 >= 0.f ? v0 : v0 * p.slope;
      v1 = v1 >= 0.f ? v1 : v1 * p.slope;
      sO[pc * LROW + nh * 32 + l31] = v0;
      if (!NHALF) sO[pc * LROW + 32 + l31] = v1;
    }
    __builtin_amdgcn_wave_barrier();
    if (yy < p.H) {
      const int ch = NHALF ? nh * 8 + (lane & 7) : (lane & 15);
#pragma unroll
      for (int i = 0; i < (NHALF ? 4 : 8); ++i) {
        const int pc = NHALF ? (lane >> 3) + 8 * i : (lane >> 4) + 4 * i;
        if (x0 + pc < p.W) {
          float4 v = *reinterpret_cast<const float4*>(sO + pc * LROW + ch * 4);
          long long pix = row_base + pc;
          if (MASK) {
            if (p.Mk) {
              v.x *= res[i].x > 0.f ? 1.f : p.mk_slope; v.y *= res[i].y > 0.f ? 1.f : p.mk_slope;
              v.z *= res[i].z > 0.f ? 1.f : p.mk_slope; v.w *= res[i].w > 0.f ? 1.f : p.mk_slope;
            }
            if (p.R1) {   // not on the hot path: late load
              float4 r = *reinterpret_cast<const float4*>(p.R1 + pix * p.r1_stride + p.r1_choff + ch * 4);
              v.x += r.x; v.y += r.y; v.z += r.z; v.w += r.w;
            }
          } else {
            if (p.R1) { v.x += res[i].x; v.y += res[i].y; v.z += res[i].z; v.w += res[i].w; }
          }
          if (p.R2) {
            float4 r = *reinterpret_cast<const float4*>(p.R2 + pix * p.r2_stride + p.r2_choff + ch * 4);
            v.x += r.x; v.y += r.y; v.z += r.z; v.w += r.w;
          }
          *reinterpret_cast<float4*>(p.Y + pix * p.y_stride + p.y_choff + ch * 4) = v;
        }
      }
    }
    STAMP(2);   // transposition + output stores issued
    if (!has_next) break;
    // every wave is done with its sO reads before the halo region is overwritten (LDS-only barrier, stores keep flying)
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    STAMP(3);   // barrier after the epilogue
    tile = next; img = nimg; y0 = ny0; x0 = nx0;
  }
#ifdef LFSR_CONV_DIAG
  if (dbgbuf && tid == 0)
    for (int k = 0; k < 6; ++k) dbgbuf[blockIdx.x * 8 + k] = (float)seg[k];
#endif
}

}  // namespace

// internal entry used by lfsr_conv3x3_fwd (gemm_gather.hip)
int lfsr_conv3x3_halo_launch(const float* x, int x_stride, int x_choff, const float* w_packed, float* y, int y_stride, int y_choff,
                             const float* r1, int r1_stride, int r1_choff, const float* r2, int r2_stride, int r2_choff,
                             const float* mk, int mk_stride, int mk_choff, float mk_slope,
                             int n_img, int h, int w, float slope, hipStream_t st) {
  static std::atomic<bool> attr_set[64];   // per device: the >64 KB dynamic-LDS opt-in is a per-device function attribute
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return LFSR_E_ARG;
  if (!attr_set[dev]) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_conv3x3_halo<false, false>), hipFuncAttributeMaxDynamicSharedMemorySize, SMEM_BYTES);
    if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_conv3x3_halo<true, false>), hipFuncAttributeMaxDynamicSharedMemorySize, SMEM_BYTES);
    if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_conv3x3_halo<false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, SMEM_BYTES);
    if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_conv3x3_halo<true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, SMEM_BYTES);
    if (e != hipSuccess) return LFSR_HIP_ERR(e);
    attr_set[dev] = true;
  }
  ConvArgs p{};
  p.X = x; p.x_stride = x_stride; p.x_choff = x_choff; p.Wp = w_packed;
  p.Y = y; p.y_stride = y_stride; p.y_choff = y_choff;
  p.R1 = r1; p.r1_stride = r1_stride; p.r1_choff = r1_choff; p.R2 = r2; p.r2_stride = r2_stride; p.r2_choff = r2_choff;
  p.Mk = mk; p.mk_stride = mk_stride; p.mk_choff = mk_choff; p.mk_slope = mk_slope;
  p.n_img = n_img; p.H = h; p.W = w; p.tiles_y = (h + TR - 1) / TR; p.tiles_x = (w + TC - 1) / TC; p.slope = slope;
#ifdef LFSR_CONV_DIAG
  p.dbg = g_lfsr_diag_buf;
#endif
  long long nblk = (long long)n_img * p.tiles_y * p.tiles_x;
  if (nblk <= 0 || nblk > 0x7fffffffLL) return LFSR_E_ARG;
  int ncu = 256;
  {
    static std::atomic<int> cus[64];
    if (!cus[dev]) {
      int v = 0;
      if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) cus[dev] = v; else cus[dev] = 256;
    }
    ncu = cus[dev];
  }
  // persistent: one 141-KB-LDS block per CU walks tiles blockIdx.x, +grid, ... (uniform cost, no queue needed).  Tiles beyond
  // the last full round (L = ntiles % ncu) go to a second, channel-split launch when that halves the tail (2L <= ncu).
  int tail = (int)(nblk % ncu);
  if (nblk < ncu || 2 * tail > ncu || lfsr_sel("LFSR_CONV_NOTAIL")) tail = 0;
  const int body = (int)nblk - tail;
  if (body > 0) {
    p.tile_begin = 0; p.tile_count = body;
    unsigned grid = (unsigned)(body < ncu ? body : ncu);
    if (mk) hipLaunchKernelGGL((k_conv3x3_halo<true, false>), dim3(grid), dim3(512), SMEM_BYTES, st, p);
    else hipLaunchKernelGGL((k_conv3x3_halo<false, false>), dim3(grid), dim3(512), SMEM_BYTES, st, p);
    LFSR_CHECK_LAUNCH();
  }
  if (tail > 0) {
    p.tile_begin = body; p.tile_count = tail;
    if (mk) hipLaunchKernelGGL((k_conv3x3_halo<true, true>), dim3(2 * tail), dim3(512), SMEM_BYTES, st, p);
    else hipLaunchKernelGGL((k_conv3x3_halo<false, true>), dim3(2 * tail), dim3(512), SMEM_BYTES, st, p);
    LFSR_CHECK_LAUNCH();
  }
  return LFSR_OK;
}

// channel-split launch over tiles [tile_begin, tile_begin + tile_count) only: the tail of the Winograd kernel (conv3x3_wino.hip)
int lfsr_conv3x3_halo_tail_launch(const float* x, int x_stride, int x_choff, const float* w_packed, float* y, int y_stride, int y_choff,
                                  const float* r1, int r1_stride, int r1_choff, const float* r2, int r2_stride, int r2_choff,
                                  const float* mk, int mk_stride, int mk_choff, float mk_slope,
                                  int n_img, int h, int w, float slope, int tile_begin, int tile_count, hipStream_t st) {
  static std::atomic<bool> attr_set[64];
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return LFSR_E_ARG;
  if (!attr_set[dev]) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_conv3x3_halo<false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, SMEM_BYTES);
    if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_conv3x3_halo<true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, SMEM_BYTES);
    if (e != hipSuccess) return LFSR_HIP_ERR(e);
    attr_set[dev] = true;
  }
  if (tile_count <= 0) return LFSR_OK;
  ConvArgs p{};
  p.X = x; p.x_stride = x_stride; p.x_choff = x_choff; p.Wp = w_packed;
  p.Y = y; p.y_stride = y_stride; p.y_choff = y_choff;
  p.R1 = r1; p.r1_stride = r1_stride; p.r1_choff = r1_choff; p.R2 = r2; p.r2_stride = r2_stride; p.r2_choff = r2_choff;
  p.Mk = mk; p.mk_stride = mk_stride; p.mk_choff = mk_choff; p.mk_slope = mk_slope;
  p.n_img = n_img; p.H = h; p.W = w; p.tiles_y = (h + TR - 1) / TR; p.tiles_x = (w + TC - 1) / TC; p.slope = slope;
  p.tile_begin = tile_begin; p.tile_count = tile_count;
  if (mk) hipLaunchKernelGGL((k_conv3x3_halo<true, true>), dim3(2 * tile_count), dim3(512), SMEM_BYTES, st, p);
  else hipLaunchKernelGGL((k_conv3x3_halo<false, true>), dim3(2 * tile_count), dim3(512), SMEM_BYTES, st, p);
  LFSR_CHECK_LAUNCH();
  return LFSR_OK;
}
