// Per-view 3x3 conv, zero pad 1, 64 -> 64 channels, VCL layout -- Winograd F(4x4, 3x3) on the fp32 MFMA pipe, SYMMETRIC-wave form.
// Reference: the MacPI convs "k3, dilation A, padding A" of model/SR/DistgSSR.py:22,47,64,79-83,101 (per-view 3x3 in VCL),
// EPIT.py:24-32,136-142 / LFT.py:36-46 (Conv3d(1,3,3)).  Same arithmetic as conv3x3_wino4.hip (the same transforms, the same packed
// U = G g G^t, the same fma sequences), a different division of labour.
//
// What round 2 measured on the specialised kernel (4 MFMA "consumer" waves + 4 "producer" waves, conv3x3_wino4.hip):
// fp32 MFMAs and VALU instructions share a SIMD's FMA lanes, and while one wave streams v_mfma_f32_16x16x4_f32 back to back the OTHER
// wave of that SIMD retires practically no VALU instruction (its LDS / memory instructions do issue).  The producers' input transform and
// epilogue arithmetic therefore ran only after the consumers' MFMA stream of a chunk had ended -- one wave per SIMD at a time, at half the
// VALU issue rate a SIMD sustains with two waves -- and the consumers idled meanwhile: 36 k cycles per tile for 18.4 k cycles of MFMAs.
//
// Here all 8 waves are alike and the two kinds of work alternate in PHASES that every wave is in at the same time:
//   MFMA phase   wave (ns, ph): output channels 16 ns .. + 15 of all 16 Winograd tiles, transform rows xi = 3 ph .. 3 ph + 2 = 18 of the 36
//                positions: 72 accumulator registers, 144 MFMAs per 32 input channels.  Two MFMA waves per SIMD share its matrix pipe.
//   VALU phase   every thread: one (Winograd tile, channel) item of the input transform (packed-math column pass), its share of the halo
//                staging, and its share of the previous tile's epilogue (exchange plane -> residuals / mask -> 16-B stores).  Two VALU
//                waves per SIMD fill the FMA lanes that one wave alone leaves half empty.
// A tile = 2 steps of 32 input channels (V[2][16 tiles][16 ch][36] in LDS as before, now ONE 32-channel buffer).  The inverse transform
// At M A is split with the positions: each wave forms the row pass of its three xi rows and its partial column sums, the two waves of a pair
// exchange partials through the LDS exchange planes (each finishes two of the four planes), LeakyReLU there.
#include <stdlib.h>

#include <type_traits>

#include "lfsr_internal.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

#ifdef LFSR_CONV_DIAG
#define SSTAMP(k) do { if (wave == 0) { long long t_ = clock64(); seg[k] += (unsigned)(t_ - tprev); tprev = t_; } } while (0)
#else
#define SSTAMP(k) do { } while (0)
#endif

namespace {

constexpr int TS = 584;                 // floats per tile in a V buffer: 16 channels x 36 positions + 8 (2336 B = 32 mod 256)
constexpr int VBUF = 16 * TS;           // one 16-channel chunk
constexpr int HPIX = 10 * 34;           // raw halo of an 8 x 32 tile
constexpr int HBUF = HPIX * 16;         // one 16-channel chunk, 64 B per pixel
constexpr int SMEM_BYTES = (2 * VBUF + 4 * 4096 + HBUF + 256) * 4;   // V (32 channels), exchange planes, raw halo + landing zone: 163072
constexpr int OOB = (int)0x80000000u;
constexpr int UREC = 1152;              // floats of U per (stage, ns, ph): 4 x 64 x 4 (positions in fours) + 64 x 2 (the last two)

struct Wino4sArgs {
  const float* X; int x_stride; int x_choff;
  const float* Wu;   // [16 stages][4 ns][2 ph]{[4 q][64 lanes][4], [64 lanes][2]}   (lfsr_pack_wino4s)
  float* Y; int y_stride; int y_choff;
  const float* R1; int r1_stride; int r1_choff;
  const float* R2; int r2_stride; int r2_choff;
  const float* Mk; int mk_stride; int mk_choff; float mk_slope;
  int n_img, H, W, ntiles;
  float slope;
  float* dbg;
};

#define LDS_BARRIER() do { __builtin_amdgcn_sched_barrier(0); asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); __builtin_amdgcn_sched_barrier(0); } while (0)

__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* base, int bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, bytes, 0x00020000);
}
__device__ __forceinline__ f32x4 bload4(__amdgpu_buffer_rsrc_t r, int voff, int soff) {
  return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0));
}
__device__ __forceinline__ f32x2 bload2(__amdgpu_buffer_rsrc_t r, int voff, int soff) {
  return __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(r, voff, soff, 0));
}
__device__ __forceinline__ void bstore4(__amdgpu_buffer_rsrc_t r, int voff, f32x4 v) {
  __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), r, voff, 0, 0);
}

// one 6-vector of the input transform: t = Bt d  (12 operations, integer coefficients: exact products) -- as conv3x3_wino4.hip
__device__ __forceinline__ void bt6(float& d0, float& d1, float& d2, float& d3, float& d4, float& d5) {
  const float a = fmaf(-4.f, d2, d4), b = fmaf(-4.f, d1, d3);
  const float c = d4 - d2, e = d3 - d1;
  const float t0 = fmaf(4.f, d0, fmaf(-5.f, d2, d4));
  const float t5 = fmaf(4.f, d1, fmaf(-5.f, d3, d5));
  d0 = t0; d1 = a + b; d2 = a - b; d3 = fmaf(2.f, e, c); d4 = fmaf(-2.f, e, c); d5 = t5;
}
// one 6-vector of the output transform: y = At m  (4 results in m0..m3) -- as conv3x3_wino4.hip
__device__ __forceinline__ void at6(f32x4& m0, f32x4& m1, f32x4& m2, f32x4& m3, const f32x4 m4, const f32x4 m5) {
  const f32x4 s12 = m1 + m2, d12 = m1 - m2, s34 = m3 + m4, d34 = m3 - m4;
  m0 = (m0 + s12) + s34;
  m1 = d12 + 2.f * d34;
  m2 = s12 + 4.f * s34;
  m3 = (d12 + 8.f * d34) + m5;
}

// accumulator j (0..17) of a wave <-> position: ph = 0: p = j;  ph = 1: j < 16 -> p = 20 + j, j = 16, 17 -> p = 18, 19 (so that in both
// halves entries 0..3 are 16-B groups of four positions and entry 4 the 8-B pair: one MFMA-phase code for both)
__host__ __device__ constexpr int pos_of(int ph, int j) { return ph == 0 ? j : (j < 16 ? 20 + j : 2 + j); }
__host__ __device__ constexpr int acc_of(int ph, int p) { return ph == 0 ? p : (p >= 20 ? p - 20 : p - 2); }

template <bool MASK, bool HAS_E, bool HAS_L>
__global__ __launch_bounds__(512) void k_conv3x3_wino4s(Wino4sArgs p) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* const sV = smem;                 // V[2 chunks of the step][16 tiles][16 ch][36 positions]
  float* const sX = smem + 2 * VBUF;      // epilogue exchange: [4 rows a][64 pixels][64 channels]
  float* const sH = sX + 4 * 4096;        // raw halo of one 16-channel chunk: [340 pixels][16 channels], then the landing zone
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, w4 = wave & 3, ph = wave >> 2;
  const int l15 = lane & 15, kk = lane >> 4;
  const int nblk = gridDim.x;
  // a block walks a CONTIGUOUS range of tiles
  int tile = (int)(((long long)blockIdx.x * p.ntiles) / nblk);
  const int tile_end = (int)(((long long)(blockIdx.x + 1) * p.ntiles) / nblk);
#ifdef LFSR_CONV_DIAG
  unsigned seg[16] = {};
  long long tprev = clock64();
#endif

  // ================================================= VALU-phase state (every thread) ==========================================
  const float* const Ep = MASK ? p.Mk : p.R1;
  const float* const Lp = MASK ? p.R1 : p.R2;
  const int e_stride = MASK ? p.mk_stride : p.r1_stride, e_choff = MASK ? p.mk_choff : p.r1_choff;
  const int l_stride = MASK ? p.r1_stride : p.r2_stride, l_choff = MASK ? p.r1_choff : p.r2_choff;
  const int img_px = p.H * p.W;
  // one descriptor per (operand, image): rows above / below the image are out-of-range offsets; an absent operand / tile has extent 0
  auto img_rsrc = [&](const float* base, int stride, int img) {
    const bool ok = base != nullptr && img >= 0;
    return make_rsrc(base + (ok ? (long long)img * img_px * stride : 0), ok ? img_px * stride * 4 : 0);
  };
  // ---- halo staging slots: slot i of a thread = (halo pixel, 16-B quarter of the chunk's 64 B) = (idx >> 2, idx & 3), idx = tid + 512 i
  int hrel[3], hcol[3];
  float* hdst[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const int px = (tid + 512 * i) >> 2, cq = tid & 3;
    const int r = __mul24(px, 1928) >> 16;   // px / 34 for px < 384
    const int c = px - r * 34;
    hrel[i] = (((r - 1) * p.W + (c - 1)) * p.x_stride + p.x_choff) * 4 + cq * 16;
    hcol[i] = px < HPIX ? c - 1 : -(1 << 20);
    hdst[i] = px < HPIX ? sH + px * 16 + cq * 4 : sH + HBUF + (tid & 63) * 4;   // slots 340..383: landing zone
  }
  int hx[3];
  auto halo_offsets = [&](int y0, int x0) {
    const int toff = (y0 * p.W + x0) * (p.x_stride * 4);
#pragma unroll
    for (int i = 0; i < 3; ++i) hx[i] = (unsigned)(x0 + hcol[i]) < (unsigned)p.W ? hrel[i] + toff : OOB;
  };
  f32x4 hv[2][3];   // the two 16-channel chunks of the NEXT step, requested one step ahead
  auto halo_load = [&](__amdgpu_buffer_rsrc_t rs, int step) {
#pragma unroll
    for (int k = 0; k < 2; ++k)
#pragma unroll
      for (int i = 0; i < 3; ++i) hv[k][i] = bload4(rs, hx[i], (2 * step + k) * 64);
  };
  auto halo_store = [&](int k) {
#pragma unroll
    for (int i = 0; i < 3; ++i) *reinterpret_cast<f32x4*>(hdst[i]) = hv[k][i];
  };
  // ---- input transform of one (Winograd tile, channel) item per thread and step: waves 0..3 chunk 2 s, waves 4..7 chunk 2 s + 1
  const int c16 = lane & 15, ptile = 4 * w4 + (lane >> 4), pty = ptile >> 3, ptx = ptile & 7;
  const float* const hR = sH + ((4 * pty) * 34 + 4 * ptx) * 16 + c16;
  f32x2 R[18];
  auto read_raw = [&]() {
#pragma unroll
    for (int r = 0; r < 6; ++r)
#pragma unroll
      for (int k = 0; k < 3; ++k) { R[3 * r + k].x = hR[(r * 34 + 2 * k) * 16]; R[3 * r + k].y = hR[(r * 34 + 2 * k + 1) * 16]; }
  };
  auto transform = [&]() {
#pragma unroll
    for (int k = 0; k < 3; ++k) {   // column pass, two columns per instruction (the fma sequence of bt6 per element: bit-identical)
      f32x2 &d0 = R[k], &d1 = R[3 + k], &d2 = R[6 + k], &d3 = R[9 + k], &d4 = R[12 + k], &d5 = R[15 + k];
      const f32x2 m4 = {-4.f, -4.f}, p4 = {4.f, 4.f}, m5 = {-5.f, -5.f}, p2 = {2.f, 2.f}, m2 = {-2.f, -2.f};
      const f32x2 a = __builtin_elementwise_fma(m4, d2, d4), b = __builtin_elementwise_fma(m4, d1, d3);
      const f32x2 c = d4 - d2, e = d3 - d1;
      const f32x2 t0 = __builtin_elementwise_fma(p4, d0, __builtin_elementwise_fma(m5, d2, d4));
      const f32x2 t5 = __builtin_elementwise_fma(p4, d1, __builtin_elementwise_fma(m5, d3, d5));
      d0 = t0; d1 = a + b; d2 = a - b; d3 = __builtin_elementwise_fma(p2, e, c); d4 = __builtin_elementwise_fma(m2, e, c); d5 = t5;
    }
#pragma unroll
    for (int r = 0; r < 6; ++r) {
      float d0 = R[3 * r].x, d1 = R[3 * r].y, d2 = R[3 * r + 1].x, d3 = R[3 * r + 1].y, d4 = R[3 * r + 2].x, d5 = R[3 * r + 2].y;
      bt6(d0, d1, d2, d3, d4, d5);
      R[3 * r].x = d0; R[3 * r].y = d1; R[3 * r + 1].x = d2; R[3 * r + 1].y = d3; R[3 * r + 2].x = d4; R[3 * r + 2].y = d5;
    }
  };
  float* const vW = sV + ph * VBUF + ptile * TS + c16 * 36;   // this thread's item goes to the chunk buffer of its wave half
  auto write_v = [&]() {
#pragma unroll
    for (int q = 0; q < 9; ++q) {
      f32x4 v; v.x = R[2 * q].x; v.y = R[2 * q].y; v.z = R[2 * q + 1].x; v.w = R[2 * q + 1].y;
      *reinterpret_cast<f32x4*>(vW + 4 * q) = v;
    }
  };
  // ---- drain slots of an exchange plane: slot i of a thread = pixel (row 4 i [+ a], column tid / 16) of the tile, 16-B unit tid % 16
  const int un = tid & 15;
  int pY[2], pE[2], pL[2];
  const float* dsrc[2];
  const int dcol = tid >> 4;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int px = (tid + 512 * i) >> 4;   // = 32 i + tid / 16
    const int pr = (4 * i) * p.W + dcol;
    pY[i] = (pr * p.y_stride + p.y_choff) * 4 + un * 16;
    pE[i] = (pr * e_stride + e_choff) * 4 + un * 16;
    pL[i] = (pr * l_stride + l_choff) * 4 + un * 16;
    dsrc[i] = sX + px * 64 + ((un ^ ((px >> 2) & 7)) << 2);
  }
  __amdgpu_buffer_rsrc_t rsYp = img_rsrc(nullptr, 0, -1), rsEp = rsYp, rsLp = rsYp;   // the PREVIOUS tile's image (none yet)
  int prow0 = 0, pcol0 = 0;
  const bool ragged_w = (p.W & 31) != 0;
  // one exchange plane a of the previous tile: operands requested, plane read back as whole pixels, mask / residuals, 16-B stores
  // (the forward activation was applied when the plane was finished)
  auto drain_plane = [&](int a) {
    const int rowoff = (prow0 + a) * p.W + pcol0;   // wave-uniform pixel offset of the plane within the image
    const bool bad = ragged_w && pcol0 + dcol >= p.W;
    f32x4 e[2];
    if (HAS_E) {
#pragma unroll
      for (int i = 0; i < 2; ++i) { const int offe = pE[i] + rowoff * (e_stride * 4); e[i] = bload4(rsEp, bad ? OOB : offe, 0); }
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      f32x4 v = *reinterpret_cast<const f32x4*>(dsrc[i] + a * 4096);
      if (MASK) {
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] *= e[i][k] > 0.f ? 1.f : p.mk_slope;
      } else if (HAS_E) {
        v += e[i];
      }
      if (HAS_L) { const int offl = pL[i] + rowoff * (l_stride * 4); v += bload4(rsLp, bad ? OOB : offl, 0); }
      const int offy = pY[i] + rowoff * (p.y_stride * 4);
      bstore4(rsYp, bad ? OOB : offy, v);
    }
  };

  // ================================================= MFMA-phase state ===========================================================
  const int ns = w4, ctile = l15, cty = ctile >> 3, ctx = ctile & 7;
  const __amdgpu_buffer_rsrc_t rsW = make_rsrc(p.Wu, 36 * 64 * 64 * 4);
  const int uq_off = (ns * 2 + ph) * (UREC * 4) + lane * 16;            // + stage * 8 * UREC * 4 + entry * 1024
  const int ut_off = (ns * 2 + ph) * (UREC * 4) + 4096 + lane * 8;      // + stage * 8 * UREC * 4
  const float* const vR = sV + ctile * TS + kk * 36 + (ph ? 20 : 0);    // + chunk * VBUF + s4 * 144 + 4 entry
  const float* const vT = sV + ctile * TS + kk * 36 + (ph ? 18 : 16);   // the 8-B pair of positions
  const int xw = (cty * 32 + 4 * ctx) * 64 + (((4 * ns + kk) ^ ctx) << 2);   // exchange: pixel (row cty, col 4 ctx + b), 16-B unit XOR ctx
  constexpr int STAGE_BYTES = 8 * UREC * 4;
  f32x4 Uq[2][4];
  f32x2 Ut[2];
  auto u_load = [&](int slot, int stage) {   // stage: 0..15 (wraps into the next tile: same weights)
#pragma unroll
    for (int i = 0; i < 4; ++i) Uq[slot][i] = bload4(rsW, uq_off, stage * STAGE_BYTES + i * 1024);
    Ut[slot] = bload2(rsW, ut_off, stage * STAGE_BYTES);
  };
  f32x4 acc[18];
  const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
  // 8 stages (32 input channels) of step s: V entries from LDS one entry ahead, U from the two-stage ring, refilled entry by entry
  auto mfma_phase = [&](auto S) {
    constexpr int s = decltype(S)::value;
#pragma unroll
    for (int st = 0; st < 8; ++st) {
      const int g = 8 * s + st, slot = st & 1;
      const float* vq = vR + (st >> 2) * VBUF + (st & 3) * 144;
      const float* vt = vT + (st >> 2) * VBUF + (st & 3) * 144;
      f32x4 v[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) v[i] = *reinterpret_cast<const f32x4*>(vq + 4 * i);
      const f32x2 vt2 = *reinterpret_cast<const f32x2*>(vt);
      const bool first = (s == 0 && st == 0);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const f32x4 u = Uq[slot][i];
        acc[4 * i + 0] = __builtin_amdgcn_mfma_f32_16x16x4f32(u.x, v[i].x, first ? zero4 : acc[4 * i + 0], 0, 0, 0);
        acc[4 * i + 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(u.y, v[i].y, first ? zero4 : acc[4 * i + 1], 0, 0, 0);
        acc[4 * i + 2] = __builtin_amdgcn_mfma_f32_16x16x4f32(u.z, v[i].z, first ? zero4 : acc[4 * i + 2], 0, 0, 0);
        acc[4 * i + 3] = __builtin_amdgcn_mfma_f32_16x16x4f32(u.w, v[i].w, first ? zero4 : acc[4 * i + 3], 0, 0, 0);
        Uq[slot][i] = bload4(rsW, uq_off, ((g + 2) & 15) * STAGE_BYTES + i * 1024);
      }
      {
        const f32x2 u = Ut[slot];
        acc[16] = __builtin_amdgcn_mfma_f32_16x16x4f32(u.x, vt2.x, first ? zero4 : acc[16], 0, 0, 0);
        acc[17] = __builtin_amdgcn_mfma_f32_16x16x4f32(u.y, vt2.y, first ? zero4 : acc[17], 0, 0, 0);
        Ut[slot] = bload2(rsW, ut_off, ((g + 2) & 15) * STAGE_BYTES);
      }
    }
  };

  // ================================================= the walk ====================================================================
  auto tile_origin = [&](int t, int& img, int& y0, int& x0) {
    const int tiles_x = (p.W + 31) / 32, tiles_y = (p.H + 7) / 8;
    int txx = t % tiles_x; int q = t / tiles_x;
    int tyy = q % tiles_y; img = q / tiles_y;
    y0 = tyy * 8; x0 = txx * 32;
  };
  int img, y0, x0;
  tile_origin(tile, img, y0, x0);
  __amdgpu_buffer_rsrc_t rsXc = img_rsrc(p.X, p.x_stride, img);
  halo_offsets(y0, x0);
  halo_load(rsXc, 0);           // step 0 of the first tile
  u_load(0, 0);
  u_load(1, 1);

  while (true) {
    const int next = tile + 1;
    const bool has_next = next < tile_end;
    int nimg = img, ny0 = y0, nx0 = x0 + 32;   // the next tile's origin by stepping (tiles are walked in order)
    if (nx0 >= p.W) { nx0 = 0; ny0 += 8; if (ny0 >= p.H) { ny0 = 0; nimg += 1; } }
    if (!has_next) nimg = -1;

    // ---------------- step s: VALU phase (V of channels 32 s .. 32 s + 31, two planes of the previous tile's epilogue), MFMA phase ----------------
    auto valu_phase = [&](auto S) {
      constexpr int s = decltype(S)::value;
      halo_store(0);
      LDS_BARRIER();                                  // (1) chunk 2 s staged
      if (ph == 0) read_raw();
      drain_plane(2 * s);
      LDS_BARRIER();                                  // (2) waves 0..3 have their patches
      halo_store(1);
      // the halo of the step after this one: requested as soon as its registers are free, a VALU phase + an MFMA phase ahead of its use.
      // (s = 1: the next tile's first step; no next tile: an empty descriptor, every load returns 0)
      if (s == 0) {
        halo_load(rsXc, 1);
      } else {
        const __amdgpu_buffer_rsrc_t rsXn = img_rsrc(p.X, p.x_stride, nimg);
        halo_offsets(ny0, nx0);
        halo_load(rsXn, 0);
        rsXc = rsXn;
      }
      LDS_BARRIER();                                  // (3) chunk 2 s + 1 staged
      if (ph == 1) read_raw();
      transform();
      write_v();
      drain_plane(2 * s + 1);
      LDS_BARRIER();                                  // (4) V of the step published; everyone is done with the staged halo
    };
    SSTAMP(0);
    valu_phase(std::integral_constant<int, 0>{});
    SSTAMP(1);
    mfma_phase(std::integral_constant<int, 0>{});
    SSTAMP(2);
    valu_phase(std::integral_constant<int, 1>{});
    SSTAMP(3);
    mfma_phase(std::integral_constant<int, 1>{});
    SSTAMP(4);

    // ---------------- At M A, split with the positions ----------------
    // row pass of this wave's three xi rows: M[xi][0..5] -> R[xi][0..3] (in the accumulators of nu = 0..3)
    if (ph == 0) {
#pragma unroll
      for (int x = 0; x < 3; ++x) at6(acc[6 * x], acc[6 * x + 1], acc[6 * x + 2], acc[6 * x + 3], acc[6 * x + 4], acc[6 * x + 5]);
    } else {
      at6(acc[acc_of(1, 18)], acc[acc_of(1, 19)], acc[acc_of(1, 20)], acc[acc_of(1, 21)], acc[acc_of(1, 22)], acc[acc_of(1, 23)]);
      at6(acc[acc_of(1, 24)], acc[acc_of(1, 25)], acc[acc_of(1, 26)], acc[acc_of(1, 27)], acc[acc_of(1, 28)], acc[acc_of(1, 29)]);
      at6(acc[acc_of(1, 30)], acc[acc_of(1, 31)], acc[acc_of(1, 32)], acc[acc_of(1, 33)], acc[acc_of(1, 34)], acc[acc_of(1, 35)]);
    }
    // partial column sums Y[a][b] = sum over this wave's xi of At[a][xi] R[xi][b],  At = [1 1 1 1 1 0; 0 1 -1 2 -2 0; 0 1 1 4 4 0; 0 1 -1 8 -8 1]
    f32x4 Yp[4][4];
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      if (ph == 0) {
        const f32x4 r0 = acc[b], r1 = acc[6 + b], r2 = acc[12 + b];
        Yp[0][b] = (r0 + r1) + r2; Yp[1][b] = r1 - r2; Yp[2][b] = r1 + r2; Yp[3][b] = r1 - r2;
      } else {
        const f32x4 r3 = acc[acc_of(1, 18 + b)], r4 = acc[acc_of(1, 24 + b)], r5 = acc[acc_of(1, 30 + b)];
        const f32x4 sm = r3 + r4, df = r3 - r4;
        Yp[0][b] = sm; Yp[1][b] = 2.f * df; Yp[2][b] = 4.f * sm; Yp[3][b] = 8.f * df + r5;
      }
    }
    // pair exchange through the planes: wave ph writes its partial of the two planes the OTHER wave finishes (ph = 0 finishes a = 0, 1)
    f32x4 Wp[2][4], Kp[2][4];
    int aw, ak;
    if (ph == 0) {
      aw = 2; ak = 0;
#pragma unroll
      for (int a2 = 0; a2 < 2; ++a2)
#pragma unroll
        for (int b = 0; b < 4; ++b) { Wp[a2][b] = Yp[2 + a2][b]; Kp[a2][b] = Yp[a2][b]; }
    } else {
      aw = 0; ak = 2;
#pragma unroll
      for (int a2 = 0; a2 < 2; ++a2)
#pragma unroll
        for (int b = 0; b < 4; ++b) { Wp[a2][b] = Yp[a2][b]; Kp[a2][b] = Yp[2 + a2][b]; }
    }
#pragma unroll
    for (int a2 = 0; a2 < 2; ++a2)
#pragma unroll
      for (int b = 0; b < 4; ++b) *reinterpret_cast<f32x4*>(sX + (aw + a2) * 4096 + xw + b * 64) = Wp[a2][b];
    LDS_BARRIER();
#pragma unroll
    for (int a2 = 0; a2 < 2; ++a2) {
#pragma unroll
      for (int b = 0; b < 4; ++b) {
        float* const q = sX + (ak + a2) * 4096 + xw + b * 64;
        f32x4 v = *reinterpret_cast<const f32x4*>(q) + Kp[a2][b];
        if (p.slope != 1.f) {
#pragma unroll
          for (int k = 0; k < 4; ++k) v[k] = p.slope <= 1.f ? fmaxf(v[k], v[k] * p.slope) : (v[k] >= 0.f ? v[k] : v[k] * p.slope);
        }
        *reinterpret_cast<f32x4*>(q) = v;
      }
    }
    LDS_BARRIER();   // the tile's results are in the exchange planes (drained during the next tile's VALU phases)
    SSTAMP(5);
    rsYp = img_rsrc(p.Y, p.y_stride, img); rsEp = img_rsrc(Ep, e_stride, img); rsLp = img_rsrc(Lp, l_stride, img);
    prow0 = y0; pcol0 = x0;
    if (!has_next) {
#pragma unroll
      for (int a = 0; a < 4; ++a) drain_plane(a);
      break;
    }
    tile = next; img = nimg; y0 = ny0; x0 = nx0;
  }
#ifdef LFSR_CONV_DIAG
  if (p.dbg && threadIdx.x == 0)
    for (int k = 0; k < 16; ++k) p.dbg[blockIdx.x * 64 + k] = (float)seg[k];
#endif
}

// U = G g G^t per (n, k) from the direct pack [tap][n][k] -> the per-wave records of k_conv3x3_wino4s:
// [stage s = k/4][ns = n/16][ph]{ [q = j/4][lane = 16 (k%4) + n%16][j%4] for j < 16, then [lane][j - 16] }, j = acc_of(ph, position)
__global__ __launch_bounds__(256) void k_pack_wino4s(const float* __restrict__ direct, float* __restrict__ out) {
  const int i = blockIdx.x * 256 + threadIdx.x;   // (n, k)
  if (i >= 64 * 64) return;
  const int n = i >> 6, k = i & 63;
  double g[3][3];
#pragma unroll
  for (int t = 0; t < 9; ++t) g[t / 3][t % 3] = (double)direct[(t * 64 + n) * 64 + k];
  const double G[6][3] = {{1.0 / 4, 0.0, 0.0}, {-1.0 / 6, -1.0 / 6, -1.0 / 6}, {-1.0 / 6, 1.0 / 6, -1.0 / 6},
                          {1.0 / 24, 1.0 / 12, 1.0 / 6}, {1.0 / 24, -1.0 / 12, 1.0 / 6}, {0.0, 0.0, 1.0}};
  double tmp[6][3];
#pragma unroll
  for (int a = 0; a < 6; ++a)
#pragma unroll
    for (int c = 0; c < 3; ++c) tmp[a][c] = G[a][0] * g[0][c] + G[a][1] * g[1][c] + G[a][2] * g[2][c];
  const int s = k >> 2, kq = k & 3, nsl = n >> 4, m = n & 15, ln = kq * 16 + m;
#pragma unroll
  for (int a = 0; a < 6; ++a)
#pragma unroll
    for (int b = 0; b < 6; ++b) {
      const double u = tmp[a][0] * G[b][0] + tmp[a][1] * G[b][1] + tmp[a][2] * G[b][2];
      const int pp = a * 6 + b, ph = pp >= 18, j = acc_of(ph, pp);
      float* rec = out + ((s * 4 + nsl) * 2 + ph) * UREC;
      if (j < 16) rec[((j >> 2) * 64 + ln) * 4 + (j & 3)] = (float)u;
      else rec[1024 + ln * 2 + (j - 16)] = (float)u;
    }
}

}  // namespace

int lfsr_pack_wino4s(const float* direct_packed, float* out, hipStream_t st) {
  if (!direct_packed || !out) return LFSR_E_ARG;
  hipLaunchKernelGGL(k_pack_wino4s, dim3(16), dim3(256), 0, st, direct_packed, out);
  LFSR_CHECK_LAUNCH();
  return LFSR_OK;
}

// LFSR_E_ARG = geometry not covered (operands of 1 GiB and more): the caller falls back
int lfsr_conv3x3_wino4s_launch(const float* x, int x_stride, int x_choff, const float* w_wino4s, float* y, int y_stride, int y_choff,
                               const float* r1, int r1_stride, int r1_choff, const float* r2, int r2_stride, int r2_choff,
                               const float* mk, int mk_stride, int mk_choff, float mk_slope,
                               int n_img, int h, int w, float slope, hipStream_t st) {
  static std::atomic<bool> attr_set[64];
  static std::atomic<int> cus[64];
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return LFSR_E_ARG;
  if (!attr_set[dev]) {
    const void* fns[5] = {reinterpret_cast<const void*>(k_conv3x3_wino4s<false, false, false>), reinterpret_cast<const void*>(k_conv3x3_wino4s<false, true, false>),
                          reinterpret_cast<const void*>(k_conv3x3_wino4s<false, true, true>), reinterpret_cast<const void*>(k_conv3x3_wino4s<true, true, false>),
                          reinterpret_cast<const void*>(k_conv3x3_wino4s<true, true, true>)};
    for (const void* f : fns) {
      hipError_t e = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, SMEM_BYTES);
      if (e != hipSuccess) return LFSR_HIP_ERR(e);
    }
    int v = 0;
    cus[dev] = (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) ? v : 256;
    attr_set[dev] = true;
  }
  int ms = x_stride > y_stride ? x_stride : y_stride;
  if (r1 && r1_stride > ms) ms = r1_stride;
  if (r2 && r2_stride > ms) ms = r2_stride;
  if (mk && mk_stride > ms) ms = mk_stride;
  if ((long long)h * w * ms * 4 >= (1LL << 30)) return LFSR_E_ARG;   // per-image descriptors: one image's span < 1 GiB
  if ((long long)n_img * h * w >= (1LL << 31) / 4) return LFSR_E_ARG;
  Wino4sArgs p{};
  p.X = x; p.x_stride = x_stride; p.x_choff = x_choff;
  p.Wu = w_wino4s;
  p.Y = y; p.y_stride = y_stride; p.y_choff = y_choff;
  p.R1 = r1; p.r1_stride = r1_stride; p.r1_choff = r1_choff; p.R2 = r2; p.r2_stride = r2_stride; p.r2_choff = r2_choff;
  p.Mk = mk; p.mk_stride = mk_stride; p.mk_choff = mk_choff; p.mk_slope = mk_slope;
  p.n_img = n_img; p.H = h; p.W = w; p.slope = slope;
#ifdef LFSR_CONV_DIAG
  p.dbg = g_lfsr_diag_buf;
#endif
  const long long nt = (long long)n_img * ((h + 7) / 8) * ((w + 31) / 32);
  if (nt <= 0 || nt > 0x7fffffffLL) return LFSR_E_ARG;
  p.ntiles = (int)nt;
  const int slots = cus[dev];
  const unsigned grid = (unsigned)(nt < slots ? nt : slots);
  if (!mk && !r1 && r2) { p.R1 = r2; p.r1_stride = r2_stride; p.r1_choff = r2_choff; p.R2 = nullptr; }   // a lone residual is the first operand
  if (mk && p.R1) hipLaunchKernelGGL((k_conv3x3_wino4s<true, true, true>), dim3(grid), dim3(512), SMEM_BYTES, st, p);
  else if (mk) hipLaunchKernelGGL((k_conv3x3_wino4s<true, true, false>), dim3(grid), dim3(512), SMEM_BYTES, st, p);
  else if (p.R1 && p.R2) hipLaunchKernelGGL((k_conv3x3_wino4s<false, true, true>), dim3(grid), dim3(512), SMEM_BYTES, st, p);
  else if (p.R1) hipLaunchKernelGGL((k_conv3x3_wino4s<false, true, false>), dim3(grid), dim3(512), SMEM_BYTES, st, p);
  else hipLaunchKernelGGL((k_conv3x3_wino4s<false, false, false>), dim3(grid), dim3(512), SMEM_BYTES, st, p);
  LFSR_CHECK_LAUNCH();
  return LFSR_OK;
}
