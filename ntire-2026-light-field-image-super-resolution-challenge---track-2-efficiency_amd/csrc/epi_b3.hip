// EPIConv branch (model/SR/DistgSSR.py:91-97 and its transposed application :108) at angRes 5 on the bf16 MFMA pipe with fp32 operands carried EXACTLY as
// three bf16 terms (default at A = 5; LFSR_EPI=wino keeps epi_fused.hip's fp32-MFMA Winograd F(2,5) kernel, LFSR_EPI=direct its direct fp32 form):
//   t = lrelu(conv 1x25, stride (1,5), pad 10, 64->32);  y = lrelu(1x1 32->160);  PixelShuffle1D(5)       -- both passes (H and V) in one launch.
// Arithmetic as in rowgemm_b3.hip: x = x0 + x1 + x2, w = w0 + w1 + w2 by truncation (the sums exact), the six products of order <= 2 as
// v_mfma_f32_16x16x32_bf16 with fp32 accumulation -- closer to fp64 than an fp32 FMA chain (tests: test_epiconv*, test_b3_products_against_fp64).
//
// Formulation.  In VCL the 1x25 conv is a 5-tap 1-D conv along the EPI line over 5 views x 64 channels.  With N = SOURCE positions instead of output
// positions the five taps share their B operands:  Z_dx[n][s] = sum_{v', c} W[5 dx + v'][n][c] X[v'][s][c]  for the 32 sources of a line (the two
// padding positions on either side are zero and need no columns), then  t[x][n] = sum_dx Z_dx[n][x + dx - 2]  -- one shifted add of the five
// accumulator tiles; the position index is the lane index within a row of 16, so the shift is a DPP row shift (+ the wrap from the neighbouring tile): no LDS.  So every input value is loaded ONCE (straight from global memory in B-operand order: lane =
// source position, eight consecutive channels per k-group), split ONCE (5.5 VALU per element, in the shadow of the MFMAs) and feeds 5 taps x 32 outputs.
// One wave = one EPI line = 80 accumulator registers; a persistent 512-thread block = 8 lines per group.  The weights arrive pre-split from the pack
// (lfsr_pack_epi_b3) and are staged per VIEW in LDS (two K steps, 60 KB), double-buffered behind ONE barrier per view (straight copy, fetched half by half);
// stage 2 (1x1 32 -> 160) runs the same way from the LDS-resident planes of its weights -- its B operand is the lane's own t registers as they stand, because
// the pack stores W2's columns in the order a lane holds them -- and its result goes out as 16-B stores per lane (chunk = destination view = PixelShuffle1D).
#include <stdlib.h>

#include "lfsr_internal.h"

#ifndef EB_VAR
#define EB_VAR 0   // bisecting builds (correct results): 1 select-form LeakyReLU, 2 pointer stores, 4 both waves of a SIMD split at the same point
#endif
#ifndef EB_ABL
#define EB_ABL 0   // diagnostic timing builds (WRONG results; tools/build_abl.sh): 1 no split of the next step's input, 2 one A-operand read per step, 4 no weight staging,
                   // 8 no input loads, 16 no tail (tap sum, stage 2, stores), 32 no barriers
#endif

namespace {

typedef float f32x4e __attribute__((ext_vector_type(4)));
typedef unsigned u32x4e __attribute__((ext_vector_type(4)));

constexpr int EB_LINES = 8;
constexpr int EB_STAGE_SLOTS = 5 * 3 * 4 * 32;              // 16-B slots of one (view, K step): [dx][plane][k-group][n]
constexpr int EB_STAGE_BYTES = EB_STAGE_SLOTS * 16;         // 30720
constexpr int EB_W2_SLOTS = 3 * 4 * 160;                    // [plane][k-group][n']
constexpr int EB_W2_BYTES = EB_W2_SLOTS * 16;               // 30720
constexpr int EB_SMEM = 4 * EB_STAGE_BYTES + EB_W2_BYTES;   // two views of stage-1 planes (two K steps each) + the stage-2 planes: 153600

struct EpiB3Args {
  const float* X; int x_stride; int x_choff; int x_bytes;
  const uint4* W1p;     // [view 5][K step 2][EB_STAGE_SLOTS]  (lfsr_pack_epi_b3)
  const uint4* W2p;     // [EB_W2_SLOTS]
  float* Y; int y_stride; int choffH; int choffV; int y_bytes;
  float* TH; float* TV;   // optional (lines * len, 32): post-LeakyReLU stage-1 activations saved for the backward
  int B, H, W;
  int tilesH, tilesV;     // groups per pass
  int tpiH, tpiV;         // > 0: groups ordered item by item (an item's horizontal groups, then its vertical ones)
  float slope;
};

__device__ __forceinline__ unsigned eb_hi_pair(unsigned hi_src, unsigned lo_src) { return __builtin_amdgcn_perm(hi_src, lo_src, 0x07060302u); }
__device__ __forceinline__ float eb_residual(float a) { return a - __uint_as_float(__float_as_uint(a) & 0xffff0000u); }   // exact

// eight consecutive floats -> their three bf16 planes in MFMA operand order
__device__ __forceinline__ void eb_split8(const f32x4e lo, const f32x4e hi, u32x4e& p0, u32x4e& p1, u32x4e& p2) {
  const float a[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
#pragma unroll
  for (int j = 0; j < 4; ++j) { unsigned t0, t1, t2; lfsr_split_pair(a[2 * j], a[2 * j + 1], t0, t1, t2); p0[j] = t0; p1[j] = t1; p2[j] = t2; }
}

// asm MFMA with the accumulator tied (rowgemm_b3.hip, b3_mfma: why not the builtin)
__device__ __forceinline__ void eb_mfma(f32x4e& c, const u32x4e a, const u32x4e b) {
  asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b));
}

// six products of one (A planes, B planes) pair into TWO accumulators alternately (two independent chains), smallest terms first
__device__ __forceinline__ void eb_six2(f32x4e& c0, f32x4e& c1, const u32x4e a0, const u32x4e a1, const u32x4e a2,
                                        const u32x4e x00, const u32x4e x01, const u32x4e x02, const u32x4e x10, const u32x4e x11, const u32x4e x12) {
  eb_mfma(c0, a2, x00); eb_mfma(c1, a2, x10);
  eb_mfma(c0, a0, x02); eb_mfma(c1, a0, x12);
  eb_mfma(c0, a1, x01); eb_mfma(c1, a1, x11);
  eb_mfma(c0, a1, x00); eb_mfma(c1, a1, x10);
  eb_mfma(c0, a0, x01); eb_mfma(c1, a0, x11);
  eb_mfma(c0, a0, x00); eb_mfma(c1, a0, x10);
}

// lane i of a row of 16 reads lane i + N (row_shl) / i - N (row_shr) of the same row; lanes that would read outside the row get 0
template <int CTRL>
__device__ __forceinline__ float eb_dpp0(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}
// z[x + D] for the two position tiles of a line (x = 16 nt + l15; sources outside [0, 32) contribute nothing): a shift inside the tile plus the lanes that wrap
// in from the neighbouring tile
template <int D>
__device__ __forceinline__ void eb_shift_add(f32x4e (&t)[2], const f32x4e (&z)[2]) {
  constexpr int SAME = D > 0 ? 0x100 + D : 0x110 - D;                 // row_shl:D / row_shr:-D
  constexpr int WRAP = D > 0 ? 0x110 + (16 - D) : 0x100 + (16 + D);   // row_shr:(16 - D) / row_shl:(16 + D)
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    t[0][r] += eb_dpp0<SAME>(z[0][r]);
    t[1][r] += eb_dpp0<SAME>(z[1][r]);
    if (D > 0) t[0][r] += eb_dpp0<WRAP>(z[1][r]);                     // positions 16 - D .. 15 of tile 0 read positions 0 .. D - 1 of tile 1
    else t[1][r] += eb_dpp0<WRAP>(z[0][r]);                           // positions 0 .. -D - 1 of tile 1 read positions 16 + D .. 15 of tile 0
  }
}

__global__ __launch_bounds__(512) void k_epi_b3(EpiB3Args p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  unsigned char* const sW = smem_raw;                                   // [2 views][2 K steps][EB_STAGE_BYTES]
  unsigned char* const sW2 = smem_raw + 4 * EB_STAGE_BYTES;             // [EB_W2_BYTES]
  constexpr int A = 5;
  constexpr int EOOB = (int)0x80000000u;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l15 = lane & 15, g = lane >> 4;
  const int phase = __builtin_amdgcn_readfirstlane(wave >> 2);     // the two waves of a SIMD: 0 / 1
  const int HW = p.H * p.W;
  const int ngroups = p.tilesH + p.tilesV;
  // consecutive workgroups go to consecutive XCDs: the blocks of one XCD walk a contiguous range of groups (both passes of an item share an L2)
  const int nb = (int)gridDim.x;
  const int vb = (nb & 7) ? (int)blockIdx.x : ((int)blockIdx.x & 7) * (nb >> 3) + ((int)blockIdx.x >> 3);

  const __amdgpu_buffer_rsrc_t rsX = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.X), 0, p.x_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsY = __builtin_amdgcn_make_buffer_rsrc(p.Y, 0, p.y_bytes, 0x00020000);

  // ---- group geometry (block-uniform) and this wave's line ----
  struct Line { int vert, base, len, vstride, pstride, choff; long long rowbase; int rowstep; };
  // Work units of a block: whole groups of 8 lines in its full rounds (group vb + i nb); the groups LEFT OVER after the last full round (B = 8: 320 groups on 256 CUs)
  // are cut into f = 2 or 4 sub-groups of 8 / f lines while they still fit one round, so the last round costs a wave its one line with a SIMD to itself instead of
  // a whole group time on a quarter of the chip.  Per-line arithmetic is untouched (same bits).
  const int nfr = ngroups / nb, nfull = nfr * nb, rem = ngroups - nfull;
  const int fsub = rem == 0 ? 1 : (rem * 4 <= nb ? 4 : rem * 2 <= nb ? 2 : 1);
  const int nunits = nfr + (vb < rem * fsub ? 1 : 0);                 // (block-uniform)
  auto line_of = [&](int it) -> Line {
    Line L;
    int grp, l0 = 0, nl = EB_LINES;
    if (it < nfr) grp = vb + it * nb;
    else if (it < nunits) { grp = nfull + vb / fsub; nl = EB_LINES / fsub; l0 = (vb % fsub) * nl; }
    else grp = ngroups;                                                 // past the end: an absent line, every access out of range
    int tile;
    if (p.tpiH > 0) {
      const int per = p.tpiH + p.tpiV, item = grp / per, r = grp - item * per;
      L.vert = r >= p.tpiH;
      tile = L.vert ? item * p.tpiV + (r - p.tpiH) : item * p.tpiH + r;
    } else {
      L.vert = grp >= p.tilesH;
      tile = L.vert ? grp - p.tilesH : grp;
    }
    L.len = L.vert ? p.H : p.W;
    const int across = L.vert ? p.W : p.H;
    const int nlines = p.B * A * across;
    L.vstride = L.vert ? A * HW : HW;
    L.pstride = L.vert ? p.W : 1;
    L.choff = L.vert ? p.choffV : p.choffH;
    const int ln = tile * EB_LINES + l0 + wave;
    L.base = -1; L.rowbase = 0; L.rowstep = 0;
    if (grp < ngroups && wave < nl && ln < nlines) {
      const int q = ln / across, o = ln - q * across;     // horizontal: q = b*A+u, o = y;  vertical: q = b*A+v, o = x
      if (!L.vert) L.base = q * A * HW + o * p.W;
      else { const int b = q / A, v = q - b * A; L.base = (b * A * A + v) * HW + o; }
      // rows of the saved stage-1 matrix are ordered like the gather-GEMM's: (b*A+u, y, x) / (b*A+v, y, x)
      L.rowbase = L.vert ? (long long)q * p.H * p.W + o : ((long long)q * p.H + o) * p.W;
      L.rowstep = L.vert ? p.W : 1;
    }
    return L;
  };

  // raw input, two steps deep ([source tile nt][lo / hi four floats]) and its planes, double-buffered: step j's MFMAs read planes P(j & 1) while the raw values of
  // step j + 1 are split into the other set IN THEIR SHADOW (a pair of values pinned behind each MFMA group) and the loads of step j + 2 fly
  f32x4e rawA[2][2], rawB[2][2];
  u32x4e pA0[2], pA1[2], pA2[2], pB0[2], pB1[2], pB2[2];
  auto load_x = [&](const Line& L, int j, f32x4e (&xr)[2][2]) {   // step j = 2 v' + ks of the line
    const int vv = j >> 1, ks = j & 1;
    const int soff = (vv * L.vstride * p.x_stride + 32 * ks) * 4;
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
      const int s = 16 * nt + l15;
      const int off = (L.base >= 0 && s < L.len) ? ((L.base + s * L.pstride) * p.x_stride + p.x_choff + 8 * g) * 4 : EOOB;
      xr[nt][0] = __builtin_bit_cast(f32x4e, __builtin_amdgcn_raw_buffer_load_b128(rsX, off, soff, 0));
      xr[nt][1] = __builtin_bit_cast(f32x4e, __builtin_amdgcn_raw_buffer_load_b128(rsX, off == EOOB ? EOOB : off + 16, soff, 0));
    }
  };
  // weights: one LDS buffer per VIEW (both K steps, 60 KB), double-buffered: ONE barrier per view.  The next view's image is fetched half by half (its K step ks during
  // this view's step ks) and stored into the other buffer at the end of the step -- that buffer was read last during the previous view, whose barrier every wave has passed
  uint4 wr[4];
  auto load_w = [&](int half_idx) {      // half_idx = 2 view + ks
    const uint4* src = p.W1p + half_idx * EB_STAGE_SLOTS;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int idx = tid + 512 * i;
      wr[i] = idx < EB_STAGE_SLOTS ? src[idx] : make_uint4(0u, 0u, 0u, 0u);
    }
  };
  auto store_w = [&](int buf, int ks) {
    uint4* dst = reinterpret_cast<uint4*>(sW + (buf * 2 + ks) * EB_STAGE_BYTES);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int idx = tid + 512 * i;
      if (idx < EB_STAGE_SLOTS) dst[idx] = wr[i];
    }
  };

  int it = 0;
  if (nunits == 0) return;              // (block-uniform: before any barrier)
  Line L = line_of(0);
  load_x(L, 0, rawA);
  load_x(L, 1, rawB);
  load_w(0);
  {   // stage-2 weight planes: once per block
    uint4 w2r[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) { const int idx = tid + 512 * i; w2r[i] = idx < EB_W2_SLOTS ? p.W2p[idx] : make_uint4(0u, 0u, 0u, 0u); }
#pragma unroll
    for (int i = 0; i < 4; ++i) { const int idx = tid + 512 * i; if (idx < EB_W2_SLOTS) reinterpret_cast<uint4*>(sW2)[idx] = w2r[i]; }
  }
  store_w(0, 0);
  load_w(1);
  store_w(0, 1);
#pragma unroll
  for (int nt = 0; nt < 2; ++nt) eb_split8(rawA[nt][0], rawA[nt][1], pA0[nt], pA1[nt], pA2[nt]);
  __syncthreads();
  int cur = 0;                                          // buffer of the current view
  const int aoff = (g * 32 + l15) * 16;                 // this lane's A-operand slot inside a (dx, plane) block: k-group g, weight row l15 (+ 16 mt)
  const unsigned char* const w2b = sW2 + (g * 160 + l15) * 16;

  for (;;) {
    const bool more = it + 1 < nunits;  // (block-uniform)
    const Line Ln = line_of(it + 1);    // (past the end: an absent line, every access out of range)
    f32x4e acc[5][2][2];
#pragma unroll
    for (int dx = 0; dx < 5; ++dx)
#pragma unroll
      for (int mt = 0; mt < 2; ++mt) { acc[dx][mt][0] = f32x4e{0.f, 0.f, 0.f, 0.f}; acc[dx][mt][1] = acc[dx][mt][0]; }
#pragma unroll
    for (int dx = 0; dx < 5; ++dx) asm volatile("s_nop 1" : "+v"(acc[dx][0][0]), "+v"(acc[dx][0][1]), "+v"(acc[dx][1][0]), "+v"(acc[dx][1][1]));

    // one step: MFMAs of step j on the planes (u0, u1, u2); raw values rs (step j + 1) split into (d0, d1, d2); loads of step j + 2 into rd
    auto step = [&](int j, u32x4e (&u0)[2], u32x4e (&u1)[2], u32x4e (&u2)[2], f32x4e (&rs)[2][2], u32x4e (&d0)[2], u32x4e (&d1)[2], u32x4e (&d2)[2],
                    f32x4e (&rd)[2][2]) {
      const int vv = j >> 1, ks = j & 1;
      if (!(EB_ABL & 8)) {
        if (j + 2 < 10) load_x(L, j + 2, rd);
        else load_x(Ln, j + 2 - 10, rd);
      }
      if (!(EB_ABL & 4)) load_w(2 * (vv + 1 < A ? vv + 1 : 0) + ks);
      asm volatile("s_nop 4" : "+v"(u0[0]), "+v"(u1[0]), "+v"(u2[0]), "+v"(u0[1]), "+v"(u1[1]), "+v"(u2[1]));
      const unsigned char* wb = sW + (cur * 2 + ks) * EB_STAGE_BYTES + aoff;
      // A operands (the weights' planes of tap dx, row tile mt) one group ahead of their MFMAs
      u32x4e af[2][3];
      auto load_a = [&](int c, u32x4e (&a)[3]) {
        const int dx = c >> 1, mt = c & 1;
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) a[pl] = *reinterpret_cast<const u32x4e*>(wb + ((dx * 3 + pl) * 128 + 16 * mt) * 16);
      };
      load_a(0, af[0]);
#pragma unroll
      for (int c = 0; c < 10; ++c) {
        const int dx = c >> 1, mt = c & 1;
        if (c + 1 < 10 && !(EB_ABL & 2)) load_a(c + 1, af[(c + 1) & 1]);
        constexpr int ABL2 = (EB_ABL & 2) ? 0 : 1;
        eb_six2(acc[dx][mt][0], acc[dx][mt][1], af[(c & 1) * ABL2][0], af[(c & 1) * ABL2][1], af[(c & 1) * ABL2][2], u0[0], u1[0], u2[0], u0[1], u1[1], u2[1]);
        // The next step's split (88 VALU) as ONE burst, at a point of the step that differs between the two waves of a SIMD (wave w and w + 4 share SIMD w & 3):
        // beside a streaming MFMA wave a VALU instruction issues only every ~13-20 cycles, and interleaved with the wave's OWN MFMAs it holds those up as well
        // (ablation, profiles/r03_logs/c8_epi_b3_ablations.log: the split sprinkled between the MFMA groups cost 52 of 234 us).  Phase-shifted, one wave's burst
        // runs under the other wave's MFMA stream, which keeps the matrix pipe busy alone.  Volatile asm statements keep their order: "defining" the inputs and
        // "using" the outputs pins the burst behind MFMA group `c`.
        if (!(EB_ABL & 1) && c == ((phase && !(EB_VAR & 4)) ? 5 : 0)) {
          asm volatile("" : "+v"(rs[0][0]), "+v"(rs[0][1]), "+v"(rs[1][0]), "+v"(rs[1][1]));
#pragma unroll
          for (int nt = 0; nt < 2; ++nt) eb_split8(rs[nt][0], rs[nt][1], d0[nt], d1[nt], d2[nt]);
          asm volatile("" : "+v"(d0[0]), "+v"(d1[0]), "+v"(d2[0]), "+v"(d0[1]), "+v"(d1[1]), "+v"(d2[1]));
        }
      }
      if (!(EB_ABL & 4)) store_w(cur ^ 1, ks);
      if (ks) { if (!(EB_ABL & 32)) __syncthreads(); cur ^= 1; }
    };
#pragma unroll 1
    for (int jj = 0; jj < 10; jj += 2) {
      step(jj, pA0, pA1, pA2, rawB, pB0, pB1, pB2, rawA);
      step(jj + 1, pB0, pB1, pB2, rawA, pA0, pA1, pA2, rawB);
    }

    // ---- t[x][n] = sum_dx Z_dx[n][x + dx - 2]: the centre tap as it stands, the others shifted along the position index = the lane index within a row of 16:
    //      DPP row shifts inside a tile plus the lanes that wrap in from the neighbouring tile (no LDS round trip) ----
#pragma unroll
    for (int dx = 0; dx < 5; ++dx)
      asm volatile("s_nop 15\n\ts_nop 15" : "+v"(acc[dx][0][0]), "+v"(acc[dx][0][1]), "+v"(acc[dx][1][0]), "+v"(acc[dx][1][1]));
    if (EB_ABL & 16) {                  // (keep the accumulators live: one store of their sum)
      f32x4e sacc = acc[0][0][0];
#pragma unroll
      for (int dx = 0; dx < 5; ++dx) sacc += acc[dx][0][1] + acc[dx][1][0] + acc[dx][1][1] + acc[dx][0][0];
      if (L.base >= 0 && l15 < L.len) *reinterpret_cast<f32x4e*>(p.Y + (long long)(L.base + l15 * L.pstride) * p.y_stride + L.choff + 4 * g) = sacc;
      if (!more) break;
      ++it; L = Ln;
      continue;
    }
    f32x4e tv[2][2];                    // [mt][nt]: channels 16 mt + 4 g .. + 3 of position 16 nt + l15
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
      tv[mt][0] = acc[2][mt][0]; tv[mt][1] = acc[2][mt][1];
      eb_shift_add<-2>(tv[mt], acc[0][mt]);
      eb_shift_add<-1>(tv[mt], acc[1][mt]);
      eb_shift_add<1>(tv[mt], acc[3][mt]);
      eb_shift_add<2>(tv[mt], acc[4][mt]);
    }
    {
      float* tsave = L.vert ? p.TV : p.TH;
      const bool save = tsave && L.base >= 0;
#pragma unroll
      for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
#pragma unroll
          for (int r = 0; r < 4; ++r) { const float v = tv[mt][nt][r]; tv[mt][nt][r] = (EB_VAR & 1) ? (v >= 0.f ? v : v * p.slope) : fmaxf(v, v * p.slope); }     // LeakyReLU, 0 <= slope <= 1 (checked by the launcher)
          const int pos = 16 * nt + l15;
          if (save && pos < L.len) *reinterpret_cast<f32x4e*>(tsave + (L.rowbase + (long long)pos * L.rowstep) * 32 + 16 * mt + 4 * g) = tv[mt][nt];
        }
      if (save) asm volatile("s_nop 1" : "+v"(tv[0][0]), "+v"(tv[0][1]), "+v"(tv[1][0]), "+v"(tv[1][1]));     // (same store-data hazard as below)
    }
    // ---- stage 2: y[n'][x] = lrelu(sum_k W2[n'][k] t[x][k]), K = 32 = one K step.  The lane (position, g) already HOLDS eight of its position's channels
    //      (16 mt + 4 g + r): they are its B operand as they stand once the pack stores W2's columns in that order (k slot 8 g + 4 mt + r <-> channel 16 mt + 4 g + r) ----
    u32x4e t0[2], t1[2], t2[2];
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) eb_split8(tv[0][nt], tv[1][nt], t0[nt], t1[nt], t2[nt]);
    asm volatile("s_nop 4" : "+v"(t0[0]), "+v"(t1[0]), "+v"(t2[0]), "+v"(t0[1]), "+v"(t1[1]), "+v"(t2[1]));
    int yoff[2];                        // byte offset of this lane's pixel (chunk 0) in Y; out of range for an absent pixel (the store is dropped)
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
      const int pos = 16 * nt + l15;
      yoff[nt] = (L.base >= 0 && pos < L.len) ? ((L.base + pos * L.pstride) * p.y_stride + L.choff + 4 * g) * 4 : EOOB;
    }
    const int ychunk = L.vstride * p.y_stride * 4;      // bytes between destination views (wave-uniform)
    u32x4e w2f[2][2][3];                // [parity of mp][h][plane]
    auto load_w2 = [&](int mp, u32x4e (&a)[2][3]) {
#pragma unroll
      for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) a[h][pl] = *reinterpret_cast<const u32x4e*>(w2b + ((pl * 4) * 160 + 16 * (2 * mp + h)) * 16);
    };
    load_w2(0, w2f[0]);
#pragma unroll
    for (int mp = 0; mp < 5; ++mp) {    // destination view = chunk mp: its 32 channels = weight-row tiles 2 mp, 2 mp + 1
      if (mp + 1 < 5) load_w2(mp + 1, w2f[(mp + 1) & 1]);
      f32x4e o[2][2];
#pragma unroll
      for (int h = 0; h < 2; ++h) { o[h][0] = f32x4e{0.f, 0.f, 0.f, 0.f}; o[h][1] = o[h][0]; }
      asm volatile("s_nop 1" : "+v"(o[0][0]), "+v"(o[0][1]), "+v"(o[1][0]), "+v"(o[1][1]));
#pragma unroll
      for (int h = 0; h < 2; ++h)
        eb_six2(o[h][0], o[h][1], w2f[mp & 1][h][0], w2f[mp & 1][h][1], w2f[mp & 1][h][2], t0[0], t1[0], t2[0], t0[1], t1[1], t2[1]);
      asm volatile("s_nop 15\n\ts_nop 15" : "+v"(o[0][0]), "+v"(o[0][1]), "+v"(o[1][0]), "+v"(o[1][1]));
#pragma unroll
      for (int nt = 0; nt < 2; ++nt)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          f32x4e v = o[h][nt];
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] = (EB_VAR & 1) ? (v[r] >= 0.f ? v[r] : v[r] * p.slope) : fmaxf(v[r], v[r] * p.slope);
          if (EB_VAR & 2) { if (yoff[nt] != EOOB) *reinterpret_cast<f32x4e*>(reinterpret_cast<char*>(p.Y) + (long long)yoff[nt] + (long long)mp * ychunk + 64 * h) = v; }
          else __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4e, v), rsY, yoff[nt], __builtin_amdgcn_readfirstlane(mp * ychunk + 64 * h), 0);
          // A 16-B store (SGPR offset) followed by a write of its data registers needs a wait state; the compiler pads it for instructions it can see, but the next
          // writer may be one of the asm MFMAs (their accumulators are allocated over dead registers).  Found the hard way: ~10 wrong values per 3 M.  Holding the
          // data registers live across a one-cycle wait closes it.
          asm volatile("s_nop 1" : "+v"(v));
        }
    }
    if (!more) break;
    ++it;
    L = Ln;
  }
}

// ---- pack: the three bf16 planes of both stages' weights in exactly the order the kernel stages / reads them ----
__device__ __forceinline__ void eb_planes8(const float* src, uint4& p0, uint4& p1, uint4& p2) {
  unsigned h0[8], h1[8], h2[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const float a = src[j];
    const float r = eb_residual(a), q = eb_residual(r);
    h0[j] = __float_as_uint(a) >> 16; h1[j] = __float_as_uint(r) >> 16; h2[j] = __float_as_uint(q) >> 16;
  }
  p0 = make_uint4(h0[0] | (h0[1] << 16), h0[2] | (h0[3] << 16), h0[4] | (h0[5] << 16), h0[6] | (h0[7] << 16));
  p1 = make_uint4(h1[0] | (h1[1] << 16), h1[2] | (h1[3] << 16), h1[4] | (h1[5] << 16), h1[6] | (h1[7] << 16));
  p2 = make_uint4(h2[0] | (h2[1] << 16), h2[2] | (h2[3] << 16), h2[4] | (h2[5] << 16), h2[6] | (h2[7] << 16));
}

// kind 0: w1 direct pack [tap 5 dx + v'][n 32][c 64] -> [v'][ks][dx][plane][g][n] slots;  kind 1: w2 direct pack [n' 160][k 32] -> [plane][g][n'] slots
__device__ __forceinline__ void eb_pack_one(const float* __restrict__ src, uint4* __restrict__ dst, int kind, int i) {
  uint4 p0, p1, p2;
  if (kind == 0) {
    if (i >= 5 * 2 * 5 * 4 * 32) return;             // (v', ks, dx, g, n)
    const int n = i & 31, g = (i >> 5) & 3, dx = (i >> 7) % 5, r = (i >> 7) / 5, ks = r & 1, v = r >> 1;
    eb_planes8(src + ((5 * dx + v) * 32 + n) * 64 + 32 * ks + 8 * g, p0, p1, p2);
    uint4* o = dst + (v * 2 + ks) * EB_STAGE_SLOTS + (dx * 3 * 4 + g) * 32 + n;
    o[0] = p0; o[4 * 32] = p1; o[8 * 32] = p2;
  } else {
    if (i >= 4 * 160) return;                         // (g, n')
    const int n = i % 160, g = i / 160;
    float kk[8];                                      // k slot 8 g + j <-> hidden channel 16 (j >> 2) + 4 g + (j & 3): the order in which a lane of k_epi_b3 holds its t values
#pragma unroll
    for (int jx = 0; jx < 8; ++jx) kk[jx] = src[n * 32 + 16 * (jx >> 2) + 4 * g + (jx & 3)];
    eb_planes8(kk, p0, p1, p2);
    uint4* o = dst + g * 160 + n;
    o[0] = p0; o[4 * 160] = p1; o[8 * 160] = p2;
  }
}

__global__ __launch_bounds__(256) void k_pack_epi_b3(const float* __restrict__ src, uint4* __restrict__ dst, int kind) {
  eb_pack_one(src, dst, kind, blockIdx.x * 256 + threadIdx.x);
}

__global__ __launch_bounds__(256) void k_pack_epi_b3_batch(const LfsrPackDesc* __restrict__ tab) {
  const LfsrPackDesc d = tab[blockIdx.y];
  eb_pack_one(d.src, reinterpret_cast<uint4*>(d.dst), d.kind, blockIdx.x * 256 + threadIdx.x);
}

}  // namespace

// kind 0: EPIConv.0 (direct pack -> LFSR_EPI_B3_W1_FLOATS), kind 1: EPIConv.2 (direct pack -> LFSR_EPI_B3_W2_FLOATS)
int lfsr_pack_epi_b3(const float* direct_packed, float* out, int kind, hipStream_t st) {
  if (!direct_packed || !out || (kind != 0 && kind != 1) || ((uintptr_t)out & 15)) return LFSR_E_ARG;
  hipLaunchKernelGGL(k_pack_epi_b3, dim3(kind == 0 ? 25 : 3), dim3(256), 0, st, direct_packed, reinterpret_cast<uint4*>(out), kind);
  LFSR_CHECK_LAUNCH();
  return LFSR_OK;
}

int lfsr_pack_epi_b3_batch(const LfsrPackDesc* table_dev, int n, hipStream_t st) {
  if (!table_dev || n <= 0) return n == 0 ? LFSR_OK : LFSR_E_ARG;
  hipLaunchKernelGGL(k_pack_epi_b3_batch, dim3(25, (unsigned)n), dim3(256), 0, st, table_dev);
  LFSR_CHECK_LAUNCH();
  return LFSR_OK;
}

// LFSR_E_ARG = not covered (the caller keeps epi_fused.hip's fp32-MFMA kernels)
int lfsr_epi_b3_launch(const float* x, int x_stride, int x_choff, const float* w1_planes, const float* w2_planes, float* y, int y_stride,
                       int choffH, int choffV, float* t_h, float* t_v, int B, int A, int h, int w, int which, float slope, hipStream_t st) {
  if (A != 5 || h > 32 || w > 32 || h <= 0 || w <= 0 || B <= 0) return LFSR_E_ARG;
  if ((x_stride | x_choff | y_stride | choffH | choffV) & 3) return LFSR_E_ARG;
  if (((uintptr_t)x | (uintptr_t)y | (uintptr_t)w1_planes | (uintptr_t)w2_planes | (uintptr_t)t_h | (uintptr_t)t_v) & 15) return LFSR_E_ARG;
  if ((long long)B * A * A * h * w * x_stride * 4 >= (1LL << 31) || (long long)B * A * A * h * w * y_stride * 4 >= (1LL << 31)) return LFSR_E_ARG;   // 32-bit byte offsets
  if (!(slope >= 0.f && slope <= 1.f)) return LFSR_E_ARG;      // LeakyReLU as max(v, slope v)
  static std::atomic<bool> attr_set[64];
  static std::atomic<int> cus[64];
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return LFSR_E_ARG;
  if (!attr_set[dev]) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_epi_b3), hipFuncAttributeMaxDynamicSharedMemorySize, EB_SMEM);
    if (e != hipSuccess) return LFSR_HIP_ERR(e);
    int v = 0;
    cus[dev] = (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) ? v : 256;
    attr_set[dev] = true;
  }
  EpiB3Args p{};
  p.X = x; p.x_stride = x_stride; p.x_choff = x_choff; p.x_bytes = (int)((long long)B * A * A * h * w * x_stride * 4);
  p.W1p = reinterpret_cast<const uint4*>(w1_planes); p.W2p = reinterpret_cast<const uint4*>(w2_planes);
  p.Y = y; p.y_stride = y_stride; p.choffH = choffH; p.choffV = choffV; p.TH = t_h; p.TV = t_v;
  p.y_bytes = (int)((long long)B * A * A * h * w * y_stride * 4);
  p.B = B; p.H = h; p.W = w; p.slope = slope;
  p.tilesH = (which & 1) ? (B * A * h + EB_LINES - 1) / EB_LINES : 0;
  p.tilesV = (which & 2) ? (B * A * w + EB_LINES - 1) / EB_LINES : 0;
  const int ngroups = p.tilesH + p.tilesV;
  if (ngroups <= 0) return LFSR_E_ARG;
  if (which == 3 && (A * h) % EB_LINES == 0 && (A * w) % EB_LINES == 0) { p.tpiH = A * h / EB_LINES; p.tpiV = A * w / EB_LINES; }
  int grid = cus[dev];
  if (grid > ngroups) grid = ngroups;
  hipLaunchKernelGGL(k_epi_b3, dim3((unsigned)grid), dim3(512), EB_SMEM, st, p);
  LFSR_CHECK_LAUNCH();
  return LFSR_OK;
}
