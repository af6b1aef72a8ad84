// Per-view 3x3 conv, zero pad 1, 64 -> 64 channels, VCL layout -- Winograd F(4x4, 3x3) with the 36 position-GEMMs on the bf16 MFMA pipe,
// fp32 operands carried EXACTLY as three bf16 terms each ("split" form; LFSR_CONV3X3=wino4b).
// Reference: the MacPI convs "k3, dilation A, padding A" of model/SR/DistgSSR.py:22,47,64,79-83,101 (per-view 3x3 in VCL),
// EPIT.py:24-32,136-142 / LFT.py:36-46 (Conv3d(1,3,3)).
//
//   Y = At [ sum_c (G g G^t) . (Bt d B) ] A     exactly as conv3x3_wino4.hip (same transforms, same tile walk, same epilogue);
//
// what differs is how a product u * v of the position-GEMMs (u = transformed weight, v = transformed input, both fp32) is formed:
//   v = v0 + v1 + v2,  u = u0 + u1 + u2   with every term a bf16 number (8 significant bits each: 24 bits = the whole fp32 significand; the
//   input is split by truncation, which is exact: v - v0 and (v - v0) - v1 are representable; the weights at pack time by round-to-nearest)
//   u v ~= u0 v0 + u1 v1 + u0 v1 + u1 v0 + u0 v2 + u2 v0          (the three dropped products are <= 2^-24 |u v|: below the rounding of an fp32 product)
// accumulated in fp32 by the matrix pipe: v_mfma_f32_16x16x32_bf16 with K = 16 channels x the term pair (v0, v1) . (u0, u1), and four
// v_mfma_f32_16x16x16_bf16 for the cross terms.  Measured against fp64 on random data the six-term sum is 8x closer than an fp32 dot product
// (3e-8 vs 2.7e-7 relative, tests/test_oracle_vs_golden.py::test_split_bf16_products) -- this is an fp32 kernel whose multiplier is the bf16 array.
//
// Why: beside ANY streaming MFMA wave (fp32 or bf16, tools/micro/micro_bf16.hip) the partner wave's VALU instructions issue every 12-15 cycles
// instead of 4-7, so the fp32 form (18.4 k MFMA cycles per tile) leaves no room for the producers' transform; the split form needs 7-11 k.
//
// Block = 8 waves on one CU, persistent over a contiguous range of 8 x 32-pixel tiles (16 Winograd tiles), specialised as in conv3x3_wino4.hip:
//   waves 0..3  CONSUMERS: wave ns owns output channels 16 ns .. 16 ns + 15 of all 16 Winograd tiles, 36 accumulators of 4 registers;
//               U terms stream from L2 (24 B per lane and (position, 16-channel chunk): (u0, u1) as 16 B + u2 as 8 B), V terms from LDS.
//   waves 4..7  PRODUCERS: thread = (Winograd tile, channel quad) of a 16-channel chunk, wave = one quarter of the 36 positions
//               (xi half x nu half): 5 x 5 raw pixels x 4 channels from the staged halo (ds_read_b128), Bt d B for its 9 positions,
//               the three-term split (22 VALU per position and quad) and the V records; plus all global traffic (halo, epilogue operands, stores).
// V is single-buffered in halves (positions 0..17 | 18..35): the consumers work on one half while the producers overwrite the other with the
// next chunk's records, which they hold in registers until the half is free.  Two barriers per chunk + one per tile for the exchange planes.
#include <stdlib.h>
#include <type_traits>
#include <utility>

#include "lfsr_internal.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));

#ifndef W4B_OPS
#define W4B_OPS 31     // (debug builds) which of the five product groups run: 1 (u0,u1).(v0,v1), 2 u0 v1, 4 u1 v0, 8 u0 v2, 16 u2 v0
#endif
#ifndef W4B_ABL
#define W4B_ABL 0      // (timing builds, wrong results) 1: no three-term split, 2: no input transform in the steady state, 4: no U refills, 8: no V reads after a half's first group
#endif
#ifndef W4B_GRP
#define W4B_GRP 3      // positions per MFMA group (2 or 3: must divide 18)
#endif
#ifndef W4B_SWAP
#define W4B_SWAP 0     // 1: u0 v1 + u1 v0 as ONE v_mfma_f32_16x16x32_bf16 on the pair (v1, v0) read from LDS a second time (ds_read2_b64), instead of two K = 16 MFMAs
#endif
#ifndef W4B_U2
#define W4B_U2 1       // 0 (timing builds: weights carried to 16 significant bits only): no u2 term -- 16 B instead of 24 B of U per lane and step
#endif
#ifndef W4B_URING
#define W4B_URING 6    // (position, chunk) steps of U terms in flight per consumer wave (6 registers each); must divide 36
#endif

namespace {

constexpr int VA_BYTES = 36 * 1024;     // VA[position][lane] = (v0, v1) of (tile = lane & 15, channel quad = lane >> 4): 16 B
constexpr int V2_BYTES = 36 * 512;      // V2[position][lane] = v2: 8 B
constexpr int PLANES_BYTES = 4 * 4096 * 4;
constexpr int HROW = 36;                // staged halo: 10 rows x 36 column slots (columns in phase-major order), 64 B per pixel
constexpr int HBYTES = 10 * HROW * 64;
constexpr int SMEM_BYTES = VA_BYTES + V2_BYTES + PLANES_BYTES + HBYTES + 1024;   // + landing zone of the surplus staging slots: 144896
constexpr int HPIX = 10 * 34;
constexpr int U1_BYTES = 144 * 4 * 64 * 16, U2_BYTES = 144 * 4 * 64 * 8;
constexpr int OOB = (int)0x80000000u;

struct Wino4bArgs {
  const float* X; int x_stride; int x_choff; int x_bytes;
  int y_bytes, r1_bytes, r2_bytes, mk_bytes;
  float* dbg;        // (LFSR_CONV_DIAG builds: the stamp buffer set by lfsr_diag_set_buffer; else unused)
  const float* Wu;   // U1 [144 steps = chunk * 36 + position][4 ns][64 lanes][(u0, u1) x 4 channels bf16] then U2 [144][4][64][u2 x 4]   (lfsr_pack_wino4b)
  float* Y; int y_stride; int y_choff;
  const float* R1; int r1_stride; int r1_choff;
  const float* R2; int r2_stride; int r2_choff;
  const float* Mk; int mk_stride; int mk_choff; float mk_slope;
  int n_img, H, W, tiles_y, tiles_x, ntiles;
  float slope;
};

#ifdef LFSR_CONV_DIAG
// diagnostic build only: consumer wave 0 / producer wave 4 accumulate s_memtime deltas per segment (64 floats per block: consumer 0..31, producer 32..63)
#define STAMP(k) do { long long t_ = clock64(); seg[k] += (unsigned)(t_ - tprev); tprev = t_; } while (0)
#else
#define STAMP(k) do { } while (0)
#endif
#define LDS_BARRIER() do { __builtin_amdgcn_sched_barrier(0); asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); __builtin_amdgcn_sched_barrier(0); } while (0)

__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* base, int bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, bytes, 0x00020000);
}
__device__ __forceinline__ f32x4 bload4(__amdgpu_buffer_rsrc_t r, int voff, int soff) {
  return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0));
}
__device__ __forceinline__ void bstore4(__amdgpu_buffer_rsrc_t r, int voff, f32x4 v) {
  __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), r, voff, 0, 0);
}
__device__ __forceinline__ unsigned hi_pair(unsigned odd, unsigned even) {   // (upper 16 bits of odd) << 16 | upper 16 bits of even: two truncated bf16
  return __builtin_amdgcn_perm(odd, even, 0x07060302u);
}
__device__ __forceinline__ float trunc_residual(float v) {   // v - bf16_trunc(v): exact
  return v - __uint_as_float(__float_as_uint(v) & 0xffff0000u);
}
// the three bf16 terms of four channels: va = (v0 pair, v0 pair, v1 pair, v1 pair), x2 = (v2 pair, v2 pair)
__device__ __forceinline__ void split3(const f32x4 v, u32x4& va, u32x2& x2) {
  // (scalar copies first: __builtin_bit_cast applied to a vector ELEMENT expression reads element 0 with this compiler)
  const float a0 = v.x, a1 = v.y, a2 = v.z, a3 = v.w;
  const float r0 = trunc_residual(a0), r1 = trunc_residual(a1), r2 = trunc_residual(a2), r3 = trunc_residual(a3);
  const float q0 = trunc_residual(r0), q1 = trunc_residual(r1), q2 = trunc_residual(r2), q3 = trunc_residual(r3);
  va.x = hi_pair(__float_as_uint(a1), __float_as_uint(a0));
  va.y = hi_pair(__float_as_uint(a3), __float_as_uint(a2));
  va.z = hi_pair(__float_as_uint(r1), __float_as_uint(r0));
  va.w = hi_pair(__float_as_uint(r3), __float_as_uint(r2));
  x2.x = hi_pair(__float_as_uint(q1), __float_as_uint(q0));
  x2.y = hi_pair(__float_as_uint(q3), __float_as_uint(q2));
}

// The MFMAs are issued through asm statements with the accumulator tied ("+v": vDst = SrcC, disjoint from the A / B registers).  With the
// builtins the register allocator may give v_mfma_f32_16x16x16_bf16 a vDst that overlaps its SrcB / SrcC PARTIALLY (it models 4-register results as
// read-before-write); on gfx950 that returns garbage in the overlapping registers (found with the W4B_OPS debug builds: every product group that
// used the builtin corrupted accumulator registers 0 and 1).  hipcc neither pads nor reorders asm: dependent MFMAs are two instructions apart
// (three positions interleaved), and the accumulators are next read by VALU instructions behind a barrier.
__device__ __forceinline__ void mfma32(f32x4& c, const u32x4 a, const u32x4 b) {
  asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b));
}
__device__ __forceinline__ void mfma32_first(f32x4& c, const u32x4 a, const u32x4 b) {
  asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, 0" : "=&v"(c) : "v"(a), "v"(b));
}
__device__ __forceinline__ void mfma16(f32x4& c, const u32x2 a, const u32x2 b) {
  asm volatile("v_mfma_f32_16x16x16_bf16 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b));
}

// LDS reads of the consumers as asm statements with immediate offsets (the compiler neither counts nor merges them: lds_wait<N>() = all but
// the N youngest are done); compile-time loops (static_for) make the offsets constant expressions
template <int OFF> __device__ __forceinline__ u32x4 lds_read_b128(unsigned addr) { u32x4 v; asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF)); return v; }
template <int OFF> __device__ __forceinline__ u32x2 lds_read_b64(unsigned addr) { u32x2 v; asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF)); return v; }
template <int OFF8> __device__ __forceinline__ u32x4 lds_read2_b64_swapped(unsigned addr) {   // (8 B at OFF8 * 8 + 8, 8 B at OFF8 * 8): the pair (v1, v0) of a VA record
  u32x4 v; asm volatile("ds_read2_b64 %0, %1 offset0:%2 offset1:%3" : "=v"(v) : "v"(addr), "n"(OFF8 + 1), "n"(OFF8)); return v;
}
template <int N> __device__ __forceinline__ void lds_wait() { asm volatile("s_waitcnt lgkmcnt(%0)" :: "n"(N)); }
template <typename F, int... I> __device__ __forceinline__ void static_for_impl(F& f, std::integer_sequence<int, I...>) { (f(std::integral_constant<int, I>{}), ...); }
template <int N, typename F> __device__ __forceinline__ void static_for(F&& f) { static_for_impl(f, std::make_integer_sequence<int, N>{}); }

// one 6-vector of the output transform: y = At m  (4 results), as conv3x3_wino4.hip
__device__ __forceinline__ void at6(f32x4& m0, f32x4& m1, f32x4& m2, f32x4& m3, const f32x4 m4, const f32x4 m5) {
  const f32x4 s12 = m1 + m2, d12 = m1 - m2, s34 = m3 + m4, d34 = m3 - m4;
  m0 = (m0 + s12) + s34;
  m1 = d12 + 2.f * d34;
  m2 = s12 + 4.f * s34;
  m3 = (d12 + 8.f * d34) + m5;
}

// three of the six results of t = Bt d (rows / columns 0..5 of the 6-vector d), the same fma sequences as conv3x3_wino4.hip's bt6:
//   LOWER half (HALF = 0): t0 t1 t2 from d0..d4;   UPPER half (HALF = 1): t3 t4 t5 from d1..d5.   q[0..4] = d[HALF .. HALF + 4]
template <int HALF>
__device__ __forceinline__ void bt_half(const f32x4 q0, const f32x4 q1, const f32x4 q2, const f32x4 q3, const f32x4 q4, f32x4& o0, f32x4& o1, f32x4& o2) {
  const f32x4 m4 = {-4.f, -4.f, -4.f, -4.f}, p4 = {4.f, 4.f, 4.f, 4.f}, m5 = {-5.f, -5.f, -5.f, -5.f}, p2 = {2.f, 2.f, 2.f, 2.f}, m2 = {-2.f, -2.f, -2.f, -2.f};
  if (HALF == 0) {   // q = d0 d1 d2 d3 d4
    const f32x4 a = __builtin_elementwise_fma(m4, q2, q4), b = __builtin_elementwise_fma(m4, q1, q3);
    o0 = __builtin_elementwise_fma(p4, q0, __builtin_elementwise_fma(m5, q2, q4));
    o1 = a + b; o2 = a - b;
  } else {           // q = d1 d2 d3 d4 d5
    const f32x4 c = q3 - q1, e = q2 - q0;
    o0 = __builtin_elementwise_fma(p2, e, c);
    o1 = __builtin_elementwise_fma(m2, e, c);
    o2 = __builtin_elementwise_fma(p4, q0, __builtin_elementwise_fma(m5, q2, q4));
  }
}

// ============================================================ PRODUCER =====================================================================
// XH / NH: the wave's quarter of the transform domain: positions p = 6 xi + nu with xi in {3 XH .. 3 XH + 2}, nu in {3 NH .. 3 NH + 2}
template <bool MASK, bool HAS_E, bool HAS_L, int XH, int NH>
__device__ __forceinline__ void producer(const Wino4bArgs& p, char* const smem, int tile, const int tile_end) {
  char* const sVA = smem;
  char* const sV2 = smem + VA_BYTES;
  float* const sX = reinterpret_cast<float*>(smem + VA_BYTES + V2_BYTES);
  char* const sH = smem + VA_BYTES + V2_BYTES + PLANES_BYTES;
  const int tid = threadIdx.x & 255, lane = tid & 63;
  const int t16 = lane & 15, g = lane >> 4, pty = t16 >> 3, ptx = t16 & 7;
  const float* const Ep = MASK ? p.Mk : p.R1;
  const float* const Lp = MASK ? p.R1 : p.R2;
  const int e_stride = MASK ? p.mk_stride : p.r1_stride, e_choff = MASK ? p.mk_choff : p.r1_choff;
  const int l_stride = MASK ? p.r1_stride : p.r2_stride, l_choff = MASK ? p.r1_choff : p.r2_choff;
  const int img_px = p.H * p.W;
  // one descriptor per (operand, image): rows above / below the image are out-of-range offsets; an absent operand / tile gets extent 0
  auto img_rsrc = [&](const float* base, int stride, int img) {
    const bool ok = base != nullptr && img >= 0;
    return make_rsrc(base + (ok ? (long long)img * img_px * stride : 0), ok ? img_px * stride * 4 : 0);
  };
  // ---- halo staging slots: slot i of a thread = (halo pixel, 16-B quarter of the chunk's 64 B) = (idx >> 2, idx & 3), idx = tid + 256 i.
  // Staged image: pixel (row, col) quarter q at ((row * 36 + (col & 3) * 9 + (col >> 2)) * 64 + 16 (q ^ (((row >> 2) & 1) << 1)): columns in
  // phase-major order and the quarter swizzled by the row group, so that the 64 lanes (16 tiles x 4 quads) of a patch read -- same patch
  // position, tile origins 4 columns / 4 rows apart -- fall into 64 different 16-B slots per 1 KB: ds_read_b128 without bank conflicts
  int hrel[6], hcol[6];
  char* hdst[6];
#pragma unroll
  for (int i = 0; i < 6; ++i) {
    const int px = (tid + 256 * i) >> 2, cq = tid & 3;
    const int r = __mul24(px, 1928) >> 16;   // px / 34 for px < 384
    const int c = px - r * 34;
    hrel[i] = (((r - 1) * p.W + (c - 1)) * p.x_stride + p.x_choff) * 4 + cq * 16;
    hcol[i] = px < HPIX ? c - 1 : -(1 << 20);
    hdst[i] = px < HPIX ? sH + (r * HROW + (c & 3) * 9 + (c >> 2)) * 64 + 16 * (cq ^ (((r >> 2) & 1) << 1)) : sH + HBYTES + (tid & 63) * 16;
  }
  int hx[6];
  auto halo_offsets = [&](int y0, int x0) {
    const int toff = (y0 * p.W + x0) * (p.x_stride * 4);   // wave-uniform
#pragma unroll
    for (int i = 0; i < 6; ++i) hx[i] = (unsigned)(x0 + hcol[i]) < (unsigned)p.W ? hrel[i] + toff : OOB;
  };
  f32x4 hv0[6], hv1[6];
  auto halo_load = [&](f32x4 (&hv)[6], __amdgpu_buffer_rsrc_t rs, int chunk) {
#pragma unroll
    for (int i = 0; i < 6; ++i) hv[i] = bload4(rs, hx[i], chunk * 64);
  };
  auto halo_store = [&](f32x4 (&hv)[6]) {
#pragma unroll
    for (int i = 0; i < 6; ++i) *reinterpret_cast<f32x4*>(hdst[i]) = hv[i];
  };
  // ---- this thread's raw patch: rows 4 pty + r (r = XH .. XH + 4), columns 4 ptx + j (j = NH .. NH + 4), channel quad g.
  // Two bases: rows r < 4 lie in row group pty, rows 4, 5 in row group pty + 1 (the quarter swizzle)
  const char* const hb0 = sH + ((4 * pty) * HROW + ptx) * 64 + 16 * (g ^ ((pty & 1) << 1));
  const char* const hb1 = sH + ((4 * pty) * HROW + ptx) * 64 + 16 * (g ^ (((pty + 1) & 1) << 1));
  f32x4 S[3][5];        // after stage 1: xi = 3 XH + a, column NH + b
  auto stage1 = [&]() {
#pragma unroll
    for (int b = 0; b < 5; ++b) {
      const int j = NH + b;
      f32x4 d[5];
#pragma unroll
      for (int a = 0; a < 5; ++a) {
        const int r = XH + a;
        d[a] = *reinterpret_cast<const f32x4*>((r < 4 ? hb0 : hb1) + (r * HROW + (j & 3) * 9 + (j >> 2)) * 64);
      }
      bt_half<XH>(d[0], d[1], d[2], d[3], d[4], S[0][b], S[1][b], S[2][b]);
    }
  };
  f32x4 Ov[9];                // V of this thread's 9 positions (fp32), 4 channels each
  u32x4 RA[9]; u32x2 R2[9];   // ... and their three-term records, held until their half of V is free
  auto stage2 = [&]() {
#pragma unroll
    for (int a = 0; a < 3; ++a) bt_half<NH>(S[a][0], S[a][1], S[a][2], S[a][3], S[a][4], Ov[3 * a + 0], Ov[3 * a + 1], Ov[3 * a + 2]);
  };
  auto split_all = [&]() {
#pragma unroll
    for (int q = 0; q < 9; ++q) {
      if (W4B_ABL & 1) { RA[q] = __builtin_bit_cast(u32x4, Ov[q]); R2[q] = u32x2{RA[q].x, RA[q].y}; }
      else split3(Ov[q], RA[q], R2[q]);
    }
  };
  auto write_v = [&]() {
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
      for (int b = 0; b < 3; ++b) {
        const int pos = 6 * (3 * XH + a) + 3 * NH + b;
        *reinterpret_cast<u32x4*>(sVA + pos * 1024 + lane * 16) = RA[3 * a + b];
        *reinterpret_cast<u32x2*>(sV2 + pos * 512 + lane * 8) = R2[3 * a + b];
      }
  };
  // ---- drain slots of an exchange plane (as conv3x3_wino4.hip): slot i = pixel (row 4 (i >> 1) [+ a], column 16 (i & 1) + tid / 16), 16-B unit tid % 16
  const int un = tid & 15;
  int pY[4], pE[4], pL[4], dcol[4];
  const float* dsrc[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int px = (tid + 256 * i) >> 4;
    const int pr = (4 * (px >> 5)) * p.W + (px & 31);
    dcol[i] = px & 31;
    pY[i] = (pr * p.y_stride + p.y_choff) * 4 + un * 16;
    pE[i] = (pr * e_stride + e_choff) * 4 + un * 16;
    pL[i] = (pr * l_stride + l_choff) * 4 + un * 16;
    dsrc[i] = sX + px * 64 + ((un ^ ((px >> 2) & 7)) << 2);
  }
  __amdgpu_buffer_rsrc_t rsYp = img_rsrc(nullptr, 0, -1), rsEp = rsYp, rsLp = rsYp;   // the PREVIOUS tile's image (none yet)
  int prow0 = 0, pcol0 = 0;
  const bool ragged_w = (p.W & 31) != 0;
  int oy[4];
  f32x4 e[4];
  auto drain_request = [&](int a) {
    const int rowoff = (prow0 + a) * p.W + pcol0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const bool bad = ragged_w && pcol0 + dcol[i] >= p.W;
      const int offy = pY[i] + rowoff * (p.y_stride * 4), offe = pE[i] + rowoff * (e_stride * 4);
      oy[i] = bad ? OOB : offy;
      if (HAS_E) e[i] = bload4(rsEp, bad ? OOB : offe, 0);
    }
  };
  auto drain_plane = [&](int a) {
    const int rowoff = (prow0 + a) * p.W + pcol0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      f32x4 v = *reinterpret_cast<const f32x4*>(dsrc[i] + a * 4096);
      if (p.slope <= 1.f) {
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] = fmaxf(v[k], v[k] * p.slope);
      } else {
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] = v[k] >= 0.f ? v[k] : v[k] * p.slope;
      }
      if (MASK) {
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] *= e[i][k] > 0.f ? 1.f : p.mk_slope;
      } else if (HAS_E) {
        v += e[i];
      }
      if (HAS_L) { const int offl = pL[i] + rowoff * (l_stride * 4); v += bload4(rsLp, oy[i] == OOB ? OOB : offl, 0); }
      bstore4(rsYp, oy[i], v);
    }
  };

  // ---- prologue: chunk 0 of the first tile staged, transformed and published; chunks 1 (staged), 2, 3 (registers) on their way
  int img, y0, x0;
  { int txx = tile % p.tiles_x; int q = tile / p.tiles_x; int tyy = q % p.tiles_y; img = q / p.tiles_y; y0 = tyy * 8; x0 = txx * 32; }
  __amdgpu_buffer_rsrc_t rsXc = img_rsrc(p.X, p.x_stride, img), rsXn = rsXc;
  halo_offsets(y0, x0);
  halo_load(hv0, rsXc, 0);
  halo_load(hv1, rsXc, 1);
  halo_store(hv0);
  halo_load(hv0, rsXc, 2);
  LDS_BARRIER();   // (P1: halo of chunk 0 staged)
  stage1();
  stage2();
  LDS_BARRIER();   // (P2: everyone has read it)
  halo_store(hv1);
  halo_load(hv1, rsXc, 3);
  split_all();
  write_v();
  bool pending = false;   // (XH = 1) records waiting for the upper half of V
  LDS_BARRIER();   // (P3 = "H2 of chunk -1": V of chunk 0 published, halo of chunk 1 staged)
  bool first_tile = true;
#ifdef LFSR_CONV_DIAG
  unsigned seg[16] = {};
  long long tprev = clock64();
#endif
  while (true) {
    const int next = tile + 1;
    const bool has_next = next < tile_end;
    int nimg = img, ny0 = y0, nx0 = x0 + 32;
    if (nx0 >= p.W) { nx0 = 0; ny0 += 8; if (ny0 >= p.H) { ny0 = 0; nimg += 1; } }
    if (!has_next) nimg = -1;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      // ---- X_c: the consumers are on the lower half of chunk c.  Upper-half records of chunk c (held since Y_{c-1}) -> V; stage 1 of chunk c + 1
      if (XH == 1 && pending) write_v();
      __builtin_amdgcn_sched_barrier(0);
      if (!(W4B_ABL & 2)) { stage1(); stage2(); }
      __builtin_amdgcn_sched_barrier(0);
      drain_request(c);
      __builtin_amdgcn_sched_barrier(0);
      STAMP(0);
      if (c == 0 && !first_tile) { LDS_BARRIER(); }   // (E: the previous tile's results are in the exchange planes)
      STAMP(1);
      LDS_BARRIER();   // (H1: lower half of V free, staged halo free)
      STAMP(2);
      // ---- Y_c: the consumers are on the upper half of chunk c.  Halo of chunk c + 2 -> LDS, the halo of chunk c + 4 (the next tile's chunk c)
      // requested; stage 2 + split of chunk c + 1; lower-half records -> V; one exchange plane of the previous tile drained
      f32x4 (&hv)[6] = (c & 1) ? hv1 : hv0;
      halo_store(hv);
      if (c == 0) { rsXn = img_rsrc(p.X, p.x_stride, nimg); halo_offsets(ny0, nx0); }
      __builtin_amdgcn_sched_barrier(0);
      split_all();
      if (XH == 0) write_v(); else pending = true;
      __builtin_amdgcn_sched_barrier(0);
      STAMP(3);
      drain_plane(c);
      __builtin_amdgcn_sched_barrier(0);
      halo_load(hv, rsXn, c);
      __builtin_amdgcn_sched_barrier(0);
      STAMP(4);
      LDS_BARRIER();   // (H2: upper half of V free; lower half of chunk c + 1 published, halo of chunk c + 2 staged)
      STAMP(5);
    }
    rsYp = img_rsrc(p.Y, p.y_stride, img); rsEp = img_rsrc(Ep, e_stride, img); rsLp = img_rsrc(Lp, l_stride, img);
    prow0 = y0; pcol0 = x0;
    first_tile = false;
    if (!has_next) {
      LDS_BARRIER();   // (E of the last tile)
#pragma unroll
      for (int a = 0; a < 4; ++a) { drain_request(a); drain_plane(a); }
      break;
    }
    tile = next; img = nimg; y0 = ny0; x0 = nx0;
    rsXc = rsXn;
  }
#ifdef LFSR_CONV_DIAG
  if (p.dbg && threadIdx.x == 256 + 64 * (2 * XH + NH)) for (int k = 0; k < 8; ++k) p.dbg[blockIdx.x * 64 + 32 + 8 * (2 * XH + NH) + k] = (float)seg[k];
#endif
}

// HAS_E / HAS_L: the first (residual R1, or the saved activation of the LeakyReLU' mask) / second epilogue operand exists.
template <bool MASK, bool HAS_E, bool HAS_L>
__global__ __launch_bounds__(512) void k_conv3x3_wino4b(Wino4bArgs p) {
  extern __shared__ __attribute__((aligned(16))) char smem_b[];
  const int lane = threadIdx.x & 63, w4 = (threadIdx.x >> 6) & 3;
  const int nblk = gridDim.x;
  int tile = (int)(((long long)blockIdx.x * p.ntiles) / nblk);
  const int tile_end = (int)(((long long)(blockIdx.x + 1) * p.ntiles) / nblk);
  if (threadIdx.x >= 256) {
    const int part = __builtin_amdgcn_readfirstlane(w4);
    if (part == 0) producer<MASK, HAS_E, HAS_L, 0, 0>(p, smem_b, tile, tile_end);
    else if (part == 1) producer<MASK, HAS_E, HAS_L, 0, 1>(p, smem_b, tile, tile_end);
    else if (part == 2) producer<MASK, HAS_E, HAS_L, 1, 0>(p, smem_b, tile, tile_end);
    else producer<MASK, HAS_E, HAS_L, 1, 1>(p, smem_b, tile, tile_end);
    return;
  }
  // ======================================================= CONSUMER ==========================================================
  const unsigned sVAa = (unsigned)(size_t)(smem_b + lane * 16);              // LDS byte addresses (the low 32 bits of a shared-memory pointer)
  const unsigned sV2a = (unsigned)(size_t)(smem_b + VA_BYTES + lane * 8);
  float* const sX = reinterpret_cast<float*>(smem_b + VA_BYTES + V2_BYTES);
  const int ctile = lane & 15, kk = lane >> 4, cty = ctile >> 3, ctx = ctile & 7;
  const int ns = w4;
  const __amdgpu_buffer_rsrc_t rsW = make_rsrc(p.Wu, U1_BYTES + U2_BYTES);
  const int uoff1 = ns * 1024 + lane * 16, uoff2 = U1_BYTES + ns * 512 + lane * 8;
  const int xw = (cty * 32 + 4 * ctx) * 64 + (((4 * ns + kk) ^ ctx) << 2);   // exchange: pixel (row cty, col 4 ctx + b), 16-B unit XOR ctx
  u32x4 U1[W4B_URING]; u32x2 U2[W4B_URING];
#pragma unroll
  for (int i = 0; i < W4B_URING; ++i) {
    U1[i] = __builtin_amdgcn_raw_buffer_load_b128(rsW, uoff1, i * 4096, 0);
    if (W4B_U2) U2[i] = __builtin_bit_cast(u32x2, __builtin_amdgcn_raw_buffer_load_b64(rsW, uoff2, i * 2048, 0));
  }
  f32x4 acc[36];
  LDS_BARRIER();   // (P1)
  LDS_BARRIER();   // (P2)
  LDS_BARRIER();   // (P3)
#ifdef LFSR_CONV_DIAG
  unsigned seg[16] = {};
  long long tprev = clock64();
#endif
  while (true) {
    const bool has_next = tile + 1 < tile_end;
    // W4B_GRP positions per group, their MFMAs interleaved so that an accumulator is used every W4B_GRP-th instruction; the V records of
    // group g + 1 are requested before the MFMAs of group g (not across the half barrier: the other half is being written until then).
    // The LDS reads are asm statements (the swapped pair (v1, v0) must really be read again, not rebuilt with v_mov): waits are counted here
    static_for<8>([&](auto CH) {
      constexpr int c = decltype(CH)::value >> 1, half = decltype(CH)::value & 1;
      constexpr int G = W4B_GRP, NG = 18 / G, RPP = W4B_SWAP ? 3 : 2;   // RPP = LDS reads per position
      // (the swapped pair lives in one 2048-B window per half: ds_read2_b64 offsets are 8-bit, in units of 8 B, so its base moves with the position)
      u32x4 va[2][G], vs[2][G]; u32x2 v2[2][G];
      auto request = [&](auto BUF, auto POS) {
        constexpr int buf = decltype(BUF)::value, pos = decltype(POS)::value;
        static_for<G>([&](auto I) {
          constexpr int i = decltype(I)::value;
          if ((W4B_ABL & 8) && (pos + i) % 18 >= G) return;   // (timing build: only the first group of a half is read)
          va[buf][i] = lds_read_b128<(pos + i) * 1024>(sVAa);
          if (W4B_SWAP) vs[buf][i] = lds_read2_b64_swapped<0>(sVAa + (pos + i) * 1024);
          v2[buf][i] = lds_read_b64<(pos + i) * 512>(sV2a);
        });
      };
      request(std::integral_constant<int, 0>{}, std::integral_constant<int, 18 * half>{});
      static_for<NG>([&](auto G3) {
        constexpr int g3 = decltype(G3)::value, pos0 = 18 * half + G * g3, cur = g3 & 1, nxt = cur ^ 1;
        if constexpr (g3 < NG - 1) request(std::integral_constant<int, nxt>{}, std::integral_constant<int, pos0 + G>{});
        u32x4 u1[G]; u32x2 u2[G];
#pragma unroll
        for (int i = 0; i < G; ++i) {
          const int t = c * 36 + pos0 + i;
          u1[i] = U1[t % W4B_URING]; if (W4B_U2) u2[i] = U2[t % W4B_URING];
        }
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (g3 < NG - 1 && !(W4B_ABL & 8)) lds_wait<G * RPP>(); else lds_wait<0>();
        // (u0, u1) . (v0, v1)
#pragma unroll
        for (int i = 0; i < G; ++i) {
          if (c == 0) mfma32_first(acc[pos0 + i], u1[i], va[cur][i]); else mfma32(acc[pos0 + i], u1[i], va[cur][i]);
        }
        // u0 v1 + u1 v0, u0 v2, u2 v0
        if (W4B_SWAP) {
#pragma unroll
          for (int i = 0; i < G; ++i) if (W4B_OPS & 2) mfma32(acc[pos0 + i], u1[i], vs[cur][i]);
        } else {
#pragma unroll
          for (int i = 0; i < G; ++i) if (W4B_OPS & 2) mfma16(acc[pos0 + i], u32x2{u1[i].x, u1[i].y}, u32x2{va[cur][i].z, va[cur][i].w});
#pragma unroll
          for (int i = 0; i < G; ++i) if (W4B_OPS & 4) mfma16(acc[pos0 + i], u32x2{u1[i].z, u1[i].w}, u32x2{va[cur][i].x, va[cur][i].y});
        }
#pragma unroll
        for (int i = 0; i < G; ++i) if (W4B_OPS & 8) mfma16(acc[pos0 + i], u32x2{u1[i].x, u1[i].y}, v2[cur][i]);
#pragma unroll
        for (int i = 0; i < G; ++i) if ((W4B_U2 != 0) & ((W4B_OPS & 16) != 0)) mfma16(acc[pos0 + i], u2[i], u32x2{va[cur][i].x, va[cur][i].y});
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < G; ++i) {
          const int t = c * 36 + pos0 + i;
          const int tn = (t + W4B_URING) % 144;     // (wraps into the next tile: same weights)
          if (W4B_ABL & 4) continue;   // (timing build: the U ring is never refilled)
          U1[t % W4B_URING] = __builtin_amdgcn_raw_buffer_load_b128(rsW, uoff1, tn * 4096, 0);
          if (W4B_U2) U2[t % W4B_URING] = __builtin_bit_cast(u32x2, __builtin_amdgcn_raw_buffer_load_b64(rsW, uoff2, tn * 2048, 0));
        }
        __builtin_amdgcn_sched_barrier(0);
      });
      STAMP(half);
      LDS_BARRIER();   // (H1 / H2)
      STAMP(2 + half);
    });
    // At M A in registers; one output row a of every Winograd tile per round -> exchange planes
#pragma unroll
    for (int nu = 0; nu < 6; ++nu) at6(acc[nu], acc[6 + nu], acc[12 + nu], acc[18 + nu], acc[24 + nu], acc[30 + nu]);
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      float* const xb = sX + a * 4096;
      at6(acc[6 * a], acc[6 * a + 1], acc[6 * a + 2], acc[6 * a + 3], acc[6 * a + 4], acc[6 * a + 5]);
#pragma unroll
      for (int b = 0; b < 4; ++b) *reinterpret_cast<f32x4*>(xb + xw + b * 64) = acc[6 * a + b];
    }
    STAMP(4);
    LDS_BARRIER();   // (E: the tile's results are in the exchange planes; the producers drain them during the next tile)
    STAMP(5);
    if (!has_next) break;
    tile += 1;
  }
#ifdef LFSR_CONV_DIAG
  if (p.dbg && (threadIdx.x & 63) == 0) for (int k = 0; k < 8; ++k) p.dbg[blockIdx.x * 64 + 8 * w4 + k] = (float)seg[k];
#endif
}

// rne bf16 of a float, returned as a float
__device__ __forceinline__ float rne_bf16(float v) {
  unsigned u = __builtin_bit_cast(unsigned, v);
  u = (u + 0x7fffu + ((u >> 16) & 1u)) & 0xffff0000u;
  return __builtin_bit_cast(float, u);
}

// U = G g G^t per (n, k) from the direct pack [tap][n][k], split into three bf16 terms:
//   U1 [step = (k / 16) * 36 + p][ns = n / 16][lane = 16 ((k / 4) % 4) + n % 16][(u0, u1)][k % 4]  (16 B per lane),  U2 [step][ns][lane][k % 4] = u2 (8 B per lane)
__global__ __launch_bounds__(256) void k_pack_wino4b(const float* __restrict__ direct, unsigned short* __restrict__ out) {
  const int i = blockIdx.x * 256 + threadIdx.x;   // (n, k)
  if (i >= 64 * 64) return;
  const int n = i >> 6, k = i & 63;
  double gq[3][3];
#pragma unroll
  for (int t = 0; t < 9; ++t) gq[t / 3][t % 3] = (double)direct[(t * 64 + n) * 64 + k];
  const double G[6][3] = {{1.0 / 4, 0.0, 0.0}, {-1.0 / 6, -1.0 / 6, -1.0 / 6}, {-1.0 / 6, 1.0 / 6, -1.0 / 6},
                          {1.0 / 24, 1.0 / 12, 1.0 / 6}, {1.0 / 24, -1.0 / 12, 1.0 / 6}, {0.0, 0.0, 1.0}};
  double tmp[6][3];
#pragma unroll
  for (int a = 0; a < 6; ++a)
#pragma unroll
    for (int c = 0; c < 3; ++c) tmp[a][c] = G[a][0] * gq[0][c] + G[a][1] * gq[1][c] + G[a][2] * gq[2][c];
  const int chunk = k >> 4, grp = (k >> 2) & 3, ki = k & 3, nsl = n >> 4, m = n & 15;
  const int ln = grp * 16 + m;
  unsigned short* const out2 = out + U1_BYTES / 2;
#pragma unroll
  for (int a = 0; a < 6; ++a)
#pragma unroll
    for (int b = 0; b < 6; ++b) {
      const float u = (float)(tmp[a][0] * G[b][0] + tmp[a][1] * G[b][1] + tmp[a][2] * G[b][2]);
      const float u0 = rne_bf16(u), r1 = u - u0, u1 = rne_bf16(r1), u2 = rne_bf16(r1 - u1);
      const int step = chunk * 36 + a * 6 + b;
      const size_t rec = (size_t)(step * 4 + nsl) * 64 + ln;
      out[rec * 8 + ki] = (unsigned short)(__builtin_bit_cast(unsigned, u0) >> 16);
      out[rec * 8 + 4 + ki] = (unsigned short)(__builtin_bit_cast(unsigned, u1) >> 16);
      out2[rec * 4 + ki] = (unsigned short)(__builtin_bit_cast(unsigned, u2) >> 16);
    }
}

}  // namespace

int lfsr_pack_wino4b(const float* direct_packed, float* out, hipStream_t st) {
  if (!direct_packed || !out) return LFSR_E_ARG;
  static_assert((U1_BYTES + U2_BYTES) == LFSR_CONV3_WINO4B_FLOATS * 4, "pack size");
  hipLaunchKernelGGL(k_pack_wino4b, dim3(16), dim3(256), 0, st, direct_packed, reinterpret_cast<unsigned short*>(out));
  LFSR_CHECK_LAUNCH();
  return LFSR_OK;
}

// LFSR_E_ARG = geometry not covered (operands of 1 GiB and more): the caller falls back
int lfsr_conv3x3_wino4b_launch(const float* x, int x_stride, int x_choff, const float* w_wino4b, float* y, int y_stride, int y_choff,
                               const float* r1, int r1_stride, int r1_choff, const float* r2, int r2_stride, int r2_choff,
                               const float* mk, int mk_stride, int mk_choff, float mk_slope,
                               int n_img, int h, int w, float slope, hipStream_t st) {
  static std::atomic<bool> attr_set[64];
  static std::atomic<int> cus[64];
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return LFSR_E_ARG;
  if (!attr_set[dev]) {
    const void* fns[5] = {reinterpret_cast<const void*>(k_conv3x3_wino4b<false, false, false>), reinterpret_cast<const void*>(k_conv3x3_wino4b<false, true, false>),
                          reinterpret_cast<const void*>(k_conv3x3_wino4b<false, true, true>), reinterpret_cast<const void*>(k_conv3x3_wino4b<true, true, false>),
                          reinterpret_cast<const void*>(k_conv3x3_wino4b<true, true, true>)};
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
  if ((long long)n_img * h * w * ms * 4 >= (1LL << 30)) return LFSR_E_ARG;
  Wino4bArgs p{};
  p.X = x; p.x_stride = x_stride; p.x_choff = x_choff; p.x_bytes = (int)((long long)n_img * h * w * x_stride * 4);
  p.Wu = w_wino4b;
  p.Y = y; p.y_stride = y_stride; p.y_choff = y_choff;
  p.R1 = r1; p.r1_stride = r1_stride; p.r1_choff = r1_choff; p.R2 = r2; p.r2_stride = r2_stride; p.r2_choff = r2_choff;
  p.Mk = mk; p.mk_stride = mk_stride; p.mk_choff = mk_choff; p.mk_slope = mk_slope;
  const long long npix4 = (long long)n_img * h * w * 4;
  p.y_bytes = (int)(npix4 * y_stride); p.r1_bytes = r1 ? (int)(npix4 * r1_stride) : 0; p.r2_bytes = r2 ? (int)(npix4 * r2_stride) : 0;
  p.mk_bytes = mk ? (int)(npix4 * mk_stride) : 0;
#ifdef LFSR_CONV_DIAG
  p.dbg = g_lfsr_diag_buf;
#endif
  p.n_img = n_img; p.H = h; p.W = w; p.tiles_y = (h + 7) / 8; p.tiles_x = (w + 31) / 32; p.slope = slope;
  const long long nt = (long long)n_img * p.tiles_y * p.tiles_x;
  if (nt <= 0 || nt > 0x7fffffffLL) return LFSR_E_ARG;
  p.ntiles = (int)nt;
  const int slots = cus[dev];
  const unsigned grid = (unsigned)(nt < slots ? nt : slots);
  if (!mk && !r1 && r2) { p.R1 = r2; p.r1_stride = r2_stride; p.r1_choff = r2_choff; p.r1_bytes = p.r2_bytes; p.R2 = nullptr; p.r2_bytes = 0; }   // a lone residual is the first operand
  if (mk && p.R1) hipLaunchKernelGGL((k_conv3x3_wino4b<true, true, true>), dim3(grid), dim3(512), SMEM_BYTES, st, p);
  else if (mk) hipLaunchKernelGGL((k_conv3x3_wino4b<true, true, false>), dim3(grid), dim3(512), SMEM_BYTES, st, p);
  else if (p.R1 && p.R2) hipLaunchKernelGGL((k_conv3x3_wino4b<false, true, true>), dim3(grid), dim3(512), SMEM_BYTES, st, p);
  else if (p.R1) hipLaunchKernelGGL((k_conv3x3_wino4b<false, true, false>), dim3(grid), dim3(512), SMEM_BYTES, st, p);
  else hipLaunchKernelGGL((k_conv3x3_wino4b<false, false, false>), dim3(grid), dim3(512), SMEM_BYTES, st, p);
  LFSR_CHECK_LAUNCH();
  return LFSR_OK;
}
