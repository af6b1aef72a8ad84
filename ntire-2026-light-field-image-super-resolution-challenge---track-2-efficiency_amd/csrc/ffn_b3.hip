// Fused transformer feed-forward block on the bf16 MFMA pipe with fp32 operands carried EXACTLY as three bf16 terms (default; LFSR_FFN=f32 keeps ffn_fused.hip's
// fp32-MFMA kernel):   y = res + W2 . relu(W1 . LN(x))        (EPIT.py:84-90,126 / LFT.py:151-156,202 / LFT.py:216-221,243)
// Every fp32 operand -- tokens, both weight matrices and the hidden activations -- is split by truncation into x0 + x1 + x2 (bf16 each, the sum exact); the six
// products of order <= 2 run as v_mfma_f32_32x32x16_bf16 with fp32 accumulation (rowgemm_b3.hip has the error figures: below the fp32-MFMA kernel's).
// Per chunk of 32 hidden units and per wave (32 token rows): GEMM 1 computes the hidden tile TRANSPOSED (A = W1 rows, B = token rows), so a lane (row, half) ends
// up holding hidden units 8 q + 4 half + r of its own row -- which is exactly a B operand of GEMM 2 (A = W2 rows) once the chunk's W2 columns are stored in LDS in
// that order: ReLU and the three-term split happen in place in the accumulator registers and the fp32 kernel's transposition through LDS disappears.
// 96 MFMAs of 32 cycles per chunk and wave against 128 of 64; the weight chunks are split into planes while they are staged (double-buffered, one barrier per chunk).
#include <stdlib.h>

#include "lfsr_internal.h"

#ifndef FB_ABL
#define FB_ABL 0   // diagnostic timing builds (WRONG results; tools/build_abl.sh, tools/ffn_time.py): 1 token rows loaded in the first round only, 2 LayerNorm + split in the
#endif             // first round only, 4 no residual loads, 8 no stores, 16 no weight restaging after the first chunk
namespace {

typedef float f32x16c __attribute__((ext_vector_type(16)));
typedef unsigned u32x4c __attribute__((ext_vector_type(4)));

constexpr int FOOB3 = (int)0x80000000u;

struct FfnB3Args {
  const float* X; int x_stride; int x_choff;
  const float* W1;                              // [H][K1] fp32 (lfsr_pack_conv_weight, taps = 1)
  const float* W2;                              // [N2][H]
  const float* R; int r_stride; int r_choff;
  float* Y; int y_stride; int y_choff;
  long long M; int H;
  int x_bytes, y_bytes, r_bytes;
  float slope;
  const float* ln_g; const float* ln_b; float ln_eps;
  const unsigned short* Wsplit;                 // non-null: the chunks' LDS images (three planes of W1 and W2 per chunk, operand order) made once by k_ffn_presplit
};

__device__ __forceinline__ unsigned c3_hi_pair(unsigned hi_src, unsigned lo_src) { return __builtin_amdgcn_perm(hi_src, lo_src, 0x07060302u); }
__device__ __forceinline__ float c3_residual(float a) { return a - __uint_as_float(__float_as_uint(a) & 0xffff0000u); }
__device__ __forceinline__ void c3_split8(const float a[8], u32x4c& p0, u32x4c& p1, u32x4c& p2) {
#pragma unroll
  for (int j = 0; j < 4; ++j) { unsigned t0, t1, t2; lfsr_split_pair(a[2 * j], a[2 * j + 1], t0, t1, t2); p0[j] = t0; p1[j] = t1; p2[j] = t2; }
}
// asm MFMA, accumulator tied (rowgemm_b3.hip, b3_mfma: why not the builtin)
__device__ __forceinline__ void c3_mfma(f32x16c& c, const u32x4c a, const u32x4c b) {
  asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b));
}
// MFMA results -> VALU reads: the wait the compiler would insert for a builtin (tied to the accumulator so it stays between the two)
__device__ __forceinline__ void c3_settle(f32x16c& c) { asm volatile("s_nop 15\n\ts_nop 15" : "+v"(c)); }

template <int K1, int N2, bool PRE>
__global__ __launch_bounds__(512) void k_ffn_b3(FfnB3Args p) {
  constexpr int KS1 = K1 / 16, NT2 = N2 / 32;
  constexpr int R1H = K1 + 8, R2H = 40;                          // LDS row strides in bf16: rows start in distinct 16-B slots over 16 consecutive rows
  constexpr int P1 = 32 * R1H, P2 = N2 * R2H;                    // bf16 per plane
  constexpr int BUFH = 3 * (P1 + P2);                            // bf16 per weight buffer
  constexpr int W1G = 32 * K1 / 8, W2G = N2 * 4;                 // groups of 8 consecutive floats per chunk slice
  constexpr int W1L = W1G / 512 > 0 ? W1G / 512 : 1, W2L = W2G / 512 > 0 ? W2G / 512 : 1;
  static_assert(W1G % 512 == 0 || W1G < 512, "W1 slice");
  extern __shared__ __attribute__((aligned(16))) unsigned short sw[];      // [2][ W1 planes [3][32][R1H] | W2 planes [3][N2][R2H] ]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int half = lane >> 5, l31 = lane & 31;
  const int nch = p.H / 32;

  const long long gtot = (p.M + 31) / 32;
  const long long gbase = gtot / gridDim.x, grem = gtot % gridDim.x;
  const long long gn = gbase + ((long long)blockIdx.x < grem ? 1 : 0);
  const long long gstart = (long long)blockIdx.x * gbase + ((long long)blockIdx.x < grem ? (long long)blockIdx.x : grem);
  const int rounds = (int)((gn + 7) / 8);
  const int gpr = rounds ? (int)((gn + rounds - 1) / rounds) : 0;
  if (rounds == 0) return;

  // weight chunk c: fetched as fp32 into registers at the start of a chunk, split and stored into the other buffer at its end
  float4 w1r[PRE ? 1 : W1L][2], w2r[PRE ? 1 : W2L][2];
  auto fetch_chunk = [&](int c) {
    if constexpr (!PRE) {
#pragma unroll
    for (int i = 0; i < W1L; ++i) {
      const int idx = tid + 512 * i;
      if (idx < W1G) {
        const int r = idx / (K1 / 8), q = idx - r * (K1 / 8);
        const float* src = p.W1 + ((long long)c * 32 + r) * K1 + q * 8;
        w1r[i][0] = *reinterpret_cast<const float4*>(src); w1r[i][1] = *reinterpret_cast<const float4*>(src + 4);
      }
    }
#pragma unroll
    for (int i = 0; i < W2L; ++i) {
      const int idx = tid + 512 * i;
      if (idx < W2G) {
        const int n = idx >> 2, q = idx & 3;
        const float* src = p.W2 + (long long)n * p.H + c * 32 + q * 8;
        w2r[i][0] = *reinterpret_cast<const float4*>(src); w2r[i][1] = *reinterpret_cast<const float4*>(src + 4);
      }
    }
    }
  };
  auto store_chunk = [&](unsigned short* buf) {
    if constexpr (!PRE) {
#pragma unroll
    for (int i = 0; i < W1L; ++i) {
      const int idx = tid + 512 * i;
      if (idx < W1G) {
        const int r = idx / (K1 / 8), q = idx - r * (K1 / 8);
        const float a[8] = {w1r[i][0].x, w1r[i][0].y, w1r[i][0].z, w1r[i][0].w, w1r[i][1].x, w1r[i][1].y, w1r[i][1].z, w1r[i][1].w};
        u32x4c p0, p1, p2;
        c3_split8(a, p0, p1, p2);
        unsigned short* d = buf + r * R1H + q * 8;
        *reinterpret_cast<u32x4c*>(d) = p0; *reinterpret_cast<u32x4c*>(d + P1) = p1; *reinterpret_cast<u32x4c*>(d + 2 * P1) = p2;
      }
    }
    // W2 columns of the chunk in GEMM-2 operand order: slot (ks, kg, e, r) <- hidden unit 16 ks + 8 e + 4 kg + r.  A thread holds units 8 q .. 8 q + 7 of row n
    // (q = 2 ks + e): its two groups of four go to slots 16 ks + 8 kg + 4 e + (0..3), kg = 0 (units 8 q .. + 3) and kg = 1 (units 8 q + 4 .. + 7)
#pragma unroll
    for (int i = 0; i < W2L; ++i) {
      const int idx = tid + 512 * i;
      if (idx < W2G) {
        const int n = idx >> 2, q = idx & 3, ks = q >> 1, e = q & 1;
        const float a[8] = {w2r[i][0].x, w2r[i][0].y, w2r[i][0].z, w2r[i][0].w, w2r[i][1].x, w2r[i][1].y, w2r[i][1].z, w2r[i][1].w};
        u32x4c p0, p1, p2;
        c3_split8(a, p0, p1, p2);
        unsigned short* d = buf + 3 * P1 + n * R2H + 16 * ks + 4 * e;
        typedef unsigned u32x2c __attribute__((ext_vector_type(2)));
        *reinterpret_cast<u32x2c*>(d) = u32x2c{p0.x, p0.y};          *reinterpret_cast<u32x2c*>(d + 8) = u32x2c{p0.z, p0.w};
        *reinterpret_cast<u32x2c*>(d + P2) = u32x2c{p1.x, p1.y};     *reinterpret_cast<u32x2c*>(d + P2 + 8) = u32x2c{p1.z, p1.w};
        *reinterpret_cast<u32x2c*>(d + 2 * P2) = u32x2c{p2.x, p2.y}; *reinterpret_cast<u32x2c*>(d + 2 * P2 + 8) = u32x2c{p2.z, p2.w};
      }
    }
    }
  };

  // pre-split form: a chunk's LDS image is copied as it stands, 16 B per thread and piece, in two halves so that no more than four pieces are in flight
  // (the first half is stored after GEMM 1, the second at the chunk's end; the image of chunk c starts at c * BUFH)
  constexpr int IMG16 = BUFH / 8;                    // 16-B pieces per chunk image
  constexpr int NQ = (IMG16 + 511) / 512, NQH = (NQ + 1) / 2;
  u32x4c vq[PRE ? NQH : 1];
  constexpr bool pre = PRE;
  auto fetch_img = [&](int c, int half_i) {
    if constexpr (PRE)
#pragma unroll
    for (int i = 0; i < NQH; ++i) {
      const int piece = tid + 512 * (half_i * NQH + i);
      if (half_i * NQH + i < NQ && piece < IMG16) vq[i] = *reinterpret_cast<const u32x4c*>(p.Wsplit + ((long long)c * IMG16 + piece) * 8);
    }
  };
  auto store_img = [&](unsigned short* buf, int half_i) {
    if constexpr (PRE)
#pragma unroll
    for (int i = 0; i < NQH; ++i) {
      const int piece = tid + 512 * (half_i * NQH + i);
      if (half_i * NQH + i < NQ && piece < IMG16) *reinterpret_cast<u32x4c*>(buf + piece * 8) = vq[i];
    }
  };
  if (pre) { fetch_img(0, 0); store_img(sw, 0); fetch_img(0, 1); store_img(sw, 1); }
  else { fetch_chunk(0); store_chunk(sw); }
  // gamma | beta of the LayerNorm in LDS: read from global memory at the top of every round they were NEWER than the previous round's stores, and a wait for them
  // (vmcnt: in issue order, stores counted) was a wait for those stores to drain
  float* const sgb = reinterpret_cast<float*>(sw + 2 * BUFH);
  if (p.ln_g) for (int i = tid; i < 2 * K1; i += 512) sgb[i] = i < K1 ? p.ln_g[i] : p.ln_b[i - K1];
  __syncthreads();

  typedef float f32x4g __attribute__((ext_vector_type(4)));
  const __amdgpu_buffer_rsrc_t rsX = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.X), 0, p.x_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsY = __builtin_amdgcn_make_buffer_rsrc(p.Y, 0, p.y_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsR = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.R), 0, p.r_bytes, 0x00020000);   // null residual: loads return 0
  u32x4c x0[KS1], x1[KS1], x2[KS1];
  f32x16c accy[NT2];
  // The token rows of round rd + 1 are asked for at the START of round rd's epilogue, in front of its residual loads and stores: their latency overlaps the residual's, and
  // the next round does not begin by draining those stores (vmcnt counts stores, in issue order: a wait for loads issued BEHIND them would include them).  Ablation
  // (profiles/r03_logs/c17_ffn_abl.log, M = 819 200): rows loaded in the first round only -43 of 727 us, no residual loads -45, no stores -60 more.
  // Every wave runs the loads and the LayerNorm + split, with or without a group of rows (out-of-range offsets: zeros, no traffic): the raw rows and the planes are then
  // defined on every path, so their registers are free where the other is live (raw rows: epilogue -> split; planes: split -> last chunk).
  float xr[KS1][8];
  auto load_x = [&](int rdn) {
    const long long gq = gstart + (long long)rdn * gpr + wave;
    const bool act = rdn < rounds && wave < gpr && gq < gstart + gn;
    // (the group's base in the VGPR offset: the bounds check that turns rows past M into zeros covers the VGPR and immediate offsets only -- an SGPR offset is added unchecked)
    const int xo = (((int)(gq * 32) + l31) * p.x_stride + p.x_choff + 8 * half) * 4;
#pragma unroll
    for (int s = 0; s < KS1; ++s)
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        f32x4g v = {1.f, 2.f, 3.f, 4.f};
        if (!((FB_ABL & 1) && rdn > 0)) v = __builtin_bit_cast(f32x4g, __builtin_amdgcn_raw_buffer_load_b128(rsX, act ? xo + (16 * s + 4 * e) * 4 : FOOB3, 0, 0));
        xr[s][4 * e] = v.x; xr[s][4 * e + 1] = v.y; xr[s][4 * e + 2] = v.z; xr[s][4 * e + 3] = v.w;
      }
  };
#ifdef FB_STAG      // lab: blocks start (blockIdx & 7) * FB_STAG * 3.9 us apart -- are the rounds' memory bursts of all CUs in lockstep what the memory operations cost?
  for (int i = 0; i < (int)((blockIdx.x >> 3) & 7) * FB_STAG; ++i) __builtin_amdgcn_s_sleep(127);
#endif
  load_x(0);
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int i = 0; i < 4 * NT2; ++i)      // as many dropped stores as an epilogue issues behind its row loads: the first round meets the loop head in the same counter state
    __builtin_amdgcn_raw_buffer_store_b128(u32x4c{0u, 0u, 0u, 0u}, rsY, FOOB3 + 16 * i, 0, 0);      // (distinct addresses: identical stores are merged into one)
  int step = 0;
  for (int rd = 0; rd < rounds; ++rd) {
    const long long g = gstart + (long long)rd * gpr + wave;
    const bool active = wave < gpr && g < gstart + gn;
    const long long m0 = g * 32;
    if (!((FB_ABL & 2) && rd > 0)) {
      // lane (row l31, k-group half): eight consecutive channels per K step -- the B-operand order of the 32 x 32 x 16 MFMA
      if (p.ln_g) {     // LayerNorm in registers: a row lives in lanes l31 and l31 + 32
        float sm = 0.f;
#pragma unroll
        for (int s = 0; s < KS1; ++s)
#pragma unroll
          for (int j = 0; j < 8; ++j) sm += xr[s][j];
        sm += __shfl_xor(sm, 32);
        const float mu = sm * (1.0f / K1);
        float q2 = 0.f;
#pragma unroll
        for (int s = 0; s < KS1; ++s)
#pragma unroll
          for (int j = 0; j < 8; ++j) { xr[s][j] -= mu; q2 = fmaf(xr[s][j], xr[s][j], q2); }
        q2 += __shfl_xor(q2, 32);
        const float rstd = 1.0f / sqrtf(q2 * (1.0f / K1) + p.ln_eps);
#pragma unroll
        for (int s = 0; s < KS1; ++s)
#pragma unroll
          for (int e = 0; e < 2; ++e) {
            const float4 gv = *reinterpret_cast<const float4*>(sgb + 16 * s + 8 * half + 4 * e);
            const float4 bv = *reinterpret_cast<const float4*>(sgb + K1 + 16 * s + 8 * half + 4 * e);
            xr[s][4 * e] = xr[s][4 * e] * rstd * gv.x + bv.x; xr[s][4 * e + 1] = xr[s][4 * e + 1] * rstd * gv.y + bv.y;
            xr[s][4 * e + 2] = xr[s][4 * e + 2] * rstd * gv.z + bv.z; xr[s][4 * e + 3] = xr[s][4 * e + 3] * rstd * gv.w + bv.w;
          }
      }
#pragma unroll
      for (int s = 0; s < KS1; ++s) c3_split8(xr[s], x0[s], x1[s], x2[s]);
#pragma unroll
      for (int s = 0; s < KS1; ++s) asm volatile("s_nop 4" : "+v"(x0[s]), "+v"(x1[s]), "+v"(x2[s]));
    }
#pragma unroll
    for (int t = 0; t < NT2; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) accy[t][r] = 0.f;

    for (int c = 0; c < nch; ++c, ++step) {
      const unsigned short* buf = sw + (step & 1) * BUFH;
      const bool last_step = rd == rounds - 1 && c == nch - 1;
      const int cn = c + 1 < nch ? c + 1 : 0;
      unsigned short* const bufn = sw + ((step + 1) & 1) * BUFH;
      const bool restage = !last_step && !((FB_ABL & 16) && step > 0);
      if (restage) { if (pre) fetch_img(cn, 0); else fetch_chunk(cn); }   // flies under this chunk's MFMAs
      if (active) {
        // GEMM 1 (transposed): h[hidden][row], A = W1 rows of the chunk (lane = hidden unit l31, k-group half), B = the token planes
        f32x16c h;
#pragma unroll
        for (int r = 0; r < 16; ++r) h[r] = 0.f;
        asm volatile("s_nop 4" : "+v"(h));
        const unsigned short* a1 = buf + l31 * R1H + 8 * half;
#pragma unroll
        for (int s = 0; s < KS1; ++s) {
          const u32x4c w0 = *reinterpret_cast<const u32x4c*>(a1 + 16 * s);
          const u32x4c w1 = *reinterpret_cast<const u32x4c*>(a1 + 16 * s + P1);
          const u32x4c w2 = *reinterpret_cast<const u32x4c*>(a1 + 16 * s + 2 * P1);
          c3_mfma(h, w2, x0[s]); c3_mfma(h, w0, x2[s]); c3_mfma(h, w1, x1[s]);
          c3_mfma(h, w1, x0[s]); c3_mfma(h, w0, x1[s]); c3_mfma(h, w0, x0[s]);
        }
        c3_settle(h);
        if (pre && restage) { store_img(bufn, 0); fetch_img(cn, 1); }      // (wave-uniform; every wave of the block passes here or in the branch below)
        // activation + split in place: registers 8 ks .. 8 ks + 7 of the lane = hidden units 16 ks + 8 e + 4 half + r = the eight k slots of GEMM 2's step ks
        u32x4c h0[2], h1[2], h2[2];
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
          float hv[8];
#pragma unroll
          for (int j = 0; j < 8; ++j) { const float v = h[8 * ks + j]; hv[j] = v >= 0.f ? v : v * p.slope; }
          c3_split8(hv, h0[ks], h1[ks], h2[ks]);
        }
        asm volatile("s_nop 4" : "+v"(h0[0]), "+v"(h0[1]), "+v"(h1[0]), "+v"(h1[1]), "+v"(h2[0]), "+v"(h2[1]));   // every plane tied: no split instruction may sink behind the wait
        // GEMM 2 (transposed): y[n][row] += W2[n][hidden chunk] h[hidden][row]
        const unsigned short* a2 = buf + 3 * P1 + l31 * R2H + 8 * half;
#pragma unroll
        for (int t = 0; t < NT2; ++t) {
#pragma unroll
          for (int ks = 0; ks < 2; ++ks) {
            const unsigned short* aq = a2 + t * 32 * R2H + 16 * ks;
            const u32x4c w0 = *reinterpret_cast<const u32x4c*>(aq);
            const u32x4c w1 = *reinterpret_cast<const u32x4c*>(aq + P2);
            const u32x4c w2 = *reinterpret_cast<const u32x4c*>(aq + 2 * P2);
            c3_mfma(accy[t], w2, h0[ks]); c3_mfma(accy[t], w0, h2[ks]); c3_mfma(accy[t], w1, h1[ks]);
            c3_mfma(accy[t], w1, h0[ks]); c3_mfma(accy[t], w0, h1[ks]); c3_mfma(accy[t], w0, h0[ks]);
          }
        }
      }
      if (restage) {
        if (pre) { if (!active) { store_img(bufn, 0); fetch_img(cn, 1); } store_img(bufn, 1); }
        else store_chunk(bufn);
      }
      __syncthreads();
    }
    // epilogue: lane (row l31, half) holds channels 32 t + 8 q + 4 half + r of its row: 16-B residual loads and stores through buffer descriptors
    load_x(rd + 1);
    __builtin_amdgcn_sched_barrier(0);      // (the scheduler otherwise sinks these loads in between the stores below -- and a wait for them becomes a wait for stores again)
    {     // (every wave, with every offset out of range for a wave without a group: one instruction stream, so the compiler can count the waits at the round's top)
#pragma unroll
      for (int t = 0; t < NT2; ++t) c3_settle(accy[t]);
      const int m0i = active ? (int)m0 : 0, Mi = (int)p.M;
      const bool ok = active && m0i + l31 < Mi;
      const int yv = (l31 * p.y_stride + p.y_choff + 4 * half) * 4, rvo = (l31 * p.r_stride + p.r_choff + 4 * half) * 4;
      const int ys = __builtin_amdgcn_readfirstlane(m0i * p.y_stride * 4), rs = __builtin_amdgcn_readfirstlane(m0i * p.r_stride * 4);
      // every residual load of the group before the first store: a wait for a load issued behind a store would include the store's drain (one such chain per 32 columns before)
      f32x4g rv[NT2][4];
#pragma unroll
      for (int t = 0; t < NT2; ++t)
#pragma unroll
        for (int q = 0; q < 4; ++q)
          rv[t][q] = (FB_ABL & 4) ? f32x4g{0.f, 0.f, 0.f, 0.f} : __builtin_bit_cast(f32x4g, __builtin_amdgcn_raw_buffer_load_b128(rsR, ok ? rvo + (32 * t + 8 * q) * 4 : FOOB3, rs, 0));
#pragma unroll
      for (int t = 0; t < NT2; ++t)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const f32x4g o = {accy[t][4 * q] + rv[t][q].x, accy[t][4 * q + 1] + rv[t][q].y, accy[t][4 * q + 2] + rv[t][q].z, accy[t][4 * q + 3] + rv[t][q].w};
          if (!(FB_ABL & 8) || o.x == 123.456f) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4c, o), rsY, ok ? yv + (32 * t + 8 * q) * 4 : FOOB3, ys, 0);
        }
    }
  }
}

// The same block with EVERY chunk's weight planes resident in LDS (K1 = N2 = 64, H = 128: four chunk images of 29 KB): nothing is restaged, so the round loop needs no barrier
// at all and the eight waves run their own row groups independently -- their loads, LayerNorm / split, MFMAs and stores drift apart and overlap, where the chunk-barrier form
// has all eight change phase together (LFT's angular transformer: 241 us per launch with a matrix-pipe time of 76 us and a memory time of 110-130 us).
// Needs the pre-split image (lfsr_ffn_b3_presplit); straight-line loop body with buffer descriptors (counted waits), the next group's rows asked for behind the split.
template <int K1, int N2, int NCH>
__global__ __launch_bounds__(512) void k_ffn_b3_res(FfnB3Args p) {
  constexpr int KS1 = K1 / 16, NT2 = N2 / 32;
  constexpr int R1H = K1 + 8, R2H = 40;
  constexpr int P1 = 32 * R1H, P2 = N2 * R2H;
  constexpr int BUFH = 3 * (P1 + P2);
  extern __shared__ __attribute__((aligned(16))) unsigned short sw[];      // [NCH][ W1 planes [3][32][R1H] | W2 planes [3][N2][R2H] ], then gamma | beta
  float* const sgb = reinterpret_cast<float*>(sw + NCH * BUFH);
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int half = lane >> 5, l31 = lane & 31;
  for (int i = tid; i < NCH * BUFH / 8; i += 512) *reinterpret_cast<u32x4c*>(sw + i * 8) = *reinterpret_cast<const u32x4c*>(p.Wsplit + (long long)i * 8);
  if (p.ln_g) for (int i = tid; i < 2 * K1; i += 512) sgb[i] = i < K1 ? p.ln_g[i] : p.ln_b[i - K1];
  __syncthreads();

  // this block's row groups (32 rows each): [gstart, gend); wave w takes gstart + w, + 8, ...
  const long long gtot = (p.M + 31) / 32;
  const long long gbase = gtot / gridDim.x, grem = gtot % gridDim.x;
  const int gn = (int)(gbase + ((long long)blockIdx.x < grem ? 1 : 0));
  const int gstart = (int)((long long)blockIdx.x * gbase + ((long long)blockIdx.x < grem ? (long long)blockIdx.x : grem));
  const int gend = gstart + gn;
  typedef float f32x4g __attribute__((ext_vector_type(4)));
  const __amdgpu_buffer_rsrc_t rsX = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.X), 0, p.x_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsY = __builtin_amdgcn_make_buffer_rsrc(p.Y, 0, p.y_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsR = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.R), 0, p.r_bytes, 0x00020000);   // null residual: loads return 0
  float xr[KS1][8];
  auto load_x = [&](int g) __attribute__((always_inline)) {      // (a group past the block's last: out-of-range offsets, zeros without traffic; the group's base sits in the VGPR offset)
    const int xo = g < gend ? ((g * 32 + l31) * p.x_stride + p.x_choff + 8 * half) * 4 : FOOB3;
#pragma unroll
    for (int s = 0; s < KS1; ++s)
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        const f32x4g v = __builtin_bit_cast(f32x4g, __builtin_amdgcn_raw_buffer_load_b128(rsX, xo == FOOB3 ? FOOB3 : xo + (16 * s + 4 * e) * 4, 0, 0));
        xr[s][4 * e] = v.x; xr[s][4 * e + 1] = v.y; xr[s][4 * e + 2] = v.z; xr[s][4 * e + 3] = v.w;
      }
  };
  int g = gstart + wave;
  load_x(g);
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int i = 0; i < 4 * NT2; ++i)      // as many dropped stores (distinct addresses) as an epilogue issues behind its row loads: the first group meets the loop head in the same counter state
    __builtin_amdgcn_raw_buffer_store_b128(u32x4c{0u, 0u, 0u, 0u}, rsY, FOOB3 + 16 * i, 0, 0);
  for (; g < gend; g += 8) {
    u32x4c x0[KS1], x1[KS1], x2[KS1];
    if (p.ln_g) {     // LayerNorm in registers: a row lives in lanes l31 and l31 + 32 (the arithmetic of k_ffn_b3)
      float sm = 0.f;
#pragma unroll
      for (int s = 0; s < KS1; ++s)
#pragma unroll
        for (int j = 0; j < 8; ++j) sm += xr[s][j];
      sm += __shfl_xor(sm, 32);
      const float mu = sm * (1.0f / K1);
      float q2 = 0.f;
#pragma unroll
      for (int s = 0; s < KS1; ++s)
#pragma unroll
        for (int j = 0; j < 8; ++j) { xr[s][j] -= mu; q2 = fmaf(xr[s][j], xr[s][j], q2); }
      q2 += __shfl_xor(q2, 32);
      const float rstd = 1.0f / sqrtf(q2 * (1.0f / K1) + p.ln_eps);
#pragma unroll
      for (int s = 0; s < KS1; ++s)
#pragma unroll
        for (int e = 0; e < 2; ++e) {
          const float4 gv = *reinterpret_cast<const float4*>(sgb + 16 * s + 8 * half + 4 * e);
          const float4 bv = *reinterpret_cast<const float4*>(sgb + K1 + 16 * s + 8 * half + 4 * e);
          xr[s][4 * e] = xr[s][4 * e] * rstd * gv.x + bv.x; xr[s][4 * e + 1] = xr[s][4 * e + 1] * rstd * gv.y + bv.y;
          xr[s][4 * e + 2] = xr[s][4 * e + 2] * rstd * gv.z + bv.z; xr[s][4 * e + 3] = xr[s][4 * e + 3] * rstd * gv.w + bv.w;
        }
    }
#pragma unroll
    for (int s = 0; s < KS1; ++s) c3_split8(xr[s], x0[s], x1[s], x2[s]);
#pragma unroll
    for (int s = 0; s < KS1; ++s) asm volatile("s_nop 4" : "+v"(x0[s]), "+v"(x1[s]), "+v"(x2[s]));
    load_x(g + 8);                                               // the wave's next group: in flight during this one's GEMMs
    __builtin_amdgcn_sched_barrier(0);
    f32x16c accy[NT2];
#pragma unroll
    for (int t = 0; t < NT2; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) accy[t][r] = 0.f;
#pragma unroll
    for (int t = 0; t < NT2; ++t) asm volatile("s_nop 4" : "+v"(accy[t]));
#pragma unroll 1
    for (int c = 0; c < NCH; ++c) {
      const unsigned short* buf = sw + c * BUFH;
      // GEMM 1 (transposed): h[hidden][row], A = W1 rows of the chunk (lane = hidden unit l31, k-group half), B = the token planes
      f32x16c h;
#pragma unroll
      for (int r = 0; r < 16; ++r) h[r] = 0.f;
      asm volatile("s_nop 4" : "+v"(h));
      const unsigned short* a1 = buf + l31 * R1H + 8 * half;
#pragma unroll
      for (int s = 0; s < KS1; ++s) {
        const u32x4c w0 = *reinterpret_cast<const u32x4c*>(a1 + 16 * s);
        const u32x4c w1 = *reinterpret_cast<const u32x4c*>(a1 + 16 * s + P1);
        const u32x4c w2 = *reinterpret_cast<const u32x4c*>(a1 + 16 * s + 2 * P1);
        c3_mfma(h, w2, x0[s]); c3_mfma(h, w0, x2[s]); c3_mfma(h, w1, x1[s]);
        c3_mfma(h, w1, x0[s]); c3_mfma(h, w0, x1[s]); c3_mfma(h, w0, x0[s]);
      }
      c3_settle(h);
      // activation + split in place: registers 8 ks .. 8 ks + 7 of the lane = hidden units 16 ks + 8 e + 4 half + r = the eight k slots of GEMM 2's step ks
      u32x4c h0[2], h1[2], h2[2];
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        float hv[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) { const float v = h[8 * ks + j]; hv[j] = v >= 0.f ? v : v * p.slope; }
        c3_split8(hv, h0[ks], h1[ks], h2[ks]);
      }
      asm volatile("s_nop 4" : "+v"(h0[0]), "+v"(h0[1]), "+v"(h1[0]), "+v"(h1[1]), "+v"(h2[0]), "+v"(h2[1]));
      // GEMM 2 (transposed): y[n][row] += W2[n][hidden chunk] h[hidden][row]
      const unsigned short* a2 = buf + 3 * P1 + l31 * R2H + 8 * half;
#pragma unroll
      for (int t = 0; t < NT2; ++t) {
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
          const unsigned short* aq = a2 + t * 32 * R2H + 16 * ks;
          const u32x4c w0 = *reinterpret_cast<const u32x4c*>(aq);
          const u32x4c w1 = *reinterpret_cast<const u32x4c*>(aq + P2);
          const u32x4c w2 = *reinterpret_cast<const u32x4c*>(aq + 2 * P2);
          c3_mfma(accy[t], w2, h0[ks]); c3_mfma(accy[t], w0, h2[ks]); c3_mfma(accy[t], w1, h1[ks]);
          c3_mfma(accy[t], w1, h0[ks]); c3_mfma(accy[t], w0, h1[ks]); c3_mfma(accy[t], w0, h0[ks]);
        }
      }
    }
    // epilogue: lane (row l31, half) holds channels 32 t + 8 q + 4 half + r of its row; every residual load in front of the first store
#pragma unroll
    for (int t = 0; t < NT2; ++t) c3_settle(accy[t]);
    const int m0i = g * 32, Mi = (int)p.M;
    const bool ok = m0i + l31 < Mi;
    const int yv = ((m0i + l31) * p.y_stride + p.y_choff + 4 * half) * 4, rvo = ((m0i + l31) * p.r_stride + p.r_choff + 4 * half) * 4;
    f32x4g rv[NT2][4];
#pragma unroll
    for (int t = 0; t < NT2; ++t)
#pragma unroll
      for (int q = 0; q < 4; ++q) rv[t][q] = __builtin_bit_cast(f32x4g, __builtin_amdgcn_raw_buffer_load_b128(rsR, ok ? rvo + (32 * t + 8 * q) * 4 : FOOB3, 0, 0));
#pragma unroll
    for (int t = 0; t < NT2; ++t)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const f32x4g o = {accy[t][4 * q] + rv[t][q].x, accy[t][4 * q + 1] + rv[t][q].y, accy[t][4 * q + 2] + rv[t][q].z, accy[t][4 * q + 3] + rv[t][q].w};
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4c, o), rsY, ok ? yv + (32 * t + 8 * q) * 4 : FOOB3, 0, 0);
      }
  }
}

// one block per chunk: the chunk's LDS image (as store_chunk builds it) written to global memory once per weight set
template <int K1, int N2>
__global__ __launch_bounds__(512) void k_ffn_presplit(const float* __restrict__ W1, const float* __restrict__ W2, int H, unsigned short* __restrict__ out) {
  constexpr int R1H = K1 + 8, R2H = 40, P1 = 32 * R1H, P2 = N2 * R2H, BUFH = 3 * (P1 + P2);
  constexpr int W1G = 32 * K1 / 8, W2G = N2 * 4;
  const int c = blockIdx.x, tid = threadIdx.x;
  unsigned short* buf = out + (long long)c * BUFH;
  for (int i = tid; i < BUFH / 8; i += 512) *reinterpret_cast<u32x4c*>(buf + i * 8) = u32x4c{0u, 0u, 0u, 0u};    // (padding halves defined)
  __syncthreads();
  for (int idx = tid; idx < W1G; idx += 512) {
    const int r = idx / (K1 / 8), q = idx - r * (K1 / 8);
    const float* src = W1 + ((long long)c * 32 + r) * K1 + q * 8;
    const float4 lo = *reinterpret_cast<const float4*>(src), hi = *reinterpret_cast<const float4*>(src + 4);
    const float a[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
    u32x4c p0, p1, p2;
    c3_split8(a, p0, p1, p2);
    unsigned short* d = buf + r * R1H + q * 8;
    *reinterpret_cast<u32x4c*>(d) = p0; *reinterpret_cast<u32x4c*>(d + P1) = p1; *reinterpret_cast<u32x4c*>(d + 2 * P1) = p2;
  }
  for (int idx = tid; idx < W2G; idx += 512) {
    const int n = idx >> 2, q = idx & 3, ks = q >> 1, e = q & 1;
    const float* src = W2 + (long long)n * H + c * 32 + q * 8;
    const float4 lo = *reinterpret_cast<const float4*>(src), hi = *reinterpret_cast<const float4*>(src + 4);
    const float a[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
    u32x4c p0, p1, p2;
    c3_split8(a, p0, p1, p2);
    unsigned short* d = buf + 3 * P1 + n * R2H + 16 * ks + 4 * e;
    typedef unsigned u32x2c __attribute__((ext_vector_type(2)));
    *reinterpret_cast<u32x2c*>(d) = u32x2c{p0.x, p0.y};          *reinterpret_cast<u32x2c*>(d + 8) = u32x2c{p0.z, p0.w};
    *reinterpret_cast<u32x2c*>(d + P2) = u32x2c{p1.x, p1.y};     *reinterpret_cast<u32x2c*>(d + P2 + 8) = u32x2c{p1.z, p1.w};
    *reinterpret_cast<u32x2c*>(d + 2 * P2) = u32x2c{p2.x, p2.y}; *reinterpret_cast<u32x2c*>(d + 2 * P2 + 8) = u32x2c{p2.z, p2.w};
  }
}

template <int K1, int N2, bool PRE>
int launch_ffn_b3(const FfnB3Args& p, hipStream_t st) {
  constexpr int smem = 2 * 3 * (32 * (K1 + 8) + N2 * 40) * 2 + 2 * K1 * 4;      // two weight buffers + gamma | beta
  static std::atomic<bool> attr_set[64];
  static std::atomic<int> cus[64];
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return LFSR_E_ARG;
  if (!attr_set[dev]) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_ffn_b3<K1, N2, PRE>), hipFuncAttributeMaxDynamicSharedMemorySize, smem);
    if (e != hipSuccess) return LFSR_HIP_ERR(e);
    int v = 0;
    cus[dev] = (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) ? v : 256;
    attr_set[dev] = true;
  }
  const long long groups = (p.M + 31) / 32;
  long long grid = cus[dev];
  if (grid > (groups + 7) / 8) grid = (groups + 7) / 8;
  hipLaunchKernelGGL((k_ffn_b3<K1, N2, PRE>), dim3((unsigned)grid), dim3(512), smem, st, p);
  LFSR_CHECK_LAUNCH();
  return LFSR_OK;
}

template <int K1, int N2, int NCH>
int launch_ffn_b3_res(const FfnB3Args& p, hipStream_t st) {
  constexpr int smem = NCH * 3 * (32 * (K1 + 8) + N2 * 40) * 2 + 2 * K1 * 4;
  static_assert(smem <= 160 * 1024, "all chunks resident");
  static std::atomic<bool> attr_set[64];
  static std::atomic<int> cus[64];
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return LFSR_E_ARG;
  if (!attr_set[dev]) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_ffn_b3_res<K1, N2, NCH>), hipFuncAttributeMaxDynamicSharedMemorySize, smem);
    if (e != hipSuccess) return LFSR_HIP_ERR(e);
    int v = 0;
    cus[dev] = (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) ? v : 256;
    attr_set[dev] = true;
  }
  const long long groups = (p.M + 31) / 32;
  long long grid = cus[dev];
  if (grid > (groups + 7) / 8) grid = (groups + 7) / 8;
  hipLaunchKernelGGL((k_ffn_b3_res<K1, N2, NCH>), dim3((unsigned)grid), dim3(512), smem, st, p);
  LFSR_CHECK_LAUNCH();
  return LFSR_OK;
}

}  // namespace

// LFSR_E_ARG = shape not covered (the caller runs the fp32-MFMA kernel)
int lfsr_ffn_b3_launch(const float* x, int x_stride, int x_choff, const float* ln_g, const float* ln_b, float ln_eps, const float* w1_packed, const float* w2_packed,
                       const float* res, int res_stride, int res_choff, float* y, int y_stride, int y_choff,
                       long long M, int K1, int H, int N2, float slope, hipStream_t st, const void* wsplit) {
  if (!x || !w1_packed || !w2_packed || !y || M <= 0 || H <= 0 || H % 32 || ((uintptr_t)wsplit & 15)) return LFSR_E_ARG;
  if ((x_stride | x_choff | y_stride | y_choff) & 3 || (res && ((res_stride | res_choff) & 3))) return LFSR_E_ARG;
  if (((uintptr_t)x | (uintptr_t)y | (uintptr_t)res | (uintptr_t)w1_packed | (uintptr_t)w2_packed | (uintptr_t)ln_g | (uintptr_t)ln_b) & 15) return LFSR_E_ARG;
  if (x_stride < x_choff + K1 || y_stride < y_choff + N2 || (res && res_stride < res_choff + N2)) return LFSR_E_ARG;
  const long long span = (long long)(x_stride > y_stride ? (x_stride > res_stride ? x_stride : res_stride) : (y_stride > res_stride ? y_stride : res_stride));
  if ((M + 64) * span * 4 >= (1LL << 31)) return LFSR_E_ARG;      // (32-bit byte offsets, a last partial group of rows included)
  FfnB3Args p{};
  p.X = x; p.x_stride = x_stride; p.x_choff = x_choff; p.W1 = w1_packed; p.W2 = w2_packed;
  p.R = res; p.r_stride = res_stride; p.r_choff = res_choff; p.Y = y; p.y_stride = y_stride; p.y_choff = y_choff;
  p.M = M; p.H = H; p.slope = slope; p.ln_g = ln_g; p.ln_b = ln_b; p.ln_eps = ln_eps; p.Wsplit = (const unsigned short*)wsplit;
  p.x_bytes = (int)(M * x_stride * 4); p.y_bytes = (int)(M * y_stride * 4); p.r_bytes = res ? (int)(M * res_stride * 4) : 0;
  if (K1 == 128 && N2 == 128) return wsplit ? launch_ffn_b3<128, 128, true>(p, st) : launch_ffn_b3<128, 128, false>(p, st);
  if (K1 == 64 && N2 == 64) {
    // H = 128 with the pre-split image: all four chunks resident in LDS, no barrier in the loop (LFSR_FFN=chunks: the chunk-staging form)
    const char* fsel = lfsr_sel("LFSR_FFN");
    if (wsplit && H == 128 && !(fsel && fsel[0] == 'c')) return launch_ffn_b3_res<64, 64, 4>(p, st);
    return wsplit ? launch_ffn_b3<64, 64, true>(p, st) : launch_ffn_b3<64, 64, false>(p, st);
  }
  return LFSR_E_ARG;
}

// bytes of / fill the pre-split weight image of one feed-forward block (0 / LFSR_E_ARG: shape not covered)
size_t lfsr_ffn_b3_presplit_bytes(int K1, int H, int N2) {
  if (H <= 0 || H % 32) return 0;
  if (K1 == 128 && N2 == 128) return (size_t)(H / 32) * 3 * (32 * (128 + 8) + 128 * 40) * 2;
  if (K1 == 64 && N2 == 64) return (size_t)(H / 32) * 3 * (32 * (64 + 8) + 64 * 40) * 2;
  return 0;
}
int lfsr_ffn_b3_presplit(const float* w1_packed, const float* w2_packed, int K1, int H, int N2, void* out, hipStream_t st) {
  if (!w1_packed || !w2_packed || !out || ((uintptr_t)out & 15) || !lfsr_ffn_b3_presplit_bytes(K1, H, N2)) return LFSR_E_ARG;
  if (K1 == 128) hipLaunchKernelGGL((k_ffn_presplit<128, 128>), dim3((unsigned)(H / 32)), dim3(512), 0, st, w1_packed, w2_packed, H, (unsigned short*)out);
  else hipLaunchKernelGGL((k_ffn_presplit<64, 64>), dim3((unsigned)(H / 32)), dim3(512), 0, st, w1_packed, w2_packed, H, (unsigned short*)out);
  LFSR_CHECK_LAUNCH();
  return LFSR_OK;
}
