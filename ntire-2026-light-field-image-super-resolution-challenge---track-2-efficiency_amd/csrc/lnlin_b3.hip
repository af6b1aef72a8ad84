// LayerNorm + attention in-projection (q | k from LayerNorm(x + pe), v from x: EPIT.py:113-121, LFT.py:190-197) with the WEIGHTS in registers and the token rows through LDS,
// on the bf16 MFMA pipe with fp32 operands carried exactly as three bf16 terms (the arithmetic of rowgemm_b3.hip: six products of order <= 2 per K step, fp32 accumulation).
//
// Why a second form.  k_rowgemm_b3<128, true, 128> keeps a 128-column weight panel in LDS (98 KB of the 160: the three planes of 256 columns would be 196 KB), so N = 384 is three
// panels, each of which loads, norms and splits the token rows again, and every 16 x 16 x 32 MFMA reads its weight operand from LDS -- 512 B per MFMA, the LDS port saturates
// exactly when the matrix pipe does.  Measured at the LFT geometry (M = 819 200; profiles/r03_logs/c15_lin_abl.log): arithmetic alone 426 us against a matrix-pipe time of 219 us.
// Here the roles are swapped: a wave owns NTW = 3 column tiles (48 of the 384 columns) and keeps their three planes in 144 registers for the whole launch; the block norms and
// splits a stage of 32 token rows ONCE (every thread eight values of one row: a row is the 16 lanes of a DPP row, the reductions are four DPP adds) into two plane sets in LDS
// (LayerNorm'd for the q | k tiles, raw for the v tiles), and all eight waves read them as B operands: 3 KB of LDS reads feed 18 MFMAs (170 B per MFMA), the rows come from HBM
// once.  Stages are double-buffered (2 x 48 KB) with one barrier per stage; the rows of stage i + 2 are in flight while stage i + 1 is split and stage i multiplied.
//
// LDS image of a plane: [row 32][slot 16][8 bf16], slot j = 8-value k-group j of the row stored at (j ^ (row & 15)): the producers' 16-B writes (16 lanes = one row = 256
// contiguous bytes, permuted) and the consumers' B-operand reads (16 lanes = 16 rows at one k-group: 16 different slots) both cover a 256-B bank row once.
#include <stdlib.h>
#include <type_traits>

#include "lfsr_internal.h"

#ifndef LL_PHASE
#define LL_PHASE 0   // 1: waves 4..7 multiply first and norm / split afterwards (waves 0..3 the other way round)
#endif
#ifndef LL_STPOL
#define LL_STPOL 0   // cache-policy bits of the output stores (2 = nt)
#endif
#ifndef LL_ABL
#define LL_ABL 0   // diagnostic timing builds (WRONG results; tools/build_abl.sh): 1 no LayerNorm / split (planes written once), 2 no MFMAs, 4 no stores, 8 no row loads after the prologue
#endif

namespace {

typedef float f32x4l __attribute__((ext_vector_type(4)));
typedef unsigned u32x4l __attribute__((ext_vector_type(4)));

struct LnLinArgs {
  const float* X; int x_stride; int x_choff;
  const float* Wp;       // [N rows][128] fp32 (packed, k contiguous)
  float* Y; int y_stride; int y_choff;
  float* Y2; int y2_stride; int y2_choff; int split_n;      // columns n >= split_n go to Y2 (column n - split_n); Y2 == nullptr: everything to Y
  long long M; int N;
  const float* ln_g; const float* ln_b; float ln_eps; int ln_cols;
  const float* pe; int pe_stride; int pe_rows; int pe_div;
  int nstages;
};

__device__ __forceinline__ void ll_mfma(f32x4l& c, const u32x4l a, const u32x4l b) {      // accumulator tied (rowgemm_b3.hip, b3_mfma: why not the builtin)
  asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b));
}
__device__ __forceinline__ void ll_split8(const float (&a)[8], u32x4l& p0, u32x4l& p1, u32x4l& p2) {
#pragma unroll
  for (int j = 0; j < 4; ++j) { unsigned t0, t1, t2; lfsr_split_pair(a[2 * j], a[2 * j + 1], t0, t1, t2); p0[j] = t0; p1[j] = t1; p2[j] = t2; }
}
// sum over the NG (16 or 8) lanes of a token row inside a DPP row, the same bits in every lane (each level adds a pair symmetrically)
template <int NG>
__device__ __forceinline__ float ll_row_sum(float v) {
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xf, 0xf, true));    // quad_perm [1, 0, 3, 2]
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xf, 0xf, true));    // quad_perm [2, 3, 0, 1]
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xf, 0xf, true));   // row_half_mirror
  if (NG == 16) v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xf, 0xf, true));   // row_mirror
  return v;
}

constexpr int LL_ROWS = 32;
// K = 128: 512 threads = 8 waves, a row = 16 lanes (k-groups) = a DPP row, N = 8 x 3 x 16 = 384.  K = 64 (LFT's angular transformer, LFT.py:236-241): 256 threads = 4 waves,
// a row = 8 lanes, N = 4 x 3 x 16 = 192; the slot swizzle takes (row >> 1) & 7 (two rows fill a 256-B bank row)
template <int K> struct LL {
  static constexpr int KS = K / 32, NG = K / 8, NTH = LL_ROWS * NG, NW = NTH / 64;
  static constexpr int PLANE = LL_ROWS * NG * 8;       // bf16 per plane
  static constexpr int SET = 3 * PLANE, BUF = 2 * SET;
  static constexpr int SMEM = 2 * BUF * 2 + 2 * K * 4; // two stages x two sets x three planes + gamma | beta
  __device__ static __forceinline__ int swz(int row) { return K == 128 ? (row & 15) : ((row >> 1) & 7); }
};

// PE (compile time): a positional encoding is added in front of the LayerNorm -- as a run-time branch around two loads it cost the loop its counted waits
// (the compiler merges the two paths' counters conservatively: the head of the loop waited for stores)
template <int K, int NTW, bool PE>
__global__ __launch_bounds__(LL<K>::NTH) void k_lnlin_b3(LnLinArgs p) {
  constexpr int LL_K = K, LL_KS = LL<K>::KS, LL_NG = LL<K>::NG, LL_NTH = LL<K>::NTH, LL_PLANE = LL<K>::PLANE, LL_SET = LL<K>::SET, LL_BUF = LL<K>::BUF;
  extern __shared__ __attribute__((aligned(16))) unsigned short sll[];
  float* const sgb = reinterpret_cast<float*>(sll + 2 * LL_BUF);
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);      // (in an SGPR: what depends on the wave alone branches as a scalar)
  const int l15 = lane & 15, g = lane >> 4;

  // ---- this wave's weight planes: column tile ct covers channels 16 (NTW wave + ct) .. + 15; A-operand order (lane = channel l15, k-group g), all K steps: 36 x 4 registers
  u32x4l wf[NTW][LL_KS][3];
#pragma unroll
  for (int ct = 0; ct < NTW; ++ct) {
    const int n = 16 * (NTW * wave + ct) + l15;
#pragma unroll
    for (int s = 0; s < LL_KS; ++s) {
      const float* src = p.Wp + (long long)n * LL_K + 32 * s + 8 * g;
      const float4 lo = *reinterpret_cast<const float4*>(src), hi = *reinterpret_cast<const float4*>(src + 4);
      const float a[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
      ll_split8(a, wf[ct][s][0], wf[ct][s][1], wf[ct][s][2]);
    }
  }
#pragma unroll
  for (int ct = 0; ct < NTW; ++ct)
#pragma unroll
    for (int s = 0; s < LL_KS; ++s) asm volatile("s_nop 4" : "+v"(wf[ct][s][0]), "+v"(wf[ct][s][1]), "+v"(wf[ct][s][2]));     // VALU write -> asm MFMA read
  for (int i = tid; i < 2 * LL_K; i += LL_NTH) sgb[i] = i < LL_K ? p.ln_g[i] : p.ln_b[i - LL_K];

  // ---- producer role: thread = (row pr of the stage, k-group pj of eight values); a row = the 16 lanes of a DPP row
  const int pr = tid / LL_NG, pj = tid % LL_NG;
  typedef float f32x4g __attribute__((ext_vector_type(4)));
  const __amdgpu_buffer_rsrc_t rsX = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.X), 0, (int)(p.M * p.x_stride * 4), 0x00020000);
  const int xoff = (pr * p.x_stride + p.x_choff + 8 * pj) * 4;
  struct Raw { f32x4g x0, x1, e0, e1; };
  // the positional-encoding row of this thread's token, (m / pe_div) % pe_rows, kept by stepping: the block's stages are gridDim.x apart, i.e. m advances by a constant
  // per stage -- no integer division (a 64-bit one is a branchy call) in the loop
  const unsigned dm = (unsigned)(LL_ROWS * gridDim.x);
  const unsigned pdiv = PE ? (unsigned)p.pe_div : 1u, prows = PE ? (unsigned)p.pe_rows : 1u;
  const unsigned d_q = dm / pdiv, d_r = dm - d_q * pdiv, d_qm = d_q % prows;      // (block-uniform)
  unsigned pe_r, pe_row;                                                            // m % pe_div and (m / pe_div) % pe_rows of the NEXT stage to be loaded
  {
    const unsigned m = (unsigned)(blockIdx.x * LL_ROWS + pr);
    const unsigned q = m / pdiv;
    pe_r = m - q * pdiv; pe_row = q % prows;
  }
  long long st_load = blockIdx.x;                                                   // the next stage to be loaded
  auto load_rows = [&]() __attribute__((always_inline)) -> Raw {        // (possibly past the end: rows >= M come back as zeros without traffic)
    Raw r;
    const long long stc = st_load < p.nstages ? st_load : p.nstages;
    const int o = xoff + (int)(stc * LL_ROWS) * p.x_stride * 4;      // (the stage's base in the VGPR offset: the bounds check does not cover an SGPR offset)
    r.x0 = __builtin_bit_cast(f32x4g, __builtin_amdgcn_raw_buffer_load_b128(rsX, o, 0, 0));
    r.x1 = __builtin_bit_cast(f32x4g, __builtin_amdgcn_raw_buffer_load_b128(rsX, o + 16, 0, 0));
    r.e0 = f32x4g{0.f, 0.f, 0.f, 0.f}; r.e1 = r.e0;
    if constexpr (PE) {
      const float* pp = p.pe + (long long)pe_row * p.pe_stride + 8 * pj;
      r.e0 = *reinterpret_cast<const f32x4g*>(pp); r.e1 = *reinterpret_cast<const f32x4g*>(pp + 4);
      pe_r += d_r;
      const unsigned c = pe_r >= pdiv ? 1u : 0u;
      pe_r -= c ? pdiv : 0u;
      pe_row += d_qm + c;
      pe_row -= pe_row >= prows ? prows : 0u;
    }
    st_load += gridDim.x;
    return r;
  };
  unsigned short* const pdst = sll + (pr * LL_NG + (pj ^ LL<K>::swz(pr))) * 8;      // this thread's slot in a plane
  auto produce = [&](const Raw& r, int buf) __attribute__((always_inline)) {
    unsigned short* d = pdst + buf * LL_BUF;
    const float a[8] = {r.x0.x, r.x0.y, r.x0.z, r.x0.w, r.x1.x, r.x1.y, r.x1.z, r.x1.w};
    u32x4l q0, q1, q2;
    ll_split8(a, q0, q1, q2);                                     // the raw rows (v's operand)
    *reinterpret_cast<u32x4l*>(d + LL_SET) = q0; *reinterpret_cast<u32x4l*>(d + LL_SET + LL_PLANE) = q1; *reinterpret_cast<u32x4l*>(d + LL_SET + 2 * LL_PLANE) = q2;
    // nn.LayerNorm(128) of x + pe: two-pass, the row's 128 values sit in its 16 lanes
    float v[8] = {a[0] + r.e0.x, a[1] + r.e0.y, a[2] + r.e0.z, a[3] + r.e0.w, a[4] + r.e1.x, a[5] + r.e1.y, a[6] + r.e1.z, a[7] + r.e1.w};
    float sm = ((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7]));
    sm = ll_row_sum<LL_NG>(sm);
    const float mu = sm * (1.0f / LL_K);
    float q = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) { v[j] -= mu; q = fmaf(v[j], v[j], q); }
    q = ll_row_sum<LL_NG>(q);
    const float rstd = 1.0f / sqrtf(q * (1.0f / LL_K) + p.ln_eps);
    const float4 g0 = *reinterpret_cast<const float4*>(sgb + 8 * pj), g1 = *reinterpret_cast<const float4*>(sgb + 8 * pj + 4);
    const float4 b0 = *reinterpret_cast<const float4*>(sgb + LL_K + 8 * pj), b1 = *reinterpret_cast<const float4*>(sgb + LL_K + 8 * pj + 4);
    const float gm[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w}, bt[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = v[j] * rstd * gm[j] + bt[j];
    ll_split8(v, q0, q1, q2);
    *reinterpret_cast<u32x4l*>(d) = q0; *reinterpret_cast<u32x4l*>(d + LL_PLANE) = q1; *reinterpret_cast<u32x4l*>(d + 2 * LL_PLANE) = q2;
  };

  // ---- consumer role: output tile (row tile rt of the stage, column tile ct): lane (row l15, g) receives channels 16 tile + 4 g .. + 3 of its row
  unsigned offY[NTW];
  constexpr unsigned OOB = 0x80000000u;
  const bool two = p.Y2 != nullptr;
#pragma unroll
  for (int ct = 0; ct < NTW; ++ct) {
    const int tile = NTW * wave + ct, n = 16 * tile + 4 * g;
    const bool second = two && n >= p.split_n;
    offY[ct] = second ? (unsigned)((l15 * p.y2_stride + p.y2_choff + n - p.split_n) * 4) : (unsigned)((l15 * p.y_stride + p.y_choff + n) * 4);
  }
  const __amdgpu_buffer_rsrc_t rsY = __builtin_amdgcn_make_buffer_rsrc(p.Y, 0, (int)(p.M * p.y_stride * 4), 0x00020000);
  const __amdgpu_buffer_rsrc_t rsY2 = __builtin_amdgcn_make_buffer_rsrc(two ? p.Y2 : p.Y, 0, two ? (int)(p.M * p.y2_stride * 4) : 0, 0x00020000);
  const unsigned short* const bsrc = sll + (l15 * LL_NG) * 8;   // row l15 of a plane; + 16 rows per row tile; slot (4 s + g) ^ swz(row) (the same for rows l15 and l15 + 16)
  // NLN (compile time): how many of the wave's column tiles take the LayerNorm'd rows (tiles ascend: those come first, the raw ones behind) -- a run-time choice inside the
  // MFMA sequence would put register copies of the accumulators right behind the asm MFMAs, which the compiler does not know to be MFMAs (tools/check_asm_mfma_hazards.py)
  auto consume = [&](long long st, int buf, auto nln_tag) __attribute__((always_inline)) {
    constexpr int NLN = decltype(nln_tag)::value;
    constexpr bool need_ln = NLN > 0, need_raw = NLN < NTW;
    const unsigned short* bb = bsrc + buf * LL_BUF;
    // Two accumulator sets, one per row tile: a store's data registers are read some time after the store is issued, so the compiler guards the next write to them with a
    // (counted) wait for that store to COMPLETE -- with one set every row tile began by waiting out the stores of the one before (the "no stores" ablation build: -120 of 550 us)
    f32x4l acc[2][NTW];
#pragma unroll
    for (int q = 0; q < 2 * LL_KS; ++q) {
      const int rt = q / LL_KS, s4 = q % LL_KS;
      if (s4 == 0) {
#pragma unroll
        for (int ct = 0; ct < NTW; ++ct) acc[rt][ct] = f32x4l{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ct = 0; ct < NTW; ++ct) asm volatile("s_nop 1" : "+v"(acc[rt][ct]));
      }
      const unsigned short* bq = bb + (rt * 16 * LL_NG + ((4 * s4 + g) ^ LL<K>::swz(l15))) * 8;
      u32x4l xl[3], xw[3];
      if (need_ln) { xl[0] = *reinterpret_cast<const u32x4l*>(bq); xl[1] = *reinterpret_cast<const u32x4l*>(bq + LL_PLANE); xl[2] = *reinterpret_cast<const u32x4l*>(bq + 2 * LL_PLANE); }
      if (need_raw) { xw[0] = *reinterpret_cast<const u32x4l*>(bq + LL_SET); xw[1] = *reinterpret_cast<const u32x4l*>(bq + LL_SET + LL_PLANE); xw[2] = *reinterpret_cast<const u32x4l*>(bq + LL_SET + 2 * LL_PLANE); }
#pragma unroll
      for (int ct = 0; ct < NTW; ++ct) {
        if (LL_ABL & 2) { asm volatile("" : "+v"(acc[rt][ct]) : "v"(wf[ct][s4][0]), "v"(wf[ct][s4][1]), "v"(wf[ct][s4][2])); continue; }
        // D[channel][row]: A = the weight rows, B = the token rows.  Smallest terms first (the order of rowgemm_b3.hip)
        const u32x4l (&xb)[3] = ct < NLN ? xl : xw;
        ll_mfma(acc[rt][ct], wf[ct][s4][2], xb[0]); ll_mfma(acc[rt][ct], wf[ct][s4][0], xb[2]); ll_mfma(acc[rt][ct], wf[ct][s4][1], xb[1]);
        ll_mfma(acc[rt][ct], wf[ct][s4][1], xb[0]); ll_mfma(acc[rt][ct], wf[ct][s4][0], xb[1]); ll_mfma(acc[rt][ct], wf[ct][s4][0], xb[0]);
      }
      if (s4 == LL_KS - 1) {
#pragma unroll
        for (int ct = 0; ct < NTW; ++ct) asm volatile("s_nop 15\n\ts_nop 7" : "+v"(acc[rt][ct]));      // MFMA results -> the stores' data
        if (!(LL_ABL & 4) || acc[rt][0][0] == 123.456f) {
          const long long m0 = st * LL_ROWS + rt * 16;      // rows >= M: out of the descriptor's range, dropped (VGPR offset)
#pragma unroll
          for (int ct = 0; ct < NTW; ++ct) {
            const bool second = two && 16 * (NTW * wave + ct) >= p.split_n;      // (wave-uniform)
            const unsigned ro = (unsigned)((int)m0 * (second ? p.y2_stride : p.y_stride) * 4);
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4l, acc[rt][ct]), second ? rsY2 : rsY, offY[ct] + ro, 0, LL_STPOL);
          }
        }
        __builtin_amdgcn_sched_barrier(0);      // (the stores stay here: sunk behind the next row tile's MFMAs they would still be in flight when their registers are written again)
      }
    }
    // (row tile 0's set kept alive to here: dead after its stores, the register allocator would hand the same registers to row tile 1)
#pragma unroll
    for (int ct = 0; ct < NTW; ++ct) asm volatile("" : : "v"(acc[0][ct]));
  };

  // ---- pipeline: stage k of this block is global stage blockIdx.x + k gridDim.x
  auto run = [&](auto nln_tag) __attribute__((always_inline)) {      // (four copies of the loop; called out of line the lambda would take its captures -- the weight planes -- through scratch memory)
#ifdef LL_STAG      // lab: blocks start (blockIdx & 15) * LL_STAG * 0.49 us apart
    for (int i = 0; i < (int)(blockIdx.x & 15) * LL_STAG; ++i) __builtin_amdgcn_s_sleep(16);
#endif
    Raw r = load_rows();
    __syncthreads();                                           // gamma | beta staged
    produce(r, 0);
    r = load_rows();
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < 2 * NTW; ++i)      // as many dropped stores (distinct addresses: identical ones are merged) as a stage issues behind its row loads: the first
      __builtin_amdgcn_raw_buffer_store_b128(u32x4l{0u, 0u, 0u, 0u}, rsY, OOB + 16u * i, 0, 0);      // stage meets the loop head in the same counter state as every other one
    __syncthreads();
    int buf = 0;
    for (long long st = blockIdx.x; st < p.nstages; st += gridDim.x, buf ^= 1) {
      // stage st is multiplied out of buffer buf while the next one (its rows arrived during the previous multiply) is normed and split into the other
#if LL_PHASE
      // the two waves of a SIMD (w and w + 4) take the stage's two jobs in opposite order: while one multiplies, the other norms, splits and talks to memory
      if (wave < 4) {
        if (!(LL_ABL & 1) || st == blockIdx.x) produce(r, buf ^ 1);
        if (!(LL_ABL & 8)) r = load_rows();
        consume(st, buf, nln_tag);
      } else {
        consume(st, buf, nln_tag);
        if (!(LL_ABL & 1) || st == blockIdx.x) produce(r, buf ^ 1);
        if (!(LL_ABL & 8)) r = load_rows();
      }
#else
      if (!(LL_ABL & 1) || st == blockIdx.x) produce(r, buf ^ 1);
      if (!(LL_ABL & 8)) r = load_rows();
      consume(st, buf, nln_tag);
#endif
      __syncthreads();
    }
  };
  int nln = (p.ln_cols >> 4) - NTW * wave;                     // (an SGPR: one scalar branch per launch)
  nln = nln < 0 ? 0 : nln > NTW ? NTW : nln;
  static_assert(NTW == 3, "dispatch below");
  switch (nln) {
    case 0: run(std::integral_constant<int, 0>{}); break;
    case 1: run(std::integral_constant<int, 1>{}); break;
    case 2: run(std::integral_constant<int, 2>{}); break;
    default: run(std::integral_constant<int, 3>{}); break;
  }
}

template <int K>
int launch_lnlin(const LnLinArgs& p, long long nst, hipStream_t st) {
  static std::atomic<bool> attr_set[64];
  static std::atomic<int> cus[64];
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return LFSR_E_ARG;
  if (!attr_set[dev]) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_lnlin_b3<K, 3, false>), hipFuncAttributeMaxDynamicSharedMemorySize, LL<K>::SMEM);
    if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_lnlin_b3<K, 3, true>), hipFuncAttributeMaxDynamicSharedMemorySize, LL<K>::SMEM);
    if (e != hipSuccess) return LFSR_HIP_ERR(e);
    int v = 0;
    cus[dev] = (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) ? v : 256;
    attr_set[dev] = true;
  }
  long long grid = (long long)cus[dev] * (K == 128 ? 1 : 3);      // K = 64: 256-thread blocks of 49 KB, three per CU
  if (grid > nst) grid = nst;
  if (p.pe) hipLaunchKernelGGL((k_lnlin_b3<K, 3, true>), dim3((unsigned)grid), dim3(LL<K>::NTH), LL<K>::SMEM, st, p);
  else hipLaunchKernelGGL((k_lnlin_b3<K, 3, false>), dim3((unsigned)grid), dim3(LL<K>::NTH), LL<K>::SMEM, st, p);
  LFSR_CHECK_LAUNCH();
  return LFSR_OK;
}

}  // namespace

// LFSR_E_ARG = shape not covered (the caller keeps the panel form, rowgemm_b3.hip): (K, N) = (128, 384) or (64, 192) (waves x 3 column tiles), ln_cols / split_n multiples of 16
int lfsr_lnlin_b3_launch(const float* x, int x_stride, int x_choff, int K, const float* w_packed, const float* ln_g, const float* ln_b, float ln_eps, int ln_cols,
                         const float* pe, int pe_stride, int pe_rows, int pe_div, float* y, int y_stride, int y_choff,
                         float* y2, int y2_stride, int y2_choff, int split_n, long long M, int N, hipStream_t st) {
  if (!((K == 128 && N == 384) || (K == 64 && N == 192)) || !x || !w_packed || !ln_g || !ln_b || !y || M <= 0 || ln_cols < 0 || ln_cols > N || ln_cols % 16) return LFSR_E_ARG;
  if (y2 && (split_n % 16 || split_n <= 0 || split_n >= N)) return LFSR_E_ARG;
  if ((x_stride | x_choff | y_stride | y_choff) & 3 || (y2 && ((y2_stride | y2_choff) & 3)) || (pe && ((pe_stride & 3) || pe_rows <= 0 || pe_div <= 0))) return LFSR_E_ARG;
  if (x_stride < x_choff + K || y_stride < y_choff + (y2 ? split_n : N) || (y2 && y2_stride < y2_choff + N - split_n)) return LFSR_E_ARG;
  if (((uintptr_t)y | (uintptr_t)y2 | (uintptr_t)x | (uintptr_t)pe | (uintptr_t)ln_g | (uintptr_t)ln_b | (uintptr_t)w_packed) & 15) return LFSR_E_ARG;
  if ((M + 64) * (long long)x_stride * 4 >= (1LL << 31) || (M + 64) * (long long)y_stride * 4 >= (1LL << 31) || (y2 && (M + 64) * (long long)y2_stride * 4 >= (1LL << 31)))
    return LFSR_E_ARG;      // (32-bit buffer offsets, and 2^31 as the out-of-range marker)
  LnLinArgs p{};
  p.X = x; p.x_stride = x_stride; p.x_choff = x_choff; p.Wp = w_packed; p.Y = y; p.y_stride = y_stride; p.y_choff = y_choff;
  p.Y2 = y2; p.y2_stride = y2_stride; p.y2_choff = y2_choff; p.split_n = split_n; p.M = M; p.N = N;
  p.ln_g = ln_g; p.ln_b = ln_b; p.ln_eps = ln_eps; p.ln_cols = ln_cols; p.pe = pe; p.pe_stride = pe_stride; p.pe_rows = pe_rows; p.pe_div = pe_div;
  const long long nst = (M + LL_ROWS - 1) / LL_ROWS;
  if (nst > 0x7fffffffLL / LL_ROWS) return LFSR_E_ARG;
  p.nstages = (int)nst;
  return K == 128 ? launch_lnlin<128>(p, nst, st) : launch_lnlin<64>(p, nst, st);
}
