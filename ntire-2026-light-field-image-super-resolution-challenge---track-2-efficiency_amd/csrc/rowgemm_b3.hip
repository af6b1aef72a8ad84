// Row-streaming GEMM on the bf16 MFMA pipe with fp32 operands carried EXACTLY as three bf16 terms (default; LFSR_ROWGEMM=f32 keeps rowgemm.hip's fp32-MFMA kernel).
//   Y[m, n0:n0+64] = act(X[m, 0:K] W^T) (+res)        -- the transformer linears of EPIT.py:110-128 / LFT.py:188-246 (no bias)
// x = x0 + x1 + x2 and w = w0 + w1 + w2 by truncation (each term a bf16, the sum exact); the six products of order <= 2 (x0w0, x0w1, x1w0, x1w1, x0w2, x2w0) run as
// v_mfma_f32_16x16x32_bf16 with fp32 accumulation -- the dropped terms are below 2^-24 relative, the dot product is closer to fp64 than an fp32 FMA chain
// (tests/test_oracle_vs_golden.py::test_split_bf16_products).  Six K = 32 MFMAs of 17 cycles replace eight 16x16x4 fp32 MFMAs of 32 cycles: 2.5x less matrix-pipe time.
// Edge semantics: finite inputs only differ from an fp32 FMA chain by rounding; an infinite input yields NaN (inf - inf in the residual) where fp32 would carry the infinity;
// residual terms below the bf16 denormal range (|x| < 2^-110) are flushed -- both outside anything the models produce.
// Structure: a persistent 256-thread block keeps the three bf16 planes of its 64 x K weight panel in LDS (split once per block from the packed fp32 weights); each wave
// owns 16 token rows per tile and loads them straight from global memory in B-operand order (lane = row, eight consecutive k per k-group: two 16-B loads per K step),
// splits them in registers (5.5 VALU per element) and runs 4 column sub-tiles x 6 products per K step.  No A image in LDS, no barrier in the tile loop.
#include <stdlib.h>

#include "lfsr_internal.h"

#ifndef RB_VAR
#define RB_VAR 1   // timing variants (correct results): 1 residual rows loaded in the epilogue (default: out-projection at the LFT geometry 274 us against 291 us with the loads at the top of the tile)
#endif
#ifndef RB_EARLY
#define RB_EARLY 1   // 1: the next tile's rows requested in front of this tile's split (two raw-row sets); 0: behind it (one set)
#endif
#ifndef RB_ABL
#define RB_ABL 0   // diagnostic timing builds (WRONG results; tools/build_abl.sh): 1 no LayerNorm / split VALU, 2 no MFMAs, 4 no stores (and no residual loads), 8 no row loads after the first
#endif

namespace {

typedef float f32x4b __attribute__((ext_vector_type(4)));
typedef unsigned u32x4b __attribute__((ext_vector_type(4)));

struct RowGemmB3Args {
  const float* X; int x_stride; int x_choff;
  const float* Wp;       // [N rows][K] fp32 (packed, k contiguous)
  const float* R1; int r1_stride; int r1_choff;
  int r1_mask; float mk_slope;      // r1_mask: R1 is a saved activation, y = v * (R1 > 0 ? 1 : mk_slope) (the LeakyReLU' of a data gradient) instead of y = v + R1
  float* Y; int y_stride; int y_choff;
  long long M; int N;
  float slope;
  // LN form (as RowGemmArgs in rowgemm.hip): panels n0 < ln_cols see LayerNorm(x (+ pe)), the others the raw rows; panels n0 >= split_n store to Y2
  const float* ln_g; const float* ln_b; float ln_eps; int ln_cols;
  const float* pe; int pe_stride; int pe_rows; int pe_div;
  float* Y2; int y2_stride; int y2_choff; int split_n;
};

__device__ __forceinline__ unsigned b3_hi_pair(unsigned hi_src, unsigned lo_src) {      // (hi_src's upper half) << 16 | lo_src's upper half
  return __builtin_amdgcn_perm(hi_src, lo_src, 0x07060302u);
}
__device__ __forceinline__ float b3_residual(float a) { return a - __uint_as_float(__float_as_uint(a) & 0xffff0000u); }   // exact

// eight consecutive floats -> their three bf16 planes in MFMA operand order (element j in half j & 1 of register j / 2)
__device__ __forceinline__ void b3_split8(const float4 lo, const float4 hi, u32x4b& p0, u32x4b& p1, u32x4b& p2) {
  const float a[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
#pragma unroll
  for (int j = 0; j < 4; ++j) { unsigned t0, t1, t2; lfsr_split_pair(a[2 * j], a[2 * j + 1], t0, t1, t2); p0[j] = t0; p1[j] = t1; p2[j] = t2; }
}

// asm MFMA with the accumulator tied ("+v": vDst = SrcC, disjoint from the A / B registers).  With the builtins the register allocator may give a bf16 MFMA a vDst that
// overlaps its SrcB / SrcC PARTIALLY (it models 4-register results as read-before-write); on gfx950 that returns garbage in the overlapping registers (found in round 2 with
// debug builds of a split-bf16 form of the 3x3 conv, since retired: every product group whose result landed on such an allocation was wrong).  Every b3 kernel issues its
// MFMAs this way.
__device__ __forceinline__ void b3_mfma(f32x4b& c, const u32x4b a, const u32x4b b) {
  asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b));
}

// NB = 64: 256 threads (4 waves x 16 rows), up to three blocks per CU; NB = 128: 512 threads (8 waves x 16 rows), one block per CU -- the token rows are read and split
// once per 128 output columns instead of once per 64 (there is no barrier in the tile loop, so eight waves of one block overlap as well as four waves of two)
// KV < K: operand rows hold KV valid values (fuse.0 of DistgSSR: KV = 144 in five K steps of 32): weight columns and x values k >= KV are taken as zero
template <int K, bool LN = false, int NB = 64, int KV = K>
__global__ __launch_bounds__(NB * 4) void k_rowgemm_b3(RowGemmB3Args p) {
  constexpr int NTH = NB * 4, BMR = NB, NT = NB / 16;
  constexpr int KS = K / 32;                // K steps
  // LDS image of a weight plane: [K step s][k-group g][row n][8 bf16] -- consecutive rows 16 B apart, the four k-groups NB x 16 B (a multiple of 256 B) apart.
  // ds_read_b128 serves the lanes in four groups of 16 ({0-3, 12-15, 20-27}, {4-11, 16-19, 28-31}, ...: MI355X_MICROARCH.md, LDS table), each holding every
  // row l15 once and two k-groups: with the k-groups at +16 B inside a padded row (the first layout) rows r and r +- 1 of different k-groups shared a 16-B
  // slot -- SQ_LDS_BANK_CONFLICT was 49 % of SQ_LDS_IDX_ACTIVE; here every group covers the 16 slots of a 256-B bank row exactly once
  constexpr int PLANE = NB * K;             // bf16 per plane (no padding)
  extern __shared__ __attribute__((aligned(16))) unsigned short swb[];     // [3 planes][K / 8 (K step, k-group)][NB rows][8], then (LN) gamma[K], beta[K] as fp32
  float* const sgb = reinterpret_cast<float*>(swb + 3 * PLANE);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l15 = lane & 15, g = lane >> 4;
  const int n0 = blockIdx.y * NB;
  const long long ntiles = (p.M + BMR - 1) / BMR;

  // the block's weight panel: split into planes, 8 consecutive k per thread-iteration
  for (int i = tid; i < NB * (K / 8); i += NTH) {
    const int r = i / (K / 8), c = i - r * (K / 8);
    const bool ok = n0 + r < p.N && c * 8 < KV;
    const float* src = p.Wp + (long long)(ok ? n0 + r : 0) * KV + (ok ? c * 8 : 0);
    float4 lo = *reinterpret_cast<const float4*>(src), hi = *reinterpret_cast<const float4*>(src + 4);
    if (!ok) { lo = make_float4(0.f, 0.f, 0.f, 0.f); hi = lo; }
    u32x4b w0, w1, w2;
    b3_split8(lo, hi, w0, w1, w2);
    const int slot = (c * NB + r) * 8;        // c = 4 s + g
    *reinterpret_cast<u32x4b*>(swb + 0 * PLANE + slot) = w0;
    *reinterpret_cast<u32x4b*>(swb + 1 * PLANE + slot) = w1;
    *reinterpret_cast<u32x4b*>(swb + 2 * PLANE + slot) = w2;
  }
  if constexpr (LN) for (int i = tid; i < 2 * K; i += NTH) sgb[i] = i < K ? p.ln_g[i] : p.ln_b[i - K];      // (read per tile: LDS latency instead of an L1 round trip)
  __syncthreads();

  // this lane's token row of a tile and its 8-float groups: rows past M read as zero through the buffer descriptor
  typedef float f32x4g __attribute__((ext_vector_type(4)));
  const __amdgpu_buffer_rsrc_t rsX = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.X), 0, (int)(p.M * p.x_stride * 4), 0x00020000);
  const int offL = ((wave * 16 + l15) * p.x_stride + p.x_choff + 8 * g) * 4;
  // RB_EARLY (plain row-GEMMs): two sets of raw rows -- the next tile's rows are asked for BEFORE this tile's split (into the other set) instead of behind it, so a tile's
  // loads have the split, the MFMAs and the epilogue of the previous tile to arrive in
  constexpr bool EARLY = RB_EARLY && !LN;
  float4 xrA[KS][2], xrB[EARLY ? KS : 1][2];
  float4 pr[LN ? KS : 1][2];
  const bool do_ln = LN && n0 < p.ln_cols;
  auto prefetch = [&](long long tile, float4 (&xr)[KS][2]) {
    // (the tile's base goes into the VGPR offset: the descriptor's bounds check covers the VGPR and immediate offsets only, an SGPR offset is added unchecked)
    const int ot = offL + (int)(tile * BMR) * p.x_stride * 4;
#pragma unroll
    for (int s = 0; s < KS; ++s)
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        f32x4g v = __builtin_bit_cast(f32x4g, __builtin_amdgcn_raw_buffer_load_b128(rsX, ot + (32 * s + 4 * e) * 4, 0, 0));
        if constexpr (KV < K) if (32 * s + 32 > KV && 32 * s + 8 * g >= KV) v = f32x4g{0.f, 0.f, 0.f, 0.f};     // (the neighbouring row's values: finite, but not ours)
        xr[s][e] = make_float4(v.x, v.y, v.z, v.w);
      }
    if constexpr (LN) if (do_ln && p.pe) {
      const int m = (int)(tile * BMR) + wave * 16 + l15;
      const float* pp = p.pe + (long long)((m / p.pe_div) % p.pe_rows) * p.pe_stride + 8 * g;
#pragma unroll
      for (int s = 0; s < KS; ++s) { pr[s][0] = *reinterpret_cast<const float4*>(pp + 32 * s); pr[s][1] = *reinterpret_cast<const float4*>(pp + 32 * s + 4); }
    }
  };
  // Output (and residual) rows through buffer descriptors as well: rows past M and column groups past N are dropped (read as zero) by the bounds check, so the
  // tile loop is straight-line code -- the compiler can then COUNT its waits (vmcnt counts stores too, in issue order): the top of the loop waits for the prefetched
  // rows with the NT stores of the previous tile still in flight, where the branchy form waited for vmcnt(0), i.e. for those stores to drain
  // (profiles/r03_logs/c15_lin_abl.log: loads + stores alone 450 us, arithmetic alone 426 us, together 710 us at the LFT geometry).
  const bool second = p.Y2 && n0 >= p.split_n;        // (block-uniform)
  const int ys = second ? p.y2_stride : p.y_stride, yc = second ? p.y2_choff - p.split_n : p.y_choff;
  const __amdgpu_buffer_rsrc_t rsY = __builtin_amdgcn_make_buffer_rsrc(second ? p.Y2 : p.Y, 0, (int)(p.M * ys * 4), 0x00020000);
  const __amdgpu_buffer_rsrc_t rsR = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.R1 ? p.R1 : p.X), 0, p.R1 ? (int)(p.M * p.r1_stride * 4) : 0, 0x00020000);
  constexpr unsigned OOB = 0x80000000u;               // (the launchers keep every extent below 2^31 bytes: OOB + a tile's base stays out of range, without wrapping)
  unsigned offY[NT], offR[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    const int n = n0 + 16 * t + 4 * g;
    const bool ok = n + 3 < p.N;
    offY[t] = ok ? (unsigned)(((wave * 16 + l15) * ys + yc + n) * 4) : OOB;
    offR[t] = ok && p.R1 ? (unsigned)(((wave * 16 + l15) * p.r1_stride + p.r1_choff + n) * 4) : OOB;
  }
  long long tile = blockIdx.x;
  prefetch(tile < ntiles ? tile : ntiles, xrA);
#pragma unroll
  for (int t = 0; t < NT; ++t)      // NT dropped stores: the first tile meets the loop head in the same counter state as every other one
    __builtin_amdgcn_raw_buffer_store_b128(u32x4b{0u, 0u, 0u, 0u}, rsY, OOB, 0, 0);
  const unsigned short* wl = swb + (g * NB + l15) * 8;         // this lane's A-operand slot: k-group g, weight row l15 (+16 rows per sub-tile, + 4 NB slots per K step)
  auto body = [&](float4 (&xr)[KS][2], float4 (&xn)[KS][2]) {
    // the residual rows of THIS tile, asked for before the arithmetic (older than the next prefetch: the epilogue waits for them alone)
    const int so = (int)(tile * BMR);
    f32x4g rv[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) rv[t] = f32x4g{-0.f, -0.f, -0.f, -0.f};      // (v + -0 = v for every v, the sign of a zero included)
    auto load_res = [&]() {
      if (p.R1 && !(RB_ABL & 4))
#pragma unroll
        for (int t = 0; t < NT; ++t) rv[t] = __builtin_bit_cast(f32x4g, __builtin_amdgcn_raw_buffer_load_b128(rsR, offR[t] + (unsigned)(so * p.r1_stride * 4), 0, 0));
    };
    if (!(RB_VAR & 1)) load_res();
    if constexpr (EARLY) {
      if (!(RB_ABL & 8)) prefetch(tile + gridDim.x < ntiles ? tile + gridDim.x : ntiles, xn);
      __builtin_amdgcn_sched_barrier(0);      // (the scheduler otherwise sinks these loads in between the split's instructions: the point is that they are in flight before it)
    }
    if constexpr (LN) if (do_ln && !(RB_ABL & 1)) {
      // nn.LayerNorm(K) of the lane's row: its K values sit in the four lanes (row l15, g = 0..3), 8 KS each -- an in-lane sum and two wave shuffles per pass
      float sm = 0.f;
#pragma unroll
      for (int s = 0; s < KS; ++s)
#pragma unroll
        for (int e = 0; e < 2; ++e) {
          if (p.pe) { xr[s][e].x += pr[s][e].x; xr[s][e].y += pr[s][e].y; xr[s][e].z += pr[s][e].z; xr[s][e].w += pr[s][e].w; }
          sm += (xr[s][e].x + xr[s][e].y) + (xr[s][e].z + xr[s][e].w);
        }
      sm += __shfl_xor(sm, 16);
      sm += __shfl_xor(sm, 32);
      const float mu = sm * (1.0f / K);
      float q2 = 0.f;
#pragma unroll
      for (int s = 0; s < KS; ++s)
#pragma unroll
        for (int e = 0; e < 2; ++e) {
          xr[s][e].x -= mu; xr[s][e].y -= mu; xr[s][e].z -= mu; xr[s][e].w -= mu;
          q2 += (xr[s][e].x * xr[s][e].x + xr[s][e].y * xr[s][e].y) + (xr[s][e].z * xr[s][e].z + xr[s][e].w * xr[s][e].w);
        }
      q2 += __shfl_xor(q2, 16);
      q2 += __shfl_xor(q2, 32);
      const float rstd = 1.0f / sqrtf(q2 * (1.0f / K) + p.ln_eps);
#pragma unroll
      for (int s = 0; s < KS; ++s)
#pragma unroll
        for (int e = 0; e < 2; ++e) {
          const float4 gv = *reinterpret_cast<const float4*>(sgb + 32 * s + 8 * g + 4 * e);
          const float4 bv = *reinterpret_cast<const float4*>(sgb + K + 32 * s + 8 * g + 4 * e);
          xr[s][e] = make_float4(xr[s][e].x * rstd * gv.x + bv.x, xr[s][e].y * rstd * gv.y + bv.y, xr[s][e].z * rstd * gv.z + bv.z, xr[s][e].w * rstd * gv.w + bv.w);
        }
    }
    u32x4b x0[KS], x1[KS], x2[KS];
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      if constexpr (RB_ABL & 1) {
        x0[s] = u32x4b{__float_as_uint(xr[s][0].x), __float_as_uint(xr[s][0].y), __float_as_uint(xr[s][0].z), __float_as_uint(xr[s][0].w)};
        x1[s] = u32x4b{__float_as_uint(xr[s][1].x), __float_as_uint(xr[s][1].y), __float_as_uint(xr[s][1].z), __float_as_uint(xr[s][1].w)};
        x2[s] = x0[s];
      } else b3_split8(xr[s][0], xr[s][1], x0[s], x1[s], x2[s]);
    }
    if constexpr (!EARLY) if (!(RB_ABL & 8)) prefetch(tile + gridDim.x < ntiles ? tile + gridDim.x : ntiles, xn);      // (past the end: rows >= M, zeros without traffic; !EARLY: xn is xr)
    f32x4b acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[t] = f32x4b{0.f, 0.f, 0.f, 0.f};
    // the split's VALU results feed the asm MFMAs below and the compiler does not see an MFMA there: the wait states a VALU write needs before an MFMA reads the
    // register are inserted by hand, TIED to every plane so that the scheduler cannot sink a split instruction behind them (found with the LN form: the first
    // sub-tile's columns were wrong when the split was scheduled right in front of the first MFMA)
#pragma unroll
    for (int s = 0; s < KS; ++s) asm volatile("s_nop 4" : "+v"(x0[s]), "+v"(x1[s]), "+v"(x2[s]));
#pragma unroll
    for (int t = 0; t < NT; t += 4) asm volatile("s_nop 1" : "+v"(acc[t]), "+v"(acc[t + 1]), "+v"(acc[t + 2]), "+v"(acc[t + 3]));       // (the zeroed accumulators are MFMA sources too)
#pragma unroll
    for (int s = 0; s < KS; ++s) {
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        const unsigned short* wq = wl + (4 * s * NB + t * 16) * 8;
        const u32x4b w0 = *reinterpret_cast<const u32x4b*>(wq);
        const u32x4b w1 = *reinterpret_cast<const u32x4b*>(wq + PLANE);
        const u32x4b w2 = *reinterpret_cast<const u32x4b*>(wq + 2 * PLANE);
        // D[channel][row]: A = the weight rows, B = the token rows.  Smallest terms first.
        if constexpr (RB_ABL & 2) { asm volatile("" : "+v"(acc[t]) : "v"(w0), "v"(w1), "v"(w2), "v"(x0[s]), "v"(x1[s]), "v"(x2[s])); continue; }
        b3_mfma(acc[t], w2, x0[s]);
        b3_mfma(acc[t], w0, x2[s]);
        b3_mfma(acc[t], w1, x1[s]);
        b3_mfma(acc[t], w1, x0[s]);
        b3_mfma(acc[t], w0, x1[s]);
        b3_mfma(acc[t], w0, x0[s]);
      }
    }
#pragma unroll
    for (int t = 0; t < NT; t += 4)      // MFMA results -> VALU reads below (the wait the compiler would pad for a builtin), tied to the accumulators
      asm volatile("s_nop 15\n\ts_nop 15" : "+v"(acc[t]), "+v"(acc[t + 1]), "+v"(acc[t + 2]), "+v"(acc[t + 3]));
    // epilogue: lane (row l15, g) holds channels n0 + 16 t + 4 g .. + 3 of its row
    if (RB_VAR & 1) load_res();
    if (!(RB_ABL & 4) || acc[0][0] == 123.456f) {
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        float v[4] = {acc[t][0], acc[t][1], acc[t][2], acc[t][3]};
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] = v[k] >= 0.f ? v[k] : v[k] * p.slope;
        if (p.r1_mask) {      // (block-uniform)
          v[0] *= rv[t].x > 0.f ? 1.f : p.mk_slope; v[1] *= rv[t].y > 0.f ? 1.f : p.mk_slope; v[2] *= rv[t].z > 0.f ? 1.f : p.mk_slope; v[3] *= rv[t].w > 0.f ? 1.f : p.mk_slope;
        } else {
          v[0] += rv[t].x; v[1] += rv[t].y; v[2] += rv[t].z; v[3] += rv[t].w;      // (-0 without a residual; +0 for the dropped column groups of a residual)
        }
        __builtin_amdgcn_raw_buffer_store_b128(u32x4b{__float_as_uint(v[0]), __float_as_uint(v[1]), __float_as_uint(v[2]), __float_as_uint(v[3])}, rsY, offY[t] + (unsigned)(so * ys * 4), 0, 0);
      }
    }
  };
  if constexpr (EARLY) {
    for (;;) {
      if (tile >= ntiles) break;
      body(xrA, xrB); tile += gridDim.x;
      if (tile >= ntiles) break;
      body(xrB, xrA); tile += gridDim.x;
    }
  } else {
    for (; tile < ntiles; tile += gridDim.x) body(xrA, xrA);
  }
}

template <int K, bool LN = false, int NB = 64, int KV = K>
int launch_b3(const RowGemmB3Args& p, hipStream_t st) {
  constexpr int smem = 3 * NB * K * 2 + (LN ? 2 * K * 4 : 0);
  constexpr int per_cu = NB == 128 ? 1 : LN ? 2 : (smem <= 52 * 1024 ? 3 : 2);      // (the LN form of K = 128 needs 189 VGPRs: two 256-thread blocks per CU)
  static std::atomic<bool> attr_set[64];
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return LFSR_E_ARG;
  if (!attr_set[dev]) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_rowgemm_b3<K, LN, NB, KV>), hipFuncAttributeMaxDynamicSharedMemorySize, smem);
    if (e != hipSuccess) return LFSR_HIP_ERR(e);
    attr_set[dev] = true;
  }
  const long long ntiles = (p.M + NB - 1) / NB;
  const int nby = (p.N + NB - 1) / NB;
  int gx = 256 * per_cu / nby;
  if (gx > 8) gx &= ~7;
  if (gx < 1) gx = 1;
  if (gx > ntiles) gx = (int)ntiles;
  hipLaunchKernelGGL((k_rowgemm_b3<K, LN, NB, KV>), dim3((unsigned)gx, (unsigned)nby), dim3(NB * 4), smem, st, p);
  LFSR_CHECK_LAUNCH();
  return LFSR_OK;
}

}  // namespace

// LFSR_E_ARG = shape not covered (the caller runs the fp32-MFMA row-GEMM)
int lfsr_rowgemm_b3_launch(const float* x, int x_stride, int x_choff, int K, const float* w_packed, const float* res, int res_stride, int res_choff,
                           float* y, int y_stride, int y_choff, long long M, int N, float slope, hipStream_t st) {
  if ((x_stride | x_choff) & 3 || N % 64 || (y_stride | y_choff) & 3 || (res && ((res_stride | res_choff) & 3))) return LFSR_E_ARG;
  if (((uintptr_t)y | (uintptr_t)x | (uintptr_t)res | (uintptr_t)w_packed) & 15) return LFSR_E_ARG;
  if ((M + 128) * (long long)x_stride * 4 >= (1LL << 31) || (M + 128) * (long long)y_stride * 4 >= (1LL << 31) || (res && (M + 128) * (long long)res_stride * 4 >= (1LL << 31)))
    return LFSR_E_ARG;      // (32-bit buffer offsets, and 2^31 as the out-of-range marker)
  RowGemmB3Args p{};
  p.X = x; p.x_stride = x_stride; p.x_choff = x_choff; p.Wp = w_packed; p.R1 = res; p.r1_stride = res_stride; p.r1_choff = res_choff;
  p.Y = y; p.y_stride = y_stride; p.y_choff = y_choff; p.M = M; p.N = N; p.slope = slope;
  const char* nsel = lfsr_sel("LFSR_B3_NB");       // "128": 128-column panels where N allows (A/B runs: EPIT 818 -> 800, LFT 1636 -> 1646 patches/s; not the default)
  const bool wide = nsel && nsel[0] == '1' && N % 128 == 0;
  switch (K) {
    case 64: return wide ? launch_b3<64, false, 128>(p, st) : launch_b3<64>(p, st);
    case 128: return wide ? launch_b3<128, false, 128>(p, st) : launch_b3<128>(p, st);
    case 144: return launch_b3<160, false, 64, 144>(p, st);      // DistgSSR's fuse.0 (DistgSSR.py:99): 144 valid of 160 operand columns
    default: return LFSR_E_ARG;
  }
}

// dx = (dy W) . lrelu'(mk): the data gradient of a 1x1 conv with 64 outputs behind a saved activation mk of N channels (DistgSSR fuse.0: N = 144; train.py:256-264 runs
// autograd through DistgSSR.py:99); wT_packed [N][64].  LFSR_E_ARG = shape not covered (the caller keeps the fp32-MFMA kernel)
int lfsr_rowgemm_b3_dgrad_launch(const float* dy, int dy_stride, int dy_choff, const float* wT_packed, const float* mk, int mk_stride, int mk_choff, float mk_slope,
                                 float* dx, int dx_stride, int dx_choff, long long M, int N, hipStream_t st) {
  if (!dy || !wT_packed || !mk || !dx || M <= 0 || N <= 0 || N % 16) return LFSR_E_ARG;
  if ((dy_stride | dy_choff | dx_stride | dx_choff | mk_stride | mk_choff) & 3) return LFSR_E_ARG;
  if (((uintptr_t)dy | (uintptr_t)dx | (uintptr_t)mk | (uintptr_t)wT_packed) & 15) return LFSR_E_ARG;
  if ((M + 128) * (long long)dy_stride * 4 >= (1LL << 31) || (M + 128) * (long long)dx_stride * 4 >= (1LL << 31) || (M + 128) * (long long)mk_stride * 4 >= (1LL << 31)) return LFSR_E_ARG;
  RowGemmB3Args p{};
  p.X = dy; p.x_stride = dy_stride; p.x_choff = dy_choff; p.Wp = wT_packed; p.R1 = mk; p.r1_stride = mk_stride; p.r1_choff = mk_choff; p.r1_mask = 1; p.mk_slope = mk_slope;
  p.Y = dx; p.y_stride = dx_stride; p.y_choff = dx_choff; p.M = M; p.N = N; p.slope = 1.0f;
  return launch_b3<64>(p, st);
}

// LayerNorm + attention in-projection in one launch on the three-term bf16 form (argument meaning as lfsr_rowgemm_ln_launch in rowgemm.hip)
int lfsr_rowgemm_b3_ln_launch(const float* x, int x_stride, int x_choff, int K, const float* w_packed, const float* ln_g, const float* ln_b, float ln_eps, int ln_cols,
                              const float* pe, int pe_stride, int pe_rows, int pe_div, float* y, int y_stride, int y_choff,
                              float* y2, int y2_stride, int y2_choff, int split_n, long long M, int N, hipStream_t st) {
  if (!x || !w_packed || !ln_g || !ln_b || !y || M <= 0 || N <= 0 || N % 64 || ln_cols % 64 || (y2 && (split_n % 64 || split_n <= 0 || split_n >= N))) return LFSR_E_ARG;
  if ((x_stride | x_choff | y_stride | y_choff) & 3 || (y2 && ((y2_stride | y2_choff) & 3)) || (pe && ((pe_stride & 3) || pe_rows <= 0 || pe_div <= 0))) return LFSR_E_ARG;
  if (x_stride < x_choff + K || y_stride < y_choff + (y2 ? split_n : N) || (y2 && y2_stride < y2_choff + N - split_n)) return LFSR_E_ARG;
  if (((uintptr_t)y | (uintptr_t)y2 | (uintptr_t)x | (uintptr_t)pe | (uintptr_t)ln_g | (uintptr_t)ln_b | (uintptr_t)w_packed) & 15) return LFSR_E_ARG;
  if ((M + 128) * (long long)x_stride * 4 >= (1LL << 31) || (M + 128) * (long long)y_stride * 4 >= (1LL << 31) || (y2 && (M + 128) * (long long)y2_stride * 4 >= (1LL << 31)))
    return LFSR_E_ARG;
  {   // K = 128, N = 384 (the q | k | v projections of EPIT and of LFT's spatial transformer): the weights-in-registers form (lnlin_b3.hip); LFSR_LNLIN=0: the panel form below
    const char* lsel = lfsr_sel("LFSR_LNLIN");
    if (!(lsel && lsel[0] == '0')) {
      const int rc = lfsr_lnlin_b3_launch(x, x_stride, x_choff, K, w_packed, ln_g, ln_b, ln_eps, ln_cols, pe, pe_stride, pe_rows, pe_div, y, y_stride, y_choff,
                                          y2, y2_stride, y2_choff, split_n, M, N, st);
      if (rc != LFSR_E_ARG) return rc;
    }
  }
  RowGemmB3Args p{};
  p.X = x; p.x_stride = x_stride; p.x_choff = x_choff; p.Wp = w_packed; p.Y = y; p.y_stride = y_stride; p.y_choff = y_choff; p.M = M; p.N = N; p.slope = 1.0f;
  p.ln_g = ln_g; p.ln_b = ln_b; p.ln_eps = ln_eps; p.ln_cols = ln_cols; p.pe = pe; p.pe_stride = pe_stride; p.pe_rows = pe_rows; p.pe_div = pe_div;
  p.Y2 = y2; p.y2_stride = y2_stride; p.y2_choff = y2_choff; p.split_n = split_n;
  // 128-column panels by default here (the norm and the split are repeated per panel: EPIT 818 -> 831, LFT 1636 -> 1680 patches/s with them; LFSR_B3_NB=64: 64-column panels)
  const char* nsel = lfsr_sel("LFSR_B3_NB");
  const bool wide = !(nsel && nsel[0] == '6') && ln_cols % 128 == 0 && (!y2 || split_n % 128 == 0);
  switch (K) {
    case 64: return wide ? launch_b3<64, true, 128>(p, st) : launch_b3<64, true>(p, st);
    case 128: return wide ? launch_b3<128, true, 128>(p, st) : launch_b3<128, true>(p, st);
    default: return LFSR_E_ARG;
  }
}
