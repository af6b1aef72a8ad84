// Row-streaming GEMM with LDS-resident weights, fp32 MFMA: Y[m, n0:n0+NB] = act(X[m, 0:K] W^T (+bias)) (+res), for the 1x1 convs /
// nn.Linear layers whose rows are plain pixel or token vectors (DistgSSR fuse.0 144->64, DistgSSR.py:99; the transformer
// linears of EPIT.py:110-128 / LFT.py:188-246).  These layers sit near the HBM ridge (AI 16-40 FLOP/B), so the kernel is built
// like a streaming op: a persistent 512-thread block keeps its NB x K weight panel in LDS for its whole life and walks 128-row
// tiles; the next tile's rows are prefetched into registers (16-B loads, rows contiguous) while the current one runs its
// MFMAs; one A image in LDS (rows padded by 4 floats: conflict-free ds_read_b128), two barriers per tile.
// Against the generic gather-GEMM this removes the per-stage weight re-staging (a 128-row block re-read the whole panel) and
// the 64-float K staging granularity.
#include <stdlib.h>

#include "gemm_gather_kernel.h"
#include "lfsr_internal.h"

namespace {

struct RowGemmArgs {
  const float* X; int x_stride; int x_choff;
  const float* Wp;       // [N rows][K] (packed, k contiguous)
  const float* bias;
  const float* R1; int r1_stride; int r1_choff;
  const float* Mk; int mk_stride; int mk_choff; float mk_slope;   // backward: v *= (Mk > 0 ? 1 : mk_slope), the LeakyReLU' mask of a saved activation
  float* Y; int y_stride; int y_choff;
  long long M; int N;
  float slope;
  // LN form: the rows feeding the panels n0 < ln_cols are LayerNorm(x (+ pe)) (nn.LayerNorm(K), eps ln_eps); the others see the raw rows
  const float* ln_g; const float* ln_b; float ln_eps; int ln_cols;
  const float* pe; int pe_stride; int pe_rows; int pe_div;        // pe row of token m: (m / pe_div) % pe_rows
  // second output: the panels n0 >= split_n store to Y2 (column n - split_n)
  float* Y2; int y2_stride; int y2_choff; int split_n;
};

template <int K, int NB, int BMR, bool LN = false>
__global__ __launch_bounds__(BMR * 4) void k_rowgemm(RowGemmArgs p) {
  constexpr int LR = K + 4;                 // LDS row stride (floats)
  constexpr int NTH = BMR * 4;              // BMR/32 row groups x 2 column halves x 64 lanes
  constexpr int NT = NB / 64;               // 32-col tiles per wave (waves: 4 row groups x 2 column halves)
  constexpr int CPR = K / 4;                // 16-B chunks per row
  constexpr int AL = (BMR * CPR + NTH - 1) / NTH;   // A chunks per thread
  constexpr int WL = (NB * CPR + NTH - 1) / NTH;
  extern __shared__ __attribute__((aligned(16))) float smr[];
  float* sW = smr;                          // [NB][LR]
  float* sA = smr + NB * LR;                // [BMR][LR]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int half = lane >> 5, l31 = lane & 31;
  const int rg = wave % (BMR / 32), ng = wave / (BMR / 32);
  const int n0 = blockIdx.y * NB;
  const long long ntiles = (p.M + BMR - 1) / BMR;

  // weight panel (rows beyond N are zero in the packed buffer's padding only up to Npad32: guard).  All of a thread's loads are
  // issued before its first LDS store: written as load-then-store per element the compiler waits for every load in turn
  // (WL serial L2 round trips, ~6 us per launch)
  {
    float4 wv[WL];
#pragma unroll
    for (int i = 0; i < WL; ++i) {
      const int idx = tid + NTH * i, r = idx / CPR, c = idx - r * CPR;
      const bool ok = idx < NB * CPR && n0 + r < p.N;
      const float4 v = *reinterpret_cast<const float4*>(p.Wp + (long long)(ok ? n0 + r : 0) * K + (ok ? c : 0) * 4);
      wv[i] = ok ? v : make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int i = 0; i < WL; ++i) {
      const int idx = tid + NTH * i, r = idx / CPR, c = idx - r * CPR;
      if (idx < NB * CPR) *reinterpret_cast<float4*>(sW + r * LR + c * 4) = wv[i];
    }
  }

  // row-tile loads through a buffer descriptor sized to the tensor: a thread's AL byte offsets inside a tile are constants, the tile
  // adds a wave-uniform offset, rows past M are out of range and read as zero -- no branches, no 64-bit address arithmetic between
  // the barrier and the MFMAs (the launcher checks the tensor spans < 2 GiB)
  typedef float f32x4g __attribute__((ext_vector_type(4)));
  const __amdgpu_buffer_rsrc_t rsX = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.X), 0, (int)(p.M * p.x_stride * 4), 0x00020000);
  int offX[AL];
#pragma unroll
  for (int i = 0; i < AL; ++i) {
    const int idx = tid + NTH * i, r = idx / CPR, c = idx - r * CPR;
    offX[i] = idx < BMR * CPR ? (r * p.x_stride + p.x_choff + c * 4) * 4 : (int)0x80000000u;
  }
  float4 ra[AL];
  // LN form: a row is spread over CPR consecutive lanes, one 16-B chunk each (chunk tid % CPR of the rows tid / CPR + i NTH / CPR) -- the layout and the
  // reduction order of k_layernorm, so the fused form returns the same bits as LayerNorm launch + plain row-GEMM
  static_assert(!LN || (NTH % CPR == 0 && (BMR * CPR) % NTH == 0 && (CPR & (CPR - 1)) == 0 && CPR <= 64), "LN form: whole rows per lane group");
  const bool do_ln = LN && n0 < p.ln_cols;
  float4 lng = make_float4(0.f, 0.f, 0.f, 0.f), lnb = lng;
  float4 rp[LN ? AL : 1];
  if (do_ln) {
    lng = *reinterpret_cast<const float4*>(p.ln_g + (tid % CPR) * 4);
    lnb = *reinterpret_cast<const float4*>(p.ln_b + (tid % CPR) * 4);
  }
  auto prefetch = [&](long long tile) {
    const int s4 = (int)(tile * BMR) * p.x_stride * 4;
#pragma unroll
    for (int i = 0; i < AL; ++i) {
      const f32x4g v = __builtin_bit_cast(f32x4g, __builtin_amdgcn_raw_buffer_load_b128(rsX, offX[i] + s4, 0, 0));     // (VGPR offset: the bounds check does not cover an SGPR offset)
      ra[i] = make_float4(v.x, v.y, v.z, v.w);
    }
    if constexpr (LN) if (do_ln && p.pe) {
#pragma unroll
      for (int i = 0; i < AL; ++i) {
        const int m = (int)(tile * BMR) + tid / CPR + i * (NTH / CPR);
        rp[i] = *reinterpret_cast<const float4*>(p.pe + (long long)((m / p.pe_div) % p.pe_rows) * p.pe_stride + (tid % CPR) * 4);
      }
    }
  };

  long long tile = blockIdx.x;
  if (tile < ntiles) prefetch(tile);
  const float* aRow = sA + (rg * 32 + l31) * LR + 4 * half;
  const float* bRow = sW + (ng * (NB / 2) + l31) * LR + 4 * half;
  for (; tile < ntiles; tile += gridDim.x) {
    if constexpr (LN) if (do_ln) {
#pragma unroll
      for (int i = 0; i < AL; ++i) {
        float4 v = ra[i];
        if (p.pe) { v.x += rp[i].x; v.y += rp[i].y; v.z += rp[i].z; v.w += rp[i].w; }
        float sm = v.x + v.y + v.z + v.w;
#pragma unroll
        for (int o = CPR / 2; o > 0; o >>= 1) sm += __shfl_xor(sm, o, CPR);
        const float mu = sm * (1.0f / K);
        const float dx = v.x - mu, dy = v.y - mu, dz = v.z - mu, dw = v.w - mu;
        float q2 = dx * dx + dy * dy + dz * dz + dw * dw;
#pragma unroll
        for (int o = CPR / 2; o > 0; o >>= 1) q2 += __shfl_xor(q2, o, CPR);
        const float rstd = 1.0f / sqrtf(q2 * (1.0f / K) + p.ln_eps);
        ra[i] = make_float4(dx * rstd * lng.x + lnb.x, dy * rstd * lng.y + lnb.y, dz * rstd * lng.z + lnb.z, dw * rstd * lng.w + lnb.w);
      }
    }
#pragma unroll
    for (int i = 0; i < AL; ++i) {
      int idx = tid + NTH * i;
      if (idx < BMR * CPR) { int r = idx / CPR, c = idx - r * CPR; *reinterpret_cast<float4*>(sA + r * LR + c * 4) = ra[i]; }
    }
    __syncthreads();
    if (tile + gridDim.x < ntiles) prefetch(tile + gridDim.x);
    // (backward) the LeakyReLU' mask operand of THIS tile's epilogue is requested here, so its HBM latency runs under the MFMAs
    float4 mkv[NT][4];
    if (p.Mk) {
      const long long mm = tile * BMR + rg * 32 + l31;
#pragma unroll
      for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int n = n0 + ng * (NB / 2) + t * 32 + 4 * half + 8 * q;
          mkv[t][q] = (mm < p.M && n + 3 < p.N) ? *reinterpret_cast<const float4*>(p.Mk + mm * p.mk_stride + p.mk_choff + n) : make_float4(1.f, 1.f, 1.f, 1.f);
        }
    }
    f32x16 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
#pragma unroll 6
    for (int j = 0; j < K / 8; ++j) {
      float4 a = *reinterpret_cast<const float4*>(aRow + 8 * j);
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        float4 b = *reinterpret_cast<const float4*>(bRow + t * 32 * LR + 8 * j);
        // operands swapped (A = the weight rows, B = the x rows): D[channel][row m], so a lane holds runs of four consecutive CHANNELS of
        // one row -- the epilogue stores 16 B per lane with no transposition
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(b.x, a.x, acc[t], 0, 0, 0);
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(b.y, a.y, acc[t], 0, 0, 0);
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(b.z, a.z, acc[t], 0, 0, 0);
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(b.w, a.w, acc[t], 0, 0, 0);
      }
    }
    // epilogue straight from the accumulators: register group q of tile t = channels n0' + 32 t + 8 q + 4 half .. + 3 of row m0 + 32 rg + l31:
    // one 16-B store per group (bias / residual as 16-B loads, all of a tile's residual loads issued before the first use)
    const long long m = tile * BMR + rg * 32 + l31;
    const bool second = p.Y2 && n0 >= p.split_n;        // (block-uniform)
    float* const Yp = second ? p.Y2 : p.Y;
    const int ys = second ? p.y2_stride : p.y_stride, yc = second ? p.y2_choff - p.split_n : p.y_choff;
    if (m < p.M) {
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        const int nb = n0 + ng * (NB / 2) + t * 32 + 4 * half;
        float4 rv[4];
        if (p.R1) {
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const int n = nb + 8 * q;
            rv[q] = n + 3 < p.N ? *reinterpret_cast<const float4*>(p.R1 + m * p.r1_stride + p.r1_choff + n) : make_float4(0.f, 0.f, 0.f, 0.f);
          }
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int n = nb + 8 * q;
          if (n >= p.N) continue;
          float v[4] = {acc[t][4 * q], acc[t][4 * q + 1], acc[t][4 * q + 2], acc[t][4 * q + 3]};
          if (n + 3 < p.N) {
            if (p.bias) { const float4 bs = *reinterpret_cast<const float4*>(p.bias + n); v[0] += bs.x; v[1] += bs.y; v[2] += bs.z; v[3] += bs.w; }
#pragma unroll
            for (int k = 0; k < 4; ++k) v[k] = v[k] >= 0.f ? v[k] : v[k] * p.slope;
            if (p.R1) { v[0] += rv[q].x; v[1] += rv[q].y; v[2] += rv[q].z; v[3] += rv[q].w; }
            if (p.Mk) {
              const float4 mk = mkv[t][q];
              v[0] *= mk.x > 0.f ? 1.f : p.mk_slope; v[1] *= mk.y > 0.f ? 1.f : p.mk_slope; v[2] *= mk.z > 0.f ? 1.f : p.mk_slope; v[3] *= mk.w > 0.f ? 1.f : p.mk_slope;
            }
            *reinterpret_cast<float4*>(Yp + m * ys + yc + n) = make_float4(v[0], v[1], v[2], v[3]);
          } else {   // ragged N (not a multiple of 4): element by element
            for (int k = 0; k < 4 && n + k < p.N; ++k) {
              float x = v[k] + (p.bias ? p.bias[n + k] : 0.f);
              x = x >= 0.f ? x : x * p.slope;
              if (p.R1) x += p.R1[m * p.r1_stride + p.r1_choff + n + k];
              if (p.Mk) x *= p.Mk[m * p.mk_stride + p.mk_choff + n + k] > 0.f ? 1.f : p.mk_slope;
              Yp[m * ys + yc + n + k] = x;
            }
          }
        }
      }
    }
    __syncthreads();   // everyone is done with sA before the next tile's rows overwrite it
  }
}

template <int K, int NB, int BMR, bool LN = false>
int launch_rowgemm(const RowGemmArgs& p, hipStream_t st) {
  constexpr int smem = (NB + BMR) * (K + 4) * 4;
  constexpr int per_cu = smem <= 78 * 1024 ? 2 : 1;
  static std::atomic<bool> attr_set[64];
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return LFSR_E_ARG;
  if (!attr_set[dev]) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_rowgemm<K, NB, BMR, LN>), hipFuncAttributeMaxDynamicSharedMemorySize, smem);
    if (e != hipSuccess) return LFSR_HIP_ERR(e);
    attr_set[dev] = true;
  }
  const long long ntiles = (p.M + BMR - 1) / BMR;
  const int nby = (p.N + NB - 1) / NB;
  int gx = 256 * per_cu / nby;              // fill every CU with the blocks its LDS admits
  if (gx > 8) gx &= ~7;                     // a multiple of 8: the nby panel blocks of a row tile land on one XCD (block id = x + gx y) and share its L2
  if (gx < 1) gx = 1;
  if (gx > ntiles) gx = (int)ntiles;
  hipLaunchKernelGGL((k_rowgemm<K, NB, BMR, LN>), dim3((unsigned)gx, (unsigned)nby), dim3(BMR * 4), smem, st, p);
  LFSR_CHECK_LAUNCH();
  return LFSR_OK;
}

}  // namespace

// returns LFSR_E_ARG when the shape is not covered (caller falls back to the gather-GEMM)
// the data gradient of a 64-output 1x1 conv whose input has N = 144 channels (DistgSSR fuse.0: dCAT = (dF W) . lrelu'(CAT)): one 192-column
// panel (the weight rows beyond N are zero-filled, stores beyond N are skipped), X streamed once, the mask applied in the epilogue
int lfsr_rowgemm_dgrad144_launch(const float* dy, int dy_stride, int dy_choff, const float* wT_packed, const float* mk, int mk_stride, int mk_choff, float mk_slope,
                                 float* dx, int dx_stride, int dx_choff, long long M, hipStream_t st) {
  if ((dy_stride | dy_choff | dx_stride | dx_choff) & 3 || (mk && ((mk_stride | mk_choff) & 3))) return LFSR_E_ARG;
  if (((uintptr_t)dy | (uintptr_t)dx | (uintptr_t)mk | (uintptr_t)wT_packed) & 15) return LFSR_E_ARG;
  if (M * (long long)dy_stride * 4 >= (1LL << 31) || M * (long long)dx_stride * 4 >= (1LL << 31)) return LFSR_E_ARG;
  {   // default: the three-term bf16 form (rowgemm_b3.hip, three 64-column panels of which the last holds 16 columns); LFSR_ROWGEMM=f32 / lfsr_set_arithmetic(f32): the kernel below
    const char* rsel = lfsr_sel("LFSR_ROWGEMM");
    const char* dsel = lfsr_sel("LFSR_DGRAD_PW");      // "f32": the fp32-MFMA kernel for this data gradient only (A/B runs; "gather": see bwd_ops.hip)
    if (mk && !(rsel && (rsel[0] == 'f' || rsel[0] == '1')) && !(dsel && dsel[0] == 'f') && !lfsr_arith_f32()) {
      const int rc = lfsr_rowgemm_b3_dgrad_launch(dy, dy_stride, dy_choff, wT_packed, mk, mk_stride, mk_choff, mk_slope, dx, dx_stride, dx_choff, M, 144, st);
      if (rc != LFSR_E_ARG) return rc;
    }
  }
  RowGemmArgs p{};
  p.X = dy; p.x_stride = dy_stride; p.x_choff = dy_choff; p.Wp = wT_packed; p.Mk = mk; p.mk_stride = mk_stride; p.mk_choff = mk_choff; p.mk_slope = mk_slope;
  p.Y = dx; p.y_stride = dx_stride; p.y_choff = dx_choff; p.M = M; p.N = 144; p.slope = 1.0f;
  return launch_rowgemm<64, 192, 64>(p, st);
}

// LayerNorm + projections in one launch (BasicTrans.forward, EPIT.py:113-121; SpaTrans / AngTrans, LFT.py:190-197 / :236-241): the weight rows n < ln_cols
// (q | k) see LayerNorm(x (+ pe)), the rows n >= ln_cols (v) see x itself; columns n >= split_n go to y2.  K = 64 or 128, N and split_n multiples of 64.
int lfsr_rowgemm_ln_launch(const float* x, int x_stride, int x_choff, int K, const float* w_packed, const float* ln_g, const float* ln_b, float ln_eps, int ln_cols,
                           const float* pe, int pe_stride, int pe_rows, int pe_div, float* y, int y_stride, int y_choff,
                           float* y2, int y2_stride, int y2_choff, int split_n, long long M, int N, hipStream_t st) {
  LfsrOpTimer op_t("linear_ln", K, N, st);
  if (!x || !w_packed || !ln_g || !ln_b || !y || M <= 0 || N <= 0 || N % 64 || ln_cols % 64 || (y2 && (split_n % 64 || split_n <= 0 || split_n >= N))) return LFSR_E_ARG;
  if ((x_stride | x_choff | y_stride | y_choff) & 3 || (y2 && ((y2_stride | y2_choff) & 3)) || (pe && ((pe_stride & 3) || pe_rows <= 0 || pe_div <= 0))) return LFSR_E_ARG;
  if (x_stride < x_choff + K || y_stride < y_choff + (y2 ? split_n : N) || (y2 && y2_stride < y2_choff + N - split_n)) return LFSR_E_ARG;
  if (((uintptr_t)y | (uintptr_t)y2 | (uintptr_t)x | (uintptr_t)pe | (uintptr_t)ln_g | (uintptr_t)ln_b) & 15) return LFSR_E_ARG;
  if ((M + 256) * (long long)x_stride * 4 >= (1LL << 31)) return LFSR_E_ARG;
  {   // default: the three-term bf16 form (rowgemm_b3.hip); LFSR_ROWGEMM=f32 keeps the fp32-MFMA kernel below (bit-identical to LayerNorm launch + fp32 row-GEMM)
    const char* rsel = lfsr_sel("LFSR_ROWGEMM");
    if (!(rsel && (rsel[0] == 'f' || rsel[0] == '1')) && !lfsr_arith_f32()) {
      const int rc = lfsr_rowgemm_b3_ln_launch(x, x_stride, x_choff, K, w_packed, ln_g, ln_b, ln_eps, ln_cols, pe, pe_stride, pe_rows, pe_div, y, y_stride, y_choff,
                                               y2, y2_stride, y2_choff, split_n, M, N, st);
      if (rc != LFSR_E_ARG) return rc;
    }
  }
  RowGemmArgs p{};
  p.X = x; p.x_stride = x_stride; p.x_choff = x_choff; p.Wp = w_packed; p.Y = y; p.y_stride = y_stride; p.y_choff = y_choff; p.M = M; p.N = N; p.slope = 1.0f;
  p.ln_g = ln_g; p.ln_b = ln_b; p.ln_eps = ln_eps; p.ln_cols = ln_cols; p.pe = pe; p.pe_stride = pe_stride; p.pe_rows = pe_rows; p.pe_div = pe_div;
  p.Y2 = y2; p.y2_stride = y2_stride; p.y2_choff = y2_choff; p.split_n = split_n;
  switch (K) {
    case 64: return launch_rowgemm<64, 64, 64, true>(p, st);
    case 128: return launch_rowgemm<128, 64, 64, true>(p, st);
    default: return LFSR_E_ARG;
  }
}

int lfsr_rowgemm_launch(const float* x, int x_stride, int x_choff, int K, const float* w_packed, const float* bias,
                        const float* res, int res_stride, int res_choff, float* y, int y_stride, int y_choff, long long M, int N, float slope, hipStream_t st) {
  if ((x_stride | x_choff) & 3 || N % 32) return LFSR_E_ARG;
  // the epilogue stores / loads 16 B per lane: every operand row and channel offset a multiple of four floats, 16-B aligned bases
  if ((y_stride | y_choff) & 3 || (res && ((res_stride | res_choff) & 3))) return LFSR_E_ARG;
  if (((uintptr_t)y | (uintptr_t)x | (uintptr_t)res | (uintptr_t)bias) & 15) return LFSR_E_ARG;
  if ((M + 256) * (long long)x_stride * 4 >= (1LL << 31)) return LFSR_E_ARG;   // 32-bit byte offsets into x (caller falls back to the gather-GEMM)
  RowGemmArgs p{};
  p.X = x; p.x_stride = x_stride; p.x_choff = x_choff; p.Wp = w_packed; p.bias = bias; p.R1 = res; p.r1_stride = res_stride; p.r1_choff = res_choff;
  p.Y = y; p.y_stride = y_stride; p.y_choff = y_choff; p.M = M; p.N = N; p.slope = slope;
  // 64-row tiles, 64-column panels: <= 76 KB of LDS -> two 256-thread blocks per CU, which is what keeps HBM loads in flight
  // while the other block runs its MFMAs
  if (N % 64) return LFSR_E_ARG;
  // LFSR_ROWGEMM=128 (N a multiple of 128: the transformer projections): 128-row x 128-column tiles, one 512-thread block per CU, each wave 32 rows
  // x 64 columns -- X streamed once per 128 output columns, a B fragment feeding two MFMA column tiles.  Measured on EPIT (B = 8, two runs each in
  // one call, profiles/r02_logs/ab_bench_lines.json: bench11_epit*.json): 637-639 patches/s against 659-660 for the 64 x 64 form -- one block per CU hides less HBM latency
  // than two; the 64 x 64 form stays the default
  const char* rsel = lfsr_sel("LFSR_ROWGEMM");
  // the bias-free K = 64 / 128 linears (the transformers' projections) run on the bf16 MFMA pipe with their fp32 operands split EXACTLY into three bf16 terms
  // (rowgemm_b3.hip; error against fp64 below this file's fp32-MFMA kernel: tools/b3_accuracy.py); LFSR_ROWGEMM=f32 keeps the fp32-MFMA form (A/B runs), 128 its wide tiles
  if (!(rsel && (rsel[0] == 'f' || rsel[0] == '1')) && !lfsr_arith_f32() && !bias && (K == 64 || K == 128 || K == 144)) {
    const int rc = lfsr_rowgemm_b3_launch(x, x_stride, x_choff, K, w_packed, res, res_stride, res_choff, y, y_stride, y_choff, M, N, slope, st);
    if (rc != LFSR_E_ARG) return rc;
  }
  const bool wide = N % 128 == 0 && rsel && rsel[0] == '1';
  switch (K) {
    case 64: return wide ? launch_rowgemm<64, 128, 128>(p, st) : launch_rowgemm<64, 64, 64>(p, st);
    case 128: return wide ? launch_rowgemm<128, 128, 128>(p, st) : launch_rowgemm<128, 64, 64>(p, st);
    case 144: return launch_rowgemm<144, 64, 64>(p, st);
    default: return LFSR_E_ARG;
  }
}

extern "C" int lfsr_linear_ln_fwd(const float* x, int x_stride, int x_choff, int K, const float* w_packed, const float* gamma, const float* beta,
                                  float eps, int ln_cols, const float* pe, int pe_stride, int pe_rows, int pe_div,
                                  float* y, int y_stride, int y_choff, float* y2, int y2_stride, int y2_choff, int split_n,
                                  long long M, int N, void* stream) {
  return lfsr_rowgemm_ln_launch(x, x_stride, x_choff, K, w_packed, gamma, beta, eps, ln_cols, pe, pe_stride, pe_rows, pe_div, y, y_stride, y_choff,
                                y2, y2_stride, y2_choff, split_n, M, N, lfsr_stream(stream));
}
