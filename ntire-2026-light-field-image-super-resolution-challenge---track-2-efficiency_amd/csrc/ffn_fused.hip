// Fused transformer feed-forward block on token rows (EPIT.py:84-90,126 / LFT.py:151-156,202 / LFT.py:216-221,243):
//   y = res + W2 . relu(W1 . xn)          xn = LayerNorm'd tokens (K1), hidden H = 2 K1, output N2 = K1, no biases
// As two GEMM launches the (M x H) hidden activations make a round trip through HBM (210 MB out + 210 MB back at EPIT's
// BASELINE geometry, more than the layer's own input and output); here they never leave the CU.  The hidden dimension is walked
// in chunks of 32: a wave keeps the A fragments of its 32 token rows in registers (straight from global memory, lane = row), runs
// GEMM 1 against the chunk's W1 rows (32 x K1, LDS) into ONE accumulator tile, applies ReLU, turns the 32 x 32 tile into A-operand
// order through a wave-private LDS tile and immediately runs GEMM 2 against the chunk's W2 columns (N2 x 32, LDS) into the N2/32
// output accumulators that live across all chunks.  The two weight slices of the next chunk are fetched into registers at the
// start of a chunk and written to the other LDS buffer at its end: one barrier per chunk, 128 MFMAs per wave between barriers
// (K1 = N2 = 128), 0.28 LDS reads per MFMA.
#include <stdlib.h>

#include "lfsr_internal.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

namespace {

constexpr int FOOB = (int)0x80000000u;   // byte offset beyond every descriptor: loads return 0, stores are dropped

struct FfnArgs {
  const float* X; int x_stride; int x_choff;    // normalised tokens (M, K1)
  const float* W1;                              // [H][K1]  (lfsr_pack_conv_weight, taps = 1)
  const float* W2;                              // [N2][H]
  const float* R; int r_stride; int r_choff;    // residual (M, N2) or null
  float* Y; int y_stride; int y_choff;
  long long M; int H;
  int y_bytes, r_bytes;                         // true byte spans (descriptor extents; 0 = absent)
  float slope;                                  // hidden activation: 0 = ReLU
  const float* ln_g; const float* ln_b; float ln_eps;   // non-null: X holds the RAW tokens and xn = LayerNorm(X) is formed in registers (feed_forward.0)
};

template <int K1, int N2>
__global__ __launch_bounds__(512) void k_ffn_fused(FfnArgs p) {
  constexpr int R1 = K1 + 4, R2 = 36, TR = 36;                 // LDS row strides (floats): conflict-free ds_read_b128
  constexpr int KJ = K1 / 8, NT2 = N2 / 32;
  constexpr int W1F4 = 32 * K1 / 4, W2F4 = N2 * 8;             // float4 per chunk slice
  constexpr int W1L = W1F4 / 512, W2L = W2F4 / 512;            // per thread
  static_assert(W1F4 % 512 == 0 && W2F4 % 512 == 0, "slice sizes are multiples of the block");
  constexpr int BUF = 32 * R1 + N2 * R2;                       // floats per weight buffer
  extern __shared__ __attribute__((aligned(16))) float sm[];
  float* sT = sm + 2 * BUF;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int half = lane >> 5, l31 = lane & 31;
  const int nch = p.H / 32;

  // this block's contiguous range of 32-row groups, walked in rounds of <= 8 groups (one per wave), rounds balanced
  const long long gtot = (p.M + 31) / 32;
  const long long gbase = gtot / gridDim.x, grem = gtot % gridDim.x;
  const long long gn = gbase + ((long long)blockIdx.x < grem ? 1 : 0);
  const long long gstart = (long long)blockIdx.x * gbase + ((long long)blockIdx.x < grem ? (long long)blockIdx.x : grem);
  const int rounds = (int)((gn + 7) / 8);
  const int gpr = rounds ? (int)((gn + rounds - 1) / rounds) : 0;
  if (rounds == 0) return;

  float4 w1r[W1L], w2r[W2L];
  auto fetch_chunk = [&](int c) {
#pragma unroll
    for (int i = 0; i < W1L; ++i) {
      const int idx = tid + 512 * i, r = idx / (K1 / 4), q = idx - r * (K1 / 4);
      w1r[i] = *reinterpret_cast<const float4*>(p.W1 + ((long long)c * 32 + r) * K1 + q * 4);
    }
#pragma unroll
    for (int i = 0; i < W2L; ++i) {
      const int idx = tid + 512 * i, n = idx >> 3, q = idx & 7;
      w2r[i] = *reinterpret_cast<const float4*>(p.W2 + (long long)n * p.H + c * 32 + q * 4);
    }
  };
  auto store_chunk = [&](float* buf) {
#pragma unroll
    for (int i = 0; i < W1L; ++i) {
      const int idx = tid + 512 * i, r = idx / (K1 / 4), q = idx - r * (K1 / 4);
      *reinterpret_cast<float4*>(buf + r * R1 + q * 4) = w1r[i];
    }
#pragma unroll
    for (int i = 0; i < W2L; ++i) {
      const int idx = tid + 512 * i, n = idx >> 3, q = idx & 7;
      *reinterpret_cast<float4*>(buf + 32 * R1 + n * R2 + q * 4) = w2r[i];
    }
  };

  fetch_chunk(0);
  store_chunk(sm);
  __syncthreads();

  float* st = sT + wave * 32 * TR;
  const __amdgpu_buffer_rsrc_t rsY = __builtin_amdgcn_make_buffer_rsrc(p.Y, 0, p.y_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsR = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.R), 0, p.r_bytes, 0x00020000);   // null residual: loads return 0
  float4 xa[KJ];
  f32x16 accy[NT2];
  int step = 0;
  for (int rd = 0; rd < rounds; ++rd) {
    const long long g = gstart + (long long)rd * gpr + wave;
    const bool active = wave < gpr && g < gstart + gn;
    const long long m0 = g * 32;
    if (active) {
      const long long m = m0 + l31;
      const float* src = p.X + (m < p.M ? m : p.M - 1) * p.x_stride + p.x_choff + 4 * half;
#pragma unroll
      for (int j = 0; j < KJ; ++j) xa[j] = *reinterpret_cast<const float4*>(src + 8 * j);
      // make the compiler retire these loads HERE: left pending across the chunk loop's header, its in-order vmcnt bookkeeping
      // would wait for every younger load too -- i.e. for the weight prefetch of the next chunk -- inside each chunk's GEMM 1
#pragma unroll
      for (int j = 0; j < KJ; ++j) asm volatile("" :: "v"(xa[j].x), "v"(xa[j].y), "v"(xa[j].z), "v"(xa[j].w));
      if (p.ln_g) {
        // a token row lives in lanes l31 and l31 + 32 (alternating groups of four channels): two-pass LayerNorm with one cross-half exchange per pass
        float sm = 0.f;
#pragma unroll
        for (int j = 0; j < KJ; ++j) sm += (xa[j].x + xa[j].y) + (xa[j].z + xa[j].w);
        sm += __shfl_xor(sm, 32);
        const float mu = sm * (1.0f / K1);
        float q2 = 0.f;
#pragma unroll
        for (int j = 0; j < KJ; ++j) {
          xa[j].x -= mu; xa[j].y -= mu; xa[j].z -= mu; xa[j].w -= mu;
          q2 += (xa[j].x * xa[j].x + xa[j].y * xa[j].y) + (xa[j].z * xa[j].z + xa[j].w * xa[j].w);
        }
        q2 += __shfl_xor(q2, 32);
        const float rstd = 1.0f / sqrtf(q2 * (1.0f / K1) + p.ln_eps);
#pragma unroll
        for (int j = 0; j < KJ; ++j) {
          const float4 gv = *reinterpret_cast<const float4*>(p.ln_g + 8 * j + 4 * half);
          const float4 bv = *reinterpret_cast<const float4*>(p.ln_b + 8 * j + 4 * half);
          xa[j] = make_float4(xa[j].x * rstd * gv.x + bv.x, xa[j].y * rstd * gv.y + bv.y, xa[j].z * rstd * gv.z + bv.z, xa[j].w * rstd * gv.w + bv.w);
        }
#pragma unroll
        for (int j = 0; j < KJ; ++j) asm volatile("" :: "v"(xa[j].x), "v"(xa[j].y), "v"(xa[j].z), "v"(xa[j].w));
      }
    }
#pragma unroll
    for (int t = 0; t < NT2; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) accy[t][r] = 0.f;

    for (int c = 0; c < nch; ++c, ++step) {
      const float* buf = sm + (step & 1) * BUF;
      const bool last_step = rd == rounds - 1 && c == nch - 1;
      if (!last_step) fetch_chunk(c + 1 < nch ? c + 1 : 0);   // flies under this chunk's MFMAs
      if (active) {
        // GEMM 1: hidden tile (32 rows x 32 hidden units of chunk c)
        f32x16 h;
#pragma unroll
        for (int r = 0; r < 16; ++r) h[r] = 0.f;
        const float* b1 = buf + l31 * R1 + 4 * half;
#pragma unroll
        for (int j = 0; j < KJ; ++j) {
          const float4 b = *reinterpret_cast<const float4*>(b1 + 8 * j);
          h = __builtin_amdgcn_mfma_f32_32x32x2f32(xa[j].x, b.x, h, 0, 0, 0);
          h = __builtin_amdgcn_mfma_f32_32x32x2f32(xa[j].y, b.y, h, 0, 0, 0);
          h = __builtin_amdgcn_mfma_f32_32x32x2f32(xa[j].z, b.z, h, 0, 0, 0);
          h = __builtin_amdgcn_mfma_f32_32x32x2f32(xa[j].w, b.w, h, 0, 0, 0);
        }
        // activation, then C layout (row = (r&3) + 8(r>>2) + 4 half, col = lane & 31) -> A-operand order through the wave tile
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          float v = h[r];
          v = v >= 0.f ? v : v * p.slope;
          st[((r & 3) + 8 * (r >> 2) + 4 * half) * TR + l31] = v;
        }
        __builtin_amdgcn_wave_barrier();
        float4 a2[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) a2[j] = *reinterpret_cast<const float4*>(st + l31 * TR + 8 * j + 4 * half);
        __builtin_amdgcn_wave_barrier();
        // GEMM 2: output accumulators += hidden chunk x W2[:, chunk]
        const float* b2 = buf + 32 * R1 + l31 * R2 + 4 * half;
#pragma unroll
        for (int t = 0; t < NT2; ++t) {
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const float4 b = *reinterpret_cast<const float4*>(b2 + t * 32 * R2 + 8 * j);
            accy[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a2[j].x, b.x, accy[t], 0, 0, 0);
            accy[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a2[j].y, b.y, accy[t], 0, 0, 0);
            accy[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a2[j].z, b.z, accy[t], 0, 0, 0);
            accy[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a2[j].w, b.w, accy[t], 0, 0, 0);
          }
        }
      }
      if (!last_step) store_chunk(sm + ((step + 1) & 1) * BUF);
      __syncthreads();
    }
    // epilogue: + residual, store from the accumulators (32 consecutive channels of two rows per instruction).  Buffer
    // addressing: the row goes into the scalar offset, the per-lane part (row-half, channel) is ONE register that is never
    // rewritten -- with per-element 64-bit addresses in recycled registers every store had to wait for the previous one to
    // retire (vmcnt(0) in front of each) -- and a tile's residual loads are all issued before their first use
    if (active) {
      const int m0i = (int)m0, Mi = (int)p.M;
      const bool full = m0i + 32 <= Mi;
#pragma unroll
      for (int t = 0; t < NT2; ++t) {
        const int n = t * 32 + l31;
        const int yv = (4 * half * p.y_stride + p.y_choff + n) * 4, rvo = (4 * half * p.r_stride + p.r_choff + n) * 4;
        float rv[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int ri = (r & 3) + 8 * (r >> 2);
          const bool ok = full || m0i + ri + 4 * half < Mi;
          rv[r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsR, ok ? rvo : FOOB, __builtin_amdgcn_readfirstlane((m0i + ri) * p.r_stride * 4), 0));
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int ri = (r & 3) + 8 * (r >> 2);
          const bool ok = full || m0i + ri + 4 * half < Mi;
          __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, accy[t][r] + rv[r]), rsY, ok ? yv : FOOB,
                                                __builtin_amdgcn_readfirstlane((m0i + ri) * p.y_stride * 4), 0);
        }
      }
    }
  }
}

template <int K1, int N2>
int launch_ffn(const FfnArgs& p, hipStream_t st) {
  constexpr int smem = (2 * (32 * (K1 + 4) + N2 * 36) + 8 * 32 * 36) * 4;
  static std::atomic<bool> attr_set[64];
  static std::atomic<int> cus[64];
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return LFSR_E_ARG;
  if (!attr_set[dev]) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_ffn_fused<K1, N2>), hipFuncAttributeMaxDynamicSharedMemorySize, smem);
    if (e != hipSuccess) return LFSR_HIP_ERR(e);
    int v = 0;
    cus[dev] = (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) ? v : 256;
    attr_set[dev] = true;
  }
  const long long groups = (p.M + 31) / 32;
  long long grid = cus[dev];
  if (grid > (groups + 7) / 8) grid = (groups + 7) / 8;
  hipLaunchKernelGGL((k_ffn_fused<K1, N2>), dim3((unsigned)grid), dim3(512), smem, st, p);
  LFSR_CHECK_LAUNCH();
  return LFSR_OK;
}

}  // namespace

// y = res + W2 . act(W1 . LayerNorm(x)) when ln_g / ln_b are given (x the raw tokens), else the plain form
int lfsr_ffn_ln_launch(const float* x, int x_stride, int x_choff, const float* ln_g, const float* ln_b, float ln_eps, const float* w1_packed, const float* w2_packed,
                       const float* res, int res_stride, int res_choff, float* y, int y_stride, int y_choff,
                       long long M, int K1, int H, int N2, float slope, hipStream_t st, const void* wsplit) {
  LfsrOpTimer op_t("ffn", K1, H, st);
  if (!x || !w1_packed || !w2_packed || !y || M <= 0 || H <= 0 || H % 32 || (x_stride | x_choff) & 3) return LFSR_E_ARG;
  if ((ln_g != nullptr) != (ln_b != nullptr)) return LFSR_E_ARG;
  {   // default: the three-term bf16 form (ffn_b3.hip); LFSR_FFN=f32 keeps this file's fp32-MFMA kernel (A/B runs)
    const char* fsel = lfsr_sel("LFSR_FFN");
    if (!(fsel && fsel[0] == 'f') && !lfsr_arith_f32()) {
      const int rc = lfsr_ffn_b3_launch(x, x_stride, x_choff, ln_g, ln_b, ln_eps, w1_packed, w2_packed, res, res_stride, res_choff, y, y_stride, y_choff, M, K1, H, N2, slope, st, wsplit);
      if (rc != LFSR_E_ARG) return rc;
    }
  }
  if (x_stride < x_choff + K1 || y_stride < y_choff + N2 || (res && res_stride < res_choff + N2)) return LFSR_E_ARG;
  if (M * (long long)(y_stride > res_stride ? y_stride : res_stride) * 4 >= (1LL << 31)) return LFSR_E_ARG;   // 32-bit byte offsets in the epilogue
  FfnArgs p{};
  p.X = x; p.x_stride = x_stride; p.x_choff = x_choff; p.W1 = w1_packed; p.W2 = w2_packed;
  p.R = res; p.r_stride = res_stride; p.r_choff = res_choff; p.Y = y; p.y_stride = y_stride; p.y_choff = y_choff;
  p.M = M; p.H = H; p.slope = slope; p.ln_g = ln_g; p.ln_b = ln_b; p.ln_eps = ln_eps;
  p.y_bytes = (int)(M * y_stride * 4); p.r_bytes = res ? (int)(M * res_stride * 4) : 0;
  if ((ln_g != nullptr) != (ln_b != nullptr) || (((uintptr_t)ln_g | (uintptr_t)ln_b) & 15)) return LFSR_E_ARG;
  if (K1 == 128 && N2 == 128) return launch_ffn<128, 128>(p, st);
  if (K1 == 64 && N2 == 64) return launch_ffn<64, 64>(p, st);
  return LFSR_E_ARG;
}

extern "C" int lfsr_ffn_fwd(const float* x, int x_stride, int x_choff, const float* w1_packed, const float* w2_packed,
                            const float* res, int res_stride, int res_choff, float* y, int y_stride, int y_choff,
                            long long M, int K1, int H, int N2, float slope, void* stream) {
  return lfsr_ffn_ln_launch(x, x_stride, x_choff, nullptr, nullptr, 0.f, w1_packed, w2_packed, res, res_stride, res_choff, y, y_stride, y_choff, M, K1, H, N2, slope,
                            lfsr_stream(stream));
}

extern "C" int lfsr_ffn_ln_fwd(const float* x, int x_stride, int x_choff, const float* gamma, const float* beta, float eps,
                               const float* w1_packed, const float* w2_packed, const float* res, int res_stride, int res_choff,
                               float* y, int y_stride, int y_choff, long long M, int K1, int H, int N2, float slope, void* stream) {
  if (!gamma || !beta) return LFSR_E_ARG;
  return lfsr_ffn_ln_launch(x, x_stride, x_choff, gamma, beta, eps, w1_packed, w2_packed, res, res_stride, res_choff, y, y_stride, y_choff, M, K1, H, N2, slope,
                            lfsr_stream(stream));
}
