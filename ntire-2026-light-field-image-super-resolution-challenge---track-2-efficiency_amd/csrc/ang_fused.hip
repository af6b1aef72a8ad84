// Fused AngConv branch (model/SR/DistgSSR.py:84-90) on VCL, one launch:
//   t = lrelu(conv AxA stride A, 64 -> 16)  -- contracts each macro-pixel: K = A*A views x 64 channels
//   y = lrelu(1x1 16 -> 16*A*A), PixelShuffle(A) -- scattered to the A*A views' 16-channel slice of the concat buffer
// The branch is HBM-bound (AI ~ 8 FLOP/B: it streams every view's 64 channels once and writes 16 per view), so the kernel is
// built around bytes in flight, not MFMA rate: a wave owns 16 macro-pixels (the 16 A-rows of v_mfma_f32_16x16x4_f32, N = 16
// exactly -- no padded columns), takes its A fragments STRAIGHT from global memory (lane (i, kq) loads the 16 B of channels
// 16j + 4kq .. +3 of its pixel in view `tap`; k order permuted identically for A and W), one row of A views ahead, 8 waves per
// CU; both weight matrices sit in LDS for the life of the block (109 KB + 26 KB at A = 5), there is no barrier after the preamble.
// The 16x16 result is LeakyReLU'd, saved (the backward needs it), turned into A-operand order through a 1-KB wave-private LDS
// tile, multiplied by the 1x1 weights view by view and stored 64 B per pixel and view.
#include "lfsr_internal.h"

typedef float f32x4a __attribute__((ext_vector_type(4)));

namespace {

constexpr int W1ROW = 68;   // floats per (tap, n) row of W1 in LDS: 272 B -> the 16 n-rows of a fragment read hit 16 distinct 16-B slots
constexpr int TROW = 20;    // wave-private t tile row stride

struct AngArgs {
  const float* X; int x_stride; int x_choff;
  const float* W1;   // [A*A][32][64] (lfsr_pack_conv_weight, perm 0; rows n >= 16 are padding)
  const float* W2;   // [>= 16*A*A][16] (perm 1, ch 16: row n' = view * 16 + c)
  float* T;          // (B*h*w, 16) post-LeakyReLU stage-1 activations
  float* Y; int y_stride; int y_choff;
  int B, A, H, W;
  float slope;
};

__global__ __launch_bounds__(512) void k_ang_fused(AngArgs p) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int AA = p.A * p.A, HW = p.H * p.W, M = p.B * HW;
  float* sW1 = sm;                               // [AA][16][W1ROW]
  float* sW2 = sm + AA * 16 * W1ROW;             // [16*AA][16]
  float* sT = sW2 + 16 * AA * 16;                // [waves][16][TROW]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nwave = blockDim.x >> 6;
  const int i = lane & 15, kq = lane >> 4;

  const int ngroups = (M + 15) / 16, gstride = (int)gridDim.x * nwave;
  const int g0 = (int)blockIdx.x * nwave + wave;
  const long long tstep = (long long)HW * p.x_stride;
  float4 fb[2][5][4];   // two register sets, alternated by the (fully unrolled) row loop: no copies, the loads of row u+1 stay in flight
  // fragment loads of one row of views of group g (macro-pixel m = 16 g + i, view `tap` -> VCL pixel (b*AA + tap)*HW + yx)
  auto load_row = [&](int g, int u, float4 (*f)[4]) {
    const int m = g * 16 + i;
    const bool ok = m < M;
    const int mb = ok ? m / HW : 0, myx = ok ? m - mb * HW : 0;
    const float* src = p.X + ((long long)mb * AA * HW + myx) * p.x_stride + p.x_choff + 4 * kq;
#pragma unroll
    for (int v = 0; v < 5; ++v)
#pragma unroll
      for (int j = 0; j < 4; ++j)
        f[v][j] = (ok && v < p.A) ? *reinterpret_cast<const float4*>(src + (u * p.A + v) * tstep + 16 * j) : make_float4(0.f, 0.f, 0.f, 0.f);
  };
  {   // W1: 16 float4 per (tap, n) row; all of a thread's loads are issued before its first LDS store (one L2 round trip, not 13); the
      // first group's first row of views is requested in between, so its HBM latency runs under the weight staging (at B = 32 a wave owns
      // exactly one group: the preamble used to sit serially in front of every byte the wave streams)
    float4 wv[13];
#pragma unroll
    for (int q = 0; q < 13; ++q) {
      const int idx = tid + q * 512, row = idx >> 4, c = idx & 15, tap = row >> 4, n = row & 15;
      wv[q] = idx < AA * 256 ? *reinterpret_cast<const float4*>(p.W1 + ((long long)tap * 32 + n) * 64 + c * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    float4 w2v[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int idx = tid + q * 512;
      w2v[q] = idx < AA * 64 ? reinterpret_cast<const float4*>(p.W2)[idx] : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    if (g0 < ngroups) load_row(g0, 0, fb[0]);
#pragma unroll
    for (int q = 0; q < 13; ++q) {
      const int idx = tid + q * 512;
      if (idx < AA * 256) *reinterpret_cast<float4*>(sW1 + (idx >> 4) * W1ROW + (idx & 15) * 4) = wv[q];
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int idx = tid + q * 512;
      if (idx < AA * 64) reinterpret_cast<float4*>(sW2)[idx] = w2v[q];
    }
  }
  __syncthreads();

  float* st = sT + wave * 16 * TROW;
  const float* b1 = sW1 + i * W1ROW + 4 * kq;     // + tap * 16 * W1ROW + 16 j
  const float* b2 = sW2 + i * 16 + 4 * kq;        // + view * 256

  for (int g = g0; g < ngroups; g += gstride) {
    const int m = g * 16 + i;
    const bool ok = m < M;
    const int mb = ok ? m / HW : 0, myx = ok ? m - mb * HW : 0;
    // the views are taken A at a time (one row of views): the next A views' fragments (A x 64 B per lane, 20 KB per wave at A = 5)
    // are in flight while the current ones feed the MFMAs -- with 8 such waves a CU keeps ~160 KB of HBM reads outstanding
    if (g != g0) load_row(g, 0, fb[0]);
    f32x4a acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int u = 0; u < 5; ++u) {
      if (u >= p.A) break;
      if (u + 1 < p.A) load_row(g, u + 1, fb[(u + 1) & 1]);
#pragma unroll
      for (int v = 0; v < 5; ++v) {
        if (v >= p.A) break;
        const float* bt = b1 + (u * p.A + v) * 16 * W1ROW;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float4 b = *reinterpret_cast<const float4*>(bt + 16 * j);
          const float4 a = fb[u & 1][v][j];
          acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, b.x, acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, b.y, acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, b.z, acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, b.w, acc, 0, 0, 0);
        }
      }
    }
    // C layout: row (macro-pixel) = 4 (lane >> 4) + reg, column (channel) = lane & 15.  Everything that leaves the wave goes
    // through the wave-private tile so that a lane stores 16 B (row i, channels 4 kq .. + 3): one store instruction per view
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      float v = acc[r];
      v = v >= 0.f ? v : v * p.slope;
      st[(4 * kq + r) * TROW + i] = v;
    }
    const int b_i = ok ? mb : 0;
    const long long dbase = ok ? (long long)b_i * AA * HW + myx : -1;   // destination pixel of row i, view 0
    __builtin_amdgcn_wave_barrier();
    const float4 a2 = *reinterpret_cast<const float4*>(st + i * TROW + 4 * kq);   // A operand: row i, k = 4 kq .. + 3
    if (ok) *reinterpret_cast<float4*>(p.T + (long long)m * 16 + 4 * kq) = a2;
    __builtin_amdgcn_wave_barrier();
    for (int view = 0; view < AA; ++view) {
      const float4 b = *reinterpret_cast<const float4*>(b2 + view * 256);
      f32x4a o = {0.f, 0.f, 0.f, 0.f};
      o = __builtin_amdgcn_mfma_f32_16x16x4f32(a2.x, b.x, o, 0, 0, 0);
      o = __builtin_amdgcn_mfma_f32_16x16x4f32(a2.y, b.y, o, 0, 0, 0);
      o = __builtin_amdgcn_mfma_f32_16x16x4f32(a2.z, b.z, o, 0, 0, 0);
      o = __builtin_amdgcn_mfma_f32_16x16x4f32(a2.w, b.w, o, 0, 0, 0);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float v = o[r];
        v = v >= 0.f ? v : v * p.slope;
        st[(4 * kq + r) * TROW + i] = v;
      }
      __builtin_amdgcn_wave_barrier();
      const float4 z = *reinterpret_cast<const float4*>(st + i * TROW + 4 * kq);
      __builtin_amdgcn_wave_barrier();
      if (ok) *reinterpret_cast<float4*>(p.Y + (dbase + (long long)view * HW) * p.y_stride + p.y_choff + 4 * kq) = z;
    }
  }
}

}  // namespace

size_t lfsr_ang_fused_smem(int A, int waves) {
  const int AA = A * A;
  return (size_t)(AA * 16 * W1ROW + 16 * AA * 16 + waves * 16 * TROW) * 4;
}

bool lfsr_ang_fused_ok(int A) { return A >= 1 && A <= 5 && lfsr_ang_fused_smem(A, 8) <= 160 * 1024; }

int lfsr_ang_fused_launch(const float* x, int x_stride, int x_choff, const float* w1_packed, const float* w2_packed, float* t, float* y,
                          int y_stride, int y_choff, int B, int A, int h, int w, float slope, hipStream_t st) {
  if (!lfsr_ang_fused_ok(A)) return LFSR_E_ARG;
  static std::atomic<bool> attr_set[64];
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return LFSR_E_ARG;
  if (!attr_set[dev]) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_ang_fused), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return LFSR_HIP_ERR(e);
    attr_set[dev] = true;
  }
  AngArgs p{};
  p.X = x; p.x_stride = x_stride; p.x_choff = x_choff; p.W1 = w1_packed; p.W2 = w2_packed; p.T = t;
  p.Y = y; p.y_stride = y_stride; p.y_choff = y_choff; p.B = B; p.A = A; p.H = h; p.W = w; p.slope = slope;
  const long long ngroups = ((long long)B * h * w + 15) / 16;
  const int waves = 8;
  long long grid = (ngroups + waves - 1) / waves;
  if (grid > 256) grid = 256;
  hipLaunchKernelGGL(k_ang_fused, dim3((unsigned)grid), dim3(waves * 64), lfsr_ang_fused_smem(A, waves), st, p);
  LFSR_CHECK_LAUNCH();
  return LFSR_OK;
}
