// Batched weight repack: one launch per pack kind, driven by a device-side descriptor table (LfsrPackDesc, lfsr_internal.h).
// The index maps are those of k_pack_weight (gemm_gather.hip), k_pack_weight_T (wgrad.hip) and k_pack_chunkT (bwd_ops.hip): reference layouts
// (O, C, kh, kw) of train.py's state_dict -> the packed layouts the kernels read.
#include "lfsr_internal.h"

namespace {

__global__ __launch_bounds__(256) void k_pack_generic_batch(const LfsrPackDesc* __restrict__ tab) {
  const LfsrPackDesc d = tab[blockIdx.y];
  const float* __restrict__ w = d.src;
  float* __restrict__ out = d.dst;
  const long long stride = (long long)gridDim.x * blockDim.x;
  if (d.kind == 0) {          // out[(t * Npad + n) * C + c] = n < O ? w[(nref * C + c) * T + t] : 0
    const long long total = (long long)d.T * d.Npad * d.C;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
      const int c = (int)(i % d.C);
      const long long t2 = i / d.C;
      const int n = (int)(t2 % d.Npad), t = (int)(t2 / d.Npad);
      float v = 0.f;
      if (n < d.O) {
        int nref = n;
        if (d.perm == 1) { const int r2 = d.O / d.ch, q = n / d.ch, cc = n - q * d.ch; nref = cc * r2 + q; }
        v = w[((long long)nref * d.C + c) * d.T + t];
      }
      out[i] = v;
    }
  } else if (d.kind == 1) {   // out[(tp * Cpad + k) * O + n] = k < C ? w[(n * C + k) * T + t] : 0,  t = flip ? T - 1 - tp : tp   (Npad holds Cpad)
    const long long total = (long long)d.T * d.Npad * d.O;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
      const int n = (int)(i % d.O);
      const long long t2 = i / d.O;
      const int k = (int)(t2 % d.Npad), tp = (int)(t2 / d.Npad);
      const int t = d.flip ? d.T - 1 - tp : tp;
      out[i] = k < d.C ? w[((long long)n * d.C + k) * d.T + t] : 0.f;
    }
  } else {                    // chunkT: out[(q * Cpad + k) * ch + c] = k < C ? w[nref * C + k] : 0,  nref = perm ? c * r2 + q : q * ch + c
    const int r2 = d.O / d.ch;
    const long long total = (long long)r2 * d.Npad * d.ch;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
      const int c = (int)(i % d.ch);
      const long long t = i / d.ch;
      const int k = (int)(t % d.Npad), q = (int)(t / d.Npad);
      const int nref = d.perm ? c * r2 + q : q * d.ch + c;
      out[i] = k < d.C ? w[(long long)nref * d.C + k] : 0.f;
    }
  }
}

}  // namespace

int lfsr_pack_generic_batch(const LfsrPackDesc* table_dev, int n, hipStream_t st) {
  if (!table_dev || n <= 0) return n == 0 ? LFSR_OK : LFSR_E_ARG;
  hipLaunchKernelGGL(k_pack_generic_batch, dim3(8, (unsigned)n), dim3(256), 0, st, table_dev);
  LFSR_CHECK_LAUNCH();
  return LFSR_OK;
}
