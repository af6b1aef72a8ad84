// LF_InterNet forward (model/SR/LF_InterNet.py:33-141) on VCL buffers: host driver + its three small kernels.
// The network keeps two feature maps: the "spatial" one (per view pixel, VCL) and the "angular" one (per LR pixel,
// rows (b,y,x)).  Both are kept as channels 0..63 of 128-wide rows whose channels 64..127 receive the other branch's
// contribution, so the reference's torch.cat((x, other), 1) before every squeeze conv (:62-63) is free:
//   Spa2Ang (AxA stride A, 64->64)  = IN_ANG gather-GEMM, ReLU, into the angular rows' upper half
//   Ang2Spa (1x1 64->A*A*64 + PixelShuffle(A)) = pointwise GEMM with OUT_VIEWS scatter into the spatial rows' upper half
//   AngConvSq (1x1 128->64)         = row-wise GEMM over the 128-wide angular rows
//   SpaConvSq (dilated 3x3 128->64) = per-view 3x3 gather-GEMM over the 128-wide spatial rows
// ReconBlock's PreConv -> MacPI2SAI -> PixelShuffle(s) -> FinalConv chain is linear and is folded to one 3x3 conv 64->s^2
// whose epilogue scatters straight into the HR SAI mosaic.
#include "gemm_gather_kernel.h"
#include "param_table.h"

namespace {

// AngFE (LF_InterNet.py:24-25): conv AxA stride A, 1 -> 64, on the MacPI of the input == per LR pixel, a 64 x A^2 matvec
// over the A^2 views of the SAI mosaic.  16 threads per LR pixel, 4 channels each.
__global__ __launch_bounds__(256) void k_angfe(const float* __restrict__ x, const float* __restrict__ w, float* __restrict__ y, int y_stride, int y_choff,
                                              int B, int A, int h, int wd) {
  extern __shared__ float sw[];   // [64][A*A]
  const int AA = A * A;
  for (int i = threadIdx.x; i < 64 * AA; i += 256) sw[i] = w[i];
  __syncthreads();
  const long long nlr = (long long)B * h * wd;
  const int Wm = A * wd, Hm = A * h;
  for (long long g = (long long)blockIdx.x * 256 + threadIdx.x; g < nlr * 16; g += (long long)gridDim.x * 256) {
    long long pix = g >> 4;
    int c4 = (int)(g & 15) * 4;
    int xx = (int)(pix % wd);
    long long t = pix / wd;
    int yy = (int)(t % h);
    int b = (int)(t / h);
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    for (int u = 0; u < A; ++u)
      for (int v = 0; v < A; ++v) {
        float xv = x[(long long)b * Hm * Wm + (long long)(u * h + yy) * Wm + v * wd + xx];
        int k = u * A + v;
        a0 = fmaf(xv, sw[(c4 + 0) * AA + k], a0); a1 = fmaf(xv, sw[(c4 + 1) * AA + k], a1);
        a2 = fmaf(xv, sw[(c4 + 2) * AA + k], a2); a3 = fmaf(xv, sw[(c4 + 3) * AA + k], a3);
      }
    *reinterpret_cast<float4*>(y + pix * y_stride + y_choff + c4) = make_float4(a0, a1, a2, a3);
  }
}

// copy a 64-channel slice between row-major buffers (the torch.cat of the per-block outputs, LF_InterNet.py:97-104)
__global__ __launch_bounds__(256) void k_copy64(const float* __restrict__ src, int s_stride, int s_choff, float* __restrict__ dst, int d_stride, int d_choff, long long M) {
  for (long long g = (long long)blockIdx.x * 256 + threadIdx.x; g < M * 16; g += (long long)gridDim.x * 256) {
    long long r = g >> 4;
    int c4 = (int)(g & 15) * 4;
    *reinterpret_cast<float4*>(dst + r * d_stride + d_choff + c4) = *reinterpret_cast<const float4*>(src + r * s_stride + s_choff + c4);
  }
}

// wf[tap][ij (pad 32)][k] = sum_c wfinal[c] * wpre[c*s2 + ij][k][tap]   (fp64 accumulation)
__global__ void k_fold_recon(const float* __restrict__ wpre, const float* __restrict__ wfin, float* __restrict__ wf, int s2) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;   // over 9*32*64
  if (i >= 9 * 32 * 64) return;
  int k = i & 63, ij = (i >> 6) & 31, tap = i >> 11;
  double a = 0.0;
  if (ij < s2)
    for (int c = 0; c < 64; ++c) a += (double)wfin[c] * (double)wpre[(((long long)c * s2 + ij) * 64 + k) * 9 + tap];
  wf[i] = (float)a;
}

inline unsigned cap_grid(long long total, unsigned cap = 8192) {
  unsigned g = lfsr_blocks(total, 256);
  return g > cap ? cap : g;
}

}  // namespace

struct lfsr_internet {
  int A, s, ngroups, nlayers;
  LfsrParamTable P;
  size_t off_wf = 0;
  bool finalized = false;
};

extern "C" {

int lfsr_internet_create(lfsr_internet** out, int A, int scale, int n_groups, int n_layers) {
  if (!out || A <= 0 || A > 9 || scale < 2 || scale > 4 || n_groups <= 0 || n_layers <= 0) return LFSR_E_ARG;
  lfsr_internet* c = new lfsr_internet();
  c->A = A; c->s = scale; c->ngroups = n_groups; c->nlayers = n_layers;
  const int AA = A * A;
  LfsrParamTable& P = c->P;
  P.add("AngFE.0.weight", 64, 1, AA, 0, 0, true);
  P.add("SpaFE.0.weight", 64, 1, 9, 0, 0, true);
  for (int g = 0; g < n_groups; ++g)
    for (int l = 0; l < n_layers; ++l) {
      std::string p = "CascadeInterBlock.body." + std::to_string(g) + ".chained_layers." + std::to_string(l) + ".";
      P.add(p + "Spa2Ang.weight", 64, 64, AA);
      P.add(p + "Ang2Spa.0.weight", AA * 64, 64, 1, 1, 64);
      P.add(p + "AngConvSq.weight", 64, 128, 1);
      P.add(p + "SpaConvSq.weight", 64, 128, 9);
    }
  P.add("BottleNeck.AngBottle.weight", 64, 64 * n_groups, 1);
  P.add("BottleNeck.Ang2Spa.0.weight", AA * 64, 64, 1, 1, 64);
  P.add("BottleNeck.SpaBottle.weight", 64, 64 * (n_groups + 1), 9);
  P.add("ReconBlock.PreConv.weight", 64 * scale * scale, 64, 9, 0, 0, true);
  P.add("ReconBlock.FinalConv.weight", 1, 64, 1, 0, 0, true);
  c->off_wf = P.reserve(9 * 32 * 64);
  *out = c;
  return LFSR_OK;
}

void lfsr_internet_destroy(lfsr_internet* c) { delete c; }
size_t lfsr_internet_packed_bytes(const lfsr_internet* c) { return c ? c->P.packed_floats * sizeof(float) : 0; }
int lfsr_internet_set_packed(lfsr_internet* c, void* packed, size_t bytes) { if (!c) return LFSR_E_ARG; c->finalized = false; return c->P.set_packed(packed, bytes); }
int lfsr_internet_load_param(lfsr_internet* c, const char* key, const float* data, size_t numel, void* stream) {
  if (!c) return LFSR_E_ARG;
  c->finalized = false;
  return c->P.load(key, data, numel, stream);
}
int lfsr_internet_finalize(lfsr_internet* c, void* stream) {
  if (!c || !c->P.packed || !c->P.all_loaded()) return LFSR_E_ARG;
  hipLaunchKernelGGL(k_fold_recon, dim3((9 * 32 * 64 + 255) / 256), dim3(256), 0, lfsr_stream(stream), c->P.w("ReconBlock.PreConv.weight"),
                     c->P.w("ReconBlock.FinalConv.weight"), c->P.packed + c->off_wf, c->s * c->s);
  LFSR_CHECK_LAUNCH();
  c->finalized = true;
  return LFSR_OK;
}

static void internet_layout(const lfsr_internet* c, int B, int h, int w, size_t off[8], size_t* total) {
  const size_t npix = (size_t)B * c->A * c->A * h * w, nlr = (size_t)B * h * w;
  size_t o = 0;
  auto take = [&](size_t f) { size_t r = o; o += LfsrParamTable::align64(f); return r; };
  off[0] = take(npix * 64);                         // XS0: SpaFE output (final skip)
  off[1] = take(npix * 128); off[2] = take(npix * 128);   // spatial rows, ping-pong
  off[3] = take(nlr * 128); off[4] = take(nlr * 128);     // angular rows, ping-pong
  off[5] = take(npix * 64 * (c->ngroups + 1));      // collected spatial outputs + BottleNeck's Ang2Spa slice
  off[6] = take(nlr * 64 * c->ngroups);             // collected angular outputs
  off[7] = take(npix * 64);                         // BottleNeck output
  *total = o;
}

size_t lfsr_internet_workspace_bytes(const lfsr_internet* c, int B, int h, int w) {
  if (!c || B <= 0 || h <= 0 || w <= 0) return 0;
  size_t off[8], tot;
  internet_layout(c, B, h, w, off, &tot);
  return tot * sizeof(float);
}

int lfsr_internet_forward(lfsr_internet* c, const float* x, float* out, int B, int h, int w, void* workspace, size_t workspace_bytes, void* stream) {
  if (!c || !x || !out || !workspace || B <= 0 || h <= 0 || w <= 0 || !c->finalized || ((uintptr_t)workspace & 15)) return LFSR_E_ARG;
  size_t off[8], tot;
  internet_layout(c, B, h, w, off, &tot);
  if (workspace_bytes < tot * sizeof(float)) return LFSR_E_WS;
  const int A = c->A, AA = A * A, nimg = B * AA, G = c->ngroups;
  const long long npix = (long long)nimg * h * w, nlr = (long long)B * h * w;
  if (npix >= (1LL << 31) / (64 * (G + 1))) return LFSR_E_ARG;
  float* ws = (float*)workspace;
  float *XS0 = ws + off[0], *S[2] = {ws + off[1], ws + off[2]}, *Ar[2] = {ws + off[3], ws + off[4]}, *CS = ws + off[5], *CA = ws + off[6], *BO = ws + off[7];
  const LfsrParamTable& P = c->P;
  hipStream_t st = lfsr_stream(stream);
  const int cs_stride = 64 * (G + 1), ca_stride = 64 * G;
  int rc;
#define RC(call) do { rc = (call); if (rc) return rc; } while (0)
  auto gemm = [&](auto launcher, const float* X, int xs, int xo, const float* Wp, float* Y, int ys, int yo, const float* R1, int r1s, int r1o,
                  int M, int N, int ntaps, int CH, float slope) -> int {
    GemmArgs p{};
    p.X = X; p.x_stride = xs; p.x_choff = xo; p.Wp = Wp; p.Y = Y; p.y_stride = ys; p.y_choff = yo; p.R1 = R1; p.r1_stride = r1s; p.r1_choff = r1o;
    p.M = M; p.N = N; p.Npad = npad32(N); p.A = A; p.AA = AA; p.H = h; p.W = w; p.ntaps = ntaps; p.CH = CH; p.slope = slope; p.S = c->s;
    return launcher(p, st);
  };
  // feature extraction (LF_InterNet.py:35-36)
  hipLaunchKernelGGL(k_angfe, dim3(cap_grid(nlr * 16)), dim3(256), 64 * AA * sizeof(float), st, x, P.w("AngFE.0.weight"), Ar[0], 128, 0, B, A, h, w);
  LFSR_CHECK_LAUNCH();
  RC(lfsr_initconv_fwd(x, P.w("SpaFE.0.weight"), S[0], 128, 0, B, A, h, w, stream));
  hipLaunchKernelGGL(k_copy64, dim3(cap_grid(npix * 16)), dim3(256), 0, st, S[0], 128, 0, XS0, 64, 0, npix);
  LFSR_CHECK_LAUNCH();
  int cur = 0;
  for (int g = 0; g < G; ++g) {
    for (int l = 0; l < c->nlayers; ++l) {
      std::string p = "CascadeInterBlock.body." + std::to_string(g) + ".chained_layers." + std::to_string(l) + ".";
      const int nxt = cur ^ 1;
      // buffer_ang2 = ReLU(Spa2Ang(xs)) -> angular rows [64:128]
      RC(gemm(launch_gemm<IN_ANG, OUT_SAME, 64, 2>, S[cur], 128, 0, P.w(p + "Spa2Ang.weight"), Ar[cur], 128, 64, nullptr, 0, 0, (int)nlr, 64, AA, 64, 0.0f));
      // buffer_spa2 = PixelShuffle(Ang2Spa(xa)) -> spatial rows [64:128] of every view
      RC(gemm(launch_gemm<IN_SAME, OUT_VIEWS, 64, 2>, Ar[cur], 128, 0, P.w(p + "Ang2Spa.0.weight"), S[cur], 128, 64, nullptr, 0, 0, (int)nlr, AA * 64, 1, 64, 1.0f));
      // out_a = ReLU(AngConvSq(cat(xa, ang2))) + xa ; out_s = ReLU(SpaConvSq(cat(xs, spa2))) + xs
      RC(lfsr_linear_fwd(Ar[cur], 128, 0, 128, P.w(p + "AngConvSq.weight"), nullptr, Ar[cur], 128, 0, Ar[nxt], 128, 0, nlr, 64, 0.0f, stream));
      RC(gemm(launch_gemm<IN_CONV3, OUT_SAME, 128, 2>, S[cur], 128, 0, P.w(p + "SpaConvSq.weight"), S[nxt], 128, 0, S[cur], 128, 0, (int)npix, 64, 9, 64, 0.0f));
      cur = nxt;
    }
    hipLaunchKernelGGL(k_copy64, dim3(cap_grid(nlr * 16)), dim3(256), 0, st, Ar[cur], 128, 0, CA, ca_stride, 64 * g, nlr);
    LFSR_CHECK_LAUNCH();
    hipLaunchKernelGGL(k_copy64, dim3(cap_grid(npix * 16)), dim3(256), 0, st, S[cur], 128, 0, CS, cs_stride, 64 * g, npix);
    LFSR_CHECK_LAUNCH();
  }
  // BottleNeck (LF_InterNet.py:119-124)
  RC(lfsr_linear_fwd(CA, ca_stride, 0, 64 * G, P.w("BottleNeck.AngBottle.weight"), nullptr, nullptr, 0, 0, Ar[0], 128, 0, nlr, 64, 0.0f, stream));
  RC(gemm(launch_gemm<IN_SAME, OUT_VIEWS, 64, 2>, Ar[0], 128, 0, P.w("BottleNeck.Ang2Spa.0.weight"), CS, cs_stride, 64 * G, nullptr, 0, 0, (int)nlr, AA * 64, 1, 64, 1.0f));
  if (G != 4) return LFSR_E_ARG;   // SpaBottle instantiated for 5 x 64 input channels
  RC(gemm(launch_gemm<IN_CONV3, OUT_SAME, 320, 2>, CS, cs_stride, 0, P.w("BottleNeck.SpaBottle.weight"), BO, 64, 0, XS0, 64, 0, (int)npix, 64, 9, 64, 0.0f));
  // ReconBlock (LF_InterNet.py:136-141), folded: 3x3 conv 64 -> s^2, epilogue = MacPI2SAI + PixelShuffle(s)
  RC(gemm(launch_gemm<IN_CONV3, OUT_PS_HR, 64, 1>, BO, 64, 0, P.packed + c->off_wf, out, 1, 0, nullptr, 0, 0, (int)npix, c->s * c->s, 9, 1, 1.0f));
#undef RC
  return LFSR_OK;
}

}  // extern "C"
