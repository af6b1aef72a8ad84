// "Next" rows N2 / N3 of SURVEY 8f: the two host-side steps that bracket the hot path in the reference's training loop,
// moved onto the device so they no longer stall it.
//  * per-view PSNR and SSIM of cal_metrics (utils/utils.py:91-134): the reference copies label and output to the CPU and runs
//    B x A x A skimage calls per training iteration (train.py:273).  Here one block per view accumulates in fp64:
//    PSNR = 10 log10(1 / MSE) (data_range 1), SSIM = skimage.metrics.structural_similarity(gaussian_weights=True, sigma 1.5,
//    use_sample_covariance=True, data_range 1): 11-tap separable Gaussian with scipy 'reflect' borders, K1 .01 K2 .03,
//    mean over the map cropped by 5 pixels.
//  * masked angular pre-training (utils/masked_pretraining.py:85-139, hooked at train.py:250-251): fill whole views of the
//    LR SAI mosaic with a constant.
#include "lfsr_common.h"

namespace {

__device__ __forceinline__ int reflect_idx(int i, int n) {   // scipy.ndimage mode='reflect': d c b a | a b c d | d c b a
  int p = 2 * n;
  int r = i % p;
  if (r < 0) r += p;
  return r < n ? r : p - 1 - r;
}

// one block per (b, u, v) view; views are H x W windows of (B,1,A*H,A*W) SAI mosaics
__global__ __launch_bounds__(256) void k_view_metrics(const float* __restrict__ lab, const float* __restrict__ out, double* __restrict__ psnr, double* __restrict__ ssim,
                                                     int A, int H, int W, int want_ssim, int TR) {
  extern __shared__ double sm[];
  __shared__ double red[256];
  const int view = blockIdx.x % (A * A), b = blockIdx.x / (A * A), u = view / A, v = view % A;
  const long long Wm = (long long)A * W, base = ((long long)b * A * H + (long long)u * H) * Wm + (long long)v * W;
  const int tid = threadIdx.x;
  // ---- PSNR ----
  double se = 0.0;
  for (int i = tid; i < H * W; i += 256) {
    int y = i / W, x = i - y * W;
    double d = (double)lab[base + y * Wm + x] - (double)out[base + y * Wm + x];
    se += d * d;
  }
  red[tid] = se;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) { if (tid < s) red[tid] += red[tid + s]; __syncthreads(); }
  if (tid == 0) { double mse = red[0] / ((double)H * W); psnr[blockIdx.x] = mse > 0.0 ? 10.0 * log10(1.0 / mse) : INFINITY; }
  if (!want_ssim) return;
  __syncthreads();
  // ---- SSIM: separable 11-tap Gaussian of x, y, xx, yy, xy (horizontal pass into LDS, vertical pass on the fly) ----
  double g[11];
  {
    double sum = 0.0;
    for (int k = 0; k < 11; ++k) { double d = k - 5; g[k] = exp(-0.5 * d * d / (1.5 * 1.5)); sum += g[k]; }
    for (int k = 0; k < 11; ++k) g[k] /= sum;
  }
  // row tiles of TR output rows: LDS holds the horizontally filtered planes of rows [y0-5, y0+TR+5) as [5][TR+10][W]
  double* hx = sm;
  const double C1 = 0.01 * 0.01, C2 = 0.03 * 0.03, cov_norm = 121.0 / 120.0;
  const int ch = H - 10, cw = W - 10;
  double acc = 0.0;
  const int RW = (TR + 10) * W;
  for (int y0 = 5; y0 < H - 5; y0 += TR) {
    const int rows_out = min(TR, H - 5 - y0);
    __syncthreads();
    for (int i = tid; i < (rows_out + 10) * W; i += 256) {
      int ry = i / W, x = i - ry * W;
      int y = reflect_idx(y0 - 5 + ry, H);
      double ax = 0, ay = 0, axx = 0, ayy = 0, axy = 0;
      for (int k = 0; k < 11; ++k) {
        int xs = reflect_idx(x + k - 5, W);
        double p = (double)lab[base + y * Wm + xs], q = (double)out[base + y * Wm + xs];
        ax += g[k] * p; ay += g[k] * q; axx += g[k] * p * p; ayy += g[k] * q * q; axy += g[k] * p * q;
      }
      hx[i] = ax; hx[RW + i] = ay; hx[2 * RW + i] = axx; hx[3 * RW + i] = ayy; hx[4 * RW + i] = axy;
    }
    __syncthreads();
    for (int i = tid; i < rows_out * cw; i += 256) {
      int ry = i / cw, x = 5 + i % cw;
      double ux = 0, uy = 0, uxx = 0, uyy = 0, uxy = 0;
      for (int k = 0; k < 11; ++k) {
        int o = (ry + k) * W + x;
        ux += g[k] * hx[o]; uy += g[k] * hx[RW + o]; uxx += g[k] * hx[2 * RW + o]; uyy += g[k] * hx[3 * RW + o]; uxy += g[k] * hx[4 * RW + o];
      }
      double vx = cov_norm * (uxx - ux * ux), vy = cov_norm * (uyy - uy * uy), vxy = cov_norm * (uxy - ux * uy);
      acc += ((2 * ux * uy + C1) * (2 * vxy + C2)) / ((ux * ux + uy * uy + C1) * (vx + vy + C2));
    }
  }
  red[tid] = acc;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) { if (tid < s) red[tid] += red[tid + s]; __syncthreads(); }
  if (tid == 0) ssim[blockIdx.x] = (ch > 0 && cw > 0) ? red[0] / ((double)ch * cw) : 0.0;
}

__global__ __launch_bounds__(256) void k_mask_views(const float* __restrict__ x, float* __restrict__ y, const unsigned char* __restrict__ mask, float fill,
                                                   long long total, int A, int h, int w) {
  const int Wm = A * w, Hm = A * h;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    int col = (int)(i % Wm);
    int row = (int)((i / Wm) % Hm);
    y[i] = mask[(row / h) * A + col / w] ? fill : x[i];
  }
}

// per-view fill values (mask_value = 'mean' with several masked views: each view gets its own mean, masked_pretraining.py:121-123)
__global__ __launch_bounds__(256) void k_mask_views_fill(const float* __restrict__ x, float* __restrict__ y, const unsigned char* __restrict__ mask,
                                                        const float* __restrict__ fill, long long total, int A, int h, int w) {
  const int Wm = A * w, Hm = A * h;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    int col = (int)(i % Wm);
    int row = (int)((i / Wm) % Hm);
    const int v = (row / h) * A + col / w;
    y[i] = mask[v] ? fill[v] : x[i];
  }
}


// "Next" row N4 of SURVEY 8f, the output tail of test() (train.py:329-341, inference.py:205-216; utils/utils.py:191-204):
//   rgb = uint8( clip( ycbcr2rgb(cat(Sr_SAI_y, Sr_SAI_cbcr)), 0, 1 ) * 255 ),  split into A x A views (h, w, 3)
// The reference does this on the CPU in float64 (numpy); here one thread per mosaic pixel evaluates the same affine map in
// fp64 with the host-computed inverse matrix (same operation order) and writes the view-major (A, A, h, w, 3) uint8 tensor
// that the BMP writer consumes, so the 13 MB mosaic never crosses PCIe as floats.
__global__ __launch_bounds__(256) void k_ycbcr2rgb_views(const float* __restrict__ yp, const float* __restrict__ cbcr, unsigned char* __restrict__ out,
                                                        int A, int h, int w, double m00, double m01, double m02, double m10, double m11, double m12,
                                                        double m20, double m21, double m22, double o0, double o1, double o2) {
  const long long Hm = (long long)A * h, Wm = (long long)A * w, npix = Hm * Wm;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < npix; i += (long long)gridDim.x * 256) {
    const long long Y = i / Wm, X = i - Y * Wm;
    const double c0 = (double)yp[i], c1 = (double)cbcr[i], c2 = (double)cbcr[npix + i];
    // explicit round-to-nearest multiplies and adds in numpy's order: no fma contraction, so the uint8 truncation below sees
    // bit-identical doubles
    double r = __dsub_rn(__dadd_rn(__dadd_rn(__dmul_rn(m00, c0), __dmul_rn(m01, c1)), __dmul_rn(m02, c2)), o0);
    double g = __dsub_rn(__dadd_rn(__dadd_rn(__dmul_rn(m10, c0), __dmul_rn(m11, c1)), __dmul_rn(m12, c2)), o1);
    double b = __dsub_rn(__dadd_rn(__dadd_rn(__dmul_rn(m20, c0), __dmul_rn(m21, c1)), __dmul_rn(m22, c2)), o2);
    r = __dmul_rn(fmin(fmax(r, 0.0), 1.0), 255.0); g = __dmul_rn(fmin(fmax(g, 0.0), 1.0), 255.0); b = __dmul_rn(fmin(fmax(b, 0.0), 1.0), 255.0);
    const int u = (int)(Y / h), y = (int)(Y - (long long)u * h), v = (int)(X / w), x = (int)(X - (long long)v * w);
    unsigned char* o = out + ((((long long)u * A + v) * h + y) * w + x) * 3;
    o[0] = (unsigned char)r; o[1] = (unsigned char)g; o[2] = (unsigned char)b;   // astype('uint8'): truncation
  }
}

}  // namespace

extern "C" {

int lfsr_view_metrics(const float* label, const float* out, double* psnr, double* ssim, int B, int A, int H, int W, void* stream) {
  if (!label || !out || !psnr || B <= 0 || A <= 0 || H <= 0 || W <= 0) return LFSR_E_ARG;
  int TR = 16;
  while (ssim && TR > 1 && (size_t)5 * (TR + 10) * W * sizeof(double) > 150 * 1024) TR >>= 1;
  size_t smem = ssim ? (size_t)5 * (TR + 10) * W * sizeof(double) : 0;
  if (smem > 150 * 1024 || (ssim && (H < 11 || W < 11))) return LFSR_E_ARG;   // skimage needs win_size 11 <= view size
  if (ssim && smem > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_view_metrics), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    if (e != hipSuccess) return LFSR_HIP_ERR(e);
  }
  hipLaunchKernelGGL(k_view_metrics, dim3((unsigned)(B * A * A)), dim3(256), smem, lfsr_stream(stream), label, out, psnr, ssim, A, H, W, ssim ? 1 : 0, TR);
  LFSR_CHECK_LAUNCH();
  return LFSR_OK;
}

int lfsr_mask_views(const float* x, float* y, const unsigned char* mask, float fill, int B, int C, int A, int h, int w, void* stream) {
  if (!x || !y || !mask || B <= 0 || C <= 0 || A <= 0 || h <= 0 || w <= 0) return LFSR_E_ARG;
  long long total = (long long)B * C * A * h * A * w;
  unsigned grid = lfsr_blocks(total, 256);
  if (grid > 8192) grid = 8192;
  hipLaunchKernelGGL(k_mask_views, dim3(grid), dim3(256), 0, lfsr_stream(stream), x, y, mask, fill, total, A, h, w);
  LFSR_CHECK_LAUNCH();
  return LFSR_OK;
}

int lfsr_mask_views_fill(const float* x, float* y, const unsigned char* mask, const float* fill, int B, int C, int A, int h, int w, void* stream) {
  if (!x || !y || !mask || !fill || B <= 0 || C <= 0 || A <= 0 || h <= 0 || w <= 0) return LFSR_E_ARG;
  long long total = (long long)B * C * A * h * A * w;
  unsigned grid = lfsr_blocks(total, 256);
  if (grid > 8192) grid = 8192;
  hipLaunchKernelGGL(k_mask_views_fill, dim3(grid), dim3(256), 0, lfsr_stream(stream), x, y, mask, fill, total, A, h, w);
  LFSR_CHECK_LAUNCH();
  return LFSR_OK;
}

}  // extern "C"

extern "C" int lfsr_ycbcr2rgb_views(const float* y, const float* cbcr, unsigned char* out, int A, int h, int w, const double* minv255,
                                    const double* offset, void* stream) {
  if (!y || !cbcr || !out || !minv255 || !offset || A <= 0 || h <= 0 || w <= 0) return LFSR_E_ARG;
  const long long npix = (long long)A * h * A * w;
  hipLaunchKernelGGL(k_ycbcr2rgb_views, dim3(lfsr_blocks(npix, 256)), dim3(256), 0, lfsr_stream(stream), y, cbcr, out, A, h, w,
                     minv255[0], minv255[1], minv255[2], minv255[3], minv255[4], minv255[5], minv255[6], minv255[7], minv255[8],
                     offset[0], offset[1], offset[2]);
  LFSR_CHECK_LAUNCH();
  return LFSR_OK;
}
