// Backward (data-gradient) wrappers over the gather-GEMM kernel, plus the small head / init_conv backward kernels.
// Every data gradient of DistgSSR is again a gather-GEMM (the transposed index map of a gather is a gather), with the
// weights packed transposed (lfsr_pack_weight_T / _chunkT) and two epilogue extras: the LeakyReLU' mask taken from the
// saved forward activation, and in-place accumulation into the gradient buffer (R1 aliasing Y).
#include <stdlib.h>

#include "gemm_gather_kernel.h"
#include "lfsr_internal.h"

namespace {

// dOut (B,1,A*h*s,A*w*s) HR mosaic -> g16[p][ij] rows (p = VCL pixel), and df[p][k] = sum_ij g16[p][ij] wf[ij][k]
template <int S>
__global__ __launch_bounds__(256) void k_head_bwd(const float* __restrict__ dout, const float* __restrict__ wf, float* __restrict__ df,
                                                 float* __restrict__ g16, int B, int A, int h, int w) {
  __shared__ float sw[S * S * 64];
  for (int i = threadIdx.x; i < S * S * 64; i += 256) sw[i] = wf[i];
  __syncthreads();
  const long long npix = (long long)B * A * A * h * w;
  const int Wo = A * w * S, Ho = A * h * S;
  for (long long g = (long long)blockIdx.x * 256 + threadIdx.x; g < npix * 16; g += (long long)gridDim.x * 256) {
    long long pix = g >> 4;
    int c4 = (int)(g & 15) * 4;
    int x = (int)(pix % w);
    long long t = pix / w;
    int y = (int)(t % h);
    t /= h;
    int view = (int)(t % (A * A));
    int b = (int)(t / (A * A));
    int u = view / A, v = view - u * A;
    const float* ob = dout + ((long long)b * Ho + (long long)(u * h + y) * S) * Wo + (long long)(v * w + x) * S;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
#pragma unroll
    for (int i = 0; i < S; ++i)
#pragma unroll
      for (int j = 0; j < S; ++j) {
        float gv = ob[(long long)i * Wo + j];
        const float* wr = sw + (i * S + j) * 64 + c4;
        a0 = fmaf(gv, wr[0], a0); a1 = fmaf(gv, wr[1], a1); a2 = fmaf(gv, wr[2], a2); a3 = fmaf(gv, wr[3], a3);
        if (c4 == 0) g16[pix * 16 + i * S + j] = gv;
      }
    if (c4 == 0 && S * S < 16)
      for (int q = S * S; q < 16; ++q) g16[pix * 16 + q] = 0.f;
    *reinterpret_cast<float4*>(df + pix * 64 + c4) = make_float4(a0, a1, a2, a3);
  }
}

// per-block column sums of a row-major (M, N<=16) matrix: partial[blk][N]
__global__ __launch_bounds__(256) void k_colsum(const float* __restrict__ g, int M, int N, float* __restrict__ partial, int rows_per_blk) {
  __shared__ float red[256];
  const int col = threadIdx.x & 15, r0 = threadIdx.x >> 4;
  const int m0 = blockIdx.x * rows_per_blk, m1 = min(M, m0 + rows_per_blk);
  float s = 0.f;
  if (col < N)
    for (int m = m0 + r0; m < m1; m += 16) s += g[(long long)m * N + col];
  red[threadIdx.x] = s;
  __syncthreads();
  if (threadIdx.x < 16) {
    float t = 0.f;
    for (int r = 0; r < 16; ++r) t += red[r * 16 + threadIdx.x];
    if (threadIdx.x < N) partial[blockIdx.x * 16 + threadIdx.x] = t;
  }
}

// gradients of the folded head back to upsample.0.{weight,bias} and upsample.2.weight (DistgSSR.py:24-27)
__global__ void k_head_fold_bwd(const float* __restrict__ dWf, const float* __restrict__ partial, int nblk, const float* __restrict__ w0,
                                const float* __restrict__ b0, const float* __restrict__ w2, float* __restrict__ dw0, float* __restrict__ db0,
                                float* __restrict__ dw2, int s2) {
  __shared__ float dbf[16];
  if (threadIdx.x < 16) {
    float t = 0.f;
    if ((int)threadIdx.x < s2)
      for (int b = 0; b < nblk; ++b) t += partial[b * 16 + threadIdx.x];
    dbf[threadIdx.x] = t;
  }
  __syncthreads();
  const int C = 64;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < C * s2 * C; i += gridDim.x * blockDim.x) {
    int k = i % C, row = i / C, c = row / s2, ij = row - c * s2;
    dw0[i] = w2[c] * dWf[ij * C + k];
  }
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < C * s2; i += gridDim.x * blockDim.x) {
    int c = i / s2, ij = i - c * s2;
    db0[i] = w2[c] * dbf[ij];
  }
  if (blockIdx.x == 0 && threadIdx.x < C) {
    int c = threadIdx.x;
    double a = 0.0;
    for (int ij = 0; ij < s2; ++ij) {
      for (int k = 0; k < C; ++k) a += (double)w0[((long long)c * s2 + ij) * C + k] * (double)dWf[ij * C + k];
      a += (double)b0[c * s2 + ij] * (double)dbf[ij];
    }
    dw2[c] = (float)a;
  }
}

// xg[p][0..8] = the 9 zero-padded 3x3 neighbours of LR pixel p inside its view (SAI mosaic input), xg[p][9..15] = 0
__global__ __launch_bounds__(256) void k_init_gather9(const float* __restrict__ x, float* __restrict__ xg, int B, int A, int h, int w) {
  const long long npix = (long long)B * A * A * h * w;
  const int Wm = A * w, Hm = A * h;
  for (long long g = (long long)blockIdx.x * 256 + threadIdx.x; g < npix * 16; g += (long long)gridDim.x * 256) {
    long long pix = g >> 4;
    int k = (int)(g & 15);
    float v = 0.f;
    if (k < 9) {
      int xx = (int)(pix % w);
      long long t = pix / w;
      int yy = (int)(t % h);
      t /= h;
      int view = (int)(t % (A * A));
      int b = (int)(t / (A * A));
      int u = view / A, vv = view - u * A;
      int sy = yy + k / 3 - 1, sx = xx + k % 3 - 1;
      if (sy >= 0 && sy < h && sx >= 0 && sx < w) v = x[(long long)b * Hm * Wm + (long long)(u * h + sy) * Wm + vv * w + sx];
    }
    xg[g] = v;
  }
}

// chunked transposed pack for the dgrad of a 1x1 conv followed by a (1-D) pixel shuffle:
// w (O = r2*ch, C) -> out[q (r2)][k (Cpad32)][c (ch)] = w[nref(q,c)][k],  nref = perm ? c*r2+q : q*ch+c
__global__ __launch_bounds__(256) void k_pack_chunkT(const float* __restrict__ w, float* __restrict__ out, int O, int C, int ch, int perm, int Cpad) {
  const int r2 = O / ch;
  const long long total = (long long)r2 * Cpad * ch;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    int c = (int)(i % ch);
    long long t = i / ch;
    int k = (int)(t % Cpad);
    int q = (int)(t / Cpad);
    int nref = perm ? c * r2 + q : q * ch + c;
    out[i] = k < C ? w[(long long)nref * C + k] : 0.f;
  }
}

__global__ __launch_bounds__(256) void k_add_inplace(float4* __restrict__ a, const float4* __restrict__ b, long long n4) {
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
    float4 x = a[i], y = b[i];
    a[i] = make_float4(x.x + y.x, x.y + y.y, x.z + y.z, x.w + y.w);
  }
}

}  // namespace

// ---- EPIConv.0 data gradient, EPI-line form -------------------------------------------------------------------------------------------
// dX[line, view v', x'][c] += sum_{dx, n} dE[line, x' + pad - dx][n] * W[tap = A dx + v'][n][c]   (the adjoint of the 1 x A^2, stride-A conv of
// DistgSSR.py:91-97 along an EPI line).  The gather form (k_gemm_gather<IN_LINE_*, OUT_EPI*>) re-gathers dE rows per tap and scatters 64-channel
// rows per view.  Here a persistent 512-thread block takes chunks of 5 EPI lines (<= 32 pixels each): the lines' dE (36 x 32 with two pixels of
// zero halo) are staged in LDS once per chunk, then one phase per destination view v': its 5 x 32 x 64 weights are staged (double-buffered) and
// wave (pb, cb) computes D[c 16][x' 16] of every line with v_mfma_f32_16x16x4_f32 (A = W^t, B = dE, K = n): a lane holds four consecutive CHANNELS
// of one pixel, so the accumulate-into-dX epilogue is one 16-B load + one 16-B store per line.  Both operands sit in LDS with the K index permuted
// (n = 4 s + g stored at 8 g + s), so a lane reads the eight K steps of a (tap, line) pair with two ds_read_b128.  (SQ_LDS_BANK_CONFLICT shows 65 %
// conflict cycles for these padded-row images; a conflict-free [s / 4][row][g][s % 4] image was built and measured 0.5 % SLOWER on the training step in the
// same call -- its scattered staging stores cost more than the reads save; LDS is not what bounds this kernel.)
constexpr int ED_WR = 36, ED_L = 5;

struct EpiDgradArgs {
  const float* G;            // dE rows ((q h + y) w + x), 32 channels
  const float* Wd;           // direct pack [A*A taps][32][64]
  float* Y; int y_stride; int y_choff;     // dX (read-modify-write)
  int g_bytes, y_bytes;
  int B, A, H, W, vert;
};

template <int A>
__global__ __launch_bounds__(512) void k_epi0_dgrad_lines(EpiDgradArgs p) {
  extern __shared__ __attribute__((aligned(16))) float smd[];
  constexpr int WBUF = A * 64 * ED_WR;            // floats per weight buffer
  float* const sE = smd + 2 * WBUF;               // [ED_L][36][ED_WR]
  typedef float f32x4d __attribute__((ext_vector_type(4)));
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l15 = lane & 15, g = lane >> 4;
  const int pb = wave >> 2, cb = wave & 3;
  const int HW = p.H * p.W;
  const int len = p.vert ? p.H : p.W, across = p.vert ? p.W : p.H;
  const int nlines = p.B * A * across;
  const int vstride = p.vert ? A * HW : HW, pstride = p.vert ? p.W : 1;
  constexpr int DOOB = (int)0x80000000u;
  const __amdgpu_buffer_rsrc_t rsG = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.G), 0, p.g_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsY = __builtin_amdgcn_make_buffer_rsrc(p.Y, 0, p.y_bytes, 0x00020000);
  const int nchunks = (nlines + ED_L - 1) / ED_L;
  float4 wr[A];
  auto load_w = [&](int vv) {
#pragma unroll
    for (int i = 0; i < A; ++i) {   // idx = tid + 512 i over (dx, n, c4): dx = i, n = tid >> 4, c4 = tid & 15
      wr[i] = *reinterpret_cast<const float4*>(p.Wd + ((long long)(i * A + vv) * 32 + (tid >> 4)) * 64 + (tid & 15) * 4);
    }
  };
  auto store_w = [&](float* buf) {
    const int n = tid >> 4, c0 = (tid & 15) * 4, kperm = (n & 3) * 8 + (n >> 2);
#pragma unroll
    for (int i = 0; i < A; ++i) {
      float* d = buf + (i * 64 + c0) * ED_WR + kperm;
      d[0] = wr[i].x; d[ED_WR] = wr[i].y; d[2 * ED_WR] = wr[i].z; d[3 * ED_WR] = wr[i].w;
    }
  };
  for (int chunk = blockIdx.x; chunk < nchunks; chunk += gridDim.x) {
    const int line0 = chunk * ED_L;
    __syncthreads();                               // the previous chunk's phases are done with sE and the weight buffers
    // ---- the chunk's dE lines -> sE (halo pixels and missing lines zero), K index permuted
    for (int idx = tid; idx < ED_L * 36 * 8; idx += 512) {
      const int n4 = idx & 7, px = (idx >> 3) % 36, ln = idx / (36 * 8);
      const int line = line0 + ln, t = px - 2;
      int off = DOOB;
      if (line < nlines && t >= 0 && t < len) {
        const int q = line / across, o = line - q * across;
        const int mbase = p.vert ? q * HW + o : (q * p.H + o) * p.W;
        off = ((mbase + t * pstride) * 32 + n4 * 4) * 4;
      }
      const f32x4d v = __builtin_bit_cast(f32x4d, __builtin_amdgcn_raw_buffer_load_b128(rsG, off, 0, 0));
      float* d = sE + (ln * 36 + px) * ED_WR + n4;   // n = 4 n4 + k -> (n & 3) * 8 + (n >> 2) = 8 k + n4
      d[0] = v.x; d[8] = v.y; d[16] = v.z; d[24] = v.w;
    }
    load_w(0);
    store_w(smd);
    // this lane's pixel of each line of the chunk (x' = pb 16 + l15) in dX: byte offset of its 4 channels, or out of range
    int yoff[ED_L];
#pragma unroll
    for (int ln = 0; ln < ED_L; ++ln) {
      const int line = line0 + ln, xq = pb * 16 + l15;
      yoff[ln] = DOOB;
      if (line < nlines && xq < len) {
        const int q = line / across, o = line - q * across;
        int base;
        if (!p.vert) base = q * A * HW + o * p.W;
        else { const int b = q / A, v = q - b * A; base = (b * A * A + v) * HW + o; }
        yoff[ln] = ((base + xq * pstride) * p.y_stride + p.y_choff + cb * 16 + 4 * g) * 4;
      }
    }
#pragma unroll 1
    for (int vv = 0; vv < A; ++vv) {
      __syncthreads();                             // weights of view vv (and, for vv = 0, the dE lines) are staged; the other buffer is free
      if (vv + 1 < A) load_w(vv + 1);
      const float* sW = smd + (vv & 1) * WBUF;
      const int vbytes = vv * vstride * p.y_stride * 4;   // wave-uniform
      f32x4d old[ED_L];
#pragma unroll
      for (int ln = 0; ln < ED_L; ++ln) old[ln] = __builtin_bit_cast(f32x4d, __builtin_amdgcn_raw_buffer_load_b128(rsY, yoff[ln], vbytes, 0));
      f32x4d acc[ED_L];
#pragma unroll
      for (int ln = 0; ln < ED_L; ++ln) acc[ln] = f32x4d{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int dx = 0; dx < A; ++dx) {
        const float* ap = sW + (dx * 64 + cb * 16 + l15) * ED_WR + 8 * g;
        const f32x4d a0 = *reinterpret_cast<const f32x4d*>(ap), a1 = *reinterpret_cast<const f32x4d*>(ap + 4);
#pragma unroll
        for (int ln = 0; ln < ED_L; ++ln) {
          const float* bp = sE + (ln * 36 + pb * 16 + l15 + 4 - dx) * ED_WR + 8 * g;
          const f32x4d b0 = *reinterpret_cast<const f32x4d*>(bp), b1 = *reinterpret_cast<const f32x4d*>(bp + 4);
          acc[ln] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.x, b0.x, acc[ln], 0, 0, 0);
          acc[ln] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.y, b0.y, acc[ln], 0, 0, 0);
          acc[ln] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.z, b0.z, acc[ln], 0, 0, 0);
          acc[ln] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.w, b0.w, acc[ln], 0, 0, 0);
          acc[ln] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.x, b1.x, acc[ln], 0, 0, 0);
          acc[ln] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.y, b1.y, acc[ln], 0, 0, 0);
          acc[ln] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.z, b1.z, acc[ln], 0, 0, 0);
          acc[ln] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.w, b1.w, acc[ln], 0, 0, 0);
        }
      }
#pragma unroll
      for (int ln = 0; ln < ED_L; ++ln)
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(__attribute__((ext_vector_type(4))) unsigned int, old[ln] + acc[ln]), rsY, yoff[ln], vbytes, 0);
      if (vv + 1 < A) store_w(smd + ((vv + 1) & 1) * WBUF);   // (the other buffer: last read in phase vv - 1, which every wave left before this phase's barrier)
    }
  }
}

// LFSR_E_ARG = not covered (the caller keeps the gather-GEMM).  LFSR_DGRAD_EPI=gather forces the gather form (A/B runs).
int lfsr_epi0_dgrad_launch(const float* dE, const float* w_direct, float* dx, int dx_stride, int dx_choff, int B, int A, int h, int w, int vert, hipStream_t st) {
  LfsrOpTimer op_t("epi0_dgrad", B, h * w, st);
  if (!dE || !w_direct || !dx || B <= 0 || h <= 0 || w <= 0 || ((dx_stride | dx_choff) & 3)) return LFSR_E_ARG;
  const char* sel = lfsr_sel("LFSR_DGRAD_EPI");
  if (A != 5 || (vert ? h : w) > 32 || (sel && sel[0] == 'g')) return LFSR_E_ARG;
  if ((long long)B * A * A * h * w * dx_stride * 4 >= (1LL << 31)) return LFSR_E_ARG;
  static std::atomic<bool> attr_set[64];
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return LFSR_E_ARG;
  constexpr int smem = (2 * 5 * 64 * ED_WR + ED_L * 36 * ED_WR) * 4;
  if (!attr_set[dev]) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_epi0_dgrad_lines<5>), hipFuncAttributeMaxDynamicSharedMemorySize, smem);
    if (e != hipSuccess) return LFSR_HIP_ERR(e);
    attr_set[dev] = true;
  }
  EpiDgradArgs p{};
  p.G = dE; p.Wd = w_direct; p.Y = dx; p.y_stride = dx_stride; p.y_choff = dx_choff;
  p.g_bytes = (int)((long long)B * A * h * w * 32 * 4); p.y_bytes = (int)((long long)B * A * A * h * w * dx_stride * 4);
  p.B = B; p.A = A; p.H = h; p.W = w; p.vert = vert;
  const long long nchunks = ((long long)B * A * (vert ? w : h) + ED_L - 1) / ED_L;
  hipLaunchKernelGGL(k_epi0_dgrad_lines<5>, dim3((unsigned)(nchunks < 256 ? nchunks : 256)), dim3(512), smem, st, p);
  LFSR_CHECK_LAUNCH();
  return LFSR_OK;
}

// ---- AngConv.0 data gradient, streaming form -------------------------------------------------------------------------------------------------
// dX[b, view, y, x][c] += sum_n dA[b, y, x][n] * W[view][n][c]   (the adjoint of the A x A, stride-A conv over the MacPI: DistgSSR.py:84-90).  K is only 16, so this
// is a read-modify-write stream over dX (25 views x 64 channels per LR pixel) with 16 FMAs per element: plain VALU, 16 B per lane, no MFMA, no gather-GEMM.
// Block = one (b, view) image chunk, 16 pixels x 16 channel quads per pass.  Round 4: the thread's 16 x 4 weights live in registers for the block's life (they were
// sixteen LDS reads per pixel) and FOUR pixels are in flight per thread -- their read-modify-write loads and gradient rows all issued before the first FMA -- where the
// round-3 kernel kept one load -> 64 FMA -> store chain per thread and paid a memory round trip per pixel (114 us for a 105-MB stream whose floor is ~25 us).
// Same FMA order per element (n = 0..15 onto the old value): bit-equal to the round-3 kernel.
__global__ __launch_bounds__(256) void k_ang0_dgrad(const float* __restrict__ dA, const float* __restrict__ Wd, float* __restrict__ dX, int dx_stride, int dx_choff,
                                                    int AA, int HW, int chunks) {
  const int tid = threadIdx.x, c4 = tid & 15, pr = tid >> 4;
  const int img = blockIdx.x / chunks, chunk = blockIdx.x - img * chunks;   // img = b * AA + view
  const int b = img / AA, view = img - b * AA;
  float4 wv[16];   // direct pack [tap = view][Npad = 32][64]: rows n < 16, this thread's channel quad
#pragma unroll
  for (int n = 0; n < 16; ++n) wv[n] = *reinterpret_cast<const float4*>(Wd + ((long long)view * 32 + n) * 64 + c4 * 4);
  const int per = (HW + chunks - 1) / chunks;
  const int p0 = chunk * per, p1 = p0 + per < HW ? p0 + per : HW;
  constexpr int U = 4;
  auto fma16 = [&](float4& acc, const float4 (&g)[4]) {
    const float gv[16] = {g[0].x, g[0].y, g[0].z, g[0].w, g[1].x, g[1].y, g[1].z, g[1].w, g[2].x, g[2].y, g[2].z, g[2].w, g[3].x, g[3].y, g[3].z, g[3].w};
#pragma unroll
    for (int n = 0; n < 16; ++n) {
      acc.x = fmaf(gv[n], wv[n].x, acc.x); acc.y = fmaf(gv[n], wv[n].y, acc.y);
      acc.z = fmaf(gv[n], wv[n].z, acc.z); acc.w = fmaf(gv[n], wv[n].w, acc.w);
    }
  };
  int q0 = p0;
  for (; q0 + 16 * U <= p1; q0 += 16 * U) {       // whole groups of 64 pixels (a block-uniform trip count): no control flow between the loads and the stores
    float4 acc[U], g[U][4];
    float4* dst[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int px = q0 + pr + 16 * u;
      dst[u] = reinterpret_cast<float4*>(dX + ((long long)img * HW + px) * dx_stride + dx_choff + c4 * 4);
      acc[u] = *dst[u];
      const float4* ga = reinterpret_cast<const float4*>(dA + ((long long)b * HW + px) * 16);
      g[u][0] = ga[0]; g[u][1] = ga[1]; g[u][2] = ga[2]; g[u][3] = ga[3];
    }
    __builtin_amdgcn_sched_barrier(0);            // (every load of the group is in flight before the first FMA: the scheduler otherwise sinks them between the pixels)
#pragma unroll
    for (int u = 0; u < U; ++u) fma16(acc[u], g[u]);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int u = 0; u < U; ++u) *dst[u] = acc[u];
  }
  for (int px = q0 + pr; px < p1; px += 16) {      // ragged rest (< 64 pixels)
    float4* dst = reinterpret_cast<float4*>(dX + ((long long)img * HW + px) * dx_stride + dx_choff + c4 * 4);
    float4 acc = *dst;
    const float4* ga = reinterpret_cast<const float4*>(dA + ((long long)b * HW + px) * 16);
    const float4 g[4] = {ga[0], ga[1], ga[2], ga[3]};
    fma16(acc, g);
    *dst = acc;
  }
}

// LFSR_DGRAD_ANG=gather keeps the gather-GEMM (A/B runs); LFSR_E_ARG = not covered
int lfsr_ang0_dgrad_launch(const float* dA16, const float* w_direct, float* dx, int dx_stride, int dx_choff, int B, int A, int h, int w, hipStream_t st) {
  LfsrOpTimer op_t("ang0_dgrad", B, h * w, st);
  if (!dA16 || !w_direct || !dx || B <= 0 || A <= 0 || h <= 0 || w <= 0 || ((dx_stride | dx_choff) & 3)) return LFSR_E_ARG;
  const char* sel = lfsr_sel("LFSR_DGRAD_ANG");
  if (sel && sel[0] == 'g') return LFSR_E_ARG;
  const int AA = A * A, HW = h * w;
  int chunks = (HW + 255) / 256;            // >= 256 pixels per block
  if (chunks < 1) chunks = 1;
  const long long nblk = (long long)B * AA * chunks;
  if (nblk > 0x7fffffffLL) return LFSR_E_ARG;
  hipLaunchKernelGGL(k_ang0_dgrad, dim3((unsigned)nblk), dim3(256), 0, st, dA16, w_direct, dx, dx_stride, dx_choff, AA, HW, chunks);
  LFSR_CHECK_LAUNCH();
  return LFSR_OK;
}

int lfsr_add_inplace(float* a, const float* b, long long n, hipStream_t st) {
  if (n & 3) return LFSR_E_ARG;
  unsigned grid = lfsr_blocks(n / 4, 256);
  if (grid > 8192) grid = 8192;
  hipLaunchKernelGGL(k_add_inplace, dim3(grid), dim3(256), 0, st, reinterpret_cast<float4*>(a), reinterpret_cast<const float4*>(b), n / 4);
  LFSR_CHECK_LAUNCH();
  return LFSR_OK;
}

int lfsr_bwd_gemm(const LfsrGemm& g, hipStream_t st) {
  LfsrOpTimer op_t("bwd_gemm", g.in_mode * 16 + g.out_mode, g.N, st);
  GemmArgs p{};
  p.X = g.X; p.x_stride = g.x_stride; p.x_choff = g.x_choff; p.Wp = g.Wp; p.bias = nullptr;
  p.Y = g.Y; p.y_stride = g.y_stride; p.y_choff = g.y_choff;
  p.R1 = g.R1; p.r1_stride = g.r1_stride; p.r1_choff = g.r1_choff;
  p.Mk = g.Mk; p.mk_stride = g.mk_stride; p.mk_choff = g.mk_choff; p.mk_slope = g.mk_slope;
  p.M = g.M; p.N = g.N; p.Npad = npad32(g.N); p.A = g.A; p.AA = g.A * g.A; p.H = g.h; p.W = g.w; p.ntaps = g.ntaps; p.CH = g.CH; p.slope = 1.0f;
  if ((g.x_stride | g.x_choff) & 3) return LFSR_E_ARG;
  // fuse.0 dgrad (64 -> 144, masked by the saved concat buffer): the row-streaming kernel (LFSR_NO_ROWGEMM keeps the gather-GEMM: A/B runs)
  if (g.in_mode == LFSR_IN_SAME && g.out_mode == LFSR_OUT_SAME && g.cin == 64 && g.N == 144 && g.ntaps == 1 && !g.R1 && g.M >= 2048 && !lfsr_sel("LFSR_NO_ROWGEMM") && !(lfsr_sel("LFSR_DGRAD_PW") && lfsr_sel("LFSR_DGRAD_PW")[0] == 'g')) {     // (LFSR_DGRAD_PW=gather: the gather-GEMM for this data gradient only -- the forward keeps its kernel)
    const int rc = lfsr_rowgemm_dgrad144_launch(g.X, g.x_stride, g.x_choff, g.Wp, g.Mk, g.mk_stride, g.mk_choff, g.mk_slope, g.Y, g.y_stride, g.y_choff, g.M, st);
    if (rc != LFSR_E_ARG) return rc;
  }
#define BG(IM, OM, CI, NT) if (g.in_mode == IM && g.out_mode == OM && g.cin == CI && p.Npad % (32 * NT) == 0) return launch_gemm<IM, OM, CI, NT>(p, st);
  BG(IN_SAME, OUT_SAME, 64, 1)      // fuse.0 dgrad (64 -> 144)
  BG(IN_ANG, OUT_SAME, 16, 1)       // AngConv.2 dgrad (A*A x 16 -> 16)
  BG(IN_SAME, OUT_VIEWS, 16, 2)     // AngConv.0 dgrad (16 -> A*A x 64), accumulates into dX
  BG(IN_CHK_H, OUT_SAME, 32, 1)     // EPIConv.2 dgrad, horizontal
  BG(IN_CHK_V, OUT_SAME, 32, 1)     //                  vertical
  BG(IN_LINE_H, OUT_EPIH, 32, 2)    // EPIConv.0 dgrad, horizontal (32 -> A x 64), accumulates into dX
  BG(IN_LINE_V, OUT_EPIV, 32, 2)    //                  vertical
  BG(IN_CONV3, OUT_SAME, 64, 2)     // 3x3 dgrad fallback
#undef BG
  return LFSR_E_ARG;
}

// ... with a second residual operand (dx = conv^T(dy) + r1 + r2, no mask): the group skip of DistgSSR's backward rides on the data gradient of the group's first block
// instead of a separate read-modify-write pass over dx.  LFSR_E_ARG = not covered (the caller adds r2 itself).
int lfsr_conv3x3_bwd_data_r2(const float* dy, int dy_stride, const float* wT_packed, float* dx, const float* r1, const float* r2, int n_img, int h, int w, hipStream_t st) {
  if ((dy_stride & 3) || !r1 || !r2) return LFSR_E_ARG;
  const char* sel = lfsr_conv3_dgrad_sel();
  if (sel) return LFSR_E_ARG;      // (lab selections of another 3x3 form: the plain path)
  LfsrOpTimer op_t("conv3x3_dgrad", n_img, h * w, st);
  return lfsr_conv3x3_wino4_launch(dy, dy_stride, 0, wT_packed + LFSR_CONV3_DIRECT_FLOATS + LFSR_CONV3_WINO2_FLOATS, dx, 64, 0, r1, 64, 0, r2, 64, 0, nullptr, 0, 0, 1.0f,
                                   n_img, h, w, 1.0f, st);
}

int lfsr_conv3x3_bwd_data(const float* dy, int dy_stride, int dy_choff, const float* wT_packed, float* dx, int dx_stride, int dx_choff,
                          const float* r1, int r1_stride, int r1_choff, const float* mk, int mk_stride, int mk_choff, float mk_slope,
                          int n_img, int h, int w, hipStream_t st) {
  LfsrOpTimer op_t("conv3x3_dgrad", n_img, h * w, st);
  const bool al = !((dy_stride | dy_choff | dx_stride | dx_choff) & 3) && (!r1 || !((r1_stride | r1_choff) & 3)) && (!mk || !((mk_stride | mk_choff) & 3));
  {
    const char* sel = lfsr_conv3_dgrad_sel();
    if (al && !(sel && (sel[0] == 'h' || sel[0] == 'g'))) {
      const int rc = lfsr_conv3x3_wino_launch(dy, dy_stride, dy_choff, wT_packed + LFSR_CONV3_DIRECT_FLOATS, wT_packed, dx, dx_stride, dx_choff, r1, r1_stride, r1_choff,
                                              nullptr, 0, 0, mk, mk_stride, mk_choff, mk_slope, n_img, h, w, 1.0f, sel, st);
      if (rc != LFSR_E_ARG) return rc;   // (E_ARG: a geometry the Winograd launchers do not cover -> the direct kernel)
    }
  }
  if (al)
    return lfsr_conv3x3_halo_launch(dy, dy_stride, dy_choff, wT_packed, dx, dx_stride, dx_choff, r1, r1_stride, r1_choff, nullptr, 0, 0,
                                    mk, mk_stride, mk_choff, mk_slope, n_img, h, w, 1.0f, st);
  LfsrGemm g{};
  g.in_mode = LFSR_IN_CONV3; g.out_mode = LFSR_OUT_SAME; g.cin = 64;
  g.X = dy; g.x_stride = dy_stride; g.x_choff = dy_choff; g.Wp = wT_packed; g.Y = dx; g.y_stride = dx_stride; g.y_choff = dx_choff;
  g.R1 = r1; g.r1_stride = r1_stride; g.r1_choff = r1_choff; g.Mk = mk; g.mk_stride = mk_stride; g.mk_choff = mk_choff; g.mk_slope = mk_slope;
  g.M = n_img * h * w; g.N = 64; g.A = 1; g.h = h; g.w = w; g.ntaps = 9; g.CH = 64;
  return lfsr_bwd_gemm(g, st);
}

int lfsr_head_bwd_data(const float* dout, const float* wf, float* df, float* g16, int B, int A, int h, int w, int s, hipStream_t st) {
  long long total = (long long)B * A * A * h * w * 16;
  unsigned grid = lfsr_blocks(total, 256);
  if (grid > 4096) grid = 4096;
  switch (s) {
    case 2: hipLaunchKernelGGL((k_head_bwd<2>), dim3(grid), dim3(256), 0, st, dout, wf, df, g16, B, A, h, w); break;
    case 3: hipLaunchKernelGGL((k_head_bwd<3>), dim3(grid), dim3(256), 0, st, dout, wf, df, g16, B, A, h, w); break;
    case 4: hipLaunchKernelGGL((k_head_bwd<4>), dim3(grid), dim3(256), 0, st, dout, wf, df, g16, B, A, h, w); break;
    default: return LFSR_E_ARG;
  }
  LFSR_CHECK_LAUNCH();
  return LFSR_OK;
}

int lfsr_colsum(const float* g, int M, int N, float* partial, int* nblk_out, hipStream_t st) {
  if (N > 16) return LFSR_E_ARG;
  int rows = 4096;
  int nblk = (M + rows - 1) / rows;
  if (nblk_out) *nblk_out = nblk;
  if (!g) return LFSR_OK;   // size query
  hipLaunchKernelGGL(k_colsum, dim3(nblk), dim3(256), 0, st, g, M, N, partial, rows);
  LFSR_CHECK_LAUNCH();
  return LFSR_OK;
}

int lfsr_head_fold_bwd(const float* dWf, const float* colsum_partial, int nblk, const float* w0, const float* b0, const float* w2,
                       float* dw0, float* db0, float* dw2, int s, hipStream_t st) {
  hipLaunchKernelGGL(k_head_fold_bwd, dim3(64), dim3(256), 0, st, dWf, colsum_partial, nblk, w0, b0, w2, dw0, db0, dw2, s * s);
  LFSR_CHECK_LAUNCH();
  return LFSR_OK;
}

int lfsr_init_gather9(const float* x, float* xg, int B, int A, int h, int w, hipStream_t st) {
  long long total = (long long)B * A * A * h * w * 16;
  unsigned grid = lfsr_blocks(total, 256);
  if (grid > 4096) grid = 4096;
  hipLaunchKernelGGL(k_init_gather9, dim3(grid), dim3(256), 0, st, x, xg, B, A, h, w);
  LFSR_CHECK_LAUNCH();
  return LFSR_OK;
}

int lfsr_pack_weight_chunkT(const float* w, float* out, int O, int C, int ch, int perm, hipStream_t st) {
  if (!w || !out || O <= 0 || C <= 0 || ch <= 0 || O % ch) return LFSR_E_ARG;
  long long total = (long long)(O / ch) * npad32(C) * ch;
  hipLaunchKernelGGL(k_pack_chunkT, dim3(lfsr_blocks(total, 256)), dim3(256), 0, st, w, out, O, C, ch, perm, npad32(C));
  LFSR_CHECK_LAUNCH();
  return LFSR_OK;
}
