// a1-a7: bit-exact integer-indexing primitives on the reference's NCHW layouts, plus the
// NCHW <-> VCL (view-major channel-last) converters.  All HBM-bound gathers: one thread per OUTPUT
// element group, stores fully coalesced, loads touch whole lines within a block.
#include "lfsr_common.h"

namespace {

// period-2n edge-including mirror (np.pad 'symmetric'); utils/utils.py:137-149 builds it from flips
__device__ __forceinline__ int sym_index(int i, int n) {
  int p = 2 * n;
  int r = i % p;
  if (r < 0) r += p;
  return r < n ? r : p - 1 - r;
}

// ---- a1 / a2 --------------------------------------------------------------------------------
// to_macpi = 1: out[b,c,y*A+u,x*A+v] = in[b,c,u*h+y,v*w+x]   (DistgSSR.py:145-155)
// to_macpi = 0: out[b,c,u*h+y,v*w+x] = in[b,c,y*A+u,x*A+v]   (DistgSSR.py:134-142)
template <typename T, int TO_MACPI>
__global__ __launch_bounds__(256) void k_sai_macpi(const T* __restrict__ in, T* __restrict__ out, int planes, int A, int h, int w) {
  const int Wd = A * w, Hd = A * h;
  const long long total = (long long)planes * Hd * Wd;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    int col = (int)(i % Wd);
    long long t = i / Wd;
    int row = (int)(t % Hd);
    long long pl = t / Hd;
    int srow, scol;
    if (TO_MACPI) {
      int y = row / A, u = row - y * A, x = col / A, v = col - x * A;
      srow = u * h + y; scol = v * w + x;
    } else {
      int u = row / h, y = row - u * h, v = col / w, x = col - v * w;
      srow = y * A + u; scol = x * A + v;
    }
    out[i] = in[(pl * Hd + srow) * Wd + scol];
  }
}


// ---- a1 / a2, vectorised: 16-B global accesses on both sides, permutation through LDS ---------------------------------
// One block = G consecutive y of one (b,c) plane = A*G input rows <-> A*G output rows (row length A*w floats, A*w % 4 == 0).
// LDS holds the SAI-side image [u][g][A*w]; the MacPI side element (g, u, x*A+v) is LDS[u][g][v*w+x].
// AT = the angular resolution when it is a compile-time constant (5: BASELINE; divisions by A become multiplies), 0 = run-time A.
// All of a thread's loads are issued before its first LDS store (up to 8 x 16 B in flight per thread, 32 KB per block); the (row, column)
// of a thread's k-th vector is stepped, not divided out.
template <int TO_MACPI, int AT>
__global__ __launch_bounds__(256) void k_sai_macpi_lds(const float* __restrict__ in, float* __restrict__ out, int A_rt, int h, int w, int G) {
  extern __shared__ float sm[];
  const int A = AT ? AT : A_rt;
  const int Wd = A * w, Hd = A * h, W4 = Wd / 4;
  const int ygroups = (h + G - 1) / G;
  const long long plane = blockIdx.x / ygroups;
  const int y0 = (blockIdx.x % ygroups) * G;
  const int g_n = min(G, h - y0);
  const float* pin = in + plane * Hd * Wd;
  float* pout = out + plane * Hd * Wd;
  const int nvec = A * g_n * W4;    // float4 per side
  const int dr = 256 / W4, dc = 256 - dr * W4;   // vector i + 256 sits dr rows and dc columns further (with carry)
  constexpr int NV = 8;                          // vectors per thread and pass
  for (int base = 0; base < nvec; base += 256 * NV) {
    // global side A (source): rows in the order they are stored in; LDS image [u][g][:] = the SAI-side rows
    float4 v[NV];
    int c4 = (base + threadIdx.x) % W4, r = (base + threadIdx.x) / W4;
    int cc[NV], rr[NV];
#pragma unroll
    for (int q = 0; q < NV; ++q) {
      cc[q] = c4; rr[q] = r;
      const bool ok = base + threadIdx.x + 256 * q < nvec;
      // TO_MACPI: source row r = u * g_n + g -> SAI row u*h + y0 + g;  else: source row r = g * A + u -> MacPI row (y0+g)*A + u
      int srow;
      if (TO_MACPI) { const int u = r / g_n, g = r - u * g_n; srow = u * h + y0 + g; }
      else srow = y0 * A + r;
      v[q] = ok ? *reinterpret_cast<const float4*>(pin + (long long)srow * Wd + c4 * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
      c4 += dc; r += dr;
      if (c4 >= W4) { c4 -= W4; r += 1; }
    }
#pragma unroll
    for (int q = 0; q < NV; ++q) {
      if (base + threadIdx.x + 256 * q < nvec) {
        int u, g;
        if (TO_MACPI) { u = rr[q] / g_n; g = rr[q] - u * g_n; } else { g = rr[q] / A; u = rr[q] - g * A; }
        *reinterpret_cast<float4*>(sm + (u * G + g) * Wd + cc[q] * 4) = v[q];
      }
    }
  }
  __syncthreads();
  {
    int c4 = threadIdx.x % W4, r = threadIdx.x / W4;
    for (int i = threadIdx.x; i < nvec; i += 256) {
      float o[4];
      long long drow;
      if (TO_MACPI) {           // MacPI rows ((y0+g)*A + u), 4 consecutive columns: element (x*A + v) = LDS[u][g][v*w + x]
        const int g = r / A, u = r - g * A;
        const float* row = sm + (u * G + g) * Wd;
#pragma unroll
        for (int e = 0; e < 4; ++e) { const int col = c4 * 4 + e, x = col / A, vv = col - x * A; o[e] = row[vv * w + x]; }
        drow = (long long)(y0 + g) * A + u;
      } else {                  // SAI rows (u*h + y0+g): element (v*w + x) = LDS[u][g][x*A + v]
        const int u = r / g_n, g = r - u * g_n;
        const float* row = sm + (u * G + g) * Wd;
#pragma unroll
        for (int e = 0; e < 4; ++e) { const int col = c4 * 4 + e, vv = col / w, x = col - vv * w; o[e] = row[x * A + vv]; }
        drow = (long long)u * h + y0 + g;
      }
      *reinterpret_cast<float4*>(pout + drow * Wd + c4 * 4) = make_float4(o[0], o[1], o[2], o[3]);
      c4 += dc; r += dr;
      if (c4 >= W4) { c4 -= W4; r += 1; }
    }
  }
}

// ---- a3, vectorised (W % 4 == 0, 4-byte elements): one thread = 4 consecutive x of one (b,c,y,i): r float4 loads (one per j
// plane), an r x 4 register transpose, r float4 stores of 4r contiguous outputs
template <int R>
__global__ __launch_bounds__(256) void k_pixel_shuffle2d_vec(const float* __restrict__ in, float* __restrict__ out, int BC, int H, int W) {
  const int W4 = W / 4;
  const long long total = (long long)BC * H * R * W4;
  for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long long)gridDim.x * 256) {
    int x4 = (int)(idx % W4);
    long long t = idx / W4;
    int i = (int)(t % R); t /= R;
    int y = (int)(t % H);
    long long bc = t / H;
    float v[R][4];
#pragma unroll
    for (int j = 0; j < R; ++j) {
      float4 q = *reinterpret_cast<const float4*>(in + ((bc * R * R + i * R + j) * H + y) * W + x4 * 4);
      v[j][0] = q.x; v[j][1] = q.y; v[j][2] = q.z; v[j][3] = q.w;
    }
    float* o = out + (bc * H * R + (long long)y * R + i) * ((long long)W * R) + (long long)x4 * 4 * R;
    float flat[4 * R];
#pragma unroll
    for (int xx = 0; xx < 4; ++xx)
#pragma unroll
      for (int j = 0; j < R; ++j) flat[xx * R + j] = v[j][xx];
#pragma unroll
    for (int q = 0; q < R; ++q) *reinterpret_cast<float4*>(o + q * 4) = make_float4(flat[4 * q], flat[4 * q + 1], flat[4 * q + 2], flat[4 * q + 3]);
  }
}

// ---- a3 for r = 4 (the x4 model) and r = 2, 4-byte elements, < 2^31 elements: one thread = ONE input position (b c, y, x) and all r*r planes.  A lane's r*r dword
// loads are coalesced across the wave (consecutive x: 256 B per instruction and plane) and all in flight before the first store; its r stores are r floats each of r
// different output rows, and CONSECUTIVE LANES STORE CONSECUTIVE 16-B (r = 4) PIECES: 1 KB contiguous per store instruction.  The round-1 form (4 consecutive x per
// thread, 16-B loads) stored 16 B per lane at a 64-B lane stride -- every store instruction touched 64 partial 64-B segments -- and paid three 64-bit divisions per
// thread; 32-bit index arithmetic here.
template <int R>
__global__ __launch_bounds__(256) void k_pixel_shuffle2d_x(const float* __restrict__ in, float* __restrict__ out, unsigned total, unsigned H, unsigned W) {
  const unsigned HW = H * W;
  for (unsigned idx = blockIdx.x * 256u + threadIdx.x; idx < total; idx += gridDim.x * 256u) {
    const unsigned bc = idx / HW, rem = idx - bc * HW, y = rem / W, x = rem - y * W;
    const float* src = in + (size_t)bc * (R * R) * HW + rem;
    float v[R][R];
#pragma unroll
    for (int i = 0; i < R; ++i)
#pragma unroll
      for (int j = 0; j < R; ++j) v[i][j] = src[(size_t)(i * R + j) * HW];
    float* dst = out + ((size_t)bc * H * R + (size_t)y * R) * ((size_t)W * R) + (size_t)x * R;
#pragma unroll
    for (int i = 0; i < R; ++i) {
      if (R == 4) *reinterpret_cast<float4*>(dst + (size_t)i * W * R) = make_float4(v[i][0], v[i][1], v[i][2], v[i][3]);
      else *reinterpret_cast<float2*>(dst + (size_t)i * W * R) = make_float2(v[i][0], v[i][1]);
    }
  }
}

// ---- a3: out[b,c,y*r+i,x*r+j] = in[b,c*r*r+i*r+j,y,x] ------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void k_pixel_shuffle2d(const T* __restrict__ in, T* __restrict__ out, int BC, int r, int H, int W) {
  const int Wo = W * r, Ho = H * r;
  const long long total = (long long)BC * Ho * Wo;
  for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long long)gridDim.x * blockDim.x) {
    int xo = (int)(idx % Wo);
    long long t = idx / Wo;
    int yo = (int)(t % Ho);
    long long bc = t / Ho;
    int y = yo / r, i = yo - y * r, x = xo / r, j = xo - x * r;
    out[idx] = in[((bc * r * r + i * r + j) * H + y) * W + x];
  }
}

// ---- a4: out[b,c,y,x*f+k] = in[b,k*C+c,y,x] ----------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void k_pixel_shuffle1d(const T* __restrict__ in, T* __restrict__ out, int B, int C, int f, int H, int W) {
  const int Wo = W * f;
  const long long total = (long long)B * C * H * Wo;
  for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long long)gridDim.x * blockDim.x) {
    int xo = (int)(idx % Wo);
    long long t = idx / Wo;
    int y = (int)(t % H);
    t /= H;
    int c = (int)(t % C);
    int b = (int)(t / C);
    int x = xo / f, k = xo - x * f;
    out[idx] = in[(((long long)b * f * C + (long long)k * C + c) * H + y) * W + x];
  }
}

// ---- a5 -----------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void k_image_extend(const T* __restrict__ in, T* __restrict__ out, int N, int h, int w, int top, int left, int Ho, int Wo) {
  const long long total = (long long)N * Ho * Wo;
  for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long long)gridDim.x * blockDim.x) {
    int xo = (int)(idx % Wo);
    long long t = idx / Wo;
    int yo = (int)(t % Ho);
    long long n = t / Ho;
    out[idx] = in[(n * h + sym_index(yo - top, h)) * w + sym_index(xo - left, w)];
  }
}

// ---- a6: sub[n1,n2,a1*P+y,a2*P+x] = data[a1*h0+sym(n1*S+y-bdr,h0), a2*w0+sym(n2*S+x-bdr,w0)] ----
template <typename T>
__global__ __launch_bounds__(256) void k_lf_divide(const T* __restrict__ in, T* __restrict__ out, int A, int h0, int w0, int P, int S, int bdr, int numU, int numV) {
  const int AP = A * P;
  const long long total = (long long)numU * numV * AP * AP;
  for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long long)gridDim.x * blockDim.x) {
    int col = (int)(idx % AP);
    long long t = idx / AP;
    int row = (int)(t % AP);
    t /= AP;
    int n2 = (int)(t % numV), n1 = (int)(t / numV);
    int a1 = row / P, y = row - a1 * P, a2 = col / P, x = col - a2 * P;
    int sy = sym_index(n1 * S + y - bdr, h0), sx = sym_index(n2 * S + x - bdr, w0);
    out[idx] = in[((long long)a1 * h0 + sy) * ((long long)A * w0) + (long long)a2 * w0 + sx];
  }
}

// ---- a7: out[a1,a2,Y,X] = sub[Y/S', X/S', a1*pz+bdr'+Y%S', a2*pz+bdr'+X%S'] ----------------------
template <typename T>
__global__ __launch_bounds__(256) void k_lf_integrate(const T* __restrict__ in, T* __restrict__ out, int A, int numU, int numV, int pz, int stride, int bdr, int h, int w) {
  const long long total = (long long)A * A * h * w;
  const int AP = A * pz;
  for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long long)gridDim.x * blockDim.x) {
    int X = (int)(idx % w);
    long long t = idx / w;
    int Y = (int)(t % h);
    t /= h;
    int a2 = (int)(t % A), a1 = (int)(t / A);
    int n1 = Y / stride, yy = Y - n1 * stride, n2 = X / stride, xx = X - n2 * stride;
    out[idx] = in[(((long long)n1 * numV + n2) * AP + a1 * pz + bdr + yy) * AP + a2 * pz + bdr + xx];
  }
}

// ---- NCHW <-> VCL --------------------------------------------------------------------------------
// One block = 64 consecutive pixels of one view-plane row group x all C channels through an LDS tile, so
// both the NCHW side (pixels contiguous) and the VCL side (channels contiguous) move whole lines.
template <int TO_VCL>
__global__ __launch_bounds__(256) void k_nchw_vcl(const float* __restrict__ src, float* __restrict__ dst, int stride, int choff,
                                                  int B, int C, int A, int h, int w, int layout) {
  __shared__ float tile[64][65];
  const int AA = A * A;
  const long long npix = (long long)B * AA * h * w;
  const long long p0 = (long long)blockIdx.x * 64;
  const int c0 = blockIdx.y * 64;
  const int Hd = A * h, Wd = A * w;
  auto nchw_off = [&](long long p, int c) -> long long {
    int x = (int)(p % w);
    long long t = p / w;
    int y = (int)(t % h);
    t /= h;
    int view = (int)(t % AA);
    int b = (int)(t / AA);
    int u = view / A, v = view - u * A;
    int row = layout == 0 ? u * h + y : y * A + u;
    int col = layout == 0 ? v * w + x : x * A + v;
    return (((long long)b * C + c) * Hd + row) * Wd + col;
  };
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;  // 64 x 4
  if (TO_VCL) {
    for (int cc = ty; cc < 64; cc += 4) {  // read: pixel fastest
      long long p = p0 + tx;
      int c = c0 + cc;
      tile[cc][tx] = (p < npix && c < C) ? src[nchw_off(p, c)] : 0.f;
    }
    __syncthreads();
    for (int pp = ty; pp < 64; pp += 4) {  // write: channel fastest
      long long p = p0 + pp;
      int c = c0 + tx;
      if (p < npix && c < C) dst[p * stride + choff + c] = tile[tx][pp];
    }
  } else {
    for (int pp = ty; pp < 64; pp += 4) {
      long long p = p0 + pp;
      int c = c0 + tx;
      tile[tx][pp] = (p < npix && c < C) ? src[p * stride + choff + c] : 0.f;
    }
    __syncthreads();
    for (int cc = ty; cc < 64; cc += 4) {
      long long p = p0 + tx;
      int c = c0 + cc;
      if (p < npix && c < C) dst[nchw_off(p, c)] = tile[cc][tx];
    }
  }
}

inline unsigned grid_for(long long total) {
  long long b = (total + 255) / 256;
  if (b > 256LL * 32) b = 256LL * 32;  // grid-stride the rest
  return (unsigned)(b < 1 ? 1 : b);
}

}  // namespace

// LFintegrate factored for patch-sharded ranks (utils/utils.py:169-178 keeps only the centre stride x stride of every view of every SR patch):
// crop runs where a patch was computed, so a quarter of the SR bytes cross xGMI; place runs on the rank that assembles the scene.
template <typename T>
__global__ __launch_bounds__(256) void k_lf_crop_tiles(const T* __restrict__ in, T* __restrict__ tiles, int A, int count, int pz, int stride, int bdr) {
  const long long total = (long long)count * A * A * stride * stride;
  const int AP = A * pz;
  for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long long)gridDim.x * blockDim.x) {
    int xx = (int)(idx % stride);
    long long t = idx / stride;
    int yy = (int)(t % stride);
    t /= stride;
    int a2 = (int)(t % A);
    t /= A;
    int a1 = (int)(t % A);
    long long n = t / A;
    tiles[idx] = in[(n * AP + a1 * pz + bdr + yy) * AP + a2 * pz + bdr + xx];
  }
}

template <typename T>
__global__ __launch_bounds__(256) void k_lf_place_tiles(const T* __restrict__ tiles, T* __restrict__ out, int A, int numV, int first, int count, int stride, int h, int w) {
  const long long total = (long long)count * A * A * stride * stride;
  for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long long)gridDim.x * blockDim.x) {
    int xx = (int)(idx % stride);
    long long t = idx / stride;
    int yy = (int)(t % stride);
    t /= stride;
    int a2 = (int)(t % A);
    t /= A;
    int a1 = (int)(t % A);
    int n = first + (int)(t / A);
    int n1 = n / numV, n2 = n - n1 * numV;
    int Y = n1 * stride + yy, X = n2 * stride + xx;
    if (Y < h && X < w) out[(((long long)a1 * A + a2) * h + Y) * w + X] = tiles[idx];
  }
}

#define DISPATCH_ELEM(KERNEL, GRID, STREAM, ...)                                                        \
  do {                                                                                                  \
    if (elem_bytes == 4)                                                                                \
      hipLaunchKernelGGL((KERNEL<uint32_t>), dim3(GRID), dim3(256), 0, STREAM, (const uint32_t*)in, (uint32_t*)out, __VA_ARGS__); \
    else                                                                                                \
      hipLaunchKernelGGL((KERNEL<uint16_t>), dim3(GRID), dim3(256), 0, STREAM, (const uint16_t*)in, (uint16_t*)out, __VA_ARGS__); \
  } while (0)

extern "C" {

const char* lfsr_version(void) { return "lfsr_hip 0.2 gfx950 (fp32 MFMA)"; }
#ifdef LFSR_CONV_DIAG
}
float* g_lfsr_diag_buf = nullptr;
extern "C" {
int lfsr_diag_set_buffer(float* buf) { g_lfsr_diag_buf = buf; return LFSR_OK; }
#endif

static int bad_elem(int e) { return !(e == 2 || e == 4); }

int lfsr_sai2macpi(const void* in, void* out, int B, int C, int A, int h, int w, int elem_bytes, void* stream) {
  if (B < 0 || C < 0 || A <= 0 || h <= 0 || w <= 0 || bad_elem(elem_bytes)) return LFSR_E_ARG;
  long long total = (long long)B * C * A * h * A * w;
  if (total == 0) return LFSR_OK;  // empty tensors carry NULL pointers
  if (!in || !out) return LFSR_E_ARG;
  if (elem_bytes == 4 && (A * w) % 4 == 0 && !(((uintptr_t)in | (uintptr_t)out) & 15)) {
    int G = 8;                                             // y rows per block: A*G*A*w*4 bytes of LDS
    while (G > 1 && (size_t)A * G * A * w * 4 > 48 * 1024) G >>= 1;
    if ((size_t)A * G * A * w * 4 <= 64 * 1024) {
      long long nblk = (long long)B * C * ((h + G - 1) / G);
      if (A == 5) hipLaunchKernelGGL((k_sai_macpi_lds<1, 5>), dim3((unsigned)nblk), dim3(256), (size_t)A * G * A * w * 4, lfsr_stream(stream), (const float*)in, (float*)out, A, h, w, G);
      else hipLaunchKernelGGL((k_sai_macpi_lds<1, 0>), dim3((unsigned)nblk), dim3(256), (size_t)A * G * A * w * 4, lfsr_stream(stream), (const float*)in, (float*)out, A, h, w, G);
      LFSR_CHECK_LAUNCH();
      return LFSR_OK;
    }
  }
  if (elem_bytes == 4)
    hipLaunchKernelGGL((k_sai_macpi<uint32_t, 1>), dim3(grid_for(total)), dim3(256), 0, lfsr_stream(stream), (const uint32_t*)in, (uint32_t*)out, B * C, A, h, w);
  else
    hipLaunchKernelGGL((k_sai_macpi<uint16_t, 1>), dim3(grid_for(total)), dim3(256), 0, lfsr_stream(stream), (const uint16_t*)in, (uint16_t*)out, B * C, A, h, w);
  LFSR_CHECK_LAUNCH();
  return LFSR_OK;
}

int lfsr_macpi2sai(const void* in, void* out, int B, int C, int A, int h, int w, int elem_bytes, void* stream) {
  if (B < 0 || C < 0 || A <= 0 || h <= 0 || w <= 0 || bad_elem(elem_bytes)) return LFSR_E_ARG;
  long long total = (long long)B * C * A * h * A * w;
  if (total == 0) return LFSR_OK;  // empty tensors carry NULL pointers
  if (!in || !out) return LFSR_E_ARG;
  if (elem_bytes == 4 && (A * w) % 4 == 0 && !(((uintptr_t)in | (uintptr_t)out) & 15)) {
    int G = 8;                                             // y rows per block: A*G*A*w*4 bytes of LDS
    while (G > 1 && (size_t)A * G * A * w * 4 > 48 * 1024) G >>= 1;
    if ((size_t)A * G * A * w * 4 <= 64 * 1024) {
      long long nblk = (long long)B * C * ((h + G - 1) / G);
      if (A == 5) hipLaunchKernelGGL((k_sai_macpi_lds<0, 5>), dim3((unsigned)nblk), dim3(256), (size_t)A * G * A * w * 4, lfsr_stream(stream), (const float*)in, (float*)out, A, h, w, G);
      else hipLaunchKernelGGL((k_sai_macpi_lds<0, 0>), dim3((unsigned)nblk), dim3(256), (size_t)A * G * A * w * 4, lfsr_stream(stream), (const float*)in, (float*)out, A, h, w, G);
      LFSR_CHECK_LAUNCH();
      return LFSR_OK;
    }
  }
  if (elem_bytes == 4)
    hipLaunchKernelGGL((k_sai_macpi<uint32_t, 0>), dim3(grid_for(total)), dim3(256), 0, lfsr_stream(stream), (const uint32_t*)in, (uint32_t*)out, B * C, A, h, w);
  else
    hipLaunchKernelGGL((k_sai_macpi<uint16_t, 0>), dim3(grid_for(total)), dim3(256), 0, lfsr_stream(stream), (const uint16_t*)in, (uint16_t*)out, B * C, A, h, w);
  LFSR_CHECK_LAUNCH();
  return LFSR_OK;
}

int lfsr_pixel_shuffle2d(const void* in, void* out, int B, int C, int r, int H, int W, int elem_bytes, void* stream) {
  if (B < 0 || C < 0 || r <= 0 || H <= 0 || W <= 0 || bad_elem(elem_bytes)) return LFSR_E_ARG;
  long long total = (long long)B * C * H * r * W * r;
  if (total == 0) return LFSR_OK;
  if (!in || !out) return LFSR_E_ARG;
  if (elem_bytes == 4 && (r == 4 || r == 2) && total < (1LL << 31) && !(((uintptr_t)in | (uintptr_t)out) & 15) && ((long long)W * r * 4) % (r == 4 ? 16 : 8) == 0) {
    const unsigned npos = (unsigned)((long long)B * C * H * W);
    unsigned grid = grid_for(npos);
    hipStream_t st = lfsr_stream(stream);
    if (r == 4) hipLaunchKernelGGL((k_pixel_shuffle2d_x<4>), dim3(grid), dim3(256), 0, st, (const float*)in, (float*)out, npos, (unsigned)H, (unsigned)W);
    else hipLaunchKernelGGL((k_pixel_shuffle2d_x<2>), dim3(grid), dim3(256), 0, st, (const float*)in, (float*)out, npos, (unsigned)H, (unsigned)W);
    LFSR_CHECK_LAUNCH();
    return LFSR_OK;
  }
  if (elem_bytes == 4 && W % 4 == 0 && r >= 2 && r <= 5 && !(((uintptr_t)in | (uintptr_t)out) & 15)) {
    long long work = (long long)B * C * H * r * (W / 4);
    unsigned grid = grid_for(work);
    hipStream_t st = lfsr_stream(stream);
    const float* fi = (const float*)in; float* fo = (float*)out;
    switch (r) {
      case 2: hipLaunchKernelGGL((k_pixel_shuffle2d_vec<2>), dim3(grid), dim3(256), 0, st, fi, fo, B * C, H, W); break;
      case 3: hipLaunchKernelGGL((k_pixel_shuffle2d_vec<3>), dim3(grid), dim3(256), 0, st, fi, fo, B * C, H, W); break;
      case 4: hipLaunchKernelGGL((k_pixel_shuffle2d_vec<4>), dim3(grid), dim3(256), 0, st, fi, fo, B * C, H, W); break;
      default: hipLaunchKernelGGL((k_pixel_shuffle2d_vec<5>), dim3(grid), dim3(256), 0, st, fi, fo, B * C, H, W); break;
    }
    LFSR_CHECK_LAUNCH();
    return LFSR_OK;
  }
  DISPATCH_ELEM(k_pixel_shuffle2d, grid_for(total), lfsr_stream(stream), B * C, r, H, W);
  LFSR_CHECK_LAUNCH();
  return LFSR_OK;
}

int lfsr_pixel_shuffle1d(const void* in, void* out, int B, int C, int f, int H, int W, int elem_bytes, void* stream) {
  if (B < 0 || C < 0 || f <= 0 || H <= 0 || W <= 0 || bad_elem(elem_bytes)) return LFSR_E_ARG;
  long long total = (long long)B * C * H * W * f;
  if (total == 0) return LFSR_OK;
  if (!in || !out) return LFSR_E_ARG;
  DISPATCH_ELEM(k_pixel_shuffle1d, grid_for(total), lfsr_stream(stream), B, C, f, H, W);
  LFSR_CHECK_LAUNCH();
  return LFSR_OK;
}

int lfsr_image_extend(const void* in, void* out, int N, int h, int w, int top, int bottom, int left, int right,
                      int elem_bytes, void* stream) {
  if (N < 0 || h <= 0 || w <= 0 || top < 0 || bottom < 0 || left < 0 || right < 0 || bad_elem(elem_bytes)) return LFSR_E_ARG;
  int Ho = h + top + bottom, Wo = w + left + right;
  long long total = (long long)N * Ho * Wo;
  if (total == 0) return LFSR_OK;
  if (!in || !out) return LFSR_E_ARG;
  DISPATCH_ELEM(k_image_extend, grid_for(total), lfsr_stream(stream), N, h, w, top, left, Ho, Wo);
  LFSR_CHECK_LAUNCH();
  return LFSR_OK;
}

int lfsr_lf_divide(const void* in, void* out, int A, int h0, int w0, int P, int S, int elem_bytes, int* num_u, int* num_v, void* stream) {
  if (A <= 0 || h0 <= 0 || w0 <= 0 || P <= 0 || S <= 0 || P < S || bad_elem(elem_bytes)) return LFSR_E_ARG;
  int bdr = (P - S) / 2;
  int numU = (h0 + bdr * 2 - 1) / S, numV = (w0 + bdr * 2 - 1) / S;
  if (num_u) *num_u = numU;
  if (num_v) *num_v = numV;
  if (!out) return LFSR_OK;
  if (!in) return LFSR_E_ARG;
  // the reference's unfold count equals numU*numV only when the padded extent admits exactly numU windows
  // (utils/utils.py:159-164 rearrange would raise otherwise)
  if ((h0 + 2 * bdr + S - 1 - P) / S + 1 != numU || (w0 + 2 * bdr + S - 1 - P) / S + 1 != numV) return LFSR_E_ARG;
  long long total = (long long)numU * numV * A * P * A * P;
  if (total == 0) return LFSR_OK;
  DISPATCH_ELEM(k_lf_divide, grid_for(total), lfsr_stream(stream), A, h0, w0, P, S, bdr, numU, numV);
  LFSR_CHECK_LAUNCH();
  return LFSR_OK;
}

int lfsr_lf_integrate(const void* in, void* out, int A, int numU, int numV, int pz, int stride, int h, int w, int elem_bytes, void* stream) {
  if (!in || !out || A <= 0 || numU <= 0 || numV <= 0 || pz <= 0 || stride <= 0 || pz < stride || h <= 0 || w <= 0 || bad_elem(elem_bytes)) return LFSR_E_ARG;
  if (h > numU * stride || w > numV * stride) return LFSR_E_ARG;
  int bdr = (pz - stride) / 2;
  long long total = (long long)A * A * h * w;
  DISPATCH_ELEM(k_lf_integrate, grid_for(total), lfsr_stream(stream), A, numU, numV, pz, stride, bdr, h, w);
  LFSR_CHECK_LAUNCH();
  return LFSR_OK;
}

int lfsr_lf_crop_tiles(const void* in, void* out, int A, int count, int pz, int stride, int elem_bytes, void* stream) {
  if (A <= 0 || count < 0 || pz <= 0 || stride <= 0 || pz < stride || bad_elem(elem_bytes)) return LFSR_E_ARG;
  long long total = (long long)count * A * A * stride * stride;
  if (total == 0) return LFSR_OK;
  if (!in || !out) return LFSR_E_ARG;
  int bdr = (pz - stride) / 2;
  DISPATCH_ELEM(k_lf_crop_tiles, grid_for(total), lfsr_stream(stream), A, count, pz, stride, bdr);
  LFSR_CHECK_LAUNCH();
  return LFSR_OK;
}

int lfsr_lf_place_tiles(const void* in, void* out, int A, int numU, int numV, int first, int count, int stride, int h, int w, int elem_bytes, void* stream) {
  if (A <= 0 || numU <= 0 || numV <= 0 || first < 0 || count < 0 || (long long)first + count > (long long)numU * numV || stride <= 0 || h <= 0 || w <= 0 ||
      h > numU * stride || w > numV * stride || bad_elem(elem_bytes)) return LFSR_E_ARG;
  long long total = (long long)count * A * A * stride * stride;
  if (total == 0) return LFSR_OK;
  if (!in || !out) return LFSR_E_ARG;
  DISPATCH_ELEM(k_lf_place_tiles, grid_for(total), lfsr_stream(stream), A, numV, first, count, stride, h, w);
  LFSR_CHECK_LAUNCH();
  return LFSR_OK;
}

int lfsr_nchw_to_vcl(const float* in, float* out, int out_stride, int out_choff, int B, int C, int A, int h, int w, int layout, void* stream) {
  if (!in || !out || B <= 0 || C <= 0 || A <= 0 || h <= 0 || w <= 0 || (layout != 0 && layout != 1) || out_stride < out_choff + C) return LFSR_E_ARG;
  long long npix = (long long)B * A * A * h * w;
  dim3 grid((unsigned)((npix + 63) / 64), (unsigned)((C + 63) / 64));
  hipLaunchKernelGGL((k_nchw_vcl<1>), grid, dim3(256), 0, lfsr_stream(stream), in, out, out_stride, out_choff, B, C, A, h, w, layout);
  LFSR_CHECK_LAUNCH();
  return LFSR_OK;
}

int lfsr_vcl_to_nchw(const float* in, int in_stride, int in_choff, float* out, int B, int C, int A, int h, int w, int layout, void* stream) {
  if (!in || !out || B <= 0 || C <= 0 || A <= 0 || h <= 0 || w <= 0 || (layout != 0 && layout != 1) || in_stride < in_choff + C) return LFSR_E_ARG;
  long long npix = (long long)B * A * A * h * w;
  dim3 grid((unsigned)((npix + 63) / 64), (unsigned)((C + 63) / 64));
  hipLaunchKernelGGL((k_nchw_vcl<0>), grid, dim3(256), 0, lfsr_stream(stream), in, out, in_stride, in_choff, B, C, A, h, w, layout);
  LFSR_CHECK_LAUNCH();
  return LFSR_OK;
}

}  // extern "C"
