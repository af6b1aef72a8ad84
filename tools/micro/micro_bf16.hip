// Micro-benchmarks behind the split-bf16 conv design (round 2):
//  (1) co-execution: wave A streams MFMAs (fp32 16x16x4 | bf16 16x16x32 | bf16 32x32x16), wave B of the same SIMD streams VALU; each alone and together
//  (2) U stream: 4 waves per CU stream 24 B per lane per step (dwordx4 + dwordx2) of an L2-resident 885 KB table, ring of 12, with 3 bf16 MFMAs per step
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

template <int MODE>   // 0 fp32 16x16x4, 1 bf16 16x16x32, 2 bf16 32x32x16
__device__ __forceinline__ void mfma_loop(int iters, float seed, float* sink) {
  if (MODE == 0) {
    f32x4 acc[8]; for (int i = 0; i < 8; ++i) acc[i] = f32x4{0, 0, 0, 0};
    float a = seed, b = seed * 0.5f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
    }
    float s = 0; for (int i = 0; i < 8; ++i) s += acc[i][0]; *sink = s;
  } else if (MODE == 1) {
    f32x4 acc[8]; for (int i = 0; i < 8; ++i) acc[i] = f32x4{0, 0, 0, 0};
    bf16x8 a, b; for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(seed + i); b[i] = (__bf16)(seed * 0.5f + i); }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[i], 0, 0, 0);
    }
    float s = 0; for (int i = 0; i < 8; ++i) s += acc[i][0]; *sink = s;
  } else if (MODE == 3) {
    typedef short s16x4 __attribute__((ext_vector_type(4)));
    f32x4 acc[8]; for (int i = 0; i < 8; ++i) acc[i] = f32x4{0, 0, 0, 0};
    s16x4 a, b; for (int i = 0; i < 4; ++i) { a[i] = (short)(0x3f80 + i); b[i] = (short)(0x3f00 + i); }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a, b, acc[i], 0, 0, 0);
    }
    float s = 0; for (int i = 0; i < 8; ++i) s += acc[i][0]; *sink = s;
  } else {
    f32x16 acc[4]; for (int i = 0; i < 4; ++i) for (int k = 0; k < 16; ++k) acc[i][k] = 0;
    bf16x8 a, b; for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(seed + i); b[i] = (__bf16)(seed * 0.5f + i); }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i], 0, 0, 0);
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i], 0, 0, 0);
    }
    float s = 0; for (int i = 0; i < 4; ++i) s += acc[i][0]; *sink = s;
  }
}

template <int VMODE>   // 0: v_fma_f32 x16 ; 1: the split sequence (cvt_pk_bf16, shifts/ands, subs) ; 2: v_pk_fma_f32 x8
__device__ __forceinline__ void valu_loop(int iters, float seed, float* sink) {
  float v[16];
  for (int i = 0; i < 16; ++i) v[i] = seed + i;
  for (int it = 0; it < iters; ++it) {
    if (VMODE == 0) {
#pragma unroll
      for (int i = 0; i < 16; ++i) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(v[i]) : "v"(seed));
    } else if (VMODE == 2) {
      typedef float f32x2 __attribute__((ext_vector_type(2)));
#pragma unroll
      for (int i = 0; i < 8; ++i) { f32x2 t = {v[2 * i], v[2 * i + 1]}; asm volatile("v_pk_fma_f32 %0, %0, %1, %0" : "+v"(t) : "v"(t)); v[2 * i] = t.x; v[2 * i + 1] = t.y; }
    } else {
#pragma unroll
      for (int i = 0; i < 16; i += 2) {   // 8 VALU per pair: cvt_pk, lshl, and, sub, sub, cvt_pk (second level), xor-combine
        unsigned pk; asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(pk) : "v"(v[i]), "v"(v[i + 1]));
        float lo, hi; asm volatile("v_lshlrev_b32 %0, 16, %1" : "=v"(lo) : "v"(pk)); asm volatile("v_and_b32 %0, 0xffff0000, %1" : "=v"(hi) : "v"(pk));
        float r0, r1; asm volatile("v_sub_f32 %0, %1, %2" : "=v"(r0) : "v"(v[i]), "v"(lo)); asm volatile("v_sub_f32 %0, %1, %2" : "=v"(r1) : "v"(v[i + 1]), "v"(hi));
        unsigned pk2; asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(pk2) : "v"(r0), "v"(r1));
        asm volatile("v_xor_b32 %0, %1, %2" : "=v"(v[i]) : "v"(pk), "v"(pk2));
        asm volatile("v_add_f32 %0, %1, %2" : "=v"(v[i + 1]) : "v"(r0), "v"(r1));
      }
    }
  }
  float s = 0; for (int i = 0; i < 16; ++i) s += v[i]; *sink = s;
}

// role bits: 1 = waves 0..3 run the MFMA loop, 2 = waves 4..7 run the VALU loop
template <int MODE, int VMODE>
__global__ __launch_bounds__(512) void k_coexec(int roles, int mi, int vi, float seed, long long* out, float* sink) {
  const int wave = threadIdx.x >> 6;
  __syncthreads();
  const long long t0 = clock64();
  if (wave < 4) { if (roles & 1) mfma_loop<MODE>(mi, seed, sink + blockIdx.x * 512 + threadIdx.x); }
  else { if (roles & 2) valu_loop<VMODE>(vi, seed, sink + blockIdx.x * 512 + threadIdx.x); }
  const long long t1 = clock64();
  if ((threadIdx.x & 63) == 0) out[blockIdx.x * 8 + wave] = t1 - t0;
}

// U stream: wave ns streams [step][ns][lane][24 B] ; 144 steps per pass
template <int WITH_MFMA, int RING>
__global__ __launch_bounds__(256) void k_ustream(const unsigned* __restrict__ U, int passes, long long* out, float* sink) {
  const int lane = threadIdx.x & 63, ns = threadIdx.x >> 6;
  __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned*>(U), 0, 144 * 4 * 64 * 24, 0x00020000);
  const int voff = ns * 64 * 24 + lane * 24;
  u32x4 a[RING]; u32x2 b[RING];
  f32x4 acc[8]; for (int i = 0; i < 8; ++i) acc[i] = f32x4{0, 0, 0, 0};
  unsigned x = 0;
#pragma unroll
  for (int i = 0; i < RING; ++i) { a[i] = __builtin_amdgcn_raw_buffer_load_b128(rs, voff, i * 6144, 0); b[i] = __builtin_bit_cast(u32x2, __builtin_amdgcn_raw_buffer_load_b64(rs, voff + 16, i * 6144, 0)); }
  __syncthreads();
  const long long t0 = clock64();
  for (int ps = 0; ps < passes; ++ps) {
#pragma unroll 1
    for (int s0 = 0; s0 < 144; s0 += RING) {
#pragma unroll
      for (int i = 0; i < RING; ++i) {
        const int sn = (s0 + i + RING) % 144;
        if (WITH_MFMA) {
          u32x4 w01 = {a[i].z, a[i].w, b[i].x, b[i].y};
          bf16x8 A0 = __builtin_bit_cast(bf16x8, a[i]), A1 = __builtin_bit_cast(bf16x8, w01);
          acc[(i * 3) & 7] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A0, A1, acc[(i * 3) & 7], 0, 0, 0);
          acc[(i * 3 + 1) & 7] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A1, A0, acc[(i * 3 + 1) & 7], 0, 0, 0);
          acc[(i * 3 + 2) & 7] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A1, A1, acc[(i * 3 + 2) & 7], 0, 0, 0);
        } else {
          x ^= a[i].x ^ a[i].y ^ a[i].z ^ a[i].w ^ b[i].x ^ b[i].y;
        }
        a[i] = __builtin_amdgcn_raw_buffer_load_b128(rs, voff, sn * 6144, 0);
        b[i] = __builtin_bit_cast(u32x2, __builtin_amdgcn_raw_buffer_load_b64(rs, voff + 16, sn * 6144, 0));
      }
    }
  }
  const long long t1 = clock64();
  float s = (float)x; for (int i = 0; i < 8; ++i) s += acc[i][0];
  for (int i = 0; i < RING; ++i) s += (float)(a[i].x ^ b[i].x);
  sink[blockIdx.x * 256 + threadIdx.x] = s;
  if (lane == 0) out[blockIdx.x * 4 + ns] = t1 - t0;
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

template <int MODE, int VMODE>
void run_coexec(const char* name, int nblk, long long* dout, float* dsink) {
  const int mi = MODE == 0 ? 2000 : (MODE == 1 || MODE == 3) ? 4000 : 2000, vi = 4000;
  const int per_m = 8, per_v = VMODE == 2 ? 8 : VMODE == 1 ? 64 : 16;
  for (int roles = 1; roles <= 3; ++roles) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL((k_coexec<MODE, VMODE>), dim3(nblk), dim3(512), 0, 0, roles, mi, vi, 1.0f, dout, dsink);
    CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<long long> h(nblk * 8); CK(hipMemcpy(h.data(), dout, h.size() * 8, hipMemcpyDeviceToHost));
    double cm = 0, cv = 0; for (int b = 0; b < nblk; ++b) { for (int w = 0; w < 4; ++w) cm += h[b * 8 + w]; for (int w = 4; w < 8; ++w) cv += h[b * 8 + w]; }
    cm /= nblk * 4; cv /= nblk * 4;
    printf("%-34s blocks %3d roles %d : mfma wave %9.0f cyc (%.1f / mfma)   valu wave %9.0f cyc (%.2f / valu)   wall %.3f ms  clock~%.2f GHz\n", name, nblk, roles,
           cm, (roles & 1) ? cm / (mi * per_m) : 0.0, cv, (roles & 2) ? cv / ((double)vi * per_v) : 0.0, ms, (cm > cv ? cm : cv) / (ms * 1e6));
  }
}

int main() {
  long long* dout; float* dsink; unsigned* dU;
  CK(hipMalloc(&dout, 256 * 8 * 8)); CK(hipMalloc(&dsink, 256 * 512 * 4)); CK(hipMalloc(&dU, 144 * 4 * 64 * 24));
  std::vector<unsigned> hu(144 * 4 * 64 * 6); for (size_t i = 0; i < hu.size(); ++i) hu[i] = 0x3f803f80u + (unsigned)(i * 2654435761u >> 20);
  CK(hipMemcpy(dU, hu.data(), hu.size() * 4, hipMemcpyHostToDevice));
  run_coexec<3, 0>("bf16 16x16x16 | v_fma_f32", 1, dout, dsink);
  run_coexec<3, 1>("bf16 16x16x16 | split seq", 256, dout, dsink);
  if (getenv("MICRO_SHORT")) return 0;
  for (int nblk : {1, 256}) {
    run_coexec<0, 0>("fp32 16x16x4  | v_fma_f32", nblk, dout, dsink);
    run_coexec<1, 0>("bf16 16x16x32 | v_fma_f32", nblk, dout, dsink);
    run_coexec<2, 0>("bf16 32x32x16 | v_fma_f32", nblk, dout, dsink);
    run_coexec<1, 1>("bf16 16x16x32 | split seq", nblk, dout, dsink);
    run_coexec<2, 1>("bf16 32x32x16 | split seq", nblk, dout, dsink);
    run_coexec<1, 2>("bf16 16x16x32 | v_pk_fma_f32", nblk, dout, dsink);
    run_coexec<0, 2>("fp32 16x16x4  | v_pk_fma_f32", nblk, dout, dsink);
  }
  for (int nblk : {1, 256}) {
    for (int variant = 0; variant < 4; ++variant) {
      const int passes = 20;
      hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
      CK(hipEventRecord(e0));
      if (variant == 0) hipLaunchKernelGGL((k_ustream<0, 12>), dim3(nblk), dim3(256), 0, 0, dU, passes, dout, dsink);
      if (variant == 1) hipLaunchKernelGGL((k_ustream<1, 12>), dim3(nblk), dim3(256), 0, 0, dU, passes, dout, dsink);
      if (variant == 2) hipLaunchKernelGGL((k_ustream<0, 24>), dim3(nblk), dim3(256), 0, 0, dU, passes, dout, dsink);
      if (variant == 3) hipLaunchKernelGGL((k_ustream<1, 24>), dim3(nblk), dim3(256), 0, 0, dU, passes, dout, dsink);
      CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      std::vector<long long> h(nblk * 4); CK(hipMemcpy(h.data(), dout, h.size() * 8, hipMemcpyDeviceToHost));
      double c = 0; for (auto v : h) c += v; c /= h.size();
      printf("ustream mfma %d ring %2d blocks %3d : %8.0f cycles per pass of 884736 B per CU = %.1f B/clk/CU   wall %.3f ms  clock~%.2f GHz\n", variant & 1, variant < 2 ? 12 : 24, nblk,
             c / passes, 884736.0 * passes / c, ms, c / (ms * 1e6));
    }
  }
  return 0;
}
