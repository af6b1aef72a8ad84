// Does the order in which a kernel walks a tensor decide how much of it the NEXT kernel finds in the memory-side cache (256 MiB)?
// W writes T (210 MB) front to back; R reads T and writes Z (210 MB), either front to back (the lines written LAST are the ones R reads last: evicted by then)
// or back to front (the lines written last are read first).  hipEvent times of R, 256 x 8 persistent-style blocks, 16 B per lane.
// build: hipcc -O3 --offload-arch=gfx950 tools/micro/mall_order.hip -o _diag/mall_order ; run: _diag/mall_order
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <algorithm>

__global__ __launch_bounds__(256) void k_write(float4* __restrict__ t, long long n4, int reverse, float v) {
  const long long per = 256LL * 16;                          // float4 per chunk (64 KB)
  const long long nchunks = n4 / per;
  for (long long c = blockIdx.x; c < nchunks; c += gridDim.x) {
    const long long cc = reverse ? nchunks - 1 - c : c;
    float4* p = t + cc * per + threadIdx.x;
#pragma unroll
    for (int i = 0; i < 16; ++i) p[i * 256] = make_float4(v, v + 1.f, v + 2.f, v + 3.f);
  }
}

__global__ __launch_bounds__(256) void k_read_write(const float4* __restrict__ t, float4* __restrict__ z, long long n4, int reverse, int write_z) {
  const long long per = 256LL * 16;
  const long long nchunks = n4 / per;
  for (long long c = blockIdx.x; c < nchunks; c += gridDim.x) {
    const long long cc = reverse ? nchunks - 1 - c : c;
    const float4* p = t + cc * per + threadIdx.x;
    float4 a[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) a[i] = p[i * 256];
    if (write_z) {
      float4* q = z + cc * per + threadIdx.x;
#pragma unroll
      for (int i = 0; i < 16; ++i) q[i * 256] = make_float4(a[i].x + 1.f, a[i].y, a[i].z, a[i].w);
    } else {
      float s = 0.f;
#pragma unroll
      for (int i = 0; i < 16; ++i) s += a[i].x;
      if (s == 12345.678f) z[0] = a[0];
    }
  }
}

int main() {
  const long long n4 = 210LL * 1000 * 1000 / 16 / 4096 * 4096;   // ~210 MB
  float4 *t, *z, *x;
  hipMalloc(&t, n4 * 16); hipMalloc(&z, n4 * 16); hipMalloc(&x, n4 * 16);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int grid = 256 * 8;
  for (int write_z = 0; write_z < 2; ++write_z)
    for (int extra = 0; extra < 2; ++extra)        // extra: W also reads another 210-MB tensor x first (like a conv reading its input while it writes)
      for (int rev = 0; rev < 2; ++rev) {
        std::vector<float> ms;
        for (int it = 0; it < 12; ++it) {
          if (extra) hipLaunchKernelGGL(k_read_write, dim3(grid), dim3(256), 0, 0, x, t, n4, 0, 1);    // t written while x is read (interleaved traffic)
          else hipLaunchKernelGGL(k_write, dim3(grid), dim3(256), 0, 0, t, n4, 0, (float)it);
          hipEventRecord(e0, 0);
          hipLaunchKernelGGL(k_read_write, dim3(grid), dim3(256), 0, 0, t, z, n4, rev, write_z);
          hipEventRecord(e1, 0);
          hipEventSynchronize(e1);
          float m; hipEventElapsedTime(&m, e0, e1);
          if (it >= 2) ms.push_back(m);
        }
        std::sort(ms.begin(), ms.end());
        const double bytes = (double)n4 * 16 * (write_z ? 2 : 1);
        printf("producer %s, consumer reads %s%s: median %.1f us  (%.2f TB/s)\n", extra ? "reads x + writes T" : "writes T", rev ? "BACK TO FRONT" : "front to back",
               write_z ? " and writes Z" : "", ms[ms.size() / 2] * 1e3, bytes / (ms[ms.size() / 2] * 1e-3) / 1e12);
      }
  return 0;
}
