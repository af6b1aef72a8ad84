// which (lane, element) of B pairs with a given (lane, element) of A, and where D lands: v_mfma_f32_16x16x32_bf16 and v_mfma_f32_16x16x16_bf16
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
__global__ void k(float* out) {   // block = one wave; blockIdx.x = (K32: la * 8 + ea for la in {0,1,16,17,33}) ...
  const int lane = threadIdx.x, cfg = blockIdx.x, k32 = cfg < 40;
  const int c = k32 ? cfg : cfg - 40;
  const int las[5] = {0, 1, 16, 17, 33};
  const int la = las[c / 8], ea = c % 8;
  f32x4 acc = {0, 0, 0, 0};
  if (k32) {
    bf16x8 a, b;
    for (int e = 0; e < 8; ++e) { a[e] = (__bf16)((lane == la && e == ea) ? 1.0f : 0.0f); b[e] = (__bf16)((lane % 16 == 3) ? (float)((lane / 16) * 8 + e + 1) : 0.0f); }
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc, 0, 0, 0);
  } else if (ea < 4) {
    s16x4 a, b;
    for (int e = 0; e < 4; ++e) {
      float av = (lane == la && e == ea) ? 1.0f : 0.0f, bv = (lane % 16 == 3) ? (float)((lane / 16) * 4 + e + 1) : 0.0f;
      a[e] = (short)(__builtin_bit_cast(unsigned, av) >> 16); b[e] = (short)(__builtin_bit_cast(unsigned, bv) >> 16);
    }
    acc = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a, b, acc, 0, 0, 0);
  }
  for (int r = 0; r < 4; ++r) out[(cfg * 64 + lane) * 4 + r] = acc[r];
}
int main() {
  float* d; hipMalloc(&d, 80 * 64 * 4 * 4);
  hipLaunchKernelGGL(k, dim3(80), dim3(64), 0, 0, d); hipDeviceSynchronize();
  static float h[80 * 64 * 4]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  const int las[5] = {0, 1, 16, 17, 33};
  for (int cfg = 0; cfg < 80; ++cfg) {
    const int c = cfg % 40;
    if (cfg >= 40 && c % 8 >= 4) continue;
    printf("%s A(lane %2d, elem %d):", cfg < 40 ? "K32" : "K16", las[c / 8], c % 8);
    for (int l = 0; l < 64; ++l) for (int r = 0; r < 4; ++r) if (h[(cfg * 64 + l) * 4 + r] != 0.0f) printf("  D(lane %d, reg %d) = %g", l, r, h[(cfg * 64 + l) * 4 + r]);
    printf("\n");
  }
  return 0;
}
