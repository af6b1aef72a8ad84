// Shared host-side helpers for liblfsr_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <atomic>
#include "../../include/lfsr_hip.h"

#define LFSR_HIP_ERR(e) (-(1000 + (int)(e)))
#define LFSR_CHECK_LAUNCH()                                   \
  do {                                                        \
    hipError_t e__ = hipGetLastError();                       \
    if (e__ != hipSuccess) return LFSR_HIP_ERR(e__);          \
  } while (0)

static inline hipStream_t lfsr_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }
// A/B ("lab") selectors: LFSR_* environment variables that pick an alternative kernel form of an operator for measurements and parity tests.  They are consulted
// ONLY when the process runs with LFSR_LAB set (read once, at the first selector lookup): the product default is one tested path per operator and no getenv on the
// launch path.  The one selection a product caller may make -- the arithmetic of the GEMMs that have an exact-three-term-bf16 form -- is an API: lfsr_set_arithmetic.
const char* lfsr_sel(const char* name);
bool lfsr_arith_f32();     // lfsr_set_arithmetic(LFSR_ARITH_F32): every GEMM on fp32 MFMA
static inline unsigned lfsr_blocks(long long n, int per) {
  long long b = (n + per - 1) / per;
  return (unsigned)(b < 1 ? 1 : b);
}
// Non-temporal hint on result stores.  Measured on one MI355X (two runs each per call): on the 3x3 conv's output stores alone (buffer_store ... nt,
// W4_STPOL in conv3x3_wino4.hip) the DistgSSR forward goes 1678.7 -> 1718 patches/s (profiles/r02_logs/ab_bench_lines.json: bench13_*.json); on the EPI / angular / row-GEMM
// kernels' stores as well (global_store ... nt through lfsr_store_stream) it FALLS to 1444 (EPIT 668 -> 539, LFT 1331 -> 982 patches/s,
// profiles/r02_logs/ab_bench_lines.json: bench14_*.json): those results are re-read by the next kernel while still in the last-level cache.  So only the conv kernel uses the
// hint; lfsr_store_stream stays for A/B builds.  LFSR_NT_STORES=0 at compile time restores plain stores everywhere.
#ifndef LFSR_NT_STORES
#define LFSR_NT_STORES 1
#endif
#if defined(__HIPCC__)
typedef float lfsr_v4f __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void lfsr_store_stream(float4* p, const float4 v) {
#if LFSR_NT_STORES
  lfsr_v4f t = {v.x, v.y, v.z, v.w};
  __builtin_nontemporal_store(t, reinterpret_cast<lfsr_v4f*>(p));
#else
  *p = v;
#endif
}
#endif


