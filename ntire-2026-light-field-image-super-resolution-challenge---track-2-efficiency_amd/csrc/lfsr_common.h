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
static inline unsigned lfsr_blocks(long long n, int per) {
  long long b = (n + per - 1) / per;
  return (unsigned)(b < 1 ? 1 : b);
}
