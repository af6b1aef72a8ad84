// Library-wide options of liblfsr_hip.so.
//  * lfsr_set_arithmetic: which arithmetic the GEMMs that have two forms run in (the product-level choice);
//  * lfsr_sel: the A/B selectors of the measurement / parity tooling (LFSR_* environment variables), live only in a process started with LFSR_LAB set.
#include <stdlib.h>

#include "lfsr_internal.h"

namespace {
std::atomic<int> g_arith{LFSR_ARITH_DEFAULT};
std::atomic<int> g_lab{-1};
}  // namespace

const char* lfsr_sel(const char* name) {
  int lab = g_lab.load(std::memory_order_relaxed);
  if (lab < 0) {
    lab = getenv("LFSR_LAB") != nullptr ? 1 : 0;
    g_lab.store(lab, std::memory_order_relaxed);
  }
  return lab ? getenv(name) : nullptr;
}

bool lfsr_arith_f32() { return g_arith.load(std::memory_order_relaxed) == LFSR_ARITH_F32; }

extern "C" {

int lfsr_set_arithmetic(int mode) {
  if (mode != LFSR_ARITH_DEFAULT && mode != LFSR_ARITH_F32) return LFSR_E_ARG;
  g_arith.store(mode);
  return LFSR_OK;
}

int lfsr_get_arithmetic(void) { return g_arith.load(); }

}  // extern "C"
