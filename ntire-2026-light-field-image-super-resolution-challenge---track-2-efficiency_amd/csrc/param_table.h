// Packed-parameter table keyed by the reference's state_dict names (host side, shared by the EPIT / LFT drivers).
#pragma once
#include <map>
#include <string>
#include <vector>

#include "lfsr_internal.h"

struct LfsrParamTable {
  struct Slot {
    size_t off = 0, floats = 0, numel = 0;
    int O = 0, C = 0, T = 0, perm = 0, ch = 0;
    bool raw = false, loaded = false;
  };
  std::map<std::string, Slot> slots;
  size_t packed_floats = 0;
  float* packed = nullptr;

  static size_t align64(size_t f) { return (f + 63) / 64 * 64; }
  void add(const std::string& k, int O, int C, int T, int perm = 0, int ch = 0, bool raw = false) {
    Slot s;
    s.O = O; s.C = C; s.T = T; s.perm = perm; s.ch = ch; s.raw = raw;
    s.numel = (size_t)O * C * T;
    s.floats = raw ? s.numel : lfsr_packed_weight_floats(O, C, T);
    s.off = packed_floats;
    packed_floats += align64(s.floats);
    slots[k] = s;
  }
  size_t reserve(size_t floats) { size_t o = packed_floats; packed_floats += align64(floats); return o; }
  const float* w(const std::string& k) const { return packed + slots.at(k).off; }
  int set_packed(void* p, size_t bytes) {
    if (!p || bytes < packed_floats * sizeof(float) || ((uintptr_t)p & 15)) return LFSR_E_ARG;
    packed = (float*)p;
    for (auto& kv : slots) kv.second.loaded = false;
    return LFSR_OK;
  }
  int load(const char* key, const float* data, size_t numel, void* stream) {
    if (!key || !data || !packed) return LFSR_E_ARG;
    auto it = slots.find(key);
    if (it == slots.end() || numel != it->second.numel) return LFSR_E_ARG;
    Slot& s = it->second;
    if (s.raw) {
      hipError_t e = hipMemcpyAsync(packed + s.off, data, numel * sizeof(float), hipMemcpyDeviceToDevice, lfsr_stream(stream));
      if (e != hipSuccess) return LFSR_HIP_ERR(e);
    } else {
      int rc = lfsr_pack_conv_weight(data, packed + s.off, s.O, s.C, s.T, s.perm, s.ch, stream);
      if (rc) return rc;
    }
    s.loaded = true;
    return LFSR_OK;
  }
  bool all_loaded() const {
    for (auto& kv : slots)
      if (!kv.second.loaded) return false;
    return true;
  }
};
