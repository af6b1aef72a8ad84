// Host driver for the DistgSSR forward (get_model.forward, model/SR/DistgSSR.py:29-36) on VCL buffers.
// Pure host code above the C-ABI operator entry points: owns the packed-weight table keyed by the
// reference's state_dict names and the launch sequence; allocates nothing on the device.
#include <map>
#include <string>
#include <vector>

#include "lfsr_common.h"

namespace {
struct Slot {
  size_t off = 0;      // float offset in the packed buffer
  size_t floats = 0;   // packed size
  size_t numel = 0;    // expected raw element count
  int O = 0, C = 0, T = 0, perm = 0, ch = 0;
  bool raw = false, loaded = false;
};
inline size_t align64(size_t f) { return (f + 63) / 64 * 64; }  // 256-B granules
}  // namespace

struct lfsr_distgssr {
  int A, s, G, NB, C;
  std::map<std::string, Slot> slots;
  size_t packed_floats = 0, off_wf = 0, off_bf = 0;
  float* packed = nullptr;
  bool finalized = false;
  bool profiling = false;
  struct Ev { int cls; hipEvent_t a, b; };
  std::vector<Ev> evs;          // recorded this profiling session
  std::vector<hipEvent_t> free_evs;
  hipEvent_t get_ev() {
    if (!free_evs.empty()) { hipEvent_t e = free_evs.back(); free_evs.pop_back(); return e; }
    hipEvent_t e = nullptr;
    (void)hipEventCreate(&e);
    return e;
  }

  void add(const std::string& k, int O, int Cc, int T, int perm, int ch, bool raw) {
    Slot sl;
    sl.O = O; sl.C = Cc; sl.T = T; sl.perm = perm; sl.ch = ch; sl.raw = raw;
    sl.numel = (size_t)O * Cc * T;
    sl.floats = raw ? sl.numel : lfsr_packed_weight_floats(O, Cc, T);
    sl.off = packed_floats;
    packed_floats += align64(sl.floats);
    slots[k] = sl;
  }
  const float* w(const std::string& k) const { return packed + slots.at(k).off; }
};

extern "C" {

int lfsr_distgssr_create(lfsr_distgssr** out, int A, int scale, int n_group, int n_block, int channels) {
  if (!out || A <= 0 || A > 15 || (scale != 2 && scale != 3 && scale != 4) || n_group <= 0 || n_block <= 0 || channels != 64) return LFSR_E_ARG;
  lfsr_distgssr* c = new lfsr_distgssr();
  c->A = A; c->s = scale; c->G = n_group; c->NB = n_block; c->C = channels;
  const int AA = A * A;
  c->add("init_conv.weight", 64, 1, 9, 0, 0, true);
  for (int g = 0; g < n_group; ++g) {
    for (int b = 0; b < n_block; ++b) {
      std::string p = "disentg.Group." + std::to_string(g) + ".Block." + std::to_string(b) + ".";
      c->add(p + "SpaConv.0.weight", 64, 64, 9, 0, 0, false);
      c->add(p + "SpaConv.2.weight", 64, 64, 9, 0, 0, false);
      c->add(p + "AngConv.0.weight", 16, 64, AA, 0, 0, false);
      c->add(p + "AngConv.2.weight", 16 * AA, 16, 1, 1, 16, false);
      c->add(p + "EPIConv.0.weight", 32, 64, AA, 0, 0, false);
      c->add(p + "EPIConv.2.weight", 32 * A, 32, 1, 0, 0, false);
      c->add(p + "fuse.0.weight", 64, 144, 1, 0, 0, false);
      c->add(p + "fuse.2.weight", 64, 64, 9, 0, 0, false);
    }
    c->add("disentg.Group." + std::to_string(g) + ".conv.weight", 64, 64, 9, 0, 0, false);
  }
  c->add("disentg.conv.weight", 64, 64, 9, 0, 0, false);
  c->add("upsample.0.weight", 64 * scale * scale, 64, 1, 0, 0, true);
  c->add("upsample.0.bias", 64 * scale * scale, 1, 1, 0, 0, true);
  c->add("upsample.2.weight", 1, 64, 1, 0, 0, true);
  c->off_wf = c->packed_floats; c->packed_floats += align64((size_t)scale * scale * 64);
  c->off_bf = c->packed_floats; c->packed_floats += align64((size_t)scale * scale);
  *out = c;
  return LFSR_OK;
}

void lfsr_distgssr_destroy(lfsr_distgssr* c) {
  if (!c) return;
  for (auto& e : c->evs) { (void)hipEventDestroy(e.a); (void)hipEventDestroy(e.b); }
  for (auto e : c->free_evs) (void)hipEventDestroy(e);
  delete c;
}

int lfsr_distgssr_profile(lfsr_distgssr* c, int enable) {
  if (!c) return LFSR_E_ARG;
  for (auto& e : c->evs) { c->free_evs.push_back(e.a); c->free_evs.push_back(e.b); }
  c->evs.clear();
  c->profiling = enable != 0;
  return LFSR_OK;
}

int lfsr_distgssr_profile_read(lfsr_distgssr* c, double* ms, long long* launches) {
  if (!c || !ms || !launches) return LFSR_E_ARG;
  for (int i = 0; i < LFSR_DISTG_NCLASS; ++i) { ms[i] = 0.0; launches[i] = 0; }
  for (auto& e : c->evs) {
    hipError_t err = hipEventSynchronize(e.b);
    if (err != hipSuccess) return LFSR_HIP_ERR(err);
    float t = 0.f;
    err = hipEventElapsedTime(&t, e.a, e.b);
    if (err != hipSuccess) return LFSR_HIP_ERR(err);
    ms[e.cls] += t;
    launches[e.cls] += 1;
    c->free_evs.push_back(e.a); c->free_evs.push_back(e.b);
  }
  c->evs.clear();
  return LFSR_OK;
}

size_t lfsr_distgssr_packed_bytes(const lfsr_distgssr* c) { return c ? c->packed_floats * sizeof(float) : 0; }

int lfsr_distgssr_set_packed(lfsr_distgssr* c, void* packed, size_t bytes) {
  if (!c || !packed || bytes < c->packed_floats * sizeof(float) || ((uintptr_t)packed & 15)) return LFSR_E_ARG;
  c->packed = (float*)packed;
  c->finalized = false;
  for (auto& kv : c->slots) kv.second.loaded = false;
  return LFSR_OK;
}

int lfsr_distgssr_load_param(lfsr_distgssr* c, const char* key, const float* data, size_t numel, void* stream) {
  if (!c || !key || !data || !c->packed) return LFSR_E_ARG;
  auto it = c->slots.find(key);
  if (it == c->slots.end()) return LFSR_E_ARG;
  Slot& sl = it->second;
  if (numel != sl.numel) return LFSR_E_ARG;
  c->finalized = false;
  if (sl.raw) {
    hipError_t e = hipMemcpyAsync(c->packed + sl.off, data, numel * sizeof(float), hipMemcpyDeviceToDevice, lfsr_stream(stream));
    if (e != hipSuccess) return LFSR_HIP_ERR(e);
  } else {
    int rc = lfsr_pack_conv_weight(data, c->packed + sl.off, sl.O, sl.C, sl.T, sl.perm, sl.ch, stream);
    if (rc) return rc;
  }
  sl.loaded = true;
  return LFSR_OK;
}

int lfsr_distgssr_finalize(lfsr_distgssr* c, void* stream) {
  if (!c || !c->packed) return LFSR_E_ARG;
  for (auto& kv : c->slots)
    if (!kv.second.loaded) return LFSR_E_ARG;
  int rc = lfsr_fold_head(c->w("upsample.0.weight"), c->w("upsample.0.bias"), c->w("upsample.2.weight"),
                          c->packed + c->off_wf, c->packed + c->off_bf, 64, c->s, stream);
  if (rc) return rc;
  c->finalized = true;
  return LFSR_OK;
}

static void ws_layout(const lfsr_distgssr* c, int B, int h, int w, size_t off[8], size_t* total) {
  const size_t npix = (size_t)B * c->A * c->A * h * w;
  size_t o = 0;
  for (int i = 0; i < 5; ++i) { off[i] = o; o += align64(npix * 64); }   // pool[0..3], T
  off[5] = o; o += align64(npix * 144);                                   // CAT
  off[6] = o; o += align64((size_t)B * h * w * 16);                       // ang stage-1
  off[7] = o; o += align64((size_t)B * c->A * h * w * 32);                // epi stage-1
  *total = o;
}

size_t lfsr_distgssr_workspace_bytes(const lfsr_distgssr* c, int B, int h, int w) {
  if (!c || B <= 0 || h <= 0 || w <= 0) return 0;
  size_t off[8], tot;
  ws_layout(c, B, h, w, off, &tot);
  return tot * sizeof(float);
}

int lfsr_distgssr_forward_taps(lfsr_distgssr* c, const float* x, float* out, int B, int h, int w, void* workspace,
                               size_t workspace_bytes, float* const* taps, void* stream) {
  if (!c || !x || !out || !workspace || B <= 0 || h <= 0 || w <= 0 || !c->finalized) return LFSR_E_ARG;
  if ((uintptr_t)workspace & 15) return LFSR_E_ARG;
  size_t off[8], tot;
  ws_layout(c, B, h, w, off, &tot);
  if (workspace_bytes < tot * sizeof(float)) return LFSR_E_WS;
  if ((long long)B * c->A * c->A * h * w >= (1LL << 31) / 144) return LFSR_E_ARG;  // pixel*stride must fit the kernels' index math
  float* ws = (float*)workspace;
  float* pool[4] = {ws + off[0], ws + off[1], ws + off[2], ws + off[3]};
  float* T = ws + off[4];
  float* CAT = ws + off[5];
  float* A16 = ws + off[6];
  float* E32 = ws + off[7];
  const int A = c->A, AA = A * A, nimg = B * AA;
  const float L = 0.1f;  // LeakyReLU(0.1), DistgSSR.py:80-101
  int rc;
#define RC(call) do { rc = (call); if (rc) return rc; } while (0)
#define PROF(cls, call)                                                              \
  do {                                                                               \
    if (c->profiling) {                                                              \
      lfsr_distgssr::Ev ev{cls, c->get_ev(), c->get_ev()};                           \
      (void)hipEventRecord(ev.a, lfsr_stream(stream));                                     \
      rc = (call);                                                                   \
      (void)hipEventRecord(ev.b, lfsr_stream(stream));                                     \
      c->evs.push_back(ev);                                                          \
    } else {                                                                         \
      rc = (call);                                                                   \
    }                                                                                \
    if (rc) return rc;                                                               \
  } while (0)
  auto tap = [&](int which, const float* buf, int stride, int C) -> int {
    if (!taps || !taps[which]) return LFSR_OK;
    return lfsr_vcl_to_nchw(buf, stride, 0, taps[which], B, C, A, h, w, 1, stream);
  };
  auto pick = [&](const float* a, const float* b2, const float* c2) -> float* {
    for (int i = 0; i < 4; ++i)
      if (pool[i] != a && pool[i] != b2 && pool[i] != c2) return pool[i];
    return nullptr;
  };
  auto conv = [&](const float* in, const std::string& key, float* o, int ostride, int ochoff, const float* res, float slope) -> int {
    return lfsr_conv3x3_fwd(in, 64, 0, c->w(key), o, ostride, ochoff, res, 64, 0, nullptr, 0, 0, nimg, h, w, slope, stream);
  };

  float* F0 = pool[0];
  PROF(4, lfsr_initconv_fwd(x, c->w("init_conv.weight"), F0, 64, 0, B, A, h, w, stream));
  RC(tap(0, F0, 64, 64));
  const float* cur = F0;
  for (int g = 0; g < c->G; ++g) {
    const float* gin = cur;
    for (int b = 0; b < c->NB; ++b) {
      std::string p = "disentg.Group." + std::to_string(g) + ".Block." + std::to_string(b) + ".";
      float* o = pick(F0, gin, cur);
      PROF(0, conv(cur, p + "SpaConv.0.weight", T, 64, 0, nullptr, L));
      PROF(0, conv(T, p + "SpaConv.2.weight", CAT, 144, 0, nullptr, L));
      PROF(1, lfsr_angconv_fwd(cur, 64, 0, c->w(p + "AngConv.0.weight"), c->w(p + "AngConv.2.weight"), A16, CAT, 144, 64, B, A, h, w, L, stream));
      PROF(2, lfsr_epiconv_hv_fwd(cur, 64, 0, c->w(p + "EPIConv.0.weight"), c->w(p + "EPIConv.2.weight"), E32, CAT, 144, 80, 112, B, A, h, w, L, stream));
      if (g == 0 && b == 0) RC(tap(4, CAT, 144, 144));
      PROF(3, lfsr_pointwise_fwd(CAT, 144, 0, 144, c->w(p + "fuse.0.weight"), nullptr, T, 64, 0, nimg * h * w, 64, L, stream));
      PROF(0, conv(T, p + "fuse.2.weight", o, 64, 0, cur, 1.0f));
      cur = o;
      if (g == 0 && b == 0) RC(tap(1, cur, 64, 64));
    }
    float* o = pick(F0, gin, cur);
    PROF(0, conv(cur, "disentg.Group." + std::to_string(g) + ".conv.weight", o, 64, 0, gin, 1.0f));
    cur = o;
    if (g == 0) RC(tap(2, cur, 64, 64));
  }
  {
    float* o = pick(F0, cur, cur);
    PROF(0, conv(cur, "disentg.conv.weight", o, 64, 0, F0, 1.0f));
    cur = o;
    RC(tap(3, cur, 64, 64));
  }
  PROF(5, lfsr_upsample_head_fwd(cur, 64, 0, c->packed + c->off_wf, c->packed + c->off_bf, x, out, B, A, h, w, c->s, stream));
#undef RC
#undef PROF
  return LFSR_OK;
}

int lfsr_distgssr_forward(lfsr_distgssr* c, const float* x, float* out, int B, int h, int w, void* workspace,
                          size_t workspace_bytes, void* stream) {
  return lfsr_distgssr_forward_taps(c, x, out, B, h, w, workspace, workspace_bytes, nullptr, stream);
}

}  // extern "C"
