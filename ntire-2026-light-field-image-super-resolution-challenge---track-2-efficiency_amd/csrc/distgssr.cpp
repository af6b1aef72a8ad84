// Host driver for the DistgSSR forward (get_model.forward, model/SR/DistgSSR.py:29-36) on VCL buffers.
// Pure host code above the C-ABI operator entry points: owns the packed-weight table keyed by the
// reference's state_dict names and the launch sequence; allocates nothing on the device.
#include <cstring>
#include <map>
#include <string>
#include <vector>

#include "lfsr_internal.h"

namespace {
struct Slot {
  size_t off = 0;      // float offset in the packed buffer
  size_t floats = 0;   // packed size
  size_t numel = 0;    // expected raw element count
  int O = 0, C = 0, T = 0, perm = 0, ch = 0;
  bool raw = false, loaded = false;
  // transposed packing used by the data gradients: 0 none, 1 pack_T flipped (3x3), 2 pack_T, 3 chunkT perm 1 ch 16, 4 chunkT perm 0 ch 32
  int kindT = 0;
  size_t offT = 0, floatsT = 0;
  size_t grad_off = 0;   // offset in the flat gradient bucket (state_dict order)
};
inline size_t align64(size_t f) { return (f + 63) / 64 * 64; }  // 256-B granules
}  // namespace

struct lfsr_distgssr {
  int A, s, G, NB, C;
  std::map<std::string, Slot> slots;
  std::vector<std::string> order;   // state_dict order
  size_t n_params = 0;
  size_t packed_floats = 0, off_wf = 0, off_bf = 0;
  float* packed = nullptr;
  bool finalized = false;
  // batched repack (lfsr_distgssr_begin_batched_load): load_param records one descriptor per pack instead of launching it; finalize uploads the
  // table when it changed (parameter addresses are stable across optimizer steps) and launches one kernel per pack kind
  bool batch_mode = false;
  std::vector<LfsrPackDesc> d_gen, d_c3, d_epi, d_epib, uploaded;
  LfsrPackDesc* table_dev = nullptr;
  size_t table_cap = 0;
  // backward: the weight gradient of a 3x3 layer runs on a side stream beside the layer's data gradient (both read the same dY); created on first use
  hipStream_t side = nullptr;
  hipEvent_t ev_fork = nullptr, ev_join = nullptr, ev_a = nullptr, ev_b = nullptr;
  bool profiling = false;
  struct Ev { int cls; hipEvent_t a, b; };
  bool profile_all = true;
  int sample_stride = 1;        // profiling mode 3: events around every sample_stride-th 3x3 conv op only
  long long sample_ctr = 0;
  std::vector<Ev> evs;          // recorded this profiling session
  std::vector<hipEvent_t> free_evs;
  hipEvent_t get_ev() {
    if (!free_evs.empty()) { hipEvent_t e = free_evs.back(); free_evs.pop_back(); return e; }
    hipEvent_t e = nullptr;
    (void)hipEventCreate(&e);
    return e;
  }

  void add(const std::string& k, int O, int Cc, int T, int perm, int ch, bool raw, int kindT = 0) {
    Slot sl;
    sl.O = O; sl.C = Cc; sl.T = T; sl.perm = perm; sl.ch = ch; sl.raw = raw; sl.kindT = kindT;
    sl.numel = (size_t)O * Cc * T;
    sl.floats = raw ? sl.numel : lfsr_packed_weight_floats(O, Cc, T);
    sl.off = packed_floats;
    packed_floats += align64(sl.floats);
    if (kindT == 1 || kindT == 2) sl.floatsT = (size_t)T * ((Cc + 31) / 32 * 32) * O;
    if (kindT == 1 && O == 64 && Cc == 64 && T == 9) sl.floatsT += LFSR_CONV3_WINO_FLOATS;   // lfsr_pack_weight_T appends the Winograd-domain copy
    if (kindT == 3 || kindT == 4) sl.floatsT = (size_t)(O / (kindT == 3 ? 16 : 32)) * ((Cc + 31) / 32 * 32) * (kindT == 3 ? 16 : 32);
    if (kindT) { sl.offT = packed_floats; packed_floats += align64(sl.floatsT); }
    sl.grad_off = n_params;
    n_params += sl.numel;
    slots[k] = sl;
    order.push_back(k);
  }
  const float* wT(const std::string& k) const { return packed + slots.at(k).offT; }
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
      c->add(p + "SpaConv.0.weight", 64, 64, 9, 0, 0, false, 1);
      c->add(p + "SpaConv.2.weight", 64, 64, 9, 0, 0, false, 1);
      c->add(p + "AngConv.0.weight", 16, 64, AA, 0, 0, false, 2);
      c->add(p + "AngConv.2.weight", 16 * AA, 16, 1, 1, 16, false, 3);
      c->add(p + "EPIConv.0.weight", 32, 64, AA, 0, 0, false, 2);
      c->add(p + "EPIConv.2.weight", 32 * A, 32, 1, 0, 0, false, 4);
      c->add(p + "fuse.0.weight", 64, 144, 1, 0, 0, false, 2);
      c->add(p + "fuse.2.weight", 64, 64, 9, 0, 0, false, 1);
    }
    c->add("disentg.Group." + std::to_string(g) + ".conv.weight", 64, 64, 9, 0, 0, false, 1);
  }
  c->add("disentg.conv.weight", 64, 64, 9, 0, 0, false, 1);
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
  if (c->table_dev) (void)hipFree(c->table_dev);
  if (c->ev_fork) (void)hipEventDestroy(c->ev_fork);
  if (c->ev_join) (void)hipEventDestroy(c->ev_join);
  if (c->ev_a) (void)hipEventDestroy(c->ev_a);
  if (c->ev_b) (void)hipEventDestroy(c->ev_b);
  if (c->side) (void)hipStreamDestroy(c->side);
  delete c;
}

int lfsr_distgssr_begin_batched_load(lfsr_distgssr* c) {
  if (!c) return LFSR_E_ARG;
  c->batch_mode = true;
  c->d_gen.clear(); c->d_c3.clear(); c->d_epi.clear(); c->d_epib.clear();
  return LFSR_OK;
}

int lfsr_distgssr_profile(lfsr_distgssr* c, int enable) {
  if (!c) return LFSR_E_ARG;
  for (auto& e : c->evs) { c->free_evs.push_back(e.a); c->free_evs.push_back(e.b); }
  c->evs.clear();
  c->profiling = enable != 0;
  c->profile_all = enable == 1;   // enable = 2: events around the 3x3 conv ops only (class 0); 3: around every 4th of them (53 ops per
  c->sample_stride = enable == 3 ? 4 : 1;   // forward and 4 are coprime: every layer is sampled once in four steps), the cheapest live measurement
  c->sample_ctr = 0;
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
  } else if (c->batch_mode && (lfsr_conv3_variant_mask() == LFSR_W_WINO4 || !(sl.O == 64 && sl.C == 64 && sl.T == 9))) {
    // record the packs; lfsr_distgssr_finalize launches them, one kernel per kind (3x3 weights with another kernel selection keep the eager path below)
    auto pad32 = [](int v) { return (v + 31) / 32 * 32; };
    float* dst = c->packed + sl.off;
    float* tdst = c->packed + sl.offT;
    const bool c3 = sl.O == 64 && sl.C == 64 && sl.T == 9 && sl.perm == 0;
    if (c3) c->d_c3.push_back(LfsrPackDesc{data, dst, dst + LFSR_CONV3_DIRECT_FLOATS + LFSR_CONV3_WINO2_FLOATS, 0, 64, 64, 9, 64, 0, 0, 0});
    else {
      c->d_gen.push_back(LfsrPackDesc{data, dst, nullptr, 0, sl.O, sl.C, sl.T, pad32(sl.O), sl.perm, sl.ch, 0});
      if (sl.O == 32 && sl.C == 64 && sl.T == 25 && sl.perm == 0) {
        c->d_epi.push_back(LfsrPackDesc{dst, dst + 25 * 32 * 64, nullptr, 0, 32, 64, 25, 32, 0, 0, 0});
        c->d_epib.push_back(LfsrPackDesc{dst, dst + 25 * 32 * 64 + LFSR_EPI_WINO_FLOATS, nullptr, 0, 32, 64, 25, 32, 0, 0, 0});     // the three bf16 planes (epi_b3.hip)
      }
      if (sl.O == 160 && sl.C == 32 && sl.T == 1 && sl.perm == 0) c->d_epib.push_back(LfsrPackDesc{dst, dst + 160 * 32, nullptr, 1, 160, 32, 1, 160, 0, 0, 0});
    }
    if (sl.kindT == 1 && c3) c->d_c3.push_back(LfsrPackDesc{data, tdst, tdst + LFSR_CONV3_DIRECT_FLOATS + LFSR_CONV3_WINO2_FLOATS, 0, 64, 64, 9, 64, 0, 0, 1});
    else if (sl.kindT == 1) c->d_gen.push_back(LfsrPackDesc{data, tdst, nullptr, 1, sl.O, sl.C, sl.T, pad32(sl.C), 0, 0, 1});
    if (sl.kindT == 2) c->d_gen.push_back(LfsrPackDesc{data, tdst, nullptr, 1, sl.O, sl.C, sl.T, pad32(sl.C), 0, 0, 0});
    if (sl.kindT == 3) c->d_gen.push_back(LfsrPackDesc{data, tdst, nullptr, 2, sl.O, sl.C, 1, pad32(sl.C), 1, 16, 0});
    if (sl.kindT == 4) c->d_gen.push_back(LfsrPackDesc{data, tdst, nullptr, 2, sl.O, sl.C, 1, pad32(sl.C), 0, 32, 0});
  } else {
    const int vmask = lfsr_conv3_variant_mask();   // only the Winograd-domain copies the selected 3x3 kernel reads (this runs once per weight and training step)
    int rc = lfsr_pack_conv_weight_m(data, c->packed + sl.off, sl.O, sl.C, sl.T, sl.perm, sl.ch, vmask, stream);
    if (rc) return rc;
    float* tdst = c->packed + sl.offT;
    if (sl.kindT == 1) rc = lfsr_pack_weight_T_m(data, tdst, sl.O, sl.C, sl.T, 1, vmask, lfsr_stream(stream));
    if (sl.kindT == 2) rc = lfsr_pack_weight_T(data, tdst, sl.O, sl.C, sl.T, 0, lfsr_stream(stream));
    if (sl.kindT == 3) rc = lfsr_pack_weight_chunkT(data, tdst, sl.O, sl.C, 16, 1, lfsr_stream(stream));
    if (sl.kindT == 4) rc = lfsr_pack_weight_chunkT(data, tdst, sl.O, sl.C, 32, 0, lfsr_stream(stream));
    if (rc) return rc;
  }
  sl.loaded = true;
  return LFSR_OK;
}

int lfsr_distgssr_finalize(lfsr_distgssr* c, void* stream) {
  if (!c || !c->packed) return LFSR_E_ARG;
  for (auto& kv : c->slots)
    if (!kv.second.loaded) return LFSR_E_ARG;
  if (c->batch_mode) {
    c->batch_mode = false;
    std::vector<LfsrPackDesc> all(c->d_gen);
    all.insert(all.end(), c->d_c3.begin(), c->d_c3.end());
    all.insert(all.end(), c->d_epi.begin(), c->d_epi.end());
    all.insert(all.end(), c->d_epib.begin(), c->d_epib.end());
    bool same = all.size() == c->uploaded.size();
    for (size_t i = 0; same && i < all.size(); ++i) same = memcmp(&all[i], &c->uploaded[i], sizeof(LfsrPackDesc)) == 0;
    if (!same) {
      if (all.size() > c->table_cap) {
        if (c->table_dev) (void)hipFree(c->table_dev);
        c->table_dev = nullptr; c->table_cap = 0;
        hipError_t e = hipMalloc(reinterpret_cast<void**>(&c->table_dev), (all.size() + 64) * sizeof(LfsrPackDesc));
        if (e != hipSuccess) return LFSR_HIP_ERR(e);
        c->table_cap = all.size() + 64;
      }
      c->uploaded = all;   // (the copy's source outlives it; a pageable-memory copy is staged before the call returns)
      hipError_t e = hipMemcpyAsync(c->table_dev, c->uploaded.data(), all.size() * sizeof(LfsrPackDesc), hipMemcpyHostToDevice, lfsr_stream(stream));
      if (e != hipSuccess) return LFSR_HIP_ERR(e);
    }
    const int ng = (int)c->d_gen.size(), n3 = (int)c->d_c3.size(), ne = (int)c->d_epi.size(), nb3 = (int)c->d_epib.size();
    int rcb = lfsr_pack_generic_batch(c->table_dev, ng, lfsr_stream(stream));
    if (!rcb) rcb = lfsr_pack_conv3_raw_wino4_batch(c->table_dev + ng, n3, lfsr_stream(stream));
    if (!rcb) rcb = lfsr_pack_epi_wino_batch(c->table_dev + ng + n3, ne, lfsr_stream(stream));   // reads the direct packs written by the first launch
    if (!rcb) rcb = lfsr_pack_epi_b3_batch(c->table_dev + ng + n3 + ne, nb3, lfsr_stream(stream));  // likewise
    if (rcb) return rcb;
  }
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
  if ((long long)B * c->A * c->A * h * w * 160 * 4 >= (1LL << 30)) return LFSR_E_ARG;  // every activation tensor < 1 GiB (the F(4x4) conv kernel's offset range; 32-bit byte offsets everywhere): callers split the batch (capi.py)
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
    if (c->profiling && (c->profile_all || ((cls) == 0 && (c->sample_stride <= 1 || (c->sample_ctr++ % c->sample_stride) == 0)))) { \
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

// ------------------------------------------------------------------------------------------------------------
// Training: forward that keeps every activation the backward needs, and the backward itself
// (autograd of model/SR/DistgSSR.py:29-111 as driven by train.py:256-264, fp32).
// ------------------------------------------------------------------------------------------------------------
namespace {
struct TrainWs {
  // saved activations
  float *F0, *D;
  std::vector<float*> S1, CAT, A16, EH, EV, FZ, OUT, GOUT;
  // backward scratch
  float *g[4], *dF, *dS1, *dCAT, *dA16, *dE32, *dE32V, *G16, *XG9, *P[4], *PA, *small;   // PA: partial slabs of the angular branch (its own: it may run beside the epipolar one)
  size_t pfloats;
  size_t total;
};

size_t max_partial_floats(const lfsr_distgssr* c, int B, int h, int w) {
  const int A = c->A, AA = A * A;
  const int npix = B * AA * h * w, nlr = B * h * w, nepi = B * A * h * w;
  size_t m = 0;
  auto up = [&](size_t v) { if (v > m) m = v; };
  up(lfsr_wgrad_partial_floats(npix, 9, 64, 64));
  up((size_t)256 * 9 * 64 * 64);
  up(lfsr_branch_bwd_partial_floats(B, A, h, w));   // (incl. the EPI-line weight gradient's one slab per block)
  up(lfsr_wgrad_partial_floats(npix, 1, 64, 144));
  up(lfsr_wgrad_partial_floats(nlr, AA, 16, 64));
  up(lfsr_wgrad_partial_floats(nlr, AA, 16, 16));
  up(lfsr_wgrad_partial_floats(nepi, AA, 32, 64));
  up(lfsr_wgrad_partial_floats(nepi, A, 32, 32));
  up(lfsr_wgrad_partial_floats(npix, 1, 16, 64));
  up(lfsr_wgrad_partial_floats(npix, 1, 64, 16));
  return m;
}

void train_layout(const lfsr_distgssr* c, int B, int h, int w, float* base, TrainWs& t) {
  const int A = c->A, AA = A * A, nb = c->G * c->NB;
  const size_t npix = (size_t)B * AA * h * w, nlr = (size_t)B * h * w, nepi = (size_t)B * A * h * w;
  size_t o = 0;
  auto take = [&](size_t floats) { float* p = base ? base + o : nullptr; o += align64(floats); return p; };
  t.F0 = take(npix * 64); t.D = take(npix * 64);
  t.S1.resize(nb); t.CAT.resize(nb); t.A16.resize(nb); t.EH.resize(nb); t.EV.resize(nb); t.FZ.resize(nb); t.OUT.resize(nb); t.GOUT.resize(c->G);
  for (int i = 0; i < nb; ++i) {
    t.S1[i] = take(npix * 64); t.CAT[i] = take(npix * 144); t.A16[i] = take(nlr * 16);
    t.EH[i] = take(nepi * 32); t.EV[i] = take(nepi * 32); t.FZ[i] = take(npix * 64); t.OUT[i] = take(npix * 64);
  }
  for (int g = 0; g < c->G; ++g) t.GOUT[g] = take(npix * 64);
  for (int i = 0; i < 4; ++i) t.g[i] = take(npix * 64);
  t.dF = take(npix * 64); t.dS1 = take(npix * 64); t.dCAT = take(npix * 144);
  t.dA16 = take(nlr * 16); t.dE32 = take(nepi * 32); t.dE32V = take(nepi * 32); t.G16 = take(npix * 16); t.XG9 = take(npix * 16);
  t.pfloats = max_partial_floats(c, B, h, w);
  for (int i = 0; i < 4; ++i) t.P[i] = take(t.pfloats);
  t.PA = take(lfsr_branch_bwd_partial_floats(B, c->A, h, w));
  t.small = take(64 * 1024);
  t.total = o;
}
}  // namespace

extern "C" {

size_t lfsr_distgssr_num_params(const lfsr_distgssr* c) { return c ? c->n_params : 0; }

int lfsr_distgssr_param_offset(const lfsr_distgssr* c, const char* key, size_t* off, size_t* numel) {
  if (!c || !key) return LFSR_E_ARG;
  auto it = c->slots.find(key);
  if (it == c->slots.end()) return LFSR_E_ARG;
  if (off) *off = it->second.grad_off;
  if (numel) *numel = it->second.numel;
  return LFSR_OK;
}

size_t lfsr_distgssr_train_workspace_bytes(const lfsr_distgssr* c, int B, int h, int w) {
  if (!c || B <= 0 || h <= 0 || w <= 0) return 0;
  TrainWs t;
  train_layout(c, B, h, w, nullptr, t);
  return t.total * sizeof(float);
}

// Diagnostic / parity aid: where forward_train left the activations the backward reads (post-LeakyReLU values, so their signs are the LeakyReLU'
// masks of the data gradients).  which: 0 S1 (SpaConv.0 out, VCL 64), 1 CAT (VCL 144: Spa | Ang | EpiH | EpiV), 2 A16 (AngConv.0 out, rows (b,y,x), 16),
// 3 EH / 4 EV (EPIConv.0 out, rows (b*A+u,y,x) / (b*A+v,y,x), 32), 5 FZ (fuse.0 out, VCL 64), 6 OUT (block output, VCL 64).  index = group * n_block + block.
int lfsr_distgssr_train_saved(const lfsr_distgssr* c, int B, int h, int w, int which, int index, size_t* offset_floats, size_t* numel) {
  if (!c || B <= 0 || h <= 0 || w <= 0 || index < 0 || index >= c->G * c->NB || !offset_floats || !numel) return LFSR_E_ARG;
  TrainWs t;
  float* const base = reinterpret_cast<float*>(uintptr_t(4096));   // any non-null base: only differences are used
  train_layout(c, B, h, w, base, t);
  const size_t npix = (size_t)B * c->A * c->A * h * w, nlr = (size_t)B * h * w, nepi = (size_t)B * c->A * h * w;
  const float* p = nullptr; size_t n = 0;
  switch (which) {
    case 0: p = t.S1[index]; n = npix * 64; break;
    case 1: p = t.CAT[index]; n = npix * 144; break;
    case 2: p = t.A16[index]; n = nlr * 16; break;
    case 3: p = t.EH[index]; n = nepi * 32; break;
    case 4: p = t.EV[index]; n = nepi * 32; break;
    case 5: p = t.FZ[index]; n = npix * 64; break;
    case 6: p = t.OUT[index]; n = npix * 64; break;
    default: return LFSR_E_ARG;
  }
  *offset_floats = (size_t)(p - base); *numel = n;
  return LFSR_OK;
}

int lfsr_distgssr_forward_train(lfsr_distgssr* c, const float* x, float* out, int B, int h, int w, void* workspace, size_t workspace_bytes, void* stream) {
  if (!c || !x || !out || !workspace || B <= 0 || h <= 0 || w <= 0 || !c->finalized || ((uintptr_t)workspace & 15)) return LFSR_E_ARG;
  TrainWs t;
  train_layout(c, B, h, w, (float*)workspace, t);
  if (workspace_bytes < t.total * sizeof(float)) return LFSR_E_WS;
  if ((long long)B * c->A * c->A * h * w * 160 * 4 >= (1LL << 30)) return LFSR_E_ARG;   // (as in the forward)
  const int A = c->A, AA = A * A, nimg = B * AA;
  const float L = 0.1f;
  hipStream_t st = lfsr_stream(stream);
  int rc;
#define RC(call) do { rc = (call); if (rc) return rc; } while (0)
  auto conv = [&](const float* in, const std::string& key, float* o, int ostride, const float* res, float slope) -> int {
    return lfsr_conv3x3_fwd(in, 64, 0, c->w(key), o, ostride, 0, res, 64, 0, nullptr, 0, 0, nimg, h, w, slope, stream);
  };
  RC(lfsr_initconv_fwd(x, c->w("init_conv.weight"), t.F0, 64, 0, B, A, h, w, stream));
  const float* cur = t.F0;
  for (int g = 0; g < c->G; ++g) {
    const float* gin = cur;
    for (int b = 0; b < c->NB; ++b) {
      const int i = g * c->NB + b;
      std::string p = "disentg.Group." + std::to_string(g) + ".Block." + std::to_string(b) + ".";
      RC(conv(cur, p + "SpaConv.0.weight", t.S1[i], 64, nullptr, L));
      RC(conv(t.S1[i], p + "SpaConv.2.weight", t.CAT[i], 144, nullptr, L));
      RC(lfsr_angconv_fwd(cur, 64, 0, c->w(p + "AngConv.0.weight"), c->w(p + "AngConv.2.weight"), t.A16[i], t.CAT[i], 144, 64, B, A, h, w, L, stream));
      if (lfsr_epi_fused_ok(A, h, w)) {
        RC(lfsr_epi_fused_launch(cur, 64, 0, c->w(p + "EPIConv.0.weight"), c->w(p + "EPIConv.2.weight"), t.CAT[i], 144, 80, 112, t.EH[i], t.EV[i], B, A, h, w, 3, L, st));
      } else {
        RC(lfsr_epiconv_gather(cur, 64, 0, c->w(p + "EPIConv.0.weight"), c->w(p + "EPIConv.2.weight"), t.EH[i], t.CAT[i], 144, 80, B, A, h, w, 0, L, st));
        RC(lfsr_epiconv_gather(cur, 64, 0, c->w(p + "EPIConv.0.weight"), c->w(p + "EPIConv.2.weight"), t.EV[i], t.CAT[i], 144, 112, B, A, h, w, 1, L, st));
      }
      RC(lfsr_pointwise_fwd(t.CAT[i], 144, 0, 144, c->w(p + "fuse.0.weight"), nullptr, t.FZ[i], 64, 0, nimg * h * w, 64, L, stream));
      RC(conv(t.FZ[i], p + "fuse.2.weight", t.OUT[i], 64, cur, 1.0f));
      cur = t.OUT[i];
    }
    RC(conv(cur, "disentg.Group." + std::to_string(g) + ".conv.weight", t.GOUT[g], 64, gin, 1.0f));
    cur = t.GOUT[g];
  }
  RC(conv(cur, "disentg.conv.weight", t.D, 64, t.F0, 1.0f));
  RC(lfsr_upsample_head_fwd(t.D, 64, 0, c->packed + c->off_wf, c->packed + c->off_bf, x, out, B, A, h, w, c->s, stream));
#undef RC
  return LFSR_OK;
}

static int distgssr_backward_impl(lfsr_distgssr* c, const float* x, const float* dout, int B, int h, int w, void* workspace, size_t workspace_bytes,
                                  float* grads, size_t n_grads, void* stream);

int lfsr_distgssr_backward(lfsr_distgssr* c, const float* x, const float* dout, int B, int h, int w, void* workspace, size_t workspace_bytes,
                           float* grads, size_t n_grads, void* stream) {
  const int rc = distgssr_backward_impl(c, x, dout, B, h, w, workspace, workspace_bytes, grads, n_grads, stream);
  // An error return between a fork and its join would leave side-stream kernels queued that still read and write the caller's workspace and gradient bucket:
  // whatever the failure was, both streams are drained before the caller gets the buffers back.
  if (rc != LFSR_OK && c && c->side) { (void)hipStreamSynchronize(c->side); (void)hipStreamSynchronize(lfsr_stream(stream)); }
  return rc;
}

static int distgssr_backward_impl(lfsr_distgssr* c, const float* x, const float* dout, int B, int h, int w, void* workspace, size_t workspace_bytes,
                                  float* grads, size_t n_grads, void* stream) {
  if (!c || !x || !dout || !workspace || !grads || B <= 0 || h <= 0 || w <= 0 || !c->finalized || n_grads != c->n_params) return LFSR_E_ARG;
  if (!(c->A & 1)) return LFSR_E_ARG;   // the EPI data gradient's line-shift gathers assume the symmetric padding of odd angRes
  TrainWs t;
  train_layout(c, B, h, w, (float*)workspace, t);
  if (workspace_bytes < t.total * sizeof(float)) return LFSR_E_WS;
  const int A = c->A, AA = A * A, nimg = B * AA;
  const int npix = nimg * h * w;
  const float L = 0.1f;
  hipStream_t st = lfsr_stream(stream);
  int rc;
#define RC(call) do { rc = (call); if (rc) return rc; } while (0)
  auto G = [&](const std::string& k) -> float* { return grads + c->slots.at(k).grad_off; };
  // Two streams for the branch gradients only (below; LFSR_BWD_OVERLAP=0 under LFSR_LAB keeps one stream).  Measured negative or neutral in round 3 and removed in
  // round 4 (profiles/r03_logs/c6_overlap.txt, train_overlap_ab.txt): the 3x3 weight gradients beside their data gradients (25.3 vs 24.5 ms: both are persistent
  // one-block-per-CU grids), fuse.0's weight gradient beside its data gradient, the slab reduces on the side stream.  Not under stream capture.
  bool overlap_br = false;
  {
    const char* osel = lfsr_sel("LFSR_BWD_OVERLAP");
    const int omode = osel ? atoi(osel) : 2;
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (omode && hipStreamIsCapturing(st, &cs) == hipSuccess && cs == hipStreamCaptureStatusNone) {
      if (!c->side) {
        bool ok = hipStreamCreateWithFlags(&c->side, hipStreamNonBlocking) == hipSuccess;
        ok = ok && hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming) == hipSuccess && hipEventCreateWithFlags(&c->ev_join, hipEventDisableTiming) == hipSuccess;
        ok = ok && hipEventCreateWithFlags(&c->ev_a, hipEventDisableTiming) == hipSuccess && hipEventCreateWithFlags(&c->ev_b, hipEventDisableTiming) == hipSuccess;
        if (!ok) { if (c->side) (void)hipStreamDestroy(c->side); c->side = nullptr; }
      }
      overlap_br = c->side != nullptr && (omode & 2);
    }
  }
  // weight gradient of a 3x3 conv: dW[tap][n][k] = sum_m g[m][n] * xin[conv3 src(m,tap)][k]
  auto wgrad3 = [&](const std::string& key, const float* xin, const float* g, int g_stride) -> int {
    int r = lfsr_wgrad_conv3_launch(g, g_stride, 0, xin, 64, 0, t.P[0], nimg, h, w, st);
    if (!r) r = lfsr_wgrad_reduce(t.P[0], lfsr_wgrad_conv3_blocks(nimg, h, w), nullptr, 0, G(key), 64, 64, 9, 0, 0, 0, 0, 0, st);
    return r;
  };
  auto dgrad3 = [&](const float* dy, int dy_stride, const std::string& key, float* dx, const float* r1, const float* r2_unused, const float* mk, int mk_stride) -> int {
    (void)r2_unused;
    return lfsr_conv3x3_bwd_data(dy, dy_stride, 0, c->wT(key), dx, 64, 0, r1, 64, 0, mk, mk_stride, 0, L, nimg, h, w, st);
  };
  auto pick = [&](const float* a, const float* b, const float* d) -> float* {
    for (int i = 0; i < 4; ++i)
      if (t.g[i] != a && t.g[i] != b && t.g[i] != d) return t.g[i];
    return nullptr;
  };

  // ---- head: out = PS(Wf f + bf) + bilinear(x) ---------------------------------------------------------------
  float* dD = t.g[0];
  RC(lfsr_head_bwd_data(dout, c->packed + c->off_wf, dD, t.G16, B, A, h, w, c->s, st));
  {
    float* dWf = t.small;                 // (s*s, 64), rows >= s*s unused
    float* colp = t.small + 16 * 64;      // column-sum partials
    RC(lfsr_wgrad_launch(LFSR_IN_SAME, LFSR_IN_SAME, t.G16, 16, 0, t.D, 64, 0, t.P[0], npix, 16, 64, 1, h, w, 1, st));
    RC(lfsr_wgrad_reduce(t.P[0], lfsr_wgrad_splits(npix, 1, 64), nullptr, 0, dWf, 16, 64, 1, 0, 0, 0, 0, 0, st));
    int nblk = 0;
    RC(lfsr_colsum(t.G16, npix, 16, colp, &nblk, st));
    if ((size_t)(16 * 64 + nblk * 16) > 64 * 1024) return LFSR_E_WS;
    RC(lfsr_head_fold_bwd(dWf, colp, nblk, c->w("upsample.0.weight"), c->w("upsample.0.bias"), c->w("upsample.2.weight"),
                          G("upsample.0.weight"), G("upsample.0.bias"), G("upsample.2.weight"), c->s, st));
  }
  // ---- cascade conv: D = conv(GOUT[last]) + F0 ----------------------------------------------------------------
  const float* last = t.GOUT[c->G - 1];
  RC(wgrad3("disentg.conv.weight", last, dD, 64));
  float* gcur = pick(dD, nullptr, nullptr);
  RC(dgrad3(dD, 64, "disentg.conv.weight", gcur, nullptr, nullptr, nullptr, 0));
  // ---- groups, reversed ----------------------------------------------------------------------------------------
  for (int g = c->G - 1; g >= 0; --g) {
    float* dG = gcur;   // gradient at the group's output; also flows through the group skip to its input
    bool skip_fused = false;
    const float* blk_last = t.OUT[g * c->NB + c->NB - 1];
    std::string gk = "disentg.Group." + std::to_string(g) + ".conv.weight";
    RC(wgrad3(gk, blk_last, dG, 64));
    float* gy = pick(dD, dG, nullptr);
    RC(dgrad3(dG, 64, gk, gy, nullptr, nullptr, nullptr, 0));
    for (int b = c->NB - 1; b >= 0; --b) {
      const int i = g * c->NB + b;
      std::string p = "disentg.Group." + std::to_string(g) + ".Block." + std::to_string(b) + ".";
      const float* Xin = b > 0 ? t.OUT[i - 1] : (g > 0 ? t.GOUT[g - 1] : t.F0);
      float* gx = pick(dD, dG, gy);
      // fuse.2 : OUT = conv(FZ) + Xin
      RC(wgrad3(p + "fuse.2.weight", t.FZ[i], gy, 64));
      RC(dgrad3(gy, 64, p + "fuse.2.weight", t.dF, nullptr, nullptr, t.FZ[i], 64));
      // fuse.0 : FZ = lrelu(1x1(CAT))
      {   // streaming kernel (every row of dF and CAT read once, one slab per block); else the generic split-K kernel
        float* Pw = t.PA;
        int rc5 = lfsr_wgrad_pw144_launch(t.dF, 64, 0, t.CAT[i], 144, 0, Pw, npix, st);
        int slabs = lfsr_wgrad_pw144_blocks(npix);
        if (rc5 == LFSR_E_ARG) {
          Pw = t.P[0];                    // generic split-K kernel: larger slabs
          rc5 = lfsr_wgrad_launch(LFSR_IN_SAME, LFSR_IN_SAME, t.dF, 64, 0, t.CAT[i], 144, 0, Pw, npix, 64, 144, 1, h, w, 1, st); slabs = lfsr_wgrad_splits(npix, 1, 144);
        }
        RC(rc5);
        RC(lfsr_wgrad_reduce(Pw, slabs, nullptr, 0, G(p + "fuse.0.weight"), 64, 144, 1, 0, 0, 0, 0, 0, st));
      }
      {
        LfsrGemm q{};
        q.in_mode = LFSR_IN_SAME; q.out_mode = LFSR_OUT_SAME; q.cin = 64; q.X = t.dF; q.x_stride = 64; q.Wp = c->wT(p + "fuse.0.weight");
        q.Y = t.dCAT; q.y_stride = 144; q.Mk = t.CAT[i]; q.mk_stride = 144; q.mk_slope = L;
        q.M = npix; q.N = 144; q.A = 1; q.h = 1; q.w = 1; q.ntaps = 1; q.CH = 144;
        RC(lfsr_bwd_gemm(q, st));
      }
      // SpaConv : CAT[0:64] = lrelu(conv(S1)), S1 = lrelu(conv(Xin))
      RC(wgrad3(p + "SpaConv.2.weight", t.S1[i], t.dCAT, 144));
      RC(dgrad3(t.dCAT, 144, p + "SpaConv.2.weight", t.dS1, nullptr, nullptr, t.S1[i], 64));
      RC(wgrad3(p + "SpaConv.0.weight", Xin, t.dS1, 64));
      // gx = gy (block skip) + dSpa; in the group's first block the group skip dG rides along as the second residual (else: one more pass over gx at the group's end)
      if (b == 0 && !skip_fused) {
        const int r2rc = lfsr_conv3x3_bwd_data_r2(t.dS1, 64, c->wT(p + "SpaConv.0.weight"), gx, gy, dG, nimg, h, w, st);
        if (r2rc == LFSR_OK) skip_fused = true; else if (r2rc != LFSR_E_ARG) return r2rc;
      }
      if (!(b == 0 && skip_fused)) RC(dgrad3(t.dS1, 64, p + "SpaConv.0.weight", gx, gy, nullptr, nullptr, 0));
      // AngConv : CAT[64:80] = PS(lrelu(1x1(A16))), A16 = lrelu(convAxA(Xin))          (branch_bwd.cpp; also exported as lfsr_angconv_bwd)
      // EPIConv (horizontal, then vertical; shared weights -> both partial sets summed in one reduce)                 (lfsr_epiconv_hv_bwd)
      const float *wa0 = c->w(p + "AngConv.0.weight"), *wa0T = c->wT(p + "AngConv.0.weight"), *wa2T = c->wT(p + "AngConv.2.weight");
      const float *we0 = c->w(p + "EPIConv.0.weight"), *we0T = c->wT(p + "EPIConv.0.weight"), *we2T = c->wT(p + "EPIConv.2.weight");
      if (!overlap_br) {
        RC(lfsr_ang_branch_bwd(t.dCAT, 144, 64, Xin, t.A16[i], wa0, wa0T, wa2T, gx, G(p + "AngConv.0.weight"), G(p + "AngConv.2.weight"), t.dA16, t.PA, B, A, h, w, L, st));
        RC(lfsr_epi_branch_bwd(t.dCAT, 144, 80, 112, Xin, t.EH[i], t.EV[i], we0, we0T, we2T, gx, G(p + "EPIConv.0.weight"), G(p + "EPIConv.2.weight"), t.dE32, t.dE32V, t.P,
                               B, A, h, w, L, st));
      } else {
        // Two streams.  The launches of these branches are small (18-30 us each, a few hundred blocks): seven of them per block in a row leave most of the chip idle.
        //   side:  [ang p1: AngConv.2 wgrad, AngConv.2 dgrad -> dA16, AngConv.0 wgrad] ......... wait(ev_b) [EPIConv.0 wgrad (reads dE_h, dE_v, x)]
        //   main:  [epi p1: EPIConv.2 wgrad + dgrad, both passes -> dE_h, dE_v] rec(ev_b) wait(ev_a) [AngConv.0 dgrad: dx +=] [EPIConv.0 dgrad H, V: dx +=] wait(ev_join)
        // The three read-modify-writes of dx stay in one stream, in the order of the one-stream form (same bits).
        if (hipEventRecord(c->ev_fork, st) != hipSuccess || hipStreamWaitEvent(c->side, c->ev_fork, 0) != hipSuccess) return LFSR_E_ARG;
        RC(lfsr_ang_branch_bwd_p1(t.dCAT, 144, 64, Xin, t.A16[i], wa2T, G(p + "AngConv.0.weight"), G(p + "AngConv.2.weight"), t.dA16, t.PA, B, A, h, w, L, c->side));
        if (hipEventRecord(c->ev_a, c->side) != hipSuccess) return LFSR_E_ARG;
        RC(lfsr_epi_branch_bwd_p1(t.dCAT, 144, 80, 112, t.EH[i], t.EV[i], we2T, G(p + "EPIConv.2.weight"), t.dE32, t.dE32V, t.P, B, A, h, w, L, st));
        if (hipEventRecord(c->ev_b, st) != hipSuccess || hipStreamWaitEvent(c->side, c->ev_b, 0) != hipSuccess) return LFSR_E_ARG;
        RC(lfsr_epi_branch_bwd_p2w(t.dE32, t.dE32V, Xin, G(p + "EPIConv.0.weight"), t.P, B, A, h, w, c->side));
        if (hipEventRecord(c->ev_join, c->side) != hipSuccess) return LFSR_E_ARG;
        if (hipStreamWaitEvent(st, c->ev_a, 0) != hipSuccess) return LFSR_E_ARG;
        RC(lfsr_ang_branch_bwd_p2(t.dA16, wa0, wa0T, gx, B, A, h, w, st));
        RC(lfsr_epi_branch_bwd_p2d(t.dE32, t.dE32V, we0, we0T, gx, B, A, h, w, st));
        if (hipStreamWaitEvent(st, c->ev_join, 0) != hipSuccess) return LFSR_E_ARG;
      }
      gy = gx;
    }
    // group skip: grad at the group's input = (through the blocks) + dG
    if (!skip_fused) RC(lfsr_add_inplace(gy, dG, (long long)npix * 64, st));
    gcur = gy;
  }
  // ---- init_conv: F0 = conv(x) ; dF0 = (through the groups) + dD (cascade skip) -------------------------------
  RC(lfsr_add_inplace(gcur, dD, (long long)npix * 64, st));
  RC(lfsr_init_gather9(x, t.XG9, B, A, h, w, st));
  RC(lfsr_wgrad_launch(LFSR_IN_SAME, LFSR_IN_SAME, gcur, 64, 0, t.XG9, 16, 0, t.P[0], npix, 64, 16, 1, h, w, 1, st));
  RC(lfsr_wgrad_reduce(t.P[0], lfsr_wgrad_splits(npix, 1, 16), nullptr, 0, G("init_conv.weight"), 64, 16, 1, 0, 0, 0, 9, 0, st));
#undef RC
  return LFSR_OK;
}

}  // extern "C"
