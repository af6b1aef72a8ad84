// C-ABI entry points of SURVEY 8(b)'s export list that are not whole-model drivers: the operator-level backward kernels
// (conv3x3 dgrad / wgrad, pointwise dgrad / wgrad, upsample-head dgrad) and the data-parallel exchange step over RCCL.
// Reference semantics: autograd of nn.Conv2d as driven by train.py:256-264 (fp32); the reference itself has no gradient exchange.
#include <dlfcn.h>

#include <mutex>

#include "lfsr_internal.h"

namespace {
inline int npad32c(int n) { return (n + 31) / 32 * 32; }
}  // namespace

extern "C" {

// ---- transposed weight packs used by the data gradients ---------------------------------------------------------------------
size_t lfsr_packed_weight_tr_floats(int O, int C, int taps) {
  size_t f = (size_t)taps * (size_t)npad32c(C) * (size_t)O;
  if (O == 64 && C == 64 && taps == 9) f += LFSR_CONV3_WINO_FLOATS;   // the 3x3 dgrad runs in Winograd form too
  return f;
}

int lfsr_pack_conv_weight_tr(const float* w, float* packed_T, int O, int C, int taps, void* stream) {
  if (!w || !packed_T || O <= 0 || C <= 0 || taps <= 0) return LFSR_E_ARG;
  return lfsr_pack_weight_T(w, packed_T, O, C, taps, taps == 9 ? 1 : 0, lfsr_stream(stream));   // 3x3: taps flipped (correlation -> its adjoint)
}

// ---- per-view 3x3 conv 64 -> 64 ------------------------------------------------------------------------------------------------
int lfsr_conv3x3_dgrad(const float* dy, int dy_stride, int dy_choff, const float* wT_packed, float* dx, int dx_stride, int dx_choff,
                       const float* r1, int r1_stride, int r1_choff, const float* act, int act_stride, int act_choff, float act_slope,
                       int n_img, int h, int w, void* stream) {
  if (!dy || !wT_packed || !dx || n_img <= 0 || h <= 0 || w <= 0) return LFSR_E_ARG;
  if (dy_stride < dy_choff + 64 || dx_stride < dx_choff + 64 || (r1 && r1_stride < r1_choff + 64) || (act && act_stride < act_choff + 64)) return LFSR_E_ARG;
  if ((long long)n_img * h * w >= (1LL << 31) / 4) return LFSR_E_ARG;
  return lfsr_conv3x3_bwd_data(dy, dy_stride, dy_choff, wT_packed, dx, dx_stride, dx_choff, r1, r1_stride, r1_choff, act, act_stride, act_choff, act_slope,
                               n_img, h, w, lfsr_stream(stream));
}

size_t lfsr_conv3x3_wgrad_workspace_floats(int n_img, int h, int w) {
  if (n_img <= 0 || h <= 0 || w <= 0) return 0;
  return (size_t)lfsr_wgrad_conv3_blocks(n_img, h, w) * 9 * 64 * 64;
}

int lfsr_conv3x3_wgrad(const float* dy, int dy_stride, int dy_choff, const float* x, int x_stride, int x_choff, float* dw,
                       float* workspace, size_t workspace_floats, int n_img, int h, int w, int accumulate, void* stream) {
  if (!dy || !x || !dw || !workspace || n_img <= 0 || h <= 0 || w <= 0) return LFSR_E_ARG;
  if (dy_stride < dy_choff + 64 || x_stride < x_choff + 64) return LFSR_E_ARG;
  if (workspace_floats < lfsr_conv3x3_wgrad_workspace_floats(n_img, h, w)) return LFSR_E_WS;
  hipStream_t st = lfsr_stream(stream);
  int rc = lfsr_wgrad_conv3_launch(dy, dy_stride, dy_choff, x, x_stride, x_choff, workspace, n_img, h, w, st);
  if (rc) return rc;
  return lfsr_wgrad_reduce(workspace, lfsr_wgrad_conv3_blocks(n_img, h, w), nullptr, 0, dw, 64, 64, 9, 0, 0, accumulate, 0, 0, st);
}

// ---- pointwise (1x1) conv: y = act(x W^T), W (cout, cin) -----------------------------------------------------------------------
// dgrad: dx[p, 0..cin) = (sum_n dy[p, n] W[n, :]) * LeakyReLU'(act[p, :]) (act = the saved activation of the PRECEDING layer, may be NULL).
// Built for the shapes the models' backward uses: cout = 64 (wT_packed from lfsr_pack_conv_weight_tr(W, cout, cin, taps = 1)).
int lfsr_pointwise_dgrad(const float* dy, int dy_stride, int dy_choff, int cout, const float* wT_packed, float* dx, int dx_stride, int dx_choff, int cin,
                         const float* act, int act_stride, int act_choff, float act_slope, long long M, void* stream) {
  if (!dy || !wT_packed || !dx || M <= 0 || M > 0x7fffffffLL || cout != 64 || cin <= 0 || cin % 4) return LFSR_E_ARG;
  if (dy_stride < dy_choff + cout || dx_stride < dx_choff + cin || (act && act_stride < act_choff + cin)) return LFSR_E_ARG;
  LfsrGemm q{};
  q.in_mode = LFSR_IN_SAME; q.out_mode = LFSR_OUT_SAME; q.cin = cout; q.X = dy; q.x_stride = dy_stride; q.x_choff = dy_choff; q.Wp = wT_packed;
  q.Y = dx; q.y_stride = dx_stride; q.y_choff = dx_choff; q.Mk = act; q.mk_stride = act_stride; q.mk_choff = act_choff; q.mk_slope = act_slope;
  q.M = (int)M; q.N = cin; q.A = 1; q.h = 1; q.w = 1; q.ntaps = 1; q.CH = cin;
  return lfsr_bwd_gemm(q, lfsr_stream(stream));
}

size_t lfsr_pointwise_wgrad_workspace_floats(long long M, int cout, int cin) {
  if (M <= 0 || M > 0x7fffffffLL || cout <= 0 || cin <= 0) return 0;
  return lfsr_wgrad_partial_floats((int)M, 1, cout, cin);
}

int lfsr_pointwise_wgrad(const float* dy, int dy_stride, int dy_choff, int cout, const float* x, int x_stride, int x_choff, int cin, float* dw,
                         float* workspace, size_t workspace_floats, long long M, int accumulate, void* stream) {
  if (!dy || !x || !dw || !workspace || M <= 0 || M > 0x7fffffffLL || cout <= 0 || cout > 64 || cin <= 0) return LFSR_E_ARG;
  if (dy_stride < dy_choff + cout || x_stride < x_choff + cin) return LFSR_E_ARG;
  if (workspace_floats < lfsr_pointwise_wgrad_workspace_floats(M, cout, cin)) return LFSR_E_WS;
  hipStream_t st = lfsr_stream(stream);
  int rc = lfsr_wgrad_launch(LFSR_IN_SAME, LFSR_IN_SAME, dy, dy_stride, dy_choff, x, x_stride, x_choff, workspace, (int)M, cout, cin, 1, 1, 1, 1, st);
  if (rc) return rc;
  return lfsr_wgrad_reduce(workspace, lfsr_wgrad_splits((int)M, 1, cin), nullptr, 0, dw, cout, cin, 1, 0, 0, accumulate, 0, 0, st);
}

// ---- upsample head (DistgSSR.py:24-27,34-35): d(out)/d(f) through PixelShuffle(s), the folded s^2 x 64 matrix and MacPI2SAI ----
// g16: scratch (B*A*A*h*w, 16) receiving the un-shuffled output gradient (rows of the folded matrix' gradient); df (pixels, 64) VCL.
int lfsr_upsample_head_dgrad(const float* dout, const float* wf, float* df, float* g16, int B, int A, int h, int w, int s, void* stream) {
  if (!dout || !wf || !df || !g16 || B <= 0 || A <= 0 || h <= 0 || w <= 0 || (s != 2 && s != 3 && s != 4)) return LFSR_E_ARG;
  return lfsr_head_bwd_data(dout, wf, df, g16, B, A, h, w, s, lfsr_stream(stream));
}

// ---- the one exchange step of data-parallel training: sum-all-reduce of the flat fp32 gradient bucket over RCCL ----
// RCCL is resolved at run time (dlopen of the librccl the process already has, i.e. PyTorch-ROCm's, else /opt/rocm's), so liblfsr_hip.so
// loads -- and every other entry point works -- on a host without it.  `comm` is an ncclComm_t (opaque here); `unique_id` is the
// 128-byte ncclUniqueId that rank 0 obtains and the host ships to the other ranks by any channel (torch.distributed store, MPI, a file).
namespace {
struct UniqueId { char b[128]; };   // ncclUniqueId (rccl.h: char internal[128]), passed BY VALUE to ncclCommInitRank
typedef int (*init_rank_fn)(void**, int, UniqueId, int);
struct Rccl {
  void* h = nullptr;
  int (*GetUniqueId)(void*) = nullptr;
  init_rank_fn CommInitRank = nullptr;
  int (*AllReduce)(const void*, void*, size_t, int, int, void*, hipStream_t) = nullptr;
  int (*CommDestroy)(void*) = nullptr;
  bool ok = false;
};
Rccl g_rccl;
std::once_flag g_rccl_once;

void rccl_load() {
  const char* names[] = {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so"};
  for (const char* n : names) {
    g_rccl.h = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
    if (g_rccl.h) break;
  }
  if (!g_rccl.h) return;
  g_rccl.GetUniqueId = reinterpret_cast<int (*)(void*)>(dlsym(g_rccl.h, "ncclGetUniqueId"));
  g_rccl.CommInitRank = reinterpret_cast<init_rank_fn>(dlsym(g_rccl.h, "ncclCommInitRank"));
  g_rccl.AllReduce = reinterpret_cast<int (*)(const void*, void*, size_t, int, int, void*, hipStream_t)>(dlsym(g_rccl.h, "ncclAllReduce"));
  g_rccl.CommDestroy = reinterpret_cast<int (*)(void*)>(dlsym(g_rccl.h, "ncclCommDestroy"));
  g_rccl.ok = g_rccl.GetUniqueId && g_rccl.CommInitRank && g_rccl.AllReduce && g_rccl.CommDestroy;
}
inline int rccl_rc(int r) { return r == 0 ? LFSR_OK : -(2000 + r); }   // -(2000 + ncclResult_t)
}  // namespace

int lfsr_comm_available(void) {
  std::call_once(g_rccl_once, rccl_load);
  return g_rccl.ok ? 1 : 0;
}

int lfsr_comm_unique_id(void* id128) {
  if (!id128) return LFSR_E_ARG;
  if (!lfsr_comm_available()) return LFSR_E_ARG;
  return rccl_rc(g_rccl.GetUniqueId(id128));
}

int lfsr_comm_init(void** comm, int world, int rank, const void* id128) {
  if (!comm || !id128 || world <= 0 || rank < 0 || rank >= world) return LFSR_E_ARG;
  if (!lfsr_comm_available()) return LFSR_E_ARG;
  UniqueId id;
  __builtin_memcpy(id.b, id128, 128);
  return rccl_rc(g_rccl.CommInitRank(comm, world, id, rank));
}

int lfsr_comm_destroy(void* comm) {
  if (!comm) return LFSR_E_ARG;
  if (!lfsr_comm_available()) return LFSR_E_ARG;
  return rccl_rc(g_rccl.CommDestroy(comm));
}

int lfsr_allreduce(void* grads, size_t n, void* comm, void* stream) {
  if (!grads || !comm || n == 0) return LFSR_E_ARG;
  if (!lfsr_comm_available()) return LFSR_E_ARG;
  return rccl_rc(g_rccl.AllReduce(grads, grads, n, /* ncclFloat32 */ 7, /* ncclSum */ 0, comm, lfsr_stream(stream)));
}

}  // extern "C"
