// Host driver for the LFT forward (get_model.forward, model/SR/LFT.py:67-98; AngTrans :206-246; SpaTrans :133-203).
// Tokens are VCL pixels: the angular transformer's sequences (25 views at one (y,x)) and the spatial transformer's
// (32x32 positions of one view) are just two stride sets for the same attention kernel; the 5x5 window mask
// (LFT.py:161-174, rebuilt on the CPU per call upstream) is a predicate, so 25 keys per query are visited, not 1024.
#include <stdlib.h>

#include "param_table.h"

struct lfsr_lft {
  int A, s, nlayer;
  LfsrParamTable P;
  std::vector<size_t> ffn_split_spa, ffn_split_ang;   // per layer: offsets (floats) of the feed-forward weights' pre-split bf16 images (ffn_b3.hip)
  bool finalized = false;
};

extern "C" {

int lfsr_lft_create(lfsr_lft** out, int A, int scale, int n_layer, int channels) {
  if (!out || A <= 0 || A > 15 || scale < 2 || scale > 4 || n_layer <= 0 || channels != 64) return LFSR_E_ARG;
  lfsr_lft* c = new lfsr_lft();
  c->A = A; c->s = scale; c->nlayer = n_layer;
  LfsrParamTable& P = c->P;
  P.add("conv_init0.0.weight", 64, 1, 9, 0, 0, true);
  for (int i : {0, 2, 4}) P.add("conv_init." + std::to_string(i) + ".weight", 64, 64, 9);
  for (int b = 0; b < n_layer; ++b) {
    std::string sp = "altblock." + std::to_string(b) + ".spa_trans.";
    P.add(sp + "MLP.weight", 128, 64, 9);                       // (128, 576) == (128, 64, 3, 3): unfold order is c*9 + tap
    P.add(sp + "MLP.weight#lo", 64, 64, 9);                     // the same weights as two 64-output 3x3 convs for the halo-tile kernel
    P.add(sp + "MLP.weight#hi", 64, 64, 9);
    P.add(sp + "norm.weight", 128, 1, 1, 0, 0, true);
    P.add(sp + "norm.bias", 128, 1, 1, 0, 0, true);
    P.add(sp + "attention.in_proj_weight", 384, 128, 1);
    P.add(sp + "attention.out_proj.weight", 128, 128, 1);
    P.add(sp + "feed_forward.0.weight", 128, 1, 1, 0, 0, true);
    P.add(sp + "feed_forward.0.bias", 128, 1, 1, 0, 0, true);
    P.add(sp + "feed_forward.1.weight", 256, 128, 1);
    P.add(sp + "feed_forward.4.weight", 128, 256, 1);
    P.add(sp + "linear.0.weight", 64, 128, 1);
    std::string an = "altblock." + std::to_string(b) + ".ang_trans.";
    P.add(an + "norm.weight", 64, 1, 1, 0, 0, true);
    P.add(an + "norm.bias", 64, 1, 1, 0, 0, true);
    P.add(an + "attention.in_proj_weight", 192, 64, 1);
    P.add(an + "attention.out_proj.weight", 64, 64, 1);
    P.add(an + "feed_forward.0.weight", 64, 1, 1, 0, 0, true);
    P.add(an + "feed_forward.0.bias", 64, 1, 1, 0, 0, true);
    P.add(an + "feed_forward.1.weight", 128, 64, 1);
    P.add(an + "feed_forward.4.weight", 64, 128, 1);
    c->ffn_split_spa.push_back(P.reserve((lfsr_ffn_b3_presplit_bytes(128, 256, 128) + 3) / 4));
    c->ffn_split_ang.push_back(P.reserve((lfsr_ffn_b3_presplit_bytes(64, 128, 64) + 3) / 4));
  }
  P.add("upsampling.0.weight", 64 * scale * scale, 64, 1, 1, 64);
  P.add("upsampling.3.weight", 1, 64, 9, 0, 0, true);
  *out = c;
  return LFSR_OK;
}

void lfsr_lft_destroy(lfsr_lft* c) { delete c; }
size_t lfsr_lft_packed_bytes(const lfsr_lft* c) { return c ? c->P.packed_floats * sizeof(float) : 0; }
int lfsr_lft_set_packed(lfsr_lft* c, void* packed, size_t bytes) { if (!c) return LFSR_E_ARG; c->finalized = false; return c->P.set_packed(packed, bytes); }
int lfsr_lft_load_param(lfsr_lft* c, const char* key, const float* data, size_t numel, void* stream) {
  if (!c) return LFSR_E_ARG;
  c->finalized = false;
  int rc = c->P.load(key, data, numel, stream);
  std::string k(key ? key : "");
  if (!rc && k.size() > 10 && k.compare(k.size() - 10, 10, "MLP.weight") == 0) {
    rc = c->P.load((k + "#lo").c_str(), data, (size_t)64 * 576, stream);
    if (!rc) rc = c->P.load((k + "#hi").c_str(), data + 64 * 576, (size_t)64 * 576, stream);
  }
  return rc;
}
int lfsr_lft_finalize(lfsr_lft* c, void* stream) {
  if (!c || !c->P.packed || !c->P.all_loaded()) return LFSR_E_ARG;
  for (int b = 0; b < c->nlayer; ++b) {      // feed-forward weights split once into the fused kernel's bf16 chunk images
    const std::string sp = "altblock." + std::to_string(b) + ".spa_trans.", an = "altblock." + std::to_string(b) + ".ang_trans.";
    int rc = lfsr_ffn_b3_presplit(c->P.w(sp + "feed_forward.1.weight"), c->P.w(sp + "feed_forward.4.weight"), 128, 256, 128, c->P.packed + c->ffn_split_spa[b], lfsr_stream(stream));
    if (!rc) rc = lfsr_ffn_b3_presplit(c->P.w(an + "feed_forward.1.weight"), c->P.w(an + "feed_forward.4.weight"), 64, 128, 64, c->P.packed + c->ffn_split_ang[b], lfsr_stream(stream));
    if (rc) return rc;
  }
  c->finalized = true;
  return LFSR_OK;
}

static void lft_layout(const lfsr_lft* c, int B, int h, int w, size_t off[20], size_t* total) {
  const size_t npix = (size_t)B * c->A * c->A * h * w;
  size_t o = 0;
  auto take = [&](size_t f) { size_t r = o; o += LfsrParamTable::align64(f); return r; };
  for (int i = 0; i < 7; ++i) off[i] = take(npix * 64);          // F0, BUF0, P, Q, C1, C2, N64
  for (int i = 7; i < 11; ++i) off[i] = take(npix * 128);        // T, TN, V, T2
  off[11] = take(npix * 256);                                    // QK / FF
  off[12] = take(npix * 64 * c->s * c->s);                       // HR mosaic
  off[13] = take((size_t)h * w * 64);                            // spa position map (h*w, 64)
  off[14] = take((size_t)c->A * c->A * 64);                      // ang PE
  off[15] = take((size_t)h * w * 128);                           // embedded spa PE (h*w, 128)
  *total = o;
}

size_t lfsr_lft_workspace_bytes(const lfsr_lft* c, int B, int h, int w) {
  if (!c || B <= 0 || h <= 0 || w <= 0) return 0;
  size_t off[20], tot;
  lft_layout(c, B, h, w, off, &tot);
  return tot * sizeof(float);
}

int lfsr_lft_forward(lfsr_lft* c, const float* x, float* out, int B, int h, int w, void* workspace, size_t workspace_bytes, void* stream) {
  if (!c || !x || !out || !workspace || B <= 0 || h <= 0 || w <= 0 || !c->finalized || ((uintptr_t)workspace & 15)) return LFSR_E_ARG;
  size_t off[20], tot;
  lft_layout(c, B, h, w, off, &tot);
  if (workspace_bytes < tot * sizeof(float)) return LFSR_E_WS;
  const int A = c->A, AA = A * A, nimg = B * AA, HW = h * w;
  const long long npix = (long long)nimg * HW;
  if (npix * 256 * 4 >= (1LL << 31)) return LFSR_E_ARG;   // every activation tensor < 2 GiB (the q | k rows are the widest): the kernels' 32-bit byte offsets; callers split the batch (capi.py)
  float* ws = (float*)workspace;
  float *F0 = ws + off[0], *BUF0 = ws + off[1], *Pb = ws + off[2], *Qb = ws + off[3], *C1 = ws + off[4], *C2 = ws + off[5], *N64 = ws + off[6];
  float *T = ws + off[7], *TN = ws + off[8], *V = ws + off[9], *T2 = ws + off[10], *QK = ws + off[11], *HR = ws + off[12];
  float *SPOS = ws + off[13], *APE = ws + off[14], *SPE = ws + off[15];
  const LfsrParamTable& P = c->P;
  const float L = 0.2f;
  int rc;
#define RC(call) do { rc = (call); if (rc) return rc; } while (0)
  auto conv = [&](const float* in, const std::string& key, float* o, const float* r1, float slope) -> int {
    return lfsr_conv3x3_fwd(in, 64, 0, P.w(key), o, 64, 0, r1, 64, 0, nullptr, 0, 0, nimg, h, w, slope, stream);
  };
  RC(lfsr_initconv_fwd(x, P.w("conv_init0.0.weight"), F0, 64, 0, B, A, h, w, stream));
  RC(conv(F0, "conv_init.0.weight", C1, nullptr, L));
  RC(conv(C1, "conv_init.2.weight", C2, nullptr, L));
  RC(conv(C2, "conv_init.4.weight", BUF0, F0, L));                         // LFT.py:81
  RC(lfsr_lft_position_fwd(SPOS, APE, A, h, w, 64, stream));              // LFT.py:84-85
  const float* cur = BUF0;
  const bool no_ffn_fused = lfsr_sel("LFSR_NO_FFN_FUSED") != nullptr;   // two-launch feed-forward (A/B runs)
  const char* psel = lfsr_sel("LFSR_FFN_PRESPLIT");
  const bool presplit = !(psel && psel[0] == '0');                    // LFSR_FFN_PRESPLIT=0: the kernel splits the weight chunks itself (A/B runs)
  const char* lf = lfsr_sel("LFSR_LN_FUSE");
  // LayerNorms formed inside the consuming kernel (see epit.cpp): feed_forward.0 inside the fused feed-forward by default; the attention norms inside the
  // q | k | v projection only with LFSR_LN_FUSE=2 (measured slower: 1464 against 143 + 795 us for SpaTrans at 32 patches); LFSR_LN_FUSE=0: all norms as launches
  const char* rgs = lfsr_sel("LFSR_ROWGEMM");
  const bool rowgemm_f32 = (rgs && (rgs[0] == 'f' || rgs[0] == '1')) || lfsr_arith_f32();
  // (late round 2) on the three-term bf16 row-GEMM with 128-column panels the fused attention norms DO pay (1636 -> 1680 patches/s): default there; LFSR_LN_FUSE=1 keeps the LayerNorm launches
  const bool ln_fuse = !(lf && lf[0] == '0'), ln_fuse_qkv = lf ? lf[0] == '2' : !rowgemm_f32;
  for (int b = 0; b < c->nlayer; ++b) {
    // ---- AngTrans (LFT.py:233-246): tokens = the A*A views at one (y, x); E = 64, 8 heads of 8, no mask -----------
    std::string an = "altblock." + std::to_string(b) + ".ang_trans.";
    float* a_out = (cur == Pb) ? Qb : Pb;
    const float* Wa = P.w(an + "attention.in_proj_weight");
    // q | k from LayerNorm(token + PE), v from the raw token (LFSR_LN_FUSE=2: one launch)
    rc = ln_fuse_qkv ? lfsr_rowgemm_ln_launch(cur, 64, 0, 64, Wa, P.w(an + "norm.weight"), P.w(an + "norm.bias"), 1e-5f, 128, APE, 64, AA, HW, T, 128, 0, C1, 64, 0, 128,
                                          npix, 192, lfsr_stream(stream))
                 : LFSR_E_ARG;
    if (rc == LFSR_E_ARG) {
      RC(lfsr_layernorm_fwd(cur, 64, 0, APE, 64, AA, HW, P.w(an + "norm.weight"), P.w(an + "norm.bias"), N64, 64, 0, npix, 64, 1e-5f, stream));
      RC(lfsr_linear_fwd(N64, 64, 0, 64, Wa, nullptr, nullptr, 0, 0, T, 128, 0, npix, 128, 1.0f, stream));                 // q | k
      RC(lfsr_linear_fwd(cur, 64, 0, 64, Wa + 128 * 64, nullptr, nullptr, 0, 0, C1, 64, 0, npix, 64, 1.0f, stream));        // v from the raw token
    } else if (rc) return rc;
    RC(lfsr_window_attn_fwd(T, 128, 0, T, 128, 64, C1, 64, 0, C2, 64, 0, 8, 8, B, h, w, (long long)AA * HW, w, 1,
                            AA, 1, HW, 0, AA, AA, 0, 1, 0, stream));
    RC(lfsr_linear_fwd(C2, 64, 0, 64, P.w(an + "attention.out_proj.weight"), nullptr, cur, 64, 0, C1, 64, 0, npix, 64, 1.0f, stream));     // + token
    const float *afg = P.w(an + "feed_forward.0.weight"), *afb = P.w(an + "feed_forward.0.bias");
    rc = (ln_fuse && !no_ffn_fused) ? lfsr_ffn_ln_launch(C1, 64, 0, afg, afb, 1e-5f, P.w(an + "feed_forward.1.weight"), P.w(an + "feed_forward.4.weight"), C1, 64, 0,
                                                         a_out, 64, 0, npix, 64, 128, 64, 0.0f, lfsr_stream(stream), presplit ? P.packed + c->ffn_split_ang[b] : nullptr)
                                    : LFSR_E_ARG;
    if (rc == LFSR_E_ARG) {
      RC(lfsr_layernorm_fwd(C1, 64, 0, nullptr, 0, 0, 1, afg, afb, N64, 64, 0, npix, 64, 1e-5f, stream));
      if (no_ffn_fused) {
        RC(lfsr_linear_fwd(N64, 64, 0, 64, P.w(an + "feed_forward.1.weight"), nullptr, nullptr, 0, 0, T, 128, 0, npix, 128, 0.0f, stream));     // ReLU
        RC(lfsr_linear_fwd(T, 128, 0, 128, P.w(an + "feed_forward.4.weight"), nullptr, C1, 64, 0, a_out, 64, 0, npix, 64, 1.0f, stream));
      } else {
        RC(lfsr_ffn_fwd(N64, 64, 0, P.w(an + "feed_forward.1.weight"), P.w(an + "feed_forward.4.weight"), C1, 64, 0, a_out, 64, 0, npix, 64, 128, 64, 0.0f, stream));
      }
    } else if (rc) return rc;
    // ---- SpaTrans (LFT.py:188-203): tokens = the h*w positions of one view; E = 128, 8 heads of 16, 5x5 window --------
    std::string sp = "altblock." + std::to_string(b) + ".spa_trans.";
    float* s_out = (a_out == Pb) ? Qb : Pb;
    RC(lfsr_conv3x3_fwd(a_out, 64, 0, P.w(sp + "MLP.weight#lo"), T, 128, 0, nullptr, 0, 0, nullptr, 0, 0, nimg, h, w, 1.0f, stream));   // unfold + MLP (tokens),
    RC(lfsr_conv3x3_fwd(a_out, 64, 0, P.w(sp + "MLP.weight#hi"), T, 128, 64, nullptr, 0, 0, nullptr, 0, 0, nimg, h, w, 1.0f, stream));  // as two 64-output convs
    RC(lfsr_conv3x3_n_fwd(SPOS, 64, 0, P.w(sp + "MLP.weight"), SPE, 128, 0, 1, h, w, 128, 1.0f, stream));              // same embedding of the PE map
    const float* Ws = P.w(sp + "attention.in_proj_weight");
    rc = ln_fuse_qkv ? lfsr_rowgemm_ln_launch(T, 128, 0, 128, Ws, P.w(sp + "norm.weight"), P.w(sp + "norm.bias"), 1e-5f, 256, SPE, 128, HW, 1, QK, 256, 0, V, 128, 0, 256,
                                          npix, 384, lfsr_stream(stream))
                 : LFSR_E_ARG;
    if (rc == LFSR_E_ARG) {
      RC(lfsr_layernorm_fwd(T, 128, 0, SPE, 128, HW, 1, P.w(sp + "norm.weight"), P.w(sp + "norm.bias"), TN, 128, 0, npix, 128, 1e-5f, stream));
      RC(lfsr_linear_fwd(TN, 128, 0, 128, Ws, nullptr, nullptr, 0, 0, QK, 256, 0, npix, 256, 1.0f, stream));
      RC(lfsr_linear_fwd(T, 128, 0, 128, Ws + 256 * 128, nullptr, nullptr, 0, 0, V, 128, 0, npix, 128, 1.0f, stream));
    } else if (rc) return rc;
    // window [i-2, i+3) x [j-2, min(h, j+3)): the column clamp uses h (LFT.py:168)
    RC(lfsr_window_attn_fwd(QK, 256, 0, QK, 256, 128, V, 128, 0, TN, 128, 0, 8, 16, nimg, 1, 1, HW, 0, 0, h, w, w, 1, 2, 3, 2, 3, h, stream));
    RC(lfsr_linear_fwd(TN, 128, 0, 128, P.w(sp + "attention.out_proj.weight"), nullptr, T, 128, 0, T2, 128, 0, npix, 128, 1.0f, stream));
    const float *sfg = P.w(sp + "feed_forward.0.weight"), *sfb = P.w(sp + "feed_forward.0.bias");
    rc = (ln_fuse && !no_ffn_fused) ? lfsr_ffn_ln_launch(T2, 128, 0, sfg, sfb, 1e-5f, P.w(sp + "feed_forward.1.weight"), P.w(sp + "feed_forward.4.weight"), T2, 128, 0,
                                                         T, 128, 0, npix, 128, 256, 128, 0.0f, lfsr_stream(stream), presplit ? P.packed + c->ffn_split_spa[b] : nullptr)
                                    : LFSR_E_ARG;
    if (rc == LFSR_E_ARG) {
      RC(lfsr_layernorm_fwd(T2, 128, 0, nullptr, 0, 0, 1, sfg, sfb, V, 128, 0, npix, 128, 1e-5f, stream));
      if (no_ffn_fused) {
        RC(lfsr_linear_fwd(V, 128, 0, 128, P.w(sp + "feed_forward.1.weight"), nullptr, nullptr, 0, 0, QK, 256, 0, npix, 256, 0.0f, stream));
        RC(lfsr_linear_fwd(QK, 256, 0, 256, P.w(sp + "feed_forward.4.weight"), nullptr, T2, 128, 0, T, 128, 0, npix, 128, 1.0f, stream));
      } else {
        RC(lfsr_ffn_fwd(V, 128, 0, P.w(sp + "feed_forward.1.weight"), P.w(sp + "feed_forward.4.weight"), T2, 128, 0, T, 128, 0, npix, 128, 256, 128, 0.0f, stream));
      }
    } else if (rc) return rc;
    // Conv3d 1x1x1 128 -> 64 (LFT.py:183-186); the network-level skip (LFT.py:91) rides on the last layer's projection
    const bool last = b == c->nlayer - 1;
    RC(lfsr_linear_fwd(T, 128, 0, 128, P.w(sp + "linear.0.weight"), nullptr, last ? BUF0 : nullptr, 64, 0, s_out, 64, 0, npix, 64, 1.0f, stream));
    cur = s_out;
  }
  if ((c->s == 2 || c->s == 4) && !lfsr_sel("LFSR_NO_UPTAIL")) {
    RC(lfsr_up_tail_fwd(cur, 64, 0, P.w("upsampling.0.weight"), P.w("upsampling.3.weight"), x, out, B, A, h, w, c->s, L, stream));
  } else {
    RC(lfsr_upsample_ps_fwd(cur, 64, 0, P.w("upsampling.0.weight"), HR, B, A, h, w, c->s, stream));
    RC(lfsr_hr_tail_fwd(HR, P.w("upsampling.3.weight"), x, out, B, A, h, w, c->s, L, stream));
  }
#undef RC
  return LFSR_OK;
}

}  // extern "C"
