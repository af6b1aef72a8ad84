// Host driver for the EPIT forward (get_model.forward, model/SR/EPIT.py:51-71; AltFilter :144-161; BasicTrans :110-128)
// on VCL buffers.  Tokens are VCL pixels; the horizontal / vertical EPI passes differ only in the strides handed to the
// attention kernel, so none of the reference's six `rearrange` copies per AltFilter exists here.
#include <stdlib.h>

#include "param_table.h"

struct lfsr_epit {
  int A, s, nblk;
  LfsrParamTable P;
  std::vector<size_t> ffn_split;      // per block: offset (floats) of the feed-forward weights' pre-split bf16 image in the packed buffer (ffn_b3.hip)
  bool finalized = false;
};

extern "C" {

int lfsr_epit_create(lfsr_epit** out, int A, int scale, int n_block, int channels) {
  if (!out || A <= 0 || A > 15 || scale < 2 || scale > 4 || n_block <= 0 || channels != 64) return LFSR_E_ARG;
  lfsr_epit* c = new lfsr_epit();
  c->A = A; c->s = scale; c->nblk = n_block;
  LfsrParamTable& P = c->P;
  P.add("conv_init0.0.weight", 64, 1, 9, 0, 0, true);
  for (int i : {0, 2, 4}) P.add("conv_init." + std::to_string(i) + ".weight", 64, 64, 9);
  for (int b = 0; b < n_block; ++b) {
    std::string p = "altblock." + std::to_string(b) + ".";
    std::string e = p + "epi_trans.";
    P.add(e + "linear_in.weight", 128, 64, 1);
    P.add(e + "norm.weight", 128, 1, 1, 0, 0, true);
    P.add(e + "norm.bias", 128, 1, 1, 0, 0, true);
    P.add(e + "attention.in_proj_weight", 384, 128, 1);
    P.add(e + "attention.out_proj.weight", 128, 128, 1);
    P.add(e + "feed_forward.0.weight", 128, 1, 1, 0, 0, true);
    P.add(e + "feed_forward.0.bias", 128, 1, 1, 0, 0, true);
    P.add(e + "feed_forward.1.weight", 256, 128, 1);
    P.add(e + "feed_forward.4.weight", 128, 256, 1);
    P.add(e + "linear_out.weight", 64, 128, 1);
    for (int i : {0, 2, 4}) P.add(p + "conv." + std::to_string(i) + ".weight", 64, 64, 9);
    c->ffn_split.push_back(P.reserve((lfsr_ffn_b3_presplit_bytes(128, 256, 128) + 3) / 4));
  }
  P.add("upsampling.0.weight", 64 * scale * scale, 64, 1, 1, 64);   // PixelShuffle order folded into the packing
  P.add("upsampling.3.weight", 1, 64, 9, 0, 0, true);
  *out = c;
  return LFSR_OK;
}

void lfsr_epit_destroy(lfsr_epit* c) { delete c; }
size_t lfsr_epit_packed_bytes(const lfsr_epit* c) { return c ? c->P.packed_floats * sizeof(float) : 0; }
int lfsr_epit_set_packed(lfsr_epit* c, void* packed, size_t bytes) { if (!c) return LFSR_E_ARG; c->finalized = false; return c->P.set_packed(packed, bytes); }
int lfsr_epit_load_param(lfsr_epit* c, const char* key, const float* data, size_t numel, void* stream) {
  if (!c) return LFSR_E_ARG;
  c->finalized = false;
  return c->P.load(key, data, numel, stream);
}
int lfsr_epit_finalize(lfsr_epit* c, void* stream) {
  if (!c || !c->P.packed || !c->P.all_loaded()) return LFSR_E_ARG;
  // the feed-forward weights of every block, split once into their three bf16 planes in the fused kernel's LDS chunk order
  for (int b = 0; b < c->nblk; ++b) {
    const std::string e = "altblock." + std::to_string(b) + ".epi_trans.";
    const int rc = lfsr_ffn_b3_presplit(c->P.w(e + "feed_forward.1.weight"), c->P.w(e + "feed_forward.4.weight"), 128, 256, 128, c->P.packed + c->ffn_split[b], lfsr_stream(stream));
    if (rc) return rc;
  }
  c->finalized = true;
  return LFSR_OK;
}

static void epit_layout(const lfsr_epit* c, int B, int h, int w, size_t off[16], size_t* total) {
  const size_t npix = (size_t)B * c->A * c->A * h * w;
  size_t o = 0;
  auto take = [&](size_t f) { size_t r = o; o += LfsrParamTable::align64(f); return r; };
  for (int i = 0; i < 8; ++i) off[i] = take(npix * 64);         // F0, BUF0, P, Q, MID, Y, C1, C2
  for (int i = 8; i < 12; ++i) off[i] = take(npix * 128);       // T, TN/O, V/FN, T2
  off[12] = take(npix * 256);                                   // QK / FF
  off[13] = take(npix * 64 * c->s * c->s);                      // HR mosaic, channel-last
  *total = o;
}

size_t lfsr_epit_workspace_bytes(const lfsr_epit* c, int B, int h, int w) {
  if (!c || B <= 0 || h <= 0 || w <= 0) return 0;
  size_t off[16], tot;
  epit_layout(c, B, h, w, off, &tot);
  return tot * sizeof(float);
}

int lfsr_epit_forward(lfsr_epit* c, const float* x, float* out, int B, int h, int w, void* workspace, size_t workspace_bytes, void* stream) {
  if (!c || !x || !out || !workspace || B <= 0 || h <= 0 || w <= 0 || !c->finalized || ((uintptr_t)workspace & 15)) return LFSR_E_ARG;
  size_t off[16], tot;
  epit_layout(c, B, h, w, off, &tot);
  if (workspace_bytes < tot * sizeof(float)) return LFSR_E_WS;
  const int A = c->A, AA = A * A, nimg = B * AA, HW = h * w;
  const long long npix = (long long)nimg * HW;
  if (npix * 256 * 4 >= (1LL << 31)) return LFSR_E_ARG;   // every activation tensor < 2 GiB (the q | k rows are the widest): the kernels' 32-bit byte offsets; callers split the batch (capi.py)
  float* ws = (float*)workspace;
  float *F0 = ws + off[0], *BUF0 = ws + off[1], *Pb = ws + off[2], *Qb = ws + off[3], *MID = ws + off[4], *Y = ws + off[5], *C1 = ws + off[6], *C2 = ws + off[7];
  float *T = ws + off[8], *TN = ws + off[9], *V = ws + off[10], *T2 = ws + off[11], *QK = ws + off[12], *HR = ws + off[13];
  const LfsrParamTable& P = c->P;
  const float L = 0.2f;   // LeakyReLU(0.2), EPIT.py:27-31,138-140
  int rc;
#define RC(call) do { rc = (call); if (rc) return rc; } while (0)
  auto conv = [&](const float* in, const std::string& key, float* o, const float* r1, const float* r2, float slope) -> int {
    return lfsr_conv3x3_fwd(in, 64, 0, P.w(key), o, 64, 0, r1, 64, 0, r2, 64, 0, nimg, h, w, slope, stream);
  };
  const char* lf = lfsr_sel("LFSR_LN_FUSE");
  // LayerNorms formed inside the consuming kernel: feed_forward.0 in the fused feed-forward (default: 293 us against 33 + 295 us, 10 launches less per forward);
  // the attention norm inside the q | k | v projection only with LFSR_LN_FUSE=2 -- measured SLOWER (334 us against 33 + 209 us: each of the four q | k column
  // panels repeats the norm of its row tile, and 384 x 128 fp32 weights do not fit one block's LDS); LFSR_LN_FUSE=0: every norm as its own launch
  const char* rgs = lfsr_sel("LFSR_ROWGEMM");
  const bool rowgemm_f32 = (rgs && (rgs[0] == 'f' || rgs[0] == '1')) || lfsr_arith_f32();
  // (late round 2) on the three-term bf16 row-GEMM with 128-column panels the fused attention norm DOES pay (818 -> 831 patches/s): default there; LFSR_LN_FUSE=1 keeps the LayerNorm launch
  const bool ln_fuse = !(lf && lf[0] == '0'), ln_fuse_qkv = lf ? lf[0] == '2' : !rowgemm_f32, no_ffn_fused = lfsr_sel("LFSR_NO_FFN_FUSED") != nullptr;
  // BasicTrans.forward (EPIT.py:110-128) over all sequences of one pass
  const char* psel = lfsr_sel("LFSR_FFN_PRESPLIT");
  const bool presplit = !(psel && psel[0] == '0');      // LFSR_FFN_PRESPLIT=0: the kernel splits the weight chunks itself (A/B runs)
  auto trans = [&](const float* X, const std::string& e, int vertical, float* Yo, int blk) -> int {
    int r;
    if ((r = lfsr_linear_fwd(X, 64, 0, 64, P.w(e + "linear_in.weight"), nullptr, nullptr, 0, 0, T, 128, 0, npix, 128, 1.0f, stream))) return r;
    const float* Win = P.w(e + "attention.in_proj_weight");
    // q | k from LayerNorm(t), v from t (LFSR_LN_FUSE=2: one launch, the norm formed on the staged rows)
    r = ln_fuse_qkv ? lfsr_rowgemm_ln_launch(T, 128, 0, 128, Win, P.w(e + "norm.weight"), P.w(e + "norm.bias"), 1e-5f, 256, nullptr, 0, 1, 1, QK, 256, 0, V, 128, 0, 256,
                                         npix, 384, lfsr_stream(stream))
                : LFSR_E_ARG;
    if (r == LFSR_E_ARG) {
      if ((r = lfsr_layernorm_fwd(T, 128, 0, nullptr, 0, 0, 1, P.w(e + "norm.weight"), P.w(e + "norm.bias"), TN, 128, 0, npix, 128, 1e-5f, stream))) return r;
      if ((r = lfsr_linear_fwd(TN, 128, 0, 128, Win, nullptr, nullptr, 0, 0, QK, 256, 0, npix, 256, 1.0f, stream))) return r;              // q | k from LN(t)
      r = lfsr_linear_fwd(T, 128, 0, 128, Win + 256 * 128, nullptr, nullptr, 0, 0, V, 128, 0, npix, 128, 1.0f, stream);                  // v from t
    }
    if (r) return r;
    // mask_field = [2A, 11] (EPIT.py:147): all angular positions, spatial window [j-5, j+6)
    if (!vertical) r = lfsr_window_attn_fwd(QK, 256, 0, QK, 256, 128, V, 128, 0, TN, 128, 0, 8, 16, B, A, w, (long long)AA * HW, HW, 1,
                                            A, h, (long long)A * HW, w, A, A, 5, 6, 0, stream);      // sequence (b, v, x); tokens (u, y)
    else r = lfsr_window_attn_fwd(QK, 256, 0, QK, 256, 128, V, 128, 0, TN, 128, 0, 8, 16, B, A, h, (long long)AA * HW, (long long)A * HW, w,
                                  A, w, HW, 1, A, A, 5, 6, 0, stream);                               // sequence (b, u, y); tokens (v, x)
    if (r) return r;
    if ((r = lfsr_linear_fwd(TN, 128, 0, 128, P.w(e + "attention.out_proj.weight"), nullptr, T, 128, 0, T2, 128, 0, npix, 128, 1.0f, stream))) return r;
    const float *fg = P.w(e + "feed_forward.0.weight"), *fb = P.w(e + "feed_forward.0.bias");
    r = (ln_fuse && !no_ffn_fused) ? lfsr_ffn_ln_launch(T2, 128, 0, fg, fb, 1e-5f, P.w(e + "feed_forward.1.weight"), P.w(e + "feed_forward.4.weight"), T2, 128, 0, T, 128, 0,
                                                        npix, 128, 256, 128, 0.0f, lfsr_stream(stream), presplit ? P.packed + c->ffn_split[blk] : nullptr)
                                   : LFSR_E_ARG;
    if (r == LFSR_E_ARG) {
      if ((r = lfsr_layernorm_fwd(T2, 128, 0, nullptr, 0, 0, 1, fg, fb, V, 128, 0, npix, 128, 1e-5f, stream))) return r;
      if (no_ffn_fused) {   // two-launch form (A/B runs): the hidden activations go through HBM
        if ((r = lfsr_linear_fwd(V, 128, 0, 128, P.w(e + "feed_forward.1.weight"), nullptr, nullptr, 0, 0, QK, 256, 0, npix, 256, 0.0f, stream))) return r;   // ReLU
        r = lfsr_linear_fwd(QK, 256, 0, 256, P.w(e + "feed_forward.4.weight"), nullptr, T2, 128, 0, T, 128, 0, npix, 128, 1.0f, stream);
      } else {
        r = lfsr_ffn_fwd(V, 128, 0, P.w(e + "feed_forward.1.weight"), P.w(e + "feed_forward.4.weight"), T2, 128, 0, T, 128, 0, npix, 128, 256, 128, 0.0f, stream);
      }
    }
    if (r) return r;
    return lfsr_linear_fwd(T, 128, 0, 128, P.w(e + "linear_out.weight"), nullptr, nullptr, 0, 0, Yo, 64, 0, npix, 64, 1.0f, stream);
  };

  RC(lfsr_initconv_fwd(x, P.w("conv_init0.0.weight"), F0, 64, 0, B, A, h, w, stream));
  RC(conv(F0, "conv_init.0.weight", C1, nullptr, nullptr, L));
  RC(conv(C1, "conv_init.2.weight", C2, nullptr, nullptr, L));
  RC(conv(C2, "conv_init.4.weight", BUF0, F0, nullptr, L));                 // lrelu(conv) + buffer   (EPIT.py:63)
  const float* cur = BUF0;
  for (int b = 0; b < c->nblk; ++b) {
    std::string p = "altblock." + std::to_string(b) + ".";
    float* o = (cur == Pb) ? Qb : Pb;
    const bool last = b == c->nblk - 1;
    for (int vert = 0; vert < 2; ++vert) {
      const float* in = vert ? MID : cur;
      RC(trans(in, p + "epi_trans.", vert, Y, b));
      RC(conv(Y, p + "conv.0.weight", C1, nullptr, nullptr, L));
      RC(conv(C1, p + "conv.2.weight", C2, nullptr, nullptr, L));
      // + shortcut (the block INPUT both times, EPIT.py:153,159); the network-level skip (:66) rides on the very last conv
      RC(conv(C2, p + "conv.4.weight", vert ? o : MID, cur, (vert && last) ? BUF0 : nullptr, 1.0f));
    }
    cur = o;
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
