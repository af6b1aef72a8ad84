// Token-wise and attention kernels of the transformer models (model/SR/EPIT.py:74-128, model/SR/LFT.py:133-246) on the
// VCL layout, plus their shared up-sampling tail.  Tokens ARE VCL pixels: every linear / LayerNorm / FFN is row-wise
// over (pixels, C) and never needs the reference's `rearrange` copies; only attention groups tokens into sequences, and
// it does so by strided addressing (sequence base + t1*stride1 + t2*stride2).
//
//  * k_layernorm      nn.LayerNorm(C) over rows (optionally of x + pe), fp32, wave-shuffle reductions.
//  * k_window_attn    softmax(q k^T / sqrt(hd) + window mask) v for `nheads` heads, one thread per (token, head), head
//                     fastest so a wave touches whole 512-B token rows.  The -inf mask the reference rebuilds on the CPU
//                     at every forward (EPIT.py:93-108 at :112; LFT.py:161-174 at :189) is a box predicate evaluated in
//                     registers: only the valid keys are ever visited (55 of 160 for EPIT, 25 of 1024 for LFT).
//  * k_hr_tail        LeakyReLU'd HR features -> 3x3 conv 64->1 over the whole HR mosaic (zero pad 1, EPIT.py:47-48 /
//                     LFT.py:55-56) + per-view bicubic skip (EPIT.py:164-169, LFT.py:263-273; a = -0.75, clamped borders).
#include <stdlib.h>

#include "gemm_gather_kernel.h"
#include "lfsr_internal.h"

namespace {

// ---------------- LayerNorm ----------------------------------------------------------------------------------------
template <int C>
__global__ __launch_bounds__(256) void k_layernorm(const float* __restrict__ x, int x_stride, int x_choff,
                                                  const float* __restrict__ pe, int pe_stride, long long pe_rows, long long pe_div,
                                                  const float* __restrict__ g, const float* __restrict__ b,
                                                  float* __restrict__ y, int y_stride, int y_choff, long long M, float eps) {
  constexpr int LPR = C / 4;            // lanes per row
  constexpr int RPB = 256 / LPR;        // rows per block
  const int lr = threadIdx.x % LPR, rr = threadIdx.x / LPR;
  const float4 gv = *reinterpret_cast<const float4*>(g + lr * 4);
  const float4 bv = *reinterpret_cast<const float4*>(b + lr * 4);
  for (long long row = (long long)blockIdx.x * RPB + rr; row < M; row += (long long)gridDim.x * RPB) {
    float4 v = *reinterpret_cast<const float4*>(x + row * x_stride + x_choff + lr * 4);
    if (pe) {
      float4 q = *reinterpret_cast<const float4*>(pe + ((row / pe_div) % pe_rows) * pe_stride + lr * 4);
      v.x += q.x; v.y += q.y; v.z += q.z; v.w += q.w;
    }
    float s = v.x + v.y + v.z + v.w;
#pragma unroll
    for (int o = LPR / 2; o > 0; o >>= 1) s += __shfl_xor(s, o, LPR);
    const float mu = s * (1.0f / C);
    float dx = v.x - mu, dy = v.y - mu, dz = v.z - mu, dw = v.w - mu;
    float q2 = dx * dx + dy * dy + dz * dz + dw * dw;
#pragma unroll
    for (int o = LPR / 2; o > 0; o >>= 1) q2 += __shfl_xor(q2, o, LPR);
    const float rstd = 1.0f / sqrtf(q2 * (1.0f / C) + eps);
    float4 o4 = make_float4(dx * rstd * gv.x + bv.x, dy * rstd * gv.y + bv.y, dz * rstd * gv.z + bv.z, dw * rstd * gv.w + bv.w);
    *reinterpret_cast<float4*>(y + row * y_stride + y_choff + lr * 4) = o4;
  }
}

// ---------------- windowed multi-head attention -----------------------------------------------------------------------
struct AttnArgs {
  const float* Q; int q_stride, q_choff;
  const float* K; int k_stride, k_choff;
  const float* V; int v_stride, v_choff;
  float* O; int o_stride, o_choff;
  int nheads;
  int ns0, ns1, ns2; long long bs0, bs1, bs2;   // sequence (s0,s1,s2) -> base pixel
  int n1, n2; long long st1, st2;               // token grid (t1,t2) and pixel strides
  int l1, r1, l2, r2;                           // key window [t-l, t+r) per dim, clipped to the grid
  int clip2;                                    // upper clip of dim 2 (LFT.py:168 clamps the column window with h, not w)
  float scale;
  float scale2;                                 // scale * log2(e): the LDS-tiled kernels run their softmax in base 2
  long long total;                              // nseq * n1 * n2 * nheads
  int loop_stage;                               // 1: the LDS-tiled kernel stages its keys with the plain loop (A/B: LFSR_ATTN_ANG=loop)
};

template <int HD>
__global__ __launch_bounds__(256) void k_window_attn(AttnArgs p) {
  const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= p.total) return;
  const int head = (int)(idx % p.nheads);
  long long t = idx / p.nheads;
  const int t2 = (int)(t % p.n2); t /= p.n2;
  const int t1 = (int)(t % p.n1); t /= p.n1;
  const int s2 = (int)(t % p.ns2); t /= p.ns2;
  const int s1 = (int)(t % p.ns1);
  const int s0 = (int)(t / p.ns1);
  const long long base = s0 * p.bs0 + s1 * p.bs1 + s2 * p.bs2;
  const long long qpix = base + t1 * p.st1 + t2 * p.st2;
  float q[HD], acc[HD];
  {
    const float4* qp = reinterpret_cast<const float4*>(p.Q + qpix * p.q_stride + p.q_choff + head * HD);
#pragma unroll
    for (int i = 0; i < HD / 4; ++i) {
      float4 v = qp[i];
      q[4 * i] = v.x * p.scale; q[4 * i + 1] = v.y * p.scale; q[4 * i + 2] = v.z * p.scale; q[4 * i + 3] = v.w * p.scale;
    }
  }
#pragma unroll
  for (int i = 0; i < HD; ++i) acc[i] = 0.f;
  float mx = -INFINITY, den = 0.f;
  const int a0 = max(0, t1 - p.l1), a1 = min(p.n1, t1 + p.r1);
  const int b0 = max(0, t2 - p.l2), b1 = min(min(p.n2, p.clip2), t2 + p.r2);
  for (int k1 = a0; k1 < a1; ++k1) {
    for (int k2 = b0; k2 < b1; ++k2) {
      const long long kpix = base + k1 * p.st1 + k2 * p.st2;
      const float4* kp = reinterpret_cast<const float4*>(p.K + kpix * p.k_stride + p.k_choff + head * HD);
      float s = 0.f;
#pragma unroll
      for (int i = 0; i < HD / 4; ++i) {
        float4 kv = kp[i];
        s = fmaf(q[4 * i], kv.x, s); s = fmaf(q[4 * i + 1], kv.y, s); s = fmaf(q[4 * i + 2], kv.z, s); s = fmaf(q[4 * i + 3], kv.w, s);
      }
      const float mn = fmaxf(mx, s);
      const float corr = expf(mx - mn);      // exp(-inf) = 0 on the first key
      const float pw = expf(s - mn);
      den = den * corr + pw;
      const float4* vp = reinterpret_cast<const float4*>(p.V + kpix * p.v_stride + p.v_choff + head * HD);
#pragma unroll
      for (int i = 0; i < HD / 4; ++i) {
        float4 vv = vp[i];
        acc[4 * i] = fmaf(pw, vv.x, acc[4 * i] * corr);
        acc[4 * i + 1] = fmaf(pw, vv.y, acc[4 * i + 1] * corr);
        acc[4 * i + 2] = fmaf(pw, vv.z, acc[4 * i + 2] * corr);
        acc[4 * i + 3] = fmaf(pw, vv.w, acc[4 * i + 3] * corr);
      }
      mx = mn;
    }
  }
  const float inv = 1.0f / den;      // an empty window gives NaN exactly like softmax over an all -inf row
  float4* op = reinterpret_cast<float4*>(p.O + qpix * p.o_stride + p.o_choff + head * HD);
#pragma unroll
  for (int i = 0; i < HD / 4; ++i) op[i] = make_float4(acc[4 * i] * inv, acc[4 * i + 1] * inv, acc[4 * i + 2] * inv, acc[4 * i + 3] * inv);
}


// ---- full attention over short sequences, heads of 8: LFT's angular transformer (the A*A = 25 views of one pixel; LFT.py:236-241) --------------------------------
// k_window_attn_lds<8, 8> gave one thread a (query, head) and an online softmax: per key two 16-B LDS reads of k, two of v, a rescale of the eight accumulators and two
// exponentials -- 243 us per launch at the LFT scene geometry with the LDS port (~150-190 us of reads) and the VALU (~140 us) both near their limits, against 150-170 us
// of HBM time.  Here a thread owns TWO queries of one head, so every k / v read serves two, and the softmax is two-pass over the N1 scores kept in registers (no rescale,
// one exponential per score); a block takes AP sequences (adjacent pixels: their token rows are adjacent in memory).
template <int N1, int AP, int NTH>
__global__ __launch_bounds__(NTH) __attribute__((amdgpu_waves_per_eu(4, 4))) void k_ang_attn_pair(AttnArgs p, long long nseq) {      // (16 waves per CU: 128 registers)
  constexpr int NQP = (N1 + 1) / 2;                  // query pairs
  constexpr int HS = 20, TS = 8 * HS + 4;            // the staged layout of k_window_attn_lds<8, 8>: (key, head) = k8 | v8 | pad 4; a key's eight heads + 4
  constexpr int SEQF = N1 * TS;                      // floats per staged sequence
  extern __shared__ __attribute__((aligned(16))) float sang[];
  const int tid = threadIdx.x;
  // ---- stage K | V of the block's AP sequences: item = (sequence a, key, head, 16-B chunk c of k8 | v8); all of a thread's loads before its first LDS store
  constexpr int NITEM = AP * N1 * 8 * 4;
  constexpr int NIT = (NITEM + NTH - 1) / NTH;
  static_assert(NTH >= AP * ((N1 + 1) / 2) * 8, "one thread per (sequence, head, query pair)");
  float4 sv[NIT];
  // the base pixels of the block's AP sequences, decoded once (three 64-bit divisions each) and shared through LDS
  long long* const sbase = reinterpret_cast<long long*>(sang + AP * SEQF);
  if (tid < AP) {
    long long t = (long long)blockIdx.x * AP + tid;
    const long long s2 = t % p.ns2; t /= p.ns2;
    const long long s1 = t % p.ns1;
    const long long s0 = t / p.ns1;
    sbase[tid] = s0 * p.bs0 + s1 * p.bs1 + s2 * p.bs2;
  }
  __syncthreads();
  auto seq_base = [&](int a) -> long long { return sbase[a]; };
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    const int i = tid + NTH * it;
    sv[it] = make_float4(0.f, 0.f, 0.f, 0.f);
    const int c = i & 3, h = (i >> 2) & 7, r = i >> 5, key = r % N1, a = r / N1;
    const long long sq = (long long)blockIdx.x * AP + a;
    if (i < NITEM && sq < nseq) {
      const long long pix = seq_base(a) + key * p.st1;
      const float* src = c < 2 ? p.K + pix * p.k_stride + p.k_choff + h * 8 + c * 4 : p.V + pix * p.v_stride + p.v_choff + h * 8 + (c - 2) * 4;
      sv[it] = *reinterpret_cast<const float4*>(src);
    }
  }
  // this thread's queries: (sequence a, head h, tokens 2 qp and 2 qp + 1)
  const int qp = tid % NQP, h = (tid / NQP) & 7, a = tid / (NQP * 8);
  const long long sq = (long long)blockIdx.x * AP + a;
  const bool act = a < AP && sq < nseq;
  const bool two = 2 * qp + 1 < N1;
  float q0[8], q1[8];
  long long qpix = 0;
  if (act) {
    qpix = seq_base(a) + 2 * qp * p.st1;
    const float4* qa = reinterpret_cast<const float4*>(p.Q + qpix * p.q_stride + p.q_choff + h * 8);
    const float4* qb = reinterpret_cast<const float4*>(p.Q + (qpix + (two ? p.st1 : 0)) * p.q_stride + p.q_choff + h * 8);
    const float4 a0 = qa[0], a1 = qa[1], b0 = qb[0], b1 = qb[1];
    q0[0] = a0.x; q0[1] = a0.y; q0[2] = a0.z; q0[3] = a0.w; q0[4] = a1.x; q0[5] = a1.y; q0[6] = a1.z; q0[7] = a1.w;
    q1[0] = b0.x; q1[1] = b0.y; q1[2] = b0.z; q1[3] = b0.w; q1[4] = b1.x; q1[5] = b1.y; q1[6] = b1.z; q1[7] = b1.w;
#pragma unroll
    for (int i = 0; i < 8; ++i) { q0[i] *= p.scale2; q1[i] *= p.scale2; }
  }
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    const int i = tid + NTH * it;
    if (i < NITEM) {
      const int c = i & 3, hh = (i >> 2) & 7, r = i >> 5, key = r % N1, aa = r / N1;
      *reinterpret_cast<float4*>(sang + aa * SEQF + key * TS + hh * HS + c * 4) = sv[it];
    }
  }
  __syncthreads();
  if (!act) return;
  const float* kvb = sang + a * SEQF + h * HS;
  // pass 1: the scores of both queries (the dot products in the order of k_window_attn_lds: bit-identical scores), their maxima
  float s0[N1], s1[N1];
  float m0 = -INFINITY, m1 = -INFINITY;
#pragma unroll
  for (int key = 0; key < N1; ++key) {
    const float4 ka = *reinterpret_cast<const float4*>(kvb + key * TS), kb = *reinterpret_cast<const float4*>(kvb + key * TS + 4);
    float x = 0.f, y = 0.f;
    x = fmaf(q0[0], ka.x, x); x = fmaf(q0[1], ka.y, x); x = fmaf(q0[2], ka.z, x); x = fmaf(q0[3], ka.w, x);
    x = fmaf(q0[4], kb.x, x); x = fmaf(q0[5], kb.y, x); x = fmaf(q0[6], kb.z, x); x = fmaf(q0[7], kb.w, x);
    y = fmaf(q1[0], ka.x, y); y = fmaf(q1[1], ka.y, y); y = fmaf(q1[2], ka.z, y); y = fmaf(q1[3], ka.w, y);
    y = fmaf(q1[4], kb.x, y); y = fmaf(q1[5], kb.y, y); y = fmaf(q1[6], kb.z, y); y = fmaf(q1[7], kb.w, y);
    s0[key] = x; s1[key] = y;
    if (key % 3 == 2 || key == N1 - 1)
      asm volatile("" : "+v"(s0[key]), "+v"(s1[key]), "+v"(s0[key > 0 ? key - 1 : 0]), "+v"(s1[key > 0 ? key - 1 : 0]), "+v"(s0[key > 1 ? key - 2 : 0]), "+v"(s1[key > 1 ? key - 2 : 0]) : : "memory");      // (three keys at a time: both scores of a key before the next group's: left alone the scheduler runs the pass per query and keeps -- spills -- every k row in between)
    m0 = fmaxf(m0, x); m1 = fmaxf(m1, y);
  }
  // pass 2: base-2 softmax (q carries log2 e) and P V
  float acc0[8], acc1[8], den0 = 0.f, den1 = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i) { acc0[i] = 0.f; acc1[i] = 0.f; }
#pragma unroll
  for (int key = 0; key < N1; ++key) {
    const float4 va = *reinterpret_cast<const float4*>(kvb + key * TS + 8), vb = *reinterpret_cast<const float4*>(kvb + key * TS + 12);
    const float w0 = __builtin_amdgcn_exp2f(s0[key] - m0), w1 = __builtin_amdgcn_exp2f(s1[key] - m1);
    den0 += w0; den1 += w1;
    acc0[0] = fmaf(w0, va.x, acc0[0]); acc0[1] = fmaf(w0, va.y, acc0[1]); acc0[2] = fmaf(w0, va.z, acc0[2]); acc0[3] = fmaf(w0, va.w, acc0[3]);
    acc0[4] = fmaf(w0, vb.x, acc0[4]); acc0[5] = fmaf(w0, vb.y, acc0[5]); acc0[6] = fmaf(w0, vb.z, acc0[6]); acc0[7] = fmaf(w0, vb.w, acc0[7]);
    acc1[0] = fmaf(w1, va.x, acc1[0]); acc1[1] = fmaf(w1, va.y, acc1[1]); acc1[2] = fmaf(w1, va.z, acc1[2]); acc1[3] = fmaf(w1, va.w, acc1[3]);
    acc1[4] = fmaf(w1, vb.x, acc1[4]); acc1[5] = fmaf(w1, vb.y, acc1[5]); acc1[6] = fmaf(w1, vb.z, acc1[6]); acc1[7] = fmaf(w1, vb.w, acc1[7]);
    if (key % 3 == 2 || key == N1 - 1)
    asm volatile("" : "+v"(acc0[0]), "+v"(acc0[1]), "+v"(acc0[2]), "+v"(acc0[3]), "+v"(acc0[4]), "+v"(acc0[5]), "+v"(acc0[6]), "+v"(acc0[7]),
                      "+v"(acc1[0]), "+v"(acc1[1]), "+v"(acc1[2]), "+v"(acc1[3]), "+v"(acc1[4]), "+v"(acc1[5]), "+v"(acc1[6]), "+v"(acc1[7]) : : "memory");      // (as above: every update of this key here)
  }
  const float i0 = 1.0f / den0, i1 = 1.0f / den1;
  float4* oa = reinterpret_cast<float4*>(p.O + qpix * p.o_stride + p.o_choff + h * 8);
  oa[0] = make_float4(acc0[0] * i0, acc0[1] * i0, acc0[2] * i0, acc0[3] * i0); oa[1] = make_float4(acc0[4] * i0, acc0[5] * i0, acc0[6] * i0, acc0[7] * i0);
  if (two) {
    float4* ob = reinterpret_cast<float4*>(p.O + (qpix + p.st1) * p.o_stride + p.o_choff + h * 8);
    ob[0] = make_float4(acc1[0] * i1, acc1[1] * i1, acc1[2] * i1, acc1[3] * i1); ob[1] = make_float4(acc1[4] * i1, acc1[5] * i1, acc1[6] * i1, acc1[7] * i1);
  }
}


// ---- windowed attention, LDS-tiled ------------------------------------------------------------------------------------------
// One block = a tile of query tokens (T1 consecutive t1 rows x all n2 columns of one sequence) x HB heads.  The K and V
// slices (HD floats each per head) of every key the tile can see -- rows [t1_lo - l1, t1_hi + r1) clipped, all columns -- are
// staged in LDS once ([key][head][k16 | v16], token stride padded by 4 floats: ds_read_b128 conflict-free), then one thread per
// (query token, head) walks its window reading LDS.  Against the L1-served k_window_attn this cuts the key traffic per query
// from ~55 x 128 B of L1 reads to LDS reads, and HBM/L2 traffic to one read of K and V per tile.
template <int HD, int HB>
__global__ __launch_bounds__(1024) void k_window_attn_lds(AttnArgs p, int T1, int ntile1) {
  extern __shared__ __attribute__((aligned(16))) float skv[];
  constexpr int HS = 2 * HD + (HD == 8 ? 4 : 0);      // floats per staged (key, head): k | v; heads of 8 padded so that the eight heads of a key sit in disjoint banks
  constexpr int TS = HB * HS + 4;                     // floats per staged key token
  const int hb = blockIdx.y * HB;                     // first head of this block
  int t = blockIdx.x;
  const int tile1 = t % ntile1; t /= ntile1;
  const int s2 = t % p.ns2; t /= p.ns2;
  const int s1 = t % p.ns1;
  const int s0 = t / p.ns1;
  const long long base = s0 * p.bs0 + s1 * p.bs1 + s2 * p.bs2;
  const int q_lo = tile1 * T1, q_hi = min(p.n1, q_lo + T1);
  const int k_lo = max(0, q_lo - p.l1), k_hi = min(p.n1, q_hi - 1 + p.r1);
  const int nkey = (k_hi - k_lo) * p.n2;
  // stage K | V : one 16-B chunk per thread-iteration; chunk c of a (key, head): c < HD/4 -> K, else V.  (Issuing four loads before their four LDS stores was
  // measured: 940 -> 1166 us for LFT's spatial attention -- the plain loop is already pipelined by the compiler)
  constexpr int CPT = 2 * HD / 4;                     // chunks per (key, head)
  if (HD == 8 && !p.loop_stage && nkey * HB * CPT <= 4 * (int)blockDim.x) {
    // short sequences (LFT's angular attention: 25 keys x 8 heads x 4 chunks = 800 items for 256 threads): all of a thread's loads before its first LDS store --
    // as a plain loop the four items of a thread are four serial memory round trips, most of this small block's life (round 3)
    float4 sv[4];
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const int i = threadIdx.x + it * blockDim.x;
      sv[it] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (i < nkey * HB * CPT) {
        int c = i % CPT, r = i / CPT, h = r % HB, key = r / HB;
        int k1 = k_lo + key / p.n2, k2 = key % p.n2;
        long long pix = base + k1 * p.st1 + k2 * p.st2;
        const float* src = c < HD / 4 ? p.K + pix * p.k_stride + p.k_choff + (hb + h) * HD + c * 4
                                      : p.V + pix * p.v_stride + p.v_choff + (hb + h) * HD + (c - HD / 4) * 4;
        sv[it] = *reinterpret_cast<const float4*>(src);
      }
    }
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const int i = threadIdx.x + it * blockDim.x;
      if (i < nkey * HB * CPT) {
        int c = i % CPT, r = i / CPT, h = r % HB, key = r / HB;
        *reinterpret_cast<float4*>(skv + key * TS + h * HS + c * 4) = sv[it];
      }
    }
  } else
  for (int i = threadIdx.x; i < nkey * HB * CPT; i += blockDim.x) {
    int c = i % CPT, r = i / CPT, h = r % HB, key = r / HB;
    int k1 = k_lo + key / p.n2, k2 = key % p.n2;
    long long pix = base + k1 * p.st1 + k2 * p.st2;
    const float* src = c < HD / 4 ? p.K + pix * p.k_stride + p.k_choff + (hb + h) * HD + c * 4
                                  : p.V + pix * p.v_stride + p.v_choff + (hb + h) * HD + (c - HD / 4) * 4;
    *reinterpret_cast<float4*>(skv + key * TS + h * HS + c * 4) = *reinterpret_cast<const float4*>(src);
  }
  __syncthreads();
  const int nq = (q_hi - q_lo) * p.n2;
  for (int it = threadIdx.x; it < nq * HB; it += blockDim.x) {
    const int h = it % HB, qi = it / HB;
    const int t1 = q_lo + qi / p.n2, t2 = qi % p.n2;
    const long long qpix = base + t1 * p.st1 + t2 * p.st2;
    float q[HD], acc[HD];
    {
      const float4* qp = reinterpret_cast<const float4*>(p.Q + qpix * p.q_stride + p.q_choff + (hb + h) * HD);
#pragma unroll
      for (int i = 0; i < HD / 4; ++i) {
        float4 v = qp[i];
        q[4 * i] = v.x * p.scale2; q[4 * i + 1] = v.y * p.scale2; q[4 * i + 2] = v.z * p.scale2; q[4 * i + 3] = v.w * p.scale2;
      }
    }
#pragma unroll
    for (int i = 0; i < HD; ++i) acc[i] = 0.f;
    float mx = -INFINITY, den = 0.f;
    const int a0 = max(0, t1 - p.l1), a1 = min(p.n1, t1 + p.r1);
    const int b0 = max(0, t2 - p.l2), b1 = min(min(p.n2, p.clip2), t2 + p.r2);
    for (int k1 = a0; k1 < a1; ++k1) {
      const float* rowp = skv + ((k1 - k_lo) * p.n2) * TS + h * HS;
      for (int k2 = b0; k2 < b1; ++k2) {
        const float4* kp = reinterpret_cast<const float4*>(rowp + k2 * TS);
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < HD / 4; ++i) {
          float4 kv = kp[i];
          s = fmaf(q[4 * i], kv.x, s); s = fmaf(q[4 * i + 1], kv.y, s); s = fmaf(q[4 * i + 2], kv.z, s); s = fmaf(q[4 * i + 3], kv.w, s);
        }
        const float mn = fmaxf(mx, s);
        const float corr = __builtin_amdgcn_exp2f(mx - mn);   // base-2 softmax (q carries log2 e): raw v_exp_f32, arguments <= 0, exp2(-inf) = 0 on the first key
        const float pw = __builtin_amdgcn_exp2f(s - mn);
        den = den * corr + pw;
#pragma unroll
        for (int i = 0; i < HD / 4; ++i) {
          float4 vv = kp[HD / 4 + i];
          acc[4 * i] = fmaf(pw, vv.x, acc[4 * i] * corr);
          acc[4 * i + 1] = fmaf(pw, vv.y, acc[4 * i + 1] * corr);
          acc[4 * i + 2] = fmaf(pw, vv.z, acc[4 * i + 2] * corr);
          acc[4 * i + 3] = fmaf(pw, vv.w, acc[4 * i + 3] * corr);
        }
        mx = mn;
      }
    }
    const float inv = 1.0f / den;
    float4* op = reinterpret_cast<float4*>(p.O + qpix * p.o_stride + p.o_choff + (hb + h) * HD);
#pragma unroll
    for (int i = 0; i < HD / 4; ++i) op[i] = make_float4(acc[4 * i] * inv, acc[4 * i + 1] * inv, acc[4 * i + 2] * inv, acc[4 * i + 3] * inv);
  }
}

// ---------------- HR tail: conv 3x3 64->1 over the HR mosaic + bicubic skip ------------------------------------------------
__device__ __forceinline__ void cubic_coeffs(float t, float c[4]) {
  const float a = -0.75f;
  float x1 = t + 1.f, x2 = t, x3 = 1.f - t, x4 = 2.f - t;
  c[0] = ((a * x1 - 5.f * a) * x1 + 8.f * a) * x1 - 4.f * a;
  c[1] = ((a + 2.f) * x2 - (a + 3.f)) * x2 * x2 + 1.f;
  c[2] = ((a + 2.f) * x3 - (a + 3.f)) * x3 * x3 + 1.f;
  c[3] = ((a * x4 - 5.f * a) * x4 + 8.f * a) * x4 - 4.f * a;
}

__global__ __launch_bounds__(256) void k_hr_tail(const float* __restrict__ f, const float* __restrict__ w, const float* __restrict__ xlr, float* __restrict__ out,
                                                int B, int A, int h, int wd, int S, float slope) {
  __shared__ float sw[9 * 64];   // [tap][c]
  for (int i = threadIdx.x; i < 9 * 64; i += 256) { int c = i / 9, t = i - c * 9; sw[t * 64 + c] = w[i]; }   // w (1,64,3,3)
  __syncthreads();
  const int Hs = A * h * S, Ws = A * wd * S, Hm = A * h, Wm = A * wd;
  const long long total = (long long)B * Hs * Ws;
  const float rs = 1.0f / (float)S;
  for (long long g = (long long)blockIdx.x * 256 + threadIdx.x; g < total; g += (long long)gridDim.x * 256) {
    const int X = (int)(g % Ws);
    long long t = g / Ws;
    const int Y = (int)(t % Hs);
    const int b = (int)(t / Hs);
    float acc = 0.f;
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
      const int yy = Y + ky - 1;
      if (yy < 0 || yy >= Hs) continue;
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) {
        const int xx = X + kx - 1;
        if (xx < 0 || xx >= Ws) continue;
        const float4* fp = reinterpret_cast<const float4*>(f + (((long long)b * Hs + yy) * Ws + xx) * 64);
        const float* wr = sw + (ky * 3 + kx) * 64;
#pragma unroll 4
        for (int c4 = 0; c4 < 16; ++c4) {
          float4 v = fp[c4];
          v.x = v.x >= 0.f ? v.x : v.x * slope; v.y = v.y >= 0.f ? v.y : v.y * slope;
          v.z = v.z >= 0.f ? v.z : v.z * slope; v.w = v.w >= 0.f ? v.w : v.w * slope;
          acc = fmaf(v.x, wr[4 * c4], acc); acc = fmaf(v.y, wr[4 * c4 + 1], acc); acc = fmaf(v.z, wr[4 * c4 + 2], acc); acc = fmaf(v.w, wr[4 * c4 + 3], acc);
        }
      }
    }
    // per-view bicubic (align_corners=False)
    const int u = Y / (h * S), yl = Y - u * h * S, v = X / (wd * S), xl = X - v * wd * S;
    const float sy = ((float)yl + 0.5f) * rs - 0.5f, sx = ((float)xl + 0.5f) * rs - 0.5f;
    const float fy = floorf(sy), fx = floorf(sx);
    float cy[4], cx[4];
    cubic_coeffs(sy - fy, cy);
    cubic_coeffs(sx - fx, cx);
    const float* img = xlr + (long long)b * Hm * Wm + (long long)(u * h) * Wm + v * wd;
    float up = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      int iy = min(max((int)fy - 1 + i, 0), h - 1);
      float rowv = 0.f;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        int ix = min(max((int)fx - 1 + j, 0), wd - 1);
        rowv = fmaf(cx[j], img[(long long)iy * Wm + ix], rowv);
      }
      up = fmaf(cy[i], rowv, up);
    }
    out[g] = acc + up;
  }
}

// sin of the even columns then cos of the odd columns, concatenated; temperature 10000 (fp64 math, fp32 result)
__device__ __forceinline__ double lft_pe(int pos, int col, int C) {
  const int half = C / 2;
  const int src = col < half ? 2 * col : 2 * (col - half) + 1;       // column of pos/grid that feeds this output column
  const double g = pow(10000.0, 2.0 * (double)(src / 2) / (double)C);
  const double a = (double)pos / g;
  return col < half ? sin(a) : cos(a);
}

__global__ void k_lft_position(float* __restrict__ spa_pe, float* __restrict__ ang_pe, int AA, int h, int w, int C) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < h * w * C) {
    int c = i % C, p = i / C, y = p / w, x = p - y * w;
    spa_pe[i] = (float)((lft_pe(y, c, C) + lft_pe(x, c, C)) / 2.0);
  } else if (i < (h * w + AA) * C) {
    int j = i - h * w * C, c = j % C, a = j / C;
    ang_pe[j] = (float)lft_pe(a, c, C);
  }
}

}  // namespace

extern "C" {

int lfsr_layernorm_fwd(const float* x, int x_stride, int x_choff, const float* pe, int pe_stride, long long pe_rows, long long pe_div, const float* gamma, const float* beta,
                       float* y, int y_stride, int y_choff, long long M, int C, float eps, void* stream) {
  LfsrOpTimer op_t("layernorm", 0, 0, lfsr_stream(stream));
  if (!x || !gamma || !beta || !y || M <= 0 || (C != 64 && C != 128)) return LFSR_E_ARG;
  if ((x_stride | x_choff | y_stride | y_choff) & 3 || (pe && ((pe_stride & 3) || pe_rows <= 0 || pe_div <= 0))) return LFSR_E_ARG;
  const int rpb = 256 / (C / 4);
  unsigned grid = lfsr_blocks(M, rpb);
  if (grid > 256u * 32) grid = 256u * 32;
  if (C == 64) hipLaunchKernelGGL((k_layernorm<64>), dim3(grid), dim3(256), 0, lfsr_stream(stream), x, x_stride, x_choff, pe, pe_stride, pe_rows, pe_div, gamma, beta, y, y_stride, y_choff, M, eps);
  else hipLaunchKernelGGL((k_layernorm<128>), dim3(grid), dim3(256), 0, lfsr_stream(stream), x, x_stride, x_choff, pe, pe_stride, pe_rows, pe_div, gamma, beta, y, y_stride, y_choff, M, eps);
  LFSR_CHECK_LAUNCH();
  return LFSR_OK;
}

int lfsr_window_attn_fwd(const float* q, int q_stride, int q_choff, const float* k, int k_stride, int k_choff, const float* v, int v_stride, int v_choff,
                         float* o, int o_stride, int o_choff, int nheads, int hd,
                         int ns0, int ns1, int ns2, long long bs0, long long bs1, long long bs2,
                         int n1, int n2, long long st1, long long st2, int l1, int r1, int l2, int r2, int clip2, void* stream) {
  LfsrOpTimer op_t("window_attn", hd, n1 * n2, lfsr_stream(stream));
  if (!q || !k || !v || !o || nheads <= 0 || (hd != 8 && hd != 16) || ns0 <= 0 || ns1 <= 0 || ns2 <= 0 || n1 <= 0 || n2 <= 0) return LFSR_E_ARG;
  if ((q_stride | q_choff | k_stride | k_choff | v_stride | v_choff | o_stride | o_choff) & 3) return LFSR_E_ARG;
  AttnArgs p{};
  p.Q = q; p.q_stride = q_stride; p.q_choff = q_choff; p.K = k; p.k_stride = k_stride; p.k_choff = k_choff;
  p.V = v; p.v_stride = v_stride; p.v_choff = v_choff; p.O = o; p.o_stride = o_stride; p.o_choff = o_choff;
  p.nheads = nheads; p.ns0 = ns0; p.ns1 = ns1; p.ns2 = ns2; p.bs0 = bs0; p.bs1 = bs1; p.bs2 = bs2;
  p.n1 = n1; p.n2 = n2; p.st1 = st1; p.st2 = st2; p.l1 = l1; p.r1 = r1; p.l2 = l2; p.r2 = r2; p.clip2 = clip2 > 0 ? clip2 : n2;
  p.scale = 1.0f / sqrtf((float)hd);
  p.scale2 = p.scale * 1.44269504088896340736f;
  p.total = (long long)ns0 * ns1 * ns2 * n1 * n2 * nheads;
  // EPI geometry (every angular position visible, <= 160 tokens per sequence, heads of 16): QK^T / softmax / PV on the matrix pipe
  // (attn_mfma.hip); LFSR_ATTN=valu keeps the VALU kernels below (A/B runs)
  {
    const char* asel = lfsr_sel("LFSR_ATTN");
    if (hd == 16 && !(asel && asel[0] == 'v') && !lfsr_sel("LFSR_ATTN_L1")) {
      const int rc = lfsr_epi_attn_mfma_launch(q, q_stride, q_choff, k, k_stride, k_choff, v, v_stride, v_choff, o, o_stride, o_choff, nheads, ns0, ns1, ns2,
                                               bs0, bs1, bs2, n1, n2, st1, st2, l1, r1, l2, r2, clip2, lfsr_stream(stream));
      if (rc != LFSR_E_ARG) return rc;
    }
    // 5 x 5 spatial windows (LFT's SpaTrans): tiles of 4 x 4 queries against the 8 x 8 keys around them on the matrix pipe (win_attn_mfma.hip)
    if (hd == 16 && !(asel && asel[0] == 'v') && !lfsr_sel("LFSR_ATTN_L1")) {
      const int rc = lfsr_win_attn_mfma_launch(q, q_stride, q_choff, k, k_stride, k_choff, v, v_stride, v_choff, o, o_stride, o_choff, nheads, ns0, ns1, ns2,
                                               bs0, bs1, bs2, n1, n2, st1, st2, l1, r1, l2, r2, clip2, lfsr_stream(stream));
      if (rc != LFSR_E_ARG) return rc;
    }
  }
  // LDS-tiled path (hd 16, heads in pairs): stage the keys a tile of queries can see once; used when the staged tile fits
  if (hd == 16 && nheads % 2 == 0 && !lfsr_sel("LFSR_ATTN_L1")) {
    constexpr int HB = 2, TS = HB * 32 + 4;
    int T1 = n1;                                            // whole sequence if it fits (EPIT: 5 x 32 tokens)
    auto smem_for = [&](int t1) { int rows = t1 + l1 + r1 - 1; if (rows > n1) rows = n1; return (size_t)rows * n2 * TS * 4; };
    while (T1 > 1 && smem_for(T1) > 96 * 1024) T1 = (T1 + 1) / 2;
    if (smem_for(T1) <= 150 * 1024) {
      const int ntile1 = (n1 + T1 - 1) / T1;
      size_t smem = smem_for(T1);
      static std::atomic<bool> attr_set[64];
      int dev = 0;
      if (hipGetDevice(&dev) == hipSuccess && dev >= 0 && dev < 64 && !attr_set[dev]) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(k_window_attn_lds<16, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024) != hipSuccess) return LFSR_E_ARG;
        attr_set[dev] = true;
      }
      int threads = T1 * n2 * HB;
      threads = (threads + 63) / 64 * 64;
      if (threads > 1024) threads = 1024;
      dim3 grid((unsigned)((long long)ns0 * ns1 * ns2 * ntile1), (unsigned)(nheads / HB));
      hipLaunchKernelGGL((k_window_attn_lds<16, 2>), grid, dim3(threads), smem, lfsr_stream(stream), p, T1, ntile1);
      LFSR_CHECK_LAUNCH();
      return LFSR_OK;
    }
  }
  // heads of 8, all eight in one block, whole (short) sequences: LFT's angular attention (25 views per pixel).  One block per sequence stages its 25 x (k | v) rows
  // once; the L1-served kernel below re-reads them for every query (it ran at the L1's 64 B/clk: 0.43 ms per launch at 32 patches).  LFSR_ATTN_ANG=l1 keeps it (A/B runs)
  if (hd == 8 && nheads == 8 && n2 == 1 && l1 >= n1 && r1 >= n1 && n1 <= 64 && !lfsr_sel("LFSR_ATTN_L1")) {
    const char* asel = lfsr_sel("LFSR_ATTN_ANG");
    if (n1 == 25 && !asel) {      // A = 5: two queries per thread, two-pass softmax, four pixels per 512-thread block (two pixels per 256-thread block: 220 against 213 us) (LFSR_ATTN_ANG=lds / loop / l1: the earlier forms)
      constexpr int AP = 4, NTH = 512, TS8 = 8 * 20 + 4;
      constexpr int smem = AP * 25 * TS8 * 4 + AP * 8;
      static std::atomic<bool> attr_set[64];
      int dev = 0;
      if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return LFSR_E_ARG;
      if (!attr_set[dev]) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_ang_attn_pair<25, AP, NTH>), hipFuncAttributeMaxDynamicSharedMemorySize, smem);
        if (e != hipSuccess) return LFSR_HIP_ERR(e);
        attr_set[dev] = true;
      }
      const long long nseq = (long long)ns0 * ns1 * ns2;
      hipLaunchKernelGGL((k_ang_attn_pair<25, AP, NTH>), dim3((unsigned)((nseq + AP - 1) / AP)), dim3(NTH), smem, lfsr_stream(stream), p, nseq);
      LFSR_CHECK_LAUNCH();
      return LFSR_OK;
    }
    if (!(asel && asel[0] == 'l' && asel[1] == '1')) {
      p.loop_stage = asel && asel[0] == 'l' && asel[1] == 'o';
      constexpr int TS8 = 8 * 20 + 4;
      const size_t smem = (size_t)n1 * TS8 * 4;
      int threads = (n1 * 8 + 63) / 64 * 64;
      dim3 grid((unsigned)((long long)ns0 * ns1 * ns2), 1);
      hipLaunchKernelGGL((k_window_attn_lds<8, 8>), grid, dim3(threads), smem, lfsr_stream(stream), p, n1, 1);
      LFSR_CHECK_LAUNCH();
      return LFSR_OK;
    }
  }
  unsigned grid = lfsr_blocks(p.total, 256);
  if (hd == 8) hipLaunchKernelGGL((k_window_attn<8>), dim3(grid), dim3(256), 0, lfsr_stream(stream), p);
  else hipLaunchKernelGGL((k_window_attn<16>), dim3(grid), dim3(256), 0, lfsr_stream(stream), p);
  LFSR_CHECK_LAUNCH();
  return LFSR_OK;
}

int lfsr_linear_fwd(const float* x, int x_stride, int x_choff, int cin, const float* w_packed, const float* bias,
                    const float* res, int res_stride, int res_choff, float* y, int y_stride, int y_choff, long long M, int N, float slope, void* stream) {
  LfsrOpTimer op_t("linear", cin, N, lfsr_stream(stream));
  if (!x || !w_packed || !y || M <= 0 || M >= (1LL << 31) || N <= 0) return LFSR_E_ARG;
  if (x_stride < x_choff + cin || y_stride < y_choff + N || (x_stride | x_choff) & 3) return LFSR_E_ARG;
  GemmArgs p{};
  p.X = x; p.x_stride = x_stride; p.x_choff = x_choff; p.Wp = w_packed; p.bias = bias;
  p.Y = y; p.y_stride = y_stride; p.y_choff = y_choff; p.R1 = res; p.r1_stride = res_stride; p.r1_choff = res_choff;
  p.M = (int)M; p.N = N; p.Npad = npad32(N); p.A = 1; p.AA = 1; p.H = 1; p.W = 1; p.ntaps = 1; p.CH = N; p.slope = slope;
  hipStream_t st = lfsr_stream(stream);
  if (M >= 2048 && !lfsr_sel("LFSR_NO_ROWGEMM")) {
    int rc = lfsr_rowgemm_launch(x, x_stride, x_choff, cin, w_packed, bias, res, res_stride, res_choff, y, y_stride, y_choff, M, N, slope, st);
    if (rc != LFSR_E_ARG) return rc;
  }
  const bool two = (p.Npad % 64) == 0;
  switch (cin) {
    case 64: return two ? launch_gemm<IN_SAME, OUT_SAME, 64, 2>(p, st) : launch_gemm<IN_SAME, OUT_SAME, 64, 1>(p, st);
    case 128: return two ? launch_gemm<IN_SAME, OUT_SAME, 128, 2>(p, st) : launch_gemm<IN_SAME, OUT_SAME, 128, 1>(p, st);
    case 256: return two ? launch_gemm<IN_SAME, OUT_SAME, 256, 2>(p, st) : launch_gemm<IN_SAME, OUT_SAME, 256, 1>(p, st);
    default: return LFSR_E_ARG;
  }
}

int lfsr_conv3x3_n_fwd(const float* x, int x_stride, int x_choff, const float* w_packed, float* y, int y_stride, int y_choff,
                       int n_img, int h, int w, int N, float slope, void* stream) {
  LfsrOpTimer op_t("conv3x3_n", n_img, h * w, lfsr_stream(stream));
  // per-view 3x3 conv 64 -> N (any N): LFT's unfold(3x3)+Linear(576->128) token embedding (LFT.py:176-182)
  if (!x || !w_packed || !y || n_img <= 0 || h <= 0 || w <= 0 || N <= 0 || x_stride < x_choff + 64 || y_stride < y_choff + N || (x_stride | x_choff) & 3) return LFSR_E_ARG;
  GemmArgs p{};
  p.X = x; p.x_stride = x_stride; p.x_choff = x_choff; p.Wp = w_packed;
  p.Y = y; p.y_stride = y_stride; p.y_choff = y_choff;
  p.M = n_img * h * w; p.N = N; p.Npad = npad32(N); p.A = 1; p.AA = 1; p.H = h; p.W = w; p.ntaps = 9; p.CH = N; p.slope = slope;
  return (p.Npad % 64) == 0 ? launch_gemm<IN_CONV3, OUT_SAME, 64, 2>(p, lfsr_stream(stream)) : launch_gemm<IN_CONV3, OUT_SAME, 64, 1>(p, lfsr_stream(stream));
}

int lfsr_lft_position_fwd(float* spa_pe, float* ang_pe, int A, int h, int w, int C, void* stream) {
  // PositionEncoding.forward (LFT.py:106-130): spa_pe (h*w, C) = (PE_h[y] + PE_w[x]) / 2 ; ang_pe (A*A, C) = PE_a[a]
  if (!spa_pe || !ang_pe || A <= 0 || h <= 0 || w <= 0 || C <= 0 || (C & 1)) return LFSR_E_ARG;
  int total = (h * w + A * A) * C;
  hipLaunchKernelGGL(k_lft_position, dim3((total + 255) / 256), dim3(256), 0, lfsr_stream(stream), spa_pe, ang_pe, A * A, h, w, C);
  LFSR_CHECK_LAUNCH();
  return LFSR_OK;
}

int lfsr_upsample_ps_fwd(const float* f, int f_stride, int f_choff, const float* w_packed, float* hr, int B, int A, int h, int w, int s, void* stream) {
  // 1x1 64 -> 64 s^2 (no bias) + PixelShuffle(s), written as the channel-last HR mosaic (B, A*h*s, A*w*s, 64); w packed with perm 1, ch 64
  if (!f || !w_packed || !hr || B <= 0 || A <= 0 || h <= 0 || w <= 0 || s <= 0 || f_stride < f_choff + 64 || (f_stride | f_choff) & 3) return LFSR_E_ARG;
  GemmArgs p{};
  p.X = f; p.x_stride = f_stride; p.x_choff = f_choff; p.Wp = w_packed;
  p.Y = hr; p.y_stride = 64; p.y_choff = 0;
  p.M = B * A * A * h * w; p.N = 64 * s * s; p.Npad = p.N; p.A = A; p.AA = A * A; p.H = h; p.W = w; p.ntaps = 1; p.CH = 64; p.slope = 1.0f; p.S = s;
  return launch_gemm<IN_SAME, OUT_PS_HR, 64, 2>(p, lfsr_stream(stream));
}

int lfsr_hr_tail_fwd(const float* hr, const float* w, const float* x_lr, float* out, int B, int A, int h, int wd, int s, float slope, void* stream) {
  if (!hr || !w || !x_lr || !out || B <= 0 || A <= 0 || h <= 0 || wd <= 0 || s <= 0) return LFSR_E_ARG;
  long long total = (long long)B * A * h * s * A * wd * s;
  unsigned grid = lfsr_blocks(total, 256);
  if (grid > 256u * 64) grid = 256u * 64;
  hipLaunchKernelGGL(k_hr_tail, dim3(grid), dim3(256), 0, lfsr_stream(stream), hr, w, x_lr, out, B, A, h, wd, s, slope);
  LFSR_CHECK_LAUNCH();
  return LFSR_OK;
}

}  // extern "C"
