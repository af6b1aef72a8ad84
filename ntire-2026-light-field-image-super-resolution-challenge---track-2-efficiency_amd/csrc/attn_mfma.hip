// Epipolar-plane self-attention on the matrix pipe (model/SR/EPIT.py:93-128: BasicTrans.gen_mask + nn.MultiheadAttention with
// q = k = LN(x), v = x, 8 heads of 16, additive -inf mask with mask_field = [2A, 11]; configs[2] of BASELINE.json).
//
// A sequence is the (n1 x n2) token grid of one EPI (EPIT: n1 = A angular positions, all of which every query sees, n2 = 32 spatial
// positions with the window [t2 - l2, t2 + r2)); tokens are enumerated t2-major (i = t2 n1 + t1), so the valid keys of a query are ONE
// contiguous band [lo, hi) of the key index, and a tile of 16 queries needs only the aligned key tiles its bands touch
// (EPIT: 5-6 of the 10).  Per (sequence, head), fp32 v_mfma_f32_16x16x4_f32 throughout:
//   S^T tile  = K_tile Q_tile^T       A = K [key = lane & 15][d = 4 g + r], B = Q [q = lane & 15][d = 4 g + r]  (g = lane >> 4, r = MFMA step:
//                                     the sum over d may run in any order as long as A and B agree)
//   softmax   over the keys of a query = over registers, over the four 16-lane groups (two wave shuffles) and over the key tiles:
//             the query sits on the lane, so no LDS round trip; all S^T tiles of a query tile stay in registers (<= 10 x 4)
//   O^T tile += V_tile^T P_tile^T      A = V^T [d = lane & 15][key = 4 g + r], B = P^T = the S^T accumulator registers AS THEY STAND
//                                     (D layout: lane (q, g), register r = key 4 g + r -- exactly the B operand of step r)
//   out[q][head 16 + 4 g .. + 3] = O^T / l : 16 B per lane.
// K [token][16] and V^T [key tile][d][16 keys] of a head are staged in LDS once per sequence: 20 KB per head, a 512-thread block = 4 heads x 2 waves (each
// wave takes every other query tile), 80 KB -> two blocks per CU.  The 16-B chunks of a row sit at swizzled positions so that the A-operand ds_read_b128
// are conflict-free for the lane groups the hardware actually serves together ({0-3, 12-15, 20-27}, {4-11, 16-19, 28-31} and the same + 32 --
// MI355X_MICROARCH.md, LDS): K chunk c of token t at (c + ((t >> 2) & 2)) & 3, V^T chunk j of row d at j ^ ((-(d >> 2)) & 3).  The round-2 rotations (by
// t >> 2 and by 4 d) were conflict-free for CONTIGUOUS groups of 16 lanes and two-way conflicted on the real ones (SQ_LDS_BANK_CONFLICT 39 % of the
// LDS-active cycles, profiles/r03_logs/pmc_epit_sq_summary.json).
// Round 4, measured and NOT adopted (profiles/r04_logs/c12_*, c14_*; one process each, 134-136 us for this kernel at B = 8): the conflict-free layouts themselves are
// worth 1 %; two / one head(s) per block (256 / 128 threads) are 3 % / 8 % slower; the three-term bf16 form (K, V split once at staging into bf16 planes, Q once per
// tile, P per key tile; two of the six products per v_mfma_f32_16x16x32_bf16) is bit-for-bit as accurate (1.7e-6 from this kernel) and 19-29 % SLOWER (162-175 us):
// per (query tile, key tile) pair the kernel issues ~28 VALU instructions beside 8 fp32 MFMAs -- mask, running maximum, exponentials -- and the split of P adds 14
// more while the matrix work it saves was not what the pair waits for.
#include <math.h>

#include "lfsr_internal.h"

typedef float f32x4a __attribute__((ext_vector_type(4)));

namespace {

struct EpiAttnArgs {
  const float* Q; int q_stride, q_choff;
  const float* K; int k_stride, k_choff;
  const float* V; int v_stride, v_choff;
  float* O; int o_stride, o_choff;
  int nheads;
  int ns1, ns2; long long bs0, bs1, bs2;
  int n1, n2; long long st1, st2;
  int l2, r2, clip2;
  float scale;     // 1 / sqrt(hd) * log2(e): the softmax runs in base 2
  int L;           // n1 * n2
};

template <int NT, int N1>      // N1: the angular resolution when known at compile time (5: the BASELINE geometry; divisions by it become multiplies), 0: read from the arguments
__global__ __launch_bounds__(512) void k_epi_attn_mfma(EpiAttnArgs p) {
  const int n1 = N1 ? N1 : p.n1;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int LR = NT * 16;                 // padded sequence length
  constexpr int HEAD_FLOATS = LR * 16 * 2;    // K [LR][16] + V^T [16][LR]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int h4 = wave >> 1, half = wave & 1, l15 = lane & 15, g = lane >> 4;
  const int hblocks = p.nheads >> 2;
  const int hq = blockIdx.x % hblocks;
  int t = blockIdx.x / hblocks;
  const int s2 = t % p.ns2; t /= p.ns2;
  const int s1 = t % p.ns1;
  const int s0 = t / p.ns1;
  const long long base = s0 * p.bs0 + s1 * p.bs1 + s2 * p.bs2;
  const int head = hq * 4 + h4;
  float* const sK = smem + h4 * HEAD_FLOATS;
  float* const sVt = sK + LR * 16;

  // ---- stage K and V^T of this head (the head's two waves: 128 threads; item = (token, 16-B chunk c of its 64-B head slice)).  All of a thread's loads are
  // issued before its first LDS store (round 3: as a loop unrolled by two the five items of a thread were three serial memory round trips per block) ----
  {
    const int t128 = half * 64 + lane;
    constexpr int NIT = (LR * 4 + 127) / 128;
    f32x4a kv[NIT], vv[NIT];
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int idx = t128 + 128 * it;
      const int tok = idx >> 2, c = idx & 3;
      kv[it] = f32x4a{0.f, 0.f, 0.f, 0.f}; vv[it] = kv[it];
      if (idx < LR * 4 && tok < p.L) {
        const int t2 = tok / n1, t1 = tok - t2 * n1;
        const long long pix = base + t1 * p.st1 + t2 * p.st2;
        kv[it] = *reinterpret_cast<const f32x4a*>(p.K + pix * p.k_stride + p.k_choff + head * 16 + 4 * c);
        vv[it] = *reinterpret_cast<const f32x4a*>(p.V + pix * p.v_stride + p.v_choff + head * 16 + 4 * c);
      }
    }
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int idx = t128 + 128 * it;
      if (idx < LR * 4) {
        const int tok = idx >> 2, c = idx & 3;
        *reinterpret_cast<f32x4a*>(sK + tok * 16 + (((c + ((tok >> 2) & 2)) & 3) << 2)) = kv[it];
#pragma unroll
        for (int jj = 0; jj < 4; ++jj)      // V^T [key tile][d = 4 c + jj][16 keys], the 16-B chunk of four keys at position chunk ^ ((-c) & 3)
          sVt[(tok >> 4) * 256 + (4 * c + jj) * 16 + (((((tok >> 2) & 3) ^ ((4 - c) & 3))) << 2) + (tok & 3)] = vv[it][jj];
      }
    }
  }
  __syncthreads();

  const int kmax = min(p.n2, p.clip2);
  // the wave's queries (every other tile of 16) are requested up front: inside the loop each tile's load was a serial memory round trip
  f32x4a qpre[(NT + 1) / 2];
#pragma unroll
  for (int i = 0; i < (NT + 1) / 2; ++i) {
    const int qtok = (half + 2 * i) * 16 + l15;
    const int qc = qtok < p.L ? qtok : p.L - 1;
    const int t2q = qc / n1, t1q = qc - t2q * n1;
    qpre[i] = *reinterpret_cast<const f32x4a*>(p.Q + (base + t1q * p.st1 + t2q * p.st2) * p.q_stride + p.q_choff + head * 16 + 4 * g);
  }
#pragma unroll
  for (int qt = half, qi = 0; qi < (NT + 1) / 2; qt += 2, ++qi) {
    if (qt >= NT || qt * 16 >= p.L) break;
    // ---- this lane's query, its band of valid keys and the (wave-uniform) range of key tiles of the whole query tile ----
    const int qtok = qt * 16 + l15;
    const bool qok = qtok < p.L;
    const int qc = qok ? qtok : p.L - 1;
    const int t2q = qc / n1, t1q = qc - t2q * n1;
    const long long qpix = base + t1q * p.st1 + t2q * p.st2;
    const f32x4a qb = qpre[qi] * p.scale;
    const int lo = max(0, t2q - p.l2) * n1, hi = min(kmax, t2q + p.r2) * n1;
    const int kbase = 4 * g - lo;
    const unsigned kwidth = (unsigned)(hi > lo ? hi - lo : 0);
    const int q2lo = (qt * 16) / n1, q2hi = min(p.L - 1, qt * 16 + 15) / n1;
    const int klo = max(0, q2lo - p.l2) * n1, khi = min(kmax, q2hi + p.r2) * n1;
    const int ktlo = klo >> 4, kthi = (khi - 1) >> 4;

    f32x4a S[NT];
    float m = -INFINITY;
#pragma unroll
    for (int kt = 0; kt < NT; ++kt) {
      // (S[kt] of a key tile outside [ktlo, kthi] is never read: the second loop skips the same tiles -- initialising it cost a register copy per element on the skipped path)
      if (kt >= ktlo && kt <= kthi) {
        const int krow = kt * 16 + l15;
        const f32x4a ka = *reinterpret_cast<const f32x4a*>(sK + krow * 16 + (((g + ((krow >> 2) & 2)) & 3) << 2));
        f32x4a acc = {0.f, 0.f, 0.f, 0.f};
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(ka.x, qb.x, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(ka.y, qb.y, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(ka.z, qb.z, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(ka.w, qb.w, acc, 0, 0, 0);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          // key = kt 16 + 4 g + r is inside [lo, hi)  <=>  (unsigned)(key - lo) < hi - lo: one add, one compare
          const float s = (unsigned)(kbase + (kt * 16 + r)) < kwidth ? acc[r] : -INFINITY;
          S[kt][r] = s;
          m = fmaxf(m, s);
        }
      }
    }
    m = fmaxf(m, __shfl_xor(m, 16));
    m = fmaxf(m, __shfl_xor(m, 32));
    float den = 0.f;
    f32x4a o = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kt = 0; kt < NT; ++kt) {
      if (kt >= ktlo && kt <= kthi) {
        f32x4a pw;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          pw[r] = __builtin_amdgcn_exp2f(S[kt][r] - m);     // raw v_exp_f32 (the argument is <= 0: no range handling needed; results below 2^-126 flush to 0).  exp2(-inf) = 0 on masked keys; an empty band gives NaN exactly like softmax over an all -inf row
          den += pw[r];
        }
        const f32x4a va = *reinterpret_cast<const f32x4a*>(sVt + kt * 256 + l15 * 16 + ((g ^ ((4 - (l15 >> 2)) & 3)) << 2));
        o = __builtin_amdgcn_mfma_f32_16x16x4f32(va.x, pw.x, o, 0, 0, 0);
        o = __builtin_amdgcn_mfma_f32_16x16x4f32(va.y, pw.y, o, 0, 0, 0);
        o = __builtin_amdgcn_mfma_f32_16x16x4f32(va.z, pw.z, o, 0, 0, 0);
        o = __builtin_amdgcn_mfma_f32_16x16x4f32(va.w, pw.w, o, 0, 0, 0);
      }
    }
    den += __shfl_xor(den, 16);
    den += __shfl_xor(den, 32);
    const float inv = 1.0f / den;
    if (qok) *reinterpret_cast<f32x4a*>(p.O + qpix * p.o_stride + p.o_choff + head * 16 + 4 * g) = o * inv;
  }
}

}  // namespace

// LFSR_E_ARG = geometry not covered (the caller falls back to the VALU kernels)
int lfsr_epi_attn_mfma_launch(const float* q, int q_stride, int q_choff, const float* k, int k_stride, int k_choff, const float* v, int v_stride, int v_choff,
                              float* o, int o_stride, int o_choff, int nheads, int ns0, int ns1, int ns2, long long bs0, long long bs1, long long bs2,
                              int n1, int n2, long long st1, long long st2, int l1, int r1, int l2, int r2, int clip2, hipStream_t st) {
  const int L = n1 * n2;
  if (nheads % 4 || L > 160 || L < 1 || l1 < n1 - 1 || r1 < n1) return LFSR_E_ARG;    // every angular position visible; <= 10 tiles of 16 tokens
  EpiAttnArgs p{};
  p.Q = q; p.q_stride = q_stride; p.q_choff = q_choff; p.K = k; p.k_stride = k_stride; p.k_choff = k_choff;
  p.V = v; p.v_stride = v_stride; p.v_choff = v_choff; p.O = o; p.o_stride = o_stride; p.o_choff = o_choff;
  p.nheads = nheads; p.ns1 = ns1; p.ns2 = ns2; p.bs0 = bs0; p.bs1 = bs1; p.bs2 = bs2;
  p.n1 = n1; p.n2 = n2; p.st1 = st1; p.st2 = st2; p.l2 = l2; p.r2 = r2; p.clip2 = clip2 > 0 ? clip2 : n2;
  p.scale = (1.0f / sqrtf(16.0f)) * 1.44269504088896340736f;
  p.L = L;
  const long long nblk = (long long)ns0 * ns1 * ns2 * (nheads / 4);
  if (nblk <= 0 || nblk > 0x7fffffffLL) return LFSR_E_ARG;
  constexpr int NT = 10;
  const int smem = 4 * NT * 16 * 16 * 2 * 4;   // 81920
  static std::atomic<bool> attr_set[64];
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return LFSR_E_ARG;
  if (!attr_set[dev]) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_epi_attn_mfma<NT, 5>), hipFuncAttributeMaxDynamicSharedMemorySize, smem);
    if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_epi_attn_mfma<NT, 0>), hipFuncAttributeMaxDynamicSharedMemorySize, smem);
    if (e != hipSuccess) return LFSR_HIP_ERR(e);
    attr_set[dev] = true;
  }
  if (n1 == 5) hipLaunchKernelGGL((k_epi_attn_mfma<NT, 5>), dim3((unsigned)nblk), dim3(512), smem, st, p);
  else hipLaunchKernelGGL((k_epi_attn_mfma<NT, 0>), dim3((unsigned)nblk), dim3(512), smem, st, p);
  LFSR_CHECK_LAUNCH();
  return LFSR_OK;
}
