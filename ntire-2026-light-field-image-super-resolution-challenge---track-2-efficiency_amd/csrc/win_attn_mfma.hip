// Spatial window self-attention on the matrix pipe (model/SR/LFT.py:161-203: SpaTrans.gen_mask + nn.MultiheadAttention, 8 heads of 16, additive -inf mask
// of the 5 x 5 window [i-2, i+3) x [j-2, min(j+3, clip)) -- the reference clamps the column window with h, LFT.py:168; configs[4] of BASELINE.json).
//
// A sequence is the (n1 x n2) token grid of one view image; a query sees at most 25 of its (up to 1024) keys.  The VALU kernel (transformer.hip:
// k_window_attn_lds) walks those 25 keys per (query, head) with eight ds_read_b128 each and sits at its LDS-read limit.  Here a TILE of 16 queries =
// a 4 x 4 block of tokens; the union of its windows is the 8 x 8 block of keys around it = four key tiles of 2 rows x 8 columns, and per (tile, head)
//   S^T tile  = K_tile Q_tile^T        A = K [key = lane & 15][d = 4 g + r], B = Q [q = lane & 15][d = 4 g + r]           (v_mfma_f32_16x16x4_f32)
//   softmax   over the 64 candidate keys of a query = over 16 registers and the four 16-lane groups (two wave shuffles); keys outside the query's own
//             5 x 5 window (or outside the image) are -inf: 25 of 64 products are used, which still beats 25 scalar key walks by far
//   O^T tile += V_tile^T P_tile^T       A = V^T [d = lane & 15][key = 4 g + r], B = P^T = the S^T accumulator registers as they stand
// exactly as attn_mfma.hip does for EPIT's band.  One 256-thread block = one (sequence, strip of 8 query rows, head): the 12 key rows the strip can see
// are staged once -- K [row][col][16] with the 16-B chunks rotated by the key index (row pitch 40 keys: the two rows of a key tile land on disjoint bank
// groups), V^T [d][row][col] with a d pitch of 484 floats (the 16 lanes of an A-operand read hit 16 distinct 16-B slots) -- 60 KB, two blocks per CU.
#include <math.h>

#include "lfsr_internal.h"

typedef float f32x4q __attribute__((ext_vector_type(4)));

#ifndef WA_XCD_RANGES
#define WA_XCD_RANGES 1   // 1: an XCD's units cover a contiguous range of (sequence, strip) pairs; 0: pairs dealt round-robin over the XCDs
#endif

namespace {

constexpr int WA_ROWS = 12, WA_KP = 40, WA_VD = 484;                    // staged key rows, key-row pitch (keys), V^T pitch per d (floats)
constexpr int WA_SMEM = (WA_ROWS * WA_KP * 16 + 16 * WA_VD) * 4;        // 61696

struct WinAttnArgs {
  const float* Q; int q_stride, q_choff;
  const float* K; int k_stride, k_choff;
  const float* V; int v_stride, v_choff;
  float* O; int o_stride, o_choff;
  int ns1, ns2; long long bs0, bs1, bs2;
  int n1, n2; long long st1, st2;
  int kmax;        // min(n2, clip2): first invalid key column
  unsigned nheads;
  long long nunits;  // sequences x strips x heads
  int remap;         // 1: the XCD-aware unit order (needs (sequences x strips) and the grid to be multiples of 8)
  int t_per_xcd;     // (sequences x strips) / 8: an XCD walks a CONTIGUOUS range of (sequence, strip) pairs -- vertically adjacent strips share four of their twelve key rows,
                     // and with the strips dealt round-robin over the XCDs (round 3) no L2 ever saw both: 1.19 x the algorithmic bytes per launch, all of it that halo
  int nstrip, ntc; // strips of 8 query rows per sequence; 4-column query tiles per row
  float scale;     // 1 / sqrt(16) * log2(e)
};

__global__ __launch_bounds__(256) void k_win_attn_mfma(WinAttnArgs p) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* const sK = smem;                                   // [12][40][16], chunk c of key kl at ((c + (kl >> 2)) & 3)
  float* const sVt = smem + WA_ROWS * WA_KP * 16;           // [16][484]: [d][row * 40 + col]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l15 = lane & 15, g = lane >> 4;
  const int ncol = 4 * p.ntc + 4;                           // staged key columns: token columns -2 .. 4 ntc + 1
  const int ntile = 2 * p.ntc;
  const int nitem = WA_ROWS * ncol * 4;
  const unsigned magic = (65536u + ncol - 1) / ncol;        // key / ncol = (key * magic) >> 16 for key < 432 (ncol <= 36)
  constexpr int NIT = (WA_ROWS * 36 * 4 + 255) / 256;       // 7: staging items per thread at the widest geometry

  // Persistent blocks over units (sequence, strip, head).  Unit order: the heads of one (sequence, strip) read the SAME token rows (a head's slice is 64 B of a
  // 512-B / 1-KB row), so they must meet in one L2: consecutive workgroups go to consecutive XCDs, hence unit id = (slot j, xcd), j = (strip-group, head) -- the
  // nheads units of a strip run at the same time on one XCD and each 128-B line comes from HBM once (with the heads as the slow dimension every line was fetched
  // once per head).  A block's units are gridDim.x apart (a multiple of 8 whenever the remap applies), so it keeps its XCD.
  struct Unit { int head, strip; int base; };
  auto unit_of = [&](unsigned u) -> Unit {          // (32-bit throughout: the launcher keeps the unit count and every pixel index below 2^31)
    Unit U;
    unsigned t;
    if (p.remap) {
      const unsigned xcd = u & 7, jx = u >> 3;
      U.head = (int)(jx % p.nheads);
      t = WA_XCD_RANGES ? xcd * (unsigned)p.t_per_xcd + jx / p.nheads : (jx / p.nheads) * 8 + xcd;
    } else {
      U.head = (int)(u % p.nheads);
      t = u / p.nheads;
    }
    U.strip = (int)(t % (unsigned)p.nstrip); t /= (unsigned)p.nstrip;
    const unsigned s2 = t % (unsigned)p.ns2; t /= (unsigned)p.ns2;
    const unsigned s1 = t % (unsigned)p.ns1;
    const unsigned s0 = t / (unsigned)p.ns1;
    U.base = (int)(s0 * (unsigned)p.bs0 + s1 * (unsigned)p.bs1 + s2 * (unsigned)p.bs2);
    return U;
  };
  // staging of a unit's 12 key rows: every load of a thread is issued before its first LDS store, and the NEXT unit's loads fly under this unit's MFMAs.
  // Addresses are 32-bit byte offsets into buffer descriptors (the launcher keeps every tensor's extent below 2^31): a thread's item offsets relative to the strip's first
  // staged row are worked out ONCE (the 64-bit pixel arithmetic per item and unit, and the SGPR spills it caused, were most of this kernel's VALU instructions); a unit adds its
  // base and turns the rows outside the image into out-of-range offsets (zeros without traffic)
  constexpr int OOBW = (int)0x80000000u;
  const __amdgpu_buffer_rsrc_t rsK = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.K), 0, 0x7fffffff, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsV = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.V), 0, 0x7fffffff, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsQ = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.Q), 0, 0x7fffffff, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsO = __builtin_amdgcn_make_buffer_rsrc(p.O, 0, 0x7fffffff, 0x00020000);
  const int st1 = (int)p.st1, st2 = (int)p.st2;
  f32x4q kv[NIT], vv[NIT];
  int kls[NIT], krow[NIT], kpix[NIT];           // LDS slot (-1: none), staged row, pixel offset relative to the strip's first staged row (invalid column: none)
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    const int idx = tid + 256 * it;
    const int key = idx >> 2;
    const int row = (int)(((unsigned)key * magic) >> 16), ci = key - row * ncol;
    const int kc = ci - 2;
    const bool ok = idx < nitem && kc >= 0 && kc < p.n2;
    kls[it] = idx < nitem ? row * WA_KP + ci : -1;
    krow[it] = ok ? row : -(1 << 20);             // (an invalid column fails every row test)
    kpix[it] = row * st1 + kc * st2;
  }
  const int kc4 = (tid & 3) * 16;                 // the 16-B chunk of the head's 64 B
  auto fetch = [&](const Unit& U, bool valid) {
    const int r0 = U.strip * 8 - 2;                         // token row of staged key row 0
    const int pb = U.base + r0 * st1;                  // pixel of (staged row 0, column 0)
    const int kb = (pb * p.k_stride + p.k_choff + U.head * 16) * 4 + kc4, vb = (pb * p.v_stride + p.v_choff + U.head * 16) * 4 + kc4;
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const bool ok = valid && (unsigned)(r0 + krow[it]) < (unsigned)p.n1;
      kv[it] = __builtin_bit_cast(f32x4q, __builtin_amdgcn_raw_buffer_load_b128(rsK, ok ? kb + kpix[it] * (p.k_stride * 4) : OOBW, 0, 0));
      vv[it] = __builtin_bit_cast(f32x4q, __builtin_amdgcn_raw_buffer_load_b128(rsV, ok ? vb + kpix[it] * (p.v_stride * 4) : OOBW, 0, 0));
    }
  };
  auto stash = [&]() {
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int kl = kls[it], c = tid & 3;                  // (256 is a multiple of 4: the chunk index does not depend on it)
      if (kl >= 0) {
        *reinterpret_cast<f32x4q*>(sK + kl * 16 + (((c + (kl >> 2)) & 3) << 2)) = kv[it];
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) sVt[(4 * c + jj) * WA_VD + kl] = vv[it][jj];
      }
    }
  };
  // this lane's query / output token of tile i (wave + 4 i): row 4 al + (l15 >> 2) of the strip, column 4 b + (l15 & 3); pixel offset relative to the strip's first row
  f32x4q qreg[4];
  int qrow[4], qpix[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int tile = wave + 4 * i;
    const int al = tile / p.ntc, b = tile - al * p.ntc;
    const int qr = 4 * al + (l15 >> 2), qc = 4 * b + (l15 & 3);
    qrow[i] = (tile < ntile && qc < p.n2) ? qr : (1 << 20);
    qpix[i] = qr * st1 + qc * st2;
  }
  auto fetch_q = [&](const Unit& U, bool valid) {
    const int pb = U.base + U.strip * 8 * st1;
    const int qb = (pb * p.q_stride + p.q_choff + U.head * 16 + 4 * g) * 4;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const bool ok = valid && U.strip * 8 + qrow[i] < p.n1;
      qreg[i] = __builtin_bit_cast(f32x4q, __builtin_amdgcn_raw_buffer_load_b128(rsQ, ok ? qb + qpix[i] * (p.q_stride * 4) : OOBW, 0, 0));
    }
  };

  unsigned u = blockIdx.x;
  if (u >= (unsigned)p.nunits) return;                                // (block-uniform)
  Unit U = unit_of(u);
  fetch_q(U, true);
  fetch(U, true);
  stash();
  __syncthreads();
  for (;;) {
    const unsigned un = u + gridDim.x;
    const bool more = un < (unsigned)p.nunits;                        // (block-uniform)
    const Unit Un = unit_of(more ? un : u);
    f32x4q qcur[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) qcur[i] = qreg[i];
    fetch(Un, more);                                        // the next unit's keys and queries: in flight during this unit's tiles
    fetch_q(Un, more);
    const int r0 = U.strip * 8 - 2;
#pragma unroll
    for (int ti = 0; ti < 4; ++ti) {
      const int tile = wave + 4 * ti;
      if (tile >= ntile) break;
      const int al = tile / p.ntc, b = tile - al * p.ntc;   // tile row inside the strip (0 / 1), tile column
      // this lane's query
      const int qr = U.strip * 8 + 4 * al + (l15 >> 2), qc = 4 * b + (l15 & 3);
      const bool qok = U.strip * 8 + qrow[ti] < p.n1;
      const f32x4q qb = qcur[ti] * p.scale;
      // registers r of a key tile kt hold key (row 4 al + 2 kt + (g >> 1), column 4 b + 4 (g & 1) + r) of the staged block
      const int kc0 = 4 * b - 2 + 4 * (g & 1);             // token column of register 0
      f32x4q S[4];
      float m = -INFINITY;
#pragma unroll
      for (int kt = 0; kt < 4; ++kt) {
        const int krow = 4 * al + 2 * kt + (l15 >> 3);
        const int kl = krow * WA_KP + 4 * b + (l15 & 7);
        const f32x4q ka = *reinterpret_cast<const f32x4q*>(sK + kl * 16 + (((g + (kl >> 2)) & 3) << 2));
        f32x4q acc = {0.f, 0.f, 0.f, 0.f};
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(ka.x, qb.x, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(ka.y, qb.y, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(ka.z, qb.z, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(ka.w, qb.w, acc, 0, 0, 0);
        const int kr = r0 + 4 * al + 2 * kt + (g >> 1);     // token row of this lane group's keys
        const bool rok = (unsigned)kr < (unsigned)p.n1 && (unsigned)(kr - qr + 2) < 5u;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int kc = kc0 + r;
          const bool ok = rok && (unsigned)kc < (unsigned)p.kmax && (unsigned)(kc - qc + 2) < 5u;
          const float sv = ok ? acc[r] : -INFINITY;
          S[kt][r] = sv;
          m = fmaxf(m, sv);
        }
      }
      m = fmaxf(m, __shfl_xor(m, 16));
      m = fmaxf(m, __shfl_xor(m, 32));
      float den = 0.f;
      f32x4q o = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int kt = 0; kt < 4; ++kt) {
        f32x4q pw;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          pw[r] = __builtin_amdgcn_exp2f(S[kt][r] - m);     // base-2 softmax (q carries log2 e); exp2(-inf) = 0 on masked keys; an empty window gives NaN like softmax over an all -inf row
          den += pw[r];
        }
        const f32x4q va = *reinterpret_cast<const f32x4q*>(sVt + l15 * WA_VD + (4 * al + 2 * kt + (g >> 1)) * WA_KP + 4 * b + 4 * (g & 1));
        o = __builtin_amdgcn_mfma_f32_16x16x4f32(va.x, pw.x, o, 0, 0, 0);
        o = __builtin_amdgcn_mfma_f32_16x16x4f32(va.y, pw.y, o, 0, 0, 0);
        o = __builtin_amdgcn_mfma_f32_16x16x4f32(va.z, pw.z, o, 0, 0, 0);
        o = __builtin_amdgcn_mfma_f32_16x16x4f32(va.w, pw.w, o, 0, 0, 0);
      }
      den += __shfl_xor(den, 16);
      den += __shfl_xor(den, 32);
      const float inv = 1.0f / den;
      {
        const int ob = ((U.base + U.strip * 8 * st1) * p.o_stride + p.o_choff + U.head * 16 + 4 * g) * 4;
        typedef unsigned u32x4q __attribute__((ext_vector_type(4)));
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4q, o * inv), rsO, qok ? ob + qpix[ti] * (p.o_stride * 4) : OOBW, 0, 0);
      }
    }
    if (!more) break;
    __syncthreads();                                        // every wave is done with this unit's keys
    stash();
    __syncthreads();
    u = un;
    U = Un;
  }
}

}  // namespace

// LFSR_E_ARG = geometry not covered (the caller keeps the VALU kernels): heads of 16, the 5 x 5 window [t-2, t+3) in both directions, at most 36 staged key columns
int lfsr_win_attn_mfma_launch(const float* q, int q_stride, int q_choff, const float* k, int k_stride, int k_choff, const float* v, int v_stride, int v_choff,
                              float* o, int o_stride, int o_choff, int nheads, int ns0, int ns1, int ns2, long long bs0, long long bs1, long long bs2,
                              int n1, int n2, long long st1, long long st2, int l1, int r1, int l2, int r2, int clip2, hipStream_t st) {
  if (l1 != 2 || r1 != 3 || l2 != 2 || r2 != 3 || n2 > 32 || n2 < 1 || n1 < 1 || nheads < 1 || nheads > 65535) return LFSR_E_ARG;
  if ((q_stride | q_choff | k_stride | k_choff | v_stride | v_choff | o_stride | o_choff) & 3) return LFSR_E_ARG;
  if (((uintptr_t)q | (uintptr_t)k | (uintptr_t)v | (uintptr_t)o) & 15) return LFSR_E_ARG;
  WinAttnArgs p{};
  p.Q = q; p.q_stride = q_stride; p.q_choff = q_choff; p.K = k; p.k_stride = k_stride; p.k_choff = k_choff;
  p.V = v; p.v_stride = v_stride; p.v_choff = v_choff; p.O = o; p.o_stride = o_stride; p.o_choff = o_choff;
  p.ns1 = ns1; p.ns2 = ns2; p.bs0 = bs0; p.bs1 = bs1; p.bs2 = bs2; p.n1 = n1; p.n2 = n2; p.st1 = st1; p.st2 = st2;
  const int clip = clip2 > 0 ? clip2 : n2;
  p.kmax = n2 < clip ? n2 : clip;
  p.nstrip = (n1 + 7) / 8; p.ntc = (n2 + 3) / 4;
  p.scale = (1.0f / sqrtf(16.0f)) * 1.44269504088896340736f;
  {   // 32-bit byte offsets: the farthest pixel any sequence touches, times the widest row, must stay below 2^31 (the caller keeps the VALU kernel otherwise)
    const long long far = (long long)(ns0 - 1) * bs0 + (long long)(ns1 - 1) * bs1 + (long long)(ns2 - 1) * bs2 + (long long)(n1 + 16) * st1 + (long long)(n2 + 8) * st2;
    const long long wide = (long long)(q_stride > k_stride ? q_stride : k_stride) > (long long)(v_stride > o_stride ? v_stride : o_stride)
                               ? (long long)(q_stride > k_stride ? q_stride : k_stride) : (long long)(v_stride > o_stride ? v_stride : o_stride);
    if (bs0 < 0 || bs1 < 0 || bs2 < 0 || st1 <= 0 || st2 <= 0 || (far + 64) * wide * 4 >= (1LL << 31)) return LFSR_E_ARG;
  }
  const long long nblk = (long long)ns0 * ns1 * ns2 * p.nstrip;
  if (nblk <= 0 || nblk * nheads > 0x7fffffffLL) return LFSR_E_ARG;
  p.nheads = (unsigned)nheads;
  p.nunits = nblk * nheads;
  static std::atomic<bool> attr_set[64];
  static std::atomic<int> cus[64];
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return LFSR_E_ARG;
  if (!attr_set[dev]) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_win_attn_mfma), hipFuncAttributeMaxDynamicSharedMemorySize, WA_SMEM);
    if (e != hipSuccess) return LFSR_HIP_ERR(e);
    int v = 0;
    cus[dev] = (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) ? v : 256;
    attr_set[dev] = true;
  }
  long long grid = 2LL * cus[dev];                          // two 60-KB blocks per CU
  if (grid > p.nunits) grid = p.nunits;
  p.remap = (nblk % 8 == 0 && grid % 8 == 0) ? 1 : 0;
  p.t_per_xcd = (int)(nblk / 8);
  hipLaunchKernelGGL(k_win_attn_mfma, dim3((unsigned)grid), dim3(256), WA_SMEM, st, p);
  LFSR_CHECK_LAUNCH();
  return LFSR_OK;
}
