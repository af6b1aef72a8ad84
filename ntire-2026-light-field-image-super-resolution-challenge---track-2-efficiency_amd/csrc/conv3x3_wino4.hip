// Per-view 3x3 conv, zero pad 1, 64 -> 64 channels, VCL layout -- Winograd F(4x4, 3x3) on the fp32 MFMA pipe.
// Reference: the MacPI convs "k3, dilation A, padding A" of model/SR/DistgSSR.py:22,47,64,79-83,101 (per-view 3x3 in VCL),
// EPIT.py:24-32,136-142 / LFT.py:36-46 (Conv3d(1,3,3)).
//
//   Y = At [ sum_c (G g G^t) . (Bt d B) ] A     per 4x4 output tile, d = its 6x6 input patch, 36 transform positions p = (xi, nu)
//
// 36 position-GEMMs per 16 outputs = 2.25 multiplies per output and channel pair instead of 9 (direct) or 4 (F(2x2,3x3)).
// The transform coefficients are small integers (Bt: 1, 2, 4, 5; At: 1, 2, 4, 8) and 1/4, 1/6, 1/12, 1/24 in G, which is applied
// in fp64 at pack time; round-off of the whole DistgSSR forward stays at 1e-6 (tools/conv_error.py).
//
// One 512-thread block per CU walks 8-row x 32-column output tiles (16 Winograd tiles each):
//  * v_mfma_f32_16x16x4_f32 with A = U (rows = 16 output channels) and B = V (columns = 16 tiles): 36 accumulators of 4 registers
//    per wave, so the whole inverse transform At M A happens in registers -- no cross-wave reduction;
//  * V[parity][tile][channel 0..15][36 positions] in LDS, tile stride 584 floats: the 16-lane groups of ds_read_b128 and the
//    8-lane groups of ds_write_b128 are both conflict-free;
//  * U never touches LDS: the pack is in fragment order [stage k/4][ns][p/4][lane][p%4], a wave streams its 9 KB per stage
//    through a register ring of 16-B fragments (L2 hits, 1 KB contiguous per wave instruction);
//  * the waves are specialised (see the kernel): four consume (MFMA), four produce (patch loads, input transform, epilogue I/O);
//  * a last round that would fill at most half of the CUs runs as HALF tiles on twice as many (round 4): a CU takes 32 of the tile's output channels, its four
//    consumer waves split the 36 positions by nu < 3 | nu >= 3 (18 accumulators, five of the nine U / V fragments per stage each) and the two partial inverse
//    transforms meet in the exchange planes, added in the association at6 itself uses -- so which tiles are split (a function of the batch) changes no bit.
//    800 tiles on 256 CUs: 3 rounds + 64 half tiles instead of 4 rounds (58.4 -> 54.2 us per op on one box, 50.2 on a faster one); 100 tiles: 16.6 -> 12.9 us.
#include <stdlib.h>
#include <type_traits>

#include "lfsr_internal.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

#ifndef W4_ABL
#define W4_ABL 0   // diagnostic timing builds (wrong results): 1 no halo loads, 2 no input transform, 4 no V writes, 16 no U loads, 32 a quarter of the epilogue
                   // stores, 256 / 512 / 1024 every store / halo load / epilogue-operand load inside the image's first 64 KB (cache hits)
#endif
#ifndef W4_VRING
#define W4_VRING 4    // depth of the V fragment ring (LDS reads W4_VRING - 1 groups ahead); must divide 36
#endif
#ifndef W4_URING
#define W4_URING 12   // depth of the U fragment ring (16-B fragments in flight per wave); must divide 144
#endif

#ifndef W4_STPOL
#define W4_STPOL (LFSR_NT_STORES ? 2 : 0)    // ... and of the output stores: nt (see lfsr_store_stream in lfsr_common.h)
#endif
#ifndef W4_AG
#define W4_AG 24      // the MFMA group (0..35) of a chunk at which the consumers join barrier A (producers: halo staged); 8..30 measured the same (profiles/r04_logs)
#endif

#ifndef W4_HAG
#define W4_HAG 12     // ... and of a half tile's chunk (0..19)
#endif
#ifndef W4_DRAIN
#define W4_DRAIN 1    // 1: the producers drain the previous tile's exchange plane BEFORE barrier A (while they would wait for the consumers to get there) instead of
                      // behind the V writes, where its four LDS round trips sat on the path that the consumers wait for at the chunk barrier
#endif
#ifndef W4_EPF
#define W4_EPF 1      // 1 (needs W4_DRAIN): the epilogue operand (residual / saved activation) of plane c + 1 is requested right after plane c has been drained: a whole
                      // chunk step ahead of its use instead of a few hundred cycles
#endif
#ifndef W4_XB
#define W4_XB 1       // 1 (needs W4_DRAIN): no chunk barrier behind a tile's last chunk -- the exchange barrier that follows At M A publishes the next tile's first V chunk too
                      // (the producers drained plane 3 before barrier A of that step, so the consumers may write the planes as soon as their MFMAs are done)
#endif
#ifndef W4_ANW
#define W4_ANW 1      // 1: the consumers join barrier A (which orders the PRODUCERS' halo stores and patch reads) without draining their own LDS reads first
#endif

#ifndef W4_STAMP_A
#define W4_STAMP_A 0  // (LFSR_CONV_DIAG builds) 1: stamps around the consumers' barrier A too -- each stamp drains the wave's LDS reads, so the stream is not what it is without them
#endif
#ifdef LFSR_CONV_DIAG
// diagnostic build only: the first consumer wave and the first producer wave accumulate s_memtime deltas per segment into the
// buffer set by lfsr_diag_set_buffer (64 floats per block: consumer 0..31, producer 32..63; producer segment k of step c = 32 + 8 c + k)
#define STAMP(k) do { if (wave == 0) { long long t_ = clock64(); seg[k] += (unsigned)(t_ - tprev); tprev = t_; } } while (0)
#define PSTAMP(k) do { if (wave == 4) { long long t_ = clock64(); seg[k] += (unsigned)(t_ - tprev); tprev = t_; } } while (0)
#else
#define STAMP(k) do { } while (0)
#define PSTAMP(k) do { } while (0)
#endif

#ifndef W4_CLK
#define W4_CLK 0      // 1 (lab builds only): every block leaves its shader-clock and 100-MHz real-time deltas in g_w4_clk (read back by lfsr_w4_clk_read): the clock the chip held
#endif
#if W4_CLK
__device__ unsigned long long g_w4_clk[2 * 1024];
extern "C" int lfsr_w4_clk_read(unsigned long long* out, int n_blocks) {
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_w4_clk), sizeof(unsigned long long) * 2 * n_blocks) == hipSuccess ? 0 : -1;
}
#endif

namespace {

constexpr int TS = 584;                 // floats per tile in a V buffer: 16 channels x 36 positions + 8 (2336 B = 32 mod 256)
constexpr int VBUF = 16 * TS;           // one parity
constexpr int HPIX = 10 * 34;           // raw halo of an 8 x 32 tile
constexpr int HBUF = HPIX * 16;         // one 16-channel chunk, 64 B per pixel
constexpr int SMEM_BYTES = (2 * VBUF + 4 * 4096 + HBUF + 256) * 4;   // V, epilogue exchange (one 64-pixel x 64-channel plane per output
                                                                                      // row of the Winograd tiles), raw halo + 1 KB landing zone: 163072
constexpr int OOB = (int)0x80000000u;

struct Wino4Args {
  const float* X; int x_stride; int x_choff; int x_bytes;   // *_bytes: the operand's true byte span (n_img * h * w * stride * 4); 0 for an absent operand,
  int y_bytes, r1_bytes, r2_bytes, mk_bytes;                //          so a mis-computed offset is a dropped access, never a stray one
  float* dbg;                                               // (LFSR_CONV_DIAG builds: the stamp buffer set by lfsr_diag_set_buffer; else unused)
  const float* Wu;   // [16 stages][4 ns][9 q][64 lanes][4]   (lfsr_pack_wino4)
  float* Y; int y_stride; int y_choff;
  const float* R1; int r1_stride; int r1_choff;
  const float* R2; int r2_stride; int r2_choff;
  const float* Mk; int mk_stride; int mk_choff; float mk_slope;
  int n_img, H, W, tiles_y, tiles_x, ntiles;
  int full_per_block, nhalf;   // nhalf > 0: every block walks full_per_block whole tiles, then the nhalf tiles left over are run as 2 nhalf HALF tiles by blocks 0 .. 2 nhalf - 1
  float slope;
};

#define LDS_BARRIER() do { __builtin_amdgcn_sched_barrier(0); asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); __builtin_amdgcn_sched_barrier(0); } while (0)
#define BARRIER_NOWAIT() do { __builtin_amdgcn_sched_barrier(0); asm volatile("s_barrier" ::: "memory"); __builtin_amdgcn_sched_barrier(0); } while (0)

__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* base, int bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, bytes, 0x00020000);
}
__device__ __forceinline__ f32x4 bload4(__amdgpu_buffer_rsrc_t r, int voff, int soff) {
  return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0));
}
// the producers' streamed operands (halo, epilogue operands: each byte is read about once) and the output, with a cache-policy field of their own
__device__ __forceinline__ f32x4 bload4s(__amdgpu_buffer_rsrc_t r, int voff, int soff) {
  return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0));
}
__device__ __forceinline__ float bload1(__amdgpu_buffer_rsrc_t r, int voff, int soff) {
  return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, voff, soff, 0));
}
__device__ __forceinline__ void bstore4(__amdgpu_buffer_rsrc_t r, int voff, f32x4 v, int soff = 0) {
  __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), r, voff, soff, W4_STPOL);
}

// Transform position (xi, nu) -> its place p in V's rows, in the U pack and in the accumulators: nu-half major (nu < 3 | nu >= 3), so that the 18 positions a wave of a
// HALF tile owns are contiguous: fragments (4 positions) 0..3 and the first two values of fragment 4 | the last two of fragment 4 and fragments 5..8.
#define WP(xi, nu) (18 * ((nu) / 3) + 3 * (xi) + (nu) % 3)

// one 6-vector of the input transform: t = Bt d  (12 operations, integer coefficients: exact products)
__device__ __forceinline__ void bt6(float& d0, float& d1, float& d2, float& d3, float& d4, float& d5) {
  const float a = fmaf(-4.f, d2, d4), b = fmaf(-4.f, d1, d3);
  const float c = d4 - d2, e = d3 - d1;
  const float t0 = fmaf(4.f, d0, fmaf(-5.f, d2, d4));
  const float t5 = fmaf(4.f, d1, fmaf(-5.f, d3, d5));
  d0 = t0; d1 = a + b; d2 = a - b; d3 = fmaf(2.f, e, c); d4 = fmaf(-2.f, e, c); d5 = t5;
}
// one 6-vector of the output transform: y = At m  (4 results)
__device__ __forceinline__ void at6(f32x4& m0, f32x4& m1, f32x4& m2, f32x4& m3, const f32x4 m4, const f32x4 m5) {
  const f32x4 s12 = m1 + m2, d12 = m1 - m2, s34 = m3 + m4, d34 = m3 - m4;
  m0 = (m0 + s12) + s34;
  m1 = d12 + 2.f * d34;
  m2 = s12 + 4.f * s34;
  m3 = d12 + __builtin_elementwise_fma(f32x4{8.f, 8.f, 8.f, 8.f}, d34, m5);   // (this association -- not (d12 + 8 d34) + m5 -- is the one a tile split between two waves by nu < 3 | nu >= 3 can form: see the half tile)
}

// HAS_E / HAS_L: the first (residual R1, or the saved activation of the LeakyReLU' mask) / second epilogue operand exists.
//
// Wave specialisation (512 threads = 8 waves, two per SIMD, <= 256 registers):
//   waves 0..3  CONSUMERS -- wave ns owns output channels 16 ns .. 16 ns + 15 of all 16 Winograd tiles: U fragments straight from
//               L2 through a register ring, V fragments from LDS, 576 MFMAs per tile into 36 accumulators, then At M A in registers
//               and the raw conv result -> the LDS exchange buffer, one output row of every Winograd tile per round.  Their vector
//               memory queue holds nothing but U loads (L2 hits): vector memory operations return in order, so a single HBM miss or
//               store burst in that queue holds up every U fragment queued behind it (measured: 5-6k cycles per tile).
//   waves 4..7  PRODUCERS -- all other global traffic.  Per 16-channel chunk step c:
//                 1. the raw halo of chunk c + 1 (340 pixels x 64 B, requested as six 16-B loads per thread two steps earlier; zero padding and ragged edges
//                    are out-of-range buffer offsets) goes from registers to LDS;
//                 2. exchange plane c of the PREVIOUS tile is drained: four 16-B units per thread read back as whole pixels (256 B contiguous per pixel),
//                    activation / LeakyReLU' mask / residuals applied, stored; then the epilogue operand of plane c + 1 is requested (a whole step ahead);
//                 3. barrier A (all eight waves; the consumers join it at MFMA group W4_AG without draining their LDS reads);
//                 4. one (Winograd tile, channel) item per thread: its 6x6 patch from the staged halo, Bt d B in registers (column pass in packed math),
//                    nine ds_write_b128 into the V buffer of the next chunk; the halo of chunk c + 3 is requested;
//                 5. chunk barrier B -- except behind the tile's last chunk, where the exchange barrier that follows the consumers' At M A does both jobs.
//               Round 4 (profiles/r04_logs): steps 2 and 4 used to run in the order 4, 2 behind barrier A, i.e. the drain's LDS round trips, three scalar
//               branches per unit on the activation form and the wait for a just-requested residual all sat on the path the consumers wait for at B:
//               184 -> 165 us (no residual) and 218 -> 187 us (residual) per op at B = 32 in one process, results bit-equal.
// Barriers per tile: two per chunk (A: halo staged | B: V of the next chunk published, this chunk's V free), the last B replaced by the exchange barrier: 8.
// ACT: the activation as a compile-time form -- 0 none (slope 1), 1 max(v, v * slope) (0 <= slope < 1), 2 the select form (any slope): the run-time choice was
// three scalar branches per drained 16-B unit on the producers' path.
// ALIGNED: H % 8 == 0 and W % 32 == 0 (every model geometry: 32 x 32 views): a tile's 8 x 32 pixels all lie inside the image, so the drain's per-plane offsets
// are wave-uniform (the buffer instruction's SGPR offset, which the range check does not cover -- hence only here) and cost no VALU instruction: 16 fewer
// per chunk step on the producers' path.
template <bool MASK, bool HAS_E, bool HAS_L, int ACT, bool ALIGNED>
__global__ __launch_bounds__(512) void k_conv3x3_wino4(Wino4Args p) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* const sV = smem;                 // V[2][16 tiles][16 ch][36]
  float* const sX = smem + 2 * VBUF;      // epilogue exchange: [4 rows a][64 pixels][64 channels]
  float* const sH = sX + 4 * 4096;        // raw halo of one chunk: [340 pixels][16 channels], then the landing zone
  const int tid = threadIdx.x & 255, lane = tid & 63, w4 = (threadIdx.x >> 6) & 3;
  const bool producer = threadIdx.x >= 256;
  const int nblk = gridDim.x;
  // a block walks a CONTIGUOUS range of tiles (the next tile's patch rows share pages and two pixel rows with this one's)
  int tile = (int)(((long long)blockIdx.x * p.ntiles) / nblk);
  int tile_end = (int)(((long long)(blockIdx.x + 1) * p.ntiles) / nblk);
  // HALF tile (the launch's last, partial round: see the launcher): tile htile for the output channels 32 hc .. 32 hc + 31 only.  Its four consumer waves split the 36
  // transform positions by nu < 3 | nu >= 3 (wave = (16-channel block, nu half)): half the MFMAs per wave, the partial inverse transforms summed through the exchange planes
  // in the association at6 uses -- every bit as in a whole tile.
  bool have_half = false;
  int htile = 0, hc = 0;
  if (p.nhalf > 0) {
    tile = (int)blockIdx.x * p.full_per_block; tile_end = tile + p.full_per_block;
    have_half = (int)blockIdx.x < 2 * p.nhalf;
    htile = p.full_per_block * nblk + ((int)blockIdx.x >> 1); hc = (int)blockIdx.x & 1;
    if ((p.nhalf & 7) == 0 && (nblk & 7) == 0) {   // the two halves of a tile on blocks b and b + 8: the same XCD (workgroups go round-robin over the eight), so the second reader of the tile's halo hits its L2
      const int b = (int)blockIdx.x;
      htile = p.full_per_block * nblk + (b >> 4) * 8 + (b & 7); hc = (b >> 3) & 1;
    }
  }
  const bool only_half = tile >= tile_end;     // (block-uniform; such a block always has a half tile)
#ifdef LFSR_CONV_DIAG
  const int wave = threadIdx.x >> 6;
  unsigned seg[64] = {};
  long long tprev = clock64();
  float* dbgbuf = p.dbg;
#endif

#if W4_CLK
  const unsigned long long clk0 = __builtin_amdgcn_s_memtime(), rt0 = __builtin_amdgcn_s_memrealtime();
#endif
  if (producer) {
    // ======================================================= PRODUCER ==========================================================
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    const int c16 = lane & 15, ptile = 4 * w4 + (lane >> 4), pty = ptile >> 3, ptx = ptile & 7;
    const float* const Ep = MASK ? p.Mk : p.R1;
    const float* const Lp = MASK ? p.R1 : p.R2;
    const int e_stride = MASK ? p.mk_stride : p.r1_stride, e_choff = MASK ? p.mk_choff : p.r1_choff;
    const int l_stride = MASK ? p.r1_stride : p.r2_stride, l_choff = MASK ? p.r1_choff : p.r2_choff;
    const int img_px = p.H * p.W;
    // one descriptor per (operand, image): base = the image's first pixel, extent = the image, so rows above / below the image are
    // out-of-range offsets (negative = huge unsigned, or >= extent) and need no per-slot test; an absent operand / tile gets extent 0
    auto img_rsrc = [&](const float* base, int stride, int img) {
      const bool ok = base != nullptr && img >= 0;
      return make_rsrc(base + (ok ? (long long)img * img_px * stride : 0), ok ? img_px * stride * 4 : 0);
    };
    auto tile_origin = [&](int t, int& img, int& y0, int& x0) {
      int txx = t % p.tiles_x; int q = t / p.tiles_x;
      int tyy = q % p.tiles_y; img = q / p.tiles_y;
      y0 = tyy * 8; x0 = txx * 32;
    };
    // ---- halo staging slots: slot i of a thread = (halo pixel, 16-B quarter of the chunk's 64 B) = (idx >> 2, idx & 3), idx = tid + 256 i
    int hrel[6], hcol[6];     // byte offset relative to the tile's pixel (y0, x0) of the image; column relative to x0 (far negative: no such slot)
#pragma unroll
    for (int i = 0; i < 6; ++i) {
      const int px = (tid + 256 * i) >> 2, cq = tid & 3;
      const int r = __mul24(px, 1928) >> 16;   // px / 34 for px < 384
      const int c = px - r * 34;
      hrel[i] = (((r - 1) * p.W + (c - 1)) * p.x_stride + p.x_choff) * 4 + cq * 16;
      hcol[i] = px < HPIX ? c - 1 : -(1 << 20);
    }
    int hx[6];
    auto halo_offsets = [&](int y0, int x0) {
      const int toff = (y0 * p.W + x0) * (p.x_stride * 4);   // wave-uniform
#pragma unroll
      for (int i = 0; i < 6; ++i) hx[i] = (unsigned)(x0 + hcol[i]) < (unsigned)p.W ? hrel[i] + toff : OOB;
    };
    f32x4 hv0[6], hv1[6];
    auto halo_load = [&](f32x4 (&hv)[6], __amdgpu_buffer_rsrc_t rs, int chunk) {
#pragma unroll
      for (int i = 0; i < 6; ++i) hv[i] = bload4s(rs, (W4_ABL & 512) ? (hx[i] & 0xffff) : hx[i], chunk * 64);   // (512: every load from the image's first 64 KB: cache hits)
    };
    float* hdst[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) {
      const int px = (tid + 256 * i) >> 2, cq = tid & 3;
      hdst[i] = (i < 5 || px < HPIX) ? sH + px * 16 + cq * 4 : sH + HBUF + (tid & 63) * 4;   // slots 340..383: landing zone
    }
    auto halo_store = [&](f32x4 (&hv)[6]) {
#pragma unroll
      for (int i = 0; i < 6; ++i) *reinterpret_cast<f32x4*>(hdst[i]) = hv[i];
    };
    // ---- input transform of one (Winograd tile, channel) item: R[3 r + k] = patch (row r, columns 2 k, 2 k + 1)
    const float* const hR = sH + ((4 * pty) * 34 + 4 * ptx) * 16 + c16;
    f32x2 R[18];
    float Vo[36];      // the item's 36 transformed values in position order (scalar results: their registers are the compiler's choice, the order costs nothing)
    auto read_raw = [&]() {
#pragma unroll
      for (int r = 0; r < 6; ++r)
#pragma unroll
        for (int k = 0; k < 3; ++k) { R[3 * r + k].x = hR[(r * 34 + 2 * k) * 16]; R[3 * r + k].y = hR[(r * 34 + 2 * k + 1) * 16]; }
    };
    auto transform = [&]() {
      // column pass Bt d down the six rows, two columns per instruction (the same fma sequence per element as bt6: bit-identical)
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        f32x2 &d0 = R[k], &d1 = R[3 + k], &d2 = R[6 + k], &d3 = R[9 + k], &d4 = R[12 + k], &d5 = R[15 + k];
        const f32x2 m4 = {-4.f, -4.f}, p4 = {4.f, 4.f}, m5 = {-5.f, -5.f}, p2 = {2.f, 2.f}, m2 = {-2.f, -2.f};
        const f32x2 a = __builtin_elementwise_fma(m4, d2, d4), b = __builtin_elementwise_fma(m4, d1, d3);
        const f32x2 c = d4 - d2, e = d3 - d1;
        const f32x2 t0 = __builtin_elementwise_fma(p4, d0, __builtin_elementwise_fma(m5, d2, d4));
        const f32x2 t5 = __builtin_elementwise_fma(p4, d1, __builtin_elementwise_fma(m5, d3, d5));
        d0 = t0; d1 = a + b; d2 = a - b; d3 = __builtin_elementwise_fma(p2, e, c); d4 = __builtin_elementwise_fma(m2, e, c); d5 = t5;
      }
#pragma unroll
      for (int r = 0; r < 6; ++r) {
        float d0 = R[3 * r].x, d1 = R[3 * r].y, d2 = R[3 * r + 1].x, d3 = R[3 * r + 1].y, d4 = R[3 * r + 2].x, d5 = R[3 * r + 2].y;
        bt6(d0, d1, d2, d3, d4, d5);
        Vo[WP(r, 0)] = d0; Vo[WP(r, 1)] = d1; Vo[WP(r, 2)] = d2; Vo[WP(r, 3)] = d3; Vo[WP(r, 4)] = d4; Vo[WP(r, 5)] = d5;
      }
    };
    float* const vW = sV + ptile * TS + c16 * 36;
    auto write_v = [&](int par) {
#pragma unroll
      for (int q = 0; q < 9; ++q) {
        f32x4 v; v.x = Vo[4 * q]; v.y = Vo[4 * q + 1]; v.z = Vo[4 * q + 2]; v.w = Vo[4 * q + 3];
        *reinterpret_cast<f32x4*>(vW + par * VBUF + 4 * q) = v;
      }
    };
    // ---- drain slots of an exchange plane: slot i of a thread = pixel (row 4 (i >> 1) [+ a], column 16 (i & 1) + tid / 16) of the tile, 16-B unit tid % 16
    const int un = tid & 15;
    int pY[4], pE[4], pL[4], dcol[4];
    const float* dsrc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int px = (tid + 256 * i) >> 4;
      const int pr = (4 * (px >> 5)) * p.W + (px & 31);
      dcol[i] = px & 31;
      pY[i] = (pr * p.y_stride + p.y_choff) * 4 + un * 16;
      pE[i] = (pr * e_stride + e_choff) * 4 + un * 16;
      pL[i] = (pr * l_stride + l_choff) * 4 + un * 16;
      dsrc[i] = sX + px * 64 + ((un ^ ((px >> 2) & 7)) << 2);
    }
    __amdgpu_buffer_rsrc_t rsYp = img_rsrc(nullptr, 0, -1), rsEp = rsYp, rsLp = rsYp;   // the PREVIOUS tile's image (none yet)
    int prow0 = 0, pcol0 = 0;                                                            // ... and its origin
    const bool ragged_w = (p.W & 31) != 0;
    static_assert(!W4_EPF || W4_DRAIN, "W4_EPF needs W4_DRAIN");
    static_assert(!W4_XB || W4_DRAIN, "W4_XB needs W4_DRAIN");
    int oy[4] = {OOB, OOB, OOB, OOB};
    f32x4 e[4] = {};
    // store offsets of plane a of the tile at (row0, col0) and the request for its first epilogue operand (rsE: that tile's image)
    int soY = 0, soE = 0;     // (ALIGNED) the requested plane's uniform byte offsets into Y and the first epilogue operand
    auto drain_request = [&](int a, int row0, int col0, __amdgpu_buffer_rsrc_t rsE) {
      const int rowoff = (row0 + a) * p.W + col0;   // wave-uniform pixel offset of the plane within the image
      if (ALIGNED) {
        soY = rowoff * (p.y_stride * 4); soE = rowoff * (e_stride * 4);
#pragma unroll
        for (int i = 0; i < 4; ++i)
          if (HAS_E) e[i] = bload4s(rsE, (W4_ABL & 1024) ? (pE[i] & 0xffff) : pE[i], (W4_ABL & 1024) ? 0 : soE);
        return;
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const bool bad = ragged_w && col0 + dcol[i] >= p.W;
        const int offy = pY[i] + rowoff * (p.y_stride * 4), offe = pE[i] + rowoff * (e_stride * 4);
        oy[i] = bad ? OOB : offy;
        if (HAS_E) e[i] = bload4s(rsE, (W4_ABL & 1024) ? (offe & 0xffff) : bad ? OOB : offe, 0);   // (1024: epilogue operand from the image's first 64 KB)
      }
    };
    auto drain_plane = [&](int a, const int (&py)[4]) {
      const int rowoff = (prow0 + a) * p.W + pcol0;
      f32x4 v[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) v[i] = *reinterpret_cast<const f32x4*>(dsrc[i] + a * 4096);   // one LDS round trip for the four units, not four
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        if (ACT == 1) {
          // max(v, v * slope) == (v >= 0 ? v : v * slope) for 0 <= slope <= 1; as an instruction, not as fmaxf: the compiler canonicalises fmaxf's LDS-loaded
          // operand with a v_max v, v of its own (12 instead of 8 instructions per unit)
#pragma unroll
          for (int k = 0; k < 4; ++k) { const float t = v[i][k] * p.slope; float r; asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(v[i][k]), "v"(t)); v[i][k] = r; }
        } else if (ACT == 2) {
#pragma unroll
          for (int k = 0; k < 4; ++k) v[i][k] = v[i][k] >= 0.f ? v[i][k] : v[i][k] * p.slope;
        }
        if (MASK) {
#pragma unroll
          for (int k = 0; k < 4; ++k) v[i][k] *= e[i][k] > 0.f ? 1.f : p.mk_slope;
        } else if (HAS_E) {
          v[i] += e[i];
        }
        if (HAS_L) {
          if (ALIGNED) v[i] += bload4s(rsLp, pL[i], rowoff * (l_stride * 4));
          else { const int offl = pL[i] + rowoff * (l_stride * 4); v[i] += bload4s(rsLp, oy[i] == OOB ? OOB : offl, 0); }
        }
        if (!(W4_ABL & 32) || i == 0) {
          if (ALIGNED) bstore4(rsYp, (W4_ABL & 256) ? (py[i] & 0xffff) : py[i], v[i], (W4_ABL & 256) ? 0 : soY);
          else bstore4(rsYp, (W4_ABL & 256) ? (oy[i] & 0xffff) : oy[i], v[i]);   // (256: every store into the image's first 64 KB)
        }
      }
    };

    int img, y0, x0;
    bool cur_half = only_half;
    if (only_half) tile = htile;
    int himg = 0, hy0 = 0, hx0 = 0;
    if (have_half) tile_origin(htile, himg, hy0, hx0);
    tile_origin(tile, img, y0, x0);
    __amdgpu_buffer_rsrc_t rsXc = img_rsrc(p.X, p.x_stride, img), rsXn = rsXc;
    halo_offsets(y0, x0);
    halo_load(hv0, rsXc, 0);
    halo_load(hv1, rsXc, 1);
    halo_store(hv0);
    LDS_BARRIER();   // (A: halo of chunk 0 staged)
    read_raw();
    halo_load(hv0, rsXc, 2);
    transform();
    write_v(0);
    LDS_BARRIER();   // (B0)
    while (true) {
      const int next = tile + 1;
      const bool next_full = !cur_half && next < tile_end;
      const bool next_half = !cur_half && !next_full && have_half;
      const bool has_next = next_full || next_half;
      // the next tile's origin by stepping (tiles are walked in order: no integer divisions -- ~100 VALU instructions -- per tile)
      int nimg = img, ny0 = y0, nx0 = x0 + 32;
      if (nx0 >= p.W) { nx0 = 0; ny0 += 8; if (ny0 >= p.H) { ny0 = 0; nimg += 1; } }
      if (next_half) { nimg = himg; ny0 = hy0; nx0 = hx0; }
      if (!has_next) nimg = -1;
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        // chunk c + 1 (chunk 0 of the next tile when c == 3): halo registers -> LDS | patch -> transform -> V of the other parity
        f32x4 (&hv)[6] = ((c + 1) & 1) ? hv1 : hv0;
        halo_store(hv);
        if (!W4_EPF) drain_request(c, prow0, pcol0, rsEp);
        __builtin_amdgcn_sched_barrier(0);
        if (W4_DRAIN) {
          drain_plane(c, pY);
          __builtin_amdgcn_sched_barrier(0);
          if (W4_EPF) {   // plane c + 1; behind plane 3 comes plane 0 of THIS tile (its epilogue operand does not depend on the conv's result)
            if (c < 3) drain_request(c + 1, prow0, pcol0, rsEp);
            else drain_request(0, y0, x0, img_rsrc(Ep, e_stride, img));
            __builtin_amdgcn_sched_barrier(0);
          }
        }
        PSTAMP(32 + 8 * c + 0);
        LDS_BARRIER();   // (A)
        PSTAMP(32 + 8 * c + 1);
        read_raw();
        if (c == 1) {   // chunk c + 3 is chunk 0 of the next tile from here on (no next tile: an empty descriptor, every load returns 0)
          rsXn = img_rsrc(p.X, p.x_stride, nimg);
          halo_offsets(ny0, nx0);
        }
        __builtin_amdgcn_sched_barrier(0);
        PSTAMP(32 + 8 * c + 2);
        if (!(W4_ABL & 2)) transform();
        if (!(W4_ABL & 4)) write_v((c + 1) & 1);
        __builtin_amdgcn_sched_barrier(0);
        PSTAMP(32 + 8 * c + 3);
        if (!W4_DRAIN) drain_plane(c, pY);
        __builtin_amdgcn_sched_barrier(0);
        if (!(W4_ABL & 1)) halo_load(hv, c == 0 ? rsXc : rsXn, (c + 3) & 3);
        __builtin_amdgcn_sched_barrier(0);
        PSTAMP(32 + 8 * c + 4);
        if (!(W4_XB && c == 3)) LDS_BARRIER();   // (B)
        PSTAMP(32 + 8 * c + 5);
      }
      if (cur_half) LDS_BARRIER();   // (half tile: the nu < 3 waves' partial results are in the exchange planes; the nu >= 3 waves add theirs)
      LDS_BARRIER();   // this tile's results are in the exchange planes (W4_XB: and the next tile's first V chunk is published)
      PSTAMP(62);
      rsYp = img_rsrc(p.Y, p.y_stride, img); rsEp = img_rsrc(Ep, e_stride, img); rsLp = img_rsrc(Lp, l_stride, img);
      prow0 = y0; pcol0 = x0;
      if (!has_next) {
        // a half tile is always a block's last: only this drain has to leave the other 32 channels' units alone (their stores -- and, ragged geometries, their second
        // operand's loads -- get an out-of-range offset)
        const bool skip = cur_half && (un >> 3) != hc;
        int pYf[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) pYf[i] = skip ? OOB : pY[i];
#pragma unroll
        for (int a = 0; a < 4; ++a) {
          if (!(W4_DRAIN && W4_EPF) || a > 0) drain_request(a, prow0, pcol0, rsEp);
          if (!ALIGNED && skip) { oy[0] = OOB; oy[1] = OOB; oy[2] = OOB; oy[3] = OOB; }
          drain_plane(a, pYf);
        }
        break;
      }
      tile = next_full ? next : htile; cur_half = next_half; img = nimg; y0 = ny0; x0 = nx0;
      rsXc = rsXn;
    }
  } else {
    // ======================================================= CONSUMER ==========================================================
    const int ctile = lane & 15, kk = lane >> 4, cty = ctile >> 3, ctx = ctile & 7;
    const int ns = w4;
    const __amdgpu_buffer_rsrc_t rsW = make_rsrc(p.Wu, 36 * 64 * 64 * 4);
    const float* const vR = sV + ctile * TS + kk * 36;     // fragment base (parity 0, stage 0): + s4 * 144 + 4 q
    const int uoff = ns * 9216 + lane * 16;                // byte offset of this lane's U fragments within a stage (q = 0)
    const int xw = (cty * 32 + 4 * ctx) * 64 + (((4 * ns + kk) ^ ctx) << 2);   // exchange: pixel (row cty, col 4 ctx + b), 16-B unit XOR ctx
    // half tile: wave = (16-channel block ob of the half's two, nu half hv); its U / V fragments are 4 hv .. 4 hv + 4 of every stage (five of the nine)
    const int ob = w4 & 1, hv = w4 >> 1, nsh = 2 * hc + ob;
    const int uoffh = nsh * 9216 + hv * 4096 + lane * 16;
    static_assert(144 % W4_URING == 0, "the ring position of a whole tile's last fragments (and so of the half tile's first, handed over by the wrap-around loads) must be 0");
    f32x4 U[W4_URING];
#pragma unroll
    for (int i = 0; i < W4_URING; ++i) U[i] = bload4(rsW, only_half ? uoffh : uoff, only_half ? ((i / 5) * 36 + (i % 5)) * 1024 : ((i / 9) * 36 + (i % 9)) * 1024);
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    f32x4 acc[36];
    LDS_BARRIER();   // (A of the prologue)
    LDS_BARRIER();   // (B0)
    STAMP(9);
    if (!only_half)
    while (true) {
      const bool has_next = tile + 1 < tile_end;
      const bool to_half = !has_next && have_half;        // the ring's wrap-around loads fetch the half tile's first fragments instead of the next whole tile's
      const int uoffn = to_half ? uoffh : uoff;
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const int par = c & 1;
        f32x4 Vq[W4_VRING];
#pragma unroll
        for (int i = 0; i < W4_VRING - 1; ++i) Vq[i] = *reinterpret_cast<const f32x4*>(vR + par * VBUF + 4 * i);
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) {
#pragma unroll
          for (int q = 0; q < 9; ++q) {
            const int t = (c * 4 + s4) * 9 + q;      // fragment group of the tile, 0..143
            const int g = s4 * 9 + q;                // ... of the chunk
            constexpr int VA = W4_VRING - 1;         // LDS reads run VA groups ahead of their use
            if (g + VA < 36) Vq[(g + VA) % W4_VRING] = *reinterpret_cast<const f32x4*>(vR + par * VBUF + ((g + VA) / 9) * 144 + ((g + VA) % 9) * 4);
            const f32x4 u = U[t % W4_URING], v = Vq[g % W4_VRING];
            const bool first = (c == 0 && s4 == 0);
            acc[4 * q + 0] = __builtin_amdgcn_mfma_f32_16x16x4f32(u.x, v.x, first ? zero4 : acc[4 * q + 0], 0, 0, 0);
            acc[4 * q + 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(u.y, v.y, first ? zero4 : acc[4 * q + 1], 0, 0, 0);
            acc[4 * q + 2] = __builtin_amdgcn_mfma_f32_16x16x4f32(u.z, v.z, first ? zero4 : acc[4 * q + 2], 0, 0, 0);
            acc[4 * q + 3] = __builtin_amdgcn_mfma_f32_16x16x4f32(u.w, v.w, first ? zero4 : acc[4 * q + 3], 0, 0, 0);
            const int tn = (t + W4_URING) % 144;     // (wraps into the next tile: same weights)
            if (t + W4_URING < 144) { if (!(W4_ABL & 16)) U[t % W4_URING] = bload4(rsW, uoff, ((tn / 9) * 36 + (tn % 9)) * 1024); }
            else U[t % W4_URING] = bload4(rsW, uoffn, to_half ? ((tn / 5) * 36 + (tn % 5)) * 1024 : ((tn / 9) * 36 + (tn % 9)) * 1024);
            __builtin_amdgcn_sched_barrier(0);       // (keeps every LDS read two groups ahead of its use)
            if (g == W4_AG) { if (W4_STAMP_A) STAMP(24 + c); if (W4_ANW) BARRIER_NOWAIT(); else LDS_BARRIER(); if (W4_STAMP_A) STAMP(28 + c); }   // (A: the producers have staged the next chunk's halo; they arrive within a group or two)
          }
        }
        STAMP(c);
        if (!(W4_XB && c == 3)) LDS_BARRIER();   // V of the next chunk published; everyone is done reading this chunk's
        STAMP(16 + c);
      }
      // At M A in registers; one output row a of every Winograd tile per round -> exchange buffer
#pragma unroll
      for (int nu = 0; nu < 6; ++nu) at6(acc[WP(0, nu)], acc[WP(1, nu)], acc[WP(2, nu)], acc[WP(3, nu)], acc[WP(4, nu)], acc[WP(5, nu)]);
#pragma unroll
      for (int a = 0; a < 4; ++a) {
        float* const xb = sX + a * 4096;
        at6(acc[WP(a, 0)], acc[WP(a, 1)], acc[WP(a, 2)], acc[WP(a, 3)], acc[WP(a, 4)], acc[WP(a, 5)]);
#pragma unroll
        for (int b = 0; b < 4; ++b) {
          f32x4 v = acc[WP(a, b)];
          *reinterpret_cast<f32x4*>(xb + xw + b * 64) = v;
        }
      }
      STAMP(8);
      LDS_BARRIER();   // the tile's results are in the exchange buffer (the producers drain it during the next tile's first chunk)
      STAMP(20);
      if (!has_next) break;
      tile += 1;
    }
    if (have_half) {
      // ============================================ the half tile: 18 positions x 16 channels x 16 Winograd tiles per wave ============================================
      // accumulator j = place - 18 hv = 3 xi + nu % 3.  hv = 0: fragments 0..3 whole (j = 4 ql + e), fragment 4's first two values (j = 16, 17);
      // hv = 1: fragment 4's last two values (j = 0, 1), fragments 5..8 whole (j = 2 + 4 (ql - 1) + e)
      const float* const vRh = vR + 16 * hv;
      auto half_mfmas = [&](auto HV) {
        constexpr int hvc = decltype(HV)::value;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const int par = c & 1;
          f32x4 Vq[W4_VRING];
#pragma unroll
          for (int i = 0; i < W4_VRING - 1; ++i) Vq[i] = *reinterpret_cast<const f32x4*>(vRh + par * VBUF + (i / 5) * 144 + (i % 5) * 4);
#pragma unroll
          for (int s4 = 0; s4 < 4; ++s4) {
#pragma unroll
            for (int ql = 0; ql < 5; ++ql) {
              const int t = (c * 4 + s4) * 5 + ql;     // fragment of the half tile, 0..79
              const int g = s4 * 5 + ql;               // ... of the chunk
              constexpr int VA = W4_VRING - 1;
              if (g + VA < 20) Vq[(g + VA) % W4_VRING] = *reinterpret_cast<const f32x4*>(vRh + par * VBUF + ((g + VA) / 5) * 144 + ((g + VA) % 5) * 4);
              const f32x4 u = U[t % W4_URING], v = Vq[g % W4_VRING];
              const bool first = (c == 0 && s4 == 0);
              if (hvc == 0 ? ql < 4 : ql > 0) {
                const int j0 = hvc == 0 ? 4 * ql : 4 * ql - 2;
                acc[j0 + 0] = __builtin_amdgcn_mfma_f32_16x16x4f32(u.x, v.x, first ? zero4 : acc[j0 + 0], 0, 0, 0);
                acc[j0 + 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(u.y, v.y, first ? zero4 : acc[j0 + 1], 0, 0, 0);
                acc[j0 + 2] = __builtin_amdgcn_mfma_f32_16x16x4f32(u.z, v.z, first ? zero4 : acc[j0 + 2], 0, 0, 0);
                acc[j0 + 3] = __builtin_amdgcn_mfma_f32_16x16x4f32(u.w, v.w, first ? zero4 : acc[j0 + 3], 0, 0, 0);
              } else if (hvc == 0) {
                acc[16] = __builtin_amdgcn_mfma_f32_16x16x4f32(u.x, v.x, first ? zero4 : acc[16], 0, 0, 0);
                acc[17] = __builtin_amdgcn_mfma_f32_16x16x4f32(u.y, v.y, first ? zero4 : acc[17], 0, 0, 0);
              } else {
                acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(u.z, v.z, first ? zero4 : acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(u.w, v.w, first ? zero4 : acc[1], 0, 0, 0);
              }
              const int tn = t + W4_URING;
              if (tn < 80) U[t % W4_URING] = bload4(rsW, uoffh, ((tn / 5) * 36 + (tn % 5)) * 1024);
              __builtin_amdgcn_sched_barrier(0);
              if (g == W4_HAG) BARRIER_NOWAIT();       // (A)
            }
          }
          if (c < 3) LDS_BARRIER();                    // (B)
        }
        // first pass of At M A (over xi) for the wave's three nu: T[a][n] -> acc[3 a + n]
#pragma unroll
        for (int n = 0; n < 3; ++n) at6(acc[n], acc[3 + n], acc[6 + n], acc[9 + n], acc[12 + n], acc[15 + n]);
        // second pass: this nu half's share of at6's four sums -- (T0 + s12) | s34,  d12 | 2 d34,  s12 | 4 s34,  d12 | fma(8, d34, T5) -- so that "share 0 + share 1"
        // is at6's own association (2 d34 and 4 s34 are exact, so adding them is what at6's fma does)
#pragma unroll
        for (int a = 0; a < 4; ++a) {
          const f32x4 t0 = acc[3 * a], t1 = acc[3 * a + 1], t2 = acc[3 * a + 2];
          if (hvc == 0) {
            const f32x4 s12 = t1 + t2, d12 = t1 - t2;
            acc[3 * a] = t0 + s12; acc[3 * a + 1] = d12; acc[3 * a + 2] = s12; acc[12 + a] = d12;
          } else {
            const f32x4 s34 = t0 + t1, d34 = t0 - t1;
            acc[3 * a] = s34; acc[3 * a + 1] = 2.f * d34; acc[3 * a + 2] = 4.f * s34; acc[12 + a] = __builtin_elementwise_fma(f32x4{8.f, 8.f, 8.f, 8.f}, d34, t2);
          }
        }
      };
      if (hv == 0) half_mfmas(std::integral_constant<int, 0>{}); else half_mfmas(std::integral_constant<int, 1>{});
      // shares: b = 0..2 in acc[3 a + b], b = 3 in acc[12 + a]
      const int xwh = (cty * 32 + 4 * ctx) * 64 + (((4 * nsh + kk) ^ ctx) << 2);
      if (hv == 0) {
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
          for (int b = 0; b < 4; ++b) *reinterpret_cast<f32x4*>(sX + a * 4096 + xwh + b * 64) = b < 3 ? acc[3 * a + b] : acc[12 + a];
      }
      LDS_BARRIER();
      if (hv == 1) {
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
          for (int b = 0; b < 4; ++b) {
            float* const q = sX + a * 4096 + xwh + b * 64;
            *reinterpret_cast<f32x4*>(q) = *reinterpret_cast<const f32x4*>(q) + (b < 3 ? acc[3 * a + b] : acc[12 + a]);
          }
      }
      LDS_BARRIER();   // the half tile's results are in the exchange planes
    }
  }
#if W4_CLK
  if (threadIdx.x == 0 && blockIdx.x < 1024) {
    g_w4_clk[2 * blockIdx.x] = __builtin_amdgcn_s_memtime() - clk0;
    g_w4_clk[2 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime() - rt0;
  }
#endif
#ifdef LFSR_CONV_DIAG
  if (dbgbuf && threadIdx.x == 0)
    for (int k = 0; k < 32; ++k) dbgbuf[blockIdx.x * 64 + k] = (float)seg[k];
  if (dbgbuf && threadIdx.x == 256)
    for (int k = 32; k < 64; ++k) dbgbuf[blockIdx.x * 64 + k] = (float)seg[k];
#endif
}

// U = G g G^t per (n, k) -> [s = k/4][ns = n/16][q = p/4][lane = 16 (k%4) + n%16][e = p%4], p = WP(xi, nu)
__device__ __forceinline__ void emit_wino4(const double (&g)[3][3], int n, int k, float* __restrict__ out) {
  const double G[6][3] = {{1.0 / 4, 0.0, 0.0}, {-1.0 / 6, -1.0 / 6, -1.0 / 6}, {-1.0 / 6, 1.0 / 6, -1.0 / 6},
                          {1.0 / 24, 1.0 / 12, 1.0 / 6}, {1.0 / 24, -1.0 / 12, 1.0 / 6}, {0.0, 0.0, 1.0}};
  double tmp[6][3];
#pragma unroll
  for (int a = 0; a < 6; ++a)
#pragma unroll
    for (int c = 0; c < 3; ++c) tmp[a][c] = G[a][0] * g[0][c] + G[a][1] * g[1][c] + G[a][2] * g[2][c];
  const int s = k >> 2, kq = k & 3, nsl = n >> 4, m = n & 15;
#pragma unroll
  for (int a = 0; a < 6; ++a)
#pragma unroll
    for (int b = 0; b < 6; ++b) {
      const double u = tmp[a][0] * G[b][0] + tmp[a][1] * G[b][1] + tmp[a][2] * G[b][2];
      const int pp = WP(a, b);
      const int ln = kq * 16 + m;
      out[((((s * 4 + nsl) * 9 + (pp >> 2)) * 64 + ln) << 2) + (pp & 3)] = (float)u;
    }
}

// ... from the direct pack [tap][n][k]
__global__ __launch_bounds__(256) void k_pack_wino4(const float* __restrict__ direct, float* __restrict__ out) {
  const int i = blockIdx.x * 256 + threadIdx.x;   // (n, k)
  if (i >= 64 * 64) return;
  const int n = i >> 6, k = i & 63;
  double g[3][3];
#pragma unroll
  for (int t = 0; t < 9; ++t) g[t / 3][t % 3] = (double)direct[(t * 64 + n) * 64 + k];
  emit_wino4(g, n, k, out);
}

// ... from the raw (O = 64, C = 64, 3, 3) weight, writing the direct pack [tap][n][k] as well: one launch per weight where the training
// step's repack used two.  TR: the transposed, tap-flipped form the data gradient reads (n = the forward's input channel, k = its output channel)
template <bool TR>
__global__ __launch_bounds__(256) void k_pack_conv3_raw(const float* __restrict__ w, float* __restrict__ direct, float* __restrict__ out) {
  const int i = blockIdx.x * 256 + threadIdx.x;   // (n, k)
  if (i >= 64 * 64) return;
  const int n = i >> 6, k = i & 63;
  double g[3][3];
#pragma unroll
  for (int t = 0; t < 9; ++t) {
    const float v = TR ? w[(k * 64 + n) * 9 + (8 - t)] : w[(n * 64 + k) * 9 + t];
    direct[(t * 64 + n) * 64 + k] = v;
    g[t / 3][t % 3] = (double)v;
  }
  emit_wino4(g, n, k, out);
}

// ... for a whole table of weights in one launch (blockIdx.y = the weight)
__global__ __launch_bounds__(256) void k_pack_conv3_raw_batch(const LfsrPackDesc* __restrict__ tab) {
  const LfsrPackDesc d = tab[blockIdx.y];
  const int i = blockIdx.x * 256 + threadIdx.x;   // (n, k)
  if (i >= 64 * 64) return;
  const int n = i >> 6, k = i & 63;
  double g[3][3];
#pragma unroll
  for (int t = 0; t < 9; ++t) {
    const float v = d.flip ? d.src[(k * 64 + n) * 9 + (8 - t)] : d.src[(n * 64 + k) * 9 + t];
    d.dst[(t * 64 + n) * 64 + k] = v;
    g[t / 3][t % 3] = (double)v;
  }
  emit_wino4(g, n, k, d.dst2);
}

}  // namespace

int lfsr_pack_conv3_raw_wino4_batch(const LfsrPackDesc* table_dev, int n, hipStream_t st) {
  if (!table_dev || n <= 0) return n == 0 ? LFSR_OK : LFSR_E_ARG;
  hipLaunchKernelGGL(k_pack_conv3_raw_batch, dim3(16, (unsigned)n), dim3(256), 0, st, table_dev);
  LFSR_CHECK_LAUNCH();
  return LFSR_OK;
}

int lfsr_pack_conv3_raw_wino4(const float* w_raw, float* direct_out, float* wino4_out, int transposed, hipStream_t st) {
  if (!w_raw || !direct_out || !wino4_out) return LFSR_E_ARG;
  if (transposed) hipLaunchKernelGGL(k_pack_conv3_raw<true>, dim3(16), dim3(256), 0, st, w_raw, direct_out, wino4_out);
  else hipLaunchKernelGGL(k_pack_conv3_raw<false>, dim3(16), dim3(256), 0, st, w_raw, direct_out, wino4_out);
  LFSR_CHECK_LAUNCH();
  return LFSR_OK;
}

int lfsr_pack_wino4(const float* direct_packed, float* out, hipStream_t st) {
  if (!direct_packed || !out) return LFSR_E_ARG;
  hipLaunchKernelGGL(k_pack_wino4, dim3(16), dim3(256), 0, st, direct_packed, out);
  LFSR_CHECK_LAUNCH();
  return LFSR_OK;
}

// LFSR_E_ARG = geometry not covered (operands of 1 GiB and more): the caller falls back to the F(2x2,3x3) kernel
int lfsr_conv3x3_wino4_launch(const float* x, int x_stride, int x_choff, const float* w_wino4, float* y, int y_stride, int y_choff,
                              const float* r1, int r1_stride, int r1_choff, const float* r2, int r2_stride, int r2_choff,
                              const float* mk, int mk_stride, int mk_choff, float mk_slope,
                              int n_img, int h, int w, float slope, hipStream_t st) {
  static std::atomic<bool> attr_set[64];
  static std::atomic<int> cus[64];
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return LFSR_E_ARG;
  if (!attr_set[dev]) {
#define W4_FNS(A, G) reinterpret_cast<const void*>(k_conv3x3_wino4<false, false, false, A, G>), reinterpret_cast<const void*>(k_conv3x3_wino4<false, true, false, A, G>), \
                     reinterpret_cast<const void*>(k_conv3x3_wino4<false, true, true, A, G>), reinterpret_cast<const void*>(k_conv3x3_wino4<true, true, false, A, G>), \
                     reinterpret_cast<const void*>(k_conv3x3_wino4<true, true, true, A, G>)
    const void* fns[30] = {W4_FNS(0, false), W4_FNS(1, false), W4_FNS(2, false), W4_FNS(0, true), W4_FNS(1, true), W4_FNS(2, true)};
#undef W4_FNS
    for (const void* f : fns) {
      hipError_t e = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, SMEM_BYTES);
      if (e != hipSuccess) return LFSR_HIP_ERR(e);
    }
    int v = 0;
    cus[dev] = (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) ? v : 256;
    attr_set[dev] = true;
  }
  int ms = x_stride > y_stride ? x_stride : y_stride;
  if (r1 && r1_stride > ms) ms = r1_stride;
  if (r2 && r2_stride > ms) ms = r2_stride;
  if (mk && mk_stride > ms) ms = mk_stride;
  if ((long long)n_img * h * w * ms * 4 >= (1LL << 30)) return LFSR_E_ARG;
  Wino4Args p{};
  p.X = x; p.x_stride = x_stride; p.x_choff = x_choff; p.x_bytes = (int)((long long)n_img * h * w * x_stride * 4);
  p.Wu = w_wino4;
  p.Y = y; p.y_stride = y_stride; p.y_choff = y_choff;
  p.R1 = r1; p.r1_stride = r1_stride; p.r1_choff = r1_choff; p.R2 = r2; p.r2_stride = r2_stride; p.r2_choff = r2_choff;
  p.Mk = mk; p.mk_stride = mk_stride; p.mk_choff = mk_choff; p.mk_slope = mk_slope;
  const long long npix4 = (long long)n_img * h * w * 4;
  p.y_bytes = (int)(npix4 * y_stride); p.r1_bytes = r1 ? (int)(npix4 * r1_stride) : 0; p.r2_bytes = r2 ? (int)(npix4 * r2_stride) : 0;
  p.mk_bytes = mk ? (int)(npix4 * mk_stride) : 0;
#ifdef LFSR_CONV_DIAG
  p.dbg = g_lfsr_diag_buf;
#endif
  p.n_img = n_img; p.H = h; p.W = w; p.tiles_y = (h + 7) / 8; p.tiles_x = (w + 31) / 32; p.slope = slope;
  const long long nt = (long long)n_img * p.tiles_y * p.tiles_x;
  if (nt <= 0 || nt > 0x7fffffffLL) return LFSR_E_ARG;
  p.ntiles = (int)nt;
  const int slots = cus[dev];
  unsigned grid = (unsigned)(nt < slots ? nt : slots);
  // A last round that would keep at most half of the CUs busy is run as half tiles on twice as many (3200 tiles on 256 CUs: 12 rounds + 128 tiles -> 12 + 256 halves;
  // 800: 3 + 64 halves; 100: 200 halves).  Which tiles are split depends on the batch; their bits do not (see the kernel).
  {
    const long long fpb = nt / slots, rem = nt % slots;
    if (rem > 0 && 2 * rem <= slots && !lfsr_sel("LFSR_CONV_NOHALF")) { p.full_per_block = (int)fpb; p.nhalf = (int)rem; grid = (unsigned)(fpb > 0 ? slots : 2 * rem); }
  }
  if (!mk && !r1 && r2) { p.R1 = r2; p.r1_stride = r2_stride; p.r1_choff = r2_choff; p.r1_bytes = p.r2_bytes; p.R2 = nullptr; p.r2_bytes = 0; }   // a lone residual is the first operand
  const int act = slope == 1.f ? 0 : (slope >= 0.f && slope < 1.f ? 1 : 2);
  const bool aligned = h % 8 == 0 && w % 32 == 0;
#define W4_GO2(M, E, L, G) do { \
    if (act == 0) hipLaunchKernelGGL((k_conv3x3_wino4<M, E, L, 0, G>), dim3(grid), dim3(512), SMEM_BYTES, st, p); \
    else if (act == 1) hipLaunchKernelGGL((k_conv3x3_wino4<M, E, L, 1, G>), dim3(grid), dim3(512), SMEM_BYTES, st, p); \
    else hipLaunchKernelGGL((k_conv3x3_wino4<M, E, L, 2, G>), dim3(grid), dim3(512), SMEM_BYTES, st, p); } while (0)
#define W4_GO(M, E, L) do { if (aligned) W4_GO2(M, E, L, true); else W4_GO2(M, E, L, false); } while (0)
  if (mk && p.R1) W4_GO(true, true, true);
  else if (mk) W4_GO(true, true, false);
  else if (p.R1 && p.R2) W4_GO(false, true, true);
  else if (p.R1) W4_GO(false, true, false);
  else W4_GO(false, false, false);
#undef W4_GO
#undef W4_GO2
  LFSR_CHECK_LAUNCH();
  return LFSR_OK;
}
