// Per-view 3x3 conv, zero pad 1, 64 -> 64 channels, VCL layout -- Winograd F(2x2, 3x3) on the fp32 MFMA pipe.
// Reference: the MacPI convs "k3, dilation A, padding A" of model/SR/DistgSSR.py:22,47,64,79-83,101 (per-view 3x3 in VCL),
// EPIT.py:24-32,136-142 / LFT.py:36-46 (Conv3d(1,3,3)).
//
//   Y = At [ sum_c (G g G^t) . (Bt d B) ] A     per 2x2 output tile, d = its 4x4 input patch, 16 transform positions p = (xi, nu)
//
// The 16 positions are 16 independent [tiles x 64] x [64 x 64] GEMMs -> 2.25x fewer MFMA flops than the 9-tap direct form,
// all arithmetic exact fp32 (products and sums; the transform coefficients are 0, +-1, +-1/2).
//
// Persistent 512-thread block (8 waves) per CU walks 8-row x 32-column output tiles (same tiling as conv3x3_halo.hip):
//  * the (8+2) x (32+2) input halo sits in LDS once, 64 floats per pixel, 16-B chunks XOR-swizzled by (pixel>>1) so that
//    the stride-2-pixel patch reads of 32 Winograd tiles are conflict-free ds_read_b128;
//  * wave (mg, xi): mg = 4-row half of the tile = 2 x 16 = 32 Winograd tiles (the 32 A-rows of v_mfma_f32_32x32x2_f32),
//    xi = one row of the 4x4 transform domain.  Per 8-channel stage it reads 2 patch rows x 4 columns (8 x b128), forms the
//    four V[xi][nu] fragments in registers (8 float4 add/sub), and runs 4 positions x 2 N-tiles x 4 = 32 MFMAs
//    (8 accumulators = 128 registers);
//  * transformed weights U stream through a 3-deep LDS ring of 16-KB units (stage j, N-tile nt: 16 positions x 32 n x 8 k,
//    packed on the host side in exactly the fragment order), unit u+1 written at the start of unit u and published by ONE
//    barrier in the middle of unit u's MFMA stream;
//  * K is the outer loop, so the channels of a finished stage are dead in the halo: the NEXT tile's halo is streamed in place,
//    16 channels (64 B per pixel) at a time, through 3 float4 registers per thread -- no second halo buffer, no seam load;
//  * epilogue: per output column parity b the waves write r_xi[b] = sum_nu M[xi][nu] A[nu][b] to LDS (over the idle weight
//    ring), the cross-xi sum  Y[a][b] = sum_xi At[a][xi] r_xi[b]  is done by the reader, which stores 16 B per lane,
//    256 B contiguous per pixel, with LeakyReLU / LeakyReLU' mask / residuals applied.
// LDS: 340 x 256 B halo + 64 KB ring/exchange = 152,576 B, one block per CU, 2 waves per SIMD.
#include <stdlib.h>

#include "lfsr_internal.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

#ifndef WINO_ABL
#define WINO_ABL 0   // diagnostic timing builds: 1 = no K-loop barriers, 2 = no halo streaming, 4 = no weight staging, 8 = no patch reads, 16 = half of them, 32 = patch reads from contiguous addresses
#endif
#ifdef LFSR_CONV_DIAG
// diagnostic build only: wave 0 accumulates s_memtime deltas per segment, written to the buffer passed as R2
#define STAMP(k) do { if (wave == 0) { long long t_ = clock64(); seg[k] += t_ - tprev; tprev = t_; } } while (0)
#else
#define STAMP(k) do { } while (0)
#endif

namespace {

constexpr int TR = 8, TC = 32;
constexpr int HALO_PIX = (TR + 2) * (TC + 2);      // 340
constexpr int HALO_FLOATS = HALO_PIX * 64;         // 21760
constexpr int UNIT_FLOATS = 4096;                  // 16 positions x 2 halves x 32 n x 4 k
constexpr int XCH_FLOATS = 8 * 32 * 64;            // exchange: 8 waves x 32 tiles x 64 channels (covers the 3-unit ring)
constexpr int SMEM_BYTES = (HALO_FLOATS + XCH_FLOATS + 256) * 4;   // 153600 (+1 KB landing zone for the unused halo slots)

struct WinoArgs {
  const float* X; int x_stride; int x_choff;
  int x_bytes, y_bytes, r1_bytes, r2_bytes, mk_bytes;   // true byte span of each operand (n_img * h * w * stride * 4 < 2 GiB; 0 = absent): descriptor extents
  float* dbg;                                           // (LFSR_CONV_DIAG builds: stamp buffer; else unused)
  const float* Wu;   // [8 stages][2 nt][16 p][2 half][32 n][4]   (lfsr_pack_wino)
  float* Y; int y_stride; int y_choff;
  const float* R1; int r1_stride; int r1_choff;
  const float* R2; int r2_stride; int r2_choff;
  const float* Mk; int mk_stride; int mk_choff; float mk_slope;
  int n_img, H, W, tiles_y, tiles_x, ntiles;
  int tile_begin;   // channel-split blocks: first tile of their range (set by the kernel)
  int nbody;        // number of persistent body blocks in the launch
  float slope;
};

__device__ __forceinline__ float4 f4add(float4 a, float4 b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }
__device__ __forceinline__ float4 f4sub(float4 a, float4 b) { return make_float4(a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w); }
__device__ __forceinline__ float4 f4fma(float s, float4 a, float4 b) {   // s = +-1: exact
  return make_float4(fmaf(s, a.x, b.x), fmaf(s, a.y, b.y), fmaf(s, a.z, b.z), fmaf(s, a.w, b.w));
}

// LDS-only barrier (global stores/loads stay in flight); the sched_barriers pin the MFMA stream around it
#define LDS_BARRIER() do { __builtin_amdgcn_sched_barrier(0); asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); __builtin_amdgcn_sched_barrier(0); } while (0)
// opaque copy: blocks loop-invariant hoisting / CSE of everything derived from x (address arithmetic is recomputed next to
// its use instead of being kept -- and spilled -- across the 16 unrolled units)
#define OPAQUE(x) asm volatile("" : "+v"(x))

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int OOB = (int)0x80000000u;   // byte offset beyond every descriptor's range: loads return 0, stores are dropped

__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* base, int bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, bytes, 0x00020000);
}
__device__ __forceinline__ float4 bload(__amdgpu_buffer_rsrc_t r, int voff, int soff) {
  const f32x4 v = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0));
  return make_float4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ void bstore(__amdgpu_buffer_rsrc_t r, int voff, float4 f) {
  f32x4 v;
  v.x = f.x; v.y = f.y; v.z = f.z; v.w = f.w;
  __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), r, voff, 0, 0);
}

// All global traffic goes through buffer descriptors with 32-bit byte offsets (the launcher checks every operand spans
// < 2 GiB): zero padding and ragged edges are out-of-range offsets instead of branches, so the K loop has no control flow.
// NHALF = true: tail launch.  When the tile count is not a multiple of the CU count the leftover tiles are given to TWO blocks
// each, block (2t + nh) computing output channels [32 nh, 32 nh + 32) of tile t: one N-tile per wave, 8 units instead of 16
// (half the MFMAs, ~0.55 of a tile time), one tile per block.
template <bool MASK, bool NHALF>
__device__ __forceinline__ void wino_tile_loop(WinoArgs p, const int bid, const int nblk) {   // bid of nblk blocks of this role
  constexpr int NU = NHALF ? 8 : 16;           // units per tile
  constexpr int NOUT = NHALF ? 2 : 4;          // epilogue outputs (float4) per thread and round
  const int nh = NHALF ? (bid & 1) : 0;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* sH = smem;                  // halo
  float* sR = smem + HALO_FLOATS;    // weight ring (3 units) / epilogue exchange

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int half = lane >> 5, l31 = lane & 31;
  const int mg = wave >> 2, xi = wave & 3;
  const int ty = l31 >> 4, tx = l31 & 15;

#ifdef LFSR_CONV_DIAG
  long long seg[32] = {};
  long long tprev = clock64();
  float* dbgbuf = p.dbg;
#endif
  const __amdgpu_buffer_rsrc_t rsX = make_rsrc(p.X, p.x_bytes), rsW = make_rsrc(p.Wu, 16 * UNIT_FLOATS * 4), rsY = make_rsrc(p.Y, p.y_bytes);
  const __amdgpu_buffer_rsrc_t rsE = make_rsrc(MASK ? p.Mk : p.R1, MASK ? p.mk_bytes : p.r1_bytes);   // prefetched epilogue operand
  const __amdgpu_buffer_rsrc_t rsL = make_rsrc(MASK ? p.R1 : p.R2, MASK ? p.r1_bytes : p.r2_bytes);   // late epilogue operand
  const int e_stride = MASK ? p.mk_stride : p.r1_stride, e_choff = MASK ? p.mk_choff : p.r1_choff;
  const int l_stride = MASK ? p.r1_stride : p.r2_stride, l_choff = MASK ? p.r1_choff : p.r2_choff;
  const bool has_e = (MASK ? p.Mk : p.R1) != nullptr, has_l = (MASK ? p.R1 : p.R2) != nullptr;

  // ---- patch addressing: t[jj] = d[ra][jj] + sg * d[rb][jj]  (row xi of Bt d) -------------------------------------
  const int ra = xi == 0 ? 0 : (xi == 2 ? 2 : 1);
  const int rb = xi == 3 ? 3 : (xi == 2 ? 1 : 2);
  const float sg = xi == 1 ? 1.f : -1.f;
  const int pixA = (4 * mg + 2 * ty + ra) * (TC + 2) + 2 * tx;   // halo pixel of patch column 0
  const int pixB = (4 * mg + 2 * ty + rb) * (TC + 2) + 2 * tx;
  // swizzle key of a halo pixel in column c: ((c >> 1) & 15) ^ ((c & 1) << 2) -- independent of the row, so the 16-lane
  // groups of ds_read_b128 ({0-3,12-15,20-27}, {4-11,16-19,28-31}, +32: tiles tx = 0..15 of either tile row) hit 16
  // distinct 16-B slots; the parity bit keeps the 8-lane groups of the ds_write_b128 halo stores (2 pixels x 4 chunks) apart
  int k0 = (tx & 15) ^ half, k1 = ((tx + 1) & 15) ^ half;   // patch columns 0 (1: ^4) and 2 (3: ^4), with the lane half folded in
  int offA = pixA * 64, offB = pixB * 64;                       // float offsets into the halo
  int offBf = ((8 * xi + half) * 32 + l31) * 4;                 // B fragments: position p = 4 xi + nu -> + nu * 256 floats
  int vtid = tid;   // laundered copy of the thread id for the per-slot arithmetic

  // ---- halo streaming slots: slice g = logical chunks 4g..4g+3; slot i -> (pixel, chunk) = (idx >> 2, idx & 3), idx = tid + 512 i
  auto tile_origin = [&](int t, int& img, int& y0, int& x0) {
    int txx = t % p.tiles_x; int q = t / p.tiles_x;
    int tyy = q % p.tiles_y; img = q / p.tiles_y;
    y0 = tyy * TR; x0 = txx * TC;
  };
  // byte offsets of the 3 slots' (pixel, chunks 4g..4g+3 at + 64 g), OOB outside the image or when there is no such tile:
  // scalar tile base + a per-thread part in 24-bit multiplies, no branches (this runs in the shadow of unit 0's MFMAs)
  auto halo_offsets = [&](int* hx, bool valid, int img, int y0, int x0) {
    const int base = (((img * p.H + y0 - 1) * p.W + x0 - 1) * p.x_stride + p.x_choff) * 4;   // wave-uniform
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      const int px = (vtid + 512 * i) >> 2, cq = vtid & 3;
      const int r = __mul24(px, 1928) >> 16;   // px / 34 for px < 384
      const int c = px - r * (TC + 2);
      const int yy = y0 + r - 1, xx = x0 + c - 1;
      const bool ok = valid && px < HALO_PIX && (unsigned)yy < (unsigned)p.H && (unsigned)xx < (unsigned)p.W;
      hx[i] = ok ? base + __mul24(__mul24(r, p.W) + c, p.x_stride * 4) + cq * 16 : OOB;
    }
  };
  // LDS float offset of halo slot i for slice 0; slice g flips the two 64-B bits: (4g + cq) ^ key = ((4g) ^ (key & 12)) | ((cq ^ key) & 3)
  auto halo_slot_base = [&](int i) -> int {
    const int px = (vtid + 512 * i) >> 2, cq = vtid & 3;
    const int col = px - (__mul24(px, 1928) >> 16) * (TC + 2);
    const int key = ((col >> 1) & 15) ^ ((col & 1) << 2);
    const int b = px * 64 + (((cq ^ key) & 3) << 2) + ((key & 12) << 2);
    return (i < 2 || px < HALO_PIX) ? b : HALO_FLOATS + XCH_FLOATS + (vtid & 63) * 4;   // slots 340..383 do not exist: landing zone
  };
  auto halo_store = [&](int g, int i, float4 v) { *reinterpret_cast<float4*>(smem + (halo_slot_base(i) ^ (16 * g))) = v; };
  // 2 rows x 4 columns of the patch, channels 8j + 4 half .. + 3; the two rows are requested half a unit apart (a burst of
  // 8 x 8 ds_read_b128 per CU holds up the B-fragment reads queued behind it -- LDS returns in order -- 4 do not)
  auto raw_read = [&](int j, int row, float4* raw) {
    const int o0 = ((2 * j) ^ k0) << 2, o1 = o0 ^ 16, o2 = ((2 * j) ^ k1) << 2, o3 = o2 ^ 16;
    const float* hR = (WINO_ABL & 32) ? sH + (tid & 63) * 4 + row * 1024 + j * 2048 - o0 : sH + (row == 0 ? offA : offB);   // (timing ablation: contiguous 16 B per lane)
    raw[4 * row + 0] = *reinterpret_cast<const float4*>(hR + o0);
    raw[4 * row + 1] = *reinterpret_cast<const float4*>(hR + 64 + o1);
    raw[4 * row + 2] = *reinterpret_cast<const float4*>(hR + 128 + o2);
    raw[4 * row + 3] = *reinterpret_cast<const float4*>(hR + 192 + o3);
  };
  auto transform = [&](const float4* raw, float4* V) {   // V[nu] = row xi of (Bt d B)
    float4 t0 = f4fma(sg, raw[4], raw[0]), t1 = f4fma(sg, raw[5], raw[1]);
    float4 t2 = f4fma(sg, raw[6], raw[2]), t3 = f4fma(sg, raw[7], raw[3]);
    V[0] = f4sub(t0, t2); V[1] = f4add(t1, t2); V[2] = f4sub(t2, t1); V[3] = f4sub(t1, t3);
  };
  // epilogue output i of round b: (row R, column 2 txo + b, 16-B chunk c); pixel index or -1 outside the image
  auto out_pixel = [&](int b, int i, int img, int y0, int x0) -> int {
    const int q = (vtid + 512 * i) >> (NHALF ? 3 : 4);
    const int R = q >> 4, col = 2 * (q & 15) + b;
    return (y0 + R < p.H && x0 + col < p.W) ? (img * p.H + y0 + R) * p.W + x0 + col : -1;
  };

  auto chunk_of = [&](int i) -> int { const int o = vtid + 512 * i; return NHALF ? (o & 7) + 8 * nh : (o & 15); };   // 16-B channel chunk of output i
  int tile = NHALF ? p.tile_begin + (bid >> 1) : bid;
  int img, y0, x0;
  tile_origin(tile, img, y0, x0);
  // the persistent stride gridDim.x as (images, tile rows, tile columns): the next tile's origin by carries, no divisions
  const int g_tx = nblk % p.tiles_x, g_q = nblk / p.tiles_x;
  const int g_ty = g_q % p.tiles_y, g_img = g_q / p.tiles_y;

  // ---- prologue: whole halo of the first tile, unit 0 into the ring, unit 1 in registers --------------------------
  float4 hv[3];
  int hx[3];
  halo_offsets(hx, true, img, y0, x0);
#pragma unroll
  for (int g = 0; g < 4; ++g) {
#pragma unroll
    for (int i = 0; i < 3; ++i) hv[i] = bload(rsX, hx[i], g * 64);
#pragma unroll
    for (int i = 0; i < 3; ++i) halo_store(g, i, hv[i]);
  }
  float4 wr0, wr1;   // staged weight unit (2 float4 per thread)
  float4 wx0, wx1, wy0, wy1;   // units 1 and 2 of the next tile, fetched during unit 15: no load is queued right behind the epilogue's stores
  {   // units 0, 1, 2 into the three ring slots
    float4* dst = reinterpret_cast<float4*>(sR);
#pragma unroll
    for (int k = 0; k < 6; ++k) dst[tid + 512 * k] = bload(rsW, tid * 16, NHALF ? ((k >> 1) * 2 + nh) * UNIT_FLOATS * 4 + (k & 1) * 8192 : 8192 * k);
  }
  __syncthreads();
  float4 raw[8];
  raw_read(0, 0, raw); raw_read(0, 1, raw);
  float4 B01[2], B23[2], nB01[2];
  float4 nV[4];   // fragments of the next stage (stage 0 of a tile: formed in the prologue / at the seam)
  transform(raw, nV);
  B01[0] = *reinterpret_cast<const float4*>(sR + offBf);
  B01[1] = *reinterpret_cast<const float4*>(sR + offBf + 256);

  f32x16 acc[4][2];
  const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};

  while (true) {
    const int next = tile + nblk;
    const bool has_next = !NHALF && next < p.ntiles;
    int nimg = 0, ny0 = 0, nx0 = 0;
    int opix0[NOUT], opix1[NOUT];   // output pixels of the two epilogue rounds and round 0's prefetched operand: set up during the
    float4 res[NOUT];               // last but one unit (its MFMAs cover the address arithmetic and the load latency; the patch registers are free)
    float4 V[4];
    if (WINO_ABL & 8) { V[0] = raw[0]; V[1] = raw[1]; V[2] = raw[2]; V[3] = raw[3]; }

#pragma unroll
    for (int u = 0; u < NU; ++u) {
      const int j = NHALF ? u : u >> 1, nt = NHALF ? 0 : u & 1;   // stage, accumulator column (NHALF: the wave's only N-tile)
      const bool sfirst = NHALF || nt == 0, slast = NHALF || nt == 1;   // first / last unit of its stage
      OPAQUE(k0); OPAQUE(k1); OPAQUE(offA); OPAQUE(offB); OPAQUE(offBf); OPAQUE(vtid);
      const float* bBase = sR + offBf;
      // S1: unit u+1 (in registers since unit u-1) -> ring; start fetching unit u+2 (unit 0 of the next tile at u = 14).
      // Units 0, 1, 2 are already in the ring when a tile starts (written at the seam from loads issued BEFORE the epilogue's
      // stores): vmcnt retires in order, so the first load issued behind the store burst (unit 3, at u = 0) is not
      // waited for until u = 2, by when the stores have drained
      if (u >= 2 && u < NU - 1 && !(WINO_ABL & 4)) {
        float4* dst = reinterpret_cast<float4*>(sR + ((u + 1) % 3) * UNIT_FLOATS);
        dst[vtid] = wr0; dst[vtid + 512] = wr1;
      }
      if (u != 1 && (NHALF ? u + 2 < NU || u == 0 : u < 15) && !(WINO_ABL & 4)) {
        const int un = NHALF ? 2 * (u == 0 ? 3 : u + 2) + nh : (u == 0 ? 3 : (u + 2) & 15);
        wr0 = bload(rsW, vtid * 16, un * UNIT_FLOATS * 4);
        wr1 = bload(rsW, vtid * 16, un * UNIT_FLOATS * 4 + 8192);
      }
      if (u == 0 && !NHALF) {   // next tile's origin (by carries) and halo offsets, in the shadow of this unit's MFMAs
        nx0 = x0 + g_tx * TC; ny0 = y0 + g_ty * TR; nimg = img + g_img;
        if (nx0 >= p.tiles_x * TC) { nx0 -= p.tiles_x * TC; ny0 += TR; }
        if (ny0 >= p.tiles_y * TR) { ny0 -= p.tiles_y * TR; nimg += 1; }
        halo_offsets(hx, has_next, nimg, ny0, nx0);
      }
      // in-place halo streaming: slice g (stages 2g, 2g+1) of the NEXT tile is requested at unit 4g+1 -- right behind this
      // unit's weight loads, so the first younger load that is waited for (vmcnt retires in order) is two units away --
      // and stored at unit 4g+4, once the barrier there says every wave is done with this tile's slice g
      if (!NHALF && (u & 3) == 1 && !(WINO_ABL & 2)) {
#pragma unroll
        for (int i = 0; i < 3; ++i) hv[i] = bload(rsX, hx[i], (u >> 2) * 64);
      }
      if (!NHALF && u == 15 && !(WINO_ABL & 4)) {
        wx0 = bload(rsW, vtid * 16, UNIT_FLOATS * 4); wx1 = bload(rsW, vtid * 16, UNIT_FLOATS * 4 + 8192);
        wy0 = bload(rsW, vtid * 16, 2 * UNIT_FLOATS * 4); wy1 = bload(rsW, vtid * 16, 2 * UNIT_FLOATS * 4 + 8192);
      }
      // second half of this unit's B fragments (published by the previous unit's barrier)
      {
        const float* bu = bBase + (u % 3) * UNIT_FLOATS;
        B23[0] = *reinterpret_cast<const float4*>(bu + 512);
        B23[1] = *reinterpret_cast<const float4*>(bu + 768);
      }
      // S2: the fragments V of stage j were formed during the previous (odd) unit, in the shadow of its MFMAs; request the
      // patch of stage j+1 now, transform it during unit (j, 1)
      if (sfirst && !(WINO_ABL & 8)) {
        if (NHALF && u > 0) transform(raw, nV);   // (tail launch: not pipelined)
        V[0] = nV[0]; V[1] = nV[1]; V[2] = nV[2]; V[3] = nV[3];
      }
      if (!NHALF && slast && j < 7 && !(WINO_ABL & 8)) raw_read(j + 1, 1, raw);
      // S3: positions nu = 0, 1
      acc[0][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(V[0].x, B01[0].x, j == 0 ? zero16 : acc[0][nt], 0, 0, 0);   // stage 0: C = 0, no zeroing pass
      acc[1][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(V[1].x, B01[1].x, j == 0 ? zero16 : acc[1][nt], 0, 0, 0);
      acc[0][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(V[0].y, B01[0].y, acc[0][nt], 0, 0, 0);
      acc[1][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(V[1].y, B01[1].y, acc[1][nt], 0, 0, 0);
      acc[0][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(V[0].z, B01[0].z, acc[0][nt], 0, 0, 0);
      acc[1][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(V[1].z, B01[1].z, acc[1][nt], 0, 0, 0);
      acc[0][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(V[0].w, B01[0].w, acc[0][nt], 0, 0, 0);
      acc[1][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(V[1].w, B01[1].w, acc[1][nt], 0, 0, 0);
      // the region between two barriers holds 16 MFMAs (S6 of the previous unit, S3 of this one).  A 32x32x2 fp32 MFMA occupies
      // the matrix pipe for 64 cycles but the SIMD's vector issue for only 8: up to ~6 VALU / LDS / VMEM instructions per wave
      // hide in each gap, a longer run does not (the two waves of a SIMD run in lock-step, neither has an MFMA to offer
      // meanwhile) -- so deal the region's other instructions out evenly, MFMA first after the barrier
#pragma unroll
      for (int k = 0; k < 16; ++k) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);
        __builtin_amdgcn_sched_group_barrier(0x080, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x010, 1, 0);
      }
      // S4: publishes unit u+1; every wave has finished unit u-1
      if (!(WINO_ABL & 1)) LDS_BARRIER();
      // S5: the patch of stage j+1 (no LDS read is left in flight at the barrier's wait), first half of the next unit's B fragments; halo slice g-1 of the next tile goes in place at unit 4g
      if (u < NU - 1) {
        const float* bn = bBase + ((u + 1) % 3) * UNIT_FLOATS;
        nB01[0] = *reinterpret_cast<const float4*>(bn);
        nB01[1] = *reinterpret_cast<const float4*>(bn + 256);
      }
      __builtin_amdgcn_sched_barrier(0);   // LDS returns in order: the B fragments must not queue behind the 8 patch reads
      if (u == NU - 2) {
#pragma unroll
        for (int i = 0; i < NOUT; ++i) {
          opix0[i] = out_pixel(0, i, img, y0, x0);
          opix1[i] = out_pixel(1, i, img, y0, x0);
          if (has_e) res[i] = bload(rsE, opix0[i] >= 0 ? (opix0[i] * e_stride + e_choff + chunk_of(i) * 4) * 4 : OOB, 0);
        }
      }
      if (sfirst && j < 7 && !(WINO_ABL & 8)) raw_read(j + 1, 0, raw);
      if (NHALF && j < 7 && !(WINO_ABL & 8)) raw_read(j + 1, 1, raw);
      if (!NHALF && slast && j < 7 && !(WINO_ABL & 8)) transform(raw, nV);   // in the shadow of this unit's last 8 and the next unit's first 8 MFMAs
      if (!NHALF && (u & 3) == 0 && u > 0 && !(WINO_ABL & 2)) {   // (no next tile: hv holds zeros, the halo is dead -- harmless)
#pragma unroll
        for (int i = 0; i < 3; ++i) halo_store((u >> 2) - 1, i, hv[i]);
      }
      // S6: positions nu = 2, 3
      acc[2][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(V[2].x, B23[0].x, j == 0 ? zero16 : acc[2][nt], 0, 0, 0);
      acc[3][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(V[3].x, B23[1].x, j == 0 ? zero16 : acc[3][nt], 0, 0, 0);
      acc[2][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(V[2].y, B23[0].y, acc[2][nt], 0, 0, 0);
      acc[3][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(V[3].y, B23[1].y, acc[3][nt], 0, 0, 0);
      acc[2][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(V[2].z, B23[0].z, acc[2][nt], 0, 0, 0);
      acc[3][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(V[3].z, B23[1].z, acc[3][nt], 0, 0, 0);
      acc[2][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(V[2].w, B23[0].w, acc[2][nt], 0, 0, 0);
      acc[3][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(V[3].w, B23[1].w, acc[3][nt], 0, 0, 0);
      if (u < NU - 1) { B01[0] = nB01[0]; B01[1] = nB01[1]; }
      STAMP(8 + u);
    }

    // ---- seam --------------------------------------------------------------------------------------------------
    LDS_BARRIER();   // K loop finished everywhere: ring free, last halo slice dead
    STAMP(1);        // seam barrier wait
    OPAQUE(vtid); OPAQUE(offBf);
    if (!NHALF) {
#pragma unroll
      for (int i = 0; i < 3; ++i) halo_store(3, i, hv[i]);
    }
    // output pixels of both rounds; the prefetched operand (residual / LeakyReLU' mask) of a round is requested before any
    // store of the previous round is queued (vmcnt retires in order: a load behind the stores would wait for them)
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      int* opix = b == 0 ? opix0 : opix1;
      // r_xi[b] = sum_nu M[xi][nu] A[nu][b]:  b = 0: M0 + M1 + M2,  b = 1: M1 - M2 - M3   -> X[wave][tile][channel]
      float* xo = sR + wave * 2048 + (4 * half) * 64 + l31;
#pragma unroll
      for (int nt = 0; nt < (NHALF ? 1 : 2); ++nt)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int m = (r & 3) + 8 * (r >> 2);
          float v = b == 0 ? (acc[0][nt][r] + acc[1][nt][r]) + acc[2][nt][r] : (acc[1][nt][r] - acc[2][nt][r]) - acc[3][nt][r];
          xo[m * 64 + (NHALF ? nh : nt) * 32] = v;
        }
      LDS_BARRIER();
      STAMP(2 + 2 * b);   // exchange write + barrier
      // Y[a][b] = sum_xi At[a][xi] r_xi[b]:  a = 0: r0 + r1 + r2,  a = 1: r1 - r2 - r3
      float4 vo[NOUT];
#pragma unroll
      for (int i = 0; i < NOUT; ++i) {
        const int c = chunk_of(i), q = (vtid + 512 * i) >> (NHALF ? 3 : 4);
        const int R = q >> 4, txo = q & 15;
        const int a = R & 1, m = ((R >> 1) & 1) * 16 + txo;
        const float* xs = sR + ((R >> 2) * 4 + a) * 2048 + m * 64 + c * 4;
        float4 e0 = *reinterpret_cast<const float4*>(xs);
        float4 e1 = *reinterpret_cast<const float4*>(xs + 2048);
        float4 e2 = *reinterpret_cast<const float4*>(xs + 4096);
        float4 v = a == 0 ? f4add(f4add(e0, e1), e2) : f4sub(f4sub(e0, e1), e2);
        v.x = v.x >= 0.f ? v.x : v.x * p.slope; v.y = v.y >= 0.f ? v.y : v.y * p.slope;
        v.z = v.z >= 0.f ? v.z : v.z * p.slope; v.w = v.w >= 0.f ? v.w : v.w * p.slope;
        if (MASK) {   // backward: LeakyReLU' from the saved activation, then the (rare) accumulate operand
          v.x *= res[i].x > 0.f ? 1.f : p.mk_slope; v.y *= res[i].y > 0.f ? 1.f : p.mk_slope;
          v.z *= res[i].z > 0.f ? 1.f : p.mk_slope; v.w *= res[i].w > 0.f ? 1.f : p.mk_slope;
        } else if (has_e) {
          v = f4add(v, res[i]);
        }
        if (has_l) v = f4add(v, bload(rsL, opix[i] >= 0 ? (opix[i] * l_stride + l_choff + c * 4) * 4 : OOB, 0));   // rare: late load
        vo[i] = v;
      }
      if (b == 0 && has_e) {   // round 1's operand, ahead of round 0's stores
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < NOUT; ++i)
          res[i] = bload(rsE, opix1[i] >= 0 ? (opix1[i] * e_stride + e_choff + chunk_of(i) * 4) * 4 : OOB, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
#pragma unroll
      for (int i = 0; i < NOUT; ++i)
        bstore(rsY, opix[i] >= 0 ? (opix[i] * p.y_stride + p.y_choff + chunk_of(i) * 4) * 4 : OOB, vo[i]);
      if (b == 1 && !has_next) break;
      LDS_BARRIER();   // exchange region free again
      STAMP(3 + 2 * b);   // exchange read, combine, stores + barrier
    }
    if (!has_next) break;
    // ---- restart the weight ring: units 0, 1, 2 (in registers since units 14 / 15) -> ring slots 0, 1, 2 -------------
    raw_read(0, 0, raw); raw_read(0, 1, raw);   // the next tile's halo has been complete since round 0's barrier
    {
      float4* dst = reinterpret_cast<float4*>(sR);
      dst[vtid] = wr0; dst[vtid + 512] = wr1;
      dst[vtid + 1024] = wx0; dst[vtid + 1536] = wx1;
      dst[vtid + 2048] = wy0; dst[vtid + 2560] = wy1;
    }
    transform(raw, nV);
    LDS_BARRIER();
    B01[0] = *reinterpret_cast<const float4*>(sR + offBf);
    B01[1] = *reinterpret_cast<const float4*>(sR + offBf + 256);
    STAMP(6);   // ring restart + barrier
    tile = next; img = nimg; y0 = ny0; x0 = nx0;
  }
#ifdef LFSR_CONV_DIAG
  if (dbgbuf && tid == 0)
    for (int k = 0; k < 32; ++k) dbgbuf[bid * 32 + k] = (float)seg[k];
#endif
}

// One launch per conv op: blocks [0, nbody) are the persistent body blocks (tiles [0, ntiles)), blocks [nbody, nbody + 2 ntail)
// the channel-split blocks of the leftover tiles -- they are dispatched as body blocks retire, without a second launch's latency.
template <bool MASK>
__global__ __launch_bounds__(512) void k_conv3x3_wino(WinoArgs p) {
  if ((int)blockIdx.x < p.nbody) {
    wino_tile_loop<MASK, false>(p, (int)blockIdx.x, p.nbody);
  } else {
    p.tile_begin = p.ntiles;
    wino_tile_loop<MASK, true>(p, (int)blockIdx.x - p.nbody, 0);
  }
}

// U = G g G^t per (n, k) from the direct pack [tap][n][k] -> [j = k/8][nt = n/32][p][half = (k/4)&1][n%32][k%4]
__global__ __launch_bounds__(256) void k_pack_wino(const float* __restrict__ direct, float* __restrict__ out) {
  const int i = blockIdx.x * 256 + threadIdx.x;   // (n, k)
  if (i >= 64 * 64) return;
  const int n = i >> 6, k = i & 63;
  double g[3][3];
#pragma unroll
  for (int t = 0; t < 9; ++t) g[t / 3][t % 3] = (double)direct[(t * 64 + n) * 64 + k];
  const double G[4][3] = {{1.0, 0.0, 0.0}, {0.5, 0.5, 0.5}, {0.5, -0.5, 0.5}, {0.0, 0.0, 1.0}};
  double tmp[4][3];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int c = 0; c < 3; ++c) tmp[a][c] = G[a][0] * g[0][c] + G[a][1] * g[1][c] + G[a][2] * g[2][c];
  const int j = k >> 3, hf = (k >> 2) & 1, e = k & 3, nt = n >> 5, n32 = n & 31;
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      double u = tmp[a][0] * G[b][0] + tmp[a][1] * G[b][1] + tmp[a][2] * G[b][2];
      const int pp = a * 4 + b;
      out[(((((j * 2 + nt) * 16 + pp) * 2 + hf) * 32 + n32) << 2) + e] = (float)u;
    }
}

}  // namespace

static int conv3_sel_mask(const char* sel) {
  if (!sel) return LFSR_W_WINO4;
  if (sel[0] == 'h' || sel[0] == 'g') return 0;                              // direct kernels: the direct pack only
  if (sel[0] == 'w' && sel[1] == 'i' && sel[2] == 'n' && sel[3] == 'o') {
    if (sel[4] == '2') return LFSR_W_WINO2;
  }
  return LFSR_W_WINO4;
}

// LFSR_CONV3X3 selects the forward kernel; LFSR_DGRAD3 (same vocabulary) the data-gradient kernel, which otherwise follows LFSR_CONV3X3
const char* lfsr_conv3_fwd_sel() { return lfsr_sel("LFSR_CONV3X3"); }
const char* lfsr_conv3_dgrad_sel() { const char* d = lfsr_sel("LFSR_DGRAD3"); return d ? d : lfsr_sel("LFSR_CONV3X3"); }

int lfsr_conv3_variant_mask() { return conv3_sel_mask(lfsr_conv3_fwd_sel()) | conv3_sel_mask(lfsr_conv3_dgrad_sel()); }

int lfsr_pack_wino_m(const float* direct_packed, float* out, int mask, hipStream_t st) {
  if (!direct_packed || !out) return LFSR_E_ARG;
  int rc = LFSR_OK;
  if (mask & LFSR_W_WINO2) {
    hipLaunchKernelGGL(k_pack_wino, dim3(16), dim3(256), 0, st, direct_packed, out);
    LFSR_CHECK_LAUNCH();
  }
  if (!rc && (mask & LFSR_W_WINO4)) rc = lfsr_pack_wino4(direct_packed, out + LFSR_CONV3_WINO2_FLOATS, st);
  return rc;
}

int lfsr_pack_wino(const float* direct_packed, float* out, hipStream_t st) { return lfsr_pack_wino_m(direct_packed, out, LFSR_W_ALL, st); }

// w_wino: the Winograd-domain pack (lfsr_pack_wino); w_direct: the [9][64][64] pack, used by the channel-split tail launch.
int lfsr_conv3x3_wino_launch(const float* x, int x_stride, int x_choff, const float* w_wino, const float* w_direct, float* y, int y_stride, int y_choff,
                             const float* r1, int r1_stride, int r1_choff, const float* r2, int r2_stride, int r2_choff,
                             const float* mk, int mk_stride, int mk_choff, float mk_slope,
                             int n_img, int h, int w, float slope, const char* sel, hipStream_t st) {
  {   // F(4x4,3x3), the default (conv3x3_wino4.hip);
      // LFSR_CONV3X3=wino2 keeps this file's F(2x2,3x3) kernel (A/B runs), as do operands the F(4x4) launchers do not cover
    const bool is_w = sel && sel[0] == 'w' && sel[1] == 'i' && sel[2] == 'n' && sel[3] == 'o';
    if (!(is_w && sel[4] == '2')) {
      const int rc = lfsr_conv3x3_wino4_launch(x, x_stride, x_choff, w_wino + LFSR_CONV3_WINO2_FLOATS, y, y_stride, y_choff, r1, r1_stride, r1_choff,
                                               r2, r2_stride, r2_choff, mk, mk_stride, mk_choff, mk_slope, n_img, h, w, slope, st);
      if (rc != LFSR_E_ARG) return rc;
    }
  }
  // this file's F(2x2) kernel runs only when LFSR_CONV3X3 selects it (the runtimes pack only the selected copies): operands the F(4x4)
  // launchers do not cover (1 GiB and more) go to the callers' direct 9-tap kernel
  if (!(conv3_sel_mask(sel) & LFSR_W_WINO2)) return LFSR_E_ARG;
  static std::atomic<bool> attr_set[64];
  static std::atomic<int> cus[64];
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return LFSR_E_ARG;
  if (!attr_set[dev]) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_conv3x3_wino<false>), hipFuncAttributeMaxDynamicSharedMemorySize, SMEM_BYTES);
    if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_conv3x3_wino<true>), hipFuncAttributeMaxDynamicSharedMemorySize, SMEM_BYTES);
    if (e != hipSuccess) return LFSR_HIP_ERR(e);
    int v = 0;
    cus[dev] = (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) ? v : 256;
    attr_set[dev] = true;
  }
  const int ncu = cus[dev];
  WinoArgs p{};
  p.X = x; p.x_stride = x_stride; p.x_choff = x_choff; p.Wu = w_wino;
  p.Y = y; p.y_stride = y_stride; p.y_choff = y_choff;
  p.R1 = r1; p.r1_stride = r1_stride; p.r1_choff = r1_choff; p.R2 = r2; p.r2_stride = r2_stride; p.r2_choff = r2_choff;
  p.Mk = mk; p.mk_stride = mk_stride; p.mk_choff = mk_choff; p.mk_slope = mk_slope;
  p.n_img = n_img; p.H = h; p.W = w; p.tiles_y = (h + TR - 1) / TR; p.tiles_x = (w + TC - 1) / TC; p.slope = slope;
  const long long nblk = (long long)n_img * p.tiles_y * p.tiles_x;
  if (nblk <= 0 || nblk > 0x7fffffffLL) return LFSR_E_ARG;
  // every operand is addressed through a buffer descriptor with 32-bit byte offsets: spans of 2 GiB and more go to the direct kernel
  {
    int ms = x_stride > y_stride ? x_stride : y_stride;
    if (r1 && r1_stride > ms) ms = r1_stride;
    if (r2 && r2_stride > ms) ms = r2_stride;
    if (mk && mk_stride > ms) ms = mk_stride;
    if ((long long)n_img * h * w * ms * 4 >= (1LL << 31) || (long long)h * w >= (1 << 24)) {
      if (!w_direct) return LFSR_E_ARG;
      return lfsr_conv3x3_halo_launch(x, x_stride, x_choff, w_direct, y, y_stride, y_choff, r1, r1_stride, r1_choff, r2, r2_stride, r2_choff,
                                      mk, mk_stride, mk_choff, mk_slope, n_img, h, w, slope, st);
    }
  }
  // tiles beyond the last full round (L = ntiles % CUs) go to a channel-split tail launch, two blocks per tile, when that halves
  // the tail (2L <= CUs); LFSR_CONV_TAIL=halo runs the tail on the direct 9-tap kernel instead (A/B)
  int tail = (int)(nblk % ncu);
  if (nblk < ncu || 2 * tail > ncu || lfsr_sel("LFSR_CONV_NOTAIL")) tail = 0;
  const char* tsel = lfsr_sel("LFSR_CONV_TAIL");
  const bool tail_direct = tsel && tsel[0] == 'h' && w_direct;
  const int body = (int)nblk - tail;
  p.ntiles = body;
  p.nbody = body < ncu ? body : ncu;
  const int wino_tail = tail_direct ? 0 : tail;
  {
    const long long npix4 = (long long)n_img * h * w * 4;   // (spans < 2 GiB: checked above)
    p.x_bytes = (int)(npix4 * x_stride); p.y_bytes = (int)(npix4 * y_stride);
    p.r1_bytes = r1 ? (int)(npix4 * r1_stride) : 0; p.r2_bytes = r2 ? (int)(npix4 * r2_stride) : 0; p.mk_bytes = mk ? (int)(npix4 * mk_stride) : 0;
  }
#ifdef LFSR_CONV_DIAG
  p.dbg = g_lfsr_diag_buf;
#endif
  const unsigned grid = (unsigned)(p.nbody + 2 * wino_tail);
  if (mk) hipLaunchKernelGGL((k_conv3x3_wino<true>), dim3(grid), dim3(512), SMEM_BYTES, st, p);
  else hipLaunchKernelGGL((k_conv3x3_wino<false>), dim3(grid), dim3(512), SMEM_BYTES, st, p);
  LFSR_CHECK_LAUNCH();
  if (tail > 0 && tail_direct)
    return lfsr_conv3x3_halo_tail_launch(x, x_stride, x_choff, w_direct, y, y_stride, y_choff, r1, r1_stride, r1_choff, r2, r2_stride, r2_choff,
                                         mk, mk_stride, mk_choff, mk_slope, n_img, h, w, slope, body, tail, st);
  return LFSR_OK;
}
