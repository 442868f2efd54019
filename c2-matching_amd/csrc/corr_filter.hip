// corr_filter.hip -- the correlation / arg-max search (ref_map_util.py:26-86) as a 16-BIT-PIPE PRE-FILTER + EXACT RE-SCORE.
//
// The contract is the exact kernel's (corr_argmax.hip): index maps and values bit-identical to the fp32 chain of the oracle
// (pixel-level fmaf chains over the channels, nine-tap sums in the oracle's order, times the fp32 inverse patch norm, first
// maximum).  That kernel is pinned to the fp32 MFMA (157 TF).  Here the SAME sliding-window sweep runs on
// v_mfma_f32_32x32x16_f16 with both maps scaled by 2^14 and split into two f16 pieces (2^14 x = x0 + x1 + e, three piece
// products q0 r0 + q1 r0 + q0 r1 into ONE accumulator; the 2^-28 goes into the candidate's scale) -- 3/16 of the matrix
// time -- and yields, per query, not the answer but a short list of candidates that provably contains it:
//
//   * F(q, n) = filter score, E(q, n) = the oracle's fp32 value.  |F - E| <= eps(q) := (ALPHA |q patch| + BETA) / 2 for
//     every n (error budget below), so   E(n*) = max E   implies   F(n*) >= max F - 2 eps.  Every lane of the sweep keeps,
//     for its share of the candidates (ref column == lane (mod x-tile)), the three largest filter scores and the indices of
//     the first two; after the sweep the half-wave that serves a query takes the maximum over its 32 lanes and every lane
//     with a score inside the band emits its candidates: one index, two, or -- third score inside the band too -- a request
//     to re-score that lane's whole candidate set; more than KSLOT entries: re-score every ref patch (never on real data).
//   * corr_resolve_kernel evaluates E for the listed candidates exactly as the oracle does (one lane per tap: a 256-term
//     sequential fmaf chain on channels-last copies of the maps; the nine taps summed in the oracle's order) and keeps the
//     larger value, then the lower index.  The list contains the true arg-max, and any candidate it may additionally
//     contain scores lower or ties with a higher index -- so the result is the oracle's, bit for bit, value included.
//   * candidates whose 3x3xC patch repeats the patch to its left or above bit for bit (the band a zero-padded Ref leaves)
//     are not scored at all (bias = -inf): they tie with a lower index in the filter AND in the exact arithmetic, so they
//     can never be the first maximum -- and they would otherwise fill the band with hundreds of exact ties.
//
// Error budget (u = 2^-24).  Exact side: 9 fmaf chains of C <= 256 terms, 4 tap additions, the product with the inverse norm:
// |E - R| <= 261 u S, S = sum |q_c r_c| / (|r| + 1e-5) <= |q patch|, R the real-number score.  Filter side: in the scaled
// domain (|2^14 x| < 65520, i.e. |x| < 3.99: channel-normalised features are <= 1) the pieces are x0 = rne_f16(2^14 x),
// x1 = rne_f16(2^14 x - x0), each flushed to zero below 2^-14 by the split itself (whatever the matrix pipe does with f16
// subnormals): |e| <= 2^-22 |x|, or 2^-28 absolute; dropped product q1 r1 <= 2^-22 |q||r|; fp32 accumulation of the MFMAs
// bounded as if every one of the 16 products of an instruction were added sequentially (272 roundings; the hardware does
// better), the tap sums, the product with the scale: (543 u + 3.1 * 2^-22) |q patch| + 2^-28 * 48 (1 + |q patch| / (|r| +
// 1e-5)).  Candidates with 1 / (|r| + 1e-5) > 2 (degenerate, all-but-zero ref patches: never with channel-normalised
// features) switch the whole launch to the exact sweep, as do inputs with |x| >= 3.99.  With a 1.25 safety factor on the
// rounding terms: 2 eps = 8.4e-5 |q patch| + 1e-6.  On the benchmark's features 9 % of the queries list two candidates,
// 1.3 % more (profiles/r04_corr_band_histogram_lr160.json); measured |F - R| is 3.5e-7.
#include "corr_filter.h"

#include <stdlib.h>

#include "c2m_common.h"

namespace c2m {
namespace corrf {

constexpr int TQ = 16, TPQ = TQ - 2, WT = 32, NWAVE = 8, NTHR = NWAVE * 64, QPIX = TQ * TQ, SLAB = QPIX * WT, NIT = TPQ;
constexpr float BAND_ALPHA = 8.4e-5f, BAND_BETA = 1.0e-6f, INV_CAP = 2.0f;
constexpr float PIECE_SCALE = 16384.0f;                 // 2^14: both maps are split in this scaled domain
constexpr float SCORE_UNSCALE = 3.725290298461914e-9f;  // 2^-28, folded into the candidate scales
constexpr float F16_MIN_NORMAL = 6.103515625e-5f;   // 2^-14
constexpr int SCAN_FLAG = 0x40000000;
constexpr int SCAN_CAP = SCAN_ITEMS;           // work-list entries (whole-lane / whole-map re-scores); more: exact sweep
constexpr int SCAN_CHUNKS = 32, SCAN_STRIPES = 64;

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

// ------------------------------------------------------------------------------------------------------------------
// preparation
// ------------------------------------------------------------------------------------------------------------------

// [B][C][HW] -> [B][HW][C] (C % 64 == 0), 64 x 64 tiles through LDS
__global__ void __launch_bounds__(256) to_nhwc_kernel(const float* __restrict__ in, int C, int HW, float* __restrict__ out) {
  __shared__ float tile[64][65];
  const int b = blockIdx.z, c0 = blockIdx.y * 64, p0 = blockIdx.x * 64;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  const float* ib = in + (size_t)b * C * HW;
  float* ob = out + (size_t)b * C * HW;
  const int p = p0 + tx;
#pragma unroll
  for (int r = 0; r < 16; ++r) tile[ty + 4 * r][tx] = p < HW ? ib[(size_t)(c0 + ty + 4 * r) * HW + p] : 0.0f;
  __syncthreads();
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int pp = p0 + ty + 4 * r;
    if (pp < HW) ob[(size_t)pp * C + c0 + tx] = tile[tx][ty + 4 * r];
  }
}

__device__ __forceinline__ void split2(float x, _Float16& h0, _Float16& h1, bool& bad);

// The query map's two preparation passes in one (round 6): the 64 x 64 transpose tile of to_nhwc_kernel also yields the f16 pieces
// (split_query_kernel's arithmetic and layout: qpl [B][2][HW][C]) -- every thread splits two runs of 8 channels of one pixel from
// the tile and stores them as 16-byte pieces.
__global__ void __launch_bounds__(256) to_nhwc_split_kernel(const float* __restrict__ in, int C, int HW, float* __restrict__ out,
                                                             _Float16* __restrict__ qpl, int* __restrict__ flags) {
  __shared__ float tile[64][65];
  const int b = blockIdx.z, c0 = blockIdx.y * 64, p0 = blockIdx.x * 64;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  const float* ib = in + (size_t)b * C * HW;
  float* ob = out + (size_t)b * C * HW;
  const int p = p0 + tx;
#pragma unroll
  for (int r = 0; r < 16; ++r) tile[ty + 4 * r][tx] = p < HW ? ib[(size_t)(c0 + ty + 4 * r) * HW + p] : 0.0f;
  __syncthreads();
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int pp = p0 + ty + 4 * r;
    if (pp < HW) ob[(size_t)pp * C + c0 + tx] = tile[tx][ty + 4 * r];
  }
  bool bad = false;
#pragma unroll
  for (int it = 0; it < 2; ++it) {
    const int item = threadIdx.x + 256 * it, px = item >> 3, c8 = item & 7;   // 64 pixels x 8 runs of 8 channels
    const int pp = p0 + px;
    if (pp < HW) {
      f16x8 q0, q1;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        _Float16 h0, h1;
        split2(tile[c8 * 8 + e][px], h0, h1, bad);
        q0[e] = h0; q1[e] = h1;
      }
      _Float16* o = qpl + ((size_t)(b * 2) * HW + pp) * C + c0 + c8 * 8;
      *reinterpret_cast<f16x8*>(o) = q0;
      *reinterpret_cast<f16x8*>(o + (size_t)HW * C) = q1;
    }
  }
  if (bad) flags[0] = 1;
}

// 2^14 x = h0 + h1 + e: h0 = rne_f16(2^14 x), h1 = rne_f16(2^14 x - h0); pieces below 2^-14 are flushed to zero here
__device__ __forceinline__ void split2(float x, _Float16& h0, _Float16& h1, bool& bad) {
  const float xs = x * PIECE_SCALE;   // exact
  const float ax = __builtin_fabsf(xs);
  bad |= !(ax < 65520.0f);   // inf / NaN / rounds to f16 infinity
  const _Float16 a = ax < F16_MIN_NORMAL ? (_Float16)0.0f : (_Float16)xs;
  const float r = xs - (float)a;      // exact
  const _Float16 c = __builtin_fabsf(r) < F16_MIN_NORMAL ? (_Float16)0.0f : (_Float16)r;
  h0 = a;
  h1 = c;
}

__device__ __forceinline__ void split8(const float* __restrict__ src, f16x8& p0, f16x8& p1, bool& bad) {
  const f32x4 a = *reinterpret_cast<const f32x4*>(src), c = *reinterpret_cast<const f32x4*>(src + 4);
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    _Float16 h0, h1;
    split2(a[e], h0, h1, bad);
    p0[e] = h0; p1[e] = h1;
    split2(c[e], h0, h1, bad);
    p0[4 + e] = h0; p1[4 + e] = h1;
  }
}

// query pieces: qn [B][HW][C] fp32 -> qpl [B][2][HW][C] f16; one thread per 8 channels
__global__ void __launch_bounds__(256) split_query_kernel(const float* __restrict__ qn, int C, int HW, long long n8,
                                                           _Float16* __restrict__ qpl, int* __restrict__ flags) {
  const long long e = (long long)blockIdx.x * 256 + threadIdx.x;
  if (e >= n8) return;
  const int c8 = (int)(e % (C / 8));
  const long long gp = e / (C / 8);
  const long long b = gp / HW, p = gp - b * HW;
  f16x8 p0, p1;
  bool bad = false;
  split8(qn + (size_t)gp * C + c8 * 8, p0, p1, bad);
  _Float16* o = qpl + ((size_t)(b * 2) * HW + p) * C + c8 * 8;
  *reinterpret_cast<f16x8*>(o) = p0;
  *reinterpret_cast<f16x8*>(o + (size_t)HW * C) = p1;
  if (bad) flags[0] = 1;
}

// ref pieces as LDS row images: rimg[b][xt][y] = [piece 2][k step C/16][k half 2][pixel 32][8 f16], pixel j = column
// xt * WP + j (zeros beyond the map).  grid (Hr, x-tiles, B), one thread per (k step, k half, pixel) and pass.
__global__ void __launch_bounds__(256) split_ref_image_kernel(const float* __restrict__ rn, int C, int Hr, int Wr, int nxt,
                                                               _Float16* __restrict__ rimg, int* __restrict__ flags) {
  const int y = blockIdx.x, xt = blockIdx.y, b = blockIdx.z;
  _Float16* img = rimg + (((size_t)b * nxt + xt) * Hr + y) * ((size_t)C * 64);
  const size_t plane = (size_t)(C / 16) * 512;   // f16 elements per piece
  bool bad = false;
  for (int u = threadIdx.x; u < C * 4; u += 256) {
    const int j = u & 31, kh = (u >> 5) & 1, t = u >> 6;
    const int x = xt * WP + j;
    f16x8 p0 = {0, 0, 0, 0, 0, 0, 0, 0}, p1 = p0;
    if (x < Wr) split8(rn + (((size_t)b * Hr + y) * Wr + x) * C + 16 * t + 8 * kh, p0, p1, bad);
    _Float16* o = img + (size_t)(t * 2 + kh) * 256 + j * 8;
    *reinterpret_cast<f16x8*>(o) = p0;
    *reinterpret_cast<f16x8*>(o + plane) = p1;
  }
  if (bad) flags[0] = 1;
}

// eq[0][b][p] = pixel p equals its LEFT neighbour over all channels (bitwise), eq[1][b][p] = equals the pixel ABOVE.
// One wave per pixel (lane = 4 channels).
__global__ void __launch_bounds__(256) pixel_eq_kernel(const float* __restrict__ rn, int C, int Hr, int Wr, long long npix,
                                                        unsigned char* __restrict__ eq) {
  const long long gp = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (gp >= npix) return;
  const int l = threadIdx.x & 63;
  const int p = (int)(gp % ((long long)Hr * Wr));
  const int y = p / Wr, x = p - y * Wr;
  typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
  bool okl = true, oku = true;
  if (l * 4 < C) {
    const u32x4* me = reinterpret_cast<const u32x4*>(rn + (size_t)gp * C) + l;
    const u32x4 v = *me;
    if (x > 0) {
      const u32x4 o = *(me - C / 4);
      okl = v[0] == o[0] && v[1] == o[1] && v[2] == o[2] && v[3] == o[3];
    }
    if (y > 0) {
      const u32x4 o = *(me - (size_t)Wr * (C / 4));
      oku = v[0] == o[0] && v[1] == o[1] && v[2] == o[2] && v[3] == o[3];
    }
  }
  okl = __all(okl) && x > 0;
  oku = __all(oku) && y > 0;
  if (l == 0) {
    eq[gp] = okl ? 1 : 0;
    eq[npix + gp] = oku ? 1 : 0;
  }
}

// scale of candidate n in the filter: 2^-28 / (|r| + 1e-5), or 0 = "not a candidate" if its patch repeats the patch to its
// left / above bit for bit
// maxcol[b] (zeroed by the caller): the last patch column of sample b that holds a candidate
__global__ void __launch_bounds__(256) cand_scale_kernel(const float* __restrict__ inv, const unsigned char* __restrict__ eq,
                                                          long long npix, int Hr, int Wr, int Hrp, int Wrp,
                                                          float* __restrict__ sc, int* __restrict__ flags, int* __restrict__ maxcol) {
  const int n0 = blockIdx.x * 256 + threadIdx.x, b = blockIdx.y;
  const bool in = n0 < Hrp * Wrp;
  const int n = in ? n0 : 0;
  const int ry = n / Wrp, rx = n - ry * Wrp;
  const unsigned char* el = eq + (size_t)b * Hr * Wr + (size_t)ry * Wr + rx;
  const unsigned char* eu = el + npix;
  bool dl = true, du = true;
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      dl = dl && el[i * Wr + j] != 0;
      du = du && eu[i * Wr + j] != 0;
    }
  const float s = inv[(size_t)b * Hrp * Wrp + n];
  const bool dup = dl || du;   // (eq is 0 in column 0 / row 0: the neighbour patch exists whenever the flags hold)
  if (in) {
    if (!dup && !(s <= INV_CAP && s > 0.0f)) flags[0] = 1;
    sc[(size_t)b * Hrp * Wrp + n] = dup ? 0.0f : s * SCORE_UNSCALE;
  }
  int mc = (in && !dup) ? rx : 0;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) mc = max(mc, __shfl_xor(mc, o, 64));
  if ((threadIdx.x & 63) == 0 && mc > 0) atomicMax(maxcol + b, mc);
}

// Trailing x-tiles whose every patch is a repeat (first patch column beyond maxcol[b]) are not swept: such a patch ties with a
// LOWER-indexed one in the filter and in the exact arithmetic alike, it can never be the first maximum (ref_map_util.py:74).
// Exact and data-dependent like the duplicate-row table it is written into: (0, Hr) = rows [0, Hr) of that (sample, x-tile) skipped.
// INVARIANT both sweeps rely on (corr_filter_kernel / corr_argmax_mfma_kernel: they subtract every entry from the step count S up
// front and only test `yn == sk.x` while walking INSIDE a tile, so a (0, Hr) entry is honoured only because the walk never
// enters such a tile): dead tiles form a TRAILING set -- `xt * WP > maxcol[b]` with maxcol = the LAST patch column holding a
// non-duplicate patch marks xt and every tile after it, and never tile 0.  A band of all-duplicate columns in the middle of
// the map is NOT marked (its tiles are swept; tests/test_corr_gpu.py::test_all_duplicate_middle_band_keeps_later_tiles_live).
// Do not mark non-trailing tiles here without teaching the sweeps to skip into the next live tile.
__global__ void dead_tiles_kernel(const int* __restrict__ maxcol, int nxt, int n, int Hr, int2* __restrict__ skip) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;   // (sample, x-tile)
  if (i >= n) return;
  const int b = i / nxt, xt = i - b * nxt;
  if (xt * WP > maxcol[b]) skip[i] = make_int2(0, Hr);
}

__global__ void __launch_bounds__(256) band_kernel(const float* __restrict__ qden, long long n, float* __restrict__ band) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i < n) band[i] = BAND_ALPHA * qden[i] + BAND_BETA;   // qden = |q patch| + 1e-5 >= |q patch|
}

// ------------------------------------------------------------------------------------------------------------------
// the filter sweep: corr_argmax_mfma_kernel's structure (one workgroup = a 16 x 16 query pixel tile, ref swept in x-tiles of
// 32 pixel columns, one pixel row per step, three-slab ring of row sums) with f16 operands:
//   A: the wave's 32 query pixels x C channels, two pieces = C/2 VGPRs (lane (i, kh) holds channels 16 t + 8 kh .. + 7 of
//      pixel(i) for k step t);  B: the ref row image [piece][t][kh][pixel][8], DMA'd as C/8 contiguous KiB per step;
//   per k step: hi += q0 r0;  lo += q0 r1' + q1' r0;   D = hi + 2^-11 lo.
// ------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ unsigned select_u32(unsigned long long lane_mask, unsigned if_set, unsigned if_clear) {
  unsigned r;
  asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(r) : "v"(if_clear), "v"(if_set), "s"(lane_mask));
  return r;
}

// C2M_CORRF_ABL (compile-time, measurement builds only -- results are wrong with any bit set; scripts/abl_corr_filter.py):
// 1 B operands read once per row and reused, 2 no ring reads in the tap rounds, 4 no tap rounds, 8 no row-sum tail (DPP + ring
// stores), 16 no MFMAs.  Round 5, C = 256, 160 x 160 maps, B = 16, no skipped rows: 25.2 ms; 1: 24.1, 2: 23.6, 3: 22.5, 4: 20.9,
// 8: 24.3, 15 (MFMAs + row DMA + barriers only): 18.7 = the pipe at the 1.42 GHz the chip holds under it; 16: 10.6.
#ifndef C2M_CORRF_ABL
#define C2M_CORRF_ABL 0
#endif
#ifndef C2M_CORRF_FAST
#define C2M_CORRF_FAST 1
#endif
template <int C, int PF>
__global__ void __launch_bounds__(NTHR, 2) corr_filter_kernel(
    const _Float16* __restrict__ qpl, const _Float16* __restrict__ rimg, int Hq, int Wq, int Hr, int Wr, int tiles_y, int tiles_x,
    const float* __restrict__ sc, const float* __restrict__ band, const int2* __restrict__ skip, int* __restrict__ cnt,
    int* __restrict__ cand) {
  constexpr int KS = C / 16;                 // k steps
  constexpr int RB = C * 128;                // bytes of one row image
  constexpr int PPW = C / 64;                // DMA pieces (1 KiB) per wave and row
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* ring = smem;                                        // [3][QPIX][WT]
  char* rbuf = reinterpret_cast<char*>(smem + 3 * SLAB);     // [2][RB]

  const int ntile = tiles_y * tiles_x;
  const int vb = xcd_remap(blockIdx.x, gridDim.x);
  const int b = vb / ntile, tile = vb - b * ntile;
  const int ty = tile / tiles_x, tx = tile - ty * tiles_x;
  const int qy0 = ty * TPQ, qx0 = tx * TPQ;
  const int Hqp = Hq - 2, Wqp = Wq - 2, Hrp = Hr - 2, Wrp = Wr - 2;
  const int nxt = (Wrp + WP - 1) / WP;

  const int tid = threadIdx.x;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l = tid & 63, hi = l >> 5, j32 = l & 31;

  // ---- resident A operands (row permutation of the exact kernel: every lane ends up with one query pixel row)
  f16x8 qa[2][KS];
  {
    const int py = min(qy0 + 2 * w + ((j32 >> 2) & 1), Hq - 1);
    const int px = min(qx0 + (j32 & 3) + 4 * (j32 >> 3), Wq - 1);
    const size_t HWq = (size_t)Hq * Wq;
    const _Float16* src = qpl + ((size_t)b * 2 * HWq + (size_t)py * Wq + px) * C + 8 * hi;
#pragma unroll
    for (int t = 0; t < KS; ++t) {
      qa[0][t] = *reinterpret_cast<const f16x8*>(src + 16 * t);
      qa[1][t] = *reinterpret_cast<const f16x8*>(src + HWq * C + 16 * t);
    }
  }

  // ---- per-lane state of query patch (row it, column qx = 2w + hi) over the candidates in ref column j32 (mod x-tile):
  // the three largest filter scores and the 16-bit codes (x-tile << 10 | patch row) of the first two
  float v1[NIT], v2[NIT], v3[NIT];
  unsigned pk[NIT];
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    v1[it] = -INFINITY; v2[it] = -INFINITY; v3[it] = -INFINITY;
    pk[it] = 0u;
  }
  const int qx = min(2 * w + hi, TPQ - 1);
  int lane_off = qx * WT + j32;                        // row sums read by the tap rounds
  int store_off = (2 * w + hi) * TQ * WT + j32;        // row sums written after the MFMAs
  asm volatile("" : "+v"(lane_off), "+v"(store_off));  // (opaque: keep the two offsets, not the lane fields they derive from)

  const int2* __restrict__ skb = skip + (size_t)b * nxt;
  int S = 0;
  for (int i = 0; i < nxt; ++i) S += Hr - (skb[i].y - skb[i].x);
  int2 sk = skb[0];

  // buffer addressing (SGPR resource + one 32-bit per-lane offset + scalar offset): no 64-bit address arithmetic on the
  // vector ALU and no 64-bit per-lane pointers in registers -- this kernel has none to spare
  const __amdgpu_buffer_rsrc_t img_rs = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<char*>(reinterpret_cast<const char*>(rimg) + (size_t)b * nxt * Hr * RB), 0, (int)((unsigned)nxt * (unsigned)Hr * (unsigned)RB),
      0x00020000);
  const __amdgpu_buffer_rsrc_t sc_rs = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(sc + (size_t)b * Hrp * Wrp), 0, (int)((unsigned)Hrp * (unsigned)Wrp * 4u), 0x00020000);
  const unsigned dma_lane = (unsigned)(w * PPW) * 1024u + (unsigned)l * 16u;
  auto issue_row = [&](int xt, int y, int buf) {
    const int so = (xt * Hr + y) * RB;
    char* d = rbuf + buf * RB + (w * PPW) * 1024;
#pragma unroll
    for (int m = 0; m < PPW; ++m)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(img_rs, (__attribute__((address_space(3))) void*)(d + m * 1024), 16, dma_lane, so + m * 1024, 0, 0);
  };
  issue_row(0, 0, 0);

  int y = 0, xt = 0, sl0 = 0;
  unsigned code = 0u;                        // candidate of this iteration (wave-uniform part of its index)
  float scale_next = 0.0f;                   // 0: no candidate
  for (int s = 0; s <= S; ++s) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();  // (A) row s landed; slab of step s-1 complete; everybody left step s-1
    float scale = scale_next;
    asm volatile("" : "+v"(scale));
    const float bias = scale == 0.0f ? -INFINITY : 0.0f;

    int yn = y + 1, xtn = xt;
    if (yn == sk.x) yn = sk.y;
    if (yn >= Hr) { yn = 0; xtn = xt + 1; }   // (a dead tile's (0, Hr) entry is never entered: dead tiles are trailing, S ends before -- dead_tiles_kernel)
    if (s + 1 < S) issue_row(xtn, yn, (s + 1) & 1);

    // candidate of the NEXT iteration: the patch row completed by THIS step
    const int ncol_nx = xt * WP + j32;
    const bool cand_nx = (s < S) && (y >= 2) && (j32 < WP) && (ncol_nx < Wrp);
    scale_next = 0.0f;
    if (cand_nx)
      scale_next = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(sc_rs, (unsigned)j32 * 4u, ((y - 2) * Wrp + xt * WP) * 4, 0));
    const unsigned code_nx = ((unsigned)xt << 10) | (unsigned)max(y - 2, 0);

    const int sl1 = (sl0 == 2) ? 0 : sl0 + 1;
    const int sl2 = (sl1 == 2) ? 0 : sl1 + 1;
    const float* a0 = ring + sl0 * SLAB + lane_off;
    const float* a1 = ring + sl1 * SLAB + TQ * WT + lane_off;
    const float* a2 = ring + sl2 * SLAB + 2 * TQ * WT + lane_off;

    const char* bsrc = rbuf + (s & 1) * RB + l * 16;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
    const unsigned cur = code, curhi = code << 16;

    // B operands: read PF k steps ahead into a ring of PF + 1 register sets
    f16x8 b0[PF + 1], b1[PF + 1];
    float hq[3];
#pragma unroll
    for (int t = 0; t < PF; ++t) {
      b0[t] = *reinterpret_cast<const f16x8*>(bsrc + t * 1024);
      b1[t] = *reinterpret_cast<const f16x8*>(bsrc + (KS + t) * 1024);
    }
    hq[0] = a0[0]; hq[1] = a1[0]; hq[2] = a2[0];
#pragma unroll
    for (int t = 0; t < KS; ++t) {
      if (!(C2M_CORRF_ABL & 16)) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(qa[1][t], b0[t % (PF + 1)], acc, 0, 0, 0);   // smallest terms first
      __builtin_amdgcn_sched_barrier(0);
      if ((C2M_CORRF_ABL & 1) && t + PF < KS) {
        b0[(t + PF) % (PF + 1)] = b0[t % (PF + 1)];
        b1[(t + PF) % (PF + 1)] = b1[t % (PF + 1)];
      } else if (t + PF < KS) {
        b0[(t + PF) % (PF + 1)] = *reinterpret_cast<const f16x8*>(bsrc + (t + PF) * 1024);
        b1[(t + PF) % (PF + 1)] = *reinterpret_cast<const f16x8*>(bsrc + (KS + t + PF) * 1024);
      }
      __builtin_amdgcn_sched_barrier(0);
      if (!(C2M_CORRF_ABL & 16)) {
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(qa[0][t], b1[t % (PF + 1)], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(qa[0][t], b0[t % (PF + 1)], acc, 0, 0, 0);
      }
      // tap-sum rounds [t NIT / KS, (t+1) NIT / KS)
#pragma unroll
      for (int r = (C2M_CORRF_ABL & 4) ? NIT : t * NIT / KS; r < (t + 1) * NIT / KS; ++r) {
        float sum = hq[0] + hq[1];
        sum = sum + hq[2];
        if (!(C2M_CORRF_ABL & 2) && r + 1 < NIT) {
          hq[0] = a0[(r + 1) * TQ * WT];
          hq[1] = a1[(r + 1) * TQ * WT];
          hq[2] = a2[(r + 1) * TQ * WT];
        }
        const float v = __builtin_fmaf(sum, scale, bias);     // -inf without a candidate
        // Round 6: a score that does not beat the lane's THIRD best changes nothing below (med3 / max / selects all return their old
        // values), and after the first few hundred candidates that is nearly every score -- so the update sits behind ONE compare and
        // a wave-uniform branch: 5 vector instructions per round instead of 13 when no lane of the wave improves.  On this chip a
        // wave's vector-ALU instructions cost matrix-pipe time whether or not another wave runs beside it, and
        // the 14 rounds per ref row were a fifth of the sweep.  (C2M_CORRF_FAST=0: the unconditional update, for A/B builds.)
        const unsigned long long m3 = __builtin_amdgcn_fcmpf(v, v3[r], 2 /* ogt */);
        if (!C2M_CORRF_FAST || m3 != 0ull) {
          // (lane masks + explicit v_cndmask: left to itself hipcc turns the two selects into divergent branches)
          const unsigned long long m1 = __builtin_amdgcn_fcmpf(v, v1[r], 2 /* ogt */), m2 = __builtin_amdgcn_fcmpf(v, v2[r], 2);
          const unsigned pa = (pk[r] << 16) | cur, pb = (pk[r] & 0xffffu) | curhi;
          pk[r] = select_u32(m1, pa, select_u32(m2, pb, pk[r]));
          v3[r] = __builtin_amdgcn_fmed3f(v2[r], v3[r], v);
          v2[r] = __builtin_amdgcn_fmed3f(v1[r], v2[r], v);
          v1[r] = __builtin_fmaxf(v1[r], v);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    code = code_nx;

    // Row sums of the three taps of a patch row, formed in registers BEFORE barrier (B) (only the ring stores need it): lane
    // (hi, j) holds D[pixel (2w+hi, r)][ref column j] in acc[r]; the value at (pixel column + 1, ref column + 1) is the next
    // register shifted by one lane (DPP wave_shl:1):  H[r] = D[r] + shl(D[r+1] + shl(D[r+2]))   (oracle tap order)
    auto shl1 = [](float x) __attribute__((always_inline)) {
      return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x130, 0xf, 0xf, true));
    };
    float hh[TPQ];
    if (C2M_CORRF_ABL & 8) {
#pragma unroll
      for (int r = 0; r < TPQ; ++r) hh[r] = acc[r];
    } else {
      float dv[16], tt[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) dv[r] = acc[r];
#pragma unroll
      for (int r = 1; r < 15; ++r) tt[r] = dv[r] + shl1(dv[r + 1]);
#pragma unroll
      for (int r = 0; r < TPQ; ++r) hh[r] = dv[r] + shl1(tt[r + 1]);
    }

    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();  // (B) every tap-sum that reads slab sl0 (about to be overwritten) is done
    {
      float* dst = ring + sl0 * SLAB + store_off;
#pragma unroll
      for (int r = 0; r < ((C2M_CORRF_ABL & 8) ? 1 : TPQ); ++r) dst[r * WT] = hh[r];
    }

    if (xtn != xt) sk = skb[min(xtn, nxt - 1)];
    y = yn;
    xt = xtn;
    sl0 = sl1;
  }

  // ---- candidate lists: the half-wave (w, hi) serves query patch (row it, column 2w + hi); lane j32 holds the states of
  // the candidates in ref columns xt * WP + j32
  const unsigned long long half_mask = hi ? 0xffffffff00000000ull : 0x00000000ffffffffull;
  const unsigned long long below = ((1ull << l) - 1ull) & half_mask;
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    float m = v1[it];
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) m = __builtin_fmaxf(m, __shfl_xor(m, o, 64));
    const int gy = qy0 + it, gx = qx0 + 2 * w + hi;
    const bool valid = w < NWAVE - 1 && gy < Hqp && gx < Wqp;
    const size_t o = (size_t)b * Hqp * Wqp + (size_t)(valid ? gy * Wqp + gx : 0);
    const float thr = m - band[o];
    const bool hot = valid && v1[it] >= thr && v1[it] > -INFINITY;
    const bool scan = hot && v3[it] >= thr;
    const bool two = hot && !scan && v2[it] >= thr;
    const unsigned long long bal1 = __ballot(hot), bal2 = __ballot(two);
    const int pos = __popcll(bal1 & below) + __popcll(bal2 & below);
    const int total = __popcll(bal1 & half_mask) + __popcll(bal2 & half_mask);
    if (valid) {
      if (total <= KSLOT) {
        if (hot) {
          const unsigned c1 = pk[it] & 0xffffu, c2 = pk[it] >> 16;
          const int n1 = (int)(c1 & 1023u) * Wrp + (int)(c1 >> 10) * WP + j32;
          const int n2 = (int)(c2 & 1023u) * Wrp + (int)(c2 >> 10) * WP + j32;
          cand[o * KSLOT + pos] = scan ? (SCAN_FLAG | j32) : n1;
          if (two) cand[o * KSLOT + pos + 1] = n2;
        }
        if (j32 == 0) cnt[o] = total;
      } else if (j32 == 0) {
        cnt[o] = -1;
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------------------------
// exact re-score.  Nine lanes per candidate (lane = tap (ti, tj): a sequential fmaf chain over the channels of query pixel
// q + (ti, tj) and ref pixel n + (ti, tj), channels-last copies), seven candidates per wave; the nine taps are exchanged
// inside the group and summed in the oracle's order.
//   corr_resolve_kernel      one group per query: its listed candidates.  Requests to re-score a whole lane's candidate set
//                            (or every ref patch) go to a work list instead; the query's running best is parked as a 64-bit
//                            key (value bits made order-preserving, then ~index: atomicMax = larger value, then lower index)
//   corr_scan_kernel         the work list, spread over a fixed grid (chunk, stripe): atomicMax into the query's key
//   corr_scan_finish_kernel  keys of the listed queries -> max_idx / max_val
// ------------------------------------------------------------------------------------------------------------------
struct ScanItem { long long q; int lane; int pad; };   // lane < 0: every ref patch

__device__ __forceinline__ unsigned long long pack_key(float v, int n) {
  const unsigned u = __builtin_bit_cast(unsigned, v);
  const unsigned ord = (u & 0x80000000u) ? ~u : (u | 0x80000000u);
  return ((unsigned long long)ord << 32) | (unsigned long long)(0xffffffffu - (unsigned)n);
}
__device__ __forceinline__ void unpack_key(unsigned long long k, float& v, int& n) {
  const unsigned ord = (unsigned)(k >> 32);
  v = __builtin_bit_cast(float, (ord & 0x80000000u) ? (ord & 0x7fffffffu) : ~ord);
  n = (int)(0xffffffffu - (unsigned)(k & 0xffffffffull));
}

struct ExactScorer {
  const f32x4* qp;      // this lane's query pixel
  const float* rb;      // ref map of the sample (channels-last)
  const float* invb;
  int C4, Wr, Wrp, ti, tj, gbase;
  __device__ __forceinline__ float operator()(int n) const {
    const int ry = n / Wrp, rx = n - ry * Wrp;
    const f32x4* rp = reinterpret_cast<const f32x4*>(rb + ((size_t)(ry + ti) * Wr + rx + tj) * (size_t)(4 * C4));
    float d = 0.0f;
#pragma unroll 8
    for (int c4 = 0; c4 < C4; ++c4) {   // (the loads of eight iterations are in flight together; the chain itself is sequential)
      const f32x4 a = qp[c4], r = rp[c4];
      d = __builtin_fmaf(a[0], r[0], d);
      d = __builtin_fmaf(a[1], r[1], d);
      d = __builtin_fmaf(a[2], r[2], d);
      d = __builtin_fmaf(a[3], r[3], d);
    }
    float t[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) t[k] = __shfl(d, gbase + k, 64);
    const float r0 = t[0] + (t[1] + t[2]), r1 = t[3] + (t[4] + t[5]), r2 = t[6] + (t[7] + t[8]);
    float sum = r0 + r1;
    sum = sum + r2;
    return sum * invb[n];
  }
};

__device__ __forceinline__ ExactScorer make_scorer(const float* qn, const float* rn, const float* inv, int C, int Hq, int Wq, int Hr,
                                                   int Wr, long long q, int tap, int gbase) {
  const int Hqp = Hq - 2, Wqp = Wq - 2, Nq = Hqp * Wqp, Nr = (Hr - 2) * (Wr - 2);
  const int b = (int)(q / Nq), qq = (int)(q - (long long)b * Nq);
  const int qy = qq / Wqp, qxx = qq - qy * Wqp;
  ExactScorer s;
  s.ti = tap / 3;
  s.tj = tap - s.ti * 3;
  s.qp = reinterpret_cast<const f32x4*>(qn + (((size_t)b * Hq + qy + s.ti) * Wq + qxx + s.tj) * C);
  s.rb = rn + (size_t)b * Hr * Wr * C;
  s.invb = inv + (size_t)b * Nr;
  s.C4 = C / 4; s.Wr = Wr; s.Wrp = Wr - 2; s.gbase = gbase;
  return s;
}

__global__ void __launch_bounds__(256) corr_resolve_kernel(const float* __restrict__ qn, const float* __restrict__ rn, int C, int Hq,
                                                            int Wq, int Hr, int Wr, const float* __restrict__ inv,
                                                            const float* __restrict__ qden, int norm_input,
                                                            const int* __restrict__ cnt, const int* __restrict__ cand,
                                                            long long nq_total, unsigned long long* __restrict__ keys,
                                                            ScanItem* __restrict__ items, int* __restrict__ flags,
                                                            int64_t* __restrict__ max_idx, float* __restrict__ max_val) {
  const int l = threadIdx.x & 63, grp = l / 9, tap = l - grp * 9;
  const long long q = ((long long)blockIdx.x * 4 + (threadIdx.x >> 6)) * 7 + grp;
  if (grp >= 7 || q >= nq_total) return;
  const int Nr = (Hr - 2) * (Wr - 2);
  const ExactScorer score = make_scorer(qn, rn, inv, C, Hq, Wq, Hr, Wr, q, tap, grp * 9);
  float best = -INFINITY;
  int bidx = 0x7fffffff;
  bool deferred = false;
  auto defer = [&](int lane) __attribute__((always_inline)) {
    if (tap == 0) {
      const int slot = atomicAdd(&flags[1], 1);
      if (slot < SCAN_CAP) items[slot] = ScanItem{q, lane, 0};
      else flags[0] = 1;   // work list full: the exact sweep takes over
    }
    deferred = true;
  };
  const int c = cnt[q];
  if (c >= 0) {
    for (int e = 0; e < c; ++e) {
      const int ent = cand[q * KSLOT + e];
      if (ent & SCAN_FLAG) {
        defer(ent & 31);
      } else if (ent < Nr) {
        const float v = score(ent);
        if (v > best || (v == best && ent < bidx)) { best = v; bidx = ent; }
      }
    }
  } else {
    defer(-1);
  }
  if (tap == 0) {
    if (deferred) {
      keys[q] = pack_key(best, bidx);
    } else {
      float v = best;
      if (norm_input) v = v / qden[q];
      max_idx[q] = (int64_t)bidx;
      max_val[q] = v;
    }
  }
}

// grid (SCAN_CHUNKS, SCAN_STRIPES): item = stripe, stripe + SCAN_STRIPES, ...; inside an item, candidate positions
// chunk * 28 + group, stepping 28 * SCAN_CHUNKS
__global__ void __launch_bounds__(256) corr_scan_kernel(const float* __restrict__ qn, const float* __restrict__ rn, int C, int Hq, int Wq,
                                                         int Hr, int Wr, const float* __restrict__ inv,
                                                         const ScanItem* __restrict__ items, const int* __restrict__ flags,
                                                         unsigned long long* __restrict__ keys) {
  const int count = min(flags[1], SCAN_CAP);
  const int l = threadIdx.x & 63, grp = l / 9, tap = l - grp * 9;
  if (grp >= 7) return;
  const int g = (threadIdx.x >> 6) * 7 + grp;   // 0..27
  const int Hrp = Hr - 2, Wrp = Wr - 2, Nr = Hrp * Wrp;
  for (int it = blockIdx.y; it < count; it += gridDim.y) {
    const ScanItem item = items[it];
    const ExactScorer score = make_scorer(qn, rn, inv, C, Hq, Wq, Hr, Wr, item.q, tap, grp * 9);
    const int jl = item.lane;
    const int N = jl < 0 ? Nr : ((jl < WP && jl < Wrp) ? ((Wrp - jl + WP - 1) / WP) * Hrp : 0);
    float best = -INFINITY;
    int bidx = 0x7fffffff;
    for (int p = blockIdx.x * 28 + g; p < N; p += 28 * gridDim.x) {
      const int n = jl < 0 ? p : (p % Hrp) * Wrp + jl + WP * (p / Hrp);
      const float v = score(n);
      if (v > best || (v == best && n < bidx)) { best = v; bidx = n; }
    }
    if (tap == 0 && bidx != 0x7fffffff) atomicMax(&keys[item.q], pack_key(best, bidx));
  }
}

__global__ void __launch_bounds__(256) corr_scan_finish_kernel(const ScanItem* __restrict__ items, const int* __restrict__ flags,
                                                                const unsigned long long* __restrict__ keys,
                                                                const float* __restrict__ qden, int norm_input,
                                                                int64_t* __restrict__ max_idx, float* __restrict__ max_val) {
  const int count = min(flags[1], SCAN_CAP);
  for (int it = blockIdx.x * 256 + threadIdx.x; it < count; it += gridDim.x * 256) {
    const long long q = items[it].q;
    float v;
    int n;
    unpack_key(keys[q], v, n);
    if (norm_input) v = v / qden[q];
    max_idx[q] = (int64_t)n;     // (several items of one query write the same pair)
    max_val[q] = v;
  }
}

template <int C>
static int launch_filter_c(hipStream_t st, const _Float16* qpl, const _Float16* rimg, int B, int Hq, int Wq, int Hr, int Wr,
                           const float* sc, const float* band, const int2* skip, int* cnt, int* cand) {
  const int tiles_y = ceil_div(Hq - 2, TPQ), tiles_x = ceil_div(Wq - 2, TPQ);
  const size_t lds = sizeof(float) * (size_t)(3 * SLAB) + 2 * (size_t)C * 128;
  // operand prefetch distance in k steps ($C2M_CORR_PF = 2: two steps ahead -- measured equal to one step on configs[2], 17.5 vs 17.6 ms, and it leaves no register to spare at C = 256)
  static const int pf = [] { const char* e = getenv("C2M_CORR_PF"); return (e && e[0] == '2') ? 2 : 1; }();
  static unsigned long long lds_set[2] = {0, 0};
  auto kern = pf == 1 ? &corr_filter_kernel<C, 1> : &corr_filter_kernel<C, 2>;
  if (int rc = ensure_dynamic_lds(reinterpret_cast<const void*>(kern), lds, lds_set[pf - 1])) return rc;
  ProfileScope prof(C2M_KERNEL_CORR_FILTER, st);
  hipLaunchKernelGGL(kern, dim3(B * tiles_y * tiles_x), dim3(NTHR), lds, st, qpl, rimg, Hq, Wq, Hr, Wr, tiles_y, tiles_x, sc, band,
                     skip, cnt, cand);
  return check_launch();
}

int launch(hipStream_t st, const float* fin, const float* fref, int B, int C, int Hq, int Wq, int Hr, int Wr, const float* inv,
           const float* qden, int norm_input, int2* skip, char* wsbase, const Ws& ws, int64_t* max_idx, float* max_val) {
  float* qn = reinterpret_cast<float*>(wsbase + ws.qn);
  float* rn = reinterpret_cast<float*>(wsbase + ws.rn);
  _Float16* qpl = reinterpret_cast<_Float16*>(wsbase + ws.qpl);
  _Float16* rimg = reinterpret_cast<_Float16*>(wsbase + ws.rimg);
  float* sc = reinterpret_cast<float*>(wsbase + ws.sb);
  float* band = reinterpret_cast<float*>(wsbase + ws.band);
  unsigned char* eq = reinterpret_cast<unsigned char*>(wsbase + ws.eq);
  int* cnt = reinterpret_cast<int*>(wsbase + ws.cnt);
  int* cand = reinterpret_cast<int*>(wsbase + ws.cand);
  int* flags = reinterpret_cast<int*>(wsbase + ws.flags);
  unsigned long long* keys = reinterpret_cast<unsigned long long*>(wsbase + ws.keys);
  ScanItem* items = reinterpret_cast<ScanItem*>(wsbase + ws.items);
  const int HWq = Hq * Wq, HWr = Hr * Wr, Hqp = Hq - 2, Wqp = Wq - 2, Hrp = Hr - 2, Wrp = Wr - 2;
  const int nxt = ceil_div(Wrp, WP);
  const long long nqp = (long long)B * Hqp * Wqp, npix_r = (long long)B * HWr;

  int* maxcol = reinterpret_cast<int*>(wsbase + ws.maxcol);
  static const int dead_tiles = [] { const char* e = getenv("C2M_CORR_DEAD_TILES"); return e ? atoi(e) : 1; }();   // (0: measurement)
  (void)hipMemsetAsync(flags, 0, 32, st);
  (void)hipMemsetAsync(maxcol, 0, sizeof(int) * (size_t)B, st);
  hipLaunchKernelGGL(to_nhwc_split_kernel, dim3(ceil_div(HWq, 64), C / 64, B), dim3(256), 0, st, fin, C, HWq, qn, qpl, flags);
  hipLaunchKernelGGL(to_nhwc_kernel, dim3(ceil_div(HWr, 64), C / 64, B), dim3(256), 0, st, fref, C, HWr, rn);
  hipLaunchKernelGGL(split_ref_image_kernel, dim3(Hr, nxt, B), dim3(256), 0, st, rn, C, Hr, Wr, nxt, rimg, flags);
  hipLaunchKernelGGL(pixel_eq_kernel, dim3((unsigned)((npix_r + 3) / 4)), dim3(256), 0, st, rn, C, Hr, Wr, npix_r, eq);
  hipLaunchKernelGGL(cand_scale_kernel, dim3(ceil_div(Hrp * Wrp, 256), B), dim3(256), 0, st, inv, eq, npix_r, Hr, Wr, Hrp, Wrp, sc,
                     flags, maxcol);
  if (dead_tiles) hipLaunchKernelGGL(dead_tiles_kernel, dim3(ceil_div(B * nxt, 64)), dim3(64), 0, st, maxcol, nxt, B * nxt, Hr, skip);
  hipLaunchKernelGGL(band_kernel, dim3((unsigned)((nqp + 255) / 256)), dim3(256), 0, st, qden, nqp, band);
  int rc = check_launch();
  if (rc != C2M_OK) return rc;
  if (C == 256) rc = launch_filter_c<256>(st, qpl, rimg, B, Hq, Wq, Hr, Wr, sc, band, skip, cnt, cand);
  else if (C == 128) rc = launch_filter_c<128>(st, qpl, rimg, B, Hq, Wq, Hr, Wr, sc, band, skip, cnt, cand);
  else rc = launch_filter_c<64>(st, qpl, rimg, B, Hq, Wq, Hr, Wr, sc, band, skip, cnt, cand);
  if (rc != C2M_OK) return rc;
  {
    ProfileScope prof(C2M_KERNEL_CORR_RESOLVE, st);
    hipLaunchKernelGGL(corr_resolve_kernel, dim3((unsigned)((nqp + 27) / 28)), dim3(256), 0, st, qn, rn, C, Hq, Wq, Hr, Wr, inv, qden,
                       norm_input, cnt, cand, nqp, keys, items, flags, max_idx, max_val);
    hipLaunchKernelGGL(corr_scan_kernel, dim3(SCAN_CHUNKS, SCAN_STRIPES), dim3(256), 0, st, qn, rn, C, Hq, Wq, Hr, Wr, inv, items, flags,
                       keys);
    hipLaunchKernelGGL(corr_scan_finish_kernel, dim3(8), dim3(256), 0, st, items, flags, keys, qden, norm_input, max_idx, max_val);
  }
  return check_launch();
}

}  // namespace corrf
}  // namespace c2m
