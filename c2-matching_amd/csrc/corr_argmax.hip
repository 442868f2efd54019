// corr_argmax.hip -- LR<->Ref 3x3 patch-feature correlation with running arg-max, for gfx950 (MI355X).
//
// Replaces sample_patches + feature_match_index (reference: mmsr/models/archs/ref_map_util.py:4-23, 26-86) and the
// caller's per-sample Python loop (mmsr/models/archs/corres_generation_arch.py:52-67) with one batched launch.
//
// The reference materialises every ref patch as a conv2d filter (9x copy of the ref map) and a [Nr x Nq] score
// volume per chunk.  Here neither exists.  The patch score is rewritten as a 9-tap diagonal sum of PIXEL-level
// channel dot products
//        S[q][n] = sum_{i,j in 0..2} D[q + (i,j)][n + (i,j)],   D[p][r] = sum_c in[c][p] * ref[c][r]
// which needs 8.6x fewer multiply-adds than the patch-level contraction.  D is produced by fp32 MFMA
// (v_mfma_f32_32x32x2_f32: a k-ordered fmaf chain, bit-identical to the C oracle's loop), never leaves the CU:
//
//   * one workgroup (8 waves) owns a 16x16 PIXEL tile of the query map = 14x14 query patches; each wave keeps its
//     32 query pixels x C channels resident in C/2 VGPRs as MFMA A-operands for the whole sweep, rows permuted so that
//     every lane ends up with one query pixel ROW in its 16 accumulator registers;
//   * the ref map is swept in x-tiles of 32 pixel columns (28 patch columns) and, inside an x-tile, one PIXEL ROW per
//     step; the row segment [C][32] is DMA'd global->LDS (double buffered, global_load_lds dwordx4) and is the
//     B-operand of all 8 waves;
//   * each step yields D[256 query pixels][32 ref pixels]; the three taps of one patch ROW are summed while the tile
//     is still in the accumulator registers (two DPP wave-shifted adds per element: H[px][j] = D[px][j] + (D[px+1][j+1] +
//     D[px+2][j+2])), and H goes to a 3-slab LDS ring (ref rows y-2, y-1, y); once row y is in, the patch row
//     ry = y-2 is complete: lanes (= ref x) add the three row sums, scale by the precomputed inverse patch norm and
//     update a per-lane, per-x-tile running (max, index) with a strict compare (candidates arrive in index order inside
//     an x-tile); the x-tile state is merged into the global one with the full (larger value, then lower index) rule.
//     Every non-MFMA instruction costs matrix time on this chip (scripts/ubench/issue_cost.hip), hence 3 LDS reads,
//     3 plain VALU ops, 1 compare and 2 selects per candidate;
//   * after the sweep a 32-lane shuffle reduction with the (larger value, then lower index) rule yields the
//     reference's "first maximum" semantics exactly, independent of the visiting order;
//   * ref pixel rows that repeat the three rows above them bit for bit (the band a zero-padded Ref leaves,
//     ref_cufed_dataset.py:99-114) are not swept: their patches can only tie with an earlier, lower-index patch
//     (ref_row_equal_kernel / ref_row_run_kernel build the per-(sample, x-tile) skip table).
//
// LDS: ring 3*256*32*4 = 96 KiB + row buffers 2*C*32*4 = 64 KiB (C=256) = all 160 KiB of a CU; 1 workgroup per CU,
// 2 waves per SIMD.  Roofline: MFMA (fp32 matrix peak 157.3 TF); HBM traffic is a few times the compulsory 53 MB/pair
// and <1 % of the HBM roofline.
#include <stdlib.h>

#include "c2m_common.h"
#include "corr_filter.h"

namespace c2m {

// ------------------------------------------------------------------------------------------------------------------
// small prologue kernels
// ------------------------------------------------------------------------------------------------------------------

// F.normalize over channels (corres_generation_arch.py:56-58): thread = pixel, coalesced over pixels per channel.
// Generic version: two passes over the pixel's channel column (the second read mostly misses L2 at 160x160x256).
// ss_out (or nullptr): the per-pixel sum of squares of the NORMALISED values, the same canonical fmaf chain (c ascending) over
// the same stored floats as pixel_sumsq_kernel -- what c2m_feature_match_index_pre_f32 takes instead of re-reading the map.
__global__ void __launch_bounds__(256) feature_normalize_kernel(const float* __restrict__ x, int C, int HW,
                                                                 float* __restrict__ out, float* __restrict__ ss_out) {
  const int p = blockIdx.x * 256 + threadIdx.x;
  if (p >= HW) return;
  const float* xb = x + (size_t)blockIdx.y * C * HW + p;
  float* ob = out + (size_t)blockIdx.y * C * HW + p;
  float ss = 0.0f;
  for (int c = 0; c < C; ++c) {
    const float v = xb[(size_t)c * HW];
    ss = fmaf(v, v, ss);
  }
  const float nrm = sqrtf(ss);
  const float den = nrm > 1e-12f ? nrm : 1e-12f;
  float s2 = 0.0f;
  for (int c = 0; c < C; ++c) {
    const float o = xb[(size_t)c * HW] / den;
    ob[(size_t)c * HW] = o;
    s2 = fmaf(o, o, s2);
  }
  if (ss_out) ss_out[(size_t)blockIdx.y * HW + p] = s2;
}

// Same arithmetic (one fmaf chain, c ascending), but the pixel's whole channel column stays in registers between the
// two passes: one HBM read + one write per element, C independent loads in flight per lane.  One wave per SIMD
// (C + a few VGPRs), which is plenty for a pure streaming kernel.
template <int C>
__global__ void __launch_bounds__(256, 1) feature_normalize_reg_kernel(const float* __restrict__ x, int HW,
                                                                        float* __restrict__ out, float* __restrict__ ss_out) {
  const int p = blockIdx.x * 256 + threadIdx.x;
  if (p >= HW) return;
  const float* xb = x + (size_t)blockIdx.y * C * HW + p;
  float* ob = out + (size_t)blockIdx.y * C * HW + p;
  float v[C];
#pragma unroll
  for (int c = 0; c < C; ++c) v[c] = xb[(size_t)c * HW];
  float ss = 0.0f;
#pragma unroll
  for (int c = 0; c < C; ++c) ss = fmaf(v[c], v[c], ss);
  const float nrm = sqrtf(ss);
  const float den = nrm > 1e-12f ? nrm : 1e-12f;
  float s2 = 0.0f;
#pragma unroll
  for (int c = 0; c < C; ++c) {
    const float o = v[c] / den;
    ob[(size_t)c * HW] = o;
    s2 = fmaf(o, o, s2);
  }
  if (ss_out) ss_out[(size_t)blockIdx.y * HW + p] = s2;
}

// per-pixel sum of squares over channels (canonical fmaf chain, c ascending)
__global__ void __launch_bounds__(256) pixel_sumsq_kernel(const float* __restrict__ x, int C, int HW,
                                                           float* __restrict__ ss) {
  const int p = blockIdx.x * 256 + threadIdx.x;
  if (p >= HW) return;
  const float* xb = x + (size_t)blockIdx.y * C * HW + p;
  float a = 0.0f;
  for (int c = 0; c < C; ++c) {
    const float v = xb[(size_t)c * HW];
    a = fmaf(v, v, a);
  }
  ss[(size_t)blockIdx.y * HW + p] = a;
}

// patch norm over (C,P,P) from the pixel sums (ref_map_util.py:63 / :80): row-major plain adds, sqrt, + 1e-5.
// invert != 0 -> 1/(norm + 1e-5) (the factor applied to ref patches); else norm + 1e-5 (the query-side divisor).
__global__ void __launch_bounds__(256) patch_norm_kernel(const float* __restrict__ ss, int H, int W, int P,
                                                          int stride, int Hp, int Wp, int invert,
                                                          float* __restrict__ out) {
  const int n = blockIdx.x * 256 + threadIdx.x;
  if (n >= Hp * Wp) return;
  const int py = n / Wp, px = n - py * Wp;
  const float* s = ss + (size_t)blockIdx.y * H * W + (size_t)(py * stride) * W + px * stride;
  float a = s[0];
  for (int i = 0; i < P; ++i)
    for (int j = 0; j < P; ++j) {
      if (i == 0 && j == 0) continue;
      a = a + s[i * W + j];
    }
  const float d = sqrtf(a) + 1e-5f;
  out[(size_t)blockIdx.y * Hp * Wp + n] = invert ? 1.0f / d : d;
}

// ------------------------------------------------------------------------------------------------------------------
// generic kernel: any patch size / strides / channel count.  One workgroup per query patch; the query patch's
// P*P*C values sit in LDS, threads stride over ref patches (coalesced over rx).  Same arithmetic as the MFMA
// kernel and the oracle -> bit-identical results.  Used for configurations the fast kernel does not cover and
// as an on-device cross-check.
// ------------------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) corr_argmax_generic_kernel(
    const float* __restrict__ fin, const float* __restrict__ fref, int C, int Hq, int Wq, int Hr, int Wr, int P,
    int sq, int sr, int Wqp, int Hrp, int Wrp, const float* __restrict__ inv, const float* __restrict__ qden,
    int64_t* __restrict__ max_idx, float* __restrict__ max_val) {
  extern __shared__ __attribute__((aligned(16))) float qlds[];  // [P*P][C]
  __shared__ float red_v[256];
  __shared__ int red_i[256];
  const int b = blockIdx.y, q = blockIdx.x;
  const int qy = q / Wqp, qx = q - qy * Wqp;
  const int HWq = Hq * Wq, HWr = Hr * Wr, Nrp = Hrp * Wrp, PP = P * P;
  const float* fi = fin + (size_t)b * C * HWq;
  const float* fr = fref + (size_t)b * C * HWr;
  for (int e = threadIdx.x; e < PP * C; e += 256) {
    const int t = e / C, c = e - t * C;
    const int i = t / P, j = t - i * P;
    qlds[e] = fi[(size_t)c * HWq + (qy * sq + i) * Wq + qx * sq + j];
  }
  __syncthreads();
  float best = -INFINITY;
  int bidx = 0x7fffffff;
  for (int n = threadIdx.x; n < Nrp; n += 256) {
    const int ry = n / Wrp, rx = n - ry * Wrp;
    float s = 0.0f;
    for (int i = 0; i < P; ++i) {          // rows left-to-right, taps of a row right-to-left (canonical order)
      float row = 0.0f;
      for (int j = P - 1; j >= 0; --j) {
        const float* r = fr + (ry * sr + i) * Wr + rx * sr + j;
        const float* ql = qlds + (i * P + j) * C;
        float d = 0.0f;
        for (int c = 0; c < C; ++c) d = fmaf(ql[c], r[(size_t)c * HWr], d);
        row = (j == P - 1) ? d : d + row;
      }
      s = (i == 0) ? row : s + row;
    }
    const float v = inv ? s * inv[(size_t)b * Nrp + n] : s;
    if (v > best || (v == best && n < bidx)) { best = v; bidx = n; }
  }
  red_v[threadIdx.x] = best;
  red_i[threadIdx.x] = bidx;
  __syncthreads();
  for (int off = 128; off > 0; off >>= 1) {
    if (threadIdx.x < off) {
      const float v2 = red_v[threadIdx.x + off];
      const int i2 = red_i[threadIdx.x + off];
      if (v2 > red_v[threadIdx.x] || (v2 == red_v[threadIdx.x] && i2 < red_i[threadIdx.x])) {
        red_v[threadIdx.x] = v2;
        red_i[threadIdx.x] = i2;
      }
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    float v = red_v[0];
    if (qden) v = v / qden[(size_t)b * gridDim.x + q];
    max_idx[(size_t)b * gridDim.x + q] = (int64_t)red_i[0];
    max_val[(size_t)b * gridDim.x + q] = v;
  }
}

// ------------------------------------------------------------------------------------------------------------------
// MFMA sliding-window kernel (patch 3, strides 1)
// ------------------------------------------------------------------------------------------------------------------
namespace corr {
constexpr int TQ = 16;            // query tile side, pixels
constexpr int TPQ = TQ - 2;       // query patches per tile side (14)
constexpr int WT = 32;            // ref x-tile width, pixels (= MFMA N)
constexpr int WP = 28;            // ref patches per x-tile; a multiple of 4 keeps every tile origin 16-byte aligned
                                  // for the dwordx4 row DMA (30 of the 32 loaded pixel columns are used)
constexpr int NWAVE = 8;
constexpr int NTHR = NWAVE * 64;
constexpr int QPIX = TQ * TQ;     // 256 query pixels per tile
constexpr int SLAB = QPIX * WT;   // floats per ring slab
constexpr int NIT = TPQ;          // 14 tap-sum rounds: round `it` = query patch ROW it, half-wave (w, hi) = patch column 2w + hi
}  // namespace corr

// DMA16: ref rows are fetched with global_load_lds_dwordx4 (needs Wr % 4 == 0 and a 16-byte aligned ref base); otherwise
// one dword per lane.
template <int C, bool DMA16>
__global__ void __launch_bounds__(corr::NTHR, 2) corr_argmax_mfma_kernel(
    const float* __restrict__ fin, const float* __restrict__ fref, int Hq, int Wq, int Hr, int Wr, int tiles_y,
    int tiles_x, const float* __restrict__ inv, const float* __restrict__ qden, int is_norm, int norm_input,
    const int2* __restrict__ skip, const int* __restrict__ need, int64_t* __restrict__ max_idx,
    float* __restrict__ max_val) {
  using namespace corr;
  // need != nullptr: this launch is the fall-back of the pre-filter path (corr_filter.hip) and runs only if the filter's
  // preparation found inputs outside its domain (*need != 0); otherwise every workgroup returns at once
  if (need != nullptr && *need == 0) return;
  constexpr int KP = C / 2;                    // MFMA k-pairs = resident A registers per lane
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* ring = smem;                          // [3][QPIX][WT]
  float* rbuf = smem + 3 * SLAB;               // [2][C][WT]

  const int ntile = tiles_y * tiles_x;
  const int vb = xcd_remap(blockIdx.x, gridDim.x);  // sample-major: tiles of one sample share an XCD's L2
  const int b = vb / ntile, tile = vb - b * ntile;
  const int ty = tile / tiles_x, tx = tile - ty * tiles_x;
  const int qy0 = ty * TPQ, qx0 = tx * TPQ;
  const int Hqp = Hq - 2, Wqp = Wq - 2, Hrp = Hr - 2, Wrp = Wr - 2;

  const int tid = threadIdx.x;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l = tid & 63, hi = l >> 5, j32 = l & 31;

  const float* fi = fin + (size_t)b * C * Hq * Wq;
  const float* fr = fref + (size_t)b * C * Hr * Wr;
  // `inv` is always a valid global pointer (workspace); is_norm says whether it holds the ref-patch inverse norms
  // (a pointer selected against nullptr degrades the load to FLAT, which shares lgkmcnt with the LDS reads)
  const bool has_inv = is_norm != 0;
  const float* __restrict__ invb = inv + (size_t)b * Hrp * Wrp;

  // ---- resident A operands.  MFMA row i of wave w is tile pixel (row 2w + ((i>>2)&1), column (i&3) + 4*(i>>3)): with the
  //      32x32 accumulator layout (lane (hi, j) holds rows (r&3) + 8*(r>>2) + 4*hi, r = 0..15) every lane then owns ONE
  //      query pixel row (2w + hi) with its 16 pixel columns in registers r = 0..15, so "one pixel to the right" is the
  //      next register of the same lane.  Lane (i = l&31, k = l>>5) of k-pair t holds in[2t + k][pixel(i)].
  float qreg[KP];
  {
    const int py = min(qy0 + 2 * w + ((j32 >> 2) & 1), Hq - 1);
    const int px = min(qx0 + (j32 & 3) + 4 * (j32 >> 3), Wq - 1);
    const float* src = fi + (size_t)hi * Hq * Wq + (size_t)py * Wq + px;
#pragma unroll
    for (int t = 0; t < KP; ++t) qreg[t] = src[(size_t)(2 * t) * Hq * Wq];
  }

  // ---- running arg-max state.  Lane (w, hi, j32) scores query patch (row it, column qx = 2w + hi) against ref column
  //      j32 of the current x-tile.  Inside one x-tile a lane meets its candidates in ascending index order, so the
  //      tile-local state (bt, bi) needs only "strictly greater" (one compare); it is merged into (best, bidx) with the
  //      full (larger value, then lower index) rule once per x-tile.  Wave 7 repeats column 13 (never stored).
  float best[NIT], bt[NIT];
  int bidx[NIT], bi[NIT];
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    best[it] = -INFINITY;
    bt[it] = -INFINITY;
    bidx[it] = 0x7fffffff;
    bi[it] = 0x7fffffff;
  }
  const int qx = min(2 * w + hi, TPQ - 1);
  const int lane_off = qx * WT + j32;   // float offset of H[(row 0, qx)][j32] inside a slab

  // Duplicate-row elimination.  skip[b][xt] = (from, to): inside x-tile xt the ref pixel rows [from - 3, to) are bitwise
  // identical over every channel and every column the tile loads (ref_row_run_kernel below), so the patch rows
  // from-2 .. to-3 are exact copies of patch row from-3: same scores, same inverse norms, higher index -- they can never
  // win the (larger value, then LOWER index) rule.  The sweep therefore jumps from row from-1 to row `to`; the ring still
  // holds the row sums of rows from-2, from-1, which are the row sums of rows to-2, to-1.  (from == to: nothing skipped.)
  // A second writer, corr_filter.hip's dead_tiles_kernel, marks whole x-tiles (0, Hr): from = 0 breaks the "from >= 3" contract
  // above and is only correct because such tiles are a TRAILING set (S below ends the walk before it would enter one; the
  // invariant and its test are documented at dead_tiles_kernel).
  const int nxt = (Wrp + WP - 1) / WP;
  const int2* __restrict__ skb = skip + (size_t)b * nxt;
  int S = 0;
  for (int i = 0; i < nxt; ++i) S += Hr - (skb[i].y - skb[i].x);
  int2 sk = skb[0];

  // DMA of ref pixel row (xt, y) into rbuf[buf] = [C][32]: wave w copies channels [C/8 * w, C/8 * (w+1))
  auto issue_row = [&](int xt, int y, int buf) {
    float* d = rbuf + buf * (C * WT) + w * (C / NWAVE) * WT;
    if constexpr (DMA16) {
      // one instruction = 8 channels x 32 pixels: lane (cs = l>>3, quad = l&7) moves pixels [4 quad, 4 quad + 4)
      const int x = min(xt * WP + 4 * (l & 7), Wr - 4);
      const float* g = fr + (size_t)(w * (C / NWAVE) + (l >> 3)) * Hr * Wr + (size_t)y * Wr + x;
#pragma unroll
      for (int m = 0; m < C / NWAVE / 8; ++m) glds_b128(g + (size_t)(8 * m) * Hr * Wr, d + (8 * m) * WT);
    } else {
      const int x = min(xt * WP + j32, Wr - 1);
      const float* g = fr + (size_t)(w * (C / NWAVE) + hi) * Hr * Wr + (size_t)y * Wr + x;
#pragma unroll
      for (int m = 0; m < C / NWAVE / 2; ++m) glds_b32(g + (size_t)(2 * m) * Hr * Wr, d + (2 * m) * WT);
    }
  };

  issue_row(0, 0, 0);

  // Step s processes ref pixel row y of x-tile xt and parks its row sums H in ring slab s % 3.  While its MFMA chain
  // runs, the lanes finish the patch row completed by step s-1 (rows of steps s-3, s-2, s-1 = slabs s%3, (s+1)%3,
  // (s+2)%3).  Iteration s == S only drains the last pending patch row (its MFMA result is never read).
  int y = 0, xt = 0;   // row / x-tile of step s
  int sl0 = 0;         // s % 3
  // candidate of the NEXT iteration, prefetched one step ahead so its global load never sits between a row DMA and
  // that DMA's wait.  scale = NaN marks "no candidate" (NaN never compares greater).
  int n = 0;
  float scale_next = __builtin_nanf("");
  for (int s = 0; s <= S; ++s) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();  // (A) row s landed in rbuf[s&1]; slab of step s-1 complete; everybody left step s-1
    float scale = scale_next;
    asm volatile("" : "+v"(scale));  // consume the prefetched load before any DMA is in flight

    // next row's DMA into the other buffer (its last readers were the MFMAs of step s-1)
    int yn = y + 1, xtn = xt;
    if (yn == sk.x) yn = sk.y;
    if (yn >= Hr) { yn = 0; xtn = xt + 1; }
    if (s + 1 < S) issue_row(xtn, yn, (s + 1) & 1);

    // candidate handled by the NEXT iteration: the patch row completed by THIS step (row y of x-tile xt).  Its inverse
    // norm is fetched now, a whole step before it is needed, so the load never sits in front of barrier (B).
    const int ncol_nx = xt * WP + j32;
    const bool cand_nx = (s < S) && (y >= 2) && (j32 < WP) && (ncol_nx < Wrp);
    const int n_nx = (y - 2) * Wrp + ncol_nx;
    scale_next = __builtin_nanf("");
    if (cand_nx) scale_next = has_inv ? invb[n_nx] : 1.0f;

    const int sl1 = (sl0 == 2) ? 0 : sl0 + 1;
    const int sl2 = (sl1 == 2) ? 0 : sl1 + 1;
    // row sums of tap row i for query patch (row it, column qx) live at a_i[it * TQ * WT]: immediate offsets only
    const float* a0 = ring + sl0 * SLAB + lane_off;                 // ref row ry     (tap row i = 0)
    const float* a1 = ring + sl1 * SLAB + TQ * WT + lane_off;       // ref row ry + 1 (i = 1: query pixel row + 1)
    const float* a2 = ring + sl2 * SLAB + 2 * TQ * WT + lane_off;   // ref row ry + 2 (i = 2)

    const float* bsrc = rbuf + (s & 1) * (C * WT) + l;  // B operand of k-pair t: rbuf[2t + hi][j32] = bsrc[t * 64]
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.0f;

    // Software pipeline over the 14 rounds.  Round `it`: the first MFMA of group `it` consumes operands fetched a whole
    // round earlier (the compiler's s_waitcnt lgkmcnt(0) in front of it is then free); only after it are the B operands
    // of group it+1 and the three row sums of round it+1 requested, to land under the remaining MFMAs of the group.
    constexpr int KPG = (KP + NIT - 1) / NIT;
    float bq[2][KPG], hq[2][3];
#pragma unroll
    for (int t = 0; t < KP / NIT; ++t) bq[0][t] = bsrc[t * 64];
    hq[0][0] = a0[0]; hq[0][1] = a1[0]; hq[0][2] = a2[0];
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int t0 = it * KP / NIT, t1 = (it + 1) * KP / NIT, t2 = (it + 2) * KP / NIT;
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(qreg[t0], bq[it & 1][0], acc, 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      if (it + 1 < NIT) {
#pragma unroll
        for (int t = t1; t < t2; ++t) bq[(it + 1) & 1][t - t1] = bsrc[t * 64];
        hq[(it + 1) & 1][0] = a0[(it + 1) * TQ * WT];
        hq[(it + 1) & 1][1] = a1[(it + 1) * TQ * WT];
        hq[(it + 1) & 1][2] = a2[(it + 1) * TQ * WT];
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int t = t0 + 1; t < t1; ++t)
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(qreg[t], bq[it & 1][t - t0], acc, 0, 0, 0);
      // (row_0 + row_1) + row_2 (oracle order); strict compare against the tile-local state
      float sum = hq[it & 1][0] + hq[it & 1][1];
      sum = sum + hq[it & 1][2];
      const float v = sum * scale;   // scale is exactly 1.0f without is_norm, NaN without a candidate
      const bool take = v > bt[it];
      bt[it] = take ? v : bt[it];
      bi[it] = take ? n : bi[it];
      __builtin_amdgcn_sched_barrier(0);
    }
    n = n_nx;

    if (y == 0 && s > 0) {
      // the rounds above finished the last patch row of an x-tile: merge its state (larger value, then lower index).
      // The empty volatile asm keeps this a real (wave-uniform) branch: if-converted it would run on every step.
      asm volatile("" ::: "memory");
#pragma unroll
      for (int it = 0; it < NIT; ++it) {
        const bool take = (bt[it] > best[it]) | ((bt[it] == best[it]) & (bi[it] < bidx[it]));
        best[it] = take ? bt[it] : best[it];
        bidx[it] = take ? bi[it] : bidx[it];
        bt[it] = -INFINITY;
        bi[it] = 0x7fffffff;
      }
    }

    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();  // (B) every tap-sum that reads slab sl0 (about to be overwritten) is done

    // Row sums of the 3 taps of a patch row, formed in registers.  Lane (hi, j) holds D[pixel (2w+hi, r)][ref column j]
    // in acc[r]; V at (pixel column + 1, ref column + 1) is the next register shifted by one lane (DPP wave_shl:1):
    //   t[r] = D[r] + shl(D[r+1])      = D[px][j] + D[px+1][j+1]
    //   H[r] = D[r] + shl(t[r+1])      = D[px][j] + (D[px+1][j+1] + D[px+2][j+2])          (oracle tap order)
    // Only patch origins px < 14, j < 28 are ever read back.
    auto shl1 = [](float x) __attribute__((always_inline)) {
      return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x130, 0xf, 0xf, true));
    };
    {
      float dv[16], tt[16], hh[TPQ];
#pragma unroll
      for (int r = 0; r < 16; ++r) dv[r] = acc[r];
#pragma unroll
      for (int r = 1; r < 15; ++r) tt[r] = dv[r] + shl1(dv[r + 1]);
#pragma unroll
      for (int r = 0; r < TPQ; ++r) hh[r] = dv[r] + shl1(tt[r + 1]);
      float* dst = ring + sl0 * SLAB + (2 * w + hi) * TQ * WT + j32;
#pragma unroll
      for (int r = 0; r < TPQ; ++r) dst[r * WT] = hh[r];
    }

    if (xtn != xt) sk = skb[min(xtn, nxt - 1)];
    y = yn;
    xt = xtn;
    sl0 = sl1;
  }

  // ---- final reduction over the 32 ref-column lanes of each half-wave, then one store per query patch ----
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    float v = best[it];
    int i = bidx[it];
#pragma unroll
    for (int m = 16; m > 0; m >>= 1) {
      const float v2 = __shfl_xor(v, m, 64);
      const int i2 = __shfl_xor(i, m, 64);
      const bool take = (v2 > v) || (v2 == v && i2 < i);
      v = take ? v2 : v;
      i = take ? i2 : i;
    }
    const int gy = qy0 + it, gx = qx0 + 2 * w + hi;
    if (j32 == 0 && w < NWAVE - 1 && gy < Hqp && gx < Wqp) {
      const size_t o = (size_t)b * Hqp * Wqp + (size_t)gy * Wqp + gx;
      if (norm_input) v = v / qden[o];
      max_idx[o] = (int64_t)i;
      max_val[o] = v;
    }
  }
}

// ------------------------------------------------------------------------------------------------------------------
// Duplicate ref rows (see corr_argmax_mfma_kernel).  A zero-padded Ref (ref_cufed_dataset.py:99-114 pads every test Ref
// to the HR size) leaves a band of identical feature rows; the reference scores every one of them and its strict '>'
// keeps the first.  ref_row_equal_kernel: eq[b][xt][y] = rows y and y+1 agree bitwise over all channels inside the
// pixel columns x-tile xt loads.  ref_row_run_kernel: the longest run per (sample, x-tile) -> (from, to).
// ------------------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) ref_row_equal_kernel(const float* __restrict__ fref, int C, int Hr, int Wr,
                                                             int nxt, int* __restrict__ eq) {
  const int y = blockIdx.x, xt = blockIdx.y, b = blockIdx.z;
  const int x0 = xt * corr::WP, ncol = min(corr::WT, Wr - x0);
  const uint32_t* r0 = reinterpret_cast<const uint32_t*>(fref) + (size_t)b * C * Hr * Wr + (size_t)y * Wr + x0;
  int ok = 1;
  for (int e = threadIdx.x; e < C * corr::WT; e += 256) {
    const int c = e / corr::WT, j = e - c * corr::WT;
    if (j < ncol) {
      const uint32_t* q = r0 + (size_t)c * Hr * Wr + j;
      ok &= (q[0] == q[Wr]) ? 1 : 0;
    }
  }
  ok = __syncthreads_and(ok);
  if (threadIdx.x == 0) eq[((size_t)b * nxt + xt) * Hr + y] = ok;
}

__global__ void ref_row_run_kernel(const int* __restrict__ eq, int Hr, int n, int2* __restrict__ skip) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;  // (sample, x-tile)
  if (i >= n) return;
  const int* e = eq + (size_t)i * Hr;
  int best0 = 0, best1 = 0, start = 0;  // rows [best0, best1] identical
  for (int y = 0; y < Hr; ++y) {
    const bool cont = (y < Hr - 1) && e[y];
    if (!cont) {
      if (y - start > best1 - best0) { best0 = start; best1 = y; }
      start = y + 1;
    }
  }
  // rows best0 .. best1 identical: patch rows best0+1 .. best1-2 duplicate patch row best0
  skip[i] = (best1 - best0 >= 3) ? make_int2(best0 + 3, best1 + 1) : make_int2(Hr, Hr);
}

// ------------------------------------------------------------------------------------------------------------------
// pre-offset builder: index_to_flow (corres_generation_arch.py:29-46) + tensor_shift x9 at scales 1, 2, 4
// (:69-104, arch_util.py:291-315).  One thread per output pixel of one (sample, shift); writes (x, y) as float2.
// ------------------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) pre_offset_kernel(const int64_t* __restrict__ max_idx, int h, int w, int s,
                                                          float2* __restrict__ out) {
  const int H = h * s, W = w * s, hp = h - 2, wp = w - 2;
  const int pix = blockIdx.x * 256 + threadIdx.x;
  if (pix >= H * W) return;
  const int k = blockIdx.y, b = blockIdx.z;
  const int y = pix / W, x = pix - y * W;
  const int ys = y - (k / 3) * s, xs = x - (k % 3) * s;
  float2 f = make_float2(0.0f, 0.0f);
  if (ys >= 0 && xs >= 0) {
    const int yy = ys / s, xx = xs / s;
    if (yy < hp && xx < wp) {
      const int64_t idx = max_idx[(size_t)b * hp * wp + (size_t)yy * wp + xx];
      f.x = (float)((int)(idx % wp) - xx) * (float)s;  // flow decoded with the QUERY grid width (:32-34)
      f.y = (float)((int)(idx / wp) - yy) * (float)s;
    }
  }
  out[((size_t)b * 9 + k) * H * W + pix] = f;
}

}  // namespace c2m

// ====================================================================================================================
// C-ABI
// ====================================================================================================================
using namespace c2m;

extern "C" int c2m_feature_normalize_ss_f32(c2m_stream_t stream, const float* x, int B, int C, int HW, float* out, float* ss_out) {
  if (!x || !out || B <= 0 || C <= 0 || HW <= 0) return C2M_ERR_INVALID_ARG;
  dim3 grid(ceil_div(HW, 256), B);
  hipStream_t st = as_stream(stream);
  if (C == 256) hipLaunchKernelGGL(feature_normalize_reg_kernel<256>, grid, dim3(256), 0, st, x, HW, out, ss_out);
  else if (C == 128) hipLaunchKernelGGL(feature_normalize_reg_kernel<128>, grid, dim3(256), 0, st, x, HW, out, ss_out);
  else if (C == 64) hipLaunchKernelGGL(feature_normalize_reg_kernel<64>, grid, dim3(256), 0, st, x, HW, out, ss_out);
  else hipLaunchKernelGGL(feature_normalize_kernel, grid, dim3(256), 0, st, x, C, HW, out, ss_out);
  return check_launch();
}

extern "C" int c2m_feature_normalize_f32(c2m_stream_t stream, const float* x, int B, int C, int HW, float* out) {
  return c2m_feature_normalize_ss_f32(stream, x, B, C, HW, out, nullptr);
}

namespace {
struct CorrWs {
  size_t ss_ref, inv, ss_in, qden, row_eq, skip, total;  // byte offsets
  int nxt;
  c2m::corrf::Ws f;   // scratch of the pre-filter path (corr_filter.h)
};
inline size_t align256(size_t v) { return (v + 255) & ~size_t(255); }
inline CorrWs corr_ws(int B, int Hq, int Wq, int Hr, int Wr, int C = c2m::corrf::CMAX) {
  CorrWs w;
  size_t o = 0;
  w.ss_ref = o; o = align256(o + sizeof(float) * (size_t)B * Hr * Wr);
  w.inv = o;    o = align256(o + sizeof(float) * (size_t)B * Hr * Wr);
  w.ss_in = o;  o = align256(o + sizeof(float) * (size_t)B * Hq * Wq);
  w.qden = o;   o = align256(o + sizeof(float) * (size_t)B * Hq * Wq);
  w.nxt = Wr > 2 ? (Wr - 2 + c2m::corr::WP - 1) / c2m::corr::WP : 1;
  w.row_eq = o; o = align256(o + sizeof(int) * (size_t)B * w.nxt * Hr);
  w.skip = o;   o = align256(o + sizeof(int2) * (size_t)B * w.nxt);
  w.f = c2m::corrf::workspace(o, B, Hq, Wq, Hr, Wr, C);
  w.total = w.f.total;
  return w;
}

template <int C>
int launch_corr_mfma(hipStream_t st, const float* fin, const float* fref, int B, int Hq, int Wq, int Hr, int Wr,
                     const float* inv, const float* qden, int* row_eq, int2* skip, int dedup, int64_t* max_idx,
                     float* max_val, const c2m::corrf::Ws* fws = nullptr, char* wsbase = nullptr, const float* qden_buf = nullptr) {
  using namespace c2m::corr;
  const int tiles_y = ceil_div(Hq - 2, TPQ), tiles_x = ceil_div(Wq - 2, TPQ);
  const size_t lds = sizeof(float) * (size_t)(3 * SLAB + 2 * C * WT);
  static unsigned long long lds_set[2] = {0, 0};
  const bool dma16 = (Wr % 4 == 0) && (reinterpret_cast<uintptr_t>(fref) % 16 == 0);
  auto kern = dma16 ? &corr_argmax_mfma_kernel<C, true> : &corr_argmax_mfma_kernel<C, false>;
  if (int rc = ensure_dynamic_lds(reinterpret_cast<const void*>(kern), lds, lds_set[dma16])) return rc;
  dim3 grid(B * tiles_y * tiles_x);
  const int nxt = ceil_div(Wr - 2, WP);
  if (dedup) {
    hipLaunchKernelGGL(ref_row_equal_kernel, dim3(Hr - 1, nxt, B), dim3(256), 0, st, fref, C, Hr, Wr, nxt, row_eq);
  } else {
    (void)hipMemsetAsync(row_eq, 0, sizeof(int) * (size_t)B * nxt * Hr, st);
  }
  hipLaunchKernelGGL(ref_row_run_kernel, dim3(ceil_div(B * nxt, 64)), dim3(64), 0, st, row_eq, Hr, B * nxt, skip);
  const int* need = nullptr;
  if (fws) {
    // pre-filter + exact re-score; the sweep below then runs only if the filter's preparation raised its flag
    if (int rc = c2m::corrf::launch(st, fin, fref, B, C, Hq, Wq, Hr, Wr, inv, qden_buf, qden ? 1 : 0, skip, wsbase, *fws, max_idx,
                                    max_val))
      return rc;
    need = reinterpret_cast<const int*>(wsbase + fws->flags);
  }
  ProfileScope prof(C2M_KERNEL_CORR_MFMA, st);
  hipLaunchKernelGGL(kern, grid, dim3(NTHR), lds, st, fin, fref, Hq, Wq, Hr, Wr, tiles_y, tiles_x, inv ? inv : fin,
                     qden ? qden : fin, inv ? 1 : 0, qden ? 1 : 0, skip, need, max_idx, max_val);
  return check_launch();
}
}  // namespace

namespace {
thread_local int g_filter_mode = -1;   // -1: $C2M_CORR_FILTER (default on), 0 / 1: forced by c2m_feature_match_set_filter
}

extern "C" int c2m_feature_match_set_filter(int mode) {
  if (mode < -1 || mode > 1) return C2M_ERR_INVALID_ARG;
  g_filter_mode = mode;
  return C2M_OK;
}

extern "C" int c2m_feature_match_filter_tables(int B, int Hq, int Wq, int Hr, int Wr, size_t* cnt_offset, size_t* cand_offset,
                                               size_t* flags_offset, int* slots) {
  if (B <= 0 || Hq <= 0 || Wq <= 0 || Hr <= 0 || Wr <= 0 || !cnt_offset || !cand_offset || !flags_offset || !slots)
    return C2M_ERR_INVALID_ARG;
  const CorrWs ws = corr_ws(B, Hq, Wq, Hr, Wr);
  *cnt_offset = ws.f.cnt;
  *cand_offset = ws.f.cand;
  *flags_offset = ws.f.flags;
  *slots = c2m::corrf::KSLOT;
  return C2M_OK;
}

extern "C" size_t c2m_feature_match_workspace_bytes(int B, int Hq, int Wq, int Hr, int Wr) {
  if (B <= 0 || Hq <= 0 || Wq <= 0 || Hr <= 0 || Wr <= 0) return 0;
  return corr_ws(B, Hq, Wq, Hr, Wr).total;
}

// channels the pre-filter's per-channel scratch is laid out for: C where the filter has a kernel, else none
static int filter_channels(int C) { return (C == 64 || C == 128 || C == 256) ? C : 0; }

extern "C" size_t c2m_feature_match_workspace_bytes_c(int B, int C, int Hq, int Wq, int Hr, int Wr) {
  if (B <= 0 || C <= 0 || Hq <= 0 || Wq <= 0 || Hr <= 0 || Wr <= 0) return 0;
  return corr_ws(B, Hq, Wq, Hr, Wr, filter_channels(C)).total;
}

extern "C" int c2m_feature_match_skip_table(int B, int Hq, int Wq, int Hr, int Wr, size_t* byte_offset, int* x_tiles) {
  if (B <= 0 || Hq <= 0 || Wq <= 0 || Hr <= 0 || Wr <= 0 || !byte_offset || !x_tiles) return C2M_ERR_INVALID_ARG;
  const CorrWs ws = corr_ws(B, Hq, Wq, Hr, Wr);
  *byte_offset = ws.skip;
  *x_tiles = ws.nxt;
  return C2M_OK;
}

extern "C" int c2m_feature_match_index_f32(c2m_stream_t stream, const float* feat_in, const float* feat_ref, int B,
                                           int C, int Hq, int Wq, int Hr, int Wr, int patch, int in_stride,
                                           int ref_stride, int is_norm, int norm_input, int force_generic,
                                           int64_t* max_idx, float* max_val, void* workspace,
                                           size_t workspace_bytes) {
  return c2m_feature_match_index_pre_f32(stream, feat_in, feat_ref, B, C, Hq, Wq, Hr, Wr, patch, in_stride, ref_stride, is_norm,
                                         norm_input, force_generic, max_idx, max_val, workspace, workspace_bytes, nullptr, nullptr);
}

extern "C" int c2m_feature_match_index_pre_f32(c2m_stream_t stream, const float* feat_in, const float* feat_ref, int B,
                                               int C, int Hq, int Wq, int Hr, int Wr, int patch, int in_stride,
                                               int ref_stride, int is_norm, int norm_input, int force_generic,
                                               int64_t* max_idx, float* max_val, void* workspace,
                                               size_t workspace_bytes, const float* ss_in_pre, const float* ss_ref_pre) {
  if (!feat_in || !feat_ref || !max_idx || !max_val) return C2M_ERR_INVALID_ARG;
  if (B <= 0 || C <= 0 || patch <= 0 || in_stride <= 0 || ref_stride <= 0) return C2M_ERR_INVALID_ARG;
  if (Hq < patch || Wq < patch || Hr < patch || Wr < patch) return C2M_ERR_INVALID_ARG;
  const CorrWs ws = corr_ws(B, Hq, Wq, Hr, Wr, filter_channels(C));   // (<= what either size query returns)
  if (!workspace || workspace_bytes < ws.total) return C2M_ERR_WORKSPACE;
  hipStream_t st = as_stream(stream);
  char* wsb = static_cast<char*>(workspace);
  float* ss_ref = reinterpret_cast<float*>(wsb + ws.ss_ref);
  float* inv = reinterpret_cast<float*>(wsb + ws.inv);
  float* ss_in = reinterpret_cast<float*>(wsb + ws.ss_in);
  float* qden = reinterpret_cast<float*>(wsb + ws.qden);

  const int Hqp = (Hq - patch) / in_stride + 1, Wqp = (Wq - patch) / in_stride + 1;
  const int Hrp = (Hr - patch) / ref_stride + 1, Wrp = (Wr - patch) / ref_stride + 1;
  int rc;
  if (is_norm) {
    // (ss_*_pre: per-pixel sums of squares the caller already holds -- c2m_feature_normalize_ss_f32 forms them, same chain, while
    // the normalised values are still in registers -- instead of a pass over the map each)
    if (!ss_ref_pre)
      hipLaunchKernelGGL(pixel_sumsq_kernel, dim3(ceil_div(Hr * Wr, 256), B), dim3(256), 0, st, feat_ref, C, Hr * Wr,
                         ss_ref);
    hipLaunchKernelGGL(patch_norm_kernel, dim3(ceil_div(Hrp * Wrp, 256), B), dim3(256), 0, st, ss_ref_pre ? ss_ref_pre : ss_ref, Hr, Wr, patch,
                       ref_stride, Hrp, Wrp, 1, inv);
    if ((rc = check_launch()) != C2M_OK) return rc;
  }
  const bool fast = !force_generic && patch == 3 && in_stride == 1 && ref_stride == 1 &&
                    (C == 64 || C == 128 || C == 256);
  // The pre-filter path (corr_filter.hip): 3/16 of the matrix time, same results.  It needs the ref-patch normalisation
  // (its error bound is relative to |r|) and shapes its 16-bit candidate codes cover.  $C2M_CORR_FILTER=0: exact sweep only.
  static const int filter_env = [] {
    const char* e = getenv("C2M_CORR_FILTER");
    return (e && e[0] == '0') ? 0 : 1;
  }();
  const int filter_on = g_filter_mode < 0 ? filter_env : g_filter_mode;
  const bool use_filter = fast && filter_on && is_norm && c2m::corrf::shapes_ok(B, C, Hq, Wq, Hr, Wr);
  if (norm_input || use_filter) {
    if (!ss_in_pre)
      hipLaunchKernelGGL(pixel_sumsq_kernel, dim3(ceil_div(Hq * Wq, 256), B), dim3(256), 0, st, feat_in, C, Hq * Wq,
                         ss_in);
    hipLaunchKernelGGL(patch_norm_kernel, dim3(ceil_div(Hqp * Wqp, 256), B), dim3(256), 0, st, ss_in_pre ? ss_in_pre : ss_in, Hq, Wq, patch,
                       in_stride, Hqp, Wqp, 0, qden);
    if ((rc = check_launch()) != C2M_OK) return rc;
  }
  const float* invp = is_norm ? inv : nullptr;
  const float* qdp = norm_input ? qden : nullptr;

  if (fast) {
    int* row_eq = reinterpret_cast<int*>(wsb + ws.row_eq);
    int2* skip = reinterpret_cast<int2*>(wsb + ws.skip);
    const c2m::corrf::Ws* fwsp = use_filter ? &ws.f : nullptr;
    static const int dedup = [] {
      const char* e = getenv("C2M_CORR_DEDUP");  // 0: score every ref row (measurement / debugging)
      return (e && e[0] == '0') ? 0 : 1;
    }();
    if (C == 256)
      return launch_corr_mfma<256>(st, feat_in, feat_ref, B, Hq, Wq, Hr, Wr, invp, qdp, row_eq, skip, dedup, max_idx,
                                   max_val, fwsp, wsb, qden);
    if (C == 128)
      return launch_corr_mfma<128>(st, feat_in, feat_ref, B, Hq, Wq, Hr, Wr, invp, qdp, row_eq, skip, dedup, max_idx,
                                   max_val, fwsp, wsb, qden);
    return launch_corr_mfma<64>(st, feat_in, feat_ref, B, Hq, Wq, Hr, Wr, invp, qdp, row_eq, skip, dedup, max_idx,
                                max_val, fwsp, wsb, qden);
  }
  const size_t lds = sizeof(float) * (size_t)patch * patch * C;
  if (lds > 60 * 1024) return C2M_ERR_UNSUPPORTED;
  ProfileScope prof(C2M_KERNEL_CORR_GENERIC, st);
  hipLaunchKernelGGL(corr_argmax_generic_kernel, dim3(Hqp * Wqp, B), dim3(256), lds, st, feat_in, feat_ref, C, Hq, Wq,
                     Hr, Wr, patch, in_stride, ref_stride, Wqp, Hrp, Wrp, invp, qdp, max_idx, max_val);
  return check_launch();
}

extern "C" int c2m_build_pre_offsets_f32(c2m_stream_t stream, const int64_t* max_idx, int B, int h, int w,
                                         float* off3, float* off2, float* off1) {
  if (!max_idx || B <= 0 || h < 3 || w < 3) return C2M_ERR_INVALID_ARG;
  hipStream_t st = as_stream(stream);
  float* outs[3] = {off3, off2, off1};
  const int scales[3] = {1, 2, 4};
  for (int i = 0; i < 3; ++i) {
    if (!outs[i]) continue;
    const int s = scales[i];
    hipLaunchKernelGGL(pre_offset_kernel, dim3(ceil_div(h * s * w * s, 256), 9, B), dim3(256), 0, st, max_idx, h, w,
                       s, reinterpret_cast<float2*>(outs[i]));
  }
  return check_launch();
}
