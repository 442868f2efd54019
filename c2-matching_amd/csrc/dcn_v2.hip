// dcn_v2.hip -- DCNv2 (modulated deformable convolution) forward / backward for gfx950 (MI355X), fp32.
//
// Replaces the reference's CUDA extension (mmsr/models/archs/DCNv2/src/cuda/dcn_v2_cuda.cu:42-172, 206-335 and
// dcn_v2_im2col_cuda.cu:25-327).  The reference materialises the im2col buffer [B][9C][HW] (236-944 MB per sample at
// 160x160 LR), runs batched cuBLAS SGEMMs on it and, in backward, loops over the batch on the host.  Here:
//
//   forward        implicit GEMM out[Co x P] = W[Co x K] . col[K x P] on fp32 MFMA (v_mfma_f32_32x32x2_f32) with the
//                  column matrix generated in registers, already in MFMA B-operand layout.  Sampling coordinates, the four
//                  corner weights and the mask are computed once per (pixel, group, tap) and reused by the group's
//                  channels (the reference recomputes them per channel thread, dcn_v2_im2col_cuda.cu:137-175).  No column
//                  buffer; bias in the epilogue (the reference spends a rank-1 GEMM on it).
//                  dcn_fwd_nhwc_kernel (8/16/32 channels per group, even group count): gathers from a zero-bordered
//                  channels-last copy of the input (float4 per 4 channels, no validity logic), weights DMA'd to LDS in
//                  chunks, two-deep gather pipeline, 8 x 4 pixel patch per wave; optional bf16-MFMA variant.  8-channel
//                  groups gather from a GROUP-MAJOR copy [dg][H+3][W+3][8] instead (Geom::in_grouped; forward and both
//                  channels-last backward kernels): a 32-byte sample run then shares its cache line with the neighbouring
//                  positions of its group instead of with three other groups -- the kernels wait on line look-ups.
//                  dcn_fwd_mfma_kernel: any other geometry, gathers from NCHW, weights from L1/L2.
//   backward data  dCol tile = W^T . gO on MFMA (gO kept in registers as the B operand), consumed in place.
//                  dcn_bwd_offmask_kernel (grad_input not requested): channels-last gathers, Wb tile in LDS, every lane
//                  owns whole (tap, group)s of its pixel -> grad_offset / grad_mask are plain stores.
//                  dcn_bwd_data_kernel (with grad_input): lanes turn their 16 (k, pixel) entries into grad_mask /
//                  grad_offset partial sums and the grad_input scatter (fp32 atomics), dcn_v2_im2col_cuda.cu:56-123, 197-327.
//                  All samples in one launch.
//   backward weight grad_weight = gO . col^T with K = all pixels of the batch: tiles of gO and of the re-generated
//                  columns are staged through LDS (lanes <-> pixels while gathering, lanes <-> rows as MFMA operands),
//                  chunk n+1 fetched under the MFMAs of chunk n, split-K over pixel ranges, fp32 atomics into grad_weight.
//
// K orders: NCHW kernels k = (g * T + tap) * CPG + c_in_group (T = kh*kw, CPG = C/dg); channels-last forward
// (tap, group, kk) with kk = 2t + hi <-> channel t + hi * CPG/2; offset/mask kernel: see weight_relayout_kernel.
// Weights are re-laid out once per call (Wt[k][Co] forward, Wb[o][k] backward, Wh[k/8][Co][8] bf16).
#include "c2m_common.h"

namespace c2m {
namespace dcn {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

struct Geom {
  int B, C, H, W, Co, kh, kw, sh, sw, ph, pw, dh, dw, dg;
  int Ho, Wo, T, CPG, Ktot, KtotPad, CoPad;
  // output of the channels-last forward kernel: planar NCHW (the reference's layout) or channels-last with pitches (in
  // floats) + optional LeakyReLU, for the fused decoder path (the lrelu after every DynAgg, ref_restoration_arch.py:152-154)
  int out_nhwc, out_pix_pitch, out_row_pitch, act;
  long long out_img_pitch;
  float slope;
  // layout of the zero-bordered staging copy the channels-last kernel gathers from: 0 = [H+3][W+3][C]; 1 = group-major
  // [dg][H+3][W+3][C/dg] (real groups).  With 8-channel groups a sample is a 32-byte run: channels-last puts every
  // (pixel, group, corner) on its own 128-byte line, group-major puts the two horizontal corners and the neighbouring
  // pixels' samples of a coherent flow on shared lines (large layer, B=16: 7.6 -> 5.9 ms).
  int in_grouped;
  int* range_flag;   // f16 x 2 GEMM: set to 1 when an output is not finite (a blended sample left |x| < 65520), or nullptr
};

// bilinear sampling state of one (pixel, group, tap)
struct Tap {
  int a1, a2, a3, a4;      // clamped linear addresses of the four corners
  float w1, w2, w3, w4;    // hh*hw, hh*lw, lh*hw, lh*lw   (dcn_v2_im2col_cuda.cu:50)
  float c1, c2, c3, c4;    // 1.0 where the corner contributes, else 0.0 (:36-47 and the in-range test :180)
  float mk;                // modulation mask
  float ah, aw;            // sample position
  int hl, wl;
  bool inside;
};

struct RawTap {
  float oh, ow, mk;  // offset_h, offset_w, mask of one (pixel, group, tap) as stored by the caller
};

__device__ __forceinline__ RawTap load_raw_tap(const Geom& g, const float* __restrict__ off_b,
                                               const float* __restrict__ msk_b, int grp, int tap, int pc) {
  const int HWo = g.Ho * g.Wo;
  const int gt = grp * g.T + tap;
  RawTap r;
  r.oh = off_b[(size_t)(2 * gt) * HWo + pc];
  r.ow = off_b[(size_t)(2 * gt + 1) * HWo + pc];
  r.mk = msk_b[(size_t)gt * HWo + pc];
  return r;
}

__device__ __forceinline__ Tap tap_from_raw_ij(const Geom& g, const RawTap& r, int i, int j, int py, int px, bool pok);

__device__ __forceinline__ Tap tap_from_raw(const Geom& g, const RawTap& r, int tap, int py, int px, bool pok) {
  const int i = tap / g.kw, j = tap - i * g.kw;
  return tap_from_raw_ij(g, r, i, j, py, px, pok);
}

__device__ __forceinline__ Tap tap_from_raw_ij(const Geom& g, const RawTap& r, int i, int j, int py, int px, bool pok) {
  Tap t;
  const float oh = r.oh, ow = r.ow;
  t.mk = pok ? r.mk : 0.0f;
  t.ah = (float)(py * g.sh - g.ph + i * g.dh) + oh;
  t.aw = (float)(px * g.sw - g.pw + j * g.dw) + ow;
  t.inside = pok && t.ah > -1.0f && t.aw > -1.0f && t.ah < (float)g.H && t.aw < (float)g.W;
  const float fh = floorf(t.ah), fw = floorf(t.aw);
  // keep the int conversion defined for wild offsets; such samples are outside anyway
  t.hl = (int)fminf(fmaxf(fh, -2.0f), (float)g.H);
  t.wl = (int)fminf(fmaxf(fw, -2.0f), (float)g.W);
  const int hh_ = t.hl + 1, wh_ = t.wl + 1;
  const float lh = t.ah - (float)t.hl, lw = t.aw - (float)t.wl;
  const float hh = 1.0f - lh, hw = 1.0f - lw;
  t.w1 = hh * hw; t.w2 = hh * lw; t.w3 = lh * hw; t.w4 = lh * lw;
  const bool r0 = t.hl >= 0, r1 = hh_ <= g.H - 1, q0 = t.wl >= 0, q1 = wh_ <= g.W - 1;
  t.c1 = (t.inside && r0 && q0) ? 1.0f : 0.0f;
  t.c2 = (t.inside && r0 && q1) ? 1.0f : 0.0f;
  t.c3 = (t.inside && r1 && q0) ? 1.0f : 0.0f;
  t.c4 = (t.inside && r1 && q1) ? 1.0f : 0.0f;
  const int y0 = min(max(t.hl, 0), g.H - 1), y1 = min(max(hh_, 0), g.H - 1);
  const int x0 = min(max(t.wl, 0), g.W - 1), x1 = min(max(wh_, 0), g.W - 1);
  t.a1 = y0 * g.W + x0; t.a2 = y0 * g.W + x1; t.a3 = y1 * g.W + x0; t.a4 = y1 * g.W + x1;
  return t;
}

__device__ __forceinline__ Tap make_tap(const Geom& g, const float* __restrict__ off_b, const float* __restrict__ msk_b,
                                        int grp, int tap, int py, int px, int pc, bool pok) {
  return tap_from_raw(g, load_raw_tap(g, off_b, msk_b, grp, tap, pc), tap, py, px, pok);
}

// dmcn_im2col_bilinear (:25-54): (w1*v1 + w2*v2 + w3*v3 + w4*v4), corners outside the image read as 0
__device__ __forceinline__ float tap_sample(const Tap& t, const float* __restrict__ im, float& v1, float& v2, float& v3,
                                            float& v4) {
  v1 = t.c1 != 0.0f ? im[t.a1] : 0.0f;
  v2 = t.c2 != 0.0f ? im[t.a2] : 0.0f;
  v3 = t.c3 != 0.0f ? im[t.a3] : 0.0f;
  v4 = t.c4 != 0.0f ? im[t.a4] : 0.0f;
  return (t.w1 * v1 + t.w2 * v2 + t.w3 * v3 + t.w4 * v4);
}

// ---------------------------------------------------------------------------------------------------------------------
// weight re-layout: W[Co][C][T] -> Wt[k][CoPad] (forward A operand rows) and Wb[CoPad2][KtotPad] (backward-data A operand)
// ---------------------------------------------------------------------------------------------------------------------
// nhwc_order != 0 selects the K order of the channels-last forward kernel instead:
//   k = (tap * dg + g) * CPG + kk,  kk = 2t + hi  <->  channel-in-group t + hi * CPG/2
__global__ void __launch_bounds__(256) weight_relayout_kernel(const float* __restrict__ w, Geom g, int CoPad2,
                                                               int nhwc_order, float* __restrict__ wt,
                                                               float* __restrict__ wb) {
  const int e = blockIdx.x * 256 + threadIdx.x;
  const int rows = max(g.CoPad, CoPad2);
  if (e >= rows * g.KtotPad) return;
  const int o = e / g.KtotPad, k = e - o * g.KtotPad;
  float v = 0.0f;
  if (o < g.Co && k < g.Ktot) {
    if (nhwc_order == 2) {
      // backward offset/mask kernel: k tile = 32 rows of one tap; row i of the tile is held by half-wave hi = (i>>2)&1 at
      // accumulator index li = 4*(i>>3) + (i&3), and each lane must own whole (or, for 32-channel groups, half) groups
      const int kt = k >> 5, i = k & 31, tpt = g.C / 32;
      const int tap = kt / tpt, gtile = kt - tap * tpt;
      const int hi = (i >> 2) & 1, li = ((i >> 3) << 2) + (i & 3);
      int grp, ch;
      if (g.CPG == 32) { grp = gtile; ch = 16 * hi + li; }
      else if (g.CPG == 16) { grp = 2 * gtile + hi; ch = li; }
      else { grp = 4 * gtile + 2 * (li >> 3) + hi; ch = li & 7; }
      v = w[((size_t)o * g.C + grp * g.CPG + ch) * g.T + tap];
    } else if (nhwc_order) {
      const int tg = k / g.CPG, kk = k - tg * g.CPG;
      const int tap = tg / g.dg, grp = tg - tap * g.dg;
      const int cig = (kk >> 1) + (kk & 1) * (g.CPG / 2);
      v = w[((size_t)o * g.C + grp * g.CPG + cig) * g.T + tap];
    } else {
      const int gt = k / g.CPG, cig = k - gt * g.CPG;
      const int grp = gt / g.T, tap = gt - grp * g.T;
      v = w[((size_t)o * g.C + grp * g.CPG + cig) * g.T + tap];
    }
  }
  if (wt && o < g.CoPad) wt[(size_t)k * g.CoPad + o] = v;
  if (wb && o < CoPad2) wb[(size_t)o * g.KtotPad + k] = v;
}

// bf16 weights of the channels-last forward kernel: Wh[K/8][CoPad][8], K order (tap, group, 16-k MFMA m, half h, e) with
// channel-in-group = h * CPG/2 + 8m + e  (`g` describes the -- possibly virtual -- grouping the kernel runs with)
__global__ void __launch_bounds__(256) weight_relayout_bf16_kernel(const float* __restrict__ w, Geom g,
                                                                    __bf16* __restrict__ wh) {
  const int e = blockIdx.x * 256 + threadIdx.x;
  if (e >= g.CoPad * g.Ktot) return;
  const int k8 = e / (g.CoPad * 8), r = e - k8 * (g.CoPad * 8);
  const int o = r >> 3, k = k8 * 8 + (r & 7);
  const int gs = k / g.CPG, pos = k - gs * g.CPG;
  const int tap = gs / g.dg, grp = gs - tap * g.dg;
  const int cig = ((pos >> 3) & 1) * (g.CPG / 2) + 8 * (pos >> 4) + (pos & 7);
  wh[e] = (__bf16)(o < g.Co ? w[((size_t)o * g.C + grp * g.CPG + cig) * g.T + tap] : 0.0f);
}

// f16 x 2 weights of the channels-last forward kernel (fp32 result on the f16 matrix pipe, as C2M_CONV_SPLIT_F16X2): S w =
// wA + w1 with the per-tensor power of two S that puts max |w| into [2^14, 2^15); Wh[K/8][2 pieces][CoPad][8] f16 in the K
// order of weight_relayout_bf16_kernel, then one float 1/S at element K*CoPad (in floats) of the image.
__global__ void __launch_bounds__(1024) weight_absmax_kernel(const float* __restrict__ w, int n, float* __restrict__ sinv_out) {
  __shared__ unsigned red[1024];
  unsigned m = 0;
  for (int i = threadIdx.x; i < n; i += 1024) m = max(m, __float_as_uint(w[i]) & 0x7fffffffu);
  red[threadIdx.x] = m;
  __syncthreads();
  for (int s = 512; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) red[threadIdx.x] = max(red[threadIdx.x], red[threadIdx.x + s]);
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    const float wmax = __uint_as_float(red[0]);
    float S = 1.0f;
    if (wmax > 0.0f && wmax <= 3.0e38f) {
      int e;
      (void)frexpf(wmax, &e);
      e = 15 - e;
      e = e < -100 ? -100 : (e > 100 ? 100 : e);
      S = ldexpf(1.0f, e);
    }
    *sinv_out = 1.0f / S;   // (power of two: exact)
  }
}
__global__ void __launch_bounds__(256) weight_relayout_f16x2_kernel(const float* __restrict__ w, Geom g, const float* __restrict__ sinv,
                                                                     _Float16* __restrict__ wh) {
  const int e = blockIdx.x * 256 + threadIdx.x;
  if (e >= g.CoPad * g.Ktot) return;
  const int k8 = e / (g.CoPad * 8), r = e - k8 * (g.CoPad * 8);
  const int o = r >> 3, k = k8 * 8 + (r & 7);
  const int gs = k / g.CPG, pos = k - gs * g.CPG;
  const int tap = gs / g.dg, grp = gs - tap * g.dg;
  const int cig = ((pos >> 3) & 1) * (g.CPG / 2) + 8 * (pos >> 4) + (pos & 7);
  const float S = 1.0f / *sinv;
  const float v = (o < g.Co ? w[((size_t)o * g.C + grp * g.CPG + cig) * g.T + tap] : 0.0f) * S;   // exact (power of two)
  const _Float16 a = (_Float16)v;
  const size_t base = ((size_t)k8 * 2 * g.CoPad + o) * 8 + (r & 7);
  wh[base] = a;
  wh[base + (size_t)g.CoPad * 8] = (_Float16)(v - (float)a);
}

// ---------------------------------------------------------------------------------------------------------------------
// forward: one wave = NT tiles of 32 output pixels x MT tiles of 32 output channels, K swept group by group, tap by tap
// ---------------------------------------------------------------------------------------------------------------------
template <int MT, int NT>
__global__ void __launch_bounds__(256, 2) dcn_fwd_mfma_kernel(const float* __restrict__ in, const float* __restrict__ wt,
                                                            const float* __restrict__ bias,
                                                            const float* __restrict__ offset,
                                                            const float* __restrict__ mask, Geom g,
                                                            float* __restrict__ out) {
  const int tid = threadIdx.x, l = tid & 63, hi = l >> 5, j = l & 31;
  const int wv = tid >> 6;
  const int b = blockIdx.y, ob = blockIdx.z;  // ob: block of MT*32 output channels
  const int HW = g.H * g.W, HWo = g.Ho * g.Wo;
  const int p0 = (blockIdx.x * 4 + wv) * (NT * 32);
  if (p0 >= HWo) return;  // whole wave out of range (no barriers in this kernel)
  const float* in_b = in + (size_t)b * g.C * HW;
  const float* off_b = offset + (size_t)b * g.dg * 2 * g.T * HWo;
  const float* msk_b = mask + (size_t)b * g.dg * g.T * HWo;
  const float* wt_o = wt + ob * (MT * 32) + j;

  int py[NT], px[NT], pc[NT];
  bool pok[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    const int p = p0 + nt * 32 + j;
    pok[nt] = p < HWo;
    pc[nt] = min(p, HWo - 1);
    py[nt] = pc[nt] / g.Wo;
    px[nt] = pc[nt] - py[nt] * g.Wo;
  }

  f32x16 acc[MT][NT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mt][nt][r] = 0.0f;

  for (int grp = 0; grp < g.dg; ++grp) {
    for (int tap = 0; tap < g.T; ++tap) {
      Tap tp[NT];
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) tp[nt] = make_tap(g, off_b, msk_b, grp, tap, py[nt], px[nt], pc[nt], pok[nt]);
      const int kbase = (grp * g.T + tap) * g.CPG;
      const float* im0 = in_b + (size_t)(grp * g.CPG + hi) * HW;
      const float* wrow = wt_o + (size_t)(kbase + hi) * g.CoPad;
      for (int kp = 0; kp < g.CPG / 2; ++kp) {
        const float* im = im0 + (size_t)(2 * kp) * HW;
        float bv[NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
          float v1, v2, v3, v4;
          bv[nt] = tap_sample(tp[nt], im, v1, v2, v3, v4) * tp[nt].mk;  // col = val * mask (:189)
        }
        const float* wr = wrow + (size_t)(2 * kp) * g.CoPad;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
          const float a = wr[mt * 32];
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bv[nt], acc[mt][nt], 0, 0, 0);
        }
      }
    }
  }

  // epilogue: out[b][o][p] = acc + bias[o];  D row i = (r&3) + 8*(r>>2) + 4*hi, column j
  float* out_b = out + (size_t)b * g.Co * HWo;
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int o = ob * (MT * 32) + mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
      if (o < g.Co) {
        const float bo = bias[o];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
          if (pok[nt]) out_b[(size_t)o * HWo + p0 + nt * 32 + j] = acc[mt][nt][r] + bo;
      }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// NCHW -> NHWC staging copy (one pass, LDS-tiled transpose).  Patch-match pre-offsets are long range and, on synthetic
// or textured data, incoherent between neighbouring pixels: in NCHW every (corner, channel) of a gather is its own
// cache line; channels-last makes one corner = one contiguous C/dg-vector and lets all groups share the pixel's line.
// ---------------------------------------------------------------------------------------------------------------------
// The copy carries a zero border: 1 pixel on the top/left, 2 on the bottom/right ((H+3) x (W+3) pixels).  A sample
// position clamped to [-1, H] x [-1, W] then always has its four corners inside the buffer and every corner the reference
// treats as "outside" (dcn_v2_im2col_cuda.cu:36-47, 180) reads 0 -- no per-corner validity logic in the hot loop.
template <typename OutT>   // float, or __bf16 for the bf16-MFMA forward (halves the gather bytes)
__global__ void __launch_bounds__(256) nchw_to_nhwc_kernel(const float* __restrict__ in, int C, int H, int W,
                                                            OutT* __restrict__ out) {
  __shared__ float tile[32][33];
  const int Wp = W + 3, PP = (H + 3) * Wp, HW = H * W;
  const int b = blockIdx.z, c0 = blockIdx.y * 32, p0 = blockIdx.x * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
  const float* ib = in + (size_t)b * C * HW;
  OutT* ob = out + (size_t)b * C * PP;
  const int pp = p0 + tx;
  const int yy = pp / Wp - 1, xx = pp - (yy + 1) * Wp - 1;
  const bool inside = pp < PP && yy >= 0 && yy < H && xx >= 0 && xx < W;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int c = c0 + ty + 8 * r;
    tile[ty + 8 * r][tx] = (c < C && inside) ? ib[(size_t)c * HW + yy * W + xx] : 0.0f;
  }
  __syncthreads();
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int p = p0 + ty + 8 * r, c = c0 + tx;
    if (c < C && p < PP) ob[(size_t)p * C + c] = (OutT)tile[tx][ty + 8 * r];
  }
}

// Same copy with 64-channel x 64-pixel tiles for C % 64 == 0: 256-byte segments on both the NCHW read and the NHWC write
// side (the 32x32 tile writes 128-byte pieces: 0.87 ms instead of 0.3 ms for the 64-channel 640x640 layer at B=16).
template <typename OutT>
__global__ void __launch_bounds__(256) nchw_to_nhwc64_kernel(const float* __restrict__ in, int C, int H, int W,
                                                              OutT* __restrict__ out) {
  __shared__ float tile[64][65];
  const int Wp = W + 3, PP = (H + 3) * Wp, HW = H * W;
  const int b = blockIdx.z, c0 = blockIdx.y * 64, p0 = blockIdx.x * 64;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;  // 64 x 4
  const float* ib = in + (size_t)b * C * HW;
  OutT* ob = out + (size_t)b * C * PP;
  const int pp = p0 + tx;
  const int yy = pp / Wp - 1, xx = pp - (yy + 1) * Wp - 1;
  const bool inside = pp < PP && yy >= 0 && yy < H && xx >= 0 && xx < W;
  const int src = yy * W + xx;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int c = c0 + ty + 4 * r;
    tile[ty + 4 * r][tx] = inside ? ib[(size_t)c * HW + src] : 0.0f;
  }
  __syncthreads();
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int p = p0 + ty + 4 * r;
    if (p < PP) ob[(size_t)p * C + c0 + tx] = (OutT)tile[tx][ty + 4 * r];
  }
}

// Group-major variant of the bordered copy: out[b][c / G][(H+3)*(W+3)][G] (Geom::in_grouped).  One thread per (group, bordered
// pixel): G plane reads (coalesced across the wave), one contiguous G-element record written.
template <typename OutT, int G>
__global__ void __launch_bounds__(256) nchw_to_grouped_kernel(const float* __restrict__ in, int C, int H, int W,
                                                               OutT* __restrict__ out) {
  const int Wp = W + 3, PP = (H + 3) * Wp, HW = H * W;
  const int pp = blockIdx.x * 256 + threadIdx.x, grp = blockIdx.y, b = blockIdx.z;
  if (pp >= PP) return;
  const int yy = pp / Wp - 1, xx = pp - (yy + 1) * Wp - 1;
  const bool inside = yy >= 0 && yy < H && xx >= 0 && xx < W;
  const float* ib = in + ((size_t)b * C + (size_t)grp * G) * HW + (inside ? yy * W + xx : 0);
  OutT v[G];
#pragma unroll
  for (int c = 0; c < G; ++c) v[c] = (OutT)(inside ? ib[(size_t)c * HW] : 0.0f);
  OutT* o = out + (((size_t)b * (C / G) + grp) * PP + pp) * G;
#pragma unroll
  for (int c = 0; c < G; ++c) o[c] = v[c];
}

// ---------------------------------------------------------------------------------------------------------------------
// forward, channels-last gathers + LDS-staged weights.  Wave tiling as dcn_fwd_mfma_kernel, K order (tap, group, kk).
//   * lane (hi, j) owns the contiguous half-run of CPG/2 channels [hi*CPG/2, (hi+1)*CPG/2) of its group: one float4
//     gather per 4 channels and corner; all taps/groups of a pixel hit the same few cache lines;
//   * the weight rows of GC consecutive groups (a "chunk", <= 32 KiB) are DMA'd global->LDS once per workgroup and shared
//     by its 4 waves as MFMA A operands (double buffered, one barrier per chunk) instead of 1 global load per MFMA;
//   * the raw offsets/mask of the NEXT (tap, group) are fetched one iteration ahead, so a group's dependent chain is
//     gather -> bilinear -> MFMA only.
// ---------------------------------------------------------------------------------------------------------------------
//   * SPLITG: CPG is a VIRTUAL group of two real deformable groups (g.CPG, g.dg describe the virtual grouping): half-wave
//     hi samples real group 2*grp + hi with that group's offsets/mask and owns all of its channels.  The memory layout,
//     the K order and the weight chunks are exactly those of a real CPG-channel group; the sampling state -- the
//     dominant VALU cost for 8- and 16-channel groups -- is computed once per 2x as many MFMAs.
//   * BF16: the GEMM runs on v_mfma_f32_32x32x16_bf16 (fp32 accumulate): `wt` then points to bf16 weights laid out
//     [K/8][CoPad][8] (weight_relayout_bf16_kernel), the blended column values are rounded to bf16 (RNE) in registers.
//     Gathers, sampling state and blend stay fp32.  For callers that ask for reduced precision (bf16 autocast).
//   * F16X2: the GEMM runs on v_mfma_f32_32x32x16_f16 with THREE products per k step (the arithmetic of the convolutions'
//     C2M_CONV_SPLIT_F16X2): the blended fp32 column value c = x0 + 2^-11 x1' (two round-to-nearest f16 pieces), the scaled
//     weight S w = wA + w1 (weight_relayout_f16x2_kernel); S w.c = wA.x0 + w1.x0 + (2^-11 wA).x1', one accumulator, times 1/S
//     in the epilogue.  3/32 of the fp32 pipe's matrix time; gathers, sampling state and blend as the fp32 kernel.  Domain
//     |c| < 65520 -- beyond it the output is not finite, which the epilogue reports into Geom::range_flag.
#ifndef C2M_DCN_F16_OCC
#define C2M_DCN_F16_OCC 3
#endif
#ifndef C2M_DCN_F16_MT8
#define C2M_DCN_F16_MT8 1   // 256 output channels share one gathered column (one wave per SIMD), as the bf16 GEMM does: the
#endif                      // f16 x 2 kernel is gather bound (small layer, B=16: 4.27 -> 2.70 ms)
#ifndef C2M_DCN_F16_OCC4
#define C2M_DCN_F16_OCC4 2  // waves per SIMD asked of the MT = 4 f16 x 2 kernels
#endif
template <int MT, int NT, int CPG, int GC, bool SPLITG, bool BF16, bool F16X2 = false>
__global__ void __launch_bounds__(256, ((BF16 || F16X2) && MT == 8) ? 1 : ((SPLITG && !BF16 && MT <= 2) ? (F16X2 ? C2M_DCN_F16_OCC : 4) : ((F16X2 && MT == 4) ? C2M_DCN_F16_OCC4 : 2))) dcn_fwd_nhwc_kernel(const float* __restrict__ inl, const float* __restrict__ wt,
                                                               const float* __restrict__ bias,
                                                               const float* __restrict__ offset,
                                                               const float* __restrict__ mask, Geom g,
                                                               float* __restrict__ out) {
  constexpr int HALF = CPG / 2;
  constexpr int ES = BF16 ? 2 : 4;            // bytes per element of the staged channels-last copy (bf16 with the bf16 GEMM)
  constexpr int EPV = 16 / ES;                // elements per 16-byte gather
  constexpr int NQ = HALF / EPV;              // gathers per corner
  constexpr int MW = MT * 32;                 // output channels of this workgroup
  constexpr int CHUNK = GC * CPG * MW / (BF16 ? 2 : 1);   // floats (4-byte units) per weight chunk
  static_assert(!(BF16 || F16X2) || HALF % 8 == 0, "a 16-bit MFMA takes 8 channels from each half-wave");
  static_assert(!(BF16 && F16X2), "one GEMM arithmetic");
  extern __shared__ __attribute__((aligned(16))) float wl[];  // fp32: [2][GC*CPG][MW]; bf16: [2][GC*CPG/8][MW][8]
  const int tid = threadIdx.x, l = tid & 63, hi = l >> 5, j = l & 31;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int b = blockIdx.y, ob = blockIdx.z;
  const int HWo = g.Ho * g.Wo;
  const int p0 = (blockIdx.x * 4 + wv) * (NT * 32);  // may lie beyond HWo for the last waves: they still hit the barriers
  const int Wp = g.W + 3;                                          // zero-bordered staging copy (nchw_to_nhwc_kernel)
  const char* in_b = reinterpret_cast<const char*>(inl) + (size_t)b * g.C * (g.H + 3) * Wp * ES;
  const int dg_real = SPLITG ? 2 * g.dg : g.dg;
  const float* off_b = offset + (size_t)b * dg_real * 2 * g.T * HWo;
  const float* msk_b = mask + (size_t)b * dg_real * g.T * HWo;
  // Everything this sample reads goes through three buffer resources (32-bit byte offsets: a per-lane VGPR part, a
  // wave-uniform SGPR part and an instruction immediate) -- no 64-bit address arithmetic on the vector ALU, which this
  // kernel is bound by.  use_nhwc() keeps each of the three ranges below 2 GiB.
  const __amdgpu_buffer_rsrc_t in_rs = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<char*>(in_b), 0, (int)((unsigned)g.C * (unsigned)(g.H + 3) * (unsigned)Wp * ES), 0x00020000);
  const __amdgpu_buffer_rsrc_t off_rs = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(off_b), 0, (int)((unsigned)dg_real * 2u * (unsigned)g.T * (unsigned)HWo * 4u), 0x00020000);
  const __amdgpu_buffer_rsrc_t msk_rs = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(msk_b), 0, (int)((unsigned)dg_real * (unsigned)g.T * (unsigned)HWo * 4u), 0x00020000);

  int pc[NT];
  float fy[NT], fx[NT];   // sample position of tap (0, 0) without offset
  bool pok[NT];
  // A wave's 32 pixels form an 8 x 4 patch when the output size allows it (else 32 consecutive pixels of a row): with
  // coherent flows the bilinear corners of vertically adjacent pixels share cache lines too (5 x 9 instead of 2 x 33 texels).
  const bool tile2d = (g.Wo % 8 == 0) && (g.Ho % 4 == 0);
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    int p = p0 + nt * 32 + j;
    pok[nt] = p < HWo;
    if (tile2d) {
      const int tile = min(p0 / 32 + nt, HWo / 32 - 1), tpr = g.Wo / 8;
      const int ty = tile / tpr, tx = tile - ty * tpr;
      p = (ty * 4 + (j >> 3)) * g.Wo + tx * 8 + (j & 7);
    }
    pc[nt] = min(p, HWo - 1);   // lanes past the end recompute the last pixel and simply do not store
    const int py = pc[nt] / g.Wo, px = pc[nt] - py * g.Wo;
    fy[nt] = (float)(py * g.sh - g.ph);
    fx[nt] = (float)(px * g.sw - g.pw);
  }

  f32x16 acc[MT][NT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mt][nt][r] = 0.0f;

  // chunk ci = weight rows [ci*GC*CPG, (ci+1)*GC*CPG) of Wt[K][CoPad], columns [ob*MW, ob*MW + MW)
  // (bf16: rows are k octets, one 16-byte piece per (octet, output channel))
  auto stage = [&](int ci, int buf) {   // 16-byte pieces, 64 per DMA instruction (wave-uniform LDS base + lane*16)
    constexpr int NPIECE = CHUNK / 4;
    constexpr int PPR = (BF16 || F16X2) ? MW : MW / 4;            // pieces per row
    constexpr int ROWS = BF16 ? GC * CPG / 8 : F16X2 ? 2 * GC * CPG / 8 : GC * CPG;   // (f16 x 2: a row per (k octet, piece))
    for (int pb = wv * 64; pb < NPIECE; pb += 256) {
      const int piece = pb + l;
      if (piece < NPIECE) {
        const int row = piece / PPR, pc_ = piece - row * PPR;
        const float* src = wt + ((size_t)(ci * ROWS + row) * g.CoPad + ob * MW) * ((BF16 || F16X2) ? 4 : 1) + pc_ * 4;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                         (__attribute__((address_space(3))) void*)(wl + buf * CHUNK + pb * 4), 16, 0, 0);
      }
    }
  };

  // Flat sequence of (tap, group) steps gs = tap*dg + grp, software-pipelined two deep:
  //   raw offsets/mask of step gs+2 are loaded, the sampling state + gathers of step gs+1 are issued, then the
  //   bilinear blend + MFMAs of step gs run on data requested one step earlier.
  const int nstep = g.T * g.dg;
  // raw offsets / mask of (tap, group): plane gt = group*T + tap of this sample, element pc.  With SPLITG half-wave hi
  // reads real group 2*grp + hi: that part of the address is per-lane but constant, so it lives in the VGPR offset.
  int rv_off[NT], rv_msk[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    rv_off[nt] = (pc[nt] + (SPLITG ? hi * 2 * g.T * HWo : 0)) * 4;
    rv_msk[nt] = (pc[nt] + (SPLITG ? hi * g.T * HWo : 0)) * 4;
  }
  auto raw_at = [&](int tap, int grp, RawTap (&r)[NT]) {
    const int gt = (SPLITG ? 2 * grp : grp) * g.T + tap;   // wave-uniform
    const int so = 2 * gt * HWo * 4;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      r[nt].oh = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(off_rs, rv_off[nt], so, 0));
      r[nt].ow = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(off_rs, rv_off[nt], so + HWo * 4, 0));
      r[nt].mk = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(msk_rs, rv_msk[nt], gt * HWo * 4, 0));
    }
  };
  struct Gath { f32x4 v1[NQ], v2[NQ], v3[NQ], v4[NQ]; };
  // Sampling state of one (pixel, group, tap): element offset of the top-left corner in the bordered copy and the four
  // bilinear weights with the modulation mask (:189) folded in.  Clamping the position to [-1, H] x [-1, W] puts every
  // "outside" corner (:36-47, :180) on the zero border, so no validity flags exist.
  struct Samp { unsigned o1; float w1, w2, w3, w4; };
  struct Wts { float w1, w2, w3, w4; };
  const float Hf = (float)g.H, Wf = (float)g.W;
  constexpr int CPGR = SPLITG ? CPG / 2 : CPG;         // channels of a real group
  const int plane = (g.H + 3) * Wp * CPGR * ES;        // group-major: bytes of one group's plane
  const int cpix = (g.in_grouped ? CPGR : g.C) * ES, crow = Wp * cpix;   // bytes to the right / lower corner
  // per-lane constant part of a gather offset: bordered pixel (1, 1) = image pixel (0, 0), plus this lane's half-run
  // (channels-last: inside the pixel's C channels; group-major: inside its real group's plane)
  const int lane_o = (Wp + 1) * cpix + (g.in_grouped ? (SPLITG ? hi * plane : hi * HALF * ES) : hi * HALF * ES);
  auto gather = [&](int grp, const Samp (&sp)[NT], Gath (&gv)[NT]) {
    // the four corners share the lane's byte offset sp.o1; group and corner displacement are wave-uniform (SGPR offset)
    const int s1 = g.in_grouped ? (SPLITG ? 2 * grp : grp) * plane : grp * CPG * ES, s2 = s1 + cpix, s3 = s1 + crow,
              s4 = s3 + cpix;
    auto ld = [&](int voff, int so, int imm) __attribute__((always_inline)) {
      return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(in_rs, voff + imm, so, 0));
    };
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      const int o1 = (int)sp[nt].o1;
      // corner-major issue order: the 16-byte pieces of one corner's channel run share a cache line and should reach the
      // texture path back to back (random flows: 15 % slower otherwise)
#pragma unroll
      for (int q = 0; q < NQ; ++q) gv[nt].v1[q] = ld(o1, s1, q * 16);
#pragma unroll
      for (int q = 0; q < NQ; ++q) gv[nt].v2[q] = ld(o1, s2, q * 16);
#pragma unroll
      for (int q = 0; q < NQ; ++q) gv[nt].v3[q] = ld(o1, s3, q * 16);
#pragma unroll
      for (int q = 0; q < NQ; ++q) gv[nt].v4[q] = ld(o1, s4, q * 16);
    }
  };

  // (tap, group) of steps gs+1 and gs+2 are tracked with wave-uniform counters (no divisions, no branches: the step loop
  // below must stay one basic block so that the compiler can wait with exact vmcnt counts).  The counters saturate at
  // the last (tap, group): the two look-ahead stages past the end re-sample it (valid addresses, results unused).
  struct Pos { int tap, grp, ti, tj; };
  auto advance = [&](Pos& p) {
    const bool last = (p.tap == g.T - 1) & (p.grp == g.dg - 1);
    const bool wrap = p.grp == g.dg - 1;           // next step starts a new tap
    const bool roww = wrap & (p.tj == g.kw - 1);   // ... in a new kernel row
    Pos n;
    n.grp = wrap ? 0 : p.grp + 1;
    n.tap = wrap ? p.tap + 1 : p.tap;
    n.tj = roww ? 0 : (wrap ? p.tj + 1 : p.tj);
    n.ti = roww ? p.ti + 1 : p.ti;
    p.grp = last ? p.grp : n.grp;
    p.tap = last ? p.tap : n.tap;
    p.tj = last ? p.tj : n.tj;
    p.ti = last ? p.ti : n.ti;
  };
  auto state = [&](const RawTap (&r)[NT], const Pos& p, Samp (&sp)[NT]) {
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      const float ah = __builtin_amdgcn_fmed3f((fy[nt] + (float)(p.ti * g.dh)) + r[nt].oh, -1.0f, Hf);
      const float aw = __builtin_amdgcn_fmed3f((fx[nt] + (float)(p.tj * g.dw)) + r[nt].ow, -1.0f, Wf);
      const float fh = floorf(ah), fw = floorf(aw);
      const float lh = ah - fh, lw = aw - fw;
      const float hw = 1.0f - lw;
      const float mh = (1.0f - lh) * r[nt].mk, ml = lh * r[nt].mk;
      sp[nt].w1 = mh * hw; sp[nt].w2 = mh * lw; sp[nt].w3 = ml * hw; sp[nt].w4 = ml * lw;
      sp[nt].o1 = (unsigned)(((int)fh * Wp + (int)fw) * cpix + lane_o);
    }
  };
  // One pipeline step: issue (state + gathers) of step gs+1 into (wB, gvB) and the raw loads of step gs+2, then blend +
  // MFMAs of step gs from (wA, gvA), whose gathers were issued one step earlier.  Called alternately with the two
  // register sets swapped, so nothing is ever copied.  Branch-free on purpose (see Pos).
  RawTap raw[NT];
  Pos p1{0, 0, 0, 0}, p2{0, 0, 0, 0};   // positions of steps gs+1 and gs+2
  auto step = [&](int ci, int gi, Wts (&wA)[NT], Gath (&gvA)[NT], Wts (&wB)[NT], Gath (&gvB)[NT]) __attribute__((always_inline)) {
    {
      Samp sp[NT];
      state(raw, p1, sp);
      gather(p1.grp, sp, gvB);
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) wB[nt] = Wts{sp[nt].w1, sp[nt].w2, sp[nt].w3, sp[nt].w4};
      raw_at(p2.tap, p2.grp, raw);
      advance(p1);
      advance(p2);
    }
    // keep the scheduler from sinking the look-ahead loads below the MFMAs they are meant to overlap with
    __builtin_amdgcn_sched_barrier(0);
    // blend step gs: w1*v1 + w2*v2 + w3*v3 + w4*v4 with the mask already inside the weights (fma chain; differs from the
    // oracle's (w1*v1 + ... ) * mask by rounding only -- DCNv2 parity is tolerance-based).  Software pipeline over the
    // k-pairs: the LDS read of the A operands and the blend of the B operand of k-pair t+1 are issued in front of the
    // MFMAs of k-pair t.
    // channel t of a gathered corner: fp32 copy -> element t of the vectors; bf16 copy -> half of a dword, widened
    auto elem = [](const f32x4 (&v)[NQ], int t) __attribute__((always_inline)) {
      if constexpr (BF16) {
        const float pair = v[t >> 3][(t >> 1) & 3];   // (bit_cast straight from the vector element reads element 0)
        const unsigned wd = __builtin_bit_cast(unsigned, pair);
        return __builtin_bit_cast(float, (t & 1) ? (wd & 0xffff0000u) : (wd << 16));
      } else {
        return v[t >> 2][t & 3];
      }
    };
    auto blend = [&](int t, float (&c)[NT]) __attribute__((always_inline)) {
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
        c[nt] = fmaf(wA[nt].w4, elem(gvA[nt].v4, t), fmaf(wA[nt].w3, elem(gvA[nt].v3, t),
                fmaf(wA[nt].w2, elem(gvA[nt].v2, t), wA[nt].w1 * elem(gvA[nt].v1, t))));
    };
    if constexpr (BF16) {
      // one MFMA = 16 k: channels [8m, 8m+8) of this lane's half-run from each half-wave.  A operand of (m, mt): the 8 bf16
      // of k octet (gi*CPG/8 + 2m + hi), output channel mt*32 + j -- one conflict-free ds_read_b128.
      const bf16x8* wrow = reinterpret_cast<const bf16x8*>(wl + (ci & 1) * CHUNK) + (gi * (CPG / 8) + hi) * MW + j;
#pragma unroll
      for (int m = 0; m < HALF / 8; ++m) {
        bf16x8 vb[NT];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          float c[NT];
          blend(8 * m + e, c);
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) vb[nt][e] = (__bf16)c[nt];
        }
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
          const bf16x8 a = wrow[(2 * m) * MW + mt * 32];
#pragma unroll
          for (int nt = 0; nt < NT; ++nt)
            acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, vb[nt], acc[mt][nt], 0, 0, 0);
        }
      }
    } else if constexpr (F16X2) {
      typedef float f32x2 __attribute__((ext_vector_type(2)));
      typedef _Float16 f16x2v __attribute__((ext_vector_type(2)));
      typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
      // A operands of (m, mt): pieces wA / w1 of k octet (gi*CPG/8 + 2m + hi), output channel mt*32 + j
      const f16x8* wrow = reinterpret_cast<const f16x8*>(wl + (ci & 1) * CHUNK) + (size_t)((gi * (CPG / 8) + hi) * 2) * MW + j;
#pragma unroll
      for (int m = 0; m < HALF / 8; ++m) {
        f16x8 b0[NT], b1[NT];
#pragma unroll
        for (int u = 0; u < 4; ++u) {   // channel pairs (8m + 2u, 8m + 2u + 1): packed blend as the fp32 kernel, then the split
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) {
            const int ch = 8 * m + 2 * u;
            const f32x4 q1 = gvA[nt].v1[ch >> 2], q2 = gvA[nt].v2[ch >> 2], q3 = gvA[nt].v3[ch >> 2], q4 = gvA[nt].v4[ch >> 2];
            const int e = ch & 3;
            const f32x2 a1 = {q1[e], q1[e + 1]}, a2 = {q2[e], q2[e + 1]}, a3 = {q3[e], q3[e + 1]}, a4 = {q4[e], q4[e + 1]};
            const f32x2 k1 = {wA[nt].w1, wA[nt].w1}, k2 = {wA[nt].w2, wA[nt].w2}, k3 = {wA[nt].w3, wA[nt].w3},
                        k4 = {wA[nt].w4, wA[nt].w4};
            const f32x2 c = __builtin_elementwise_fma(k4, a4, __builtin_elementwise_fma(k3, a3, __builtin_elementwise_fma(k2, a2, k1 * a1)));
            const f16x2v h0 = __builtin_convertvector(c, f16x2v);
            const f32x2 r = (c - __builtin_convertvector(h0, f32x2)) * 2048.0f;
            const f16x2v h1 = __builtin_convertvector(r, f16x2v);
            b0[nt][2 * u] = h0[0]; b0[nt][2 * u + 1] = h0[1];
            b1[nt][2 * u] = h1[0]; b1[nt][2 * u + 1] = h1[1];
          }
        }
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
          const f16x8 a0 = wrow[(size_t)(2 * m) * 2 * MW + mt * 32], a1 = wrow[(size_t)((2 * m) * 2 + 1) * MW + mt * 32];
          const f16x8 ad = a0 * (_Float16)(1.0f / 2048.0f);   // (exact unless subnormal: weights below 2^-13 max |w|, lost bits ~2^-39 of the largest product)
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) {
            acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, b0[nt], acc[mt][nt], 0, 0, 0);
            acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, b0[nt], acc[mt][nt], 0, 0, 0);
            acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ad, b1[nt], acc[mt][nt], 0, 0, 0);
          }
        }
      }
    } else {
    // MFMAs: A[i = o][kk] from the staged chunk, row (2t + hi) of group gi, column mt*32 + j
    const float* wrow = wl + (ci & 1) * CHUNK + j + (gi * CPG + hi) * MW;
    // The blend runs on channel PAIRS (v_pk_mul_f32 / v_pk_fma_f32: two fp32 lanes per instruction, 4 instructions per
    // pair instead of 8): pair u = channels (2u, 2u+1) of the lane's half-run, i.e. the B operands of k-pairs 2u, 2u+1.
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    auto blend2 = [&](int u, f32x2 (&c)[NT]) __attribute__((always_inline)) {
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        const f32x4 q1 = gvA[nt].v1[u >> 1], q2 = gvA[nt].v2[u >> 1], q3 = gvA[nt].v3[u >> 1], q4 = gvA[nt].v4[u >> 1];
        const int e = (u & 1) * 2;
        const f32x2 a1 = {q1[e], q1[e + 1]}, a2 = {q2[e], q2[e + 1]}, a3 = {q3[e], q3[e + 1]}, a4 = {q4[e], q4[e + 1]};
        const f32x2 k1 = {wA[nt].w1, wA[nt].w1}, k2 = {wA[nt].w2, wA[nt].w2}, k3 = {wA[nt].w3, wA[nt].w3},
                    k4 = {wA[nt].w4, wA[nt].w4};
        c[nt] = __builtin_elementwise_fma(k4, a4, __builtin_elementwise_fma(k3, a3, __builtin_elementwise_fma(k2, a2, k1 * a1)));
      }
    };
    static_assert(HALF % 2 == 0, "channel pairs");
    float aop[2][2][MT];
    f32x2 col[2][NT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) { aop[0][0][mt] = wrow[mt * 32]; aop[0][1][mt] = wrow[2 * MW + mt * 32]; }
    blend2(0, col[0]);
#pragma unroll
    for (int u = 0; u < HALF / 2; ++u) {
      if (u + 1 < HALF / 2) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
          aop[(u + 1) & 1][0][mt] = wrow[(2 * (2 * u + 2)) * MW + mt * 32];
          aop[(u + 1) & 1][1][mt] = wrow[(2 * (2 * u + 3)) * MW + mt * 32];
        }
        blend2(u + 1, col[(u + 1) & 1]);
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
          for (int nt = 0; nt < NT; ++nt)
            acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(aop[u & 1][h][mt], col[u & 1][nt][h], acc[mt][nt], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
    }
    __builtin_amdgcn_sched_barrier(0);
  };

  Wts w0[NT], w1s[NT];
  Gath gv0[NT], gv1[NT];
  {
    Pos pz{0, 0, 0, 0};
    Samp sp[NT];
    raw_at(0, 0, raw);
    state(raw, pz, sp);
    gather(0, sp, gv0);
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) w0[nt] = Wts{sp[nt].w1, sp[nt].w2, sp[nt].w3, sp[nt].w4};
    advance(p1);                 // step 1
    advance(p2); advance(p2);    // step 2
    raw_at(p1.tap, p1.grp, raw);
  }
  static_assert(GC % 2 == 0, "the step pairs (two register sets) must not straddle a weight chunk");
  const int nchunk = nstep / GC;   // GC divides dg
  stage(0, 0);
  for (int ci = 0; ci < nchunk; ++ci) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();  // chunk ci landed in wl[ci&1]; every wave is done reading wl[(ci+1)&1]
    stage(min(ci + 1, nchunk - 1), (ci + 1) & 1);   // (the last iteration re-stages its own chunk into the idle buffer)
#pragma unroll
    for (int gi = 0; gi < GC; gi += 2) {
      step(ci, gi, w0, gv0, w1s, gv1);
      step(ci, gi + 1, w1s, gv1, w0, gv0);
    }
  }

  if constexpr (F16X2) {
    // times 1/S (a power of two: exact); an output that is not finite means a blended sample left the f16 x 2 domain
    const float sinv = wt[(size_t)g.Ktot * g.CoPad];
    bool bad = false;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          acc[mt][nt][r] *= sinv;
          bad |= !(fabsf(acc[mt][nt][r]) <= 3.0e38f);
        }
    if (bad && g.range_flag != nullptr) *g.range_flag = 1;   // (rare, idempotent)
  }
  if (g.out_nhwc) {
    // channels-last store (+ activation): the lane's rows (r & 3) are 4 consecutive output channels = one float4
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      const int py = pc[nt] / g.Wo, px = pc[nt] - py * g.Wo;
      float* op = out + (size_t)b * g.out_img_pitch + (size_t)py * g.out_row_pitch + (size_t)px * g.out_pix_pitch;
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int qd = 0; qd < 4; ++qd) {
          const int o = ob * MW + mt * 32 + 8 * qd + 4 * hi;
          if (o < g.Co && pok[nt]) {   // Co % 4 == 0 on this path (checked by the launcher)
            f32x4 v;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              v[e] = acc[mt][nt][4 * qd + e] + bias[o + e];
              if (g.act == 2) v[e] = v[e] > 0.0f ? v[e] : v[e] * g.slope;
              else if (g.act == 1) v[e] = fmaxf(v[e], 0.0f);
            }
            *reinterpret_cast<f32x4*>(op + o) = v;
          }
        }
    }
    return;
  }
  float* out_b = out + (size_t)b * g.Co * HWo;
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int o = ob * MW + mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
      if (o < g.Co) {
        const float bo = bias[o];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
          if (pok[nt]) out_b[(size_t)o * HWo + pc[nt]] = acc[mt][nt][r] + bo;
      }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// backward (data): dCol = W^T . gO per 32(k) x 32(pixel) tile on MFMA, then grad_mask / grad_offset / grad_input
// COH = CoPad2 / 2 = number of k-pairs of the reduction over output channels (gO tile lives in COH registers)
// ---------------------------------------------------------------------------------------------------------------------
template <int COH>
__global__ void __launch_bounds__(256) dcn_bwd_data_kernel(const float* __restrict__ in, const float* __restrict__ wb,
                                                            const float* __restrict__ offset,
                                                            const float* __restrict__ mask,
                                                            const float* __restrict__ gout, Geom g,
                                                            float* __restrict__ gin, float* __restrict__ goff,
                                                            float* __restrict__ gmsk) {
  const bool want_gin = gin != nullptr;  // grad_input is optional: in C2-Matching the warped ref feature never needs it
  const int tid = threadIdx.x, l = tid & 63, hi = l >> 5, j = l & 31;
  const int wv = tid >> 6;
  const int b = blockIdx.y;
  const int HW = g.H * g.W, HWo = g.Ho * g.Wo;
  const int p0 = (blockIdx.x * 4 + wv) * 32;
  if (p0 >= HWo) return;
  const int p = p0 + j;
  const bool pok = p < HWo;
  const int pc = min(p, HWo - 1);
  const int py = pc / g.Wo, px = pc - py * g.Wo;
  const float* in_b = in + (size_t)b * g.C * HW;
  const float* off_b = offset + (size_t)b * g.dg * 2 * g.T * HWo;
  const float* msk_b = mask + (size_t)b * g.dg * g.T * HWo;
  const float* go_b = gout + (size_t)b * g.Co * HWo;
  float* gin_b = want_gin ? gin + (size_t)b * g.C * HW : nullptr;
  float* goff_b = goff + (size_t)b * g.dg * 2 * g.T * HWo;
  float* gmsk_b = gmsk + (size_t)b * g.dg * g.T * HWo;

  // B operand: gO[o = 2t + hi][pixel j], resident
  float gq[COH];
#pragma unroll
  for (int t = 0; t < COH; ++t) {
    const int o = 2 * t + hi;
    gq[t] = (o < g.Co && pok) ? go_b[(size_t)o * HWo + pc] : 0.0f;
  }

  // k tiles are independent (their gradients accumulate atomically): blockIdx.z splits them so that small batches
  // (training: 4 samples per GPU at 40x40 LR) still fill the 256 CUs
  const int nkt = g.KtotPad / 32;
  const int kt_per = (nkt + gridDim.z - 1) / gridDim.z;
  const int kt_end = min(nkt, (int)(blockIdx.z + 1) * kt_per);
  for (int kt = blockIdx.z * kt_per; kt < kt_end; ++kt) {
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
    const float* wa = wb + (size_t)hi * g.KtotPad + kt * 32 + j;  // A[i = k row][kk = o parity]
#pragma unroll
    for (int t = 0; t < COH; ++t) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(wa[(size_t)(2 * t) * g.KtotPad], gq[t], acc, 0, 0, 0);

    // lane owns rows i = e + 8*rq + 4*hi (e = 0..3) of column j: four quads of four consecutive k
    int gt_prev = -1;
    Tap tp;
#pragma unroll
    for (int rq = 0; rq < 4; ++rq) {
      const int k0 = kt * 32 + 8 * rq + 4 * hi;
      if (k0 < g.Ktot) {  // CPG % 4 == 0: a quad never straddles a (group, tap)
        const int gt = k0 / g.CPG, cig0 = k0 - gt * g.CPG;
        const int grp = gt / g.T, tap = gt - grp * g.T;
        if (gt != gt_prev) { tp = make_tap(g, off_b, msk_b, grp, tap, py, px, pc, pok); gt_prev = gt; }
        float mval = 0.0f, oh = 0.0f, ow = 0.0f;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float dc = acc[rq * 4 + e];
          const int c = grp * g.CPG + cig0 + e;
          const float* im = in_b + (size_t)c * HW;
          float v1, v2, v3, v4;
          const float bil = tap_sample(tp, im, v1, v2, v3, v4);
          // grad_mask (:307-310): sum_c dCol * (unmasked) bilinear value, only for in-range samples
          mval += dc * bil;
          // dmcn_get_coordinate_weight (:82-123); v* are already 0 for corners that do not contribute
          const float ghl = (float)(tp.hl + 1) - tp.ah, glh = tp.ah - (float)tp.hl;
          const float gwl = (float)(tp.wl + 1) - tp.aw, glw = tp.aw - (float)tp.wl;
          float wh = 0.0f, ww = 0.0f;
          wh += -1.0f * gwl * v1; wh += -1.0f * glw * v2; wh += gwl * v3; wh += glw * v4;
          ww += -1.0f * ghl * v1; ww += ghl * v2; ww += -1.0f * glh * v3; ww += glh * v4;
          const float top = dc * tp.mk;
          oh += wh * top;   // val += weight * dCol * mask (:317)
          ow += ww * top;
          // col2im scatter (:197-254): the four bilinear corners inside the image
          if (want_gin) {
            float* gi = gin_b + (size_t)c * HW;
            if (tp.c1 != 0.0f) atomicAdd(gi + tp.a1, tp.w1 * top);
            if (tp.c2 != 0.0f) atomicAdd(gi + tp.a2, tp.w2 * top);
            if (tp.c3 != 0.0f) atomicAdd(gi + tp.a3, tp.w3 * top);
            if (tp.c4 != 0.0f) atomicAdd(gi + tp.a4, tp.w4 * top);
          }
        }
        if (tp.inside) {
          atomicAdd(gmsk_b + (size_t)gt * HWo + pc, mval);
          atomicAdd(goff_b + (size_t)(2 * gt) * HWo + pc, oh);
          atomicAdd(goff_b + (size_t)(2 * gt + 1) * HWo + pc, ow);
        }
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// backward (offset / mask gradients only; grad_input not requested -- the C2-Matching case).
// Same MFMA product as dcn_bwd_data_kernel (dCol tile = Wb^T . gO, gO resident), but built on the forward's footing:
//   * gathers from the zero-bordered channels-last copy, float4 per 4 channels, no validity logic;
//   * the K rows of a tile are ordered (weight_relayout_kernel, order 2) so that every lane owns complete (tap, group)s of
//     its pixel -- both halves of one for 32-channel groups, combined with one cross-half add: grad_offset and grad_mask
//     (dcn_v2_im2col_cuda.cu:257-330) are plain coalesced stores, written exactly once -- no atomics, no zero fill;
//   * raw offsets, sampling state and gathers of a tile are issued in front of its MFMA chain (COH MFMAs cover them).
// ---------------------------------------------------------------------------------------------------------------------
template <int COH, int CPG>
__global__ void __launch_bounds__(256) dcn_bwd_offmask_kernel(const float* __restrict__ inl, const float* __restrict__ wb,
                                                               const float* __restrict__ offset,
                                                               const float* __restrict__ mask,
                                                               const float* __restrict__ gout, Geom g,
                                                               float* __restrict__ goff, float* __restrict__ gmsk) {
  constexpr int NGL = CPG == 8 ? 2 : 1;   // groups (half a group for CPG = 32) per lane and k tile
  constexpr int CPL = 16 / NGL;           // channels per lane and group
  constexpr int NQ = CPL / 4;
  constexpr int ROWS = 2 * COH;           // rows of Wb (output channels, padded) = rows of a staged tile
  constexpr int TILE = ROWS * 32;         // floats per staged k tile: Wb[o][kt*32 .. kt*32+32)
  extern __shared__ __attribute__((aligned(16))) float wtile[];   // [2][ROWS][32]
  const int tid = threadIdx.x, l = tid & 63, hi = l >> 5, j = l & 31;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int b = blockIdx.y;
  const int HWo = g.Ho * g.Wo;
  const int p0 = (blockIdx.x * 4 + wv) * 32;   // may lie beyond HWo for the last waves: they still hit the barriers
  int p = p0 + j;
  const bool pok = p < HWo;
  if ((g.Wo % 8 == 0) && (g.Ho % 4 == 0)) {   // 8 x 4 pixel patch per wave (gather locality, as in the forward kernel)
    const int tile = min(p0 / 32, HWo / 32 - 1), tpr = g.Wo / 8;
    const int ty = tile / tpr, tx = tile - ty * tpr;
    p = (ty * 4 + (j >> 3)) * g.Wo + tx * 8 + (j & 7);
  }
  const int pc = min(p, HWo - 1);
  const int py = pc / g.Wo, px = pc - py * g.Wo;
  const int Wp = g.W + 3;
  const float* in_b = inl + (size_t)b * g.C * (g.H + 3) * Wp;
  const float* off_b = offset + (size_t)b * g.dg * 2 * g.T * HWo;
  const float* msk_b = mask + (size_t)b * g.dg * g.T * HWo;
  const float* go_b = gout + (size_t)b * g.Co * HWo;
  float* goff_b = goff + (size_t)b * g.dg * 2 * g.T * HWo;
  float* gmsk_b = gmsk + (size_t)b * g.dg * g.T * HWo;
  const float fy = (float)(py * g.sh - g.ph), fx = (float)(px * g.sw - g.pw);
  const float Hf = (float)g.H, Wf = (float)g.W;

  // B operand: gO[o = 2t + hi][pixel j], resident
  float gq[COH];
#pragma unroll
  for (int t = 0; t < COH; ++t) {
    const int o = 2 * t + hi;
    gq[t] = (o < g.Co && pok) ? go_b[(size_t)o * HWo + pc] : 0.0f;
  }

  const int tpt = g.C / 32;        // k tiles per tap
  const int nkt = g.T * tpt;
  const int kt_per = (nkt + gridDim.z - 1) / gridDim.z;   // k tiles are independent: grid.z splits them for small batches
  const int kt_beg = blockIdx.z * kt_per, kt_end = min(nkt, (int)(blockIdx.z + 1) * kt_per);

  // A operands: the Wb tile of k tile kt (ROWS x 128 bytes) is DMA'd into LDS once per workgroup and shared by its four
  // waves (one dword per MFMA straight from global memory left each MFMA waiting for its own load).  One instruction
  // moves 8 rows (lane = row l>>3, 16-byte piece l&7); wave wv moves rows [wv*ROWS/4, (wv+1)*ROWS/4).
  auto stage = [&](int kt, int buf) {
    const float* src = wb + (size_t)(wv * (ROWS / 4) + (l >> 3)) * g.KtotPad + kt * 32 + 4 * (l & 7);
    float* dst = wtile + buf * TILE + wv * (ROWS / 4) * 32;
#pragma unroll
    for (int m = 0; m < ROWS / 32; ++m) glds_b128(src + (size_t)(8 * m) * g.KtotPad, dst + 8 * m * 32);
  };
  // raw offsets / mask of a tile, fetched one tile ahead
  struct Raw { float oh[NGL], ow[NGL], mk[NGL]; };
  auto group_of = [&](int gtile, int lg) { return CPG == 32 ? gtile : (CPG == 16 ? 2 * gtile + hi : 4 * gtile + 2 * lg + hi); };
  auto load_raw = [&](int kt, Raw& r) {
    const int tap = kt / tpt, gtile = kt - tap * tpt;
#pragma unroll
    for (int lg = 0; lg < NGL; ++lg) {
      const int gt = group_of(gtile, lg) * g.T + tap;
      r.oh[lg] = off_b[(size_t)(2 * gt) * HWo + pc];
      r.ow[lg] = off_b[(size_t)(2 * gt + 1) * HWo + pc];
      r.mk[lg] = msk_b[(size_t)gt * HWo + pc];
    }
  };

  Raw raw;
  if (kt_beg < kt_end) {
    stage(kt_beg, 0);
    load_raw(kt_beg, raw);
  }
  for (int kt = kt_beg; kt < kt_end; ++kt) {
    const int buf = (kt - kt_beg) & 1;
    const int tap = kt / tpt, gtile = kt - tap * tpt;
    const int ti = tap / g.kw, tj = tap - ti * g.kw;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();   // tile kt landed in wtile[buf]; every wave is done with wtile[buf ^ 1]
    if (kt + 1 < kt_end) stage(kt + 1, buf ^ 1);

    int gt[NGL];
    float hh[NGL], hw[NGL], lh[NGL], lw[NGL], mk[NGL];
    bool inside[NGL];
    f32x4 v[NGL][4][NQ];
#pragma unroll
    for (int lg = 0; lg < NGL; ++lg) {
      const int grp = group_of(gtile, lg);
      gt[lg] = grp * g.T + tap;
      mk[lg] = raw.mk[lg];
      const float ar = (fy + (float)(ti * g.dh)) + raw.oh[lg], ac = (fx + (float)(tj * g.dw)) + raw.ow[lg];
      inside[lg] = ar > -1.0f && ac > -1.0f && ar < Hf && ac < Wf;   // (:180, :82-87)
      const float ah = __builtin_amdgcn_fmed3f(ar, -1.0f, Hf), aw = __builtin_amdgcn_fmed3f(ac, -1.0f, Wf);
      const float fh = floorf(ah), fw = floorf(aw);
      lh[lg] = ah - fh; lw[lg] = aw - fw; hh[lg] = 1.0f - lh[lg]; hw[lg] = 1.0f - lw[lg];
      // channels-last, or (8-channel groups, Geom::in_grouped) the group-major copy: see dcn_fwd_nhwc_kernel
      const unsigned pixs = g.in_grouped ? (unsigned)CPG : (unsigned)g.C;
      const unsigned gbase = g.in_grouped ? (unsigned)(grp * (g.H + 3) * Wp * CPG) : (unsigned)(grp * CPG);
      const unsigned o1 = (unsigned)((int)fh * Wp + (int)fw + Wp + 1) * pixs + gbase + (CPG == 32 ? 16 * hi : 0);
      const unsigned o2 = o1 + pixs, o3 = o1 + (unsigned)Wp * pixs, o4 = o3 + pixs;
#pragma unroll
      for (int q = 0; q < NQ; ++q) {
        v[lg][0][q] = *reinterpret_cast<const f32x4*>(in_b + o1 + 4 * q);
        v[lg][1][q] = *reinterpret_cast<const f32x4*>(in_b + o2 + 4 * q);
        v[lg][2][q] = *reinterpret_cast<const f32x4*>(in_b + o3 + 4 * q);
        v[lg][3][q] = *reinterpret_cast<const f32x4*>(in_b + o4 + 4 * q);
      }
    }
    if (kt + 1 < kt_end) load_raw(kt + 1, raw);
    __builtin_amdgcn_sched_barrier(0);

    // dCol tile: A[i = k row j][kk = o parity hi] of k-pair t = wtile[buf][2t + hi][j]; operands of 8 k-pairs are read one
    // group ahead of the MFMAs that use them
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
    const float* wa = wtile + buf * TILE + l;   // + t * 64
    constexpr int AG = 8;
    float aq[2][AG];
#pragma unroll
    for (int t = 0; t < AG; ++t) aq[0][t] = wa[t * 64];
#pragma unroll
    for (int tg = 0; tg < COH / AG; ++tg) {
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(aq[tg & 1][0], gq[tg * AG], acc, 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      if (tg + 1 < COH / AG) {
#pragma unroll
        for (int t = 0; t < AG; ++t) aq[(tg + 1) & 1][t] = wa[((tg + 1) * AG + t) * 64];
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int t = 1; t < AG; ++t)
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(aq[tg & 1][t], gq[tg * AG + t], acc, 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }

    // acc[li] = dCol of this lane's channel li (weight_relayout_kernel, order 2)
#pragma unroll
    for (int lg = 0; lg < NGL; ++lg) {
      const float w1 = hh[lg] * hw[lg], w2 = hh[lg] * lw[lg], w3 = lh[lg] * hw[lg], w4 = lh[lg] * lw[lg];
      float mval = 0.0f, soh = 0.0f, sow = 0.0f;
#pragma unroll
      for (int c = 0; c < CPL; ++c) {
        const float dc = acc[lg * CPL + c];
        const float v1 = v[lg][0][c >> 2][c & 3], v2 = v[lg][1][c >> 2][c & 3];
        const float v3 = v[lg][2][c >> 2][c & 3], v4 = v[lg][3][c >> 2][c & 3];
        // grad_mask (:307-310): dCol * unmasked bilinear value
        mval = fmaf(dc, fmaf(w4, v4, fmaf(w3, v3, fmaf(w2, v2, w1 * v1))), mval);
        // dmcn_get_coordinate_weight (:82-123); border corners read 0
        soh = fmaf(dc, fmaf(lw[lg], v4 - v2, hw[lg] * (v3 - v1)), soh);
        sow = fmaf(dc, fmaf(lh[lg], v4 - v3, hh[lg] * (v2 - v1)), sow);
      }
      soh *= mk[lg];   // val += weight * dCol * mask (:317), mask factored out of the channel sum
      sow *= mk[lg];
      if (CPG == 32) {   // the other half-wave holds the other 16 channels of the same (pixel, group, tap)
        mval += __shfl_xor(mval, 32, 64);
        soh += __shfl_xor(soh, 32, 64);
        sow += __shfl_xor(sow, 32, 64);
      }
      if (!inside[lg]) { mval = 0.0f; soh = 0.0f; sow = 0.0f; }
      if (pok && (CPG != 32 || hi == 0)) {
        gmsk_b[(size_t)gt[lg] * HWo + pc] = mval;
        goff_b[(size_t)(2 * gt[lg]) * HWo + pc] = soh;
        goff_b[(size_t)(2 * gt[lg] + 1) * HWo + pc] = sow;
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// backward (weight): grad_weight[o][k] += sum_pixels gO[o][p] * col[k][p]; one workgroup = 32 k rows x all Co x a
// range of 64-pixel chunks (split-K).  Chunks are staged in LDS: lanes <-> pixels while gathering, rows as operands.
// ---------------------------------------------------------------------------------------------------------------------
constexpr int PCH = 64;        // pixels per chunk
constexpr int LDP = PCH + 1;   // padded row stride: conflict-free column reads

// NHWC: `in` is the zero-bordered channels-last copy (nchw_to_nhwc*_kernel) and CPG % 8 == 0: the 8 column rows a lane
// regenerates are 8 consecutive channels of one (group, tap) = two float4 gathers per corner instead of 32 scattered
// dword loads, and the sampling state needs no validity logic.
template <int MT, bool NHWC>  // MT = CoPad / 32
__global__ void __launch_bounds__(256) dcn_bwd_weight_kernel(const float* __restrict__ in,
                                                              const float* __restrict__ offset,
                                                              const float* __restrict__ mask,
                                                              const float* __restrict__ gout, Geom g, int chunks_per_b,
                                                              int nsplit, float* __restrict__ gw) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* go_l = lds;                         // [MT*32][LDP]
  float* col_l = lds + MT * 32 * LDP;        // [32][LDP]
  const int tid = threadIdx.x, l = tid & 63, hi = l >> 5, j = l & 31;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int kt = blockIdx.x, split = blockIdx.y;
  const int HW = g.H * g.W, HWo = g.Ho * g.Wo;
  constexpr int MPW = (MT + 3) / 4;  // m-tiles per wave
  constexpr int KS = !NHWC ? 1 : (MT == 1 ? 4 : (MT == 2 ? 2 : 1));   // pixel-pair split across waves (channels-last path)
  f32x16 acc[MPW];
#pragma unroll
  for (int s = 0; s < MPW; ++s)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[s][r] = 0.0f;

  // rows of go_l beyond Co stay zero
  for (int e = tid; e < MT * 32 * LDP; e += 256) go_l[e] = 0.0f;
  __syncthreads();

  const int total = g.B * chunks_per_b;
  if constexpr (NHWC) {
    // Chunk pipeline: the gO rows and the gathered corners of chunk n+1 are requested right after the barrier that
    // publishes chunk n and land in registers while the MFMAs of chunk n run; they are blended / written to LDS after
    // the second barrier.
    constexpr int NGO = MT * 8;          // gO rows per wave and chunk (rows wv, wv+4, ...)
    const int Wp = g.W + 3;
    const int k0 = kt * 32 + 8 * wv;     // this wave regenerates column rows k0 .. k0+7: 8 channels of one (group, tap)
    const int gt = k0 / g.CPG, cig0 = k0 - gt * g.CPG;
    const int grp = gt / g.T, tap = gt - grp * g.T;
    const int ti = tap / g.kw, tj = tap - ti * g.kw;
    float gor[NGO], w1, w2, w3, w4;
    f32x4 cv[4][2];
    auto fetch = [&](int ch) {
      const int b = ch / chunks_per_b, pc0 = (ch - b * chunks_per_b) * PCH;
      const int p = pc0 + l;
      const bool pok = p < HWo;
      const int pc = min(p, HWo - 1);
      const int py = pc / g.Wo, px = pc - py * g.Wo;
      const float* in_b = in + (size_t)b * g.C * (g.H + 3) * Wp;
      const float* off_b = offset + (size_t)b * g.dg * 2 * g.T * HWo;
      const float* msk_b = mask + (size_t)b * g.dg * g.T * HWo;
      const float* go_b = gout + (size_t)b * g.Co * HWo;
      const float oh = off_b[(size_t)(2 * gt) * HWo + pc], ow = off_b[(size_t)(2 * gt + 1) * HWo + pc];
      const float mk = pok ? msk_b[(size_t)gt * HWo + pc] : 0.0f;
#pragma unroll
      for (int r = 0; r < NGO; ++r) {
        const int o = wv + 4 * r;
        gor[r] = (pok && o < g.Co) ? go_b[(size_t)o * HWo + pc] : 0.0f;
      }
      const float ah = __builtin_amdgcn_fmed3f((float)(py * g.sh - g.ph + ti * g.dh) + oh, -1.0f, (float)g.H);
      const float aw = __builtin_amdgcn_fmed3f((float)(px * g.sw - g.pw + tj * g.dw) + ow, -1.0f, (float)g.W);
      const float fh = floorf(ah), fw = floorf(aw);
      const float lh = ah - fh, lw = aw - fw, hw = 1.0f - lw;
      const float mh = (1.0f - lh) * mk, ml = lh * mk;
      w1 = mh * hw; w2 = mh * lw; w3 = ml * hw; w4 = ml * lw;
      const int pixs = g.in_grouped ? g.CPG : g.C;   // floats between horizontally adjacent positions
      const int gbase = g.in_grouped ? grp * (g.H + 3) * Wp * g.CPG : grp * g.CPG;
      const float* s1 = in_b + (size_t)(((int)fh * Wp + (int)fw + Wp + 1) * pixs + gbase + cig0);
      const float* s3 = s1 + Wp * pixs;
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        cv[0][q] = *reinterpret_cast<const f32x4*>(s1 + 4 * q);
        cv[1][q] = *reinterpret_cast<const f32x4*>(s1 + pixs + 4 * q);
        cv[2][q] = *reinterpret_cast<const f32x4*>(s3 + 4 * q);
        cv[3][q] = *reinterpret_cast<const f32x4*>(s3 + pixs + 4 * q);
      }
    };
    if (split < total) fetch(split);
    for (int ch = split; ch < total; ch += nsplit) {
#pragma unroll
      for (int r = 0; r < NGO; ++r) go_l[(wv + 4 * r) * LDP + l] = gor[r];
#pragma unroll
      for (int q = 0; q < 2; ++q)
#pragma unroll
        for (int e = 0; e < 4; ++e)
          col_l[(8 * wv + 4 * q + e) * LDP + l] = fmaf(w4, cv[3][q][e], fmaf(w3, cv[2][q][e], fmaf(w2, cv[1][q][e], w1 * cv[0][q][e])));
      __syncthreads();
      if (ch + nsplit < total) fetch(ch + nsplit);
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (KS > 1) {
        // fewer than 4 m-tiles: the waves also split the chunk's pixel pairs (partial sums meet in the final atomics)
        const int mt = wv % MT, tb = (wv / MT) * (PCH / 2 / KS);
        const float* ar = go_l + (mt * 32 + j) * LDP + hi;
        const float* br = col_l + j * LDP + hi;
#pragma unroll
        for (int t = 0; t < PCH / 2 / KS; ++t)
          acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(ar[2 * (tb + t)], br[2 * (tb + t)], acc[0], 0, 0, 0);
      } else {
#pragma unroll
        for (int s = 0; s < MPW; ++s) {
          const int mt = wv + 4 * s;
          if (mt < MT) {
            const float* ar = go_l + (mt * 32 + j) * LDP + hi;   // A[i = o][kk = pixel parity]
            const float* br = col_l + j * LDP + hi;              // B[kk = pixel parity][j = k row]
#pragma unroll
            for (int t = 0; t < PCH / 2; ++t) acc[s] = __builtin_amdgcn_mfma_f32_32x32x2f32(ar[2 * t], br[2 * t], acc[s], 0, 0, 0);
          }
        }
      }
      __builtin_amdgcn_sched_barrier(0);
      __syncthreads();
    }
  } else {
    for (int ch = split; ch < total; ch += nsplit) {
      const int b = ch / chunks_per_b, pc0 = (ch - b * chunks_per_b) * PCH;
      const int p = pc0 + l;
      const bool pok = p < HWo;
      const int pc = min(p, HWo - 1);
      const int py = pc / g.Wo, px = pc - py * g.Wo;
      const float* in_b = in + (size_t)b * g.C * HW;
      const float* off_b = offset + (size_t)b * g.dg * 2 * g.T * HWo;
      const float* msk_b = mask + (size_t)b * g.dg * g.T * HWo;
      const float* go_b = gout + (size_t)b * g.Co * HWo;
      // gO chunk: wave wv loads rows wv, wv+4, ... (coalesced along pixels)
      for (int o = wv; o < g.Co; o += 4) go_l[o * LDP + l] = pok ? go_b[(size_t)o * HWo + pc] : 0.0f;
      // column rows 8wv .. 8wv+7 of this k tile
      int gt_prev = -1;
      Tap tp;
      for (int rr = 0; rr < 8; ++rr) {
        const int row = 8 * wv + rr, k = kt * 32 + row;
        float v = 0.0f;
        if (k < g.Ktot) {
          const int gt = k / g.CPG, cig = k - gt * g.CPG;
          const int grp = gt / g.T, tap = gt - grp * g.T;
          if (gt != gt_prev) { tp = make_tap(g, off_b, msk_b, grp, tap, py, px, pc, pok); gt_prev = gt; }
          float v1, v2, v3, v4;
          v = tap_sample(tp, in_b + (size_t)(grp * g.CPG + cig) * HW, v1, v2, v3, v4) * tp.mk;
        }
        col_l[row * LDP + l] = v;
      }
      __syncthreads();
#pragma unroll
      for (int s = 0; s < MPW; ++s) {
        const int mt = wv + 4 * s;
        if (mt < MT) {
          const float* ar = go_l + (mt * 32 + j) * LDP + hi;   // A[i = o][kk = pixel parity]
          const float* br = col_l + j * LDP + hi;              // B[kk = pixel parity][j = k row]
#pragma unroll
          for (int t = 0; t < PCH / 2; ++t) acc[s] = __builtin_amdgcn_mfma_f32_32x32x2f32(ar[2 * t], br[2 * t], acc[s], 0, 0, 0);
        }
      }
      __syncthreads();
    }
  }

  const int k = kt * 32 + j;
  if (k < g.Ktot) {
    const int gt = k / g.CPG, cig = k - gt * g.CPG;
    const int grp = gt / g.T, tap = gt - grp * g.T;
    const int c = grp * g.CPG + cig;
#pragma unroll
    for (int s = 0; s < MPW; ++s) {
      const int mt = KS > 1 ? wv % MT : wv + 4 * s;
      if (mt < MT) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int o = mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
          if (o < g.Co) atomicAdd(gw + ((size_t)o * g.C + c) * g.T + tap, acc[s][r]);
        }
      }
    }
  }
}

// grad_bias[o] = sum_{b,p} gO[b][o][p]   (Sgemv with ones, dcn_v2_cuda.cu:324-329)
__global__ void __launch_bounds__(256) dcn_bias_grad_kernel(const float* __restrict__ gout, int B, int Co, int HWo,
                                                             float* __restrict__ gb) {
  __shared__ float red[256];
  const int o = blockIdx.x;
  float s = 0.0f;
  for (int b = 0; b < B; ++b) {
    const float* r = gout + ((size_t)b * Co + o) * HWo;
    for (int p = threadIdx.x; p < HWo; p += 256) s += r[p];
  }
  red[threadIdx.x] = s;
  __syncthreads();
  for (int off = 128; off > 0; off >>= 1) {
    if (threadIdx.x < off) red[threadIdx.x] += red[threadIdx.x + off];
    __syncthreads();
  }
  if (threadIdx.x == 0) gb[o] = red[0];
}

// ---------------------------------------------------------------------------------------------------------------------
// offset / mask assembly of DCN_sep_pre_multi_offset.forward (dcn_v2.py:229-245) in one pass
// ---------------------------------------------------------------------------------------------------------------------
constexpr int kAbsSlots = 256;  // abs_sum is an array of this many doubles: spreads the atomics of ~10^5 workgroups
constexpr int kFusePix = 4;     // pixels per thread

__global__ void __launch_bounds__(256) fuse_offsets_kernel(const float* __restrict__ conv_out,
                                                            const float2* __restrict__ pre, int dg, int K, int HW,
                                                            float* __restrict__ offset, float* __restrict__ mask,
                                                            double* __restrict__ abs_sum) {
  __shared__ float red[256];
  const int ch = blockIdx.y, b = blockIdx.z;  // ch over 3*dg*K conv channels
  const int n2 = 2 * dg * K;
  const float* src = conv_out + ((size_t)b * 3 * dg * K + ch) * HW;
  float a = 0.0f;
#pragma unroll
  for (int e = 0; e < kFusePix; ++e) {
    const int p = (blockIdx.x * kFusePix + e) * 256 + threadIdx.x;
    if (p < HW) {
      const float v = src[p];
      if (ch < n2) {
        float add = 0.0f;
        if (pre) {
          const float2 f = pre[((size_t)b * K + (ch >> 1) % K) * HW + p];  // (x, y); even offset channels are dy
          add = (ch & 1) ? f.x : f.y;
        }
        offset[((size_t)b * n2 + ch) * HW + p] = v + add;
        a += fabsf(v);
      } else {
        mask[((size_t)b * dg * K + (ch - n2)) * HW + p] = 1.0f / (1.0f + expf(-v));
      }
    }
  }
  if (abs_sum && ch < n2) {
    red[threadIdx.x] = a;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
      if (threadIdx.x < off) red[threadIdx.x] += red[threadIdx.x + off];
      __syncthreads();
    }
    if (threadIdx.x == 0) atomicAdd(abs_sum + ((blockIdx.x * 31 + blockIdx.y * 7 + blockIdx.z) & (kAbsSlots - 1)), (double)red[0]);
  }
}

}  // namespace dcn
}  // namespace c2m

// =====================================================================================================================
// C-ABI
// =====================================================================================================================
using namespace c2m;
using c2m::dcn::Geom;

namespace {
inline size_t align256(size_t v) { return (v + 255) & ~size_t(255); }

int make_geom(Geom& g, int B, int C, int H, int W, int Co, int kh, int kw, int sh, int sw, int ph, int pw, int dh,
              int dw, int dg) {
  if (B <= 0 || C <= 0 || H <= 0 || W <= 0 || Co <= 0 || kh <= 0 || kw <= 0 || sh <= 0 || sw <= 0 || ph < 0 || pw < 0 ||
      dh <= 0 || dw <= 0 || dg <= 0 || C % dg != 0)
    return C2M_ERR_INVALID_ARG;
  g.B = B; g.C = C; g.H = H; g.W = W; g.Co = Co; g.kh = kh; g.kw = kw; g.sh = sh; g.sw = sw; g.ph = ph; g.pw = pw;
  g.dh = dh; g.dw = dw; g.dg = dg;
  g.in_grouped = 0;
  g.range_flag = nullptr;
  g.out_nhwc = 0; g.out_pix_pitch = 0; g.out_row_pitch = 0; g.out_img_pitch = 0; g.act = 0; g.slope = 0.0f;
  g.Ho = (H + 2 * ph - (dh * (kh - 1) + 1)) / sh + 1;
  g.Wo = (W + 2 * pw - (dw * (kw - 1) + 1)) / sw + 1;
  if (g.Ho <= 0 || g.Wo <= 0) return C2M_ERR_INVALID_ARG;
  g.T = kh * kw;
  g.CPG = C / dg;
  g.Ktot = dg * g.T * g.CPG;
  g.KtotPad = (g.Ktot + 31) / 32 * 32;
  return C2M_OK;
}

// forward tiling: MT m-tiles of 32 output channels per wave (1, 2, 4 or 8), NT pixel tiles so that MT*NT*16 <= 128 acc regs
inline int fwd_mt(int Co) {
  const int need = (Co + 31) / 32;
  return need <= 1 ? 1 : need <= 2 ? 2 : need <= 4 ? 4 : 8;
}
inline int copad_fwd(int Co) {
  const int mt = fwd_mt(Co);
  return (Co + mt * 32 - 1) / (mt * 32) * (mt * 32);
}
inline int copad2(int Co) { return Co <= 64 ? 64 : Co <= 128 ? 128 : Co <= 256 ? 256 : -1; }

template <int MT, int NT, int CPG, int GC, bool SPLITG, bool BF16, bool F16X2>
int launch_fwd_nhwc(hipStream_t st, const float* inl, const float* wt, const float* bias, const float* off,
                    const float* msk, const Geom& g, float* out) {
  const int HWo = g.Ho * g.Wo;
  const size_t lds = (BF16 ? 2 : 4) * 2 * (size_t)GC * CPG * MT * 32;
  static unsigned long long lds_set = 0;
  if (lds > 48 * 1024)
    if (int rc = ensure_dynamic_lds(reinterpret_cast<const void*>(&dcn::dcn_fwd_nhwc_kernel<MT, NT, CPG, GC, SPLITG, BF16, F16X2>), lds, lds_set))
      return rc;
  dim3 grid(ceil_div(HWo, 4 * NT * 32), g.B, g.CoPad / (MT * 32));
  hipLaunchKernelGGL((dcn::dcn_fwd_nhwc_kernel<MT, NT, CPG, GC, SPLITG, BF16, F16X2>), grid, dim3(256), lds, st, inl, wt, bias, off, msk, g, out);
  return C2M_OK;
}

// groups per weight chunk: the largest power of two dividing dg with chunk <= 32 KiB (and >= 4 KiB so that every wave's
// quarter is a whole number of 1 KiB DMA pieces)
template <int MT, int NT, int CPG, bool SPLITG, bool BF16, bool F16X2>
int pick_gc_fwd_nhwc(hipStream_t st, const float* inl, const float* wt, const float* bias, const float* off,
                     const float* msk, const Geom& g, float* out) {
  constexpr int ROWB = CPG * MT * 32 * (BF16 ? 2 : 4);  // bytes of one group's weight rows
#ifndef C2M_DCN_F16_CHUNK4_KB
#define C2M_DCN_F16_CHUNK4_KB 32
#endif
  // groups that fit a 32 KiB chunk (64 KiB for the one-wave-per-SIMD MT = 8 variant)
  constexpr int FIT = ((MT == 8 ? 64 : (F16X2 && MT == 4 && CPG == 16) ? C2M_DCN_F16_CHUNK4_KB : 32) * 1024) / ROWB;
  if constexpr (FIT >= 8) {
    if (g.dg % 8 == 0) return launch_fwd_nhwc<MT, NT, CPG, 8, SPLITG, BF16, F16X2>(st, inl, wt, bias, off, msk, g, out);
  }
  if constexpr (FIT >= 4) {
    if (g.dg % 4 == 0) return launch_fwd_nhwc<MT, NT, CPG, 4, SPLITG, BF16, F16X2>(st, inl, wt, bias, off, msk, g, out);
  }
  static_assert(FIT >= 2, "two groups' weight rows must fit one chunk");
  return launch_fwd_nhwc<MT, NT, CPG, 2, SPLITG, BF16, F16X2>(st, inl, wt, bias, off, msk, g, out);   // use_nhwc() guarantees an even dg
}

template <int CPG, bool SPLITG, bool BF16, bool F16X2 = false>
int dispatch_fwd_nhwc(hipStream_t st, int mt, const float* inl, const float* wt, const float* bias, const float* off,
                      const float* msk, const Geom& g, float* out) {
  // Register budget (256 VGPRs at 2 waves/SIMD): MT*NT*16 accumulators + two generations of gathered corners
  // (NT*4*CPG/2 values each) + column values.  Co > 128 is split over grid.z (each workgroup re-gathers: cheap next to
  // the MFMA work of >= 128 output channels).
  // one pixel tile per wave everywhere: for 8-channel groups that means ~110 VGPRs = 4 waves/SIMD, which hides the gather
  // latency better than a second pixel tile amortises the LDS weight reads (large layer, B=16: 10.4 -> 8.4 ms)
  constexpr int NT2 = 1;
  switch (mt) {
    case 1: return pick_gc_fwd_nhwc<1, NT2, CPG, SPLITG, BF16, F16X2>(st, inl, wt, bias, off, msk, g, out);
    case 2: return pick_gc_fwd_nhwc<2, NT2, CPG, SPLITG, BF16, F16X2>(st, inl, wt, bias, off, msk, g, out);
    case 8:
      // bf16: the kernel is gather/blend bound, so all 256 output channels share one gathered column (one wave per SIMD,
      // 128 accumulator registers) instead of splitting Co over grid.z and gathering twice.  (fp32: measured, no gain.)
      if constexpr (BF16 && CPG == 32) return pick_gc_fwd_nhwc<8, 1, CPG, SPLITG, BF16, false>(st, inl, wt, bias, off, msk, g, out);
      if constexpr (F16X2 && CPG == 32 && C2M_DCN_F16_MT8 != 0) return pick_gc_fwd_nhwc<8, 1, CPG, SPLITG, false, true>(st, inl, wt, bias, off, msk, g, out);
      [[fallthrough]];
    default: return pick_gc_fwd_nhwc<4, 1, CPG, SPLITG, BF16, F16X2>(st, inl, wt, bias, off, msk, g, out);
  }
}

// channels-last fast path: 8/16/32 channels per deformable group and an even number of groups (its step pairs and weight
// chunks cover two groups at a time); everything else takes the NCHW kernel
// (and samples whose staged copy or offset planes reach 2 GiB: the kernel addresses them with 32-bit buffer offsets)
inline bool use_nhwc(const Geom& g) {
  const unsigned long long lim = 1ull << 31;
  const unsigned long long copy = 4ull * g.C * (g.H + 3) * (g.W + 3), offs = 4ull * g.dg * 2 * g.T * g.Ho * g.Wo;
  return (g.CPG == 8 || g.CPG == 16 || g.CPG == 32) && g.dg % 2 == 0 && copy < lim && offs < lim;
}

template <int MT, int NT>
void launch_fwd(hipStream_t st, const float* in, const float* wt, const float* bias, const float* off, const float* msk,
                const Geom& g, float* out) {
  const int HWo = g.Ho * g.Wo;
  dim3 grid(ceil_div(HWo, 4 * NT * 32), g.B, g.CoPad / (MT * 32));
  hipLaunchKernelGGL((dcn::dcn_fwd_mfma_kernel<MT, NT>), grid, dim3(256), 0, st, in, wt, bias, off, msk, g, out);
}
}  // namespace

extern "C" size_t c2m_dcn_v2_forward_workspace_bytes(int B, int C, int H, int W, int Co, int kh, int kw, int dg) {
  Geom g;
  if (make_geom(g, B, C, H, W, Co, kh, kw, 1, 1, kh, kw, 1, 1, dg) != C2M_OK) return 0;
  size_t n = align256(sizeof(float) * (size_t)g.KtotPad * copad_fwd(Co));
  if (use_nhwc(g)) n += align256(sizeof(float) * (size_t)B * C * (H + 3) * (W + 3));  // zero-bordered channels-last copy
  return n;
}

namespace {
// zero-bordered channels-last copy of `input` (fp32, or bf16 for the bf16-MFMA forward)
template <typename OutT>
void launch_nhwc_copy(hipStream_t st, const float* input, int B, int C, int H, int W, OutT* out, bool grouped8 = false) {
  if (grouped8)
    hipLaunchKernelGGL((dcn::nchw_to_grouped_kernel<OutT, 8>), dim3(ceil_div((H + 3) * (W + 3), 256), C / 8, B), dim3(256), 0, st,
                       input, C, H, W, out);
  else if (C % 64 == 0)
    hipLaunchKernelGGL(dcn::nchw_to_nhwc64_kernel<OutT>, dim3(ceil_div((H + 3) * (W + 3), 64), C / 64, B), dim3(256), 0, st,
                       input, C, H, W, out);
  else
    hipLaunchKernelGGL(dcn::nchw_to_nhwc_kernel<OutT>, dim3(ceil_div((H + 3) * (W + 3), 32), ceil_div(C, 32), B), dim3(256), 0,
                       st, input, C, H, W, out);
}

// Options of the fused decoder path (c2m_dcn_v2_forward_nhwc_f32): the caller already holds the zero-bordered channels-last
// copy (shared with the offset convolutions) and the re-laid-out weights (cached while the weights do not change).
struct FwdExt {
  const float* inl = nullptr;   // bordered channels-last input [B][H+3][W+3][C]; nullptr: made here from `input`
  int in_grouped = 0;           // `inl` is group-major [B][dg][H+3][W+3][C/dg] instead (8-channel groups only)
  const float* wt = nullptr;    // weights in the forward kernel's layout; nullptr: re-laid-out here from `weight`
  int out_nhwc = 0, out_pix_pitch = 0, out_row_pitch = 0, act = 0;
  long long out_img_pitch = 0;
  float slope = 0.0f;
  int f16x2 = 0;                // `wt` holds the f16 x 2 image (c2m_dcn_v2_relayout_f16x2): GEMM on the f16 matrix pipe
  int* range_flag = nullptr;
};

int dcn_forward(c2m_stream_t stream, const float* input, const float* weight, const float* bias, const float* offset,
                const float* mask, int B, int C, int H, int W, int Co, int kh, int kw, int sh, int sw, int ph, int pw,
                int dh, int dw, int dg, float* output, void* workspace, size_t workspace_bytes, bool want_bf16,
                const FwdExt& ext = FwdExt()) {
  if ((!input && !ext.inl) || (!weight && !ext.wt) || !bias || !offset || !mask || !output) return C2M_ERR_INVALID_ARG;
  Geom g;
  int rc = make_geom(g, B, C, H, W, Co, kh, kw, sh, sw, ph, pw, dh, dw, dg);
  if (rc != C2M_OK) return rc;
  if (g.CPG % 2 != 0) return C2M_ERR_UNSUPPORTED;  // k-pairs of the fp32 MFMA never straddle a (group, tap)
  g.CoPad = copad_fwd(Co);
  const bool nhwc = use_nhwc(g);
  if ((ext.inl || ext.wt || ext.out_nhwc) && (!nhwc || want_bf16)) return C2M_ERR_UNSUPPORTED;
  if (ext.out_nhwc && (Co % 4 != 0 || ext.out_pix_pitch % 4 != 0 || ext.out_row_pitch % 4 != 0 || ext.out_img_pitch % 4 != 0 ||
                       ((uintptr_t)output & 15)))
    return C2M_ERR_UNSUPPORTED;
  g.out_nhwc = ext.out_nhwc; g.out_pix_pitch = ext.out_pix_pitch; g.out_row_pitch = ext.out_row_pitch;
  g.out_img_pitch = ext.out_img_pitch; g.act = ext.act; g.slope = ext.slope;
  g.range_flag = ext.range_flag;
  const size_t wbytes = ext.wt ? 0 : align256(sizeof(float) * (size_t)g.KtotPad * g.CoPad);
  const size_t need = wbytes + ((nhwc && !ext.inl) ? align256(sizeof(float) * (size_t)B * C * (H + 3) * (W + 3)) : 0);
  if (need > 0 && (!workspace || workspace_bytes < need)) return C2M_ERR_WORKSPACE;
  hipStream_t st = as_stream(stream);
  float* wt = ext.wt ? const_cast<float*>(ext.wt) : static_cast<float*>(workspace);
  float* inl = ext.inl ? const_cast<float*>(ext.inl) : reinterpret_cast<float*>(static_cast<char*>(workspace) + wbytes);
  // 8-channel groups are processed as virtual groups of two (see SPLITG): the kernel and the weight re-layout see the
  // virtual grouping, which leaves the K order a plain (tap, group, kk) order over real channels.  (Measured, B=16: large
  // layer 16.4 -> 13.7 ms on random flows, 7.6 -> 7.1 ms on coherent ones; 16-channel groups lose 2-10 %, so they stay.)
  const bool split = nhwc && g.CPG == 8 && g.dg % 4 == 0;   // the virtual grouping must keep an even group count
  Geom gk = g;
  if (split) { gk.CPG = 2 * g.CPG; gk.dg = g.dg / 2; }
  // bf16 MFMA variant: channels-last geometries whose half-run is a multiple of 8 channels; anything else computes in fp32
  const bool bf16 = want_bf16 && nhwc && gk.CPG >= 16;
  // 8-channel groups gather from a group-major copy (see Geom::in_grouped); a caller-provided copy says which it is
  if (ext.in_grouped && !(ext.inl && nhwc && g.CPG == 8)) return C2M_ERR_UNSUPPORTED;
  const bool grouped = ext.inl ? ext.in_grouped != 0 : (nhwc && g.CPG == 8);
  g.in_grouped = gk.in_grouped = grouped ? 1 : 0;
  if (nhwc && !ext.inl) {
    if (bf16) launch_nhwc_copy(st, input, B, C, H, W, reinterpret_cast<__bf16*>(inl), grouped);
    else launch_nhwc_copy(st, input, B, C, H, W, inl, grouped);
  }
  if (!ext.wt) {
    if (bf16)
      hipLaunchKernelGGL(dcn::weight_relayout_bf16_kernel, dim3(ceil_div(g.CoPad * g.Ktot, 256)), dim3(256), 0, st, weight, gk,
                         reinterpret_cast<__bf16*>(wt));
    else
      hipLaunchKernelGGL(dcn::weight_relayout_kernel, dim3(ceil_div(g.CoPad * g.KtotPad, 256)), dim3(256), 0, st, weight, gk, 0,
                         nhwc ? 1 : 0, wt, (float*)nullptr);
  }
  if ((rc = check_launch()) != C2M_OK) return rc;
  if (ext.f16x2 && (!ext.wt || !nhwc || gk.CPG < 16)) return C2M_ERR_UNSUPPORTED;
  gk.range_flag = g.range_flag;
  if (nhwc) {
    ProfileScope prof(C2M_KERNEL_DCN_FWD, st);
    if (ext.f16x2) {
      if (split) rc = dispatch_fwd_nhwc<16, true, false, true>(st, fwd_mt(Co), inl, wt, bias, offset, mask, gk, output);
      else if (g.CPG == 16) rc = dispatch_fwd_nhwc<16, false, false, true>(st, fwd_mt(Co), inl, wt, bias, offset, mask, g, output);
      else rc = dispatch_fwd_nhwc<32, false, false, true>(st, fwd_mt(Co), inl, wt, bias, offset, mask, g, output);
    } else if (bf16) {
      if (split) rc = dispatch_fwd_nhwc<16, true, true>(st, fwd_mt(Co), inl, wt, bias, offset, mask, gk, output);
      else if (g.CPG == 16) rc = dispatch_fwd_nhwc<16, false, true>(st, fwd_mt(Co), inl, wt, bias, offset, mask, g, output);
      else rc = dispatch_fwd_nhwc<32, false, true>(st, fwd_mt(Co), inl, wt, bias, offset, mask, g, output);
    } else if (split) {
      rc = dispatch_fwd_nhwc<16, true, false>(st, fwd_mt(Co), inl, wt, bias, offset, mask, gk, output);
    } else {
      switch (g.CPG) {
        case 8: rc = dispatch_fwd_nhwc<8, false, false>(st, fwd_mt(Co), inl, wt, bias, offset, mask, g, output); break;
        case 16: rc = dispatch_fwd_nhwc<16, false, false>(st, fwd_mt(Co), inl, wt, bias, offset, mask, g, output); break;
        default: rc = dispatch_fwd_nhwc<32, false, false>(st, fwd_mt(Co), inl, wt, bias, offset, mask, g, output); break;
      }
    }
    if (rc != C2M_OK) return rc;
  } else {
    ProfileScope prof(C2M_KERNEL_DCN_FWD, st);
    switch (fwd_mt(Co)) {
      case 1: launch_fwd<1, 4>(st, input, wt, bias, offset, mask, g, output); break;
      case 2: launch_fwd<2, 4>(st, input, wt, bias, offset, mask, g, output); break;
      case 4: launch_fwd<4, 2>(st, input, wt, bias, offset, mask, g, output); break;
      default: launch_fwd<8, 1>(st, input, wt, bias, offset, mask, g, output); break;
    }
  }
  return check_launch();
}
}  // namespace

extern "C" int c2m_nchw_to_nhwc_bordered_f32(c2m_stream_t stream, const float* input, int B, int C, int H, int W, float* out) {
  if (!input || !out || B <= 0 || C <= 0 || H <= 0 || W <= 0) return C2M_ERR_INVALID_ARG;
  launch_nhwc_copy(as_stream(stream), input, B, C, H, W, out);
  return check_launch();
}

extern "C" size_t c2m_dcn_v2_relayout_bytes(int C, int Co, int kh, int kw, int dg) {
  Geom g;
  if (make_geom(g, 1, C, 8, 8, Co, kh, kw, 1, 1, kh / 2, kw / 2, 1, 1, dg) != C2M_OK || !use_nhwc(g)) return 0;
  return sizeof(float) * (size_t)g.KtotPad * copad_fwd(Co);
}

extern "C" int c2m_dcn_v2_relayout_f32(c2m_stream_t stream, const float* weight, int C, int Co, int kh, int kw, int dg,
                                       float* wt) {
  if (!weight || !wt) return C2M_ERR_INVALID_ARG;
  Geom g;
  int rc = make_geom(g, 1, C, 8, 8, Co, kh, kw, 1, 1, kh / 2, kw / 2, 1, 1, dg);
  if (rc != C2M_OK) return rc;
  if (!use_nhwc(g)) return C2M_ERR_UNSUPPORTED;
  g.CoPad = copad_fwd(Co);
  Geom gk = g;
  if (g.CPG == 8 && g.dg % 4 == 0) { gk.CPG = 2 * g.CPG; gk.dg = g.dg / 2; }
  hipLaunchKernelGGL(dcn::weight_relayout_kernel, dim3(ceil_div(g.CoPad * g.KtotPad, 256)), dim3(256), 0, as_stream(stream),
                     weight, gk, 0, 1, wt, (float*)nullptr);
  return check_launch();
}

extern "C" int c2m_dcn_v2_forward_nhwc_f32(c2m_stream_t stream, const float* input_bordered, const float* wt,
                                           const float* bias, const float* offset, const float* mask, int B, int C, int H,
                                           int W, int Co, int kh, int kw, int sh, int sw, int ph, int pw, int dh, int dw,
                                           int dg, float* output, int out_nhwc, int out_pix_pitch, int out_row_pitch,
                                           long long out_img_pitch, int act, float slope, int input_grouped) {
  FwdExt ext;
  ext.inl = input_bordered; ext.in_grouped = input_grouped; ext.wt = wt; ext.out_nhwc = out_nhwc; ext.out_pix_pitch = out_pix_pitch;
  ext.out_row_pitch = out_row_pitch; ext.out_img_pitch = out_img_pitch; ext.act = act; ext.slope = slope;
  return dcn_forward(stream, nullptr, nullptr, bias, offset, mask, B, C, H, W, Co, kh, kw, sh, sw, ph, pw, dh, dw, dg, output,
                     nullptr, 0, false, ext);
}

// ---- f16 x 2 GEMM (channels-last path, >= 16 channels per -- possibly virtual -- group)
static bool f16x2_geom(Geom& g, Geom& gk, int C, int Co, int kh, int kw, int dg) {
  if (make_geom(g, 1, C, 8, 8, Co, kh, kw, 1, 1, kh / 2, kw / 2, 1, 1, dg) != C2M_OK || !use_nhwc(g)) return false;
  g.CoPad = copad_fwd(Co);
  gk = g;
  if (g.CPG == 8 && g.dg % 4 == 0) { gk.CPG = 2 * g.CPG; gk.dg = g.dg / 2; }
  return gk.CPG >= 16 && g.Ktot % 16 == 0;
}

extern "C" size_t c2m_dcn_v2_relayout_f16x2_bytes(int C, int Co, int kh, int kw, int dg) {
  Geom g, gk;
  if (!f16x2_geom(g, gk, C, Co, kh, kw, dg)) return 0;
  return (size_t)g.Ktot * g.CoPad * 4 + 256;   // two f16 pieces per weight + the float 1/S behind them
}

extern "C" int c2m_dcn_v2_relayout_f16x2(c2m_stream_t stream, const float* weight, int C, int Co, int kh, int kw, int dg,
                                         void* wt) {
  if (!weight || !wt) return C2M_ERR_INVALID_ARG;
  Geom g, gk;
  if (!f16x2_geom(g, gk, C, Co, kh, kw, dg)) return C2M_ERR_UNSUPPORTED;
  hipStream_t st = as_stream(stream);
  float* sinv = static_cast<float*>(wt) + (size_t)g.Ktot * g.CoPad;
  const long long n = (long long)Co * C * kh * kw;
  if (n > 0x7fffffffLL) return C2M_ERR_UNSUPPORTED;
  hipLaunchKernelGGL(dcn::weight_absmax_kernel, dim3(1), dim3(1024), 0, st, weight, (int)n, sinv);
  hipLaunchKernelGGL(dcn::weight_relayout_f16x2_kernel, dim3(ceil_div(g.CoPad * g.Ktot, 256)), dim3(256), 0, st, weight, gk, sinv,
                     static_cast<_Float16*>(wt));
  return check_launch();
}

extern "C" int c2m_dcn_v2_forward_nhwc_f16x2(c2m_stream_t stream, const float* input_bordered, const void* wt,
                                             const float* bias, const float* offset, const float* mask, int B, int C, int H,
                                             int W, int Co, int kh, int kw, int sh, int sw, int ph, int pw, int dh, int dw,
                                             int dg, float* output, int out_nhwc, int out_pix_pitch, int out_row_pitch,
                                             long long out_img_pitch, int act, float slope, int input_grouped, int* range_flag) {
  FwdExt ext;
  ext.inl = input_bordered; ext.in_grouped = input_grouped; ext.wt = static_cast<const float*>(wt); ext.out_nhwc = out_nhwc;
  ext.out_pix_pitch = out_pix_pitch; ext.out_row_pitch = out_row_pitch; ext.out_img_pitch = out_img_pitch; ext.act = act;
  ext.slope = slope; ext.f16x2 = 1; ext.range_flag = range_flag;
  return dcn_forward(stream, nullptr, nullptr, bias, offset, mask, B, C, H, W, Co, kh, kw, sh, sw, ph, pw, dh, dw, dg, output,
                     nullptr, 0, false, ext);
}

extern "C" int c2m_dcn_v2_forward_f32(c2m_stream_t stream, const float* input, const float* weight, const float* bias,
                                      const float* offset, const float* mask, int B, int C, int H, int W, int Co, int kh,
                                      int kw, int sh, int sw, int ph, int pw, int dh, int dw, int dg, float* output,
                                      void* workspace, size_t workspace_bytes) {
  return dcn_forward(stream, input, weight, bias, offset, mask, B, C, H, W, Co, kh, kw, sh, sw, ph, pw, dh, dw, dg, output,
                     workspace, workspace_bytes, false);
}

extern "C" int c2m_dcn_v2_forward_bf16mma_f32(c2m_stream_t stream, const float* input, const float* weight,
                                              const float* bias, const float* offset, const float* mask, int B, int C,
                                              int H, int W, int Co, int kh, int kw, int sh, int sw, int ph, int pw, int dh,
                                              int dw, int dg, float* output, void* workspace, size_t workspace_bytes) {
  return dcn_forward(stream, input, weight, bias, offset, mask, B, C, H, W, Co, kh, kw, sh, sw, ph, pw, dh, dw, dg, output,
                     workspace, workspace_bytes, true);
}

namespace {
template <int MT, bool NHWC>
int launch_bwd_weight_l(hipStream_t st, dim3 grid, const float* in, const float* off, const float* msk, const float* go,
                        const Geom& g, int chunks_per_b, int nsplit, float* gw) {
  const size_t lds = sizeof(float) * (size_t)(MT * 32 + 32) * dcn::LDP;
  static unsigned long long lds_set = 0;
  if (lds > 48 * 1024)
    if (int rc = ensure_dynamic_lds(reinterpret_cast<const void*>(&dcn::dcn_bwd_weight_kernel<MT, NHWC>), lds, lds_set)) return rc;
  hipLaunchKernelGGL((dcn::dcn_bwd_weight_kernel<MT, NHWC>), grid, dim3(256), lds, st, in, off, msk, go, g, chunks_per_b, nsplit, gw);
  return C2M_OK;
}
// `inl` != nullptr selects the channels-last variant (in = bordered copy)
template <int MT>
int launch_bwd_weight(hipStream_t st, dim3 grid, const float* in, const float* inl, const float* off, const float* msk,
                      const float* go, const Geom& g, int chunks_per_b, int nsplit, float* gw) {
  if (inl) return launch_bwd_weight_l<MT, true>(st, grid, inl, off, msk, go, g, chunks_per_b, nsplit, gw);
  return launch_bwd_weight_l<MT, false>(st, grid, in, off, msk, go, g, chunks_per_b, nsplit, gw);
}

struct BwdWs {
  size_t wb, inl, total;
  int CoPad2;
  bool nhwc;   // geometry admits the atomic-free offset/mask kernel (used when grad_input is not requested)
};
inline BwdWs bwd_ws(const Geom& g) {
  BwdWs w;
  w.CoPad2 = copad2(g.Co);
  w.nhwc = (g.CPG == 8 || g.CPG == 16 || g.CPG == 32) && g.C % 32 == 0;
  w.wb = 0;
  w.inl = align256(sizeof(float) * (size_t)(w.CoPad2 > 0 ? w.CoPad2 : 0) * g.KtotPad);
  w.total = w.inl + (w.nhwc ? align256(sizeof(float) * (size_t)g.B * g.C * (g.H + 3) * (g.W + 3)) : 0);
  return w;
}

template <int COH, int CPG>
int launch_bwd_offmask_cpg(hipStream_t st, dim3 grid, const float* inl, const float* wb, const float* off,
                           const float* msk, const float* go, const Geom& g, float* goff, float* gmsk) {
  const size_t lds = sizeof(float) * 2 * (size_t)(2 * COH) * 32;   // two staged Wb tiles
  static unsigned long long lds_set = 0;
  if (lds > 48 * 1024)
    if (int rc = ensure_dynamic_lds(reinterpret_cast<const void*>(&dcn::dcn_bwd_offmask_kernel<COH, CPG>), lds, lds_set))
      return rc;
  hipLaunchKernelGGL((dcn::dcn_bwd_offmask_kernel<COH, CPG>), grid, dim3(256), lds, st, inl, wb, off, msk, go, g, goff, gmsk);
  return C2M_OK;
}
template <int COH>
int launch_bwd_offmask(hipStream_t st, dim3 grid, const float* inl, const float* wb, const float* off, const float* msk,
                       const float* go, const Geom& g, float* goff, float* gmsk) {
  switch (g.CPG) {
    case 8: return launch_bwd_offmask_cpg<COH, 8>(st, grid, inl, wb, off, msk, go, g, goff, gmsk);
    case 16: return launch_bwd_offmask_cpg<COH, 16>(st, grid, inl, wb, off, msk, go, g, goff, gmsk);
    default: return launch_bwd_offmask_cpg<COH, 32>(st, grid, inl, wb, off, msk, go, g, goff, gmsk);
  }
}
}  // namespace

extern "C" size_t c2m_dcn_v2_backward_workspace_bytes(int B, int C, int H, int W, int Co, int kh, int kw, int sh, int sw,
                                                      int ph, int pw, int dh, int dw, int dg) {
  Geom g;
  if (make_geom(g, B, C, H, W, Co, kh, kw, sh, sw, ph, pw, dh, dw, dg) != C2M_OK) return 0;
  return bwd_ws(g).total;
}

extern "C" int c2m_dcn_v2_backward_f32(c2m_stream_t stream, const float* input, const float* weight, const float* bias,
                                       const float* offset, const float* mask, const float* grad_output, int B, int C,
                                       int H, int W, int Co, int kh, int kw, int sh, int sw, int ph, int pw, int dh,
                                       int dw, int dg, float* grad_input, float* grad_offset, float* grad_mask,
                                       float* grad_weight, float* grad_bias, void* workspace, size_t workspace_bytes) {
  (void)bias;
  if (!input || !weight || !offset || !mask || !grad_output || !grad_offset || !grad_mask || !grad_weight || !grad_bias)
    return C2M_ERR_INVALID_ARG;  // grad_input may be NULL: the scatter into the input gradient is then skipped
  Geom g;
  int rc = make_geom(g, B, C, H, W, Co, kh, kw, sh, sw, ph, pw, dh, dw, dg);
  if (rc != C2M_OK) return rc;
  const BwdWs ws = bwd_ws(g);
  if (g.CPG % 4 != 0 || ws.CoPad2 < 0) return C2M_ERR_UNSUPPORTED;
  if (!workspace || workspace_bytes < ws.total) return C2M_ERR_WORKSPACE;
  g.CoPad = (Co + 31) / 32 * 32;
  hipStream_t st = as_stream(stream);
  float* wb = reinterpret_cast<float*>(static_cast<char*>(workspace) + ws.wb);
  const int HW = H * W, HWo = g.Ho * g.Wo;

  // grad_input not requested + channels-last geometry: offset/mask gradients by plain stores (no zero fill needed)
  const bool offmask = !grad_input && ws.nhwc;
  hipError_t e = grad_input ? hipMemsetAsync(grad_input, 0, sizeof(float) * (size_t)B * C * HW, st) : hipSuccess;
  if (e == hipSuccess && !offmask) e = hipMemsetAsync(grad_offset, 0, sizeof(float) * (size_t)B * dg * 2 * g.T * HWo, st);
  if (e == hipSuccess && !offmask) e = hipMemsetAsync(grad_mask, 0, sizeof(float) * (size_t)B * dg * g.T * HWo, st);
  if (e == hipSuccess) e = hipMemsetAsync(grad_weight, 0, sizeof(float) * (size_t)Co * C * g.T, st);
  if (e != hipSuccess) { set_last_error(e); return C2M_ERR_LAUNCH; }

  hipLaunchKernelGGL(dcn::weight_relayout_kernel, dim3(ceil_div(max(g.CoPad, ws.CoPad2) * g.KtotPad, 256)), dim3(256), 0,
                     st, weight, g, ws.CoPad2, offmask ? 2 : 0, (float*)nullptr, wb);
  if ((rc = check_launch()) != C2M_OK) return rc;
  // zero-bordered channels-last copy of the input for the gathers of the offset/mask and weight kernels
  float* inl = ws.nhwc ? reinterpret_cast<float*>(static_cast<char*>(workspace) + ws.inl) : nullptr;
  // (8-channel groups: group-major, as in the forward -- a 32-byte sample run then shares its line with its neighbours)
  const bool bgrouped = inl && g.CPG == 8 && [] { const char* e = getenv("C2M_DCN_BWD_GROUPED"); return !(e && e[0] == '0'); }();
  g.in_grouped = bgrouped ? 1 : 0;
  if (inl) launch_nhwc_copy(st, input, B, C, H, W, inl, bgrouped);
  if (offmask) {
    ProfileScope prof(C2M_KERNEL_DCN_BWD_DATA, st);
    const int nkt = g.KtotPad / 32;
    int nz = ceil_div(2048, ceil_div(HWo, 128) * B);
    nz = nz < 1 ? 1 : (nz > nkt ? nkt : nz);
    dim3 grid(ceil_div(HWo, 128), B, nz);
    switch (ws.CoPad2) {
      case 64: rc = launch_bwd_offmask<32>(st, grid, inl, wb, offset, mask, grad_output, g, grad_offset, grad_mask); break;
      case 128: rc = launch_bwd_offmask<64>(st, grid, inl, wb, offset, mask, grad_output, g, grad_offset, grad_mask); break;
      default: rc = launch_bwd_offmask<128>(st, grid, inl, wb, offset, mask, grad_output, g, grad_offset, grad_mask); break;
    }
    if (rc != C2M_OK) return rc;
  } else {
    ProfileScope prof(C2M_KERNEL_DCN_BWD_DATA, st);
    const int nkt = g.KtotPad / 32;
    int nz = ceil_div(2048, ceil_div(HWo, 128) * B);
    nz = nz < 1 ? 1 : (nz > nkt ? nkt : nz);
    dim3 grid(ceil_div(HWo, 128), B, nz);
    switch (ws.CoPad2) {
      case 64: hipLaunchKernelGGL((dcn::dcn_bwd_data_kernel<32>), grid, dim3(256), 0, st, input, wb, offset, mask, grad_output, g, grad_input, grad_offset, grad_mask); break;
      case 128: hipLaunchKernelGGL((dcn::dcn_bwd_data_kernel<64>), grid, dim3(256), 0, st, input, wb, offset, mask, grad_output, g, grad_input, grad_offset, grad_mask); break;
      default: hipLaunchKernelGGL((dcn::dcn_bwd_data_kernel<128>), grid, dim3(256), 0, st, input, wb, offset, mask, grad_output, g, grad_input, grad_offset, grad_mask); break;
    }
  }
  if ((rc = check_launch()) != C2M_OK) return rc;
  {
    ProfileScope prof(C2M_KERNEL_DCN_BWD_WEIGHT, st);
    const int chunks_per_b = ceil_div(HWo, dcn::PCH);
    const int nkt = g.KtotPad / 32;
    int nsplit = ceil_div(1024, nkt);
    if (nsplit > B * chunks_per_b) nsplit = B * chunks_per_b;
    const int MT = g.CoPad / 32;
    dim3 grid(nkt, nsplit);
    switch (MT) {
      case 1: rc = launch_bwd_weight<1>(st, grid, input, inl, offset, mask, grad_output, g, chunks_per_b, nsplit, grad_weight); break;
      case 2: rc = launch_bwd_weight<2>(st, grid, input, inl, offset, mask, grad_output, g, chunks_per_b, nsplit, grad_weight); break;
      case 3: rc = launch_bwd_weight<3>(st, grid, input, inl, offset, mask, grad_output, g, chunks_per_b, nsplit, grad_weight); break;
      case 4: rc = launch_bwd_weight<4>(st, grid, input, inl, offset, mask, grad_output, g, chunks_per_b, nsplit, grad_weight); break;
      case 5: rc = launch_bwd_weight<5>(st, grid, input, inl, offset, mask, grad_output, g, chunks_per_b, nsplit, grad_weight); break;
      case 6: rc = launch_bwd_weight<6>(st, grid, input, inl, offset, mask, grad_output, g, chunks_per_b, nsplit, grad_weight); break;
      case 7: rc = launch_bwd_weight<7>(st, grid, input, inl, offset, mask, grad_output, g, chunks_per_b, nsplit, grad_weight); break;
      default: rc = launch_bwd_weight<8>(st, grid, input, inl, offset, mask, grad_output, g, chunks_per_b, nsplit, grad_weight); break;
    }
    if (rc != C2M_OK) return rc;
  }
  if ((rc = check_launch()) != C2M_OK) return rc;
  hipLaunchKernelGGL(dcn::dcn_bias_grad_kernel, dim3(Co), dim3(256), 0, st, grad_output, B, Co, HWo, grad_bias);
  return check_launch();
}

extern "C" int c2m_dcn_fuse_offsets_f32(c2m_stream_t stream, const float* conv_out, const float* pre_offset, int B,
                                        int dg, int K, int H, int W, float* offset, float* mask, double* abs_sum) {
  if (!conv_out || !offset || !mask || B <= 0 || dg <= 0 || K <= 0 || H <= 0 || W <= 0) return C2M_ERR_INVALID_ARG;
  dim3 grid(ceil_div(H * W, 256 * dcn::kFusePix), 3 * dg * K, B);
  hipLaunchKernelGGL(dcn::fuse_offsets_kernel, grid, dim3(256), 0, as_stream(stream), conv_out,
                     reinterpret_cast<const float2*>(pre_offset), dg, K, H * W, offset, mask, abs_sum);
  return check_launch();
}
