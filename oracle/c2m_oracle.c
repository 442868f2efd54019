/*
 * c2m_oracle.c -- CPU restatement (plain C) of the C2-Matching restoration hot path.
 *
 * THIS FILE IS TEST INFRASTRUCTURE.  Only tests/, __graft_entry__.smoke() and the
 * `cpu_baseline` leg of bench.py may load it.  The product path (c2-matching_amd/) never
 * links, imports or falls back to anything in oracle/.
 *
 * What it restates (all citations relative to the upstream reference checkout):
 *   - F.normalize over channels            mmsr/models/archs/corres_generation_arch.py:56-58
 *   - sample_patches / feature_match_index mmsr/models/archs/ref_map_util.py:4-23, 26-86
 *   - index_to_flow + 9 shifts x 3 scales  mmsr/models/archs/corres_generation_arch.py:29-46, 69-104
 *                                          mmsr/models/archs/arch_util.py:291-315 (tensor_shift)
 *   - DCNv2 forward                        mmsr/models/archs/DCNv2/src/cuda/dcn_v2_im2col_cuda.cu:25-54,125-195
 *                                          mmsr/models/archs/DCNv2/src/cuda/dcn_v2_cuda.cu:86-163
 *   - DCNv2 backward                       .../dcn_v2_im2col_cuda.cu:56-123,197-327, .../dcn_v2_cuda.cu:259-330
 *
 * Parity pinning:
 *   - correlation: pinned against golden vectors produced by importing the reference's own
 *     ref_map_util.py in the build container (tests/golden/make_golden.py -> tests/golden/ npz files).
 *   - DCNv2: the reference has NO CPU implementation (src/dcn_v2.h:38,72 throw) and no tests, and its CUDA
 *     sources cannot be built here (THC headers, nvcc).  PARITY UNPINNED BY THE REFERENCE for DCNv2; this
 *     restatement is pinned by known-answer tests instead (zero offset == conv2d, integer offsets == shifted
 *     gather, out-of-range == bias only, fp64 autograd of an independent torch restatement).
 *
 * Canonical floating-point order of the correlation (the HIP kernels reproduce it bit for bit):
 *   D[p][r]   = fmaf chain over channels c = 0..C-1, starting from +0:  acc = fmaf(in[c][p], ref[c][r], acc)
 *   S[q][n]   = (row_0 + row_1) + ... + row_{P-1},  row_i = t_i0 + (t_i1 + (... + t_i,P-1)) with t_ij = D[q+(i,j)][n+(i,j)]:
 *               taps of one patch row are added right-to-left, the rows left-to-right, plain fp32 adds (this is the order in
 *               which the HIP kernel forms the row sums in registers with two DPP shifted adds before parking them in LDS)
 *   ss[pix]   = fmaf chain over c of x*x ;  patch_ss = row-major plain adds of ss over the PxP window
 *   inv[n]    = 1.0f / (sqrtf(patch_ss[n]) + 1e-5f)            (ref_map_util.py:62-63, scale factored out of
 *                                                               the inner product -- differs from the reference's
 *                                                               scale-then-convolve by rounding only)
 *   corr[q][n]= S[q][n] * inv[n]   (is_norm) ;  arg-max = largest value, LOWEST n on ties (ref_map_util.py:69-76)
 *   max_val   = corr / (sqrtf(patch_ss_in[q]) + 1e-5f)  when norm_input (ref_map_util.py:78-84)
 * The reference's own order is whatever oneDNN's conv2d picks; both are fp32 and agree on every index of the
 * golden fixtures (see tests/test_oracle_golden.py, which also records the top-2 margins).
 *
 * Build:  make -C oracle      (gcc -O2 -mavx2 -mfma -ffp-contract=off -fopenmp -shared -fPIC)
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#ifdef _OPENMP
#include <omp.h>
#endif

#define C2M_OK 0
#define C2M_EINVAL 1
#define C2M_ENOMEM 2

int c2m_oracle_abi_version(void) { return 1; }

int c2m_oracle_num_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}

void c2m_oracle_set_num_threads(int n) {
#ifdef _OPENMP
  if (n > 0) omp_set_num_threads(n);
#else
  (void)n;
#endif
}

/* ------------------------------------------------------------------------------------------------ */
/* F.normalize(x.reshape(C,-1), dim=0): every pixel's C-vector divided by max(||v||2, 1e-12).       */
/* corres_generation_arch.py:56-58                                                                   */
/* ------------------------------------------------------------------------------------------------ */
int c2m_oracle_feature_normalize(const float* x, int C, int HW, float* out) {
  if (!x || !out || C <= 0 || HW <= 0) return C2M_EINVAL;
#pragma omp parallel for schedule(static)
  for (int p = 0; p < HW; ++p) {
    float ss = 0.0f;
    for (int c = 0; c < C; ++c) {
      float v = x[(size_t)c * HW + p];
      ss = fmaf(v, v, ss);
    }
    float nrm = sqrtf(ss);
    float den = nrm > 1e-12f ? nrm : 1e-12f;
    for (int c = 0; c < C; ++c) out[(size_t)c * HW + p] = x[(size_t)c * HW + p] / den;
  }
  return C2M_OK;
}

/* per-pixel sum of squares over channels, canonical fmaf chain */
static void pixel_sumsq(const float* x, int C, int HW, float* ss) {
#pragma omp parallel for schedule(static)
  for (int p = 0; p < HW; ++p) {
    float a = 0.0f;
    for (int c = 0; c < C; ++c) {
      float v = x[(size_t)c * HW + p];
      a = fmaf(v, v, a);
    }
    ss[p] = a;
  }
}

/* L2 norm of every PxP patch (over C,P,P), patches row-major with the given stride:
 * ref_map_util.py:19-22 (unfold order) and :63 / :80 (norm over dims 0,1,2).            */
int c2m_oracle_patch_norms(const float* x, int C, int H, int W, int P, int stride, float* norms) {
  if (!x || !norms || C <= 0 || P <= 0 || stride <= 0 || H < P || W < P) return C2M_EINVAL;
  int Hp = (H - P) / stride + 1, Wp = (W - P) / stride + 1;
  float* ss = (float*)malloc(sizeof(float) * (size_t)H * W);
  if (!ss) return C2M_ENOMEM;
  pixel_sumsq(x, C, H * W, ss);
  for (int py = 0; py < Hp; ++py)
    for (int px = 0; px < Wp; ++px) {
      float a = 0.0f;
      int first = 1;
      for (int i = 0; i < P; ++i)
        for (int j = 0; j < P; ++j) {
          float v = ss[(py * stride + i) * W + px * stride + j];
          if (first) { a = v; first = 0; } else { a = a + v; }
        }
      norms[py * Wp + px] = sqrtf(a);
    }
  free(ss);
  return C2M_OK;
}

/* one query pixel row of the pixel-level correlation D[px][r] for all Wq pixels of row y */
static void d_row(const float* fin, const float* fref, int C, int Wq, int HWq, int y, int Nr, float* Drow) {
  enum { PB = 8, RB = 1024 };
  memset(Drow, 0, sizeof(float) * (size_t)Wq * Nr);
#pragma omp parallel for schedule(dynamic, 1) collapse(2)
  for (int r0 = 0; r0 < Nr; r0 += RB)
    for (int p0 = 0; p0 < Wq; p0 += PB) {
      int rn = Nr - r0 < RB ? Nr - r0 : RB;
      int pn = Wq - p0 < PB ? Wq - p0 : PB;
      for (int c = 0; c < C; ++c) {
        const float* rr = fref + (size_t)c * Nr + r0;
        for (int pp = 0; pp < pn; ++pp) {
          float q = fin[(size_t)c * HWq + (size_t)y * Wq + p0 + pp];
          float* acc = Drow + (size_t)(p0 + pp) * Nr + r0;
#pragma omp simd
          for (int r = 0; r < rn; ++r) acc[r] = fmaf(q, rr[r], acc[r]);
        }
      }
    }
}

/*
 * feature_match_index (ref_map_util.py:26-86).  feat_in (C,Hq,Wq), feat_ref (C,Hr,Wr), fp32, contiguous.
 * Outputs max_idx int64 [Hqp*Wqp], max_val fp32 [Hqp*Wqp] with Hqp=(Hq-P)/in_stride+1 etc.
 * Only query patch rows qy in [qrow_begin, qrow_end) are computed (others left untouched): lets bench.py time
 * a bounded sample of a large pair.  Pass 0 and Hqp for everything.
 */
int c2m_oracle_feature_match_index(const float* fin, const float* fref, int C, int Hq, int Wq, int Hr, int Wr,
                                   int P, int in_stride, int ref_stride, int is_norm, int norm_input,
                                   int qrow_begin, int qrow_end, int64_t* max_idx, float* max_val) {
  if (!fin || !fref || !max_idx || !max_val) return C2M_EINVAL;
  if (C <= 0 || P <= 0 || in_stride <= 0 || ref_stride <= 0) return C2M_EINVAL;
  if (Hq < P || Wq < P || Hr < P || Wr < P) return C2M_EINVAL;
  const int Hqp = (Hq - P) / in_stride + 1, Wqp = (Wq - P) / in_stride + 1;
  const int Hrp = (Hr - P) / ref_stride + 1, Wrp = (Wr - P) / ref_stride + 1;
  if (qrow_begin < 0 || qrow_end > Hqp || qrow_begin > qrow_end) return C2M_EINVAL;
  const int Nr = Hr * Wr, Nrp = Hrp * Wrp;

  float* inv = (float*)malloc(sizeof(float) * (size_t)Nrp);
  float* qnorm = (float*)malloc(sizeof(float) * (size_t)Hqp * Wqp);
  float** ring = (float**)calloc((size_t)P, sizeof(float*));
  int* tag = (int*)malloc(sizeof(int) * (size_t)P);
  if (!inv || !qnorm || !ring || !tag) { free(inv); free(qnorm); free(ring); free(tag); return C2M_ENOMEM; }
  int rc = c2m_oracle_patch_norms(fref, C, Hr, Wr, P, ref_stride, inv);
  if (rc == C2M_OK) rc = c2m_oracle_patch_norms(fin, C, Hq, Wq, P, in_stride, qnorm);
  for (int i = 0; i < P && rc == C2M_OK; ++i) {
    ring[i] = (float*)malloc(sizeof(float) * (size_t)Wq * Nr);
    tag[i] = -1;
    if (!ring[i]) rc = C2M_ENOMEM;
  }
  if (rc == C2M_OK) {
    for (int n = 0; n < Nrp; ++n) inv[n] = 1.0f / (inv[n] + 1e-5f);

    for (int qy = qrow_begin; qy < qrow_end; ++qy) {
      for (int i = 0; i < P; ++i) {
        int y = qy * in_stride + i;
        if (tag[y % P] != y) { d_row(fin, fref, C, Wq, Hq * Wq, y, Nr, ring[y % P]); tag[y % P] = y; }
      }
#pragma omp parallel for schedule(static)
      for (int qx = 0; qx < Wqp; ++qx) {
        float best = 0.0f;
        int64_t bidx = 0;
        int have = 0;
        for (int ry = 0; ry < Hrp; ++ry)
          for (int rx = 0; rx < Wrp; ++rx) {
            float s = 0.0f;
            for (int i = 0; i < P; ++i) {
              const float* Dr = ring[(qy * in_stride + i) % P];
              float row = 0.0f;
              for (int j = P - 1; j >= 0; --j) {
                float d = Dr[(size_t)(qx * in_stride + j) * Nr + (size_t)(ry * ref_stride + i) * Wr + rx * ref_stride + j];
                row = (j == P - 1) ? d : d + row;
              }
              s = (i == 0) ? row : s + row;
            }
            int n = ry * Wrp + rx;
            float v = is_norm ? s * inv[n] : s;
            if (!have || v > best) { best = v; bidx = n; have = 1; } /* n ascending: strict > keeps the lowest n */
          }
        if (norm_input) best = best / (qnorm[qy * Wqp + qx] + 1e-5f);
        max_idx[qy * Wqp + qx] = bidx;
        max_val[qy * Wqp + qx] = best;
      }
    }
  }
  for (int i = 0; i < P; ++i) free(ring[i]);
  free(ring); free(tag); free(inv); free(qnorm);
  return rc;
}

/* ------------------------------------------------------------------------------------------------ */
/* index_to_flow (corres_generation_arch.py:29-46) followed by the nine tensor_shift copies at the   */
/* three scales (:69-104, arch_util.py:291-315).  max_idx is the (h-2)x(w-2) map of ONE sample;      */
/* off3 [9,h,w,2], off2 [9,2h,2w,2], off1 [9,4h,4w,2], last dim (x, y).                               */
/* ------------------------------------------------------------------------------------------------ */
static void shifted_scale(const int64_t* max_idx, int h, int w, int s, float* out) {
  const int hp = h - 2, wp = w - 2; /* query grid; flow decoded with the QUERY width (index_to_flow :32-34) */
  const int H = h * s, W = w * s;
  for (int k = 0; k < 9; ++k) {
    const int sh = (k / 3) * s, sw = (k % 3) * s;
    for (int y = 0; y < H; ++y)
      for (int x = 0; x < W; ++x) {
        float fx = 0.0f, fy = 0.0f;
        int ys = y - sh, xs = x - sw; /* tensor_shift: new[sh:, sw:] = old[:H-sh, :W-sw] */
        if (ys >= 0 && xs >= 0) {
          int yy = ys / s, xx = xs / s; /* repeat_interleave(s) */
          if (yy < hp && xx < wp) {      /* F.pad(..., (0,0,0,2,0,2)) leaves zeros bottom/right */
            int64_t idx = max_idx[yy * wp + xx];
            fx = (float)((idx % wp) - xx) * (float)s;
            fy = (float)((idx / wp) - yy) * (float)s;
          }
        }
        out[(((size_t)k * H + y) * W + x) * 2 + 0] = fx;
        out[(((size_t)k * H + y) * W + x) * 2 + 1] = fy;
      }
  }
}

int c2m_oracle_build_pre_offsets(const int64_t* max_idx, int h, int w, float* off3, float* off2, float* off1) {
  if (!max_idx || h < 3 || w < 3) return C2M_EINVAL;
  if (off3) shifted_scale(max_idx, h, w, 1, off3);
  if (off2) shifted_scale(max_idx, h, w, 2, off2);
  if (off1) shifted_scale(max_idx, h, w, 4, off1);
  return C2M_OK;
}

/* ------------------------------------------------------------------------------------------------ */
/* DCNv2                                                                                             */
/* ------------------------------------------------------------------------------------------------ */

/* dmcn_im2col_bilinear, dcn_v2_im2col_cuda.cu:25-54 */
static float bilinear(const float* im, int H, int W, float h, float w) {
  int hl = (int)floorf(h), wl = (int)floorf(w);
  int hh_ = hl + 1, wh_ = wl + 1;
  float lh = h - hl, lw = w - wl, hh = 1 - lh, hw = 1 - lw;
  float v1 = 0, v2 = 0, v3 = 0, v4 = 0;
  if (hl >= 0 && wl >= 0) v1 = im[hl * W + wl];
  if (hl >= 0 && wh_ <= W - 1) v2 = im[hl * W + wh_];
  if (hh_ <= H - 1 && wl >= 0) v3 = im[hh_ * W + wl];
  if (hh_ <= H - 1 && wh_ <= W - 1) v4 = im[hh_ * W + wh_];
  float w1 = hh * hw, w2 = hh * lw, w3 = lh * hw, w4 = lh * lw;
  return (w1 * v1 + w2 * v2 + w3 * v3 + w4 * v4);
}

/* dmcn_get_gradient_weight, dcn_v2_im2col_cuda.cu:56-80 */
static float scatter_weight(float ah, float aw, int h, int w, int H, int W) {
  if (ah <= -1 || ah >= H || aw <= -1 || aw >= W) return 0;
  int hl = (int)floorf(ah), wl = (int)floorf(aw), hh = hl + 1, wh = wl + 1;
  float weight = 0;
  if (h == hl && w == wl) weight = (h + 1 - ah) * (w + 1 - aw);
  if (h == hl && w == wh) weight = (h + 1 - ah) * (aw + 1 - w);
  if (h == hh && w == wl) weight = (ah + 1 - h) * (w + 1 - aw);
  if (h == hh && w == wh) weight = (ah + 1 - h) * (aw + 1 - w);
  return weight;
}

/* dmcn_get_coordinate_weight, dcn_v2_im2col_cuda.cu:82-123 */
static float coord_weight(float ah, float aw, int H, int W, const float* im, int dir) {
  if (ah <= -1 || ah >= H || aw <= -1 || aw >= W) return 0;
  int hl = (int)floorf(ah), wl = (int)floorf(aw), hh = hl + 1, wh = wl + 1;
  float weight = 0;
  if (dir == 0) {
    if (hl >= 0 && wl >= 0) weight += -1 * (wl + 1 - aw) * im[hl * W + wl];
    if (hl >= 0 && wh <= W - 1) weight += -1 * (aw - wl) * im[hl * W + wh];
    if (hh <= H - 1 && wl >= 0) weight += (wl + 1 - aw) * im[hh * W + wl];
    if (hh <= H - 1 && wh <= W - 1) weight += (aw - wl) * im[hh * W + wh];
  } else {
    if (hl >= 0 && wl >= 0) weight += -1 * (hl + 1 - ah) * im[hl * W + wl];
    if (hl >= 0 && wh <= W - 1) weight += (hl + 1 - ah) * im[hl * W + wh];
    if (hh <= H - 1 && wl >= 0) weight += -1 * (ah - hl) * im[hh * W + wl];
    if (hh <= H - 1 && wh <= W - 1) weight += (ah - hl) * im[hh * W + wh];
  }
  return weight;
}

typedef struct {
  int B, C, H, W, Co, kh, kw, sh, sw, ph, pw, dh, dw, dg, Ho, Wo;
} dcn_geom;

static int dcn_geom_init(dcn_geom* g, int B, int C, int H, int W, int Co, int kh, int kw, int sh, int sw, int ph,
                         int pw, int dh, int dw, int dg) {
  if (B <= 0 || C <= 0 || H <= 0 || W <= 0 || Co <= 0 || kh <= 0 || kw <= 0 || sh <= 0 || sw <= 0 || ph < 0 ||
      pw < 0 || dh <= 0 || dw <= 0 || dg <= 0 || C % dg != 0)
    return C2M_EINVAL;
  g->B = B; g->C = C; g->H = H; g->W = W; g->Co = Co; g->kh = kh; g->kw = kw; g->sh = sh; g->sw = sw;
  g->ph = ph; g->pw = pw; g->dh = dh; g->dw = dw; g->dg = dg;
  g->Ho = (H + 2 * ph - (dh * (kh - 1) + 1)) / sh + 1; /* dcn_v2_cuda.cu:86-87 */
  g->Wo = (W + 2 * pw - (dw * (kw - 1) + 1)) / sw + 1;
  if (g->Ho <= 0 || g->Wo <= 0) return C2M_EINVAL;
  return C2M_OK;
}

/* modulated_deformable_im2col_gpu_kernel, dcn_v2_im2col_cuda.cu:125-195, one sample.
 * col[(c*kh*kw + tap)][y*Wo + x] */
static void im2col_sample(const dcn_geom* g, const float* in, const float* off, const float* msk, float* col) {
  const int K = g->kh * g->kw, HWo = g->Ho * g->Wo, cpg = g->C / g->dg;
#pragma omp parallel for schedule(static)
  for (int c = 0; c < g->C; ++c) {
    const int grp = c / cpg;
    const float* im = in + (size_t)c * g->H * g->W;
    const float* o = off + (size_t)grp * 2 * K * HWo;
    const float* m = msk + (size_t)grp * K * HWo;
    for (int y = 0; y < g->Ho; ++y)
      for (int x = 0; x < g->Wo; ++x)
        for (int i = 0; i < g->kh; ++i)
          for (int j = 0; j < g->kw; ++j) {
            const int t = i * g->kw + j, p = y * g->Wo + x;
            const float oh = o[(size_t)(2 * t) * HWo + p], ow = o[(size_t)(2 * t + 1) * HWo + p];
            const float mk = m[(size_t)t * HWo + p];
            const float h_im = (y * g->sh - g->ph) + i * g->dh + oh;
            const float w_im = (x * g->sw - g->pw) + j * g->dw + ow;
            float val = 0;
            if (h_im > -1 && w_im > -1 && h_im < g->H && w_im < g->W) val = bilinear(im, g->H, g->W, h_im, w_im);
            col[((size_t)c * K + t) * HWo + p] = val * mk;
          }
  }
}

/* dcn_v2_cuda_forward, dcn_v2_cuda.cu:42-172: out[b] = bias (rank-1 GEMM :123-137) + W[Co x CK] . col[b] (:149-163) */
/* round-to-nearest-even to bfloat16 (8 significand bits), result widened back to float */
static float round_bf16(float v) {
  uint32_t u;
  memcpy(&u, &v, 4);
  if ((u & 0x7fffffffu) > 0x7f800000u) return v; /* NaN */
  u += 0x7fffu + ((u >> 16) & 1u);
  u &= 0xffff0000u;
  memcpy(&v, &u, 4);
  return v;
}

static int dcn_forward_impl(const float* in, const float* weight, const float* bias, const float* offset,
                            const float* mask, int B, int C, int H, int W, int Co, int kh, int kw, int sh, int sw,
                            int ph, int pw, int dh, int dw, int dg, float* out, int bf16_cols) {
  dcn_geom g;
  if (!in || !weight || !bias || !offset || !mask || !out) return C2M_EINVAL;
  int rc = dcn_geom_init(&g, B, C, H, W, Co, kh, kw, sh, sw, ph, pw, dh, dw, dg);
  if (rc) return rc;
  const int K = kh * kw, HWo = g.Ho * g.Wo, CK = C * K;
  float* col = (float*)malloc(sizeof(float) * (size_t)CK * HWo);
  if (!col) return C2M_ENOMEM;
  for (int b = 0; b < B; ++b) {
    im2col_sample(&g, in + (size_t)b * C * H * W, offset + (size_t)b * dg * 2 * K * HWo,
                  mask + (size_t)b * dg * K * HWo, col);
    if (bf16_cols)
      for (size_t e = 0; e < (size_t)CK * HWo; ++e) col[e] = round_bf16(col[e]);
    float* ob = out + (size_t)b * Co * HWo;
#pragma omp parallel for schedule(static)
    for (int o = 0; o < Co; ++o) {
      float* orow = ob + (size_t)o * HWo;
      for (int p = 0; p < HWo; ++p) orow[p] = 0.0f;
      for (int k = 0; k < CK; ++k) {
        const float wv = weight[(size_t)o * CK + k];
        const float* cr = col + (size_t)k * HWo;
#pragma omp simd
        for (int p = 0; p < HWo; ++p) orow[p] = fmaf(wv, cr[p], orow[p]);
      }
      for (int p = 0; p < HWo; ++p) orow[p] = orow[p] + bias[o];
    }
  }
  free(col);
  return C2M_OK;
}

int c2m_oracle_dcn_v2_forward(const float* in, const float* weight, const float* bias, const float* offset,
                              const float* mask, int B, int C, int H, int W, int Co, int kh, int kw, int sh, int sw,
                              int ph, int pw, int dh, int dw, int dg, float* out) {
  return dcn_forward_impl(in, weight, bias, offset, mask, B, C, H, W, Co, kh, kw, sh, sw, ph, pw, dh, dw, dg, out, 0);
}

/* Checker for the bf16-MFMA variant of the HIP forward (c2m_dcn_v2_forward_bf16mma_f32; the reference has no reduced-
 * precision path): the same operator with the column matrix rounded to bfloat16 before the fp32-accumulated product.
 * The caller hands over `in` and `weight` already rounded to bfloat16 (what the kernel's staging copy / weight re-layout
 * hold); sampling positions, bilinear weights, mask and bias stay float32. */
int c2m_oracle_dcn_v2_forward_bf16cols(const float* in, const float* weight, const float* bias, const float* offset,
                                       const float* mask, int B, int C, int H, int W, int Co, int kh, int kw, int sh,
                                       int sw, int ph, int pw, int dh, int dw, int dg, float* out) {
  return dcn_forward_impl(in, weight, bias, offset, mask, B, C, H, W, Co, kh, kw, sh, sw, ph, pw, dh, dw, dg, out, 1);
}

/* dcn_v2_cuda_backward, dcn_v2_cuda.cu:206-335 (per-sample loop :259-330).  All grads are OVERWRITTEN
 * (the reference starts from zeros_like, :251-255).                                                     */
int c2m_oracle_dcn_v2_backward(const float* in, const float* weight, const float* bias, const float* offset,
                               const float* mask, const float* grad_out, int B, int C, int H, int W, int Co, int kh,
                               int kw, int sh, int sw, int ph, int pw, int dh, int dw, int dg, float* grad_in,
                               float* grad_offset, float* grad_mask, float* grad_weight, float* grad_bias) {
  dcn_geom g;
  (void)bias;
  if (!in || !weight || !offset || !mask || !grad_out || !grad_in || !grad_offset || !grad_mask || !grad_weight ||
      !grad_bias)
    return C2M_EINVAL;
  int rc = dcn_geom_init(&g, B, C, H, W, Co, kh, kw, sh, sw, ph, pw, dh, dw, dg);
  if (rc) return rc;
  const int K = kh * kw, HWo = g.Ho * g.Wo, CK = C * K, cpg = C / dg, HW = H * W;
  float* col = (float*)malloc(sizeof(float) * (size_t)CK * HWo);
  if (!col) return C2M_ENOMEM;
  memset(grad_in, 0, sizeof(float) * (size_t)B * C * HW);
  memset(grad_weight, 0, sizeof(float) * (size_t)Co * CK);
  memset(grad_bias, 0, sizeof(float) * (size_t)Co);

  for (int b = 0; b < B; ++b) {
    const float* inb = in + (size_t)b * C * HW;
    const float* offb = offset + (size_t)b * dg * 2 * K * HWo;
    const float* mskb = mask + (size_t)b * dg * K * HWo;
    const float* gob = grad_out + (size_t)b * Co * HWo;
    float* gib = grad_in + (size_t)b * C * HW;
    float* gofb = grad_offset + (size_t)b * dg * 2 * K * HWo;
    float* gmb = grad_mask + (size_t)b * dg * K * HWo;

    /* dCol = W^T . gO[b]  (Sgemm 'n','t', dcn_v2_cuda.cu:273-276) */
#pragma omp parallel for schedule(static)
    for (int k = 0; k < CK; ++k) {
      float* cr = col + (size_t)k * HWo;
      for (int p = 0; p < HWo; ++p) cr[p] = 0.0f;
      for (int o = 0; o < Co; ++o) {
        const float wv = weight[(size_t)o * CK + k];
        const float* gr = gob + (size_t)o * HWo;
#pragma omp simd
        for (int p = 0; p < HWo; ++p) cr[p] = fmaf(wv, gr[p], cr[p]);
      }
    }

    /* col2im_coord: grad_offset / grad_mask, dcn_v2_im2col_cuda.cu:256-327 */
#pragma omp parallel for schedule(static)
    for (int oc = 0; oc < dg * 2 * K; ++oc) {
      const int grp = oc / (2 * K), offc = oc % (2 * K), t = offc / 2, dir = offc % 2;
      const int i = t / kw, j = t % kw;
      for (int y = 0; y < g.Ho; ++y)
        for (int x = 0; x < g.Wo; ++x) {
          const int p = y * g.Wo + x;
          const float oh = offb[((size_t)grp * 2 * K + 2 * t) * HWo + p];
          const float ow = offb[((size_t)grp * 2 * K + 2 * t + 1) * HWo + p];
          const float mk = mskb[((size_t)grp * K + t) * HWo + p];
          float inv_h = (y * sh - ph) + i * dh + oh, inv_w = (x * sw - pw) + j * dw + ow;
          const int outside = (inv_h <= -1 || inv_w <= -1 || inv_h >= H || inv_w >= W);
          if (outside) inv_h = inv_w = -2;
          float val = 0, mval = 0;
          for (int cc = 0; cc < cpg; ++cc) {
            const int c = grp * cpg + cc;
            const float dc = col[((size_t)c * K + t) * HWo + p];
            const float* im = inb + (size_t)c * HW;
            if (!outside) mval += dc * bilinear(im, H, W, inv_h, inv_w);
            const float wgt = coord_weight(inv_h, inv_w, H, W, im, dir);
            val += wgt * dc * mk;
          }
          gofb[(size_t)oc * HWo + p] = val;
          if (dir == 0) gmb[((size_t)grp * K + t) * HWo + p] = mval;
        }
    }

    /* col2im: grad_input scatter, dcn_v2_im2col_cuda.cu:197-254 (deterministic order here; atomics there) */
#pragma omp parallel for schedule(static)
    for (int c = 0; c < C; ++c) {
      const int grp = c / cpg;
      float* gi = gib + (size_t)c * HW;
      for (int t = 0; t < K; ++t) {
        const int i = t / kw, j = t % kw;
        for (int y = 0; y < g.Ho; ++y)
          for (int x = 0; x < g.Wo; ++x) {
            const int p = y * g.Wo + x;
            const float oh = offb[((size_t)grp * 2 * K + 2 * t) * HWo + p];
            const float ow = offb[((size_t)grp * 2 * K + 2 * t + 1) * HWo + p];
            const float mk = mskb[((size_t)grp * K + t) * HWo + p];
            const float ch = (y * sh - ph) + i * dh + oh, cw = (x * sw - pw) + j * dw + ow;
            const float top = col[((size_t)c * K + t) * HWo + p] * mk;
            const int ih = (int)ch, iw = (int)cw;
            for (int dy = -2; dy <= 2; ++dy)
              for (int dx = -2; dx <= 2; ++dx) {
                const int yy = ih + dy, xx = iw + dx;
                if (yy >= 0 && yy < H && xx >= 0 && xx < W && fabsf(ch - yy) < 1 && fabsf(cw - xx) < 1)
                  gi[yy * W + xx] += scatter_weight(ch, cw, yy, xx, H, W) * top;
              }
          }
      }
    }

    /* re-im2col, then grad_weight += gO[b] . col^T, grad_bias += rowsum(gO[b])  (dcn_v2_cuda.cu:302-329) */
    im2col_sample(&g, inb, offb, mskb, col);
#pragma omp parallel for schedule(static)
    for (int o = 0; o < Co; ++o) {
      const float* gr = gob + (size_t)o * HWo;
      for (int k = 0; k < CK; ++k) {
        const float* cr = col + (size_t)k * HWo;
        float a = 0.0f;
        for (int p = 0; p < HWo; ++p) a = fmaf(gr[p], cr[p], a);
        grad_weight[(size_t)o * CK + k] += a;
      }
      float s = 0.0f;
      for (int p = 0; p < HWo; ++p) s += gr[p];
      grad_bias[o] += s;
    }
  }
  free(col);
  return C2M_OK;
}
