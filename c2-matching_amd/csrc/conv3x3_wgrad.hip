// conv3x3_wgrad.hip -- weight gradient of the 3x3 / stride 1 / pad 1 convolution on channels-last fp32 tensors (gfx950).
//
// Training path of SURVEY.md 8f row 3 (stage-3 step, ref_restoration_model.py:192-269): with the forward (conv3x3_split.hip)
// and the data gradient (the same kernel on rotated / transposed weights) this takes the decoder's convolutions off MIOpen.
//
//     dW[co][ci][dy][dx] = sum over (b, y, x) of  G[b][y][x][co] * X[b][y + dy - 1][x + dx - 1][ci]          (X zero outside)
//
// = a GEMM [Cout x P] . [P x 9 Cin] whose contraction runs over PIXELS, so on channels-last tensors both MFMA operands of a
// k-step (one pixel per half-wave) are 32 consecutive channels of that pixel: v_mfma_f32_32x32x2_f32 (fp32: an exact fmaf
// chain) with A = G[pixel][32 couts], B = X[shifted pixel][32 cins], no transposition anywhere.
//   * workgroup = 3 waves = one (64-cout block, 32-cin block); wave w = kernel row dy = w: 3 taps x 2 cout tiles = 6
//     accumulators.  The contraction is split over pixel segments (32 pixels of one image row): a workgroup walks its slice of
//     the segments, staging G (32 px x 64 co) and the three X rows (34 px x 32 ci, zero outside the image) through LDS with
//     the next segment's global loads in flight; per k-step 2 + 3 ds_read_b32 feed 6 MFMAs.
//   * partial sums go to a workspace [slice][Cout][Cin][9]; wgrad_reduce_kernel adds the slices (deterministic: no atomics).
// X may be the concatenation of two sources (cat(content, ref): ref_restoration_arch.py:147): a 32-channel block comes from
// one of them.
#include <stdio.h>
#include <stdlib.h>

#include "c2m_common.h"
#include "conv3x3_shared.h"

namespace c2m {
namespace conv {
namespace wgrad {

constexpr int SEG = 32;            // pixels per segment
constexpr int GCO = 64, XCI = 32;  // channel blocks
constexpr int G_FLOATS = SEG * GCO;              // 2048
constexpr int X_FLOATS = 3 * (SEG + 2) * XCI;    // 3264
constexpr int NG4 = G_FLOATS / 4, NX4 = X_FLOATS / 4;   // float4 pieces: 512 / 816
constexpr int NT = 192;
constexpr int LG = (NG4 + NT - 1) / NT, LX = (NX4 + NT - 1) / NT;   // per-thread pieces: 3 / 5

struct Params {
  int B, H, W, Cin, Cout;
  Src src[2];
  const float* g;          // grad of the conv output, channels-last
  int g_pix_pitch, g_row_pitch;
  long long g_img_pitch;
  float* partial;          // [nslice][Cout][Cin][9]
  int nslice, nseg, segs_x, per_slice;
};

__global__ void __launch_bounds__(NT) conv3x3_wgrad_kernel(Params p) {
  __shared__ __attribute__((aligned(16))) float lg[2][G_FLOATS];   // [pixel][co]
  __shared__ __attribute__((aligned(16))) float lx[2][X_FLOATS];   // [row dy][pixel 0..33][ci]
  const int tid = threadIdx.x, l = tid & 63, hi = l >> 5, j = l & 31;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);   // kernel row dy
  const int slice = blockIdx.x, cib = blockIdx.y, cob = blockIdx.z;
  const int ci0 = cib * XCI, co0 = cob * GCO;
  // the 32-channel block of X comes from one source
  const bool first = ci0 < p.src[0].C;
  const Src S = first ? p.src[0] : p.src[1];
  const int cs = first ? ci0 : ci0 - p.src[0].C;
  const int seg0 = slice * p.per_slice, seg1 = min(p.nseg, seg0 + p.per_slice);

  f32x4 pg[LG], px[LX];
  auto fetch = [&](int seg) __attribute__((always_inline)) {
    const int sx = seg % p.segs_x, row = seg / p.segs_x;
    const int y = row % p.H, b = row / p.H, x0 = sx * SEG;
#pragma unroll
    for (int k = 0; k < LG; ++k) {
      const int q = tid + k * NT;           // piece: pixel q / 16, channel quad q % 16
      f32x4 v = {0.0f, 0.0f, 0.0f, 0.0f};
      if (q < NG4) {
        const int px_ = q >> 4, c4 = (q & 15) * 4, x = x0 + px_, co = co0 + c4;
        if (x < p.W && co < p.Cout) {
          const float* src = p.g + (size_t)b * p.g_img_pitch + (size_t)y * p.g_row_pitch + (size_t)x * p.g_pix_pitch + co;
          if (co + 3 < p.Cout) v = *reinterpret_cast<const f32x4*>(src);
          else for (int e = 0; e < 4 && co + e < p.Cout; ++e) v[e] = src[e];
        }
      }
      pg[k] = v;
    }
#pragma unroll
    for (int k = 0; k < LX; ++k) {
      const int q = tid + k * NT;           // piece: (row r, pixel i, channel quad): q = (r * 34 + i) * 8 + c4/4
      f32x4 v = {0.0f, 0.0f, 0.0f, 0.0f};
      if (q < NX4) {
        const int c4 = (q & 7) * 4, pi = q >> 3, r = pi / (SEG + 2), i = pi - r * (SEG + 2);
        const int yy = y + r - 1, xx = x0 + i - 1;
        if ((unsigned)yy < (unsigned)p.H && (unsigned)xx < (unsigned)p.W)
          v = *reinterpret_cast<const f32x4*>(S.ptr + (size_t)b * S.img_pitch + (size_t)yy * S.row_pitch + (size_t)xx * S.pix_pitch + cs + c4);
      }
      px[k] = v;
    }
  };
  auto stash = [&](int buf) __attribute__((always_inline)) {
#pragma unroll
    for (int k = 0; k < LG; ++k) {
      const int q = tid + k * NT;
      if (q < NG4) *reinterpret_cast<f32x4*>(&lg[buf][q * 4]) = pg[k];
    }
#pragma unroll
    for (int k = 0; k < LX; ++k) {
      const int q = tid + k * NT;
      if (q < NX4) *reinterpret_cast<f32x4*>(&lx[buf][q * 4]) = px[k];
    }
  };

  f32x16 acc[3][2];
#pragma unroll
  for (int dx = 0; dx < 3; ++dx)
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[dx][mt][r] = 0.0f;

  if (seg0 < seg1) {
    fetch(seg0);
    stash(0);
    __syncthreads();
    for (int seg = seg0, it = 0; seg < seg1; ++seg, ++it) {
      const int buf = it & 1;
      const bool more = seg + 1 < seg1;
      if (more) fetch(seg + 1);   // global loads in flight under this segment's MFMAs
      const float* G = lg[buf];
      const float* X = lx[buf] + wv * (SEG + 2) * XCI;
#pragma unroll 4
      for (int s = 0; s < SEG / 2; ++s) {
        const int pq = 2 * s + hi;
        const float a0 = G[pq * GCO + j], a1 = G[pq * GCO + 32 + j];
#pragma unroll
        for (int dx = 0; dx < 3; ++dx) {
          const float bv = X[(pq + dx) * XCI + j];
          acc[dx][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, bv, acc[dx][0], 0, 0, 0);
          acc[dx][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, bv, acc[dx][1], 0, 0, 0);
        }
      }
      if (more) {
        stash(buf ^ 1);           // (the other buffer: its readers finished before the previous barrier)
        __syncthreads();
      }
    }
  }
  // D[row = co (r & 3) + 8 (r >> 2) + 4 hi][col = ci j] -> partial[slice][co][ci][dy * 3 + dx]
  float* out = p.partial + (size_t)slice * p.Cout * p.Cin * 9;
#pragma unroll
  for (int dx = 0; dx < 3; ++dx)
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int co = co0 + mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi, ci = ci0 + j;
        if (co < p.Cout && ci < p.Cin) out[((size_t)co * p.Cin + ci) * 9 + wv * 3 + dx] = acc[dx][mt][r];
      }
}

__global__ void __launch_bounds__(256) wgrad_reduce_kernel(const float* __restrict__ partial, int nslice, long long n,
                                                            float* __restrict__ gw) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  float s = 0.0f;
  for (int k = 0; k < nslice; ++k) s += partial[(size_t)k * n + i];
  gw[i] = s;
}

}  // namespace wgrad
}  // namespace conv
}  // namespace c2m

using namespace c2m;

namespace {
int wgrad_slices(long long nseg, int Cin, int Cout) {
  // ~768 workgroups per launch (256 CUs x 3 resident) but at least 8 segments each: every slice costs a partial image of
  // Cout x Cin x 9 floats written and read again (at 1024 slices that traffic, not the MFMAs, set the kernel's time)
  const long long blocks = (long long)(Cin / conv::wgrad::XCI) * ((Cout + conv::wgrad::GCO - 1) / conv::wgrad::GCO);
  long long s = (768 + blocks - 1) / blocks;
  s = std::min<long long>(s, (nseg + 7) / 8);
  return (int)std::max<long long>(1, s);
}
}  // namespace

extern "C" size_t c2m_conv3x3_wgrad_workspace_bytes(int B, int H, int W, int Cin, int Cout) {
  if (B <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0) return 0;
  const long long nseg = (long long)B * H * ((W + conv::wgrad::SEG - 1) / conv::wgrad::SEG);
  return (size_t)wgrad_slices(nseg, Cin, Cout) * Cout * Cin * 9 * sizeof(float);
}

extern "C" int c2m_conv3x3_wgrad_f32(c2m_stream_t stream, const c2m_conv_src* src, int nsrc, const float* grad_out,
                                     int g_pix_pitch, int g_row_pitch, long long g_img_pitch, int B, int H, int W, int Cin,
                                     int Cout, float* grad_weight, void* workspace, size_t workspace_bytes) {
  if (!src || nsrc < 1 || nsrc > 2 || !grad_out || !grad_weight || B <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0)
    return C2M_ERR_INVALID_ARG;
  int csum = 0;
  for (int s = 0; s < nsrc; ++s) {
    if (!src[s].ptr || src[s].C <= 0 || src[s].C % conv::wgrad::XCI != 0 || src[s].pix_pitch % 4 != 0 || src[s].row_pitch % 4 != 0 ||
        src[s].img_pitch % 4 != 0 || ((uintptr_t)src[s].ptr & 15))
      return C2M_ERR_UNSUPPORTED;   // 32-channel blocks, 16-byte pieces
    csum += src[s].C;
  }
  if (csum != Cin) return C2M_ERR_INVALID_ARG;
  if (g_pix_pitch % 4 != 0 || g_row_pitch % 4 != 0 || g_img_pitch % 4 != 0 || ((uintptr_t)grad_out & 15)) return C2M_ERR_UNSUPPORTED;
  const size_t need = c2m_conv3x3_wgrad_workspace_bytes(B, H, W, Cin, Cout);
  if (!workspace || workspace_bytes < need) return C2M_ERR_WORKSPACE;
  conv::wgrad::Params p;
  p.B = B; p.H = H; p.W = W; p.Cin = Cin; p.Cout = Cout;
  for (int s = 0; s < 2; ++s) {
    const int k = s < nsrc ? s : 0;
    p.src[s].ptr = src[k].ptr; p.src[s].C = s < nsrc ? src[k].C : 0; p.src[s].pix_pitch = src[k].pix_pitch;
    p.src[s].row_pitch = src[k].row_pitch; p.src[s].img_pitch = src[k].img_pitch;
  }
  p.g = grad_out; p.g_pix_pitch = g_pix_pitch; p.g_row_pitch = g_row_pitch; p.g_img_pitch = g_img_pitch;
  p.partial = static_cast<float*>(workspace);
  p.segs_x = (W + conv::wgrad::SEG - 1) / conv::wgrad::SEG;
  const long long nseg = (long long)B * H * p.segs_x;
  if (nseg > 0x7fffffffLL) return C2M_ERR_INVALID_ARG;
  p.nseg = (int)nseg;
  p.nslice = wgrad_slices(nseg, Cin, Cout);
  p.per_slice = (int)((nseg + p.nslice - 1) / p.nslice);
  hipStream_t st = as_stream(stream);
  {
    ProfileScope prof(C2M_KERNEL_CONV3X3_WGRAD, st);
    hipLaunchKernelGGL(conv::wgrad::conv3x3_wgrad_kernel, dim3(p.nslice, Cin / conv::wgrad::XCI, (Cout + conv::wgrad::GCO - 1) / conv::wgrad::GCO),
                       dim3(conv::wgrad::NT), 0, st, p);
  }
  const long long n = (long long)Cout * Cin * 9;
  hipLaunchKernelGGL(conv::wgrad::wgrad_reduce_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, p.partial, p.nslice, n,
                     grad_weight);
  return check_launch();
}
