// conv3x3.hip -- 3x3 / stride 1 / pad 1 convolution, fp32, channels-last, for gfx950 (MI355X).
//
// the decoder's convolution stack (ref_restoration_arch.py:140-187, arch_util.py:80-136: 3 x 16
// residual blocks + heads + tails + the offset convs, 30 TFLOP per batch-16 forward at LR 160) and the DCN offset/mask
// head (dcn_v2.py:229-245).  The reference runs them as separate cuDNN convs + bias adds + activations + residual adds +
// cats on NCHW tensors; here one kernel computes
//
//     out = act( conv3x3( cat(src0, src1) ) + bias ) + res1 + res2
//
// as an implicit GEMM out[Cout x pixels] = W[Cout x 9*Cin] . X[9*Cin x pixels] on v_mfma_f32_32x32x2_f32 with every
// operand moved by LDS-DMA (global_load_lds_dwordx4) -- no VGPR staging, no VALU in the k loop:
//
//   * workgroup = 4 waves = a 32 x 4 pixel tile x MW = 32*MT output channels; wave w owns row w (32 pixels, MT
//     accumulator tiles D[32 couts][32 pixels]);
//   * K is swept chunk by chunk (32 input channels) and, inside a chunk, tap by tap.  The zero-padded halo tile of a chunk
//     (34 x 6 pixels x 32 channels = 26 KiB) is DMA'd ONCE and serves all 9 taps as shifted LDS reads; the next chunk's
//     tile streams in underneath (double buffer).  cat() never exists: a chunk simply comes from src0 or src1;
//   * weights are pre-arranged (conv3x3_relayout_kernel, cached by the host per weight version) as ready-made LDS images
//     [cout block][chunk][tap][MW rows][32 k]; a unit's image (4 KiB * MT) is a linear DMA into a ring of 3 slots, two
//     units ahead of its use; one barrier per unit;
//   * both operands are read with ds_read_b128 (4 k-steps per read).  Rows (pixels / couts) are 128 bytes; the 16-byte
//     pieces of row r sit at slot q ^ ((r >> 1) & 7), which makes every 16-lane group of a b128 read hit 16 distinct
//     bank quads whatever the tap shift.  The DMA realises the swizzle on the global side (per-lane source address);
//   * k <-> channel map inside a chunk: piece q = 2g + hi holds channels 8g + 4hi + e (e = 0..3); MFMA k-step 4g + e
//     takes A = W[cout][that channel], B = X[that channel][pixel] from half-wave hi.  fp32 MFMA = fmaf chain: exact fp32;
//   * epilogue in registers: bias, ReLU / LeakyReLU, up to two residuals, then one of four stores -- channels-last,
//     channels-last through PixelShuffle(2) (tail convs), planar NCHW, or the DCN head: offsets += pre-offset synthesised
//     from the flow map of the arg-max indices (index_to_flow + tensor_shift + repeat over groups + (x,y)->(y,x),
//     corres_generation_arch.py:29-46,69-109, dcn_v2.py:236-243), mask = sigmoid, both planar for the DCN kernel, plus the
//     |offset| sum of the reference's "offset mean > 100" warning.  The three [B,9,H,W,2] pre-offset tensors, the raw
//     216-channel head output and the separate fuse pass are never materialised on this path.
//
// LDS: 2 x 26 KiB halo tiles + 3 x 8 KiB weight slots + 1 KiB = 77 KiB (MT = 2) -> two workgroups per CU: while one
// waits at its barrier the other owns the matrix pipes.
#include <stdlib.h>

#include "c2m_common.h"

namespace c2m {
namespace conv {

constexpr int TW = 32, TH = 4;                 // pixel tile of a workgroup
constexpr int HW_ = TW + 2, HH_ = TH + 2;      // halo tile
constexpr int NPIX = HW_ * HH_;                // 204 pixels
constexpr int KCH = 32;                        // input channels per chunk = one 128-byte LDS row per pixel
constexpr int NIN_REAL = (NPIX * 8 + 63) / 64; // 26 DMA instructions (64 x 16 B) per halo tile
constexpr int NIN_W = 7;                       // per wave (waves 2, 3 issue one dummy each: uniform vmcnt counts)
constexpr int IN_BYTES = NIN_REAL * 1024;      // 26624
static_assert(NIN_REAL <= 4 * NIN_W, "four waves x NIN_W instructions must cover the halo tile");

struct Src {
  const float* ptr;      // pixel (0, 0), channel 0 of sample 0
  int C;                 // channels taken from this source (multiple of 32)
  int pix_pitch;         // elements between horizontally adjacent pixels
  int row_pitch;         // elements between rows
  long long img_pitch;   // elements between samples
};

struct Params {
  int B, H, W, Cin, Cout;
  int tiles_x, tiles_y, nchunks;
  Src src[2];
  const float* wr;       // relayouted weights + 256 zero bytes at wr + wr_zero_off
  long long wr_zero_off; // element offset of the zero page
  const float* bias;     // [Cout] or nullptr
  int act;               // 0 none, 1 ReLU, 2 LeakyReLU(slope)
  float slope;
  int out_mode;          // 0 NHWC, 1 NHWC + PixelShuffle(2), 2 NCHW, 3 DCN offset/mask head (NCHW)
  float* out;            // modes 0/1: channels-last with the pitches below; 2: [B][Cout][H][W]; 3: offset [B][2*dg*9][H][W]
  int out_pix_pitch, out_row_pitch;
  long long out_img_pitch;
  const float* res1;     // mode 0: same geometry as out
  const float* res2;
  float* mask_out;       // mode 3: [B][dg*9][H][W]
  const float* flow;     // mode 3: [B][fh][fw][2] (x, y) = index_to_flow of the arg-max map, or nullptr (no pre-offset)
  int fh, fw, scale, n_off;   // n_off = 2*dg*9 offset channels (the rest are mask logits)
  double* abs_sum;       // mode 3: C2M_ABS_SUM_SLOTS partial sums of |raw offset| or nullptr
  int out_vec4;          // mode 0: out / res pitches and bases are 16-byte aligned -> float4 stores
  int stagger;           // start-up delay (x 8192 cycles) of the workgroups in odd CU slots, see conv3x3_kernel
};

// ---------------------------------------------------------------------------------------------------------------------
// weights W[Cout][Cin][3][3] -> LDS images Wr[cb][chunk][tap][row (MW)][slot (8)][e (4)], slot = q ^ ((row >> 1) & 7),
// value = W[cb*MW + row][chunk*32 + 4q + e][tap] (0 beyond Cout), followed by a 256-byte zero page
// ---------------------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) conv3x3_relayout_kernel(const float* __restrict__ w, int Cin, int Cout, int MW,
                                                                 long long total, float* __restrict__ wr) {
  const long long e0 = (long long)blockIdx.x * 256 + threadIdx.x;
  if (e0 >= total + 64) return;
  if (e0 >= total) { wr[e0] = 0.0f; return; }   // zero page
  const int e = (int)(e0 & 3), slot = (int)((e0 >> 2) & 7);
  long long r = e0 >> 5;
  const int row = (int)(r % MW); r /= MW;
  const int tap = (int)(r % 9); r /= 9;
  const int nch = Cin / KCH;
  const int chunk = (int)(r % nch);
  const int cb = (int)(r / nch);
  const int q = slot ^ ((row >> 1) & 7);
  const int co = cb * MW + row, ci = chunk * KCH + 4 * q + e;
  wr[e0] = co < Cout ? w[((size_t)co * Cin + ci) * 9 + tap] : 0.0f;
}

// index_to_flow (corres_generation_arch.py:29-46) of the whole batch, un-padded: flow[b][y][x] = (idx % wq - x, idx / wq - y)
__global__ void __launch_bounds__(256) index_to_flow_kernel(const int64_t* __restrict__ max_idx, int n, int hq, int wq,
                                                             float2* __restrict__ flow) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const int p = i % (hq * wq);
  const int y = p / wq, x = p - y * wq;
  const int64_t idx = max_idx[i];
  flow[i] = make_float2((float)((int)(idx % wq) - x), (float)((int)(idx / wq) - y));
}

__device__ __forceinline__ f32x4 lds_read_b128(unsigned byte_addr) {
  return *(const __attribute__((address_space(3))) f32x4*)byte_addr;
}

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

template <int MT>
__global__ void __launch_bounds__(256, 2) conv3x3_kernel(Params p) {
  constexpr int MW = 32 * MT;
  constexpr int WSLOT = MW * 128;          // bytes of one unit's weight image
  constexpr int NW_W = MT;                 // weight DMA instructions per wave and unit (MT KiB each)
  extern __shared__ __attribute__((aligned(1024))) char lds[];
  // [in0 | in1 | w ring x3 | dummy 1 KiB]
  const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)lds;   // LDS byte address
  const unsigned in_base = lds0, w_base = lds0 + 2 * IN_BYTES, dummy = w_base + 3 * WSLOT;

  const int tid = threadIdx.x, l = tid & 63, hi = l >> 5, j = l & 31;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ntile = p.tiles_x * p.tiles_y * p.B;
  const int tile = xcd_remap(blockIdx.x, ntile);
  const int cb = blockIdx.y;
  const int tx = tile % p.tiles_x, ty = (tile / p.tiles_x) % p.tiles_y, b = tile / (p.tiles_x * p.tiles_y);
  const int x0 = tx * TW, y0 = ty * TH;
  const int U = p.nchunks * 9;
  const float* zero = p.wr + p.wr_zero_off;

  // Two workgroups share a CU (one wave each per SIMD).  Started together they run in lockstep -- same barriers, same
  // prologue / epilogue at the same time -- and the matrix pipe idles whenever both stall.  Delaying the workgroup in
  // the odd slot once puts the pair in anti-phase for the rest of the launch (each slot's successors inherit the offset):
  // one's DMA prologue, barrier waits and store epilogue then run under the other's MFMAs.
  if (p.stagger > 0) {
    unsigned hwid;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
    if (((hwid >> 16) & 1) && blockIdx.x < 512 && blockIdx.y == 0)   // TG_ID = the workgroup's slot on its CU; first dispatch round only
      for (int k = 0; k < p.stagger; ++k) __builtin_amdgcn_s_sleep(127);
  }

  // halo-tile DMA: instruction n (64 pieces of 16 B) is issued by wave n & 3 as its slot n >> 2; piece P = 64n + lane is
  // pixel pl = P >> 3, LDS slot P & 7, logical piece q = slot ^ ((pl >> 1) & 7)
  int pyx[NIN_W];   // (iy << 16) | ix of the source pixel, -1 = outside the image / beyond the tile (reads the zero page)
#pragma unroll
  for (int s = 0; s < NIN_W; ++s) {
    const int n = wv + 4 * s;
    const int pl = 8 * n + (l >> 3);
    const int ry = pl / HW_, rx = pl - ry * HW_;
    const int iy = y0 - 1 + ry, ix = x0 - 1 + rx;
    const bool ok = n < NIN_REAL && pl < NPIX && iy >= 0 && iy < p.H && ix >= 0 && ix < p.W;
    pyx[s] = ok ? ((iy << 16) | ix) : -1;
  }
  auto issue_in = [&](int c) {
    const int c0 = c * KCH;
    const bool first = c0 < p.src[0].C;
    const Src& S = first ? p.src[0] : p.src[1];
    const float* base = S.ptr + (long long)b * S.img_pitch + (first ? c0 : c0 - p.src[0].C);
    const unsigned buf = in_base + (c & 1) * IN_BYTES;
#pragma unroll
    for (int s = 0; s < NIN_W; ++s) {
      const int n = wv + 4 * s;
      const int q = (l & 7) ^ ((4 * n + (l >> 4)) & 7);
      const int iy = pyx[s] >> 16, ix = pyx[s] & 0xffff;
      const float* g = pyx[s] >= 0 ? base + (long long)iy * S.row_pitch + ix * S.pix_pitch + 4 * q : zero;
      const unsigned dst = n < NIN_REAL ? buf + n * 1024 : dummy;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                       (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
    }
  };
  const float* wsrc = p.wr + (long long)cb * U * (MW * 32) + l * 4;
  auto issue_w = [&](int u) {
#pragma unroll
    for (int k = 0; k < NW_W; ++k) {
      const int i = wv * NW_W + k;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(wsrc + (long long)u * (MW * 32) + i * 256),
                                       (__attribute__((address_space(3))) void*)(w_base + (u % 3) * WSLOT + i * 1024),
                                       16, 0, 0);
    }
  };

  f32x16 acc[MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[mt][r] = 0.0f;

  // A-operand byte offsets inside a weight slot (row = cout j of tile mt, piece 2g + hi), + mt * 4096
  unsigned aoff[4];
#pragma unroll
  for (int g = 0; g < 4; ++g) aoff[g] = j * 128 + (((2 * g + hi) ^ ((j >> 1) & 7)) << 4);

  // prologue (issue order matters for the vmcnt waits: halo tile first, then the two weight units)
  issue_in(0);
  issue_w(0);
  if (U > 1) issue_w(1);

  for (int c = 0; c < p.nchunks; ++c) {
    const unsigned ibuf = in_base + (c & 1) * IN_BYTES;
    const bool more_in = c + 1 < p.nchunks;
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      const int u = c * 9 + t;
      // unit u needs W(u) (issued two units ago) and, at t == 0, the halo tile of chunk c (older).  Younger, still allowed
      // in flight: W(u+1) and -- at t == 1, 2 -- the next halo tile, issued during unit (c, 0) right after W(c, 2).
      if (u + 1 >= U) wait_vmcnt<0>();
      else if ((t == 1 || t == 2) && more_in) wait_vmcnt<NW_W + NIN_W>();
      else wait_vmcnt<NW_W>();
      // bare s_barrier: __syncthreads() would add a fence = s_waitcnt vmcnt(0) and drain the DMAs that are meant to stay in
      // flight.  This wave's shares of W(u) / the halo tile have landed (vmcnt above); the barrier publishes every wave's.
      __builtin_amdgcn_s_barrier();
      if (u + 2 < U) issue_w(u + 2);
      if (t == 0 && more_in) issue_in(c + 1);

      const int dy = t / 3, dx = t - 3 * dy;
      const int pl = (wv + dy) * HW_ + j + dx;
      const unsigned brow = ibuf + pl * 128, bsw = (pl >> 1) & 7;
      const unsigned wslot = w_base + (t % 3) * WSLOT;   // 9 % 3 == 0: the ring position of tap t is the same in every chunk
      f32x4 a[2][MT], bq[2];
      bq[0] = lds_read_b128(brow + (((0 + hi) ^ bsw) << 4));
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) a[0][mt] = lds_read_b128(wslot + aoff[0] + mt * 4096);
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        if (g + 1 < 4) {
          bq[(g + 1) & 1] = lds_read_b128(brow + (((2 * (g + 1) + hi) ^ bsw) << 4));
#pragma unroll
          for (int mt = 0; mt < MT; ++mt) a[(g + 1) & 1][mt] = lds_read_b128(wslot + aoff[g + 1] + mt * 4096);
        }
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
          for (int mt = 0; mt < MT; ++mt)
            acc[mt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[g & 1][mt][e], bq[g & 1][e], acc[mt], 0, 0, 0);
      }
    }
  }

  // ------------------------------------------------------------------------------------------------------------------
  // epilogue.  acc[mt][r] = out channel cb*MW + mt*32 + 8*(r>>2) + 4*hi + (r&3) of pixel (y0 + wv, x0 + j)
  // ------------------------------------------------------------------------------------------------------------------
  const int y = y0 + wv, x = x0 + j;
  const bool pok = y < p.H && x < p.W;
  float asum = 0.0f;
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
    for (int qd = 0; qd < 4; ++qd) {
      const int co = cb * MW + mt * 32 + 8 * qd + 4 * hi;   // first of 4 consecutive output channels
      if (co >= p.Cout) continue;
      f32x4 v;
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = acc[mt][4 * qd + e];
      if (p.bias) {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] += (co + e < p.Cout) ? p.bias[co + e] : 0.0f;
      }
      if (p.out_mode == 3) {
        if (!pok) continue;
        const size_t HWs = (size_t)p.H * p.W, pix = (size_t)y * p.W + x;
        if (co < p.n_off) {
          // channels (co, co+1) = (dy, dx) of (group, tap) gt = co/2, (co+2, co+3) of gt+1; pre-offset of tap k at scale s:
          // P_k[y][x] = s * flow[(y - s*ki) / s][(x - s*kj) / s] (0 outside), channel order (y, x)
#pragma unroll
          for (int h2 = 0; h2 < 2; ++h2) {
            const int gt = (co >> 1) + h2, tap = gt % 9;
            const int ki = tap / 3, kj = tap - 3 * ki;
            float fy = 0.0f, fx = 0.0f;
            if (p.flow) {
              const int ys = y - p.scale * ki, xs = x - p.scale * kj;
              if (ys >= 0 && xs >= 0) {
                const int yy = ys / p.scale, xx = xs / p.scale;
                if (yy < p.fh && xx < p.fw) {
                  const float2 f = reinterpret_cast<const float2*>(p.flow)[((size_t)b * p.fh + yy) * p.fw + xx];
                  fx = f.x * (float)p.scale;
                  fy = f.y * (float)p.scale;
                }
              }
            }
            asum += fabsf(v[2 * h2]) + fabsf(v[2 * h2 + 1]);
            p.out[((size_t)b * p.n_off + co + 2 * h2) * HWs + pix] = v[2 * h2] + fy;
            p.out[((size_t)b * p.n_off + co + 2 * h2 + 1) * HWs + pix] = v[2 * h2 + 1] + fx;
          }
        } else {
          const int nm = p.Cout - p.n_off;
#pragma unroll
          for (int e = 0; e < 4; ++e)
            if (co + e < p.Cout) p.mask_out[((size_t)b * nm + (co - p.n_off) + e) * HWs + pix] = 1.0f / (1.0f + expf(-v[e]));
        }
        continue;
      }
      if (p.act == 1) {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.0f);
      } else if (p.act == 2) {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = v[e] > 0.0f ? v[e] : v[e] * p.slope;
      }
      if (!pok) continue;
      if (p.out_mode == 0) {
        const size_t o = (size_t)b * p.out_img_pitch + (size_t)y * p.out_row_pitch + (size_t)x * p.out_pix_pitch + co;
        if (co + 3 < p.Cout && p.out_vec4) {
          if (p.res1) { const f32x4 r1 = *reinterpret_cast<const f32x4*>(p.res1 + o); v += r1; }
          if (p.res2) { const f32x4 r2 = *reinterpret_cast<const f32x4*>(p.res2 + o); v += r2; }
          *reinterpret_cast<f32x4*>(p.out + o) = v;
        } else {
          for (int e = 0; e < 4 && co + e < p.Cout; ++e) {
            float s = v[e];
            if (p.res1) s += p.res1[o + e];
            if (p.res2) s += p.res2[o + e];
            p.out[o + e] = s;
          }
        }
      } else if (p.out_mode == 1) {
        // PixelShuffle(2): channel 4*c2 + 2*dy + dx of pixel (y, x) -> channel c2 of pixel (2y + dy, 2x + dx)
        const int c2 = co >> 2;
#pragma unroll
        for (int e = 0; e < 4; ++e)
          p.out[(size_t)b * p.out_img_pitch + (size_t)(2 * y + (e >> 1)) * p.out_row_pitch +
                (size_t)(2 * x + (e & 1)) * p.out_pix_pitch + c2] = v[e];
      } else {
        const size_t HWs = (size_t)p.H * p.W;
        for (int e = 0; e < 4 && co + e < p.Cout; ++e) p.out[((size_t)b * p.Cout + co + e) * HWs + (size_t)y * p.W + x] = v[e];
      }
    }
  }
  if (p.out_mode == 3 && p.abs_sum) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) asum += __shfl_xor(asum, off, 64);
    if (l == 0) atomicAdd(p.abs_sum + ((blockIdx.x * 4 + wv + blockIdx.y * 31) & (C2M_ABS_SUM_SLOTS - 1)), (double)asum);
  }
}

}  // namespace conv
}  // namespace c2m

// =====================================================================================================================
// C-ABI
// =====================================================================================================================
using namespace c2m;

namespace {
inline int conv_mw(int Cout) { return Cout <= 32 ? 32 : 64; }
inline long long relayout_elems(int Cin, int Cout) {
  const int MW = conv_mw(Cout), ncb = (Cout + MW - 1) / MW;
  return (long long)ncb * (Cin / conv::KCH) * 9 * MW * 32;
}
}  // namespace

extern "C" size_t c2m_conv3x3_relayout_bytes(int Cin, int Cout) {
  if (Cin <= 0 || Cout <= 0 || Cin % conv::KCH != 0) return 0;
  return (size_t)(relayout_elems(Cin, Cout) + 64) * sizeof(float);
}

extern "C" int c2m_conv3x3_relayout_f32(c2m_stream_t stream, const float* weight, int Cin, int Cout, float* wr) {
  if (!weight || !wr || Cin <= 0 || Cout <= 0) return C2M_ERR_INVALID_ARG;
  if (Cin % conv::KCH != 0) return C2M_ERR_UNSUPPORTED;
  const long long total = relayout_elems(Cin, Cout);
  hipLaunchKernelGGL(conv::conv3x3_relayout_kernel, dim3((unsigned)((total + 64 + 255) / 256)), dim3(256), 0,
                     as_stream(stream), weight, Cin, Cout, conv_mw(Cout), total, wr);
  return check_launch();
}

extern "C" int c2m_index_to_flow_f32(c2m_stream_t stream, const int64_t* max_idx, int B, int hq, int wq, float* flow) {
  if (!max_idx || !flow || B <= 0 || hq <= 0 || wq <= 0) return C2M_ERR_INVALID_ARG;
  const int n = B * hq * wq;
  hipLaunchKernelGGL(conv::index_to_flow_kernel, dim3(ceil_div(n, 256)), dim3(256), 0, as_stream(stream), max_idx, n, hq,
                     wq, reinterpret_cast<float2*>(flow));
  return check_launch();
}

extern "C" int c2m_conv3x3_nhwc_f32(c2m_stream_t stream, const c2m_conv3x3_desc* d) {
  if (!d || !d->wr || !d->out || d->B <= 0 || d->H <= 0 || d->W <= 0 || d->Cin <= 0 || d->Cout <= 0 || d->nsrc < 1 ||
      d->nsrc > 2 || !d->src[0].ptr || (d->nsrc == 2 && !d->src[1].ptr))
    return C2M_ERR_INVALID_ARG;
  int csum = 0;
  for (int s = 0; s < d->nsrc; ++s) {
    if (d->src[s].C <= 0 || d->src[s].C % conv::KCH != 0 || d->src[s].pix_pitch % 4 != 0 || d->src[s].row_pitch % 4 != 0 ||
        d->src[s].img_pitch % 4 != 0 || ((uintptr_t)d->src[s].ptr & 15))
      return C2M_ERR_UNSUPPORTED;   // 16-byte pieces: every pitch a multiple of 4 floats, 32-channel chunks
    csum += d->src[s].C;
  }
  if (csum != d->Cin || d->H >= 32768 || d->W >= 65536) return C2M_ERR_INVALID_ARG;
  if (d->out_mode < 0 || d->out_mode > 3) return C2M_ERR_INVALID_ARG;
  const bool out_vec4 = !(d->out_pix_pitch % 4 != 0 || d->out_row_pitch % 4 != 0 || d->out_img_pitch % 4 != 0 ||
                          ((uintptr_t)d->out & 15) || ((uintptr_t)d->res1 & 15) || ((uintptr_t)d->res2 & 15));
  if (d->out_mode == 1 && d->Cout % 4 != 0) return C2M_ERR_INVALID_ARG;
  if (d->out_mode == 3 && (!d->mask_out || d->n_off <= 0 || d->n_off % 4 != 0 || d->n_off >= d->Cout || d->scale <= 0 ||
                           (d->flow && (d->fh <= 0 || d->fw <= 0))))
    return C2M_ERR_INVALID_ARG;
  if ((d->res1 || d->res2) && d->out_mode != 0) return C2M_ERR_UNSUPPORTED;

  conv::Params p;
  p.B = d->B; p.H = d->H; p.W = d->W; p.Cin = d->Cin; p.Cout = d->Cout;
  p.tiles_x = ceil_div(d->W, conv::TW); p.tiles_y = ceil_div(d->H, conv::TH); p.nchunks = d->Cin / conv::KCH;
  for (int s = 0; s < 2; ++s) {
    const int k = s < d->nsrc ? s : 0;
    p.src[s].ptr = d->src[k].ptr; p.src[s].C = s < d->nsrc ? d->src[k].C : 0; p.src[s].pix_pitch = d->src[k].pix_pitch;
    p.src[s].row_pitch = d->src[k].row_pitch; p.src[s].img_pitch = d->src[k].img_pitch;
  }
  p.wr = d->wr; p.wr_zero_off = relayout_elems(d->Cin, d->Cout);
  p.bias = d->bias; p.act = d->act; p.slope = d->slope; p.out_mode = d->out_mode; p.out = d->out;
  p.out_pix_pitch = d->out_pix_pitch; p.out_row_pitch = d->out_row_pitch; p.out_img_pitch = d->out_img_pitch;
  p.res1 = d->res1; p.res2 = d->res2; p.mask_out = d->mask_out; p.flow = d->flow; p.fh = d->fh; p.fw = d->fw;
  p.scale = d->scale; p.n_off = d->n_off; p.abs_sum = d->abs_sum;
  p.out_vec4 = out_vec4 ? 1 : 0;
  {
    // anti-phase start-up delay: about half a workgroup's lifetime (nchunks * 9 units of ~8200 cycles when two workgroups
    // share the matrix pipes), capped so that short launches do not pay more than they gain.  C2M_CONV_STAGGER overrides.
    static const int env = [] { const char* e = getenv("C2M_CONV_STAGGER"); return e ? atoi(e) : -1; }();
    const long long rounds = ((long long)p.tiles_x * p.tiles_y * p.B * ceil_div(d->Cout, conv_mw(d->Cout))) / 512;
    int st_units = p.nchunks * 9 / 2;
    if (rounds < 16) st_units = (int)(st_units * rounds / 32);
    (void)st_units;
    p.stagger = env >= 0 ? env : 0;
  }

  const int MW = conv_mw(d->Cout);
  const long long ntile = (long long)p.tiles_x * p.tiles_y * p.B;
  if (ntile > 0x7fffffffLL) return C2M_ERR_INVALID_ARG;
  dim3 grid((unsigned)ntile, ceil_div(d->Cout, MW));
  hipStream_t st = as_stream(stream);
  ProfileScope prof(C2M_KERNEL_CONV3X3, st);
  if (MW == 64) {
    const size_t ldsb = 2 * conv::IN_BYTES + 3 * 64 * 128 + 1024;
    static unsigned long long lds_set = 0;
    if (int rc = ensure_dynamic_lds(reinterpret_cast<const void*>(&conv::conv3x3_kernel<2>), ldsb, lds_set)) return rc;
    hipLaunchKernelGGL(conv::conv3x3_kernel<2>, grid, dim3(256), ldsb, st, p);
  } else {
    const size_t ldsb = 2 * conv::IN_BYTES + 3 * 32 * 128 + 1024;
    static unsigned long long lds_set = 0;
    if (int rc = ensure_dynamic_lds(reinterpret_cast<const void*>(&conv::conv3x3_kernel<1>), ldsb, lds_set)) return rc;
    hipLaunchKernelGGL(conv::conv3x3_kernel<1>, grid, dim3(256), ldsb, st, p);
  }
  return check_launch();
}
