// conv3x3_split.hip -- 3x3 / stride 1 / pad 1 convolution, fp32 in / fp32 out, on the 16-BIT matrix pipe of gfx950 (MI355X).
//
// Same contract as conv3x3.hip (decoder stack ref_restoration_arch.py:140-187, arch_util.py:80-136,
// DCN offset/mask head dcn_v2.py:229-245):   out = act( conv3x3( cat(src0, src1) ) + bias ) + res1 + res2
// on channels-last fp32 tensors.  What changes is the arithmetic underneath.  On CDNA4 the fp32 MFMA
// (v_mfma_f32_32x32x2_f32) runs at the fp32 VECTOR rate (157 TF, and every VALU instruction next to it costs matrix
// time); v_mfma_f32_32x32x16_{f16,bf16} is 16x faster and has its own pipe.  Three flavours (template FL, struct Flavour):
//   FL = 3  bf16 x 3: every fp32 operand split EXACTLY into three bf16 pieces
//       x = x0 + x1 + x2,   x0 = top 16 bits of x,  x1 = top 16 bits of (x - x0),  x2 = x - x0 - x1   (8 + 8 + 8 mantissa bits)
//     and the product sum taken over the six piece pairs whose magnitude is >= 2^-16 of the leading one,
//       w.x ~= w0x0 + (w0x1 + w1x0) + (w1x1 + w0x2 + w2x0)            (dropped: w1x2, w2x1, w2x2 <= 2^-24 |w||x|)
//     6/16 of the fp32-MFMA time, full fp32 exponent range: the arithmetic of the autograd path (forward + data gradient);
//   FL = 2  f16 x 2: two round-to-nearest f16 pieces per activation (the low one stored times 2^11), weights scaled per tensor
//     by a power of two and split the same way, THREE products (comment at struct Flavour): the inference default.  Error
//     against float64 below the fp32-MFMA kernel's own (bench.py's conv_arithmetic_check); |x| < 65520, NaN beyond;
//   FL = 1  bf16: one round-to-nearest piece, one product -- the plain bf16 convolution of BASELINE configs[4] (autocast).
// All accumulate in fp32 in the MFMA.
//
// Mapping (one workgroup = 4 waves, TWO workgroups per CU = two waves per SIMD; 32 x 8 output pixels x MW = 32*MT couts):
//   * wave w owns pixel rows 2w, 2w+1 of the tile: NT = 2 pixel tiles x MT channel tiles = 2*MT accumulators f32x16;
//   * K is swept in chunks of 16 input channels (= K of one MFMA).  The zero-padded 34 x 10 halo tile of a chunk arrives as
//     fp32 in REGISTERS (buffer_load_dwordx4 with hardware zero fill outside the image, issued a whole chunk or more
//     ahead; 6 pieces per wave); every wave splits its pieces into NPX 16-bit planes in LDS laid out [plane][k half][pixel]
//     [8 x 16 bit]: a B operand (8 channels of one pixel) is one ds_read_b128, 16 consecutive lanes read 256 contiguous bytes
//     for any tap shift (conflict-free without a swizzle).  FL = 3: one plane buffer, the split is a phase (A) of its own
//     (22 VALU per 4 channels x pixel) -- while one workgroup splits, the co-resident one owns the matrix pipe.  FL = 1, 2:
//     two plane buffers, split round R of the NEXT chunk rides in the MFMA groups of taps 1, 2 of unit R / 2;
//   * three units (kernel rows) of three taps per chunk.  Weights are split once per weight version on the host side of
//     the call (conv3x3_relayout_split_kernel / _multi_kernel) into ready-made LDS images [cout block][chunk][dy][dx][image]
//     [mt][k half][32 rows][8 x 16 bit]; a unit's image (3*NPW*MT KiB) streams by LDS-DMA into a ring of 2 (FL = 3) or 3
//     slots, one or two units ahead, its pieces issued one per MFMA group of a unit's first tap; one barrier per unit;
//   * per tap: NPW*MT A reads + NPX*NT B reads (ds_read_b128, hand-placed: inline asm between sched_barrier fences, one tap
//     ahead into the other of two register sets) feed N*MT*NT MFMAs (24 for bf16 x 3, 12 for f16 x 2);
//   * persistent tiles, XCD-aware tile order, epilogue in registers with the store flavours of conv3x3.hip (channels-last
//     (+ residuals), PixelShuffle(2), planar NCHW, DCN offset/mask head) plus ReLU + MaxPool2d(2,2) (both rows of a
//     pooling window live in one lane, the horizontal neighbour one lane over).
// LDS (MT = 2): bf16 x 3: planes 31.9 KiB + ring 2 x 18 KiB + 1.25 KiB = 69 KiB; f16 x 2: planes 2 x 21.25 + ring 3 x 12 + 1.25 =
// 79.75 KiB; <= 256 registers either way.
// What bounds it: the socket's power cap -- rocm-smi reads 1 400 W of 1 400 and a shader clock of 1.64 GHz (of 2.4)
// while the 64 -> 64 body layer runs on N(0,1) tensors, 1 299 W at 2.39 GHz on zeros (1.58 against 1.13 ms, B = 16, 640^2).  Freed cycles
// become a lower clock; removed WORK (loads, stores, operand reads, MFMAs) becomes time.  History of the structure (v1 .. v5), the
// ablations and the A/B measurements: DESIGN.md 6, DESIGN_HISTORY.md.
#include <stdio.h>
#include <stdlib.h>

#include <algorithm>
#include <type_traits>

#include "c2m_common.h"
#include "conv3x3_shared.h"

namespace c2m {
namespace conv {
namespace split {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

constexpr int KC = 16;                          // input channels per chunk = K of v_mfma_f32_32x32x16_bf16
constexpr int TWX = 32, THY = 8;                // pixel tile of a workgroup
constexpr int HWc = TWX + 2, HHr = THY + 2;     // halo tile 34 x 10
constexpr int NPIX = HWc * HHr;                 // 340
constexpr int NRAW_W = (NPIX * 4 + 255) / 256;  // 6 DMA instructions per wave: 64 pieces of 16 B = (pixel, 4 fp32 channels)
constexpr int NRAW = 4 * NRAW_W;                // 24 slots (22 carry pixels; the rest read zeros) -- every wave runs the
                                                // same branch-free sequence of DMAs and split rounds
constexpr int RAW_BYTES = NRAW * 1024;
constexpr int HALFB = NPIX * 16;                // one (plane, k half) slab: 340 pixels x 16 B (the slots beyond are never stored).
                                                // ds_write_b64 is serviced in groups of 16 consecutive lanes over 32 banks ((a/4) mod 32,
                                                // MI355X_MICROARCH.md): a group of the split's stores covers 4 pixels x 16 B in each k-half
                                                // slab -- == 64 mod 128 puts the two slabs on disjoint banks (with == 0 they collided 2-way:
                                                // SQ_LDS_BANK_CONFLICT was 12 % of the LDS-active cycles)
static_assert(HALFB % 128 == 64 && HALFB % 16 == 0, "bank phase of the second k half");

// Flavour FL of the arithmetic: NPX input planes, NPW weight images, N piece products (weight image W[g] x input plane X[g]).
//   FL = 3  bf16 x 3: three exact bf16 pieces of both operands, six products (full fp32 range)
//   FL = 1  bf16: one round-to-nearest piece, one product (BASELINE configs[4])
//   FL = 2  f16 x 2: x = x0 + 2^-11 x1' + e,  x0 = rne_f16(x), x1' = rne_f16(2^11 (x - x0))   (|e| <= max(2^-22 |x|, 2^-36))
//           and, with the per-tensor power of two S that puts max |w| into [2^14, 2^15):
//           wA = rne_f16(S w), w1 = rne_f16(S w - wA), wB = 2^-11 wA (a third A OPERAND instead of a second accumulator);
//           S w.x ~= wA.x0 + w1.x0 + wB.x1'   (dropped: w1 (x - x0) <= 2^-22 |w||x|);  the epilogue multiplies by 1/S.
//           Three products instead of six.  Domain: |x| < 65520 (beyond: NaN, never a silently wrong number).
template <int FL> struct Flavour;
template <> struct Flavour<1> {
  static constexpr int NPX = 1, NPW = 1, N = 1; static constexpr bool F16 = false;
  static constexpr int W[1] = {0}; static constexpr int X[1] = {0};
};
template <> struct Flavour<3> {
  static constexpr int NPX = 3, NPW = 3, N = 6; static constexpr bool F16 = false;   // smallest terms first, the leading product last
  static constexpr int W[6] = {2, 0, 1, 1, 0, 0};
  static constexpr int X[6] = {0, 2, 1, 0, 1, 0};
};
template <> struct Flavour<2> {
  static constexpr int NPX = 2, NPW = 2, N = 3; static constexpr bool F16 = true;    // images: 0 = wA, 1 = w1; "2" = wB = 2^-11 wA,
  static constexpr int W[3] = {1, 2, 0};                                              // derived in registers (4 v_pk_mul_f16 per operand)
  static constexpr int X[3] = {0, 1, 0};
};
constexpr float F16_LO_SCALE = 2048.0f;   // 2^11: the low piece of an f16 x 2 operand is stored times this
constexpr int npw_of(int fl) { return fl == 1 ? 1 : fl; }
constexpr int npx_of(int fl) { return fl == 2 ? 2 : fl; }

// exact three-way split of four fp32 values into bf16 pairs (truncation: the residuals are exact in fp32)
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void split3(const f32x4 v, u32x2& p0, u32x2& p1, u32x2& p2) {
  const u32x4 m = {0xffff0000u, 0xffff0000u, 0xffff0000u, 0xffff0000u};
  const u32x4 vb = __builtin_bit_cast(u32x4, v);
  const f32x4 r = v - __builtin_bit_cast(f32x4, vb & m);
  const u32x4 rb = __builtin_bit_cast(u32x4, r);
  const f32x4 t = r - __builtin_bit_cast(f32x4, rb & m);
  const u32x4 tb = __builtin_bit_cast(u32x4, t);
  // [hi.b3 hi.b2 lo.b3 lo.b2]: the top halves of two consecutive channels
  p0 = u32x2{__builtin_amdgcn_perm(vb[1], vb[0], 0x07060302u), __builtin_amdgcn_perm(vb[3], vb[2], 0x07060302u)};
  p1 = u32x2{__builtin_amdgcn_perm(rb[1], rb[0], 0x07060302u), __builtin_amdgcn_perm(rb[3], rb[2], 0x07060302u)};
  p2 = u32x2{__builtin_amdgcn_perm(tb[1], tb[0], 0x07060302u), __builtin_amdgcn_perm(tb[3], tb[2], 0x07060302u)};
}

// exact two-way f16 split of four fp32 values: p0 = rne_f16(v), p1 = rne_f16(2^11 (v - p0))
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ void split2_f16(const f32x4 v, u32x2& p0, u32x2& p1) {
  const f16x4 h0 = __builtin_convertvector(v, f16x4);
  const f32x4 r = (v - __builtin_convertvector(h0, f32x4)) * F16_LO_SCALE;
  const f16x4 h1 = __builtin_convertvector(r, f16x4);
  p0 = __builtin_bit_cast(u32x2, h0);
  p1 = __builtin_bit_cast(u32x2, h1);
}

// power of two S with S * wmax in [2^14, 2^15) (1 for wmax = 0 / non-finite)
__device__ __forceinline__ float f16_weight_scale(float wmax) {
  if (!(wmax > 0.0f) || wmax > 3.0e38f) return 1.0f;
  int e;
  (void)frexpf(wmax, &e);          // wmax = m 2^e, m in [0.5, 1)
  e = 15 - e;
  e = e < -100 ? -100 : (e > 100 ? 100 : e);
  return ldexpf(1.0f, e);
}
__device__ __forceinline__ unsigned short f16_bits(float v) { return __builtin_bit_cast(unsigned short, (_Float16)v); }
__device__ __forceinline__ unsigned short f16_piece(float w, int pl, float S) {
  const float v = w * S;                       // exact (power of two)
  const _Float16 a = (_Float16)v;
  if (pl == 0) return __builtin_bit_cast(unsigned short, a);
  return f16_bits(v - (float)a);
}

__device__ __forceinline__ unsigned short bf16_piece(float w, int pl, bool rne) {
  if (rne) {   // NP = 1: round to nearest even
    unsigned u = __builtin_bit_cast(unsigned, w);
    if ((u & 0x7f800000u) == 0x7f800000u) return (unsigned short)(u >> 16);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (unsigned short)(u >> 16);
  }
  float x = w;
  for (int k = 0; k < pl; ++k) x = x - __builtin_bit_cast(float, __builtin_bit_cast(unsigned, x) & 0xffff0000u);
  return (unsigned short)(__builtin_bit_cast(unsigned, x) >> 16);
}

// weights W[Cout][Cin][3][3] -> images [cb][chunk][dy][dx][plane NP][mt MT][k half 2][row 32][e 8] (bf16),
// value = piece `plane` of W[cb*32*MT + mt*32 + row][chunk*16 + 8*half + e][dy][dx] (0 beyond Cout)
// dgrad != 0: images of the DATA-GRADIENT convolution instead (Cin/Cout = its input/output channels = the forward's
// Cout/Cin): value = piece of w[ci][co][2-dy][2-dx] with w the forward weight [Cin here][Cout here][3][3]
// fl = 2: the image is followed by a 256-byte tail of 32-bit slots: [0] float 1/S; uint bits of max |w|: [3] written by
// weight_absmax_kernel (single-tensor path: zeroed + atomicMax), [8 .. 8 + ABSMAX_SPLIT) partial maxima of
// weight_absmax_multi_kernel (each overwritten by its own workgroup: nothing to zero, no atomics)
// fl = 2 | R << 4 (R = 4, 2): f16 x 2 images of the Winograd F(R,3)-along-y kernel (conv3x3_wino16.hip): the kernel-row index dy
// becomes the transform position t = 0 .. R+1, value = piece of U_t[co][ci][dx] = sum_dy G[t][dy] w[co][ci][dy][dx] (float64,
// rounded once), G = 4 x the F(4,3) matrix for R = 4 (the kernel applies B^T / 4); |U| <= 4 (2) max |w| -> S from that bound
__device__ __forceinline__ void relayout_split_elem(const float* __restrict__ w, int Cin, int Cout, int flc, int MT, long long total,
                                                    unsigned short* __restrict__ wr, int dgrad, long long e0, int max_slot, int nslot) {
  const int fl = flc & 15, R = flc >> 4;
  const int NP = npw_of(fl);
  float S = 1.0f;
  if (fl == 2) {
    unsigned mx = 0;
    for (int i = 0; i < nslot; ++i) mx = max(mx, reinterpret_cast<const unsigned*>(wr + total)[max_slot + i]);
    S = f16_weight_scale(__builtin_bit_cast(float, mx) * (R == 4 ? 4.0f : R == 2 ? 2.0f : 1.0f));
    if (e0 == 0) reinterpret_cast<float*>(wr + total)[0] = 1.0f / S;
  }
  if (e0 >= total) return;
  const int NT = R ? R + 2 : 3;
  const int e = (int)(e0 & 7), row = (int)((e0 >> 3) & 31), half = (int)((e0 >> 8) & 1);
  long long r = e0 >> 9;
  const int mt = (int)(r % MT); r /= MT;
  const int pl = (int)(r % NP); r /= NP;
  const int dx = (int)(r % 3); r /= 3;
  const int dy = (int)(r % NT); r /= NT;
  const int nch = Cin / KC;
  const int chunk = (int)(r % nch);
  const int cb = (int)(r / nch);
  const int co = (cb * MT + mt) * 32 + row, ci = chunk * KC + 8 * half + e;
  float v = 0.0f;
  if (co < Cout) {
    if (R) {
      const float* g = w + ((size_t)co * Cin + ci) * 9 + dx;
      const double g0 = g[0], g1 = g[3], g2 = g[6];
      double u;
      if (R == 4) {
        switch (dy) {
          case 0: u = g0; break;
          case 1: u = -(g0 + g1 + g2) * (2.0 / 3.0); break;
          case 2: u = -(g0 - g1 + g2) * (2.0 / 3.0); break;
          case 3: u = g0 * (1.0 / 6.0) + g1 * (1.0 / 3.0) + g2 * (2.0 / 3.0); break;
          case 4: u = g0 * (1.0 / 6.0) - g1 * (1.0 / 3.0) + g2 * (2.0 / 3.0); break;
          default: u = 4.0 * g2; break;
        }
      } else {
        switch (dy) {
          case 0: u = g0; break;
          case 1: u = 0.5 * (g0 + g1 + g2); break;
          case 2: u = 0.5 * (g0 - g1 + g2); break;
          default: u = g2; break;
        }
      }
      v = (float)u;
    } else {
      v = dgrad ? w[((size_t)ci * Cout + co) * 9 + (2 - dy) * 3 + (2 - dx)] : w[((size_t)co * Cin + ci) * 9 + dy * 3 + dx];
    }
  }
  wr[e0] = fl == 2 ? f16_piece(v, pl, S) : bf16_piece(v, pl, fl == 1);
}

__global__ void __launch_bounds__(256) conv3x3_relayout_split_kernel(const float* __restrict__ w, int Cin, int Cout, int fl, int MT,
                                                                      long long total, unsigned short* __restrict__ wr, int dgrad) {
  relayout_split_elem(w, Cin, Cout, fl, MT, total, wr, dgrad, (long long)blockIdx.x * 256 + threadIdx.x, 3, 1);
}

// max |w| as uint bits (monotonic for non-negative floats; NaN -> large) of this thread's share, reduced over its wave
__device__ __forceinline__ unsigned absmax_wave(const float* __restrict__ w, long long n, long long first, long long stride) {
  unsigned m = 0;
  for (long long i = first; i < n; i += stride) m = max(m, __builtin_bit_cast(unsigned, w[i]) & 0x7fffffffu);
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) m = max(m, (unsigned)__shfl_xor((int)m, off, 64));
  return m;
}
__global__ void __launch_bounds__(256) weight_absmax_kernel(const float* __restrict__ w, long long n, unsigned* __restrict__ slot) {
  const unsigned m = absmax_wave(w, n, (long long)blockIdx.x * 256 + threadIdx.x, (long long)gridDim.x * 256);
  if ((threadIdx.x & 63) == 0 && m) atomicMax(slot + 3, m);   // (slot zeroed by the caller)
}

// ---- many tensors per launch (the per-forward refresh of a module's cached images: ~160 weights) ------------------------
// job = 8 x int64: {weight ptr, image ptr, Cin, Cout, fl | MT << 8 | dgrad << 16, image elements, first block, -}
constexpr int ABSMAX_SPLIT = 8;
__global__ void __launch_bounds__(256) weight_absmax_multi_kernel(const long long* __restrict__ jobs) {
  const long long* jb = jobs + (size_t)blockIdx.x * 8;
  if ((int)(jb[4] & 0xf) != 2) return;
  const long long n = jb[2] * jb[3] * 9;
  const unsigned m = absmax_wave(reinterpret_cast<const float*>(jb[0]), n, (long long)blockIdx.y * 256 + threadIdx.x, (long long)ABSMAX_SPLIT * 256);
  __shared__ unsigned part[4];
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0)
    (reinterpret_cast<unsigned*>(reinterpret_cast<unsigned short*>(jb[1]) + jb[5]))[8 + blockIdx.y] = max(max(part[0], part[1]), max(part[2], part[3]));
}
__global__ void __launch_bounds__(256) conv3x3_relayout_split_multi_kernel(const long long* __restrict__ jobs, int njobs) {
  int lo = 0, hi = njobs - 1;            // last job whose first block <= blockIdx.x
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (jobs[(size_t)mid * 8 + 6] <= (long long)blockIdx.x) lo = mid; else hi = mid - 1;
  }
  const long long* jb = jobs + (size_t)lo * 8;
  const int flags = (int)jb[4];
  relayout_split_elem(reinterpret_cast<const float*>(jb[0]), (int)jb[2], (int)jb[3], flags & 0xff, (flags >> 8) & 0xff, jb[5],
                      reinterpret_cast<unsigned short*>(jb[1]), (flags >> 16) & 1, ((long long)blockIdx.x - jb[6]) * 256 + threadIdx.x, 8,
                      ABSMAX_SPLIT);
}

// compile-time loop and LDS instructions with immediate offsets (hand-placed: hipcc re-uses operand registers and then
// issues the next tap's reads behind the last MFMA that reads them -- one exposed LDS latency per tap)
template <int I, int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>());
    static_for<I + 1, N>(f);
  }
}
template <int IMM>
__device__ __forceinline__ void lds_read128(bf16x8& d, unsigned addr) {
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(d) : "v"(addr), "n"(IMM) : "memory");
}
template <int IMM>
__device__ __forceinline__ void lds_read128f(f32x4& d, unsigned addr) {
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(d) : "v"(addr), "n"(IMM) : "memory");
}
typedef int i32x4 __attribute__((ext_vector_type(4)));
// raw buffer descriptor as four SGPR words (stride 0, bounds-checked: out-of-range lanes read zeros)
__device__ __forceinline__ i32x4 make_rsrc_words(const void* base, unsigned bytes) {
  const unsigned long long a = (unsigned long long)(uintptr_t)base;
  i32x4 r;
  r[0] = __builtin_amdgcn_readfirstlane((int)(unsigned)a);
  r[1] = __builtin_amdgcn_readfirstlane((int)((unsigned)(a >> 32) & 0xffffu));
  r[2] = __builtin_amdgcn_readfirstlane((int)bytes);
  r[3] = 0x00020000;
  return r;
}
template <int IMM>
__device__ __forceinline__ void buf_load128(bf16x8& d, unsigned voff, const i32x4 rsrc, int soff) {
  asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen offset:%4" : "=v"(d) : "v"(voff), "s"(rsrc), "s"(soff), "n"(IMM) : "memory");
}
__device__ __forceinline__ void buf_load128f(f32x4& d, unsigned voff, const i32x4 rsrc, int soff) {
  asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen" : "=v"(d) : "v"(voff), "s"(rsrc), "s"(soff) : "memory");
}
template <int IMM>
__device__ __forceinline__ void lds_write64(unsigned addr, const u32x2 v) {
  asm volatile("ds_write_b64 %0, %1 offset:%2" ::"v"(addr), "v"(v), "n"(IMM) : "memory");
}

// ABL > 0: timing-only ablations (WRONG results; $C2M_SPLIT_ABL), a bit mask: 1 no weight DMA after the prologue, 2 no halo
// DMA, 4 no split of the raw tile (only together with 2: loads into dead registers are unsafe), 8 no unit-end waits /
// barriers, 16 no MFMAs, 32 no operand reads
// IO16 (bf16 flavour, channels-last mode, one source): the source tensor holds bf16 -- 16 channels of a pixel are two 16-byte
// pieces that ARE plane entries, so the halo tile goes HBM -> LDS by LDS-DMA (no registers, no split rounds, 3 pieces per wave
// and chunk instead of 6 register loads; zero fill outside the image by the buffer's range check), two chunks ahead into a
// ring of THREE plane buffers.  Output / residual element types follow Params::io_flags (any FL = 1 launch).
template <int FL, int MT, int MODE, int ABL = 0, bool IO16 = false>
__global__ void __launch_bounds__(256, 2) conv3x3_split_kernel(Params p) {
  static_assert(!IO16 || (FL == 1 && MODE == 0 && (ABL & 4) == 0), "bf16 sources: the bf16 flavour's channels-last mode");
  constexpr int NT = 2;
  constexpr int MW = 32 * MT;
  using PR = Flavour<FL>;
  constexpr int NPX = PR::NPX, NPW = PR::NPW;
  // bytes of one plane buffer (IO16: 12 whole wave-pieces of 1 KiB -- the 24 slots behind the 680 pieces of a chunk take zeros)
  constexpr int PLB = IO16 ? 12 * 1024 : NPX * 2 * HALFB;
  // PIPE: two plane buffers -- the split of chunk c+1 is interleaved with the MFMAs of chunk c (no phase (A), no barrier for
  // it).  The bf16 x 3 flavour keeps one buffer and phase (A): two of its workgroups would not fit a CU otherwise.
  constexpr bool PIPE = FL != 3;
  // BF (round 5): the chunk loop of the pipelined fp32-tensor flavours runs WITHOUT uniform branches around its asynchronous
  // issues: past the end of the workgroup's stream the weight pieces land in the dummy page, the halo loads go through a
  // descriptor of zero records (no memory traffic, zeros into dead registers) and the split rounds write a plane buffer nobody
  // reads -- so every unit issues exactly the same vector-memory instructions and its end waits with ONE constant count.
  // (38 branches per 108 MFMAs before; the wino16 counters showed what scalar control flow costs eight waves per CU.)
#ifndef C2M_SPLIT_BF
#define C2M_SPLIT_BF 1
#endif
  constexpr bool BFA = ABL == 0 && C2M_SPLIT_BF != 0;   // any flavour: weight pieces / unit-end waits without branches
  constexpr bool BF = BFA && !IO16;                     // register-loaded halo tiles (fp32 sources)
  constexpr bool BF16S = BFA && IO16;                   // LDS-DMA halo tiles (bf16 sources)
  constexpr int NPB = IO16 ? 3 : (PIPE ? 2 : 1);
  constexpr int NRING = PIPE ? 3 : 2;           // weight ring slots; unit u's weights are issued NRING-1 units ahead
  constexpr int WTAP = NPW * MT * 1024;         // one tap's weight image: [image][mt][half][32 rows][16 B]
  constexpr int WUNIT = 3 * WTAP;               // unit = one kernel row
  constexpr int NWI = WUNIT / 1024;             // LDS-DMA instructions per unit
  constexpr int NW_W = (NWI + 3) / 4;           // per wave (the last wave pads with dummies: uniform vmcnt counts)
  extern __shared__ __attribute__((aligned(1024))) char lds[];
  // [planes (one chunk) | weight ring x2 | DMA dummy 1 KiB | bias MW floats]
  const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)lds;
  const unsigned pl_base = lds0, w_base = lds0 + NPB * PLB, dummy = w_base + NRING * WUNIT, bias_lds = dummy + 1024;

  const int tid = threadIdx.x, l = tid & 63, hi = l >> 5, j = l & 31;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ntile = p.tiles_x * p.tiles_y * p.B;
  const int tile_first = xcd_remap(blockIdx.x, gridDim.x) * p.tpw;
  const int ntl = min(p.tpw, ntile - tile_first);
  const int cb = blockIdx.y;
  const int UT = p.nchunks * 3;         // units per tile
  const int G = ntl * p.nchunks;        // chunks of this workgroup

  // ---- weights: one fetch per WORKGROUP.  Unit u of this cout block (= kernel row dy of a chunk, WUNIT contiguous bytes of the
  // re-laid-out weights) streams by LDS-DMA into ring slot u & 1 while unit u-1 is being multiplied; wave w moves
  // instructions [w*NW_W, (w+1)*NW_W).  (Round-3 measurements: every wave loading its own A operands straight from L2
  // quadruples the weight traffic -- 15 TB/s of L2 reads at full rate -- and cost 16 % of the kernel's time.)
  const __amdgpu_buffer_rsrc_t wrsrc = make_rsrc(reinterpret_cast<const char*>(p.wr) + (size_t)cb * UT * WUNIT, (unsigned)UT * WUNIT);
  const unsigned wvoff = (wv * NW_W * 64 + l) * 16;
  int wsoff = 0;   // unit the NEXT issue fetches (wraps per tile)
  auto issue_w_piece = [&](unsigned slot_off, int i, bool live = true) __attribute__((always_inline)) {
    const int n = wv * NW_W + i;
    const unsigned dst = (n < NWI && live) ? w_base + slot_off + n * 1024 : dummy;
    // (beyond the image: reads the next unit / zeros past the end of the buffer, lands in the dummy page)
    __builtin_amdgcn_raw_ptr_buffer_load_lds(wrsrc, (__attribute__((address_space(3))) void*)dst, 16, wvoff, wsoff + i * 1024, 0, 0);
  };
  auto issue_w_done = [&](bool live = true) __attribute__((always_inline)) {
    const int nx = wsoff + WUNIT == UT * WUNIT ? 0 : wsoff + WUNIT;
    wsoff = live ? nx : wsoff;
  };

  // ---- halo tile: 24 slots of 64 pieces (pixel, 4 fp32 channels); slot r of wave wv = pieces [64 (wv + 4r), +64), fetched
  // by buffer_load_dwordx4 (hardware zero fill outside the image) into a register that stays with the wave until split round
  // r of that chunk consumes it -- a whole chunk later
  struct TileCoord { int b, ty, tx; };
  auto tc_init = [&](int tile) __attribute__((always_inline)) {
    TileCoord t;
    t.tx = tile % p.tiles_x;
    t.ty = (tile / p.tiles_x) % p.tiles_y;
    t.b = tile / (p.tiles_x * p.tiles_y);
    return t;
  };
  auto tc_next = [&](TileCoord& t) __attribute__((always_inline)) {
    if (++t.tx == p.tiles_x) {
      t.tx = 0;
      if (++t.ty == p.tiles_y) { t.ty = 0; ++t.b; }
    }
  };
  TileCoord dma_tc = tc_init(tile_first), epi_tc = dma_tc;
  int dma_c = 0;   // chunk (inside its tile) the next issue_in_begin() call fetches
  unsigned ivoff[NRAW_W];
  int ib = 0, iy0 = 0, ix0 = 0;
  i32x4 rs0, rs1;
  int slotc[NRAW_W];   // ry | rx << 8 | quad << 16 | valid << 24
#pragma unroll
  for (int sl = 0; sl < NRAW_W; ++sl) {
    const int n = wv + 4 * sl, pix = 16 * n + (l >> 2);
    const int ry = pix / HWc, rx = pix - ry * HWc;
    slotc[sl] = ry | (rx << 8) | ((l & 3) << 16) | (pix < NPIX ? (1 << 24) : 0);
  }
  // IO16: piece pid = 64 (wv + 4 sl) + l = (k half pid / 340, halo pixel pid % 340) -> LDS byte pid * 16 of the plane buffer
  unsigned dvoff[3] = {kOOB, kOOB, kOOB};
  __amdgpu_buffer_rsrc_t in_rs16 = make_rsrc(p.src[0].ptr, 0u);
  auto set_source16 = [&](const Src& S) __attribute__((always_inline)) {
#pragma unroll
    for (int sl = 0; sl < 3; ++sl) {
      const int pid = 64 * (wv + 4 * sl) + l, kh = pid >= NPIX ? 1 : 0, pix = pid - kh * NPIX;
      const int ry = pix / HWc, rx = pix - ry * HWc;
      const int iy = iy0 - 1 + ry, ix = ix0 - 1 + rx;
      // (pure ALU -- an invalid lane's offset gets bit 31 set, i.e. lies beyond any num_records -- instead of a select that
      // the compiler turns into a lane-masked region per piece; valid offsets are < 2^31: checked by the host)
      const unsigned bad = (pid < 2 * NPIX && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W) ? 0u : 1u;
      dvoff[sl] = ((unsigned)(iy * S.row_pitch + ix * S.pix_pitch + 8 * kh) * 2u) | (bad << 31);
    }
  };
  auto set_source = [&](const Src& S) __attribute__((always_inline)) {
#pragma unroll
    for (int sl = 0; sl < NRAW_W; ++sl) {
      const int c = slotc[sl];
      const int iy = iy0 - 1 + (c & 0xff), ix = ix0 - 1 + ((c >> 8) & 0xff);
      const unsigned bad = ((c >> 24) != 0 && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W) ? 0u : 1u;
      ivoff[sl] = ((unsigned)(iy * S.row_pitch + ix * S.pix_pitch + 4 * ((c >> 16) & 3)) * 4u) | (bad << 31);   // (as set_source16)
    }
  };
  auto src_rsrc = [&](const Src& S, int b) __attribute__((always_inline)) {
    const unsigned bytes = (unsigned)((p.H - 1) * S.row_pitch + (p.W - 1) * S.pix_pitch + S.C) * 4u;
    return make_rsrc_words(S.ptr + (long long)b * S.img_pitch, bytes);
  };
  int in_soff = 0;
  bool in_first = true;
  auto issue_in_begin = [&]() __attribute__((always_inline)) {   // the next chunk of the workgroup's stream
    const int c0 = dma_c * KC;
    in_first = c0 < p.src[0].C;
    if (++dma_c == p.nchunks) dma_c = 0;
    if constexpr (IO16) {
      if (c0 == 0) {
        ib = dma_tc.b; iy0 = dma_tc.ty * THY; ix0 = dma_tc.tx * TWX;
        tc_next(dma_tc);
        const Src& S = p.src[0];   // (pitches of a bf16 source are in bf16 elements)
        in_rs16 = make_rsrc(reinterpret_cast<const char*>(S.ptr) + (long long)ib * S.img_pitch * 2,
                            (unsigned)((p.H - 1) * S.row_pitch + (p.W - 1) * S.pix_pitch + S.C) * 2u);
        set_source16(S);
      }
      in_soff = c0 * 2;
      return;
    }
    if (c0 == 0) {
      ib = dma_tc.b; iy0 = dma_tc.ty * THY; ix0 = dma_tc.tx * TWX;
      tc_next(dma_tc);
      rs0 = src_rsrc(p.src[0], ib);
      rs1 = src_rsrc(p.src[1], ib);
      set_source(p.src[0]);
    } else if (c0 == p.src[0].C) {
      set_source(p.src[1]);
    }
    in_soff = (in_first ? c0 : c0 - p.src[0].C) * 4;
  };
  // IO16: piece `sl` of the chunk issue_in_begin() has just set up -> plane buffer at byte offset `plane_off`
  __amdgpu_buffer_rsrc_t in_rs16_cur = in_rs16;   // BF16S: in_rs16, or a descriptor of zero records past the stream's end
  const __amdgpu_buffer_rsrc_t null_rs16 = make_rsrc(p.src[0].ptr, 0u);
  auto issue_in_dma = [&](int sl, unsigned plane_off) __attribute__((always_inline)) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(BF16S ? in_rs16_cur : in_rs16, (__attribute__((address_space(3))) void*)(pl_base + plane_off + (unsigned)(wv + 4 * sl) * 1024u), 16,
                                             dvoff[sl], in_soff, 0, 0);
  };
  f32x4 rawr[NRAW_W];
  i32x4 rs_cur = {0, 0, 0, 0x00020000};   // BF: descriptor of the chunk being fetched (set_chunk_rsrc)
  auto set_chunk_rsrc = [&](bool live) __attribute__((always_inline)) {
    const i32x4 r = in_first ? rs0 : rs1;
    rs_cur[0] = r[0];
    rs_cur[1] = r[1];
    rs_cur[2] = live ? r[2] : 0;        // zero records: every lane out of range -> zeros, no memory traffic
    rs_cur[3] = 0x00020000;
  };
  auto issue_in_piece = [&](auto slc) __attribute__((always_inline)) {
    constexpr int sl = decltype(slc)::value;
    if constexpr (BF) {
      buf_load128f(rawr[sl], ivoff[sl], rs_cur, in_soff);
    } else {
      if (in_first) buf_load128f(rawr[sl], ivoff[sl], rs0, in_soff);
      else buf_load128f(rawr[sl], ivoff[sl], rs1, in_soff);
    }
  };

  // ---- split of the wave's own raw pieces into the bf16 planes.  Round r: piece 64 (wv + 4r) + l = (pixel 16 (wv + 4r) +
  // (l >> 2), quad q = l & 3) -> plane slab (q >> 1), 8 bytes at pixel*16 + (q & 1)*8.
  const unsigned cdst = pl_base + ((l >> 1) & 1) * HALFB + (wv * 16 + (l >> 2)) * 16 + (l & 1) * 8;   // + r * 1024 + plane * 2*HALFB
  u32x2 cq[3];
  float amax = 0.0f;   // f16 x 2: largest |activation| this lane has split (domain check, Params::range_flag)
  auto conv_split = [&](const f32x4 v) __attribute__((always_inline)) {
    if constexpr (FL == 3) {
      split3(v, cq[0], cq[1], cq[2]);
    } else if constexpr (FL == 2) {
      split2_f16(v, cq[0], cq[1]);
      // (asm volatile: these must read the raw registers BEFORE the volatile asm that re-loads them is issued -- the compiler
      // does not know that load is asynchronous; plain C++ here was scheduled behind it and made it copy the load's
      // destination registers before the data had arrived)
      asm volatile("v_max3_f32 %0, %0, |%1|, |%2|\n\tv_max3_f32 %0, %0, |%3|, |%4|"
                   : "+v"(amax) : "v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]));
    } else {
      typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
      bf16x4 h;
#pragma unroll
      for (int i = 0; i < 4; ++i) h[i] = (__bf16)v[i];   // round to nearest even
      cq[0] = __builtin_bit_cast(u32x2, h);
    }
  };
  const bool last_ok = (slotc[NRAW_W - 1] >> 24) != 0;   // the last round's slots reach beyond the 340 pixels of a slab
  auto conv_store = [&](auto rr, unsigned dst) __attribute__((always_inline)) {
    constexpr int R = decltype(rr)::value;
    if (R < NRAW_W - 1 || last_ok) {
      lds_write64<R * 1024>(dst, cq[0]);
      if constexpr (NPX >= 2) lds_write64<R * 1024 + 2 * HALFB>(dst, cq[1]);
      if constexpr (NPX >= 3) lds_write64<R * 1024 + 4 * HALFB>(dst, cq[2]);
    }
  };

  // ---- operands: A = lane (cout row j, k half hi) of the ring slot's tap dx, plane pl, channel tile mt;
  //                B = pixel (row 2wv + nt + dy, column j + dx) of the halo tile, k half hi, plane pl
  const unsigned abase = w_base + hi * 512 + j * 16;
  const unsigned bbase = pl_base + hi * HALFB + (2 * wv * HWc + j) * 16;
  bf16x8 A[2][NPW][MT], Bq[2][NPX][NT];   // two operand sets: tap (dy, dx) multiplies set (dy + dx) & 1
  bf16x8 Ad[MT];                          // f16 x 2: 2^-11 wA of the current tap
  constexpr int NLB = NPX * NT, NLA = NPW * MT;
  auto load_a = [&](auto setc, auto dxc, auto kc, unsigned aslot) __attribute__((always_inline)) {   // aslot = abase + ring slot offset
    constexpr int SET = decltype(setc)::value, DX = decltype(dxc)::value, K = decltype(kc)::value;
    lds_read128<DX * WTAP + K * 1024>(A[SET][K / MT][K % MT], aslot);
  };
  auto load_b = [&](auto setc, auto dyc, auto dxc, auto kc, unsigned bcur) __attribute__((always_inline)) {   // bcur = bbase + plane buffer
    constexpr int SET = decltype(setc)::value, DY = decltype(dyc)::value, DX = decltype(dxc)::value, K = decltype(kc)::value;
    // (ablation 2048, timing only: the B operand of pixel row nt = 0 at kernel row dy >= 1 is the one row nt = 1 read at dy - 1 -- skip
    // the re-read, 12 of a chunk's 36 B reads: what holding a unit's four pixel rows in registers would save)
    if constexpr ((ABL & 2048) != 0 && (K % NT) == 0 && DY >= 1) return;
    lds_read128<(K / NT) * 2 * HALFB + ((K % NT + DY) * HWc + DX) * 16>(Bq[SET][K / NT][K % NT], bcur);
  };

  const int co_lane = cb * MW + 4 * hi;
  // f16 x 2: 1/S of the weight images (the float behind the last cout block's image)
  float w_sinv = 1.0f;
  if constexpr (FL == 2) w_sinv = *reinterpret_cast<const float*>(reinterpret_cast<const char*>(p.wr) + (size_t)gridDim.y * UT * WUNIT);
  if (tid < MW) {
    const int co = cb * MW + tid;
    *(__attribute__((address_space(3))) float*)(bias_lds + tid * 4) = (p.bias && co < p.Cout) ? p.bias[co] : 0.0f;
  }
  f32x16 acc[MT][NT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mt][nt][r] = 0.0f;

  // ABL & 1024 (-DC2M_SPLIT_TRACE builds only; scripts/trace_split.py): a timeline of two tiles per wave -- s_memtime just before and just
  // after every unit-end barrier and after the epilogue, kept in the 64 lanes of ONE register and stored once, at the kernel's
  // end, to Params::mask_out [workgroup][wave][64]; lanes 62 / 63 carry XCC_ID / HW_ID (which CU the workgroup ran on).  Every stamp sits
  // where lgkmcnt is already zero, so the s_waitcnt it needs costs the timestamp's own latency only.
  unsigned trace_v = 0u;
  int trace_it = -1;   // tile slot (0 / 1) being traced, or -1
  auto stamp = [&](int ev) __attribute__((always_inline)) {
    if constexpr ((ABL & 1024) != 0) {
      if (trace_it >= 0) {
        unsigned long long t;
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
        const unsigned tl = (unsigned)t;
        const int ln = trace_it * 26 + ev;
        trace_v = l == ln ? tl : trace_v;   // (v_writelane_b32 with two SGPR operands violates gfx9's constant-bus limit)
      }
    }
  };

  // ------------------------------------------------------------------------------------------------------------------
  // prologue: the raw pieces of chunk 0 and the weights of unit 0
  // ------------------------------------------------------------------------------------------------------------------
  issue_in_begin();
  if constexpr (IO16) {
    in_rs16_cur = in_rs16;
#pragma unroll
    for (int sl = 0; sl < 3; ++sl) issue_in_dma(sl, 0u);
  } else {
    if constexpr (BF) set_chunk_rsrc(true);
    static_for<0, NRAW_W>([&](auto rr) __attribute__((always_inline)) { issue_in_piece(rr); });
  }
#pragma unroll
  for (int i = 0; i < NW_W; ++i) issue_w_piece(0u, i);
  issue_w_done();
  if constexpr (NRING == 3) {
    if (3 * G > 1) {
#pragma unroll
      for (int i = 0; i < NW_W; ++i) issue_w_piece((unsigned)WUNIT, i);
      issue_w_done();
    }
  }
  if constexpr (IO16) {   // chunk 1 -> plane buffer 1; both landed and published before the first unit
    if (G > 1) {
      issue_in_begin();
      in_rs16_cur = in_rs16;
#pragma unroll
      for (int sl = 0; sl < 3; ++sl) issue_in_dma(sl, (unsigned)PLB);
    }
    wait_vmcnt<0>();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
  }
  wait_vmcnt<0>();
  if constexpr (PIPE && !IO16) {   // chunk 0 -> plane buffer 0; the registers re-load with chunk 1
    const bool more1 = !(ABL & 2) && G > 1;
    if (more1) issue_in_begin();
    if constexpr (BF) set_chunk_rsrc(more1);
    static_for<0, NRAW_W>([&](auto rr) __attribute__((always_inline)) {
      constexpr int R = decltype(rr)::value;
      conv_split(rawr[R]);
      conv_store(rr, cdst);
      if (more1) issue_in_piece(rr);
    });
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
  }

  // Per chunk, bf16 x 3: (A) every wave splits its raw pieces into the planes (single buffer: the co-resident workgroup owns
  // the matrix pipe meanwhile) and re-loads the registers for the next chunk; barrier; (B) three units (kernel rows) of three
  // taps.  Other flavours (PIPE): no phase (A) -- split round R of the NEXT chunk (registers -> the other plane buffer, then
  // the registers re-load with the chunk after that) rides in the MFMA groups of taps 1 and 2 of unit R / 2.
  // One tap = PR::N groups of MT*NT MFMAs, fenced by sched_barriers so that everything else stays where it is written: the
  // first groups read the next tap's operands (two sets, one tap ahead; the A operands of a unit's FIRST tap are read after
  // the barrier that publishes the unit's weights), the unit's first tap also issues the LDS-DMA pieces of the NEXT unit's
  // weights, one per group.  Unit end: own LDS ops / weight DMAs done (the two raw loads issued after them may still fly),
  // barrier (publishes W(u+1) and the other plane buffer, frees this one / the ring slot).
  constexpr int NG = PR::N;
#ifndef C2M_LPGD
#define C2M_LPGD 2
#endif
  constexpr int LPGD = C2M_LPGD;
  constexpr int LPG = NG >= 4 ? (NLA + NLB + NG - 3) / (NG - 2) : (NG == 3 ? (NLA + NLB + LPGD - 1) / LPGD : NLA + NLB);   // operand reads per group
  static_assert(NRAW_W == 6, "two split rounds per unit");
  unsigned slot_cur = 0u;   // ring slot (byte offset) of the current unit
  for (int it = 0, gc = 0; it < ntl; ++it) {
    if constexpr ((ABL & 1024) != 0) trace_it = (it >= p.co_off && it < p.co_off + 2 && p.nchunks <= 4) ? it - p.co_off : -1;
    for (int c = 0; c < p.nchunks; ++c, ++gc) {
      // PIPE: the registers hold chunk gc+1 (if any); they re-load with chunk gc+2.  Else: they hold chunk gc, re-load with gc+1
      const bool has_next = PIPE ? gc + 1 < G : true;
      const bool more_in = !(ABL & 2) && gc + (PIPE ? 2 : 1) < G;
      const unsigned pb = IO16 ? (unsigned)(gc % 3) * PLB : PIPE ? (unsigned)(gc & 1) * PLB : 0u;
      const unsigned pb_in = (unsigned)((gc + 2) % 3) * PLB;   // IO16: where chunk gc+2 lands
      const unsigned bcur = bbase + pb, cnext = PIPE ? cdst + (PLB - pb) : cdst;
      if (more_in) issue_in_begin();
      if constexpr (BF) set_chunk_rsrc(more_in);
      if constexpr (BF16S) in_rs16_cur = more_in ? in_rs16 : null_rs16;
      if constexpr (!PIPE) {
        // ---- (A) split
        if constexpr (!(ABL & 4)) {
          static_for<0, NRAW_W>([&](auto rr) __attribute__((always_inline)) {
            constexpr int R = decltype(rr)::value;
            conv_split(rawr[R]);
            conv_store(rr, cdst);
            if (BF || more_in) issue_in_piece(rr);    // same slot of the next chunk
          });
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
      }
      // ---- (B) multiply
      static_for<0, 3>([&](auto dyc) __attribute__((always_inline)) {
        constexpr int dy = decltype(dyc)::value;
        const int u = 3 * gc + dy;
        // weights of unit u + NRING - 1 -> the slot unit u - 1 has just left
        const unsigned slot_nxt = slot_cur == 0u ? (unsigned)((NRING - 1) * WUNIT) : slot_cur - (unsigned)WUNIT;
        const unsigned aslot = abase + slot_cur;
        const bool do_w = !(ABL & 1) && u + NRING - 1 < 3 * G;
        // operands of the unit's first tap: A now (its weights were published by the barrier just passed); B too at dy == 0
        static_for<0, NLA>([&](auto kc) __attribute__((always_inline)) {
          load_a(std::integral_constant<int, (dy & 1)>(), std::integral_constant<int, 0>(), kc, aslot);
        });
        if constexpr (dy == 0) {
          static_for<0, NLB>([&](auto kc) __attribute__((always_inline)) {
            load_b(std::integral_constant<int, 0>(), std::integral_constant<int, 0>(), std::integral_constant<int, 0>(), kc, bcur);
          });
        }
        static_for<0, 3>([&](auto dxc) __attribute__((always_inline)) {
          constexpr int dx = decltype(dxc)::value;
          constexpr int set = (dy + dx) & 1, nset = set ^ 1;
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
          __builtin_amdgcn_sched_barrier(0);
          static_for<0, NG>([&](auto gcnt) __attribute__((always_inline)) {
            constexpr int g = decltype(gcnt)::value;
            if constexpr (!(ABL & 32)) {
              // next tap's operands: (dy, dx+1): A and B; after the unit's last tap: only B of (dy+1, 0)
              static_for<g * LPG, (g + 1) * LPG < NLA + NLB ? (g + 1) * LPG : NLA + NLB>([&](auto kc) __attribute__((always_inline)) {
                constexpr int K = decltype(kc)::value;
                if constexpr (dx < 2) {
                  if constexpr (K < NLA) load_a(std::integral_constant<int, nset>(), std::integral_constant<int, dx + 1>(), kc, aslot);
                  else load_b(std::integral_constant<int, nset>(), dyc, std::integral_constant<int, dx + 1>(), std::integral_constant<int, K - NLA>(), bcur);
                } else if constexpr (dy < 2) {
                  if constexpr (K >= NLA)
                    load_b(std::integral_constant<int, nset>(), std::integral_constant<int, dy + 1>(), std::integral_constant<int, 0>(),
                           std::integral_constant<int, K - NLA>(), bcur);
                }
              });
            }
            if constexpr (dx == 0) {
              if constexpr (BFA) {
#pragma unroll
                for (int i = g; i < NW_W; i += NG) issue_w_piece(slot_nxt, i, do_w);
              } else {
                if (do_w) {
#pragma unroll
                  for (int i = g; i < NW_W; i += NG) issue_w_piece(slot_nxt, i);
                }
              }
            }
            if constexpr (FL == 2 && g == 0) {   // wB = 2^-11 wA of this tap (used by group 1)
#pragma unroll
              for (int mt = 0; mt < MT; ++mt) {
                const f16x8 sc = {(_Float16)(1.0f / F16_LO_SCALE), (_Float16)(1.0f / F16_LO_SCALE), (_Float16)(1.0f / F16_LO_SCALE), (_Float16)(1.0f / F16_LO_SCALE),
                                  (_Float16)(1.0f / F16_LO_SCALE), (_Float16)(1.0f / F16_LO_SCALE), (_Float16)(1.0f / F16_LO_SCALE), (_Float16)(1.0f / F16_LO_SCALE)};
                Ad[mt] = __builtin_bit_cast(bf16x8, __builtin_bit_cast(f16x8, A[set][0][mt]) * sc);
              }
            }
            if constexpr (IO16 && dx == 1 && g == NG / 2) {   // halo piece dy of chunk gc+2 (one per unit)
              if (BF16S || more_in) issue_in_dma(dy, pb_in);
            }
            if constexpr (PIPE && !IO16 && dx >= 1 && g == NG / 2 && !(ABL & 4)) {   // split round R of the next chunk
              constexpr int R = dx >= 1 ? 2 * dy + dx - 1 : 0;
              if (BF || has_next) {
                if constexpr (dy == 0) {
                  if (gc == 0) wait_vmcnt<0>();   // (chunk 1's raw pieces were issued by the prologue: no unit end since)
                }
                conv_split(rawr[R]);
                conv_store(std::integral_constant<int, R>(), cnext);
                if (BF || more_in) issue_in_piece(std::integral_constant<int, R>());
              }
            }
            if constexpr (!(ABL & 16) && !((ABL & 256) && dx == 2)) {   // (256: only two of the three taps' MFMAs -- what a Winograd F(2,3) would issue)
#pragma unroll
              for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int nq = 0; nq < NT; ++nq) {
                  const int nt = nq;
                  const bf16x8 av = PR::W[g] < NPW ? A[set][PR::W[g] < NPW ? PR::W[g] : 0][mt] : Ad[mt];
                  if constexpr (PR::F16)
                    acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, av),
                                                                         __builtin_bit_cast(f16x8, Bq[set][PR::X[g]][nt]), acc[mt][nt], 0, 0, 0);
                  else
                    acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, Bq[set][PR::X[g]][nt], acc[mt][nt], 0, 0, 0);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
          });
          if constexpr (dx == 0) {
            if constexpr (BFA) issue_w_done(do_w);
            else if (do_w) issue_w_done();
          }
        });
        if (!(ABL & 8)) {
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
          // PIPE, steady state: the weights of unit u+1 (issued in unit u-1) landed, and with them every raw load older than
          // unit u-1's; still in flight may be: raw(u-1) x 2, W(u+2) x NW_W, raw(u) x 2.  (vmcnt counts in issue order.)
          if constexpr (BF && PIPE) {
            wait_vmcnt<4 + NW_W>();   // (every unit issued its NW_W pieces and two loads, live or not: one constant count)
          } else if constexpr (BF16S) {
            wait_vmcnt<2 + NW_W>();   // (likewise: NW_W pieces and one halo piece per unit)
          } else if (IO16 && more_in) {
            wait_vmcnt<2 + NW_W>();   // in flight may be: halo piece of unit u-1, W(u+2) x NW_W, halo piece of unit u
          } else if (PIPE && !IO16 && more_in) {
            if ((ABL & 128) && c == 0 && dy == 0) wait_vmcnt<4 + NW_W + 16>();
            else wait_vmcnt<4 + NW_W>();
          } else wait_vmcnt<0>();
          stamp(2 * (3 * c + dy));
          __builtin_amdgcn_s_barrier();
          stamp(2 * (3 * c + dy) + 1);
        }
        slot_cur = slot_cur == (unsigned)((NRING - 1) * WUNIT) ? 0u : slot_cur + (unsigned)WUNIT;
      });
    }
    // ----------------------------------------------------------------------------------------------------------------
    // epilogue of the tile
    // ----------------------------------------------------------------------------------------------------------------
    const int b = epi_tc.b, y0 = epi_tc.ty * THY, x0 = epi_tc.tx * TWX;
    tc_next(epi_tc);
    // the lane's first output channel, re-read through an opaque move once per tile: left loop-invariant, hipcc hoists the generic path's
    // sixteen residual / output base addresses (pointer + channel offset, 64-bit pairs) out of the tile loop and keeps them in 32 registers
    // of the MFMA loop -- or spills them, and every reload in the epilogue then carries an s_waitcnt vmcnt(0) that drains the stores
    int co_e = co_lane;
    if constexpr (MODE == 0) asm volatile("" : "+v"(co_e));   // (the DCN head's quad stores measured 3 % slower with it)
    // Channels-last tiles that lie inside the image with all their channels (every tile of the 64 -> 64 bodies but the map's last row /
    // column of tiles): a straight-line epilogue.  Row 0's eight residual pieces are requested HERE, in front of the
    // bias / activation arithmetic, row 1's before row 0 is stored.  (The generic path below fetches each piece behind its own branches
    // and between two stores that may alias it: hipcc serialises that into 16 x (load, s_waitcnt vmcnt(0), add, store), and on gfx9 that
    // wait also drains the previous store -- 21 000 cycles per tile against 7 300 without a residual, scripts/trace_split.py.)
    constexpr bool EPI_FAST = MODE == 0 && !IO16 && (ABL & (64 | 512)) == 0;
    bool fast = false;
    size_t fpix = 0;
    f32x4 rv[MT][4];   // one pixel row's pieces at a time (both rows: 64 registers, and hipcc spills)
    auto res_load = [&](const float* r, int nt) __attribute__((always_inline)) {
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int qd = 0; qd < 4; ++qd) rv[mt][qd] = *reinterpret_cast<const f32x4*>(r + fpix + (size_t)nt * p.out_row_pitch + mt * 32 + 8 * qd);
    };
    auto res_add = [&](int nt) __attribute__((always_inline)) {
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int qd = 0; qd < 4; ++qd)
#pragma unroll
          for (int e = 0; e < 4; ++e) acc[mt][nt][4 * qd + e] += rv[mt][qd][e];
    };
    // the same for bf16 tensors (configs[4]'s bodies: bf16 output, bf16 residual): both rows' four 16-byte pieces per lane, requested here
    typedef __bf16 bf16x8r __attribute__((ext_vector_type(8)));
    constexpr bool EPI_FAST16 = IO16 && (ABL & (64 | 512)) == 0;   // (IO16 implies FL == 1, MODE == 0)
    bool fast16 = false;
    bf16x8r hres[NT][MT][2];
    if constexpr (EPI_FAST16) {
      fast16 = (p.io_flags & 6) == 6 && p.res1 != nullptr && cb * MW + MW <= p.Cout && y0 + THY <= p.H && x0 + TWX <= p.W;
      if (fast16) {
        const __bf16* rb = reinterpret_cast<const __bf16*>(p.res1) + (size_t)b * p.out_img_pitch + (size_t)(y0 + 2 * wv) * p.out_row_pitch +
                           (size_t)(x0 + j) * p.out_pix_pitch + cb * MW + 8 * hi;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
          for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int k = 0; k < 2; ++k) hres[nt][mt][k] = *reinterpret_cast<const bf16x8r*>(rb + (size_t)nt * p.out_row_pitch + mt * 32 + 16 * k);
      }
    }
    if constexpr (EPI_FAST) {
      fast = p.out_vec4 != 0 && cb * MW + MW <= p.Cout && y0 + THY <= p.H && x0 + TWX <= p.W && (FL != 1 || (p.io_flags & 14) == 0);
      fpix = (size_t)b * p.out_img_pitch + (size_t)(y0 + 2 * wv) * p.out_row_pitch + (size_t)(x0 + j) * p.out_pix_pitch + co_e;
      if (fast && p.res1 != nullptr) res_load(p.res1, 0);
    }
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int qd = 0; qd < 4; ++qd) {
        const f32x4 bv = *(const __attribute__((address_space(3))) f32x4*)(bias_lds + (mt * 32 + 8 * qd + 4 * hi) * 4);
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            if constexpr (FL == 2) acc[mt][nt][4 * qd + e] = acc[mt][nt][4 * qd + e] * w_sinv + bv[e];   // (power of two: exact)
            else acc[mt][nt][4 * qd + e] += bv[e];
          }
      }
    if constexpr ((ABL & 64) != 0) {
      // (ablation: one 16-byte store per lane and tile -- the sum of its accumulators -- instead of sixteen)
      f32x4 v = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
          for (int r = 0; r < 16; ++r) v[r & 3] += acc[mt][nt][r];
      const int y = y0 + 2 * wv, x = x0 + j;
      if (y < p.H && x < p.W) {
        const size_t o = (size_t)b * p.out_img_pitch + (size_t)y * p.out_row_pitch + (size_t)x * p.out_pix_pitch + co_e;
        if constexpr (IO16) *reinterpret_cast<f32x4*>(reinterpret_cast<__bf16*>(p.out) + 2 * (o / 2)) = v;   // (stays inside a bf16 tensor)
        else *reinterpret_cast<f32x4*>(p.out + o) = v;
      }
    } else if constexpr (MODE == 3 || MODE == 5) {
      float asum = 0.0f;
      const HeadOut ho = head_out(p, b);
      // MODE 5 (W % 4 == 0; $C2M_HEAD_QUAD=0 keeps MODE 3): 16-byte planar stores after a 4 x 4 transpose inside the lane
      // quads, pre-offsets from a per-row flow window held in registers (conv3x3_shared.h)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        const int y = y0 + 2 * wv + nt, x = x0 + j;
        const bool pok = y < p.H && x < p.W;
        HeadFlowWin fwin;
        if constexpr (MODE == 5) fwin = head_flow_window(p, b, min(y, p.H - 1), x0, l);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
          for (int qd = 0; qd < 4; ++qd) {
            const int col = co_e + mt * 32 + 8 * qd;   // channel inside this launch's slice
            if constexpr (MODE == 5) {
              if (col - 4 * hi >= p.Cout) continue;       // (wave-uniform: the whole 8-channel group lies beyond the slice)
            } else {
              if (col >= p.Cout || !pok) continue;
            }
            f32x4 v;
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = acc[mt][nt][4 * qd + e];
            if constexpr (MODE == 5) asum += dcn_head_store_quad(p, ho, y, x, col - 4 * hi, 4 * hi, v, j, fwin, pok && col + 3 < p.Cout);
            else asum += dcn_head_store(p, ho, b, y, x, col - 4 * hi, 4 * hi, v);
          }
      }
      // (Round 4 staged this epilogue -- and the planar NCHW one -- through a per-wave [32 channels][32 pixels] LDS tile so that
      // a lane stores four consecutive pixels of one channel: 16 store instructions per wave and tile instead of 64, head
      // 64 -> 216 @640^2 7.30 -> 5.72 ms.  It is NOT in the tree: on CUs with two resident workgroups one in ~10^6 of the
      // values came out wrong.  Two causes were found and fixed -- a 16-byte buffer store with a REGISTER soffset followed at
      // once by a VALU write of its data registers stores the new value for lanes 12-15 / 28-31 / 44-47 / 60-63 (hipcc guards
      // that hazard only when soffset is an immediate); type-based alias analysis let 16-byte LDS loads overtake dword LDS
      // stores -- a third (lanes 60-63 of flow-dependent channels, 32-200 values per 59 M) was not.  DESIGN.md 6.2.)
      if (p.abs_sum) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) asum += __shfl_xor(asum, off, 64);
        if (l == 0) atomicAdd(p.abs_sum + ((blockIdx.x * 4 + wv + blockIdx.y * 31 + it) & (C2M_ABS_SUM_SLOTS - 1)), (double)asum);
      }
    } else {
      if (p.act == 1) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
          for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mt][nt][r] = fmaxf(acc[mt][nt][r], 0.0f);
      } else if (p.act == 2) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
          for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mt][nt][r] = fmaxf(acc[mt][nt][r], acc[mt][nt][r] * p.slope);
      }
      if constexpr (MODE == 4) {
        // MaxPool2d(2, 2): rows 2wv, 2wv+1 are the two accumulator sets of this lane, the horizontal neighbour is lane j ^ 1
        const int yo = (y0 >> 1) + wv, xo = (x0 + j) >> 1;
        const bool pok = (y0 + 2 * wv + 1) < p.H && (x0 + j) < p.W && (j & 1) == 0;
        float* ob = p.out + (size_t)b * p.out_img_pitch + (size_t)yo * p.out_row_pitch + (size_t)xo * p.out_pix_pitch + co_e;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
          for (int qd = 0; qd < 4; ++qd) {
            f32x4 v;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const float m = fmaxf(acc[mt][0][4 * qd + e], acc[mt][1][4 * qd + e]);
              v[e] = fmaxf(m, __shfl_xor(m, 1, 64));
            }
            if (pok && co_e + mt * 32 + 8 * qd + 3 < p.Cout) *reinterpret_cast<f32x4*>(ob + mt * 32 + 8 * qd) = v;
          }
      } else if (EPI_FAST && fast) {
        static_assert(NT == 2, "two pixel rows per wave");
        auto store_row = [&](int nt) __attribute__((always_inline)) {
#pragma unroll
          for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int qd = 0; qd < 4; ++qd) {
              f32x4 v;
#pragma unroll
              for (int e = 0; e < 4; ++e) v[e] = acc[mt][nt][4 * qd + e];
              *reinterpret_cast<f32x4*>(p.out + fpix + (size_t)nt * p.out_row_pitch + mt * 32 + 8 * qd) = v;
            }
        };
        // row 0's pieces were requested at the top of the epilogue; row 1's go out before row 0 is stored
        if (p.res1 != nullptr) { res_add(0); res_load(p.res1, 1); }
        if (p.res2 != nullptr) {   // (the stage input on a body's last block)
          if (p.res1 != nullptr) res_add(1);
          res_load(p.res2, 0); res_add(0);
          res_load(p.res2, 1);
        }
        store_row(0);
        if (p.res1 != nullptr || p.res2 != nullptr) res_add(1);
        store_row(1);
      } else {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
          const int y = y0 + 2 * wv + nt, x = x0 + j;
          const bool pok = y < p.H && x < p.W;
          if (!pok) continue;
          if constexpr (MODE == 0) {
            const size_t opix = (size_t)b * p.out_img_pitch + (size_t)y * p.out_row_pitch + (size_t)x * p.out_pix_pitch;
            float* ob = p.out + opix + co_e;
            if constexpr (FL == 1) {
              if (p.io_flags & 2) {
                // bf16 output (Cout % 8 == 0): lanes j and j + 32 hold channels 8qd + 0..3 / 8qd + 4..7 of the same pixel;
                // v_permlane32_swap hands lane j the whole 8-channel group 2k and lane j + 32 the whole group 2k + 1, so
                // that a lane stores (and fetches its residuals as) 16 bytes: 8 stores per wave and tile instead of 16
                typedef __bf16 bf16x8v __attribute__((ext_vector_type(8)));
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                  for (int k = 0; k < 2; ++k) {
                    float v[8];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                      // (copies first: __builtin_bit_cast applied to a vector ELEMENT expression reads element 0 with this hipcc)
                      const float lo = acc[mt][nt][8 * k + e], up = acc[mt][nt][8 * k + 4 + e];
                      const auto r = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, lo), __builtin_bit_cast(unsigned, up), false, false);
                      const unsigned r0 = r[0], r1 = r[1];
                      v[e] = __builtin_bit_cast(float, r0);
                      v[4 + e] = __builtin_bit_cast(float, r1);
                    }
                    const int co8 = cb * MW + mt * 32 + 8 * (2 * k + hi);
                    if (co8 + 7 < p.Cout) {
                      auto res = [&](const float* r, bool is16) __attribute__((always_inline)) {
                        if (is16) {
                          const bf16x8v h = *reinterpret_cast<const bf16x8v*>(reinterpret_cast<const __bf16*>(r) + opix + co8);
#pragma unroll
                          for (int e = 0; e < 8; ++e) v[e] += (float)h[e];
                        } else {
                          const f32x4 a = *reinterpret_cast<const f32x4*>(r + opix + co8), c = *reinterpret_cast<const f32x4*>(r + opix + co8 + 4);
#pragma unroll
                          for (int e = 0; e < 4; ++e) { v[e] += a[e]; v[4 + e] += c[e]; }
                        }
                      };
                      if (EPI_FAST16 && fast16) {
#pragma unroll
                        for (int e = 0; e < 8; ++e) v[e] += (float)hres[nt][mt][k][e];
                      } else if (p.res1) res(p.res1, (p.io_flags & 4) != 0);
                      if (p.res2) res(p.res2, (p.io_flags & 8) != 0);
                      bf16x8v h;
#pragma unroll
                      for (int e = 0; e < 8; ++e) h[e] = (__bf16)v[e];
                      *reinterpret_cast<bf16x8v*>(reinterpret_cast<__bf16*>(p.out) + opix + co8) = h;
                    }
                  }
                continue;
              }
            }
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
              for (int qd = 0; qd < 4; ++qd) {
                const int co = co_e + mt * 32 + 8 * qd;
                f32x4 v;
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = acc[mt][nt][4 * qd + e];
                if constexpr ((ABL & 512) != 0) {
                  // (ablation: the ADDRESS pattern of quad-transposed stores -- lanes 4q .. 4q+3 write the four consecutive 16-byte
                  // pieces of pixel 4q + qd -- with untransposed data: what would coalesced 64-byte runs per quad buy?)
                  const int xq = x0 + 4 * (j >> 2) + qd, cq = cb * MW + mt * 32 + 4 * (4 * hi + (j & 3));
                  if (xq < p.W && cq + 3 < p.Cout) {
                    const size_t oq = (size_t)b * p.out_img_pitch + (size_t)y * p.out_row_pitch + (size_t)xq * p.out_pix_pitch + cq;
                    if (p.res1) v += *reinterpret_cast<const f32x4*>(p.res1 + oq);
                    if (p.res2) v += *reinterpret_cast<const f32x4*>(p.res2 + oq);
                    *reinterpret_cast<f32x4*>(p.out + oq) = v;
                  }
                  continue;
                }
                if constexpr (FL == 1) {
                  // bf16 flavour: the output / the residuals may hold bf16 (Params::io_flags; Cout % 4 == 0 then) -- the sum
                  // is taken in fp32 and rounded once
                  if (p.io_flags & 12) {   // fp32 output, bf16 residual(s)
                    typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
                    if (co + 3 < p.Cout) {
                      auto res = [&](const float* r, bool is16) __attribute__((always_inline)) {
                        if (is16) {
                          const bf16x4 h = *reinterpret_cast<const bf16x4*>(reinterpret_cast<const __bf16*>(r) + opix + co);
#pragma unroll
                          for (int e = 0; e < 4; ++e) v[e] += (float)h[e];
                        } else {
                          v += *reinterpret_cast<const f32x4*>(r + opix + co);
                        }
                      };
                      if (p.res1) res(p.res1, (p.io_flags & 4) != 0);
                      if (p.res2) res(p.res2, (p.io_flags & 8) != 0);
                      *reinterpret_cast<f32x4*>(ob + mt * 32 + 8 * qd) = v;
                    }
                    continue;
                  }
                }
                if (co + 3 < p.Cout) {
                  // (residuals are fetched here: the co-resident workgroup's MFMAs cover the latency)
                  if (p.res1) v += *reinterpret_cast<const f32x4*>(p.res1 + opix + co);
                  if (p.res2) v += *reinterpret_cast<const f32x4*>(p.res2 + opix + co);
                  *reinterpret_cast<f32x4*>(ob + mt * 32 + 8 * qd) = v;
                } else {
                  for (int e = 0; e < 4 && co + e < p.Cout; ++e) {
                    float sv = v[e];
                    if (p.res1) sv += p.res1[opix + co + e];
                    if (p.res2) sv += p.res2[opix + co + e];
                    ob[mt * 32 + 8 * qd + e] = sv;
                  }
                }
              }
          } else if constexpr (MODE == 1) {
            // PixelShuffle(2): channel 4*c2 + 2*dy + dx of pixel (y, x) -> channel c2 of pixel (2y + dy, 2x + dx)
            float* ob = p.out + (size_t)b * p.out_img_pitch + (size_t)(2 * y) * p.out_row_pitch +
                        (size_t)(2 * x) * p.out_pix_pitch + (co_e >> 2);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
              if (cb * MW + mt * 32 + 32 <= p.Cout && p.out_vec4 && !(p.io_flags & C2M_IO_DWORD_STORES)) {
                // whole 32-channel tile: lanes j / j + 32 hold output channels 8mt + 2qd + {0, 1} of the four output pixels;
                // two v_permlane32_swap per output pixel hand lane j channels 8mt + 0..3 and lane j + 32 channels 8mt + 4..7:
                // one 16-byte store each -- 8 stores per row instead of 32
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                  const float q0 = acc[mt][nt][e], q1 = acc[mt][nt][4 + e], q2 = acc[mt][nt][8 + e], q3 = acc[mt][nt][12 + e];
                  const auto ra = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, q0), __builtin_bit_cast(unsigned, q2), false, false);
                  const auto rb = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, q1), __builtin_bit_cast(unsigned, q3), false, false);
                  const unsigned a0 = ra[0], a1 = ra[1], b0 = rb[0], b1 = rb[1];
                  const f32x4 v = {__builtin_bit_cast(float, a0), __builtin_bit_cast(float, a1), __builtin_bit_cast(float, b0), __builtin_bit_cast(float, b1)};
                  *reinterpret_cast<f32x4*>(ob + (size_t)(e >> 1) * p.out_row_pitch + (size_t)(e & 1) * p.out_pix_pitch + mt * 8 + 3 * hi) = v;   // (ob holds + hi already: channel cb*16 + 8mt + 4hi)
                }
                continue;
              }
#pragma unroll
              for (int qd = 0; qd < 4; ++qd)
                if (co_e + mt * 32 + 8 * qd < p.Cout) {
#pragma unroll
                  for (int e = 0; e < 4; ++e)
                    ob[(size_t)(e >> 1) * p.out_row_pitch + (size_t)(e & 1) * p.out_pix_pitch + mt * 8 + 2 * qd] = acc[mt][nt][4 * qd + e];
                }
            }
          } else {
            const size_t HWs = (size_t)p.H * p.W;
            if ((p.W & 3) == 0 && !(p.io_flags & C2M_IO_DWORD_STORES)) {
              // planar output, W % 4 == 0: the quad transpose of the DCN head (conv3x3_shared.h) -- lane 4q + i stores channel
              // co + i of pixels 4q .. 4q + 3 as one 16-byte piece: 16 stores per wave and tile instead of 64.  (A quad's
              // pixels are valid together, so the lanes the `continue` above removed never exchange with active ones.)
#pragma unroll
              for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int qd = 0; qd < 4; ++qd) {
                  f32x4 v;
#pragma unroll
                  for (int e = 0; e < 4; ++e) v[e] = acc[mt][nt][4 * qd + e];
                  const f32x4 t = quad_transpose(v, j);
                  const int ch = co_e + mt * 32 + 8 * qd + (j & 3);
                  if (ch < p.Cout) *reinterpret_cast<f32x4*>(p.out + ((size_t)b * p.Cout + ch) * HWs + (size_t)y * p.W + (x & ~3)) = t;
                }
              continue;
            }
            float* ob = p.out + ((size_t)b * p.Cout + co_e) * HWs + (size_t)y * p.W + x;
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
              for (int r = 0; r < 16; ++r) {
                const int cr = mt * 32 + 8 * (r >> 2) + (r & 3);
                if (co_e + cr < p.Cout) ob[(size_t)cr * HWs] = acc[mt][nt][r];
              }
          }
        }
      }
    }
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[mt][nt][r] = 0.0f;
    stamp(24);
  }
  if constexpr (FL == 2) {
    if (p.range_flag != nullptr && !(amax < 65520.0f)) *p.range_flag = 1;   // (rare, idempotent store; inf counts)
  }
  if constexpr ((ABL & 1024) != 0) {
    const unsigned xcc = __builtin_amdgcn_s_getreg(20 | (3 << 11)), hwid = __builtin_amdgcn_s_getreg(4 | (31 << 11));   // HW_REG_XCC_ID[3:0], HW_REG_HW_ID
    asm volatile("v_writelane_b32 %0, %1, 62\n\tv_writelane_b32 %0, %2, 63" : "+v"(trace_v) : "s"(xcc), "s"(hwid));
    reinterpret_cast<unsigned*>(p.mask_out)[((size_t)blockIdx.x * 4 + wv) * 64 + l] = trace_v;
  }
}

}  // namespace split
}  // namespace conv
}  // namespace c2m

// =====================================================================================================================
// host side (called by c2m_conv3x3_nhwc_f32 / the relayout entry points in conv3x3.hip)
// =====================================================================================================================
using namespace c2m;

namespace c2m {
namespace conv {

// np = flavour: 3 bf16 x 3, 1 bf16, 2 f16 x 2 (image + 256-byte tail holding 1/S)
// np = pieces code: 3 / 1 / 2, or 2 | R << 4 (R = 4, 2): the Winograd F(R,3)-along-y images (Cout % 64 == 0)
static size_t split_image_bytes(int Cin, int Cout, int npc) {
  const int np = npc & 15, R = npc >> 4;
  if (Cin <= 0 || Cout <= 0 || Cin % split::KC != 0 || (np != 1 && np != 2 && np != 3)) return 0;
  if (R != 0 && (np != 2 || (R != 4 && R != 2) || Cout % 64 != 0)) return 0;
  const int MT = Cout <= 32 ? 1 : 2, ncb = (Cout + 32 * MT - 1) / (32 * MT);
  return (size_t)ncb * (Cin / split::KC) * (R ? R + 2 : 3) * 3 * split::npw_of(np) * MT * 1024;
}

size_t split_relayout_bytes(int Cin, int Cout, int np) {
  const size_t b = split_image_bytes(Cin, Cout, np);
  return b == 0 ? 0 : b + ((np & 15) == 2 ? 256 : 0);
}

int split_relayout(hipStream_t st, const float* weight, int Cin, int Cout, int np, void* wr, int dgrad) {
  const size_t bytes = split_image_bytes(Cin, Cout, np);
  if (bytes == 0) return C2M_ERR_UNSUPPORTED;
  const long long total = (long long)(bytes / 2);
  if ((np >> 4) != 0 && dgrad) return C2M_ERR_UNSUPPORTED;
  if ((np & 15) == 2) {   // per-tensor scale: max |w| -> tail of the image
    const long long n = (long long)Cin * Cout * 9;
    (void)hipMemsetAsync(reinterpret_cast<char*>(wr) + bytes, 0, 256, st);
    hipLaunchKernelGGL(split::weight_absmax_kernel, dim3((unsigned)std::min<long long>(64, (n + 4095) / 4096)), dim3(256), 0, st, weight, n,
                       reinterpret_cast<unsigned*>(reinterpret_cast<char*>(wr) + bytes));
  }
  hipLaunchKernelGGL(split::conv3x3_relayout_split_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, weight, Cin,
                     Cout, np, Cout <= 32 ? 1 : 2, total, reinterpret_cast<unsigned short*>(wr), dgrad);
  return check_launch();
}

int split_relayout_multi(hipStream_t st, const long long* jobs, int njobs, long long nblocks, int any_f16) {
  if (njobs <= 0 || nblocks <= 0 || nblocks > 0x7fffffffLL || !jobs) return C2M_ERR_INVALID_ARG;
  if (any_f16) hipLaunchKernelGGL(split::weight_absmax_multi_kernel, dim3((unsigned)njobs, split::ABSMAX_SPLIT), dim3(256), 0, st, jobs);
  hipLaunchKernelGGL(split::conv3x3_relayout_split_multi_kernel, dim3((unsigned)nblocks), dim3(256), 0, st, jobs, njobs);
  return check_launch();
}

thread_local int g_head_stores = -1;   // c2m_conv3x3_set_head_stores

template <int NP, int MT>
static int launch_split_mode(hipStream_t st, const Params& p, dim3 grid) {
  constexpr size_t ldsb = (size_t)((NP != 3 ? 2 : 1) * split::npx_of(NP) * 2 * split::HALFB) + (NP != 3 ? 3 : 2) * (size_t)(3 * split::npw_of(NP) * MT * 1024) + 1024 +
                          256;   // planes (x2 when pipelined), weight ring (3 / 2 slots), dummy, bias
  static unsigned long long done[5] = {};
  int rc = C2M_OK;
  auto go = [&](auto kern, unsigned long long& dn) {
    if ((rc = ensure_dynamic_lds(reinterpret_cast<const void*>(kern), ldsb, dn)) == C2M_OK)
      hipLaunchKernelGGL(kern, grid, dim3(256), ldsb, st, p);
  };
  if constexpr (NP == 1) {
    if (p.io_flags & C2M_IO_SRC_BF16) {   // bf16 source: three 12 KiB plane buffers filled by LDS-DMA
      constexpr size_t lds16 = (size_t)3 * 12 * 1024 + 3 * (size_t)(3 * MT * 1024) + 1024 + 256;
      static unsigned long long done16 = 0;
      auto go16 = [&](auto kern, unsigned long long& dn) {
        if ((rc = ensure_dynamic_lds(reinterpret_cast<const void*>(kern), lds16, dn)) == C2M_OK)
          hipLaunchKernelGGL(kern, grid, dim3(256), lds16, st, p);
      };
      if constexpr (MT == 2) {   // (timing-only ablations, as for the f16 x 2 flavour: $C2M_SPLIT_ABL16)
        static const int abl = [] {
          const char* e = getenv("C2M_SPLIT_ABL16");
          const int v = e ? atoi(e) : 0;
          if (v > 0) fprintf(stderr, "c2m: C2M_SPLIT_ABL16=%d -- the bf16-tensor conv3x3 kernel runs a timing-only ablation, its results are wrong\n", v);
          return v;
        }();
        static unsigned long long dn[9] = {};
        switch (abl) {
          case 0: break;
          case 1: go16(&split::conv3x3_split_kernel<1, 2, 0, 1, true>, dn[0]); return rc;
          case 2: go16(&split::conv3x3_split_kernel<1, 2, 0, 2, true>, dn[1]); return rc;
          case 8: go16(&split::conv3x3_split_kernel<1, 2, 0, 8, true>, dn[2]); return rc;
          case 16: go16(&split::conv3x3_split_kernel<1, 2, 0, 16, true>, dn[3]); return rc;
          case 32: go16(&split::conv3x3_split_kernel<1, 2, 0, 32, true>, dn[4]); return rc;
          case 64: go16(&split::conv3x3_split_kernel<1, 2, 0, 64, true>, dn[5]); return rc;
          case 43: go16(&split::conv3x3_split_kernel<1, 2, 0, 43, true>, dn[6]); return rc;
          case 48: go16(&split::conv3x3_split_kernel<1, 2, 0, 48, true>, dn[7]); return rc;
          case 107: go16(&split::conv3x3_split_kernel<1, 2, 0, 107, true>, dn[8]); return rc;
          default: fprintf(stderr, "c2m: unknown C2M_SPLIT_ABL16 mask\n"); return C2M_ERR_INVALID_ARG;
        }
      }
      go16(&split::conv3x3_split_kernel<1, MT, 0, 0, true>, done16);
      return rc;
    }
  }
  if constexpr (NP == 2 && MT == 2) {   // (timing-only ablations exist for the f16 x 2 flavour on 64-wide cout tiles)
    static const int abl = [] {
      const char* e = getenv("C2M_SPLIT_ABL");
      const int v = e ? atoi(e) : 0;
      if (v > 0) fprintf(stderr, "c2m: C2M_SPLIT_ABL=%d -- conv3x3 split kernel runs a timing-only ablation, its results are wrong\n", v);
      return v;
    }();
    static unsigned long long done_abl[13] = {};
    if (abl > 0 && p.out_mode == 0) {
      switch (abl) {
        case 1: go(&split::conv3x3_split_kernel<NP, 2, 0, 1>, done_abl[1]); break;
        case 2: go(&split::conv3x3_split_kernel<NP, 2, 0, 2>, done_abl[2]); break;
        case 6: go(&split::conv3x3_split_kernel<NP, 2, 0, 6>, done_abl[3]); break;
        case 8: go(&split::conv3x3_split_kernel<NP, 2, 0, 8>, done_abl[4]); break;
        case 32: go(&split::conv3x3_split_kernel<NP, 2, 0, 32>, done_abl[5]); break;
        case 39: go(&split::conv3x3_split_kernel<NP, 2, 0, 39>, done_abl[6]); break;
        case 47: go(&split::conv3x3_split_kernel<NP, 2, 0, 47>, done_abl[7]); break;
        case 48: go(&split::conv3x3_split_kernel<NP, 2, 0, 48>, done_abl[8]); break;
        case 64: go(&split::conv3x3_split_kernel<NP, 2, 0, 64>, done_abl[9]); break;
        case 128: go(&split::conv3x3_split_kernel<NP, 2, 0, 128>, done_abl[10]); break;
        case 111: go(&split::conv3x3_split_kernel<NP, 2, 0, 111>, done_abl[11]); break;
        case 112: go(&split::conv3x3_split_kernel<NP, 2, 0, 112>, done_abl[12]); break;
        case 256: go(&split::conv3x3_split_kernel<NP, 2, 0, 256>, done_abl[0]); break;
        case 512: { static unsigned long long d512 = 0; go(&split::conv3x3_split_kernel<NP, 2, 0, 512>, d512); break; }
#ifdef C2M_SPLIT_TRACE
        case 2048: { static unsigned long long d2048 = 0; go(&split::conv3x3_split_kernel<NP, 2, 0, 2048>, d2048); break; }
        case 1024: {   // timeline build: RIGHT results; every launch appends [int grid][int tpw][grid x 4 x 64 words] to $C2M_SPLIT_TRACE_FILE
          static unsigned long long d1024 = 0;
          static unsigned* tbuf = nullptr;
          const size_t tb = (size_t)grid.x * 4 * 64 * sizeof(unsigned);
          if (!tbuf && hipMalloc(&tbuf, 4096 * 4 * 64 * sizeof(unsigned)) != hipSuccess) return C2M_ERR_LAUNCH;
          if (grid.x > 4096) return C2M_ERR_UNSUPPORTED;
          (void)hipMemsetAsync(tbuf, 0, tb, st);
          Params q = p;
          q.mask_out = reinterpret_cast<float*>(tbuf);
          const char* e = getenv("C2M_SPLIT_TRACE_IT");
          q.co_off = e ? atoi(e) : 20;
          if ((rc = ensure_dynamic_lds(reinterpret_cast<const void*>(&split::conv3x3_split_kernel<NP, 2, 0, 1024>), ldsb, d1024)) == C2M_OK) {
            hipLaunchKernelGGL((split::conv3x3_split_kernel<NP, 2, 0, 1024>), grid, dim3(256), ldsb, st, q);
            const char* fn = getenv("C2M_SPLIT_TRACE_FILE");
            if (fn) {
              (void)hipStreamSynchronize(st);
              unsigned* h = (unsigned*)malloc(tb);
              (void)hipMemcpy(h, tbuf, tb, hipMemcpyDeviceToHost);
              FILE* f = fopen(fn, "ab");
              if (f) { const int hd[4] = {(int)grid.x, p.tpw, p.res1 != nullptr, p.H}; fwrite(hd, 4, 4, f); fwrite(h, 1, tb, f); fclose(f); }
              free(h);
            }
          }
          break;
        }
#endif
        default: fprintf(stderr, "c2m: unknown C2M_SPLIT_ABL mask\n"); return C2M_ERR_INVALID_ARG;
      }
      return rc;
    }
  }
  switch (p.out_mode) {
    case 0: go(&split::conv3x3_split_kernel<NP, MT, 0>, done[0]); break;
    case 1: go(&split::conv3x3_split_kernel<NP, MT, 1>, done[1]); break;
    case 2: go(&split::conv3x3_split_kernel<NP, MT, 2>, done[2]); break;
    case 3: {
      static const int env_quad = [] { const char* e = getenv("C2M_HEAD_QUAD"); return e ? atoi(e) : 1; }();
      const int head_quad = g_head_stores >= 0 ? g_head_stores : env_quad;
      static unsigned long long done5 = 0;
      if (head_quad != 0 && p.W % 4 == 0) go(&split::conv3x3_split_kernel<NP, MT, 5>, done5);
      else go(&split::conv3x3_split_kernel<NP, MT, 3>, done[3]);
      break;
    }
    default: go(&split::conv3x3_split_kernel<NP, MT, 4>, done[4]); break;
  }
  return rc;
}

// p: as filled by c2m_conv3x3_nhwc_f32 (tiles / nchunks / tpw are set here)
#ifdef C2M_EXPERIMENTAL
bool pc_supported(const Params& p);          // experimental/conv3x3_pc.hip: the same arithmetic with loader / matrix waves
int launch_pc(hipStream_t st, Params p);
#endif

int launch_split(hipStream_t st, Params p, int np) {
#ifdef C2M_EXPERIMENTAL
  // $C2M_CONV_PC: 1 = the f16 x 2 flavour's channels-last 64-cout layers on maps of >= C2M_CONV_PC_MINPIX pixels (default 320^2)
  // run the loader / matrix-wave kernel (same weight images); measured +11 % slower
  static const int env_pc = [] { const char* e = getenv("C2M_CONV_PC"); return e ? atoi(e) : 0; }();
  static const long long pc_minpix = [] { const char* e = getenv("C2M_CONV_PC_MINPIX"); return e ? atoll(e) : 320LL * 320; }();
  if (env_pc != 0 && np == 2 && (long long)p.H * p.W >= pc_minpix && pc_supported(p)) return launch_pc(st, p);
#endif
  p.tiles_x = ceil_div(p.W, split::TWX);
  p.tiles_y = ceil_div(p.H, split::THY);
  p.nchunks = p.Cin / split::KC;
  const int MT = p.Cout <= 32 ? 1 : 2, MW = 32 * MT;
  const int ncb = ceil_div(p.Cout, MW);
  const long long ntile = (long long)p.tiles_x * p.tiles_y * p.B;
  if (ntile > 0x7fffffffLL) return C2M_ERR_INVALID_ARG;
  static const int env_tpw = [] { const char* e = getenv("C2M_CONV_TPW"); return e ? atoi(e) : 0; }();
  const long long resident = 512;   // two workgroups per CU (75 KiB of LDS, <= 256 registers each)
  long long tpw = 1, best = -1;
  for (long long t = 1; t <= 16; ++t) {
    const long long wgs = ((ntile + t - 1) / t) * ncb;
    const long long cost = ((wgs + resident - 1) / resident) * t;
    if (best < 0 || cost <= best) { best = cost; tpw = t; }
  }
  if (env_tpw > 0) tpw = env_tpw;
  p.tpw = (int)tpw;
  dim3 grid((unsigned)((ntile + tpw - 1) / tpw), ncb);
  int rc;
  if (np == 3) rc = MT == 2 ? launch_split_mode<3, 2>(st, p, grid) : launch_split_mode<3, 1>(st, p, grid);
  else if (np == 2) rc = MT == 2 ? launch_split_mode<2, 2>(st, p, grid) : launch_split_mode<2, 1>(st, p, grid);
  else rc = MT == 2 ? launch_split_mode<1, 2>(st, p, grid) : launch_split_mode<1, 1>(st, p, grid);
  if (rc != C2M_OK) return rc;
  return check_launch();
}

int set_head_stores(int mode) {
  if (mode < -1 || mode > 1) return C2M_ERR_INVALID_ARG;
  g_head_stores = mode;
  return C2M_OK;
}

}  // namespace conv
}  // namespace c2m
