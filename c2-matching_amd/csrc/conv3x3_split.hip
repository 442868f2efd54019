// conv3x3_split.hip -- 3x3 / stride 1 / pad 1 convolution, fp32 in / fp32 out, on the BF16 matrix pipe of gfx950 (MI355X).
//
// Same contract as conv3x3.hip (decoder stack ref_restoration_arch.py:140-187, arch_util.py:80-136,
// DCN offset/mask head dcn_v2.py:229-245):   out = act( conv3x3( cat(src0, src1) ) + bias ) + res1 + res2
// on channels-last fp32 tensors.  What changes is the arithmetic underneath.  On CDNA4 the fp32 MFMA
// (v_mfma_f32_32x32x2_f32) runs at the fp32 VECTOR rate (157 TF, and every VALU instruction next to it costs matrix
// time); v_mfma_f32_32x32x16_bf16 is 16x faster and has its own pipe.  So every fp32 operand is split EXACTLY into three
// bf16 pieces
//       x = x0 + x1 + x2,   x0 = top 16 bits of x,  x1 = top 16 bits of (x - x0),  x2 = x - x0 - x1   (8 + 8 + 8 mantissa bits)
// and the product sum is taken over the six piece pairs whose magnitude is >= 2^-16 of the leading one,
//       w.x ~= w0x0 + (w0x1 + w1x0) + (w1x1 + w0x2 + w2x0)            (dropped: w1x2, w2x1, w2x2 <= 2^-24 |w||x|)
// accumulated in fp32 by the MFMA: 6/16 of the fp32-MFMA time for an error BELOW that of an fp32 fmaf chain (measured
// on K = 576: 6e-8 of the result scale from the dropped terms against 5e-7 for fp32 accumulation itself).  NP = 1 keeps
// only w0x0 with round-to-nearest pieces: the plain bf16 convolution of BASELINE configs[4] (bf16 inference), 1/6 of the
// matrix work again.
//
// Mapping (one workgroup = 4 waves = one wave per SIMD, 32 x 8 output pixels x MW = 32*MT output channels):
//   * wave w owns pixel rows 2w, 2w+1 of the tile: NT = 2 pixel tiles x MT channel tiles = 2*MT accumulators f32x16;
//   * K is swept in chunks of 16 input channels (= K of one MFMA).  The zero-padded 34 x 10 halo tile of a chunk arrives
//     as fp32 by LDS-DMA (buffer_load ... lds, hardware zero fill outside the image) into `raw`; every wave then splits
//     the pieces it fetched itself (no barrier between DMA and split) into NP bf16 planes laid out [plane][k half][pixel]
//     [8 bf16]: a B operand (8 channels of one pixel) is one ds_read_b128, 16 consecutive lanes read 256 contiguous
//     bytes for any tap shift (conflict-free without a swizzle).  The split of chunk c+1 (22 VALU + 4 LDS instructions
//     per 4 channels x pixel) is interleaved with the MFMAs of chunk c -- bf16 MFMAs leave ~5 issue slots per instruction
//     free -- and the planes are double buffered;
//   * weights are split once per weight version on the host side of the call (conv3x3_relayout_split_kernel) into
//     ready-made LDS images [cout block][chunk][dy][dx][plane][mt][k half][32 rows][8 bf16]; a unit = one kernel row
//     (3 taps, 3*NP*MT KiB) streams by linear DMA into a ring of 3 slots, two units ahead of its use;
//   * per tap: NP*MT A reads + NP*NT B reads (ds_read_b128) feed NPROD*MT*NT MFMAs (24 for the fp32 flavour), operands
//     fetched one tap ahead into a second register set; one barrier per unit (72 MFMAs);
//   * persistent tiles, XCD-aware tile order, epilogue in registers with the store flavours of conv3x3.hip (channels-last
//     (+ residuals), PixelShuffle(2), planar NCHW, DCN offset/mask head) plus ReLU + MaxPool2d(2,2) (both rows of a
//     pooling window live in one lane, the horizontal neighbour one lane over).
// LDS: raw 24 KiB + planes 2 x 36.75 KiB + weight ring 3 x 18 KiB + 1.25 KiB = 152.75 KiB (NP = 3, MT = 2).
#include <stdio.h>
#include <stdlib.h>

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
constexpr int HALFB = NRAW * 256 + 128;         // one (plane, k half) slab: 384 pixels x 16 B + 128 (== 128 mod 256: the
                                                // split's 8-byte stores of a half-wave then cover all 64 banks once)
static_assert(HALFB % 256 == 128, "bank phase of the second k half");

template <int NP> struct Products;
template <> struct Products<1> { static constexpr int N = 1; static constexpr int W[1] = {0}; static constexpr int X[1] = {0}; };
template <> struct Products<3> {
  static constexpr int N = 6;   // smallest terms first, the leading product last
  static constexpr int W[6] = {2, 0, 1, 1, 0, 0};
  static constexpr int X[6] = {0, 2, 1, 0, 1, 0};
};

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
__global__ void __launch_bounds__(256) conv3x3_relayout_split_kernel(const float* __restrict__ w, int Cin, int Cout, int NP, int MT,
                                                                      long long total, unsigned short* __restrict__ wr) {
  const long long e0 = (long long)blockIdx.x * 256 + threadIdx.x;
  if (e0 >= total) return;
  const int e = (int)(e0 & 7), row = (int)((e0 >> 3) & 31), half = (int)((e0 >> 8) & 1);
  long long r = e0 >> 9;
  const int mt = (int)(r % MT); r /= MT;
  const int pl = (int)(r % NP); r /= NP;
  const int dx = (int)(r % 3); r /= 3;
  const int dy = (int)(r % 3); r /= 3;
  const int nch = Cin / KC;
  const int chunk = (int)(r % nch);
  const int cb = (int)(r / nch);
  const int co = (cb * MT + mt) * 32 + row, ci = chunk * KC + 8 * half + e;
  const float v = co < Cout ? w[((size_t)co * Cin + ci) * 9 + dy * 3 + dx] : 0.0f;
  wr[e0] = bf16_piece(v, pl, NP == 1);
}

template <int NP, int MT, int MODE>
__global__ void __launch_bounds__(256, 1) conv3x3_split_kernel(Params p) {
  constexpr int NT = 2;
  constexpr int MW = 32 * MT;
  using PR = Products<NP>;
  constexpr int PLB = NP * 2 * HALFB;           // bytes of one plane buffer
  constexpr int WTAP = NP * MT * 1024;          // one tap's weight image: [plane][mt][half][32 rows][16 B]
  constexpr int WUNIT = 3 * WTAP;               // unit = one kernel row
  constexpr int NWI = WUNIT / 1024;             // DMA instructions per unit
  constexpr int NW_W = (NWI + 3) / 4;           // per wave (the last wave pads with dummies: uniform vmcnt counts)
  extern __shared__ __attribute__((aligned(1024))) char lds[];
  // [raw | planes x2 | weight ring x3 | dummy 1 KiB | bias MW floats]
  const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)lds;
  const unsigned raw_base = lds0, pl_base = lds0 + RAW_BYTES, w_base = pl_base + 2 * PLB, dummy = w_base + 3 * WUNIT,
                 bias_lds = dummy + 1024;

  const int tid = threadIdx.x, l = tid & 63, hi = l >> 5, j = l & 31;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ntile = p.tiles_x * p.tiles_y * p.B;
  const int tile_first = xcd_remap(blockIdx.x, gridDim.x) * p.tpw;
  const int ntl = min(p.tpw, ntile - tile_first);
  const int cb = blockIdx.y;
  const int UT = p.nchunks * 3;         // units per tile
  const int G = ntl * p.nchunks;        // chunks of this workgroup
  const int T = G * 3;                  // units of this workgroup

  // ---- weights: unit u of this cout block = WUNIT contiguous bytes; wave w moves instructions [w*NW_W, (w+1)*NW_W)
  const __amdgpu_buffer_rsrc_t wrsrc = make_rsrc(reinterpret_cast<const char*>(p.wr) + (size_t)cb * UT * WUNIT, (unsigned)UT * WUNIT);
  const unsigned wvoff = (wv * NW_W * 64 + l) * 16;
  int wsoff = 0;
  auto issue_w = [&](int slot) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < NW_W; ++i) {
      const int n = wv * NW_W + i;
      const unsigned dst = n < NWI ? w_base + slot * WUNIT + n * 1024 : dummy;
      // (beyond the image: reads the next unit / zeros past the end of the buffer, lands in the dummy page)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(wrsrc, (__attribute__((address_space(3))) void*)dst, 16, wvoff, wsoff + i * 1024, 0, 0);
    }
    wsoff += WUNIT;
    if (wsoff == UT * WUNIT) wsoff = 0;
  };

  // ---- halo tile: instruction n = wave + 4 * slot covers pieces [64n, 64n + 64); piece P = (pixel P >> 2, channel quad P & 3)
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
  int dma_c = 0;   // chunk (inside its tile) the next issue_in() call fetches
  unsigned ivoff[NRAW_W];
  int ib = 0, iy0 = 0, ix0 = 0;
  __amdgpu_buffer_rsrc_t rs0, rs1;
  int slotc[NRAW_W];   // ry | rx << 8 | quad << 16 | valid << 24
#pragma unroll
  for (int sl = 0; sl < NRAW_W; ++sl) {
    const int n = wv + 4 * sl, pix = 16 * n + (l >> 2);
    const int ry = pix / HWc, rx = pix - ry * HWc;
    slotc[sl] = ry | (rx << 8) | ((l & 3) << 16) | (pix < NPIX ? (1 << 24) : 0);
  }
  auto set_source = [&](const Src& S) __attribute__((always_inline)) {
#pragma unroll
    for (int sl = 0; sl < NRAW_W; ++sl) {
      const int c = slotc[sl];
      const int iy = iy0 - 1 + (c & 0xff), ix = ix0 - 1 + ((c >> 8) & 0xff);
      const bool ok = (c >> 24) != 0 && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
      ivoff[sl] = ok ? (unsigned)(iy * S.row_pitch + ix * S.pix_pitch + 4 * ((c >> 16) & 3)) * 4u : kOOB;
    }
  };
  auto src_rsrc = [&](const Src& S, int b) __attribute__((always_inline)) {
    const unsigned bytes = (unsigned)((p.H - 1) * S.row_pitch + (p.W - 1) * S.pix_pitch + S.C) * 4u;
    return make_rsrc(S.ptr + (long long)b * S.img_pitch, bytes);
  };
  auto issue_in = [&]() __attribute__((always_inline)) {   // the next chunk of the workgroup's stream -> raw
    const int c0 = dma_c * KC;
    const bool first = c0 < p.src[0].C;
    if (++dma_c == p.nchunks) dma_c = 0;
    if (c0 == 0) {
      ib = dma_tc.b; iy0 = dma_tc.ty * THY; ix0 = dma_tc.tx * TWX;
      tc_next(dma_tc);
      rs0 = src_rsrc(p.src[0], ib);
      rs1 = src_rsrc(p.src[1], ib);
      set_source(p.src[0]);
    } else if (c0 == p.src[0].C) {
      set_source(p.src[1]);
    }
    const int soff = (first ? c0 : c0 - p.src[0].C) * 4;
#pragma unroll
    for (int sl = 0; sl < NRAW_W; ++sl) {
      const int n = wv + 4 * sl;
      const unsigned dst = raw_base + n * 1024;
      if (first) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs0, (__attribute__((address_space(3))) void*)dst, 16, ivoff[sl], soff, 0, 0);
      else __builtin_amdgcn_raw_ptr_buffer_load_lds(rs1, (__attribute__((address_space(3))) void*)dst, 16, ivoff[sl], soff, 0, 0);
    }
  };

  // ---- split of the wave's own raw pieces into the bf16 planes of buffer `nb`.  Round r = DMA slot r of this wave:
  // instruction n = wv + 4r, piece 64n + l = (pixel 16n + (l >> 2), quad q = l & 3) -> plane slab (q >> 1), 8 bytes at
  // pixel*16 + (q & 1)*8.  Software-pipelined: the raw piece of round r+1 is read while round r is being split.
  const unsigned craw = raw_base + wv * 1024 + l * 16;                                       // + r * 4096
  const unsigned cdst = pl_base + ((l >> 1) & 1) * HALFB + (wv * 16 + (l >> 2)) * 16 + (l & 1) * 8;   // + r * 1024 + plane * 2*HALFB + nb * PLB
  auto conv_load = [&](int R) __attribute__((always_inline)) {
    return *(const __attribute__((address_space(3))) f32x4*)(craw + R * 4096);
  };
  auto conv_store = [&](int R, unsigned nb_off, const f32x4 v) __attribute__((always_inline)) {
    if constexpr (NP == 3) {
      u32x2 q0, q1, q2;
      split3(v, q0, q1, q2);
      *(__attribute__((address_space(3))) u32x2*)(cdst + nb_off + R * 1024) = q0;
      *(__attribute__((address_space(3))) u32x2*)(cdst + nb_off + R * 1024 + 2 * HALFB) = q1;
      *(__attribute__((address_space(3))) u32x2*)(cdst + nb_off + R * 1024 + 4 * HALFB) = q2;
    } else {
      typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
      bf16x4 h;
#pragma unroll
      for (int i = 0; i < 4; ++i) h[i] = (__bf16)v[i];   // round to nearest even
      *(__attribute__((address_space(3))) bf16x4*)(cdst + nb_off + R * 1024) = h;
    }
  };

  // ---- operand addresses
  // A: lane (cout row j, k half hi) of ring slot dy, tap dx, plane pl, channel tile mt: w_base + dy*WUNIT + dx*WTAP + (pl*MT+mt)*1024
  const unsigned abase = w_base + hi * 512 + j * 16;
  // B: pixel (row 2wv + nt + dy, column j + dx) of the halo tile, k half hi, plane pl, buffer nb
  const unsigned bbase = pl_base + hi * HALFB + (2 * wv * HWc + j) * 16;
  // three operand sets, one per kernel column dx: tap (dy, dx) multiplies set dx while set (dx + 1) % 3 is being fetched
  bf16x8 A[3][NP][MT], Bq[3][NP][NT];
  auto load_tap = [&](int dy, int dx, unsigned nb_off) __attribute__((always_inline)) {
#pragma unroll
    for (int pl = 0; pl < NP; ++pl) {
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
        A[dx][pl][mt] = *(const __attribute__((address_space(3))) bf16x8*)(abase + dy * WUNIT + dx * WTAP + (pl * MT + mt) * 1024);
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
        Bq[dx][pl][nt] = *(const __attribute__((address_space(3))) bf16x8*)(bbase + nb_off + pl * 2 * HALFB + ((nt + dy) * HWc + dx) * 16);
    }
  };

  const int co_lane = cb * MW + 4 * hi;
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

  // ------------------------------------------------------------------------------------------------------------------
  // prologue: chunk 0 is fetched, split and published before the first MFMA
  // ------------------------------------------------------------------------------------------------------------------
  issue_in();
  issue_w(0);
  issue_w(1);
  issue_w(2);
  wait_vmcnt<0>();
#pragma unroll
  for (int R = 0; R < NRAW_W; ++R) conv_store(R, 0u, conv_load(R));
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  if (G > 1) issue_in();   // the raw pieces of a wave are private to it: no barrier needed before they are overwritten
  __builtin_amdgcn_s_barrier();
  load_tap(0, 0, 0u);
  if (G > 1) wait_vmcnt<0>();   // raw(1): once per workgroup
  f32x4 rawv = conv_load(0);

  f32x4 res4[MT][NT][4];
  for (int it = 0, gc = 0; it < ntl; ++it) {
    for (int c = 0; c < p.nchunks; ++c, ++gc) {
      const unsigned nb_cur = (gc & 1) ? PLB : 0u, nb_nxt = PLB - nb_cur;
      const bool last_chunk = c == p.nchunks - 1;
#pragma unroll
      for (int dy = 0; dy < 3; ++dy) {
        const int u = 3 * gc + dy;
        if (MODE == 0 && dy == 2 && last_chunk) {   // residuals: their latency runs under the last unit's MFMAs
          const int b = epi_tc.b, y0 = epi_tc.ty * THY, x0 = epi_tc.tx * TWX;
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) {
            const int y = y0 + 2 * wv + nt, x = x0 + j;
            const bool pok = y < p.H && x < p.W;
            const size_t opix = (size_t)b * p.out_img_pitch + (size_t)y * p.out_row_pitch + (size_t)x * p.out_pix_pitch;
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
              for (int qd = 0; qd < 4; ++qd) {
                f32x4 rv = {0.0f, 0.0f, 0.0f, 0.0f};
                if (pok && co_lane + mt * 32 + 8 * qd + 3 < p.Cout) {
                  if (p.res1) rv = *reinterpret_cast<const f32x4*>(p.res1 + opix + co_lane + mt * 32 + 8 * qd);
                  if (p.res2) rv += *reinterpret_cast<const f32x4*>(p.res2 + opix + co_lane + mt * 32 + 8 * qd);
                }
                res4[mt][nt][qd] = rv;
              }
          }
        }
#pragma unroll
        for (int dx = 0; dx < 3; ++dx) {
          const int t = 3 * dy + dx;
          // the split of the NEXT chunk rides on units 0 and 1 (one round per tap; published by the barrier that ends
          // unit 1).  Past the end of the stream it splits stale bytes into a buffer nobody reads: branch-free.
          if (dy < 2) {
            const f32x4 v = rawv;
            if (t < 5) rawv = conv_load(t + 1);
            conv_store(t, nb_nxt, v);
          }
          // operands of the next tap (the last tap of a chunk: tap 0 of the next chunk, from the other plane buffer)
          if (dx < 2) load_tap(dy, dx + 1, nb_cur);
          else if (dy < 2) load_tap(dy + 1, 0, nb_cur);
          else load_tap(0, 0, nb_nxt);
#pragma unroll
          for (int pr = 0; pr < PR::N; ++pr)
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
              for (int nt = 0; nt < NT; ++nt)
                acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[dx][PR::W[pr]][mt], Bq[dx][PR::X[pr]][nt], acc[mt][nt], 0, 0, 0);
          // spread the tap's LDS traffic and the split's VALU over the MFMA issue gaps
#pragma unroll
          for (int k = 0; k < PR::N * MT * NT; ++k) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);   // MFMA
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);   // DS read
            __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);   // VALU
            __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);   // DS write
          }
          __builtin_amdgcn_sched_barrier(0);
        }
        if (u + 1 < T) {
          // own LDS stores (the split) and own DMAs (W(u+2), raw) have landed; then everybody's
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
          wait_vmcnt<0>();
          __builtin_amdgcn_s_barrier();
          if (u + 3 < T) issue_w(dy);                       // unit u+3 = same kernel row of the next chunk -> same ring slot
          if (dy == 1 && gc + 2 < G) issue_in();            // raw was consumed by the rounds of units 0 and 1
          if (dy == 2) rawv = conv_load(0);                 // first piece of the chunk after next (landed: vmcnt(0) above)
        }
      }
    }
    // ----------------------------------------------------------------------------------------------------------------
    // epilogue of the tile
    // ----------------------------------------------------------------------------------------------------------------
    const int b = epi_tc.b, y0 = epi_tc.ty * THY, x0 = epi_tc.tx * TWX;
    tc_next(epi_tc);
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int qd = 0; qd < 4; ++qd) {
        const f32x4 bv = *(const __attribute__((address_space(3))) f32x4*)(bias_lds + (mt * 32 + 8 * qd + 4 * hi) * 4);
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
          for (int e = 0; e < 4; ++e) acc[mt][nt][4 * qd + e] += bv[e];
      }
    if constexpr (MODE == 3) {
      float asum = 0.0f;
      const HeadOut ho = head_out(p, b);
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        const int y = y0 + 2 * wv + nt, x = x0 + j;
        const bool pok = y < p.H && x < p.W;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
          for (int qd = 0; qd < 4; ++qd) {
            const int col = co_lane + mt * 32 + 8 * qd;   // channel inside this launch's slice
            if (col >= p.Cout || !pok) continue;
            f32x4 v;
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = acc[mt][nt][4 * qd + e];
            asum += dcn_head_store(p, ho, b, y, x, col - 4 * hi, 4 * hi, v);
          }
      }
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
        float* ob = p.out + (size_t)b * p.out_img_pitch + (size_t)yo * p.out_row_pitch + (size_t)xo * p.out_pix_pitch + co_lane;
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
            if (pok && co_lane + mt * 32 + 8 * qd + 3 < p.Cout) *reinterpret_cast<f32x4*>(ob + mt * 32 + 8 * qd) = v;
          }
      } else {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
          const int y = y0 + 2 * wv + nt, x = x0 + j;
          const bool pok = y < p.H && x < p.W;
          if (!pok) continue;
          if constexpr (MODE == 0) {
            const size_t opix = (size_t)b * p.out_img_pitch + (size_t)y * p.out_row_pitch + (size_t)x * p.out_pix_pitch;
            float* ob = p.out + opix + co_lane;
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
              for (int qd = 0; qd < 4; ++qd) {
                const int co = co_lane + mt * 32 + 8 * qd;
                f32x4 v;
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = acc[mt][nt][4 * qd + e];
                if (co + 3 < p.Cout) {
                  v += res4[mt][nt][qd];
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
                        (size_t)(2 * x) * p.out_pix_pitch + (co_lane >> 2);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
              for (int qd = 0; qd < 4; ++qd)
                if (co_lane + mt * 32 + 8 * qd < p.Cout) {
#pragma unroll
                  for (int e = 0; e < 4; ++e)
                    ob[(size_t)(e >> 1) * p.out_row_pitch + (size_t)(e & 1) * p.out_pix_pitch + mt * 8 + 2 * qd] = acc[mt][nt][4 * qd + e];
                }
          } else {
            const size_t HWs = (size_t)p.H * p.W;
            float* ob = p.out + ((size_t)b * p.Cout + co_lane) * HWs + (size_t)y * p.W + x;
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
              for (int r = 0; r < 16; ++r) {
                const int cr = mt * 32 + 8 * (r >> 2) + (r & 3);
                if (co_lane + cr < p.Cout) ob[(size_t)cr * HWs] = acc[mt][nt][r];
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

size_t split_relayout_bytes(int Cin, int Cout, int np) {
  if (Cin <= 0 || Cout <= 0 || Cin % split::KC != 0 || (np != 1 && np != 3)) return 0;
  const int MT = Cout <= 32 ? 1 : 2, ncb = (Cout + 32 * MT - 1) / (32 * MT);
  return (size_t)ncb * (Cin / split::KC) * 9 * np * MT * 1024;
}

int split_relayout(hipStream_t st, const float* weight, int Cin, int Cout, int np, void* wr) {
  const size_t bytes = split_relayout_bytes(Cin, Cout, np);
  if (bytes == 0) return C2M_ERR_UNSUPPORTED;
  const long long total = (long long)(bytes / 2);
  hipLaunchKernelGGL(split::conv3x3_relayout_split_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, weight, Cin,
                     Cout, np, Cout <= 32 ? 1 : 2, total, reinterpret_cast<unsigned short*>(wr));
  return check_launch();
}

template <int NP, int MT>
static int launch_split_mode(hipStream_t st, const Params& p, dim3 grid) {
  constexpr size_t ldsb = split::RAW_BYTES + 2 * (size_t)(NP * 2 * split::HALFB) + 3 * (size_t)(3 * NP * MT * 1024) + 1024 + 256;
  static unsigned long long done[5] = {};
  int rc = C2M_OK;
  auto go = [&](auto kern, unsigned long long& dn) {
    if ((rc = ensure_dynamic_lds(reinterpret_cast<const void*>(kern), ldsb, dn)) == C2M_OK)
      hipLaunchKernelGGL(kern, grid, dim3(256), ldsb, st, p);
  };
  switch (p.out_mode) {
    case 0: go(&split::conv3x3_split_kernel<NP, MT, 0>, done[0]); break;
    case 1: go(&split::conv3x3_split_kernel<NP, MT, 1>, done[1]); break;
    case 2: go(&split::conv3x3_split_kernel<NP, MT, 2>, done[2]); break;
    case 3: go(&split::conv3x3_split_kernel<NP, MT, 3>, done[3]); break;
    default: go(&split::conv3x3_split_kernel<NP, MT, 4>, done[4]); break;
  }
  return rc;
}

// p: as filled by c2m_conv3x3_nhwc_f32 (tiles / nchunks / tpw are set here)
int launch_split(hipStream_t st, Params p, int np) {
  p.tiles_x = ceil_div(p.W, split::TWX);
  p.tiles_y = ceil_div(p.H, split::THY);
  p.nchunks = p.Cin / split::KC;
  const int MT = p.Cout <= 32 ? 1 : 2, MW = 32 * MT;
  const int ncb = ceil_div(p.Cout, MW);
  const long long ntile = (long long)p.tiles_x * p.tiles_y * p.B;
  if (ntile > 0x7fffffffLL) return C2M_ERR_INVALID_ARG;
  static const int env_tpw = [] { const char* e = getenv("C2M_CONV_TPW"); return e ? atoi(e) : 0; }();
  const long long resident = 256;   // one workgroup per CU
  long long tpw = 1, best = -1;
  for (long long t = 1; t <= 10; ++t) {
    const long long wgs = ((ntile + t - 1) / t) * ncb;
    const long long cost = ((wgs + resident - 1) / resident) * t;
    if (best < 0 || cost <= best) { best = cost; tpw = t; }
  }
  if (env_tpw > 0) tpw = env_tpw;
  p.tpw = (int)tpw;
  dim3 grid((unsigned)((ntile + tpw - 1) / tpw), ncb);
  int rc;
  if (np == 3) rc = MT == 2 ? launch_split_mode<3, 2>(st, p, grid) : launch_split_mode<3, 1>(st, p, grid);
  else rc = MT == 2 ? launch_split_mode<1, 2>(st, p, grid) : launch_split_mode<1, 1>(st, p, grid);
  if (rc != C2M_OK) return rc;
  return check_launch();
}

}  // namespace conv
}  // namespace c2m
