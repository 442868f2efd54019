// conv3x3_wino16.hip -- 3x3 / stride 1 / pad 1 convolution, fp32 in / fp32 out, Winograd F(R,3) ALONG Y on the f16 x 2 pieces
// of conv3x3_split.hip (gfx950 / MI355X only).  R = 4: HALF the matrix instructions of the direct f16 x 2 kernel; R = 2: 2/3.
//
// Same contract as conv3x3_split.hip's channels-last mode (the decoder's residual bodies and offset
// convolutions, ref_restoration_arch.py:140-187, arch_util.py:80-136):   out = act(conv3x3(cat(src0, src1)) + bias) + res1 + res2.
//
// Why along y.  A 1-D Winograd transform over the rows of a column touches ONE pixel column: the lane that fetched the
// (pixel column, 4 channels) pieces of R + 2 consecutive halo rows has everything the input transform needs in its own
// registers -- no cross-lane traffic, and the transformed planes V_t keep the direct kernel's LDS layout [plane][k half]
// [pixel column][8 x f16], so that the three taps ALONG X of a kernel row stay what they are there: the same B operand read
// shifted by 16 bytes.  The output transform is lane-local too (the R + 2 accumulators of an output column live in one lane).
//   y[R] = A^T ( (G g) . (B^T d) ),  per pixel column and channel pair;  B^T is applied in fp32 BEFORE the f16 split
//   (B^T/4 for R = 4: exact scaling, keeps |V| <= 2.5 max|d|), G in float64 before the weight split (4 G for R = 4),
//   the products are the three f16 x 2 products of conv3x3_split.hip (wA.x0 + w1.x0 + 2^-11 wA.x1'), A^T in fp32.
// Error against float64: BELOW the direct f16 x 2 kernel's for R = 2 and at the exact-fp32-MFMA chain's level for R = 4
// (each transform-domain accumulator sums a third / a ninth of the products of a direct accumulator; measured in
// tests/test_conv_gpu.py, predicted by scripts/sim_wino16_numerics.py before the kernel was written).
//
// Mapping: one workgroup = 4 waves, ONE per SIMD (the R + 2 transform-domain accumulators of a 64-cout tile are (R + 2) x 2 x 16
// = 192 (128) registers: the unified 512-entry file at one wave per SIMD); tile = 30 x 4R output pixels x 64 couts:
//   * wave w owns output rows [R w, R w + R) of the tile: accumulators acc[t][mt], t = 0 .. R+1, mt = 0, 1;
//   * the MFMA's 32 pixel columns are the 32 halo columns x0 - 1 .. x0 + 30 shifted by the tap: 30 valid output columns (the
//     34-column halo of a 32-wide tile would leave 16 of 272 load slots to a second, almost empty round of R + 2 loads);
//   * K in chunks of 16 input channels.  Lane (column c, channel quad q, half h) of the workgroup fetches the 2R + 2 halo rows
//     of row groups 2h, 2h + 1 as fp32 (buffer_load_dwordx4, hardware zero fill) TWO chunks ahead into one of two register
//     sets, transforms + splits them ONE chunk ahead (one (row group, t) item per tap of units 0 .. R, riding in the MFMA
//     groups) into the other of two plane buffers [plane][k half][row group][t][32 columns][8 x f16];
//   * a unit = one transform position t with its three taps dx; its weight images U_t[dx] (split once per weight version by
//     conv3x3_relayout_split_kernel, fl = 2 | R << 4) stream by LDS-DMA into a ring of FOUR slots three units ahead, landed
//     TWO units ahead -- so the A operands of a unit's first tap are prefetched during the previous unit's last tap like
//     every other operand (with one wave per SIMD nobody else covers an exposed LDS latency);
//   * per tap: 4 A + 2 B ds_read_b128 (one tap ahead, two register sets) feed 6 MFMAs (3 products x 2 cout tiles);
//   * one barrier per unit; persistent tiles, XCD-aware order; epilogue: A^T, 1/S, bias, activation, residuals, 16-byte stores.
// LDS (R = 4): planes 2 x 48.25 KiB + ring 4 x 12 KiB + bias = 144.75 KiB; R = 2: 2 x 32.25 + 48 = 112.75 KiB.
#include <stdio.h>
#include <stdlib.h>

#include <algorithm>
#include <type_traits>

#include "c2m_common.h"
#include "conv3x3_shared.h"

namespace c2m {
namespace conv {
namespace wino16 {

typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr int KC = 16;            // input channels per chunk = K of v_mfma_f32_32x32x16_f16
constexpr int TWO = 30;           // valid output columns of a tile
constexpr float F16_LO_SCALE = 2048.0f;   // the low piece of an f16 x 2 operand is stored times 2^11 (conv3x3_split.hip)

template <int R>
struct Geo {
  static constexpr int T = R + 2;            // transform positions = units per chunk
  static constexpr int TH = 4 * R;           // output rows of a tile
  static constexpr int NROW = 2 * R + 2;     // halo rows a lane fetches (row groups 2h, 2h + 1)
  static constexpr int NSLAB = 4 * T;        // (row group, t) slabs of 32 columns x 16 B
  // one (plane, k half) array; + 64: (i) the two k halves of a ds_write_b64 group land on disjoint banks (== 64 mod 128),
  // (ii) the dx = 1, 2 shifted reads of the last slab stay inside the allocation (they feed output columns 30, 31: discarded)
  static constexpr int KH = NSLAB * 512 + 64;
  static constexpr int PLB = 4 * KH;         // one plane buffer: [plane 2][k half 2]
  static constexpr float DOMAIN = R == 4 ? 26200.0f : 32760.0f;   // |V| <= 2.5 (2) max |d| must stay below 65520
};
constexpr int MT = 2, MW = 64;
constexpr int NRING = 4;
constexpr int WTAP = 2 * MT * 1024;   // one tap's weight image [image 2][mt][k half][32 rows][16 B]
constexpr int WUNIT = 3 * WTAP;       // unit = one transform position, taps dx = 0, 1, 2
constexpr int NW_W = WUNIT / 1024 / 4;   // LDS-DMA instructions per wave and unit (3)

size_t lds_bytes(int R) { return (size_t)2 * (R == 4 ? Geo<4>::PLB : Geo<2>::PLB) + (size_t)NRING * WUNIT + 256; }

template <int I, int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>());
    static_for<I + 1, N>(f);
  }
}
template <int IMM>
__device__ __forceinline__ void lds_read128(f16x8& d, unsigned addr) {
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(d) : "v"(addr), "n"(IMM) : "memory");
}
template <int IMM>
__device__ __forceinline__ void lds_write64(unsigned addr, const u32x2 v) {
  asm volatile("ds_write_b64 %0, %1 offset:%2" ::"v"(addr), "v"(v), "n"(IMM) : "memory");
}
__device__ __forceinline__ void split2_f16(const f32x4 v, u32x2& p0, u32x2& p1) {
  const f16x4 h0 = __builtin_convertvector(v, f16x4);
  const f32x4 r = (v - __builtin_convertvector(h0, f32x4)) * F16_LO_SCALE;
  const f16x4 h1 = __builtin_convertvector(r, f16x4);
  p0 = __builtin_bit_cast(u32x2, h0);
  p1 = __builtin_bit_cast(u32x2, h1);
}
__device__ __forceinline__ f32x4 fma4(float a, const f32x4 x, const f32x4 y) {   // a * x + y, one rounding per element
  f32x4 r;
#pragma unroll
  for (int e = 0; e < 4; ++e) r[e] = __builtin_fmaf(a, x[e], y[e]);
  return r;
}

template <int R>
__global__ void __launch_bounds__(256, 1) conv3x3_wino16_kernel(Params p) {
  using GE = Geo<R>;
  constexpr int T = GE::T, TH = GE::TH, NROW = GE::NROW, KH = GE::KH, PLB = GE::PLB;
  constexpr int NTAP = 3 * T;            // taps per chunk
  constexpr int NITEM = 2 * T;           // (row group of this lane, t) transform + split items per chunk
  static_assert(NITEM <= 3 * (T - 1) && NROW <= 3 * (T - 1), "items and raw loads fit the taps of units 0 .. T-2");
  extern __shared__ __attribute__((aligned(1024))) char lds[];
  const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)lds;
  const unsigned pl_base = lds0, w_base = lds0 + 2 * PLB, bias_lds = w_base + NRING * WUNIT;

  const int tid = threadIdx.x, l = tid & 63, hi = l >> 5, j = l & 31;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int half = wv >> 1, lx = (wv & 1) * 16 + (l >> 2), q = l & 3;   // loading role: halo column lx, channel quad q, row-group pair
  const int ntile = p.tiles_x * p.tiles_y * p.B;
  const int tile_first = xcd_remap(blockIdx.x, gridDim.x) * p.tpw;
  const int ntl = min(p.tpw, ntile - tile_first);
  const int cb = blockIdx.y;
  const int UT = p.nchunks * T;         // units per tile
  const int G = ntl * p.nchunks;        // chunks of this workgroup
  const int NU = G * T;                 // units of this workgroup

  // ---- weights: unit u of this cout block (WUNIT contiguous bytes) -> ring slot u & 3 by LDS-DMA, wave w moves pieces [3w, 3w + 3)
  const __amdgpu_buffer_rsrc_t wrsrc = make_rsrc(reinterpret_cast<const char*>(p.wr) + (size_t)cb * UT * WUNIT, (unsigned)UT * WUNIT);
  const unsigned wvoff = (wv * NW_W * 64 + l) * 16;
  int wsoff = 0;   // byte offset of the unit the NEXT issue fetches (wraps per tile)
  auto issue_w_piece = [&](unsigned slot_off, int i) __attribute__((always_inline)) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(wrsrc, (__attribute__((address_space(3))) void*)(w_base + slot_off + (unsigned)(wv * NW_W + i) * 1024u), 16,
                                             wvoff, wsoff + i * 1024, 0, 0);
  };
  auto issue_w_done = [&]() __attribute__((always_inline)) {
    wsoff += WUNIT;
    if (wsoff == UT * WUNIT) wsoff = 0;
  };

  // ---- halo rows: lane (lx, q, half) fetches rows 2R*half .. 2R*half + 2R + 1 of the (4R + 2)-row halo, column x0 - 1 + lx
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
  int dma_c = 0;            // chunk (inside its tile) the next issue_in_begin() sets up
  unsigned ivoff[NROW];
  int iy0 = 0, ix0 = 0, in_soff = 0;
  bool in_first = true;
  __amdgpu_buffer_rsrc_t rs0 = make_rsrc(p.src[0].ptr, 0u), rs1 = rs0;
  auto set_source = [&](const Src& S) __attribute__((always_inline)) {
    const int ix = ix0 - 1 + lx;
#pragma unroll
    for (int i = 0; i < NROW; ++i) {
      const int iy = iy0 - 1 + 2 * R * half + i;
      const bool ok = (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
      ivoff[i] = ok ? (unsigned)(iy * S.row_pitch + ix * S.pix_pitch + 4 * q) * 4u : kOOB;
    }
  };
  auto src_rsrc = [&](const Src& S, int b) __attribute__((always_inline)) {
    const unsigned bytes = (unsigned)((p.H - 1) * S.row_pitch + (p.W - 1) * S.pix_pitch + S.C) * 4u;
    return make_rsrc(S.ptr + (long long)b * S.img_pitch, bytes);
  };
  auto issue_in_begin = [&]() __attribute__((always_inline)) {   // the next chunk of the workgroup's stream
    const int c0 = dma_c * KC;
    in_first = c0 < p.src[0].C;
    if (++dma_c == p.nchunks) dma_c = 0;
    if (c0 == 0) {
      iy0 = dma_tc.ty * TH; ix0 = dma_tc.tx * TWO;
      rs0 = src_rsrc(p.src[0], dma_tc.b);
      rs1 = src_rsrc(p.src[1], dma_tc.b);
      tc_next(dma_tc);
      set_source(p.src[0]);
    } else if (c0 == p.src[0].C) {
      set_source(p.src[1]);
    }
    in_soff = (in_first ? c0 : c0 - p.src[0].C) * 4;
  };
  f32x4 raw[2][NROW];   // two register sets: chunk c lives in set c & 1 from chunk c-2 (fetch) to chunk c-1 (transform)
  float amax = 0.0f;
  auto load_row = [&](auto setc, auto ic) __attribute__((always_inline)) {
    constexpr int SET = decltype(setc)::value, I = decltype(ic)::value;
    raw[SET][I] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(in_first ? rs0 : rs1, (int)ivoff[I], in_soff, 0));
  };

  // ---- transform + split item K = a*T + t of the lane's row group a (of its pair) from register set SET -> plane buffer at
  // byte offset pb: V_t = sum_i BT[t][i] * raw[R a + i], two f16 pieces, 8 bytes each at [plane][q >> 1][slab][lx][q & 1]
  const unsigned cwr = pl_base + (q >> 1) * KH + (2 * half * T) * 512 + lx * 16 + (q & 1) * 8;
  f32x4 hold = {0.0f, 0.0f, 0.0f, 0.0f};   // V_2 / V_4, computed with V_1 / V_3 (shared sub-expressions)
  auto item = [&](auto setc, auto kc, unsigned pb) __attribute__((always_inline)) {
    constexpr int SET = decltype(setc)::value, K = decltype(kc)::value, A = K / T, t = K % T;
    const f32x4* d = &raw[SET][R * A];
    f32x4 v;
    if constexpr (R == 4) {
      if constexpr (t == 0) {
        v = fma4(0.25f, d[4], fma4(-1.25f, d[2], d[0]));
#pragma unroll
        for (int i = 0; i < T; ++i)   // (range check: every row of this row group, once)
#pragma unroll
          for (int e = 0; e < 4; ++e) amax = fmaxf(amax, __builtin_fabsf(d[i][e]));
      } else if constexpr (t == 1) {
        const f32x4 P = fma4(0.25f, d[4], -d[2]), Q = fma4(0.25f, d[3], -d[1]);
        v = P + Q;
        hold = P - Q;
      } else if constexpr (t == 3) {
        const f32x4 Rr = (d[4] - d[2]) * 0.25f, Ss = (d[3] - d[1]) * 0.5f;
        v = Rr + Ss;
        hold = Rr - Ss;
      } else if constexpr (t == 5) {
        v = fma4(0.25f, d[5], fma4(-1.25f, d[3], d[1]));
      } else {
        v = hold;
      }
    } else {
      if constexpr (t == 0) {
        v = d[0] - d[2];
#pragma unroll
        for (int i = 0; i < T; ++i)
#pragma unroll
          for (int e = 0; e < 4; ++e) amax = fmaxf(amax, __builtin_fabsf(d[i][e]));
      } else if constexpr (t == 1) {
        v = d[1] + d[2];
      } else if constexpr (t == 2) {
        v = d[2] - d[1];
      } else {
        v = d[1] - d[3];
      }
    }
    u32x2 p0, p1;
    split2_f16(v, p0, p1);
    lds_write64<K * 512>(cwr + pb, p0);
    lds_write64<K * 512 + 2 * KH>(cwr + pb, p1);
  };

  // ---- operands: A = lane (cout row j, k half hi) of the ring slot's tap dx, image pl, cout tile mt;
  //                B = column j + dx of slab (row group wv, t), k half hi, plane pl
  const unsigned abase = w_base + hi * 512 + j * 16;
  const unsigned bbase = pl_base + hi * KH + (wv * T) * 512 + j * 16;
  f16x8 A[2][2][MT], Bq[2][2];   // two operand sets: tap n = 3t + dx multiplies set n & 1
  f16x8 Ad[MT];                  // 2^-11 wA of the current tap
  auto load_a = [&](auto setc, auto dxc, auto kc, unsigned aslot) __attribute__((always_inline)) {
    constexpr int SET = decltype(setc)::value, DX = decltype(dxc)::value, K = decltype(kc)::value;
    lds_read128<DX * WTAP + K * 1024>(A[SET][K / MT][K % MT], aslot);
  };
  auto load_b = [&](auto setc, auto tc, auto dxc, auto plc, unsigned bcur) __attribute__((always_inline)) {
    constexpr int SET = decltype(setc)::value, TT = decltype(tc)::value, DX = decltype(dxc)::value, PL = decltype(plc)::value;
    lds_read128<PL * 2 * KH + TT * 512 + DX * 16>(Bq[SET][PL], bcur);
  };

  float w_sinv = *reinterpret_cast<const float*>(reinterpret_cast<const char*>(p.wr) + (size_t)gridDim.y * UT * WUNIT);   // 1/S behind the images
  if (tid < MW) {
    const int co = cb * MW + tid;
    *(__attribute__((address_space(3))) float*)(bias_lds + tid * 4) = (p.bias && co < p.Cout) ? p.bias[co] : 0.0f;
  }
  f32x16 acc[T][MT];
#pragma unroll
  for (int t = 0; t < T; ++t)
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[t][mt][r] = 0.0f;

  // ------------------------------------------------------------------------------------------------------------------
  // prologue: weights of units 0, 1, 2; chunk 0 fetched into set 0, transformed into plane buffer 0; chunk 1 fetched into set 1
  // ------------------------------------------------------------------------------------------------------------------
#pragma unroll
  for (int u = 0; u < NRING - 1; ++u)
    if (u < NU) {
#pragma unroll
      for (int i = 0; i < NW_W; ++i) issue_w_piece((unsigned)(u * WUNIT), i);
      issue_w_done();
    }
  issue_in_begin();
  static_for<0, NROW>([&](auto ic) __attribute__((always_inline)) { load_row(std::integral_constant<int, 0>(), ic); });
  if (G > 1) {
    issue_in_begin();
    static_for<0, NROW>([&](auto ic) __attribute__((always_inline)) { load_row(std::integral_constant<int, 1>(), ic); });
  }
  static_for<0, NITEM>([&](auto kc) __attribute__((always_inline)) { item(std::integral_constant<int, 0>(), kc, 0u); });
  // first operands: A of (unit 0, dx 0), B of (t 0, dx 0) -- after everything above has landed and been published
  wait_vmcnt<0>();
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  static_for<0, 2 * MT>([&](auto kc) __attribute__((always_inline)) {
    load_a(std::integral_constant<int, 0>(), std::integral_constant<int, 0>(), kc, abase);
  });
  static_for<0, 2>([&](auto plc) __attribute__((always_inline)) {
    load_b(std::integral_constant<int, 0>(), std::integral_constant<int, 0>(), std::integral_constant<int, 0>(), plc, bbase);
  });

  // ------------------------------------------------------------------------------------------------------------------
  // one chunk = T units of three taps.  PAR = chunk parity: plane buffer PAR is multiplied, items transform register set
  // 1 - PAR (chunk gc + 1) into plane buffer 1 - PAR, raw loads of chunk gc + 2 refill set PAR.
  // ------------------------------------------------------------------------------------------------------------------
  auto chunk = [&](auto parc, int gc) __attribute__((always_inline)) {
    constexpr int PAR = decltype(parc)::value;
    const bool has_next = gc + 1 < G, more_in = gc + 2 < G;
    const unsigned bcur = bbase + PAR * PLB, bnext = bbase + (1 - PAR) * PLB;
    if (more_in) issue_in_begin();
    static_for<0, T>([&](auto tc) __attribute__((always_inline)) {
      constexpr int t = decltype(tc)::value;
      const int u = gc * T + t;
      constexpr int slot = (PAR * T + t) & 3;                       // (gc*T + t) & 3: T even
      constexpr unsigned slot_cur = slot * WUNIT, slot_n1 = ((slot + 1) & 3) * WUNIT, slot_n3 = ((slot + 3) & 3) * WUNIT;
      const unsigned aslot = abase + slot_cur;
      const bool do_w = u + NRING - 1 < NU;
      static_for<0, 3>([&](auto dxc) __attribute__((always_inline)) {
        constexpr int dx = decltype(dxc)::value, n = 3 * t + dx;
        constexpr int set = n & 1, nset = set ^ 1;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        static_for<0, 3>([&](auto gcnt) __attribute__((always_inline)) {
          constexpr int g = decltype(gcnt)::value;
          // ---- next tap's operands, three reads per group in groups 0 and 1
          if constexpr (g < 2) {
            static_for<3 * g, 3 * g + 3>([&](auto kc) __attribute__((always_inline)) {
              constexpr int K = decltype(kc)::value;   // 0..3: A images, 4..5: B planes
              if constexpr (dx < 2) {
                if constexpr (K < 4) load_a(std::integral_constant<int, nset>(), std::integral_constant<int, dx + 1>(), kc, aslot);
                else load_b(std::integral_constant<int, nset>(), tc, std::integral_constant<int, dx + 1>(), std::integral_constant<int, K - 4>(), bcur);
              } else {
                // first tap of the next unit: its weights landed two units ago; at the chunk's end the B operand comes from the
                // other plane buffer, complete since the barrier of unit T-2
                if constexpr (K < 4) load_a(std::integral_constant<int, nset>(), std::integral_constant<int, 0>(), kc, abase + slot_n1);
                else if constexpr (t < T - 1) load_b(std::integral_constant<int, nset>(), std::integral_constant<int, t + 1>(), std::integral_constant<int, 0>(), std::integral_constant<int, K - 4>(), bcur);
                else load_b(std::integral_constant<int, nset>(), std::integral_constant<int, 0>(), std::integral_constant<int, 0>(), std::integral_constant<int, K - 4>(), bnext);
              }
            });
          }
          if constexpr (dx == 0) {
            if (do_w) issue_w_piece(slot_n3, g);
          }
          if constexpr (g == 0) {   // wB = 2^-11 wA of this tap (used by group 1)
            const f16x8 sc = {(_Float16)(1.0f / F16_LO_SCALE), (_Float16)(1.0f / F16_LO_SCALE), (_Float16)(1.0f / F16_LO_SCALE), (_Float16)(1.0f / F16_LO_SCALE),
                              (_Float16)(1.0f / F16_LO_SCALE), (_Float16)(1.0f / F16_LO_SCALE), (_Float16)(1.0f / F16_LO_SCALE), (_Float16)(1.0f / F16_LO_SCALE)};
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) Ad[mt] = A[set][0][mt] * sc;
          }
          if constexpr (g == 1) {
            if constexpr (n < NITEM) {
              if (has_next) item(std::integral_constant<int, 1 - PAR>(), std::integral_constant<int, n>(), (unsigned)((1 - PAR) * PLB));
            }
            if constexpr (n < NROW) {
              if (more_in) load_row(std::integral_constant<int, PAR>(), std::integral_constant<int, n>());
            }
          }
          // products: g = 0: w1 . x0, g = 1: (2^-11 wA) . x1', g = 2: wA . x0 (smallest terms first)
#pragma unroll
          for (int mt = 0; mt < MT; ++mt) {
            const f16x8 av = g == 0 ? A[set][1][mt] : (g == 1 ? Ad[mt] : A[set][0][mt]);
            acc[t][mt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(av, Bq[set][g == 1 ? 1 : 0], acc[t][mt], 0, 0, 0);
          }
          __builtin_amdgcn_sched_barrier(0);
        });
        if constexpr (dx == 0) {
          if (do_w) issue_w_done();
        }
      });
      // ---- unit end: own LDS writes done; everything issued BEFORE this unit has landed (the weights of unit u + 2 among
      // it); still in flight may be what this unit issued: W(u+3) x 3 and its raw loads.  Barrier: publishes W(u+2) and
      // (unit T-2) the other plane buffer, frees ring slot u & 3.
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      constexpr int nraw = NROW - 3 * t >= 3 ? 3 : (NROW - 3 * t > 0 ? NROW - 3 * t : 0);
      if (do_w && more_in) wait_vmcnt<NW_W + nraw>();
      else if (do_w) wait_vmcnt<NW_W>();
      else if (more_in) wait_vmcnt<nraw>();
      else wait_vmcnt<0>();
      __builtin_amdgcn_s_barrier();
    });
  };

  for (int it = 0, gc = 0; it < ntl; ++it) {
    for (int c = 0; c < p.nchunks; ++c, ++gc) {
      if (gc & 1) chunk(std::integral_constant<int, 1>(), gc);
      else chunk(std::integral_constant<int, 0>(), gc);
    }
    // ----------------------------------------------------------------------------------------------------------------
    // epilogue of the tile: y = A^T m per (cout, column), then 1/S, bias, activation, residuals, 16-byte stores
    // ----------------------------------------------------------------------------------------------------------------
    const int b = epi_tc.b, y0 = epi_tc.ty * TH, x0 = epi_tc.tx * TWO;
    tc_next(epi_tc);
    const int x = x0 + j;
    const bool xok = j < TWO && x < p.W;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int qd = 0; qd < 4; ++qd) {
        const f32x4 bv = *(const __attribute__((address_space(3))) f32x4*)(bias_lds + (mt * 32 + 8 * qd + 4 * hi) * 4);
        f32x4 o[R];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int r = 4 * qd + e;
          if constexpr (R == 4) {
            const float m0 = acc[0][mt][r], m1 = acc[1][mt][r], m2 = acc[2][mt][r], m3 = acc[3][mt][r], m4 = acc[4][mt][r], m5 = acc[5][mt][r];
            const float s1 = m1 + m2, d1 = m1 - m2, s2 = m3 + m4, d2 = m3 - m4;
            o[0][e] = (m0 + s1) + s2;
            o[1][e] = __builtin_fmaf(2.0f, d2, d1);
            o[2][e] = __builtin_fmaf(4.0f, s2, s1);
            o[3][e] = __builtin_fmaf(8.0f, d2, d1) + m5;
          } else {
            const float m0 = acc[0][mt][r], m1 = acc[1][mt][r], m2 = acc[2][mt][r], m3 = acc[3][mt][r];
            o[0][e] = (m0 + m1) + m2;
            o[1][e] = (m1 - m2) - m3;
          }
        }
        const int co = cb * MW + mt * 32 + 8 * qd + 4 * hi;
#pragma unroll
        for (int r = 0; r < R; ++r) {
          f32x4 v = o[r] * w_sinv + bv;   // (power of two: exact)
          if (p.act == 1) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.0f);
          } else if (p.act == 2) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], v[e] * p.slope);
          }
          const int y = y0 + R * wv + r;
          if (xok && y < p.H && co + 3 < p.Cout) {
            const size_t opix = (size_t)b * p.out_img_pitch + (size_t)y * p.out_row_pitch + (size_t)x * p.out_pix_pitch + co;
            if (p.res1) v += *reinterpret_cast<const f32x4*>(p.res1 + opix);
            if (p.res2) v += *reinterpret_cast<const f32x4*>(p.res2 + opix);
            *reinterpret_cast<f32x4*>(p.out + opix) = v;
          }
        }
      }
#pragma unroll
    for (int t = 0; t < T; ++t)
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][mt][r] = 0.0f;
  }
  if (p.range_flag != nullptr && !(amax < GE::DOMAIN)) *p.range_flag = 1;   // (rare, idempotent store; inf / NaN count)
}

}  // namespace wino16

// p: as filled by c2m_conv3x3_nhwc_f32 (tiles / nchunks / tpw are set here).  R = 4 or 2.
int launch_wino16(hipStream_t st, Params p, int R) {
  p.tiles_x = ceil_div(p.W, wino16::TWO);
  p.tiles_y = ceil_div(p.H, 4 * R);
  p.nchunks = p.Cin / wino16::KC;
  const int ncb = ceil_div(p.Cout, wino16::MW);
  const long long ntile = (long long)p.tiles_x * p.tiles_y * p.B;
  if (ntile > 0x7fffffffLL || (R != 4 && R != 2)) return C2M_ERR_INVALID_ARG;
  static const int env_tpw = [] { const char* e = getenv("C2M_CONV_TPW"); return e ? atoi(e) : 0; }();
  const long long resident = 256;   // one workgroup per CU
  long long tpw = 1, best = -1;
  for (long long t = 1; t <= 16; ++t) {
    const long long wgs = ((ntile + t - 1) / t) * ncb;
    const long long cost = ((wgs + resident - 1) / resident) * t;
    if (best < 0 || cost <= best) { best = cost; tpw = t; }
  }
  if (env_tpw > 0) tpw = env_tpw;
  p.tpw = (int)tpw;
  dim3 grid((unsigned)((ntile + tpw - 1) / tpw), ncb);
  const size_t ldsb = wino16::lds_bytes(R);
  static unsigned long long done4 = 0, done2 = 0;
  int rc;
  if (R == 4) {
    if ((rc = ensure_dynamic_lds(reinterpret_cast<const void*>(&wino16::conv3x3_wino16_kernel<4>), ldsb, done4)) != C2M_OK) return rc;
    hipLaunchKernelGGL(wino16::conv3x3_wino16_kernel<4>, grid, dim3(256), ldsb, st, p);
  } else {
    if ((rc = ensure_dynamic_lds(reinterpret_cast<const void*>(&wino16::conv3x3_wino16_kernel<2>), ldsb, done2)) != C2M_OK) return rc;
    hipLaunchKernelGGL(wino16::conv3x3_wino16_kernel<2>, grid, dim3(256), ldsb, st, p);
  }
  return check_launch();
}

}  // namespace conv
}  // namespace c2m
