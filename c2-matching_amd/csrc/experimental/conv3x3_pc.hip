// conv3x3_pc.hip -- the f16 x 2 split convolution (conv3x3_split.hip, flavour 2) with LOADER and MATRIX waves (round 5,
// DESIGN.md 6.10): the same arithmetic, weight images and plane layout, one workgroup of EIGHT waves per CU, specialised:
//   waves 0..3  ("matrix")  one per SIMD: 4 pixel rows x 32 pixels x 64 couts each (NT = 4, MT = 2: 8 accumulator tiles), nothing
//               in their instruction stream but ds_read_b128, the 8 v_pk_mul_f16 of a tap's wB, MFMAs, one barrier per chunk --
//               and the tile's epilogue;
//   waves 4..7  ("loader")  the SIMDs' second waves: halo tile of the next chunk (buffer_load_dwordx4 -> registers -> two f16
//               planes in LDS), the next chunk's 36 KiB of weight images (LDS-DMA), the domain check.
// Why: conv3x3_split_kernel's cycle ablations put 32 % of its cycles on halo loads and stores although both are
// asynchronous -- every wave carries them IN ORDER between its MFMAs -- while the round-5 micro-benchmarks
// (scripts/ubench/mfma_vmem_share.hip) show that a vector-memory instruction costs the OTHER wave of its SIMD ~5 ns against
// ~30 ns for the wave that issues it.  Scope of this first version: channels-last output (mode 0), fp32 tensors, one source,
// Cout = 64, Cin % 16 == 0 -- the residual bodies' 64 -> 64 layers, 2/3 of the convolution family's time.  Opt-in
// ($C2M_CONV_PC=1) until measured.
//
// Per chunk of 16 input channels (global chunk index g of the workgroup's stream of tiles), ONE barrier:
//   loader, step g:  issue W(g+1) -> weight buffer (g+1)&1 (9 DMA pieces per wave);  vmcnt(9): raw(g+1) has landed;  split it into
//                    plane buffer (g+1)&1;  issue raw(g+2) (10 loads, zero-record descriptor past the stream's end);  vmcnt(10):
//                    W(g+1) has landed;  lgkmcnt(0);  barrier
//   matrix, step g:  9 taps x 4 pixel rows x 6 MFMAs on plane buffer g&1 / weight buffer g&1;  barrier;  after a tile's last
//                    chunk: epilogue (the loaders run ahead meanwhile and wait at the next barrier)
// The barrier at the end of step g publishes planes / weights of chunk g+1 and frees those of chunk g.
// LDS: 2 x 38.25 KiB planes (34 x 18 halo pixels x 16 channels x 2 pieces) + 2 x 36 KiB weights + dummy + bias = 149.75 KiB.
#include <stdio.h>
#include <stdlib.h>

#include <type_traits>

#include "c2m_common.h"
#include "conv3x3_shared.h"

namespace c2m {
namespace conv {
namespace pc {

typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

constexpr int KC = 16;                       // input channels per chunk = K of v_mfma_f32_32x32x16_f16
constexpr int TWX = 32, THY = 16;            // pixel tile of a workgroup
constexpr int HWc = TWX + 2, HHr = THY + 2;  // halo tile 34 x 18
constexpr int NPIX = HWc * HHr;              // 612
constexpr int HALFB = NPIX * 16;             // one (plane, k half) slab
static_assert(HALFB % 128 == 64, "bank phase of the second k half (conv3x3_split.hip)");
constexpr int PLB = 4 * HALFB;               // two planes x two k halves
constexpr int MT = 2, NT = 4, MW = 64;
constexpr int WTAP = 2 * MT * 1024;          // [image][mt][k half][32 rows][16 B]
constexpr int WUNIT = 3 * WTAP, WCHUNK = 3 * WUNIT;
constexpr int NRAW = (NPIX * 4 + 255) / 256; // 10 load rounds per loader lane (piece = (pixel, 4 fp32 channels))
constexpr int NWP = WCHUNK / 1024 / 4;       // 9 DMA pieces per loader wave and chunk
static_assert(NRAW == 10 && NWP == 9, "wait counts below");
constexpr size_t LDS_BYTES = 2 * (size_t)PLB + 2 * (size_t)WCHUNK + 1024 + 256;
constexpr float F16_LO_SCALE = 2048.0f;

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
// NOP: the first load after its descriptor / scalar offset were (possibly) written by v_readfirstlane -- the wait states between
// a VALU write of an SGPR and a vector-memory read of it go INSIDE the asm string (hipcc pads nothing there)
template <bool NOP>
__device__ __forceinline__ void buf_load128f(f32x4& d, unsigned voff, const i32x4 rsrc, int soff) {
  if constexpr (NOP) asm volatile("s_nop 4\n\tbuffer_load_dwordx4 %0, %1, %2, %3 offen" : "=v"(d) : "v"(voff), "s"(rsrc), "s"(soff) : "memory");
  else asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen" : "=v"(d) : "v"(voff), "s"(rsrc), "s"(soff) : "memory");
}
// p0 = rne_f16(v), p1 = rne_f16(2^11 (v - p0))   (conv3x3_split.hip: split2_f16)
__device__ __forceinline__ void split2_f16(const f32x4 v, u32x2& p0, u32x2& p1) {
  const f16x4 h0 = __builtin_convertvector(v, f16x4);
  const f32x4 r = (v - __builtin_convertvector(h0, f32x4)) * F16_LO_SCALE;
  const f16x4 h1 = __builtin_convertvector(r, f16x4);
  p0 = __builtin_bit_cast(u32x2, h0);
  p1 = __builtin_bit_cast(u32x2, h1);
}

// C2M_PC_ABL (compile-time, measurement builds only -- results are wrong): 1 one store of a wave's 32 per tile, 2 the loaders only
// keep the barriers after the prologue (no loads, splits, DMA), 4 no per-step lgkmcnt waits, 8 no MFMAs, 16 no operand reads
#ifndef C2M_PC_ABL
#define C2M_PC_ABL 0
#endif
__global__ void __launch_bounds__(512, 2) conv3x3_pc_kernel(Params p) {
  extern __shared__ __attribute__((aligned(1024))) char lds[];
  const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)lds;
  const unsigned pl_base = lds0, w_base = lds0 + 2 * PLB, dummy = w_base + 2 * WCHUNK, bias_lds = dummy + 1024;

  const int tid = threadIdx.x, l = tid & 63, hi = l >> 5, j = l & 31;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ntile = p.tiles_x * p.tiles_y * p.B;
  const int tile_first = xcd_remap(blockIdx.x, gridDim.x) * p.tpw;
  const int ntl = min(p.tpw, ntile - tile_first);
  const int nch = p.nchunks;
  const int G = ntl * nch;   // chunks of this workgroup's stream
  const int tpi = p.tiles_x * p.tiles_y;

  if (tid < MW) *(__attribute__((address_space(3))) float*)(bias_lds + tid * 4) = p.bias ? p.bias[tid] : 0.0f;

  if (wv >= 4) {
    // =================================================================================================================
    // loader waves
    // =================================================================================================================
    const int pw = wv - 4, pl = pw * 64 + l;   // loader lane 0..255
    const int quad = l & 3;
    int ry[NRAW], rx[NRAW];
#pragma unroll
    for (int r = 0; r < NRAW; ++r) {
      const int pix = 64 * r + (pl >> 2);
      ry[r] = pix / HWc;
      rx[r] = pix - ry[r] * HWc;
    }
    const bool last_ok = 64 * (NRAW - 1) + (pl >> 2) < NPIX;   // the last round reaches beyond the 612 pixels of a slab
    // piece (pixel, quad) -> plane slab (quad >> 1), 8 bytes at pixel * 16 + (quad & 1) * 8;  + round * 1024, + plane * 2 * HALFB
    const unsigned cdst = pl_base + ((quad >> 1) & 1) * HALFB + (unsigned)(pl >> 2) * 16 + (quad & 1) * 8;
    const __amdgpu_buffer_rsrc_t wrsrc = make_rsrc(p.wr, (unsigned)nch * WCHUNK);
    const unsigned wvoff = (unsigned)(pw * NWP * 64 + l) * 16;
    const Src& S = p.src[0];
    const unsigned src_bytes = (unsigned)((p.H - 1) * S.row_pitch + (p.W - 1) * S.pix_pitch + S.C) * 4u;

    unsigned ivoff[NRAW];
    i32x4 rs = {0, 0, 0, 0x00020000};
    int in_soff = 0;
    // descriptor / per-lane offsets / channel offset of chunk q of the stream (q >= G: zero records -- zeros, no memory traffic)
    auto set_chunk = [&](int q) __attribute__((always_inline)) {
      const bool live = q < G;
      const int qq = live ? q : 0;
      const int it = qq / nch, c = qq - it * nch;
      const int tile = tile_first + it;
      const int b = tile / tpi, tr = tile - b * tpi;
      const int ty = tr / p.tiles_x, tx = tr - ty * p.tiles_x;
      const unsigned long long a = (unsigned long long)(uintptr_t)(S.ptr + (long long)b * S.img_pitch);
      rs[0] = __builtin_amdgcn_readfirstlane((int)(unsigned)a);
      rs[1] = __builtin_amdgcn_readfirstlane((int)((unsigned)(a >> 32) & 0xffffu));
      rs[2] = __builtin_amdgcn_readfirstlane(live ? (int)src_bytes : 0);
      in_soff = __builtin_amdgcn_readfirstlane(c * KC * 4);
      const int iy0 = ty * THY - 1, ix0 = tx * TWX - 1;
#pragma unroll
      for (int r = 0; r < NRAW; ++r) {
        const int iy = iy0 + ry[r], ix = ix0 + rx[r];
        const unsigned bad = ((r < NRAW - 1 || last_ok) && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W) ? 0u : 1u;
        ivoff[r] = ((unsigned)(iy * S.row_pitch + ix * S.pix_pitch + 4 * quad) * 4u) | (bad << 31);
      }
    };
    // two register sets: raw(q) lives in set q & 1 and is fetched TWO steps before its split (one step of slack was the loaders'
    // whole latency: 4.3 us per chunk with nothing else running)
    f32x4 rawr[2][NRAW];
    auto issue_raw = [&](auto setc) __attribute__((always_inline)) {
      constexpr int SET = decltype(setc)::value;
      static_for<0, NRAW>([&](auto rr) __attribute__((always_inline)) {
        constexpr int R = decltype(rr)::value;
        buf_load128f<R == 0>(rawr[SET][R], ivoff[R], rs, in_soff);
      });
    };
    float amax = 0.0f;
    auto split_store = [&](auto setc, unsigned dst) __attribute__((always_inline)) {   // set -> the two planes of the buffer at `dst`
      constexpr int SET = decltype(setc)::value;
      static_for<0, NRAW>([&](auto rr) __attribute__((always_inline)) {
        constexpr int R = decltype(rr)::value;
        u32x2 q0, q1;
        split2_f16(rawr[SET][R], q0, q1);
        asm volatile("v_max3_f32 %0, %0, |%1|, |%2|\n\tv_max3_f32 %0, %0, |%3|, |%4|"
                     : "+v"(amax) : "v"(rawr[SET][R][0]), "v"(rawr[SET][R][1]), "v"(rawr[SET][R][2]), "v"(rawr[SET][R][3]));
        if (R < NRAW - 1 || last_ok) {
          lds_write64<R * 1024>(dst, q0);
          lds_write64<R * 1024 + 2 * HALFB>(dst, q1);
        }
      });
    };
    auto issue_w = [&](int q) __attribute__((always_inline)) {   // weights of chunk q -> buffer q & 1 (q >= G: into the dummy page)
      const bool live = q < G;
      const int c = live ? q % nch : 0;
      const unsigned dst0 = w_base + (unsigned)(q & 1) * WCHUNK + (unsigned)(pw * NWP) * 1024u;
#pragma unroll
      for (int i = 0; i < NWP; ++i) {
        const unsigned dst = live ? dst0 + i * 1024 : dummy;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(wrsrc, (__attribute__((address_space(3))) void*)dst, 16, wvoff, c * WCHUNK + i * 1024, 0, 0);
      }
    };
    // nothing that reads a set may be scheduled above the wait before it  (plain lambdas: hipcc rejects "+v" operands that name a
    // captured array inside a GENERIC lambda)
    auto raw_fence0 = [&]() __attribute__((always_inline)) {
      asm volatile("" : "+v"(rawr[0][0]), "+v"(rawr[0][1]), "+v"(rawr[0][2]), "+v"(rawr[0][3]), "+v"(rawr[0][4]));
      asm volatile("" : "+v"(rawr[0][5]), "+v"(rawr[0][6]), "+v"(rawr[0][7]), "+v"(rawr[0][8]), "+v"(rawr[0][9]));
    };
    auto raw_fence1 = [&]() __attribute__((always_inline)) {
      asm volatile("" : "+v"(rawr[1][0]), "+v"(rawr[1][1]), "+v"(rawr[1][2]), "+v"(rawr[1][3]), "+v"(rawr[1][4]));
      asm volatile("" : "+v"(rawr[1][5]), "+v"(rawr[1][6]), "+v"(rawr[1][7]), "+v"(rawr[1][8]), "+v"(rawr[1][9]));
    };
    auto raw_fence = [&](auto setc) __attribute__((always_inline)) {
      if constexpr (decltype(setc)::value == 0) raw_fence0();
      else raw_fence1();
    };
    using S0 = std::integral_constant<int, 0>;
    using S1 = std::integral_constant<int, 1>;

    // ---- prologue: chunk 0 -> plane buffer 0 / weight buffer 0; chunks 1 and 2 on their way into sets 1 and 0
    set_chunk(0);
    issue_raw(S0());
    issue_w(0);
    set_chunk(1);
    issue_raw(S1());
    wait_vmcnt<NRAW>();     // raw(0) and W(0) (both older than raw(1))
    raw_fence(S0());
    __builtin_amdgcn_sched_barrier(0);
    split_store(S0(), cdst);
    __builtin_amdgcn_sched_barrier(0);
    set_chunk(2);
    issue_raw(S0());
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();

    // step g (chunk g is being multiplied): W(g+1) -> the other weight buffer; raw(g+1) (set (g+1)&1, fetched two steps ago) ->
    // the other plane buffer; raw(g+3) into the set just emptied.  In flight at the first wait, in issue order: raw(g+1) x10,
    // raw(g+2) x10, W(g+1) x9; at the second: W(g+1) x9 (with raw(g+2) before it), raw(g+3) x10.
    auto step = [&](int g, auto setc) __attribute__((always_inline)) {
      if constexpr ((C2M_PC_ABL & 2) != 0) {
        __builtin_amdgcn_s_barrier();
        return;
      }
      issue_w(g + 1);
      wait_vmcnt<NRAW + NWP>();
      raw_fence(setc);
      __builtin_amdgcn_sched_barrier(0);
      split_store(setc, cdst + (unsigned)((g + 1) & 1) * PLB);
      __builtin_amdgcn_sched_barrier(0);
      set_chunk(g + 3);
      issue_raw(setc);
      wait_vmcnt<NRAW>();
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
    };
    int g = 0;
    for (; g + 1 < G; g += 2) {
      step(g, S1());       // raw(g+1): g even -> set 1
      step(g + 1, S0());
    }
    if (g < G) step(g, S1());
    wait_vmcnt<0>();
    if (p.range_flag != nullptr && !(amax < 65520.0f)) *p.range_flag = 1;   // (rare, idempotent store; inf / NaN count)
    return;
  }

  // ===================================================================================================================
  // matrix waves
  // ===================================================================================================================
  const unsigned abase = w_base + hi * 512 + j * 16;
  const unsigned bbase = pl_base + hi * HALFB + (unsigned)((NT * wv) * HWc + j) * 16;
  const float w_sinv = *reinterpret_cast<const float*>(reinterpret_cast<const char*>(p.wr) + (size_t)nch * WCHUNK);
  f32x16 acc[MT][NT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mt][nt][r] = 0.0f;
  f16x8 A[2][2][MT];   // [set = tap & 1][image: 0 wA, 1 w1][mt]
  f16x8 Ad[MT];        // 2^-11 wA of the current tap
  f16x8 Bq[2][2];      // [set = step & 1][plane: 0 x0, 1 x1']
  auto load_a = [&](auto setc, auto tapc, auto kc, unsigned aw) __attribute__((always_inline)) {   // aw = abase + weight buffer
    constexpr int SET = decltype(setc)::value, T = decltype(tapc)::value, K = decltype(kc)::value;
    lds_read128<(T / 3) * WUNIT + (T % 3) * WTAP + K * 1024>(A[SET][K / MT][K % MT], aw);
  };
  auto load_b = [&](auto setc, auto tapc, auto ntc, auto plc, unsigned bp) __attribute__((always_inline)) {   // bp = bbase + plane buffer
    constexpr int SET = decltype(setc)::value, T = decltype(tapc)::value, N = decltype(ntc)::value, P = decltype(plc)::value;
    lds_read128<P * 2 * HALFB + ((N + T / 3) * HWc + T % 3) * 16>(Bq[SET][P], bp);
  };
  __builtin_amdgcn_s_barrier();   // (the loaders' prologue barrier: chunk 0 is in place)

  int c = 0, it = 0;
  for (int g = 0; g < G; ++g) {
    const unsigned aw = abase + (unsigned)(g & 1) * WCHUNK, bp = bbase + (unsigned)(g & 1) * PLB;
    // operands of the first step
    static_for<0, 4>([&](auto kc) __attribute__((always_inline)) { load_a(std::integral_constant<int, 0>(), std::integral_constant<int, 0>(), kc, aw); });
    static_for<0, 2>([&](auto plc) __attribute__((always_inline)) {
      load_b(std::integral_constant<int, 0>(), std::integral_constant<int, 0>(), std::integral_constant<int, 0>(), plc, bp);
    });
    static_for<0, 36>([&](auto sc) __attribute__((always_inline)) {
      constexpr int STEP = decltype(sc)::value, T = STEP / 4, N = STEP % 4;
      constexpr int aset = T & 1, bset = STEP & 1;
      if constexpr (!(C2M_PC_ABL & 4) || STEP == 0) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
      // next step's B operands; the next tap's A operands ride in the tap's last two steps
      if constexpr (STEP + 1 < 36 && !(C2M_PC_ABL & 16)) {
        static_for<0, 2>([&](auto plc) __attribute__((always_inline)) {
          load_b(std::integral_constant<int, bset ^ 1>(), std::integral_constant<int, (STEP + 1) / 4>(), std::integral_constant<int, (STEP + 1) % 4>(), plc, bp);
        });
      }
      if constexpr (T + 1 < 9 && N >= 2 && !(C2M_PC_ABL & 16)) {
        static_for<2 * (N - 2), 2 * (N - 2) + 2>([&](auto kc) __attribute__((always_inline)) {
          load_a(std::integral_constant<int, aset ^ 1>(), std::integral_constant<int, T + 1>(), kc, aw);
        });
      }
      if constexpr (N == 0) {
        const f16x8 sc2 = {(_Float16)(1.0f / F16_LO_SCALE), (_Float16)(1.0f / F16_LO_SCALE), (_Float16)(1.0f / F16_LO_SCALE), (_Float16)(1.0f / F16_LO_SCALE),
                           (_Float16)(1.0f / F16_LO_SCALE), (_Float16)(1.0f / F16_LO_SCALE), (_Float16)(1.0f / F16_LO_SCALE), (_Float16)(1.0f / F16_LO_SCALE)};
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) Ad[mt] = A[aset][0][mt] * sc2;
      }
      // per accumulator: w1.x0, wB.x1', wA.x0 (smallest terms first, as conv3x3_split.hip's Flavour<2>)
      if constexpr (!(C2M_PC_ABL & 8)) {
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) acc[mt][N] = __builtin_amdgcn_mfma_f32_32x32x16_f16(A[aset][1][mt], Bq[bset][0], acc[mt][N], 0, 0, 0);
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) acc[mt][N] = __builtin_amdgcn_mfma_f32_32x32x16_f16(Ad[mt], Bq[bset][1], acc[mt][N], 0, 0, 0);
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) acc[mt][N] = __builtin_amdgcn_mfma_f32_32x32x16_f16(A[aset][0][mt], Bq[bset][0], acc[mt][N], 0, 0, 0);
      } else {
        acc[0][N][STEP % 16] += (float)Bq[bset][0][0] + (float)Ad[0][1] + (float)A[aset][1][1][2];
      }
      __builtin_amdgcn_sched_barrier(0);
    });
    __builtin_amdgcn_s_barrier();   // chunk g+1 is in place; the buffers of chunk g are free
    if (++c < nch) continue;
    c = 0;
    // ---- epilogue of the tile
    const int tile = tile_first + it;
    ++it;
    const int b = tile / tpi, tr = tile - b * tpi;
    const int ty = tr / p.tiles_x, tx = tr - ty * p.tiles_x;
    const int y0 = ty * THY + NT * wv, x = tx * TWX + j;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int qd = 0; qd < 4; ++qd) {
        const f32x4 bv = *(const __attribute__((address_space(3))) f32x4*)(bias_lds + (mt * 32 + 8 * qd + 4 * hi) * 4);
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            float v = acc[mt][nt][4 * qd + e] * w_sinv + bv[e];   // (power of two: exact)
            if (p.act == 1) v = fmaxf(v, 0.0f);
            else if (p.act == 2) v = fmaxf(v, v * p.slope);
            acc[mt][nt][4 * qd + e] = v;
          }
      }
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      const int y = y0 + nt;
      if (y < p.H && x < p.W) {
        const size_t opix = (size_t)b * p.out_img_pitch + (size_t)y * p.out_row_pitch + (size_t)x * p.out_pix_pitch + 4 * hi;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
          for (int qd = 0; qd < 4; ++qd) {
            f32x4 v;
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = acc[mt][nt][4 * qd + e];
            const size_t o = opix + mt * 32 + 8 * qd;
            if (p.res1) v += *reinterpret_cast<const f32x4*>(p.res1 + o);
            if (p.res2) v += *reinterpret_cast<const f32x4*>(p.res2 + o);
            if ((C2M_PC_ABL & 1) && (nt + mt + qd != 0) && v[0] != 12345.678f) continue;
            *reinterpret_cast<f32x4*>(p.out + o) = v;
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

}  // namespace pc

// Is the loader / matrix-wave kernel defined for this launch?  (f16 x 2 flavour, channels-last mode, one source of fp32 tensors)
bool pc_supported(const Params& p) {
  return p.out_mode == 0 && p.Cout == 64 && p.Cin % pc::KC == 0 && p.src[1].C == 0 && p.src[0].C == p.Cin && p.out_vec4 && p.io_flags == 0 &&
         p.out2 == nullptr;
}

int launch_pc(hipStream_t st, Params p) {
  p.tiles_x = ceil_div(p.W, pc::TWX);
  p.tiles_y = ceil_div(p.H, pc::THY);
  p.nchunks = p.Cin / pc::KC;
  const long long ntile = (long long)p.tiles_x * p.tiles_y * p.B;
  if (ntile > 0x7fffffffLL) return C2M_ERR_INVALID_ARG;
  const long long resident = 256;   // one workgroup per CU
  p.tpw = (int)((ntile + resident - 1) / resident);
  dim3 grid((unsigned)((ntile + p.tpw - 1) / p.tpw));
  static unsigned long long done = 0;
  if (int rc = ensure_dynamic_lds(reinterpret_cast<const void*>(&pc::conv3x3_pc_kernel), pc::LDS_BYTES, done)) return rc;
  hipLaunchKernelGGL(pc::conv3x3_pc_kernel, grid, dim3(512), pc::LDS_BYTES, st, p);
  return check_launch();
}

}  // namespace conv
}  // namespace c2m
