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
//
// Kernels in this file (DESIGN.md 6 has the measurements):
//   conv3x3_kernel<MT, MODE>        the direct kernel described above (any shape the family supports);
//   conv3x3_wino_kernel<PX, MODE>   Winograd F(2,3) along x, 1.5x fewer MFMAs: channels-last outputs with 64-channel tiles
//                                   on maps a multiple of 32 pixels wide; MODE 3 = DCN head, MODE 4 = + ReLU + MaxPool2d(2,2);
//   conv3x3_wino4_kernel            Winograd F(4,3) along x, 2x fewer MFMAs, one wave per SIMD; decoder only (its rounding
//                                   error is ~4x the direct kernel's), maps a multiple of 64 pixels wide;
//   conv3x3_c3_kernel               the 3 -> 64 first layer of the image towers (im2col, K = 27, fused (x - mean) / std);
//   *_relayout_*_kernel, index_to_flow_kernel   weight images for the three algorithms; arg-max indices -> flow map.
#include <stdio.h>
#include <stdlib.h>

#include <type_traits>

#include "c2m_common.h"
#include "conv3x3_shared.h"

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

// MODE = Params::out_mode (compile time: each store flavour is its own kernel, the others' code is not even loaded)
//
// A workgroup processes p.tpw consecutive tiles as ONE continuous stream of chunks and units: the halo tile of the next
// tile's first chunks and its weight images are DMA'd while the current tile is still being multiplied, so only the first
// tile of a workgroup waits for memory; bias, descriptors and operand addresses are set up once.
template <int MT, int MODE>
__global__ void __launch_bounds__(256, 2) conv3x3_kernel(Params p) {
  constexpr int MW = 32 * MT;
  constexpr int WSLOT = MW * 128;          // bytes of one unit's weight image
  constexpr int NW_W = MT;                 // weight DMA instructions per wave and unit (1 KiB each)
  extern __shared__ __attribute__((aligned(1024))) char lds[];
  // [in0 | in1 | w ring x3 | dummy 1 KiB | bias MW floats]
  const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)lds;   // LDS byte address
  const unsigned in_base = lds0, w_base = lds0 + 2 * IN_BYTES, dummy = w_base + 3 * WSLOT;

  const int tid = threadIdx.x, l = tid & 63, hi = l >> 5, j = l & 31;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ntile = p.tiles_x * p.tiles_y * p.B;
  const int tile_first = xcd_remap(blockIdx.x, gridDim.x) * p.tpw;
  const int ntl = min(p.tpw, ntile - tile_first);   // >= 1 by construction of the grid
  const int cb = blockIdx.y;
  const int U = p.nchunks * 9;
  const int G = ntl * p.nchunks;                     // chunks of this workgroup; units: 9 * G

  // ------------------------------------------------------------------------------------------------------------------
  // DMA plumbing.  Every operand byte travels global -> LDS by buffer_load_dwordx4 ... lds (1 KiB per wave-instruction:
  // LDS address = wave-uniform M0 base + lane * 16, global address = descriptor base + SGPR offset + per-lane VGPR offset).
  //   weights : unit u of this cout block is one linear image of WSLOT bytes: voffset = (instr * 64 + lane) * 16,
  //             soffset = u * WSLOT.  No VALU per unit.
  //   halo    : instruction n (64 pieces of 16 B) is issued by wave n & 3 as its slot n >> 2; piece P = 64n + lane is pixel
  //             pl = P >> 3 of the 34 x 6 tile, LDS slot P & 7, logical piece q = slot ^ ((pl >> 1) & 7) (channels 4q..4q+3
  //             of the chunk).  voffset = byte offset of (pixel, 4q) inside the sample, or kOOB outside the image / tile;
  //             soffset = the chunk's channel offset.  The 7 voffsets are computed once per (tile, source tensor).
  // ------------------------------------------------------------------------------------------------------------------
  const __amdgpu_buffer_rsrc_t wrsrc = make_rsrc(p.wr + (long long)cb * U * (MW * 32), (unsigned)U * WSLOT);
  const unsigned wvoff = (wv * NW_W * 64 + l) * 16;
  // units are issued strictly in stream order: `wsoff` is the byte offset of the next unit's image inside this cout block's
  // weights (wraps after U units, once per tile); `slot` = ring slot of that unit (stream index % 3: a constant at every call
  // site, because U % 3 == 0)
  int wsoff = 0;
  auto issue_w = [&](int slot) __attribute__((always_inline)) {
    const unsigned dst = w_base + slot * WSLOT + wv * NW_W * 1024;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(wrsrc, (__attribute__((address_space(3))) void*)dst, 16, wvoff, wsoff, 0, 0);
    if constexpr (NW_W == 2)   // the instruction offset advances BOTH the global and the LDS address
      __builtin_amdgcn_raw_ptr_buffer_load_lds(wrsrc, (__attribute__((address_space(3))) void*)dst, 16, wvoff, wsoff, 1024, 0);
    wsoff += WSLOT;
    if (wsoff == U * WSLOT) wsoff = 0;
  };

  auto tile_coords = [&](int tile, int& b, int& y0, int& x0) __attribute__((always_inline)) {
    const int tx = tile % p.tiles_x, ty = (tile / p.tiles_x) % p.tiles_y;
    b = tile / (p.tiles_x * p.tiles_y);
    x0 = tx * TW;
    y0 = ty * TH;
  };
  // state of the DMA side (it runs up to two chunks -- possibly one tile -- ahead of the MFMA side)
  unsigned ivoff[NIN_W];
  int ib = 0, iy0 = 0, ix0 = 0;
  __amdgpu_buffer_rsrc_t rs0, rs1;
  auto set_source = [&](const Src& S) __attribute__((always_inline)) {
    int ry = 0, rx = 8 * wv + (l >> 3);   // pixel of slot 0 (< 34: row 0 of the halo tile); each further slot is 32 pixels on
#pragma unroll
    for (int sl = 0; sl < NIN_W; ++sl) {
      const int n = wv + 4 * sl;
      const int q = (l & 7) ^ ((4 * n + (l >> 4)) & 7);
      const int iy = iy0 - 1 + ry, ix = ix0 - 1 + rx;
      const bool ok = n < NIN_REAL && ry < HH_ && iy >= 0 && iy < p.H && ix >= 0 && ix < p.W;
      ivoff[sl] = ok ? (unsigned)(iy * S.row_pitch + ix * S.pix_pitch + 4 * q) * 4u : kOOB;
      rx += 32;
      if (rx >= HW_) { rx -= HW_; ry += 1; }
    }
  };
  auto src_rsrc = [&](const Src& S, int b) __attribute__((always_inline)) {
    const unsigned bytes = (unsigned)((p.H - 1) * S.row_pitch + (p.W - 1) * S.pix_pitch + S.C) * 4u;
    return make_rsrc(S.ptr + (long long)b * S.img_pitch, bytes);
  };
  auto issue_in = [&](int gc) __attribute__((always_inline)) {   // gc: chunk index in the workgroup's stream
    const int it = gc / p.nchunks, c0 = (gc - it * p.nchunks) * KCH;
    const bool first = c0 < p.src[0].C;
    if (c0 == 0) {
      tile_coords(tile_first + it, ib, iy0, ix0);
      rs0 = src_rsrc(p.src[0], ib);
      rs1 = src_rsrc(p.src[1], ib);
      set_source(p.src[0]);
    } else if (c0 == p.src[0].C) {
      set_source(p.src[1]);
    }
    const unsigned buf = in_base + (gc & 1) * IN_BYTES;
    const int soff = (first ? c0 : c0 - p.src[0].C) * 4;
#pragma unroll
    for (int sl = 0; sl < NIN_W; ++sl) {
      const int n = wv + 4 * sl;
      const unsigned dst = n < NIN_REAL ? buf + n * 1024 : dummy;
      if (first) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs0, (__attribute__((address_space(3))) void*)dst, 16, ivoff[sl], soff, 0, 0);
      else __builtin_amdgcn_raw_ptr_buffer_load_lds(rs1, (__attribute__((address_space(3))) void*)dst, 16, ivoff[sl], soff, 0, 0);
    }
  };

  // acc[mt][r] = out channel cb*MW + mt*32 + 8*(r>>2) + 4*hi + (r&3) of pixel (y, x).  The bias of this cout block is parked
  // in LDS (published by the first barrier) and added in the epilogue: holding it in registers across tiles costs 16*MT VGPRs
  // ------------------------------------------------------------------------------------------------------------------
  const int co_lane = cb * MW + 4 * hi;    // + mt*32 + 8*qd + e
  const unsigned bias_lds = dummy + 1024;  // MW floats
  if (tid < MW) {
    const int co = cb * MW + tid;
    *(__attribute__((address_space(3))) float*)(bias_lds + tid * 4) = (p.bias && co < p.Cout) ? p.bias[co] : 0.0f;
  }
  f32x16 acc[MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[mt][r] = 0.0f;

  // A-operand byte addresses of ring slot 0: k-quad g (row = cout j of tile mt, piece 2g + hi); + slot * WSLOT + mt * 4096 go
  // into the instruction's offset field.  Loop invariant.
  unsigned aaddr[4];
#pragma unroll
  for (int g = 0; g < 4; ++g) aaddr[g] = w_base + j * 128 + (((2 * g + hi) ^ ((j >> 1) & 7)) << 4);

  // Operand reads are inline asm with hand-placed s_waitcnt: hipcc's own counter tracking collapses to lgkmcnt(0) after
  // every other k-quad here, which waits for the reads just issued for the NEXT quad (one exposed LDS latency per 8 MFMAs).
  // LDS returns in order, so after issuing the next quad's NRD reads "lgkmcnt(NRD)" means: the current quad's have landed
  // (scalar loads that may share the counter only make this wait longer, never shorter than needed).
  constexpr int NRD = MT + 1;
  // operands of (tap t, k-quad g) of the chunk whose halo tile sits at LDS address ibuf: one b128 per operand
  auto load_ops = [&](unsigned ibuf, int t, int g, f32x4 (&a)[MT], f32x4& bq) __attribute__((always_inline)) {
    const int dy = t / 3, dx = t - 3 * dy;
    const int pl = (wv + dy) * HW_ + j + dx;
    // piece (2g + hi) ^ ((pl >> 1) & 7): the k-quad only flips bits 5-6 of the address -> one base per tap, one xor per quad
    const unsigned baddr = (ibuf + pl * 128 + (((hi ^ (pl >> 1)) & 7) << 4)) ^ (unsigned)(g << 5);
    asm volatile("ds_read_b128 %0, %1" : "=v"(bq) : "v"(baddr) : "memory");
    // 9 % 3 == 0: tap t always sits in ring slot t % 3 (t is a constant after unrolling: the switch folds)
    if constexpr (MT == 2) {
      switch (t % 3) {
        case 0:
          asm volatile("ds_read_b128 %0, %1" : "=v"(a[0]) : "v"(aaddr[g]) : "memory");
          asm volatile("ds_read_b128 %0, %1 offset:4096" : "=v"(a[1]) : "v"(aaddr[g]) : "memory");
          break;
        case 1:
          asm volatile("ds_read_b128 %0, %1 offset:8192" : "=v"(a[0]) : "v"(aaddr[g]) : "memory");
          asm volatile("ds_read_b128 %0, %1 offset:12288" : "=v"(a[1]) : "v"(aaddr[g]) : "memory");
          break;
        default:
          asm volatile("ds_read_b128 %0, %1 offset:16384" : "=v"(a[0]) : "v"(aaddr[g]) : "memory");
          asm volatile("ds_read_b128 %0, %1 offset:20480" : "=v"(a[1]) : "v"(aaddr[g]) : "memory");
          break;
      }
    } else {
      switch (t % 3) {
        case 0: asm volatile("ds_read_b128 %0, %1" : "=v"(a[0]) : "v"(aaddr[g]) : "memory"); break;
        case 1: asm volatile("ds_read_b128 %0, %1 offset:4096" : "=v"(a[0]) : "v"(aaddr[g]) : "memory"); break;
        default: asm volatile("ds_read_b128 %0, %1 offset:8192" : "=v"(a[0]) : "v"(aaddr[g]) : "memory"); break;
      }
    }
  };
  // wait until at most N LDS reads are in flight; the operands are tied to the statement so that no MFMA reading them can be
  // scheduled above it
  auto wait_ops = [&](auto n, f32x4 (&a)[MT], f32x4& bq) __attribute__((always_inline)) {
    constexpr int N = decltype(n)::value;
    if constexpr (MT == 2) asm volatile("s_waitcnt lgkmcnt(%3)" : "+v"(bq), "+v"(a[0]), "+v"(a[1]) : "n"(N));
    else asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(bq), "+v"(a[0]) : "n"(N));
  };

  // ------------------------------------------------------------------------------------------------------------------
  // main loop over the workgroup's chunk stream.  Unit gu = (chunk gc, tap t).  Invariant at the top of unit gu: the weight
  // images of units gu and gu+1 and the halo tile of chunk gc (and, from (gc, 2) on, of chunk gc+1) are visible to every
  // wave; W(gu+2) is in flight.  The barrier at the END of unit gu publishes W(gu+2) and frees ring slot gu % 3 for W(gu+3).
  // Because unit gu+1's operands are visible one unit early, the last k-quad of unit gu already fetches the first operands
  // of unit gu+1: MFMAs issue back to back across the barrier -- and across tiles.
  // ------------------------------------------------------------------------------------------------------------------
  issue_in(0);
  if (G > 1) issue_in(1);
  issue_w(0);
  issue_w(1);
  issue_w(2);
  wait_vmcnt<0>();
  __builtin_amdgcn_s_barrier();

  // two operand sets, used alternately (k-quad g of any unit reads set g & 1: 4 quads per unit keeps the parity fixed)
  f32x4 a_s[2][MT], b_s[2];
  load_ops(in_base, 0, 0, a_s[0], b_s[0]);
  f32x4 res4[MT][4];
  for (int it = 0, gc = 0; it < ntl; ++it) {
    // MFMA-side tile (epilogue addresses)
    int b, y0, x0;
    tile_coords(tile_first + it, b, y0, x0);
    const int y = y0 + wv, x = x0 + j;
    const bool pok = y < p.H && x < p.W;
    const size_t opix = (size_t)b * p.out_img_pitch + (size_t)y * p.out_row_pitch + (size_t)x * p.out_pix_pitch;
    const bool vec_res = MODE == 0 && p.out_vec4 && pok && (p.res1 || p.res2);
   for (int c = 0; c < p.nchunks; ++c, ++gc) {
    const unsigned ibuf = in_base + (gc & 1) * IN_BYTES, ibuf_next = in_base + ((gc + 1) & 1) * IN_BYTES;
    const bool more_in = gc + 1 < G;
    const bool last_chunk = c == p.nchunks - 1;   // of its tile
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      const int gu = gc * 9 + t;
      if (MODE == 0 && t == 8 && last_chunk) {   // top of the tile's last unit: residuals -- their latency runs under MFMAs
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
          for (int qd = 0; qd < 4; ++qd) res4[mt][qd] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
        if (vec_res) {
          const float* r1 = p.res1 ? p.res1 + opix + co_lane : nullptr;
          const float* r2 = p.res2 ? p.res2 + opix + co_lane : nullptr;
#pragma unroll
          for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int qd = 0; qd < 4; ++qd)
              if (co_lane + mt * 32 + 8 * qd + 3 < p.Cout) {
                if (r1) res4[mt][qd] = *reinterpret_cast<const f32x4*>(r1 + mt * 32 + 8 * qd);
                if (r2) res4[mt][qd] += *reinterpret_cast<const f32x4*>(r2 + mt * 32 + 8 * qd);
              }
        }
      }
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int cur = g & 1, nxt = cur ^ 1;
        bool fetched = true;
        if (g < 3) load_ops(ibuf, t, g + 1, a_s[nxt], b_s[nxt]);
        else if (t < 8) load_ops(ibuf, t + 1, 0, a_s[nxt], b_s[nxt]);
        else if (more_in) load_ops(ibuf_next, 0, 0, a_s[nxt], b_s[nxt]);
        else fetched = false;
        if (fetched) wait_ops(std::integral_constant<int, NRD>(), a_s[cur], b_s[cur]);
        else wait_ops(std::integral_constant<int, 0>(), a_s[cur], b_s[cur]);
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
          for (int mt = 0; mt < MT; ++mt)
            acc[mt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_s[cur][mt][e], b_s[cur][e], acc[mt], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
      if (gu + 1 < 9 * G) {
        // W(gu+2) was issued one unit ago and is the youngest DMA in flight -- except at t == 0 of chunks >= 1, where the
        // halo tile of chunk gc+1 was issued right after it (end of unit (gc-1, 8)) and may keep flying
        if (t == 0 && gc >= 1 && more_in) wait_vmcnt<NIN_W>();
        else wait_vmcnt<0>();
        // bare s_barrier: __syncthreads() would add a fence = s_waitcnt vmcnt(0) and drain DMAs that may stay in flight
        __builtin_amdgcn_s_barrier();
        if (gu + 3 < 9 * G) issue_w(t % 3);   // unit gu+3 -> the slot unit gu just vacated
        if (t == 8 && gc + 2 < G) issue_in(gc + 2);
      }
    }
   }   // chunks of the tile
    {   // + bias (LDS, one b128 per 4 consecutive channels)
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int qd = 0; qd < 4; ++qd) {
          const f32x4 bv = *(const __attribute__((address_space(3))) f32x4*)(bias_lds + (mt * 32 + 8 * qd + 4 * hi) * 4);
#pragma unroll
          for (int e = 0; e < 4; ++e) acc[mt][4 * qd + e] += bv[e];
        }
    }

    // ----------------------------------------------------------------------------------------------------------------
    // epilogue of the tile (bias is already inside acc); the accumulators restart from the bias for the next tile
    // ----------------------------------------------------------------------------------------------------------------
    if constexpr (MODE == 3) {
      float asum = 0.0f;
      const HeadOut ho = head_out(p, b);
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int qd = 0; qd < 4; ++qd) {
          const int col = co_lane + mt * 32 + 8 * qd;   // channel inside this launch's slice
          if (col >= p.Cout || !pok) continue;
          f32x4 v;
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = acc[mt][4 * qd + e];
          asum += dcn_head_store(p, ho, b, y, x, col - 4 * hi, 4 * hi, v);
        }
      if (p.abs_sum) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) asum += __shfl_xor(asum, off, 64);
        if (l == 0) atomicAdd(p.abs_sum + ((blockIdx.x * 4 + wv + blockIdx.y * 31 + it) & (C2M_ABS_SUM_SLOTS - 1)), (double)asum);
      }
    } else {
      // activation: ReLU = max(v, 0); LeakyReLU = max(v, slope * v) (0 <= slope <= 1, checked by the launcher)
      if (p.act == 1) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[mt][r] = fmaxf(acc[mt][r], 0.0f);
      } else if (p.act == 2) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[mt][r] = fmaxf(acc[mt][r], acc[mt][r] * p.slope);
      }
      if (pok) {
        if constexpr (MODE == 0) {
          float* ob = p.out + opix + co_lane;   // + mt*32 + 8*qd: immediate offsets
          if (p.out_vec4) {
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
              for (int qd = 0; qd < 4; ++qd) {
                const int co = co_lane + mt * 32 + 8 * qd;
                f32x4 v;
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = acc[mt][4 * qd + e];
                if (co + 3 < p.Cout) {
                  v += res4[mt][qd];
                  *reinterpret_cast<f32x4*>(ob + mt * 32 + 8 * qd) = v;
                  if (p.out2)
                    *reinterpret_cast<f32x4*>(p.out2 + (size_t)b * p.out2_img_pitch + (size_t)(co >> 3) * p.out2_plane_pitch +
                                              (size_t)y * p.out2_row_pitch + x * 8 + (co & 7)) = v;
                } else {
                  for (int e = 0; e < 4 && co + e < p.Cout; ++e) {
                    float sv = v[e];
                    if (p.res1) sv += p.res1[opix + co + e];
                    if (p.res2) sv += p.res2[opix + co + e];
                    ob[mt * 32 + 8 * qd + e] = sv;
                  }
                }
              }
          } else {
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
              for (int r = 0; r < 16; ++r) {
                const int cr = mt * 32 + 8 * (r >> 2) + (r & 3);
                if (co_lane + cr < p.Cout) {
                  float sv = acc[mt][r];
                  if (p.res1) sv += p.res1[opix + co_lane + cr];
                  if (p.res2) sv += p.res2[opix + co_lane + cr];
                  ob[cr] = sv;
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
                  ob[(size_t)(e >> 1) * p.out_row_pitch + (size_t)(e & 1) * p.out_pix_pitch + mt * 8 + 2 * qd] = acc[mt][4 * qd + e];
              }
        } else {
          const size_t HWs = (size_t)p.H * p.W;
          float* ob = p.out + ((size_t)b * p.Cout + co_lane) * HWs + (size_t)y * p.W + x;
#pragma unroll
          for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              const int cr = mt * 32 + 8 * (r >> 2) + (r & 3);
              if (co_lane + cr < p.Cout) ob[(size_t)cr * HWs] = acc[mt][r];
            }
        }
      }
    }
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mt][r] = 0.0f;
  }
}


// =====================================================================================================================
// Winograd F(2,3) along x (mode 0 only).  Output pixels are produced in horizontal PAIRS: for every row tap dy and the four
// transform positions xi the kernel multiplies U[dy][xi] (= G g, precomputed: g0, (g0+g1+g2)/2, (g0-g1+g2)/2, g2) with
// V[dy][xi] (= B^T d: d0-d2, d1+d2, d2-d1, d1-d3 of the four input columns 2t-1 .. 2t+2) and accumulates M[xi] over
// (dy, channels); the pair is Y0 = M0+M1+M2, Y1 = M1-M2-M3.  12 instead of 18 MFMA k-steps per output pair: 1.5x fewer
// matrix instructions, paid with 0.5 VALU instruction per MFMA for the input transform (formed in registers from four
// b128 reads per row tap and k-quad, shared by the four xi units) and a 128-instruction combine per tile.  Same machinery
// as the direct kernel: buffer-addressed LDS-DMA with hardware zero fill, 8 KiB weight units in a ring of 3 (two units
// ahead), b128 operand reads with hand-placed waits, persistent tiles.
//   workgroup = 4 waves = 64 x 4 pixels (wave = row, lane (hi, j) = pixel pair j) or 32 x 8 (wave = two rows of 16 pairs)
//   chunk = 16 input channels (64-byte pixel rows in LDS, pieces swizzled by the halo column, (x >> 2) & 3; the stride-2 pair access
//   leaves a 2-way conflict on the 8 raw reads per 64 MFMAs), unit = (chunk, dy, two xi) = 32 MFMAs per wave
// fp32 throughout; results differ from the direct kernel by the rounding of the transforms (tested at 2e-5 * scale).
// =====================================================================================================================
namespace wino {
constexpr int KC = 16;                         // channels per chunk
constexpr int NIN_W = 7;                       // halo DMA instructions per wave (25 of 28 used by the 66 x 6 tile, 22 by 34 x 10)
constexpr int IN_BYTES = 25 * 1024;            // halo buffer: 396 pixels x 64 B, rounded up to whole DMA instructions
constexpr int WIMG = 64 * 64;                  // bytes of one (dy, xi) weight image: 64 couts x 16 k
constexpr int WUNIT = 2 * WIMG;                // a unit = two transform positions of one row tap: 8 KiB, 32 MFMAs per wave
constexpr int NRING = 3;
constexpr int UPC = 6;                         // units per chunk: 3 row taps x 2 pairs of transform positions
static_assert(UPC % NRING == 0, "ring slot of a unit must be a compile-time constant");
}  // namespace wino

// weights W[Cout][Cin][3][3] -> Wr[cb][chunk][dy][xi][row 64][slot 4][e 4], slot = q ^ ((row >> 2) & 3),
// value = U[cb*64 + row][chunk*16 + 4q + e][dy][xi] (0 beyond Cout)
__global__ void __launch_bounds__(256) conv3x3_relayout_wino_kernel(const float* __restrict__ w, int Cin, int Cout,
                                                                      long long total, float* __restrict__ wr) {
  const long long e0 = (long long)blockIdx.x * 256 + threadIdx.x;
  if (e0 >= total) return;
  const int e = (int)(e0 & 3), slot = (int)((e0 >> 2) & 3);
  long long r = e0 >> 4;
  const int row = (int)(r & 63); r >>= 6;
  const int xi = (int)(r & 3); r >>= 2;
  const int dy = (int)(r % 3); r /= 3;
  const int nch = Cin / wino::KC;
  const int chunk = (int)(r % nch);
  const int cb = (int)(r / nch);
  const int q = slot ^ ((row >> 2) & 3);
  const int co = cb * 64 + row, ci = chunk * wino::KC + 4 * q + e;
  float u = 0.0f;
  if (co < Cout) {
    const float* g = w + ((size_t)co * Cin + ci) * 9 + dy * 3;
    const float g0 = g[0], g1 = g[1], g2 = g[2];
    u = xi == 0 ? g0 : xi == 1 ? ((g0 + g1) + g2) * 0.5f : xi == 2 ? ((g0 - g1) + g2) * 0.5f : g2;
  }
  wr[e0] = u;
}

// PX = pixel pairs per wave row: 16 -> workgroup tile 32 x 8 pixels (wave = two rows of 16 pairs; what the launcher uses:
// smaller halo), 32 -> 64 x 4 (wave = one row; not instantiated any more, see c2m_conv3x3_nhwc_f32)
// MODE 0: channels-last (+ activation, residuals); MODE 3: DCN offset/mask head; MODE 4: channels-last + activation +
// MaxPool2d(2, 2) (the conv1_2 / conv2_2 -> pool steps of the VGG towers: only the pooled map is written)
template <int PX, int MODE>
__global__ void __launch_bounds__(256, 2) conv3x3_wino_kernel(Params p) {
  constexpr int RW = 32 / PX;                      // pixel rows per wave
  constexpr int TWX = 2 * PX, THY = 4 * RW;        // pixel tile of a workgroup
  constexpr int HWc = TWX + 2, HHr = THY + 2;      // halo tile
  constexpr int NIN_REAL = (HWc * HHr * 4 + 63) / 64;
  static_assert(NIN_REAL <= 4 * wino::NIN_W && NIN_REAL * 1024 <= wino::IN_BYTES, "halo tile must fit the DMA plan / LDS buffer");
  constexpr int MT = 2;
  extern __shared__ __attribute__((aligned(1024))) char lds[];
  // [in0 | in1 | w ring x6 | dummy 1 KiB | bias 64 floats]
  const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)lds;
  const unsigned in_base = lds0, w_base = lds0 + 2 * wino::IN_BYTES, dummy = w_base + wino::NRING * wino::WUNIT, bias_lds = dummy + 1024;

  const int tid = threadIdx.x, l = tid & 63, hi = l >> 5, j = l & 31;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int pt = j & (PX - 1), prow = wv * RW + j / PX;   // the lane's pixel pair inside the tile: columns 2pt, 2pt+1 of row prow
  const int ntile = p.tiles_x * p.tiles_y * p.B;
  const int tile_first = xcd_remap(blockIdx.x, gridDim.x) * p.tpw;
  const int ntl = min(p.tpw, ntile - tile_first);
  const int cb = blockIdx.y;
  const int UT = p.nchunks * wino::UPC;                    // units per tile
  const int G = ntl * p.nchunks;                     // chunks of this workgroup
  const int T = G * wino::UPC;                             // units of this workgroup

  // ---- DMA plumbing (see conv3x3_kernel): weights
  const __amdgpu_buffer_rsrc_t wrsrc = make_rsrc(p.wr + (long long)cb * UT * (wino::WUNIT / 4), (unsigned)UT * wino::WUNIT);
  const unsigned wvoff = (wv * 128 + l) * 16;
  int wsoff = 0;
  auto issue_w = [&](int slot) __attribute__((always_inline)) {
    const unsigned dst = w_base + slot * wino::WUNIT + wv * 2048;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(wrsrc, (__attribute__((address_space(3))) void*)dst, 16, wvoff, wsoff, 0, 0);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(wrsrc, (__attribute__((address_space(3))) void*)dst, 16, wvoff, wsoff, 1024, 0);
    wsoff += wino::WUNIT;
    if (wsoff == UT * wino::WUNIT) wsoff = 0;
  };
  // ---- halo tile: instruction n = wave + 4 * slot covers pieces [64n, 64n + 64): pixel pl = 16n + (lane >> 2) = halo
  //      (row ry, column rx), LDS slot lane & 3 holds logical piece (lane & 3) ^ ((rx >> 2) & 3): swizzled by the COLUMN, so a
  //      raw read address is linear in the row tap (immediate offsets) and the slot constants do not depend on the tile
  // tile coordinates are tracked incrementally (a workgroup's tiles are consecutive): integer divisions run on the vector
  // ALU, and one per chunk plus three per tile were a measurable share of the per-tile overhead
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
  unsigned ivoff[wino::NIN_W];
  int ib = 0, iy0 = 0, ix0 = 0;
  __amdgpu_buffer_rsrc_t rs0, rs1;
  int slotc[wino::NIN_W];   // ry | rx << 8 | piece << 16 | valid << 24 of this lane's DMA slots (tile-independent)
#pragma unroll
  for (int sl = 0; sl < wino::NIN_W; ++sl) {
    const int n = wv + 4 * sl, pl = 16 * n + (l >> 2);
    const int ry = pl / HWc, rx = pl - ry * HWc;
    slotc[sl] = ry | (rx << 8) | ((((l & 3) ^ ((rx >> 2) & 3))) << 16) | ((n < NIN_REAL && ry < HHr) ? (1 << 24) : 0);
  }
  auto set_source = [&](const Src& S) __attribute__((always_inline)) {
#pragma unroll
    for (int sl = 0; sl < wino::NIN_W; ++sl) {
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
  auto issue_in = [&](int gc) __attribute__((always_inline)) {   // called for gc = 0, 1, 2, ... in order
    const int c0 = dma_c * wino::KC;
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
    const unsigned buf = in_base + (gc & 1) * wino::IN_BYTES;
    const int soff = (first ? c0 : c0 - p.src[0].C) * 4;
#pragma unroll
    for (int sl = 0; sl < wino::NIN_W; ++sl) {
      const int n = wv + 4 * sl;
      const unsigned dst = n < NIN_REAL ? buf + n * 1024 : dummy;
      if (first) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs0, (__attribute__((address_space(3))) void*)dst, 16, ivoff[sl], soff, 0, 0);
      else __builtin_amdgcn_raw_ptr_buffer_load_lds(rs1, (__attribute__((address_space(3))) void*)dst, 16, ivoff[sl], soff, 0, 0);
    }
  };

  const int co_lane = cb * 64 + 4 * hi;
  if (tid < 64) {
    const int co = cb * 64 + tid;
    *(__attribute__((address_space(3))) float*)(bias_lds + tid * 4) = (p.bias && co < p.Cout) ? p.bias[co] : 0.0f;
  }
  f32x16 M[4][MT];
#pragma unroll
  for (int xi = 0; xi < 4; ++xi)
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int r = 0; r < 16; ++r) M[xi][mt][r] = 0.0f;

  // A operand: row = cout j (+32 mt), piece 2g + hi, slot q ^ ((row >> 2) & 3); + ring slot * 4096 + mt * 2048 as offsets
  unsigned aaddr[2];
#pragma unroll
  for (int g = 0; g < 2; ++g) aaddr[g] = w_base + j * 64 + (((2 * g + hi) ^ ((j >> 2) & 3)) << 4);
  auto load_a = [&](int uc, int st, f32x4 (&a)[MT]) __attribute__((always_inline)) {   // uc: unit inside the chunk, st: step = 2 * xi_local + g
    const int g = st & 1;
    switch ((uc % wino::NRING) * 2 + (st >> 1)) {
      case 0: asm volatile("ds_read_b128 %0, %2 offset:0\n\tds_read_b128 %1, %2 offset:2048" : "=&v"(a[0]), "=&v"(a[1]) : "v"(aaddr[g]) : "memory"); break;
      case 1: asm volatile("ds_read_b128 %0, %2 offset:4096\n\tds_read_b128 %1, %2 offset:6144" : "=&v"(a[0]), "=&v"(a[1]) : "v"(aaddr[g]) : "memory"); break;
      case 2: asm volatile("ds_read_b128 %0, %2 offset:8192\n\tds_read_b128 %1, %2 offset:10240" : "=&v"(a[0]), "=&v"(a[1]) : "v"(aaddr[g]) : "memory"); break;
      case 3: asm volatile("ds_read_b128 %0, %2 offset:12288\n\tds_read_b128 %1, %2 offset:14336" : "=&v"(a[0]), "=&v"(a[1]) : "v"(aaddr[g]) : "memory"); break;
      case 4: asm volatile("ds_read_b128 %0, %2 offset:16384\n\tds_read_b128 %1, %2 offset:18432" : "=&v"(a[0]), "=&v"(a[1]) : "v"(aaddr[g]) : "memory"); break;
      case 5: asm volatile("ds_read_b128 %0, %2 offset:20480\n\tds_read_b128 %1, %2 offset:22528" : "=&v"(a[0]), "=&v"(a[1]) : "v"(aaddr[g]) : "memory"); break;
      default: break;
    }
  };
  auto wait_a = [&](auto n, f32x4 (&a)[MT]) __attribute__((always_inline)) {
    constexpr int N = decltype(n)::value;
    asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(a[0]), "+v"(a[1]) : "n"(N));
  };
  // input transform of one row tap for both k-quads: four raw b128 reads per quad (input columns 2j-1 .. 2j+2 of the
  // lane's pair) -> V[xi][g].  The raw reads of the next group (chunk, dy) are issued in the last step of the current
  // group's last unit (in place of that step's operand prefetch); at the top of the next unit the operand read is issued
  // first and the 32 transform instructions run under its latency.
  f32x4 V[4][2], d[2][4];
  // raw read addresses of row tap 0 in the CURRENT halo buffer; row tap dy = immediate offset dy * HWc * 64
  unsigned raddr[2][4];
#pragma unroll
  for (int g = 0; g < 2; ++g)
#pragma unroll
    for (int bcol = 0; bcol < 4; ++bcol) {
      const int x = 2 * pt + bcol;
      raddr[g][bcol] = in_base + (prow * HWc + x) * 64 + ((((2 * g + hi) ^ (x >> 2)) & 3) << 4);
    }
  int rdelta = wino::IN_BYTES;   // to the other buffer
  auto issue_raw = [&](int dy) __attribute__((always_inline)) {
#pragma unroll
    for (int g = 0; g < 2; ++g)
#pragma unroll
      for (int bcol = 0; bcol < 4; ++bcol) {
        switch (dy) {
          case 0: asm volatile("ds_read_b128 %0, %1" : "=v"(d[g][bcol]) : "v"(raddr[g][bcol]) : "memory"); break;
          case 1: asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(d[g][bcol]) : "v"(raddr[g][bcol]), "n"(HWc * 64) : "memory"); break;
          default: asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(d[g][bcol]) : "v"(raddr[g][bcol]), "n"(2 * HWc * 64) : "memory"); break;
        }
      }
  };
  auto next_buffer = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int g = 0; g < 2; ++g)
#pragma unroll
      for (int bcol = 0; bcol < 4; ++bcol) raddr[g][bcol] += rdelta;
    rdelta = -rdelta;
  };
  auto wait_raw = [&]() __attribute__((always_inline)) {
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(d[0][0]), "+v"(d[0][1]), "+v"(d[0][2]), "+v"(d[0][3]), "+v"(d[1][0]), "+v"(d[1][1]),
                 "+v"(d[1][2]), "+v"(d[1][3]));
  };
  auto vcomp = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int g = 0; g < 2; ++g) {
      V[0][g] = d[g][0] - d[g][2];
      V[1][g] = d[g][1] + d[g][2];
      V[2][g] = d[g][2] - d[g][1];
      V[3][g] = d[g][1] - d[g][3];
    }
  };

  issue_in(0);
  if (G > 1) issue_in(1);
  issue_w(0);
  issue_w(1);
  issue_w(2);
  wait_vmcnt<0>();
  __builtin_amdgcn_s_barrier();

  // Synchronisation as in conv3x3_kernel: at the top of unit gu the weight units gu and gu+1 and the halo tile of chunk gc
  // (from unit 2 of a chunk on: of chunk gc+1) are visible, W(gu+2) is in flight; the barrier at the end of unit gu
  // publishes W(gu+2) and frees ring slot gu % 3 for W(gu+3).
  f32x4 a_s[2][MT];
  issue_raw(0);
  for (int it = 0, gc = 0; it < ntl; ++it) {
    for (int c = 0; c < p.nchunks; ++c, ++gc) {
      const bool more_in = gc + 1 < G;
#pragma unroll
      for (int uc = 0; uc < wino::UPC; ++uc) {
        const int gu = gc * wino::UPC + uc;
        const int dy = uc >> 1, xh = uc & 1;          // row tap, which half of the transform positions (xi = 2 xh, 2 xh + 1)
        const bool next_group = gu + 2 - xh < T;       // a group (chunk, dy) follows this one
        if (xh == 0) {   // raw values landed (issued in the previous group's last step / before the loop): operands, then transform
          wait_raw();
          load_a(uc, 0, a_s[0]);
          vcomp();
        }
#pragma unroll
        for (int st = 0; st < 4; ++st) {               // step = (xi_local, g)
          const int cur = st & 1, nxt = cur ^ 1;
          const int xi = 2 * xh + (st >> 1), g = st & 1;
          if (st < 3) {
            load_a(uc, st + 1, a_s[nxt]);
            wait_a(std::integral_constant<int, 2>(), a_s[cur]);
          } else if (xh == 1) {
            // last step of the group: fetch the next group's raw values instead of the next unit's operands (those are read at
            // the top of the next unit, under the transform): next row tap of this chunk, or row tap 0 of the next chunk
            if (next_group) {
              if (dy < 2) {
                issue_raw(dy + 1);
              } else {
                next_buffer();     // row tap 0 of the next chunk: the other halo buffer
                issue_raw(0);
              }
              wait_a(std::integral_constant<int, 8>(), a_s[cur]);
            } else {
              wait_a(std::integral_constant<int, 0>(), a_s[cur]);
            }
          } else {
            load_a(uc + 1, 0, a_s[nxt]);               // next unit of the same group
            wait_a(std::integral_constant<int, 2>(), a_s[cur]);
          }
#pragma unroll
          for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
              M[xi][mt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_s[cur][mt][e], V[xi][g][e], M[xi][mt], 0, 0, 0);
          __builtin_amdgcn_sched_barrier(0);
        }
        if (gu + 1 < T) {
          // W(gu+2) was issued one unit ago and is the youngest DMA in flight -- except at unit 0 of chunks >= 1, where the halo
          // tile of chunk gc+1 was issued right after it (end of the previous chunk's last unit) and may keep flying
          if (uc == 0 && gc >= 1 && more_in) wait_vmcnt<wino::NIN_W>();
          else wait_vmcnt<0>();
          __builtin_amdgcn_s_barrier();
          if (gu + 3 < T) issue_w(uc % wino::NRING);   // unit gu+3 -> the slot unit gu just vacated
          if (uc == wino::UPC - 1 && gc + 2 < G) issue_in(gc + 2);
        }
      }
    }
    // ---- tile epilogue: Y0 = M0 + M1 + M2 -> pixel x, Y1 = M1 - M2 - M3 -> pixel x + 1; + bias, then the mode's store
    // (pixel coordinates are derived here, not at the top of the tile: nothing of the epilogue stays live across the units)
    int b = epi_tc.b, y0 = epi_tc.ty * THY, x0 = epi_tc.tx * TWX;
    tc_next(epi_tc);
    asm volatile("" : "+s"(b), "+s"(y0), "+s"(x0));
    const int y = y0 + prow, x = x0 + 2 * pt;
    const bool pok = y < p.H && x < p.W;   // W is even: the pair is inside or outside as a whole
    const size_t opix = (size_t)b * p.out_img_pitch + (size_t)y * p.out_row_pitch + (size_t)x * p.out_pix_pitch;
    const float* r1 = p.res1 ? p.res1 + opix + co_lane : nullptr;
    const float* r2 = p.res2 ? p.res2 + opix + co_lane : nullptr;
    float* ob = p.out + opix + co_lane;
    float asum = 0.0f;
    HeadOut ho;
    if constexpr (MODE == 3) ho = head_out(p, b);
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int qd = 0; qd < 4; ++qd) {
        const int co = co_lane + mt * 32 + 8 * qd;
        const f32x4 bv = *(const __attribute__((address_space(3))) f32x4*)(bias_lds + (mt * 32 + 8 * qd + 4 * hi) * 4);
        f32x4 y0v, y1v;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float m0 = M[0][mt][4 * qd + e], m1 = M[1][mt][4 * qd + e], m2 = M[2][mt][4 * qd + e], m3 = M[3][mt][4 * qd + e];
          y0v[e] = ((m0 + m1) + m2) + bv[e];
          y1v[e] = ((m1 - m2) - m3) + bv[e];
          if constexpr (MODE == 0 || MODE == 4) {
            if (p.act == 1) { y0v[e] = fmaxf(y0v[e], 0.0f); y1v[e] = fmaxf(y1v[e], 0.0f); }
            else if (p.act == 2) { y0v[e] = fmaxf(y0v[e], y0v[e] * p.slope); y1v[e] = fmaxf(y1v[e], y1v[e] * p.slope); }
          }
          M[0][mt][4 * qd + e] = 0.0f; M[1][mt][4 * qd + e] = 0.0f; M[2][mt][4 * qd + e] = 0.0f; M[3][mt][4 * qd + e] = 0.0f;
        }
        if constexpr (MODE == 3) {
          if (pok && co < p.Cout) {
            asum += dcn_head_store2(p, ho, b, y, x, co - 4 * hi, 4 * hi, y0v, y1v);
          }
        } else if constexpr (MODE == 4) {
          // horizontal max inside the lane (its pixel pair), vertical max with the lane 16 further on (rows 2 wv, 2 wv + 1 of
          // the wave: PX == 16); the even-row lanes store pooled pixel (y / 2, x / 2)
          static_assert(MODE != 4 || PX == 16, "the pooled epilogue pairs the two rows of a wave");
          f32x4 pv;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float hm = fmaxf(y0v[e], y1v[e]);
            pv[e] = fmaxf(hm, __shfl_xor(hm, 16, 64));
          }
          if (pok && (j & 16) == 0 && co + 3 < p.Cout)
            *reinterpret_cast<f32x4*>(p.out + (size_t)b * p.out_img_pitch + (size_t)(y >> 1) * p.out_row_pitch +
                                      (size_t)(x >> 1) * p.out_pix_pitch + co_lane + mt * 32 + 8 * qd) = pv;
        } else if (pok && co + 3 < p.Cout) {
          const int o = mt * 32 + 8 * qd;
          if (r1) { y0v += *reinterpret_cast<const f32x4*>(r1 + o); y1v += *reinterpret_cast<const f32x4*>(r1 + p.out_pix_pitch + o); }
          if (r2) { y0v += *reinterpret_cast<const f32x4*>(r2 + o); y1v += *reinterpret_cast<const f32x4*>(r2 + p.out_pix_pitch + o); }
          *reinterpret_cast<f32x4*>(ob + o) = y0v;
          *reinterpret_cast<f32x4*>(ob + p.out_pix_pitch + o) = y1v;
        }
      }
    if (MODE == 3 && p.abs_sum) {
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) asum += __shfl_xor(asum, off, 64);
      if (l == 0) atomicAdd(p.abs_sum + ((blockIdx.x * 4 + wv + blockIdx.y * 31 + it) & (C2M_ABS_SUM_SLOTS - 1)), (double)asum);
    }
  }
}


// =====================================================================================================================
// Winograd F(4,3) along x (mode 0 only).  Output pixels are produced in horizontal QUADS: six transform positions per row
// tap instead of the 12 k-steps a direct quad costs -- 2x fewer matrix instructions than the direct kernel, 1.33x fewer
// than F(2,3).  U = G g (precomputed by conv3x3_relayout_wino4_kernel), V = B^T d over the six input columns 4t-1 .. 4t+4,
// Y = A^T M with
//   G   = [1/4 0 0; -1/6 -1/6 -1/6; -1/6 1/6 -1/6; 1/24 1/12 1/6; 1/24 -1/12 1/6; 0 0 1]
//   B^T = [4 0 -5 0 1 0; 0 -4 -4 1 1 0; 0 4 -4 -1 1 0; 0 -2 -1 2 1 0; 0 2 -1 -2 1 0; 0 4 0 -5 0 1]
//   A^T = [1 1 1 1 1 0; 0 1 -1 2 -2 0; 0 1 1 4 4 0; 0 1 -1 8 -8 1]
// Six accumulator sets = 192 registers per lane: the kernel runs ONE wave per SIMD (one workgroup of 4 waves per CU, up to
// 512 registers per lane), so nothing of another wave hides its latencies -- operand reads, the input transform (packed
// fp32, ~0.25 VALU instruction per MFMA) and all DMA waits are scheduled under its own MFMAs; only the tile epilogue is
// exposed (~5 %).  Same DMA / ring / barrier machinery as the F(2,3) kernel.
//   workgroup = 4 waves = 64 x 8 pixels (wave = two rows of 16 quads), halo 66 x 10; chunk = 16 input channels;
//   unit = (chunk, dy, three xi) = 12 KiB of weights, 48 MFMAs per wave; bias enters through the initial value of M1.
// fp32 throughout; the transforms' rounding puts the result ~4x further from the exact value than the direct kernel
// (5e-6 against 1.5e-6 at |y| ~ 5): used for the decoder, not for the extractors that feed the index search.
// =====================================================================================================================
namespace wino4 {
constexpr int KC = 16;
constexpr int TWX = 64, THY = 8;
constexpr int HWc = TWX + 2, HHr = THY + 2;
constexpr int NIN_REAL = (HWc * HHr * 4 + 63) / 64;   // 42 DMA instructions of 64 x 16 bytes
constexpr int NIN_W = (NIN_REAL + 3) / 4;             // per wave
constexpr int IN_BYTES = NIN_REAL * 1024;
constexpr int WIMG = 64 * 64;                         // one (dy, xi) weight image: 64 couts x 16 k
constexpr int XPU = 3;                                // transform positions per unit
constexpr int WUNIT = XPU * WIMG;
constexpr int NRING = 3;
constexpr int UPC = 6;                                // units per chunk: 3 row taps x 2 halves of the six positions
static_assert(UPC % NRING == 0, "ring slot of a unit must be a compile-time constant");
}  // namespace wino4

// weights W[Cout][Cin][3][3] -> Wr[cb][chunk][dy][xi 6][row 64][slot 4][e 4], slot = q ^ ((row >> 2) & 3),
// value = U[cb*64 + row][chunk*16 + 4q + e][dy][xi]
__global__ void __launch_bounds__(256) conv3x3_relayout_wino4_kernel(const float* __restrict__ w, int Cin, int Cout,
                                                                       long long total, float* __restrict__ wr) {
  const long long e0 = (long long)blockIdx.x * 256 + threadIdx.x;
  if (e0 >= total) return;
  const int e = (int)(e0 & 3), slot = (int)((e0 >> 2) & 3);
  long long r = e0 >> 4;
  const int row = (int)(r & 63); r >>= 6;
  const int xi = (int)(r % 6); r /= 6;
  const int dy = (int)(r % 3); r /= 3;
  const int nch = Cin / wino4::KC;
  const int chunk = (int)(r % nch);
  const int cb = (int)(r / nch);
  const int q = slot ^ ((row >> 2) & 3);
  const int co = cb * 64 + row, ci = chunk * wino4::KC + 4 * q + e;
  float u = 0.0f;
  if (co < Cout) {
    const float* g = w + ((size_t)co * Cin + ci) * 9 + dy * 3;
    const float g0 = g[0], g1 = g[1], g2 = g[2];
    switch (xi) {
      case 0: u = g0 * 0.25f; break;
      case 1: u = ((g0 + g1) + g2) * (-1.0f / 6.0f); break;
      case 2: u = ((g0 - g1) + g2) * (-1.0f / 6.0f); break;
      case 3: u = (g0 * (1.0f / 24.0f) + g1 * (1.0f / 12.0f)) + g2 * (1.0f / 6.0f); break;
      case 4: u = (g0 * (1.0f / 24.0f) - g1 * (1.0f / 12.0f)) + g2 * (1.0f / 6.0f); break;
      default: u = g2; break;
    }
  }
  wr[e0] = u;
}

#define C2M_W4_LOAD_A(K)                                                                                              \
  case K:                                                                                                             \
    asm volatile("ds_read_b128 %0, %2 offset:%3\n\tds_read_b128 %1, %2 offset:%4"                                     \
                 : "=&v"(a[0]), "=&v"(a[1])                                                                           \
                 : "v"(aaddr[g]), "n"((K) * wino4::WIMG), "n"((K) * wino4::WIMG + 2048)                               \
                 : "memory");                                                                                         \
    break;

template <int ABL>   // 0; > 0: timing-only ablations (wrong results): 1 no barriers / DMA waits, 2 no epilogue, 3 no halo DMA
__global__ void __launch_bounds__(256, 1) conv3x3_wino4_kernel(Params p) {
  constexpr int KC = wino4::KC, TWX = wino4::TWX, THY = wino4::THY, HWc = wino4::HWc, HHr = wino4::HHr;
  constexpr int NIN_REAL = wino4::NIN_REAL, NIN_W = wino4::NIN_W, IN_BYTES = wino4::IN_BYTES, WUNIT = wino4::WUNIT;
  constexpr int XPU = wino4::XPU, NRING = wino4::NRING, UPC = wino4::UPC;
  constexpr int MT = 2;
  extern __shared__ __attribute__((aligned(1024))) char lds[];
  // [in0 | in1 | w ring x3 | dummy 1 KiB | bias 64 floats]
  const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)lds;
  const unsigned in_base = lds0, w_base = lds0 + 2 * IN_BYTES, dummy = w_base + NRING * WUNIT, bias_lds = dummy + 1024;
  // With one wave per SIMD every VALU instruction is paid for in matrix time, so all per-lane address arithmetic that does
  // not depend on the tile is done once: the 16-byte pieces of a halo pixel are swizzled by its COLUMN ((x >> 2) & 3, not
  // by the linear pixel index), which makes a raw read address linear in the row tap (immediate offsets) and in the
  // buffer (one add per chunk), and the DMA slots' (row, column, piece) are constants of the lane.

  const int tid = threadIdx.x, l = tid & 63, hi = l >> 5, j = l & 31;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int pt = j & 15, prow = wv * 2 + (j >> 4);   // the lane's pixel quad: columns 4pt .. 4pt+3 of row prow
  const int ntile = p.tiles_x * p.tiles_y * p.B;
  const int tile_first = xcd_remap(blockIdx.x, gridDim.x) * p.tpw;
  const int ntl = min(p.tpw, ntile - tile_first);
  const int cb = blockIdx.y;
  const int UT = p.nchunks * UPC;       // units per tile
  const int G = ntl * p.nchunks;        // chunks of this workgroup
  const int T = G * UPC;                // units of this workgroup

  // ---- DMA plumbing: weights, 3 KiB per wave and unit
  const __amdgpu_buffer_rsrc_t wrsrc = make_rsrc(p.wr + (long long)cb * UT * (WUNIT / 4), (unsigned)UT * WUNIT);
  const unsigned wvoff = (wv * 192 + l) * 16;
  int wsoff = 0;
  auto issue_w = [&](int slot) __attribute__((always_inline)) {
    const unsigned dst = w_base + slot * WUNIT + wv * 3072;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(wrsrc, (__attribute__((address_space(3))) void*)dst, 16, wvoff, wsoff, 0, 0);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(wrsrc, (__attribute__((address_space(3))) void*)dst, 16, wvoff, wsoff, 1024, 0);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(wrsrc, (__attribute__((address_space(3))) void*)dst, 16, wvoff, wsoff, 2048, 0);
    wsoff += WUNIT;
    if (wsoff == UT * WUNIT) wsoff = 0;
  };
  // ---- halo tile (as conv3x3_wino_kernel): instruction n = wave + 4 * slot covers pieces [64n, 64n + 64)
  // tile coordinates are tracked incrementally (a workgroup's tiles are consecutive): integer divisions run on the vector
  // ALU, and one per chunk plus three per tile were a measurable share of the per-tile overhead
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
  unsigned ivoff[NIN_W];
  int ib = 0, iy0 = 0, ix0 = 0;
  __amdgpu_buffer_rsrc_t rs0, rs1;
  // DMA slot sl of this lane: halo pixel pl = 16 (wv + 4 sl) + (l >> 2) = (row ry, column rx), LDS slot l & 3 holds logical
  // piece (l & 3) ^ ((rx >> 2) & 3).  Packed: ry | rx << 8 | piece << 16 | valid << 24.
  int slotc[NIN_W];
#pragma unroll
  for (int sl = 0; sl < NIN_W; ++sl) {
    const int n = wv + 4 * sl, pl = 16 * n + (l >> 2);
    const int ry = pl / HWc, rx = pl - ry * HWc;
    slotc[sl] = ry | (rx << 8) | ((((l & 3) ^ ((rx >> 2) & 3))) << 16) | ((n < NIN_REAL && ry < HHr) ? (1 << 24) : 0);
  }
  auto set_source = [&](const Src& S) __attribute__((always_inline)) {
#pragma unroll
    for (int sl = 0; sl < NIN_W; ++sl) {
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
  auto issue_in = [&](int gc) __attribute__((always_inline)) {   // called for gc = 0, 1, 2, ... in order
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
    const unsigned buf = in_base + (gc & 1) * IN_BYTES;
    const int soff = (first ? c0 : c0 - p.src[0].C) * 4;
#pragma unroll
    for (int sl = 0; sl < NIN_W; ++sl) {
      const int n = wv + 4 * sl;
      const unsigned dst = n < NIN_REAL ? buf + n * 1024 : dummy;
      if (first) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs0, (__attribute__((address_space(3))) void*)dst, 16, ivoff[sl], soff, 0, 0);
      else __builtin_amdgcn_raw_ptr_buffer_load_lds(rs1, (__attribute__((address_space(3))) void*)dst, 16, ivoff[sl], soff, 0, 0);
    }
  };

  const int co_lane = cb * 64 + 4 * hi;
  if (tid < 64) {
    const int co = cb * 64 + tid;
    *(__attribute__((address_space(3))) float*)(bias_lds + tid * 4) = (p.bias && co < p.Cout) ? p.bias[co] : 0.0f;
  }
  __syncthreads();
  // accumulators: M[xi][mt]; A^T (b, 0, 0, 0, 0, 0 in M1) = (b, b, b, b): the bias is the initial value of M1
  f32x16 M[6][MT];
  auto init_m = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int qd = 0; qd < 4; ++qd) {
        const f32x4 bv = *(const __attribute__((address_space(3))) f32x4*)(bias_lds + (mt * 32 + 8 * qd + 4 * hi) * 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
#pragma unroll
          for (int xi = 0; xi < 6; ++xi) M[xi][mt][4 * qd + e] = 0.0f;
          M[1][mt][4 * qd + e] = bv[e];
        }
      }
  };
  init_m();

  // A operand: row = cout j (+32 mt), piece 2g + hi, slot q ^ ((row >> 2) & 3); + ring slot * WUNIT + xi_local * WIMG
  unsigned aaddr[2];
#pragma unroll
  for (int g = 0; g < 2; ++g) aaddr[g] = w_base + j * 64 + (((2 * g + hi) ^ ((j >> 2) & 3)) << 4);
  auto load_a = [&](int uc, int st, f32x4 (&a)[MT]) __attribute__((always_inline)) {   // st = 2 * xi_local + g
    const int g = st & 1;
    switch ((uc % NRING) * XPU + (st >> 1)) {
      C2M_W4_LOAD_A(0) C2M_W4_LOAD_A(1) C2M_W4_LOAD_A(2) C2M_W4_LOAD_A(3) C2M_W4_LOAD_A(4)
      C2M_W4_LOAD_A(5) C2M_W4_LOAD_A(6) C2M_W4_LOAD_A(7) C2M_W4_LOAD_A(8)
      default: break;
    }
  };
  auto wait_a = [&](auto n, f32x4 (&a)[MT]) __attribute__((always_inline)) {
    constexpr int N = decltype(n)::value;
    asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(a[0]), "+v"(a[1]) : "n"(N));
  };
  // input transform of one row tap: six raw b128 reads per k-quad (input columns 4t-1 .. 4t+4 of the lane's quad)
  f32x4 V[6][2], d[2][6];
  // raw read addresses of row tap 0 in the CURRENT halo buffer; row tap dy = immediate offset dy * HWc * 64
  unsigned raddr[2][6];
#pragma unroll
  for (int g = 0; g < 2; ++g)
#pragma unroll
    for (int bcol = 0; bcol < 6; ++bcol) {
      const int x = 4 * pt + bcol;
      raddr[g][bcol] = in_base + (prow * HWc + x) * 64 + ((((2 * g + hi) ^ (x >> 2)) & 3) << 4);
    }
  int rdelta = IN_BYTES;   // to the other buffer
  auto issue_raw = [&](int dy) __attribute__((always_inline)) {
#pragma unroll
    for (int g = 0; g < 2; ++g)
#pragma unroll
      for (int bcol = 0; bcol < 6; ++bcol) {
        switch (dy) {
          case 0: asm volatile("ds_read_b128 %0, %1" : "=v"(d[g][bcol]) : "v"(raddr[g][bcol]) : "memory"); break;
          case 1: asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(d[g][bcol]) : "v"(raddr[g][bcol]), "n"(HWc * 64) : "memory"); break;
          default: asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(d[g][bcol]) : "v"(raddr[g][bcol]), "n"(2 * HWc * 64) : "memory"); break;
        }
      }
  };
  auto next_buffer = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int g = 0; g < 2; ++g)
#pragma unroll
      for (int bcol = 0; bcol < 6; ++bcol) raddr[g][bcol] += rdelta;
    rdelta = -rdelta;
  };
  auto wait_raw = [&]() __attribute__((always_inline)) {
    asm volatile("s_waitcnt lgkmcnt(0)"
                 : "+v"(d[0][0]), "+v"(d[0][1]), "+v"(d[0][2]), "+v"(d[0][3]), "+v"(d[0][4]), "+v"(d[0][5]), "+v"(d[1][0]),
                   "+v"(d[1][1]), "+v"(d[1][2]), "+v"(d[1][3]), "+v"(d[1][4]), "+v"(d[1][5]));
  };
  auto vcomp = [&]() __attribute__((always_inline)) {
    const f32x4 c4 = {4.0f, 4.0f, 4.0f, 4.0f}, cm4 = {-4.0f, -4.0f, -4.0f, -4.0f}, cm5 = {-5.0f, -5.0f, -5.0f, -5.0f},
                c2 = {2.0f, 2.0f, 2.0f, 2.0f}, cm2 = {-2.0f, -2.0f, -2.0f, -2.0f};
#pragma unroll
    for (int g = 0; g < 2; ++g) {
      const f32x4 t1 = __builtin_elementwise_fma(cm4, d[g][2], d[g][4]);   // d4 - 4 d2
      const f32x4 t2 = __builtin_elementwise_fma(cm4, d[g][1], d[g][3]);   // d3 - 4 d1
      const f32x4 t3 = d[g][4] - d[g][2];
      const f32x4 t4 = d[g][3] - d[g][1];
      V[0][g] = __builtin_elementwise_fma(c4, d[g][0], __builtin_elementwise_fma(cm5, d[g][2], d[g][4]));
      V[1][g] = t1 + t2;
      V[2][g] = t1 - t2;
      V[3][g] = __builtin_elementwise_fma(c2, t4, t3);
      V[4][g] = __builtin_elementwise_fma(cm2, t4, t3);
      V[5][g] = __builtin_elementwise_fma(c4, d[g][1], __builtin_elementwise_fma(cm5, d[g][3], d[g][5]));
    }
  };

  issue_in(0);
  if (G > 1) issue_in(1);
  issue_w(0);
  issue_w(1);
  issue_w(2);
  wait_vmcnt<0>();
  __builtin_amdgcn_s_barrier();

  f32x4 a_s[2][MT];
  issue_raw(0);
  for (int it = 0, gc = 0; it < ntl; ++it) {
    for (int c = 0; c < p.nchunks; ++c, ++gc) {
      const bool more_in = gc + 1 < G;
#pragma unroll
      for (int uc = 0; uc < UPC; ++uc) {
        const int gu = gc * UPC + uc;
        const int dy = uc >> 1, xh = uc & 1;          // row tap, which half of the transform positions (xi = 3 xh ..)
        const bool next_group = gu + 2 - xh < T;       // a group (chunk, dy) follows this one
        if (xh == 0) {
          wait_raw();
          load_a(uc, 0, a_s[0]);
          vcomp();
        }
#pragma unroll
        for (int st = 0; st < 2 * XPU; ++st) {         // step = (xi_local, g)
          const int cur = st & 1, nxt = cur ^ 1;
          const int xi = XPU * xh + (st >> 1), g = st & 1;
          if (st < 2 * XPU - 1) {
            load_a(uc, st + 1, a_s[nxt]);
            wait_a(std::integral_constant<int, 2>(), a_s[cur]);
          } else if (xh == 1) {
            // last step of the group: the next group's raw values are fetched now (its operands at the top of its first unit)
            if (next_group) {
              if (dy < 2) {
                issue_raw(dy + 1);
              } else {
                next_buffer();     // row tap 0 of the next chunk: the other halo buffer
                issue_raw(0);
              }
              wait_a(std::integral_constant<int, 12>(), a_s[cur]);
            } else {
              wait_a(std::integral_constant<int, 0>(), a_s[cur]);
            }
          } else {
            load_a(uc + 1, 0, a_s[nxt]);               // next unit of the same group
            wait_a(std::integral_constant<int, 2>(), a_s[cur]);
          }
#pragma unroll
          for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
              M[xi][mt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_s[cur][mt][e], V[xi][g][e], M[xi][mt], 0, 0, 0);
          __builtin_amdgcn_sched_barrier(0);
        }
        if (gu + 1 < T) {
          if constexpr (ABL != 1) {
            if (uc == 0 && gc >= 1 && more_in) wait_vmcnt<NIN_W>();
            else wait_vmcnt<0>();
            __builtin_amdgcn_s_barrier();
          }
          if (gu + 3 < T) issue_w(uc % NRING);
          if (ABL != 3 && uc == UPC - 1 && gc + 2 < G) issue_in(gc + 2);
        }
      }
    }
    // ---- tile epilogue: Y = A^T M (bias already inside M1), activation, residuals, four channels-last pixels per lane
    int b = epi_tc.b, y0 = epi_tc.ty * THY, x0 = epi_tc.tx * TWX;
    tc_next(epi_tc);
    asm volatile("" : "+s"(b), "+s"(y0), "+s"(x0));
    const int y = y0 + prow, x = x0 + 4 * pt;
    const bool pok = y < p.H && x < p.W;   // W % 64 == 0: the quad is inside or outside as a whole
    const size_t opix = (size_t)b * p.out_img_pitch + (size_t)y * p.out_row_pitch + (size_t)x * p.out_pix_pitch;
    const float* r1 = p.res1 ? p.res1 + opix + co_lane : nullptr;
    const float* r2 = p.res2 ? p.res2 + opix + co_lane : nullptr;
    float* ob = p.out + opix + co_lane;
    if (ABL == 2 && it > 0) { init_m(); continue; }
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int qd = 0; qd < 4; ++qd) {
        const int co = co_lane + mt * 32 + 8 * qd;
        f32x4 yv[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int r = 4 * qd + e;
          const float m0 = M[0][mt][r], m1 = M[1][mt][r], m2 = M[2][mt][r], m3 = M[3][mt][r], m4 = M[4][mt][r], m5 = M[5][mt][r];
          const float s12 = m1 + m2, d12 = m1 - m2, s34 = m3 + m4, d34 = m3 - m4;
          yv[0][e] = (m0 + s12) + s34;
          yv[1][e] = __builtin_fmaf(2.0f, d34, d12);
          yv[2][e] = __builtin_fmaf(4.0f, s34, s12);
          yv[3][e] = __builtin_fmaf(8.0f, d34, d12) + m5;
        }
        if (p.act == 1) {
#pragma unroll
          for (int k = 0; k < 4; ++k)
#pragma unroll
            for (int e = 0; e < 4; ++e) yv[k][e] = fmaxf(yv[k][e], 0.0f);
        } else if (p.act == 2) {
#pragma unroll
          for (int k = 0; k < 4; ++k)
#pragma unroll
            for (int e = 0; e < 4; ++e) yv[k][e] = fmaxf(yv[k][e], yv[k][e] * p.slope);
        }
        if (pok && co + 3 < p.Cout) {
          const int o = mt * 32 + 8 * qd;
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            if (r1) yv[k] += *reinterpret_cast<const f32x4*>(r1 + k * p.out_pix_pitch + o);
            if (r2) yv[k] += *reinterpret_cast<const f32x4*>(r2 + k * p.out_pix_pitch + o);
            *reinterpret_cast<f32x4*>(ob + k * p.out_pix_pitch + o) = yv[k];
          }
        }
      }
    init_m();
  }
}
#undef C2M_W4_LOAD_A

// ---------------------------------------------------------------------------------------------------------------------
// First layer of an image tower: 3 input channels (vgg conv1_1, vgg_arch.py:107-123; conv_first, ref_restoration_arch.py:30)
// -> 64 output channels.  K = 27 does not fill the 32-channel chunks of conv3x3_kernel (which then multiplies 29 zero
// channels per tap: 2.1 ms for a 640 x 640 batch of 16), so this layer is an im2col GEMM of its own: 14 k-pairs per
// 32 pixels x 64 channels, weights resident in registers, image tile (with the (x - mean) / std of the extractors applied
// while it is staged; padding is zero in the NORMALISED domain, as in the reference) in LDS.  Memory-bound on its output.
//   K order (so that both half-waves read LDS with one base register + immediates): pairs t = 0..8: (c, dy) = (t/3, t%3),
//   hi -> dx = hi;  t = 9..11: c = t-9, dx = 2, hi -> dy = hi;  t = 12: dx = 2, dy = 2, hi -> c = hi;  t = 13: hi = 0 ->
//   (c, dy, dx) = (2, 2, 2), hi = 1 -> zero weight.
// ---------------------------------------------------------------------------------------------------------------------
namespace c3 {
constexpr int CTW = 64, CTH = 8;                 // pixel tile of a workgroup (wave = two rows)
constexpr int CHW = CTW + 2, CHH = CTH + 2;      // halo tile
constexpr int PLANE = CHW * CHH;               // floats per staged channel
constexpr int NEL = 3 * PLANE;                 // 1980
constexpr int EPT = (NEL + 255) / 256;         // staged elements per thread (8)
struct Params {
  const float* in;      // [B][3][H][W]
  const float* w;       // [64][3][3][3]
  const float* bias;    // [64] or nullptr
  const float* mean;    // [3] or nullptr: (x - mean[c]) / std[c] applied to the image first
  const float* std_;
  int B, H, W, tiles_x, tiles_y;
  int act;
  float slope;
  float* out;           // channels-last, pitches in floats
  int out_pix_pitch, out_row_pitch;
  long long out_img_pitch;
  float* out2;          // optional 8-channel group-major twin (see conv::Params::out2)
  int out2_row_pitch;
  long long out2_plane_pitch, out2_img_pitch;
  unsigned out_bytes, out2_bytes;   // bytes one TILE ROW's stores may touch, from its own base (buffer-descriptor records; < 2^31)
};
}  // namespace c3

// C2M_C3_ABL (compile-time, measurement builds only; scripts/abl_c3.py): 1 one store of eight, 2 no MFMAs, 4 the image tile is
// fetched and staged once per workgroup
#ifndef C2M_C3_ABL
#define C2M_C3_ABL 0
#endif
template <int ACT, bool OUT2>
__global__ void __launch_bounds__(256, 2) conv3x3_c3_kernel(c3::Params p) {
  using namespace c3;
  typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
  __shared__ float tile[2][NEL];
  const int tid = threadIdx.x, l = tid & 63, hi = l >> 5, j = l & 31;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);

  // ---- weights: A operand of k-pair t, m-tile mt = W[mt*32 + j][k(t, hi)]
  float wreg[14][2];
#pragma unroll
  for (int t = 0; t < 14; ++t) {
    int c, dy, dx;
    bool live = true;
    if (t < 9) { c = t / 3; dy = t % 3; dx = hi; }
    else if (t < 12) { c = t - 9; dy = hi; dx = 2; }
    else if (t == 12) { c = hi; dy = 2; dx = 2; }
    else { c = 2; dy = 2; dx = 2; live = hi == 0; }
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) wreg[t][mt] = live ? p.w[(size_t)(mt * 32 + j) * 27 + c * 9 + dy * 3 + dx] : 0.0f;
  }
  // LDS byte bases of this lane for the four pair families (pixel (row 0, column j) of the wave's first row, tap (0, 0, 0))
  const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) float*)&tile[0][0];
  const unsigned pix = (unsigned)(j * 4);
  const unsigned bA = pix + hi * 4;                 // dx = hi
  const unsigned bB = pix + hi * (CHW * 4) + 8;     // dy = hi, dx = 2
  const unsigned bC = pix + hi * (PLANE * 4) + (2 * CHW + 2) * 4;   // c = hi, (dy, dx) = (2, 2)
  const unsigned bD = pix + (2 * PLANE + 2 * CHW + 2) * 4;          // (2, 2, 2) (hi = 1 multiplies a zero weight)

  // ---- staging plan: element e = tid + 256 i of the halo tile -> (channel, halo row, halo column)
  int s_off[EPT], s_ry[EPT], s_rx[EPT], s_c[EPT];
#pragma unroll
  for (int i = 0; i < EPT; ++i) {
    const int e = tid + 256 * i;
    const int c = e / PLANE, rem = e - c * PLANE;
    s_c[i] = e < NEL ? c : -1;
    s_ry[i] = rem / CHW;
    s_rx[i] = rem - s_ry[i] * CHW;
    s_off[i] = e;
  }
  float mean_[3] = {0.0f, 0.0f, 0.0f}, std3[3] = {1.0f, 1.0f, 1.0f};
  if (p.mean) {
#pragma unroll
    for (int c = 0; c < 3; ++c) { mean_[c] = p.mean[c]; std3[c] = p.std_[c]; }
  }
  const int ntile = p.tiles_x * p.tiles_y * p.B;
  // The next tile's pixels are fetched RAW with buffer loads (one descriptor per image; a halo element outside the image gets an
  // offset beyond the records and reads 0.0) right after the barrier and normalised only when they are written to LDS, one
  // tile later: no branch, no wait between the eight loads, and -- the stores below being unconditional buffer stores as well --
  // a counted wait (`vmcnt(stores issued since)`) in front of the staging writes instead of a drain of the tile's 32 stores.
  float stage[EPT];
  unsigned okm = 0u;   // bit i: element i of the staged tile lies inside the image
  const unsigned in_bytes = (unsigned)(3 * p.H * p.W) * 4u;
  auto fetch = [&](int t) __attribute__((always_inline)) {
    const int tx = t % p.tiles_x, ty = (t / p.tiles_x) % p.tiles_y, b = t / (p.tiles_x * p.tiles_y);
    const __amdgpu_buffer_rsrc_t irs =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.in + (size_t)b * 3 * p.H * p.W), 0, (int)in_bytes, 0x00020000);
    okm = 0u;
#pragma unroll
    for (int i = 0; i < EPT; ++i) {
      const int y = ty * CTH - 1 + s_ry[i], x = tx * CTW - 1 + s_rx[i];
      const unsigned bad = (s_c[i] < 0) | (y < 0) | (y >= p.H) | (x < 0) | (x >= p.W);
      const unsigned off = (unsigned)((s_c[i] * p.H + y) * p.W + x) * 4u;
      stage[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(irs, off | (bad << 31), 0, 0));
      okm |= (bad ^ 1u) << i;
    }
  };
  auto staged = [&](int i) __attribute__((always_inline)) {   // (x - mean) / std; the padding is zero in the normalised domain
    // (without mean / std: (x - 0) / 1, exact).  Element tid + 256 i lies in channel (256 i) / PLANE or the next one: the
    // constants are compile-time per round, or one select -- not a three-way choice hipcc turns into divergent branches
    const int c_lo = (256 * i) / PLANE, c_hi = (256 * i + 255) / PLANE > 2 ? 2 : (256 * i + 255) / PLANE;
    const bool up = c_hi != c_lo && tid + 256 * i >= c_hi * PLANE;
    const float m = up ? mean_[c_hi] : mean_[c_lo], sd = up ? std3[c_hi] : std3[c_lo];
    const float v = (stage[i] - m) / sd;
    return ((okm >> i) & 1u) ? v : 0.0f;
  };
  float bias4[2][4][4];
#pragma unroll
  for (int mt = 0; mt < 2; ++mt)
#pragma unroll
    for (int qd = 0; qd < 4; ++qd)
#pragma unroll
      for (int e = 0; e < 4; ++e) bias4[mt][qd][e] = p.bias ? p.bias[mt * 32 + 8 * qd + 4 * hi + e] : 0.0f;

  // every prologue load is waited for HERE: a register still pending at the loop entry would make hipcc wait for ALL memory
  // traffic (vmcnt(0): the tile fetch just issued included) at its first use in every iteration
#pragma unroll
  for (int k = 0; k < 14; ++k) asm volatile("" ::"v"(wreg[k][0]), "v"(wreg[k][1]));
#pragma unroll
  for (int mt = 0; mt < 2; ++mt)
#pragma unroll
    for (int qd = 0; qd < 4; ++qd) asm volatile("" ::"v"(bias4[mt][qd][0]), "v"(bias4[mt][qd][1]), "v"(bias4[mt][qd][2]), "v"(bias4[mt][qd][3]));
  asm volatile("" ::"v"(mean_[0]), "v"(mean_[1]), "v"(mean_[2]), "v"(std3[0]), "v"(std3[1]), "v"(std3[2]));
  // Loop shape: the FIRST tile is fetched and staged here; iteration `it` = barrier, fetch of tile it + 1 (raw), the four groups
  // of tile `it` with their 32 (64) stores, then -- in the same straight-line body, so that hipcc can count the stores issued
  // since the fetch instead of draining them -- the normalised staging writes of tile it + 1 into the other buffer, which was
  // last read in iteration it - 1, i.e. before this iteration's barrier.  Past the last tile the fetch repeats the last tile
  // (unconditional: no branch for the wait count to be merged over) into a buffer nobody reads.
  auto stage_to = [&](float* buf) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < EPT; ++i)
      if (256 * (i + 1) <= NEL || s_c[i] >= 0) buf[s_off[i]] = staged(i);   // (only the last round is partial)
  };
  int t = blockIdx.x;
  if (t < ntile) {
    fetch(t);
    stage_to(tile[0]);
  }
  for (int it = 0; t < ntile; t += gridDim.x, ++it) {
    __syncthreads();   // tile `it` staged; everybody has left tile it - 1
    if (!(C2M_C3_ABL & 4)) fetch(min(t + (int)gridDim.x, ntile - 1));
    const int tx = t % p.tiles_x, ty = (t / p.tiles_x) % p.tiles_y, b = t / (p.tiles_x * p.tiles_y);
    const unsigned tb = lds0 + ((C2M_C3_ABL & 4) ? 0 : it & 1) * (NEL * 4);
    // store descriptors start at this tile's first pixel row (and, for the twin, at each 8-channel plane): the 32-bit offsets
    // then span eight rows whatever the image size
    float* const obase = p.out + (size_t)b * p.out_img_pitch + (size_t)(ty * CTH) * p.out_row_pitch;
    const __amdgpu_buffer_rsrc_t ors = __builtin_amdgcn_make_buffer_rsrc(obase, 0, (int)p.out_bytes, 0x00020000);
    float* const obase2 = OUT2 ? p.out2 + (size_t)b * p.out2_img_pitch + (size_t)(ty * CTH) * p.out2_row_pitch : obase;
#pragma unroll
    for (int g = 0; g < 4; ++g) {   // the wave's groups: rows 2 wv + (g >> 1), columns 32 (g & 1) + j
      const int row = 2 * wv + (g >> 1), col = 32 * (g & 1);
      const unsigned gb = tb + (row * CHW + col) * 4;
      float bv[14];
#pragma unroll
      for (int k = 0; k < 9; ++k)
        bv[k] = *(const __attribute__((address_space(3))) float*)(gb + bA + ((k / 3) * PLANE + (k % 3) * CHW) * 4);
#pragma unroll
      for (int k = 0; k < 3; ++k) bv[9 + k] = *(const __attribute__((address_space(3))) float*)(gb + bB + k * PLANE * 4);
      bv[12] = *(const __attribute__((address_space(3))) float*)(gb + bC);
      bv[13] = *(const __attribute__((address_space(3))) float*)(gb + bD);
      f32x16 acc[2];
#pragma unroll
      for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[mt][r] = 0.0f;
#pragma unroll
      for (int k = 0; k < 14; ++k)
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
          if (C2M_C3_ABL & 2) acc[mt][k] += wreg[k][mt] * bv[k];
          else acc[mt] = __builtin_amdgcn_mfma_f32_32x32x2f32(wreg[k][mt], bv[k], acc[mt], 0, 0, 0);
        }
      const int y = ty * CTH + row, x = tx * CTW + col + j;
      {
        const unsigned bad = (unsigned)(y >= p.H) | (unsigned)(x >= p.W);
        const unsigned ob = ((unsigned)(row * p.out_row_pitch + x * p.out_pix_pitch + 4 * hi) * 4u) | (bad << 31);
        const unsigned ob2 = ((unsigned)(row * p.out2_row_pitch + x * 8 + 4 * hi) * 4u) | (bad << 31);
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
          for (int qd = 0; qd < 4; ++qd) {
            f32x4 v;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              v[e] = acc[mt][4 * qd + e] + bias4[mt][qd][e];
              if (ACT == 1) v[e] = fmaxf(v[e], 0.0f);
              else if (ACT == 2) v[e] = fmaxf(v[e], v[e] * p.slope);
            }
            if ((C2M_C3_ABL & 1) && (mt + qd != 0) && v[0] != 12345.678f) continue;
            // (the channel offset goes into the instruction's IMMEDIATE offset, the plane offset into the vector offset: a 128-bit
            // buffer store with an SGPR soffset whose data registers a VALU overwrites within two issue slots stores garbage in
            // lanes 12..15 / 28..31 of each half on gfx950 -- hipcc pads only the immediate-soffset form of that hazard)
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, v), ors, ob + (unsigned)((mt * 32 + 8 * qd) * 4), 0, 0);
            if (OUT2) {   // plane co >> 3 = 4 mt + qd is the same for the whole wave: it goes into the descriptor's base
              const __amdgpu_buffer_rsrc_t ors2 = __builtin_amdgcn_make_buffer_rsrc(
                  obase2 + (size_t)(mt * 4 + qd) * p.out2_plane_pitch, 0, (int)p.out2_bytes, 0x00020000);
              __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, v), ors2, ob2, 0, 0);
            }
          }
      }
    }
    if (!(C2M_C3_ABL & 4)) stage_to(tile[(it + 1) & 1]);
  }
}

}  // namespace conv
}  // namespace c2m

// =====================================================================================================================
// C-ABI
// =====================================================================================================================
using namespace c2m;

namespace c2m {
namespace conv {   // conv3x3_split.hip
size_t split_relayout_bytes(int Cin, int Cout, int np);
int split_relayout(hipStream_t st, const float* weight, int Cin, int Cout, int np, void* wr, int dgrad);
int launch_split(hipStream_t st, Params p, int np);
#ifdef C2M_EXPERIMENTAL
int launch_wino16(hipStream_t st, Params p, int R);   // experimental/conv3x3_wino16.hip (make EXPERIMENTAL=1)
#else
static inline int launch_wino16(hipStream_t, Params, int) { return C2M_ERR_UNSUPPORTED; }   // measured no-go: not in the product library
#endif
int set_head_stores(int mode);
int split_relayout_multi(hipStream_t st, const long long* jobs, int njobs, long long nblocks, int any_f16);
}  // namespace conv
}  // namespace c2m

extern "C" size_t c2m_conv3x3_relayout_split_bytes(int Cin, int Cout, int pieces) { return conv::split_relayout_bytes(Cin, Cout, pieces); }

extern "C" int c2m_conv3x3_relayout_split_f32(c2m_stream_t stream, const float* weight, int Cin, int Cout, int pieces, void* wr) {
  if (!weight || !wr) return C2M_ERR_INVALID_ARG;
  return conv::split_relayout(as_stream(stream), weight, Cin, Cout, pieces, wr, 0);
}

extern "C" int c2m_conv3x3_relayout_split_multi(c2m_stream_t stream, const long long* jobs, int njobs, long long nblocks, int any_f16) {
  return conv::split_relayout_multi(as_stream(stream), jobs, njobs, nblocks, any_f16);
}

extern "C" int c2m_conv3x3_relayout_split_dgrad_f32(c2m_stream_t stream, const float* weight, int Cin, int Cout, int pieces, void* wr) {
  if (!weight || !wr) return C2M_ERR_INVALID_ARG;
  // the data-gradient convolution maps Cout channels (of dY) to Cin channels (of dX)
  return conv::split_relayout(as_stream(stream), weight, Cout, Cin, pieces, wr, 1);
}

namespace {
inline int conv_mw(int Cout) { return Cout <= 32 ? 32 : 64; }
inline long long relayout_elems(int Cin, int Cout) {
  const int MW = conv_mw(Cout), ncb = (Cout + MW - 1) / MW;
  return (long long)ncb * (Cin / conv::KCH) * 9 * MW * 32;
}
}  // namespace

#ifndef C2M_EXPERIMENTAL
// csrc/experimental/conv3x3_resblock.hip (a whole ResidualBlockNoBN in one launch) is a measured no-go: correct,
// 6 - 7 % slower than two launches of the split kernel.  It is built by `make EXPERIMENTAL=1` only; the product library answers
// "unsupported" and callers (c2m_amd.ops.resblock3x3_wanted) stay on the two-launch path.
extern "C" int c2m_resblock3x3_supported(int, int, int) { return 0; }
extern "C" int c2m_resblock3x3_nhwc_f32(c2m_stream_t, const c2m_resblock3x3_desc*) { return C2M_ERR_UNSUPPORTED; }
#endif

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

extern "C" size_t c2m_conv3x3_relayout_wino4_bytes(int Cin, int Cout) {
  if (Cin <= 0 || Cout <= 0 || Cin % conv::wino4::KC != 0 || Cout % 64 != 0) return 0;
  return (size_t)(Cout / 64) * (Cin / conv::wino4::KC) * 18 * conv::wino4::WIMG;   // 3 row taps x 6 transform positions
}

extern "C" int c2m_conv3x3_relayout_wino4_f32(c2m_stream_t stream, const float* weight, int Cin, int Cout, float* wr) {
  if (!weight || !wr) return C2M_ERR_INVALID_ARG;
  const size_t bytes = c2m_conv3x3_relayout_wino4_bytes(Cin, Cout);
  if (bytes == 0) return C2M_ERR_UNSUPPORTED;
  const long long total = (long long)(bytes / 4);
  hipLaunchKernelGGL(conv::conv3x3_relayout_wino4_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, as_stream(stream),
                     weight, Cin, Cout, total, wr);
  return check_launch();
}

extern "C" size_t c2m_conv3x3_relayout_wino_bytes(int Cin, int Cout) {
  if (Cin <= 0 || Cout <= 0 || Cin % conv::wino::KC != 0 || Cout % 64 != 0) return 0;
  return (size_t)(Cout / 64) * (Cin / conv::wino::KC) * 12 * conv::wino::WIMG;   // 3 row taps x 4 transform positions
}

extern "C" int c2m_conv3x3_relayout_wino_f32(c2m_stream_t stream, const float* weight, int Cin, int Cout, float* wr) {
  if (!weight || !wr) return C2M_ERR_INVALID_ARG;
  const size_t bytes = c2m_conv3x3_relayout_wino_bytes(Cin, Cout);
  if (bytes == 0) return C2M_ERR_UNSUPPORTED;
  const long long total = (long long)(bytes / 4);
  hipLaunchKernelGGL(conv::conv3x3_relayout_wino_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, as_stream(stream),
                     weight, Cin, Cout, total, wr);
  return check_launch();
}

extern "C" int c2m_conv3x3_rgb64_f32(c2m_stream_t stream, const float* image, int B, int H, int W, const float* weight,
                                     const float* bias, const float* mean, const float* std_, int act, float slope, float* out,
                                     int out_pix_pitch, int out_row_pitch, long long out_img_pitch, float* out2,
                                     int out2_row_pitch, long long out2_plane_pitch, long long out2_img_pitch) {
  if (!image || !weight || !out || B <= 0 || H <= 0 || W <= 0 || (mean == nullptr) != (std_ == nullptr)) return C2M_ERR_INVALID_ARG;
  if (act < 0 || act > 2 || (act == C2M_ACT_LEAKY_RELU && !(slope >= 0.0f && slope <= 1.0f))) return C2M_ERR_UNSUPPORTED;
  if (out_pix_pitch % 4 != 0 || out_row_pitch % 4 != 0 || out_img_pitch % 4 != 0 || ((uintptr_t)out & 15)) return C2M_ERR_UNSUPPORTED;
  if (out2 && (out2_row_pitch % 4 != 0 || out2_plane_pitch % 4 != 0 || out2_img_pitch % 4 != 0 || ((uintptr_t)out2 & 15)))
    return C2M_ERR_UNSUPPORTED;
  conv::c3::Params p;
  p.in = image; p.w = weight; p.bias = bias; p.mean = mean; p.std_ = std_;
  p.B = B; p.H = H; p.W = W;
  p.tiles_x = ceil_div(W, conv::c3::CTW); p.tiles_y = ceil_div(H, conv::c3::CTH);
  p.act = act; p.slope = slope; p.out = out; p.out_pix_pitch = out_pix_pitch; p.out_row_pitch = out_row_pitch;
  p.out_img_pitch = out_img_pitch; p.out2 = out2; p.out2_row_pitch = out2_row_pitch; p.out2_plane_pitch = out2_plane_pitch;
  p.out2_img_pitch = out2_img_pitch;
  const long long ntile = (long long)p.tiles_x * p.tiles_y * B;
  if (ntile > 0x7fffffffLL) return C2M_ERR_INVALID_ARG;
  // buffer addressing inside one image (loads) / one row of tiles (stores): every byte offset, and bit 31 as the "outside" mark,
  // must fit 32 bits
  const long long in_b = 12LL * H * W,
                  out_b = 4LL * ((long long)(conv::c3::CTH - 1) * out_row_pitch + (long long)(W - 1) * out_pix_pitch + 64),
                  out2_b = out2 ? 4LL * ((long long)(conv::c3::CTH - 1) * out2_row_pitch + 8LL * W) : 0;
  if (in_b >= (1LL << 31) || out_b >= (1LL << 31) || out2_b >= (1LL << 31) || out_row_pitch < 0 || out_pix_pitch < 0) return C2M_ERR_UNSUPPORTED;
  p.out_bytes = (unsigned)out_b; p.out2_bytes = (unsigned)out2_b;
  hipStream_t st = as_stream(stream);
  ProfileScope prof(C2M_KERNEL_CONV3X3, st);
  // persistent workgroups (the weights stay in registers): 3 fit a CU (164 VGPRs), tiles strided over them
  void (*kern)(conv::c3::Params) =
      out2 ? (act == 0 ? &conv::conv3x3_c3_kernel<0, true> : act == 1 ? &conv::conv3x3_c3_kernel<1, true> : &conv::conv3x3_c3_kernel<2, true>)
           : (act == 0 ? &conv::conv3x3_c3_kernel<0, false> : act == 1 ? &conv::conv3x3_c3_kernel<1, false> : &conv::conv3x3_c3_kernel<2, false>);
  hipLaunchKernelGGL(kern, dim3((unsigned)std::min<long long>(ntile, 768)), dim3(256), 0, st, p);
  return check_launch();
}

extern "C" int c2m_index_to_flow_f32(c2m_stream_t stream, const int64_t* max_idx, int B, int hq, int wq, float* flow) {
  if (!max_idx || !flow || B <= 0 || hq <= 0 || wq <= 0) return C2M_ERR_INVALID_ARG;
  const int n = B * hq * wq;
  hipLaunchKernelGGL(conv::index_to_flow_kernel, dim3(ceil_div(n, 256)), dim3(256), 0, as_stream(stream), max_idx, n, hq,
                     wq, reinterpret_cast<float2*>(flow));
  return check_launch();
}

extern "C" int c2m_conv3x3_set_head_stores(int mode) { return conv::set_head_stores(mode); }

extern "C" int c2m_conv3x3_nhwc_f32(c2m_stream_t stream, const c2m_conv3x3_desc* d) {
  if (!d || !d->wr || !d->out || d->B <= 0 || d->H <= 0 || d->W <= 0 || d->Cin <= 0 || d->Cout <= 0 || d->nsrc < 1 ||
      d->nsrc > 2 || !d->src[0].ptr || (d->nsrc == 2 && !d->src[1].ptr))
    return C2M_ERR_INVALID_ARG;
  const bool wino4 = d->algo == C2M_CONV_WINOGRAD_F43X;
  const bool wino = d->algo == C2M_CONV_WINOGRAD_F23X || wino4;   // both: 16-channel chunks, 64-cout blocks
  const bool wino16 = d->algo == C2M_CONV_WINO_F16X2_F43Y || d->algo == C2M_CONV_WINO_F16X2_F23Y;   // conv3x3_wino16.hip
  const bool splitk = d->algo == C2M_CONV_SPLIT_BF16X3 || d->algo == C2M_CONV_BF16 || d->algo == C2M_CONV_SPLIT_F16X2 || wino16;   // 16-channel chunks, any shape
  if (wino16 && (d->out_mode != 0 || d->Cout % 64 != 0 || d->out2 || d->io_flags != 0)) return C2M_ERR_UNSUPPORTED;
  if (d->algo != 0 && !wino && !splitk) return C2M_ERR_INVALID_ARG;
  if (splitk && (d->out2 || (d->out_mode == 4 && (d->H % 2 != 0 || d->W % 2 != 0 || d->res1 || d->res2)))) return C2M_ERR_UNSUPPORTED;
  if (wino && ((d->out_mode != 0 && d->out_mode != 3 && d->out_mode != 4) || d->Cout % 64 != 0 || d->W % 32 != 0)) return C2M_ERR_UNSUPPORTED;
  if (d->out_mode == 4 && !splitk && (d->algo != C2M_CONV_WINOGRAD_F23X || d->H % 2 != 0 || d->res1 || d->res2 || d->out2)) return C2M_ERR_UNSUPPORTED;
  if (wino4 && (d->out_mode != 0 || d->W % 64 != 0)) return C2M_ERR_UNSUPPORTED;
  const int kch = (wino || splitk) ? conv::wino::KC : conv::KCH;
  int csum = 0;
  for (int s = 0; s < d->nsrc; ++s) {
    if (d->src[s].C <= 0 || d->src[s].C % kch != 0 || d->src[s].pix_pitch % 4 != 0 || d->src[s].row_pitch % 4 != 0 ||
        d->src[s].img_pitch % 4 != 0 || ((uintptr_t)d->src[s].ptr & 15))
      return C2M_ERR_UNSUPPORTED;   // 16-byte pieces: every pitch a multiple of 4 floats, 32-channel chunks
    csum += d->src[s].C;
  }
  if (csum != d->Cin || d->H >= 32768 || d->W >= 65536) return C2M_ERR_INVALID_ARG;
  if (d->out_mode < 0 || d->out_mode > 4) return C2M_ERR_INVALID_ARG;
  const bool out_vec4 = !(d->out_pix_pitch % 4 != 0 || d->out_row_pitch % 4 != 0 || d->out_img_pitch % 4 != 0 ||
                          ((uintptr_t)d->out & 15) || ((uintptr_t)d->res1 & 15) || ((uintptr_t)d->res2 & 15));
  if (d->out_mode == 1 && d->Cout % 4 != 0) return C2M_ERR_INVALID_ARG;
  const int cout_total = d->cout_total > 0 ? d->cout_total : d->Cout;
  if (d->out_mode == 3 && (!d->mask_out || d->n_off <= 0 || d->n_off % 4 != 0 || d->n_off >= cout_total || d->scale <= 0 ||
                           d->cout_offset < 0 || d->cout_offset % 4 != 0 || d->cout_offset + d->Cout > cout_total ||
                           (d->flow && (d->fh <= 0 || d->fw <= 0))))
    return C2M_ERR_INVALID_ARG;
  if ((d->res1 || d->res2) && d->out_mode != 0) return C2M_ERR_UNSUPPORTED;

  conv::Params p;
  p.B = d->B; p.H = d->H; p.W = d->W; p.Cin = d->Cin; p.Cout = d->Cout;
  // 32 x 8 pixel tiles for every width.  The kernel also instantiates as 64 x 4 (PX = 32); measured on the decoder's shapes
  // (B=16) that is 3-10 % slower -- its 66 x 6 halo re-reads 1.55x the tile from L2/HBM against 1.33x for 34 x 10, and the
  // halo traffic is what this kernel stalls on (64->64 @640x640: 3.17 ms vs 2.93 ms; without input DMA 2.76 ms).
  const bool wino64 = false;
  p.tiles_x = ceil_div(d->W, wino4 ? conv::wino4::TWX : wino ? (wino64 ? 64 : 32) : conv::TW);
  p.tiles_y = ceil_div(d->H, wino4 ? conv::wino4::THY : wino && !wino64 ? 8 : conv::TH);
  p.nchunks = d->Cin / kch;
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
  p.scale_shift = -1;
  for (int sh = 0; sh < 16; ++sh)
    if (d->scale == (1 << sh)) p.scale_shift = sh;
  if (d->out_mode == 3 && d->flow && p.scale_shift < 0) return C2M_ERR_UNSUPPORTED;   // pre-offset scales are powers of two
  // 32-bit byte offsets inside one sample's offset planes [n_off][H][W] and, separately, its mask planes (two buffer resources:
  // conv3x3_shared.h head_out) -- the same bound the DCNv2 kernel that reads them has (dcn_v2.hip use_nhwc)
  if (d->out_mode == 3 && (long long)std::max(d->n_off, cout_total - d->n_off) * d->H * d->W * 4 >= 0x7fffffffLL) return C2M_ERR_UNSUPPORTED;
  p.out_vec4 = out_vec4 ? 1 : 0;
  p.co_off = d->out_mode == 3 ? d->cout_offset : 0;
  p.cout_total = cout_total;
  p.out2 = d->out2; p.out2_row_pitch = d->out2_row_pitch; p.out2_plane_pitch = d->out2_plane_pitch;
  p.out2_img_pitch = d->out2_img_pitch;
  p.range_flag = d->range_flag;
  p.io_flags = d->io_flags;
  if (d->io_flags & C2M_IO_DWORD_STORES) {   // per-call reference store path of the split kernels' PixelShuffle / planar epilogues
    if (!splitk || (d->out_mode != 1 && d->out_mode != 2) || (d->io_flags & ~C2M_IO_DWORD_STORES)) return C2M_ERR_UNSUPPORTED;
  } else if (d->io_flags != 0) {
    if (d->algo != C2M_CONV_BF16 || d->out_mode != 0 || (d->io_flags & ~15)) return C2M_ERR_UNSUPPORTED;
    if ((d->io_flags & C2M_IO_SRC_BF16) && (d->nsrc != 1 || d->src[0].pix_pitch % 8 != 0 || d->src[0].row_pitch % 8 != 0 || d->src[0].img_pitch % 8 != 0))
      return C2M_ERR_UNSUPPORTED;   // 16-byte pieces of 8 bf16
    if (d->Cout % ((d->io_flags & C2M_IO_OUT_BF16) ? 8 : 4) != 0) return C2M_ERR_UNSUPPORTED;   // 16-byte stores
    if ((d->io_flags & C2M_IO_OUT_BF16) && (d->out_pix_pitch % 8 != 0 || d->out_row_pitch % 8 != 0 || d->out_img_pitch % 8 != 0)) return C2M_ERR_UNSUPPORTED;
  }
  if (d->out2 && (wino || d->out_mode != 0 || !out_vec4 || d->Cout % 8 != 0 || ((uintptr_t)d->out2 & 15) ||
                  d->out2_row_pitch % 4 != 0 || d->out2_plane_pitch % 4 != 0 || d->out2_img_pitch % 4 != 0))
    return C2M_ERR_UNSUPPORTED;

  const int MW = wino ? 64 : conv_mw(d->Cout);
  const long long ntile = (long long)p.tiles_x * p.tiles_y * p.B;
  if ((wino || splitk) && (d->out_mode == 0 || d->out_mode == 4) && !out_vec4) return C2M_ERR_UNSUPPORTED;
  if (ntile > 0x7fffffffLL) return C2M_ERR_INVALID_ARG;
  if (d->act == C2M_ACT_LEAKY_RELU && !(d->slope >= 0.0f && d->slope <= 1.0f)) return C2M_ERR_UNSUPPORTED;   // max(v, slope*v)
  for (int sidx = 0; sidx < d->nsrc; ++sidx) {   // 32-bit byte offsets inside one sample (buffer addressing)
    const long long ext = ((long long)(d->H - 1) * d->src[sidx].row_pitch + (long long)(d->W - 1) * d->src[sidx].pix_pitch +
                           d->src[sidx].C) * 4;
    if (ext >= 0x7fffffffLL || d->src[sidx].row_pitch < 0 || d->src[sidx].pix_pitch < 0) return C2M_ERR_UNSUPPORTED;
  }
  if (wino16) {
    ProfileScope prof(C2M_KERNEL_CONV3X3_SPLIT, as_stream(stream));
    return conv::launch_wino16(as_stream(stream), p, d->algo == C2M_CONV_WINO_F16X2_F43Y ? 4 : 2);
  }
  if (splitk) {
    ProfileScope prof(C2M_KERNEL_CONV3X3_SPLIT, as_stream(stream));
    return conv::launch_split(as_stream(stream), p, d->algo == C2M_CONV_BF16 ? 1 : (d->algo == C2M_CONV_SPLIT_F16X2 ? 2 : 3));
  }
  // tiles per workgroup: long streams amortise the set-up and the first DMA wait, but the launch is only as fast as its
  // last round of 512 resident workgroups (2 per CU; the F(4,3) kernel: 256, 1 per CU): take the tpw <= 10 with the fewest
  // "rounds x tiles" (ties: the longer stream), e.g. 51200 tiles -> 10 (10 full rounds), 12800 -> 5 (5 full rounds).
  // C2M_CONV_TPW overrides.
  static const int env_tpw = [] { const char* e = getenv("C2M_CONV_TPW"); return e ? atoi(e) : 0; }();
  const int ncb = ceil_div(d->Cout, MW);
  const long long resident = wino4 ? 256 : 512;
  long long tpw = 1, best = -1;
  for (long long t = 1; t <= 10; ++t) {
    const long long wgs = ((ntile + t - 1) / t) * ncb;
    const long long cost = ((wgs + resident - 1) / resident) * t;
    if (best < 0 || cost <= best) { best = cost; tpw = t; }
  }
  if (env_tpw > 0) tpw = env_tpw;
  p.tpw = (int)tpw;
  dim3 grid((unsigned)((ntile + tpw - 1) / tpw), ncb);
  hipStream_t st = as_stream(stream);
  ProfileScope prof(C2M_KERNEL_CONV3X3, st);
  const size_t ldsb = 2 * conv::IN_BYTES + 3 * (size_t)MW * 128 + 1024 + 256;   // halo x2, weight ring, DMA dummy, bias
  int rc = C2M_OK;
  auto go = [&](auto kern, unsigned long long& done) {
    if ((rc = ensure_dynamic_lds(reinterpret_cast<const void*>(kern), ldsb, done)) != C2M_OK) return;
    hipLaunchKernelGGL(kern, grid, dim3(256), ldsb, st, p);
  };
  static unsigned long long done[2][4] = {};
  if (wino4) {
    static unsigned long long done_w4 = 0;
    const size_t lds4 = 2 * conv::wino4::IN_BYTES + conv::wino4::NRING * conv::wino4::WUNIT + 1024 + 256;
    // C2M_CONV_ABL = 1..3: timing-only ablations of the F(4,3) kernel (see its template parameter); results are WRONG
    static const int abl = [] {
      const char* e = getenv("C2M_CONV_ABL");
      const int v = e ? atoi(e) : 0;
      if (v > 0) fprintf(stderr, "c2m: C2M_CONV_ABL=%d -- conv3x3 F(4,3) runs a timing-only ablation, its results are wrong\n", v);
      return v;
    }();
    static unsigned long long done_abl[4] = {};
    auto go4 = [&](auto kern, unsigned long long& dn) {
      if ((rc = ensure_dynamic_lds(reinterpret_cast<const void*>(kern), lds4, dn)) == C2M_OK)
        hipLaunchKernelGGL(kern, grid, dim3(256), lds4, st, p);
    };
    switch (abl) {
      case 1: go4(&conv::conv3x3_wino4_kernel<1>, done_abl[1]); break;
      case 2: go4(&conv::conv3x3_wino4_kernel<2>, done_abl[2]); break;
      case 3: go4(&conv::conv3x3_wino4_kernel<3>, done_abl[3]); break;
      default: go4(&conv::conv3x3_wino4_kernel<0>, done_w4); break;
    }
  } else if (wino) {
    static unsigned long long done_w[2][2] = {};
    const size_t ldsw = 2 * conv::wino::IN_BYTES + conv::wino::NRING * conv::wino::WUNIT + 1024 + 256;
    auto gow = [&](auto kern, unsigned long long& dn) {
      if ((rc = ensure_dynamic_lds(reinterpret_cast<const void*>(kern), ldsw, dn)) != C2M_OK) return;
      hipLaunchKernelGGL(kern, grid, dim3(256), ldsw, st, p);
    };
    static unsigned long long done_pool = 0;
    if (d->out_mode == 3) {
      gow(&conv::conv3x3_wino_kernel<16, 3>, done_w[1][1]);
    } else if (d->out_mode == 4) {
      gow(&conv::conv3x3_wino_kernel<16, 4>, done_pool);
    } else {
      gow(&conv::conv3x3_wino_kernel<16, 0>, done_w[1][0]);
    }
  } else if (MW == 64) {
    switch (d->out_mode) {
      case 0: go(&conv::conv3x3_kernel<2, 0>, done[1][0]); break;
      case 1: go(&conv::conv3x3_kernel<2, 1>, done[1][1]); break;
      case 2: go(&conv::conv3x3_kernel<2, 2>, done[1][2]); break;
      default: go(&conv::conv3x3_kernel<2, 3>, done[1][3]); break;
    }
  } else {
    switch (d->out_mode) {
      case 0: go(&conv::conv3x3_kernel<1, 0>, done[0][0]); break;
      case 1: go(&conv::conv3x3_kernel<1, 1>, done[0][1]); break;
      case 2: go(&conv::conv3x3_kernel<1, 2>, done[0][2]); break;
      default: go(&conv::conv3x3_kernel<1, 3>, done[0][3]); break;
    }
  }
  if (rc != C2M_OK) return rc;
  return check_launch();
}
