// corr_filter.h -- interface between corr_argmax.hip (exact fp32-MFMA sweep, C-ABI entry point) and corr_filter.hip (the
// 16-bit-pipe pre-filter + exact re-score that replaces the sweep whenever its preconditions hold).
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

namespace c2m {
namespace corrf {

constexpr int WP = 28;       // ref patches per x-tile (as the exact kernel)
constexpr int KSLOT = 8;     // candidate slots per query handed from the filter to the re-score kernel
constexpr int SCAN_ITEMS = 8192;   // capacity of the re-score work list
constexpr int CMAX = 256;    // the workspace is sized for the widest feature map the filter serves

// byte offsets of the filter's scratch inside the correlation workspace (relative to `base`)
struct Ws {
  size_t qn, rn;        // float [B][Hq*Wq][C], [B][Hr*Wr][C]: channels-last copies of both maps (exact re-score)
  size_t qpl;           // f16   [B][2][Hq*Wq][C]: query pieces (A operands)
  size_t rimg;          // f16   [B][x-tiles][Hr][C*64]: ref pieces as ready-made LDS row images (B operands)
  size_t sb;            // float2 [B][Hrp*Wrp]: (scale, bias) of every candidate
  size_t band;          // float [B][Hqp*Wqp]: error band of every query
  size_t eq;            // uint8 [2][B][Hr*Wr]: pixel == left neighbour / == upper neighbour (all channels, bitwise)
  size_t cnt;           // int   [B][Hqp*Wqp]: candidates per query (-1: score every ref patch)
  size_t cand;          // int   [B][Hqp*Wqp][KSLOT]
  size_t flags;         // int   [8]: [0] != 0 -> the exact sweep must run (inputs outside the filter's domain); [1] work-list length
  size_t keys;          // u64   [B][Hqp*Wqp]: running best (order-preserving value bits, ~index) of queries on the work list
  size_t items;         // {int64 query, int lane, int pad} [SCAN_ITEMS]: whole-lane / whole-map re-score requests
  size_t maxcol;        // int   [B]: last ref patch column that holds a candidate (everything right of it repeats its left neighbour)
  size_t total;
};

inline size_t al256(size_t v) { return (v + 255) & ~size_t(255); }

// C: channels the maps have (CMAX = an upper bound for callers that do not know; 0 = the filter cannot run on these maps: no
// per-channel scratch at all).  The C-INDEPENDENT tables come first, so that their offsets -- what the diagnostics entry points
// report -- do not depend on C; the per-channel blocks follow.
inline Ws workspace(size_t base, int B, int Hq, int Wq, int Hr, int Wr, int C = CMAX) {
  Ws w;
  size_t o = al256(base);
  const size_t nq = (size_t)B * Hq * Wq, nr = (size_t)B * Hr * Wr;
  const size_t nqp = (size_t)B * (Hq > 2 ? Hq - 2 : 1) * (Wq > 2 ? Wq - 2 : 1);
  const size_t nrp = (size_t)B * (Hr > 2 ? Hr - 2 : 1) * (Wr > 2 ? Wr - 2 : 1);
  const size_t nxt = Wr > 2 ? (size_t)(Wr - 2 + WP - 1) / WP : 1;
  w.sb = o;    o = al256(o + nrp * 8);
  w.band = o;  o = al256(o + nqp * 4);
  w.eq = o;    o = al256(o + 2 * nr);
  w.cnt = o;   o = al256(o + nqp * 4);
  w.cand = o;  o = al256(o + nqp * 4 * KSLOT);
  w.flags = o; o = al256(o + 32);
  w.keys = o;  o = al256(o + nqp * 8);
  w.items = o; o = al256(o + (size_t)SCAN_ITEMS * 16);
  w.maxcol = o; o = al256(o + (size_t)B * 4);
  const size_t c = (size_t)(C < 0 ? 0 : C);
  w.qn = o;    o = al256(o + nq * c * 4);
  w.rn = o;    o = al256(o + nr * c * 4);
  w.qpl = o;   o = al256(o + nq * c * 4);
  w.rimg = o;  o = al256(o + (size_t)B * nxt * Hr * c * 128);
  w.total = o;
  return w;
}

// Is the filter defined for these shapes?  (16-bit candidate codes: patch row < 1024, x-tile < 64; 32-bit byte offsets)
inline bool shapes_ok(int B, int C, int Hq, int Wq, int Hr, int Wr) {
  if (!(C == 64 || C == 128 || C == 256)) return false;
  if (Hr - 2 > 1024 || (Wr - 2 + WP - 1) / WP > 64) return false;
  const size_t lim = (size_t)1 << 31;
  const size_t nxt = (size_t)(Wr - 2 + WP - 1) / WP;
  return (size_t)Hq * Wq * C * 4 < lim && (size_t)Hr * Wr * C * 4 < lim && nxt * Hr * C * 128 < lim;
}

// Enqueue: channels-last copies + pieces + candidate scales + bands, the filter sweep, the exact re-score.  On return (in
// stream order) max_idx / max_val hold the final result unless ws.flags[0] != 0, in which case the caller's exact sweep
// (which reads the flag on the device) overwrites them.  inv: 1/(|ref patch| + 1e-5) [B][Hrp*Wrp]; ss_in: per-pixel sums of
// squares of the query map [B][Hq*Wq]; qden: |query patch| + 1e-5 [B][Hqp*Wqp]; skip: the duplicate-row table -- UPDATED
// here: trailing x-tiles without a single candidate (every patch repeats its left / upper neighbour bit for bit: the band a
// zero-padded Ref leaves on the right) are marked (0, Hr) = "no row swept", for this sweep and for the caller's exact one.
int launch(hipStream_t st, const float* fin, const float* fref, int B, int C, int Hq, int Wq, int Hr, int Wr, const float* inv,
           const float* qden, int norm_input, int2* skip, char* wsbase, const Ws& ws, int64_t* max_idx, float* max_val);

}  // namespace corrf
}  // namespace c2m
