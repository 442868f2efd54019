// conv3x3_shared.h -- declarations shared by the 3x3 convolution kernels (conv3x3.hip: fp32-MFMA direct / Winograd kernels;
// conv3x3_split.hip: the split-bf16 kernels): launch parameters, buffer-descriptor helpers, the DCN offset/mask head stores.
#pragma once
#include "c2m_common.h"

namespace c2m {
namespace conv {

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
  int scale_shift;       // log2(scale): the head's pre-offset scales are powers of two (1, 2, 4 in C2-Matching)
  double* abs_sum;       // mode 3: C2M_ABS_SUM_SLOTS partial sums of |raw offset| or nullptr
  int out_vec4;          // mode 0: out / res pitches and bases are 16-byte aligned -> float4 stores
  int tpw;               // consecutive tiles per workgroup (>= 1)
  int co_off, cout_total;// mode 3: this launch computes head channels [co_off, co_off + Cout) of cout_total
  float* out2;           // mode 0 (direct kernel, float4 stores): second copy of the output, 8-channel group-major
  int out2_row_pitch;    //   out2[b*img + (co/8)*plane + y*row + x*8 + co%8]: the layout the DCNv2 kernel gathers 8-channel
  long long out2_plane_pitch, out2_img_pitch;   // groups from (c2m_dcn_v2_forward_nhwc_f32, input_grouped)
  int io_flags;          // bf16 flavour: C2M_IO_* (element types of src[0] / out / res1 / res2)
  int* range_flag;       // f16 x 2 flavour: set to 1 if an input activation lies outside the flavour's domain (|x| >= 65520)
};

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// raw buffer descriptor (stride 0, bounds-checked: a lane whose offset lies beyond num_records reads zeros -- that is how the
// zero padding of the halo tile is produced, without a select per lane)
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* base, unsigned bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (int)bytes, 0x00020000);
}
constexpr unsigned kOOB = 0x80000000u;   // voffset of a lane that must read zeros (>= any num_records used here)

// DCN offset/mask head, one group of 4 consecutive head channels of one pixel (used by both convolution kernels).
// Channels (co, co+1) = (dy, dx) of (group, tap) gt = co/2; pre-offset of tap k at scale s = 2^sh: P_k[y][x] =
// s * flow[(y - s*ki) >> sh][(x - s*kj) >> sh] (0 outside), channel order (y, x); mask = sigmoid.  `col` = channel inside
// this launch's slice, v = conv + bias.  Returns the |raw offset| contribution for the reference's offset-mean warning.
// Planar stores go through per-sample buffer resources: byte offset = channel * H*W*4 + pixel*4 as (per-lane VGPR part:
// pixel + the lane's channel quad) + (wave-uniform SGPR part: the rest of the channel) -- no 64-bit arithmetic per store.
struct HeadOut {
  __amdgpu_buffer_rsrc_t off, msk;   // this sample's offset planes [n_off][H][W] / mask planes [nm][H][W]
};
__device__ __forceinline__ HeadOut head_out(const Params& p, int b) {
  const unsigned HWb = (unsigned)(p.H * p.W) * 4u;
  const int nm = p.cout_total - p.n_off;
  HeadOut h;
  h.off = make_rsrc(p.out + (size_t)b * p.n_off * p.H * p.W, (unsigned)p.n_off * HWb);
  h.msk = make_rsrc(p.mask_out + (size_t)b * nm * p.H * p.W, (unsigned)nm * HWb);
  return h;
}
// col_u: wave-uniform part of the slice channel (multiple of 8), lane_q = 4 * hi: the lane's quad inside it
__device__ __forceinline__ float dcn_head_store(const Params& p, const HeadOut& ho, int b, int y, int x, int col_u, int lane_q,
                                                const f32x4& v) {
  const int HWb = p.H * p.W * 4;
  const int co_u = col_u + p.co_off;            // uniform part of the head channel
  const int co = co_u + lane_q;
  const int pixb = (y * p.W + x) * 4;
  float asum = 0.0f;
  const int vo = pixb + lane_q * HWb;
  if (co_u < p.n_off) {   // wave-uniform: n_off is a multiple of 8, so both channel quads of co_u lie on the same side
#pragma unroll
    for (int h2 = 0; h2 < 2; ++h2) {
      const int gt = (co >> 1) + h2, tap = gt % 9;
      const int ki = (tap * 11) >> 5, kj = tap - 3 * ki;   // tap / 3 for tap < 9
      float fy = 0.0f, fx = 0.0f;
      if (p.flow) {
        // branch-free: an out-of-range tap reads flow entry (0, 0) of the sample and is multiplied by 0
        const int ys = y - (ki << p.scale_shift), xs = x - (kj << p.scale_shift);
        const int yy = ys >> p.scale_shift, xx = xs >> p.scale_shift;
        const bool ok = (ys >= 0) & (xs >= 0) & (yy < p.fh) & (xx < p.fw);
        const float2 f = reinterpret_cast<const float2*>(p.flow)[(size_t)b * p.fh * p.fw + (ok ? yy * p.fw + xx : 0)];
        const float sc = ok ? (float)p.scale : 0.0f;
        fx = f.x * sc;
        fy = f.y * sc;
      }
      asum += fabsf(v[2 * h2]) + fabsf(v[2 * h2 + 1]);
      __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v[2 * h2] + fy), ho.off, vo, (co_u + 2 * h2) * HWb, 0);
      __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v[2 * h2 + 1] + fx), ho.off, vo,
                                            (co_u + 2 * h2 + 1) * HWb, 0);
    }
  } else {
#pragma unroll
    for (int e = 0; e < 4; ++e)
      if (col_u + lane_q + e < p.Cout)
        // sigmoid with the hardware exp2 / reciprocal (1 ulp each): the mask is a multiplier of sampled features in a
        // tolerance-based path; the correctly rounded expf + division cost ~20 instructions per value, at every call site
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, __builtin_amdgcn_rcpf(1.0f + __expf(-v[e]))), ho.msk, vo,
                                              (co_u - p.n_off + e) * HWb, 0);
  }
  return asum;
}

// ---- The same with 16-byte planar stores (W % 4 == 0, lanes j = 0..31 of a half-wave = 32 consecutive pixels of a row, x - j a
// multiple of 4): the four lanes of a quad hold 4 channels x 4 consecutive pixels; a 4 x 4 transpose inside the quad (two DPP
// exchange stages, registers only) hands lane 4q + i channel co + i of pixels 4q .. 4q + 3, which it stores as one 16-byte
// piece of that channel's plane: 16 store instructions per wave and 64-channel tile instead of 64.  The store's whole byte
// offset travels in the VGPR (soffset = 0): a 16-byte buffer store with a REGISTER soffset followed at once by a VALU write of
// its data registers stores the new value for some lanes on this chip, and hipcc inserts the wait state only for an immediate
// soffset.
template <int CTRL>
__device__ __forceinline__ float quad_dpp(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}
// lane 4q + i, element k  <->  lane 4q + k, element i
__device__ __forceinline__ f32x4 quad_transpose(const f32x4& v, int lane) {
  const bool odd = (lane & 1) != 0, up = (lane & 2) != 0;
  const float r0 = quad_dpp<0xB1>(v[0]), r1 = quad_dpp<0xB1>(v[1]), r2 = quad_dpp<0xB1>(v[2]), r3 = quad_dpp<0xB1>(v[3]);   // lane ^ 1
  const float a0 = odd ? r1 : v[0], a1 = odd ? v[1] : r0, a2 = odd ? r3 : v[2], a3 = odd ? v[3] : r2;
  const float t0 = quad_dpp<0x4E>(a0), t1 = quad_dpp<0x4E>(a1), t2 = quad_dpp<0x4E>(a2), t3 = quad_dpp<0x4E>(a3);           // lane ^ 2
  f32x4 o;
  o[0] = up ? t2 : a0; o[1] = up ? t3 : a1; o[2] = up ? a2 : t0; o[3] = up ? a3 : t1;
  return o;
}
// Pre-offsets of one row of 32 pixels: every tap reads flow[(y >> sh) - ki][(x >> sh) - kj], i.e. the row needs a window of
// 3 x ((32 >> sh) + 2) <= 102 flow entries -- two per lane (entries l and 64 + l), fetched ONCE per tile row (already times
// the scale, 0 outside the map or without a flow) and looked up with ds_bpermute, instead of one 8-byte global load per
// (4 channels, pixel): 64 vector-memory instructions per wave and 64-channel tile.
struct HeadFlowWin {
  float fx[2], fy[2];   // this lane's window entries
  int wx_row;           // (32 >> sh) + 2: entries per window row
};
// y = the row, x0 = first pixel of the tile row (multiple of 32), l = lane
__device__ __forceinline__ HeadFlowWin head_flow_window(const Params& p, int b, int y, int x0, int l) {
  HeadFlowWin w;
  w.wx_row = (32 >> p.scale_shift) + 2;
#pragma unroll
  for (int r = 0; r < 2; ++r) {
    w.fx[r] = 0.0f; w.fy[r] = 0.0f;
    if (p.flow != nullptr && (r == 0 || p.scale_shift == 0)) {   // (wave-uniform; scale 2 / 4: 54 / 30 entries)
      const int e = l + 64 * r;
      const int wy = (e >= w.wx_row ? 1 : 0) + (e >= 2 * w.wx_row ? 1 : 0), wx = e - wy * w.wx_row;
      const int Y = (y >> p.scale_shift) - 2 + wy, X = (x0 >> p.scale_shift) - 2 + wx;
      const bool ok = (e < 3 * w.wx_row) & (Y >= 0) & (X >= 0) & (Y < p.fh) & (X < p.fw);
      const float2 f = reinterpret_cast<const float2*>(p.flow)[(size_t)b * p.fh * p.fw + (ok ? Y * p.fw + X : 0)];
      const float sc = ok ? (float)p.scale : 0.0f;
      w.fx[r] = f.x * sc;
      w.fy[r] = f.y * sc;
    }
  }
  return w;
}
// The kernel's quad-store head mode (W % 4 == 0).  Call with the whole quad active (the pixels of a quad are valid together
// when W % 4 == 0).  j = lane & 31 = x - x0.
// Branch-free on purpose: every lane of the wave takes part (ds_bpermute returns 0 from a disabled lane, the DPP exchanges
// need the whole quad); `valid` = the lane's pixel exists and its channels lie inside the slice -- other lanes' stores go to
// an out-of-range offset, which the buffer's range check drops.
__device__ __forceinline__ float dcn_head_store_quad(const Params& p, const HeadOut& ho, int y, int x, int col_u, int lane_q,
                                                     const f32x4& v, int j, const HeadFlowWin& fw, bool valid) {
  typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
  const int HWb = p.H * p.W * 4;
  const int co_u = col_u + p.co_off;
  const int co = co_u + lane_q;
  float asum = 0.0f;
  f32x4 w;
  const bool is_off = co_u < p.n_off;   // wave-uniform (n_off % 8 == 0)
  if (is_off) {
#pragma unroll
    for (int h2 = 0; h2 < 2; ++h2) {
      const int gt = (co >> 1) + h2, tap = gt % 9;
      const int ki = (tap * 11) >> 5, kj = tap - 3 * ki;
      const int e = (2 - ki) * fw.wx_row + (j >> p.scale_shift) + 2 - kj;   // window entry ((y >> sh) - ki, (x >> sh) - kj)
      const int src = (e & 63) * 4;
      const float fx0 = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(src, __builtin_bit_cast(int, fw.fx[0])));
      const float fy0 = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(src, __builtin_bit_cast(int, fw.fy[0])));
      float fx = fx0, fy = fy0;
      if (p.scale_shift == 0) {   // (wave-uniform: only the scale-1 window has more than 64 entries)
        const float fx1 = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(src, __builtin_bit_cast(int, fw.fx[1])));
        const float fy1 = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(src, __builtin_bit_cast(int, fw.fy[1])));
        fx = e >= 64 ? fx1 : fx0;
        fy = e >= 64 ? fy1 : fy0;
      }
      asum += valid ? fabsf(v[2 * h2]) + fabsf(v[2 * h2 + 1]) : 0.0f;
      w[2 * h2] = v[2 * h2] + fy;
      w[2 * h2 + 1] = v[2 * h2 + 1] + fx;
    }
  } else {
#pragma unroll
    for (int e = 0; e < 4; ++e) w[e] = __builtin_amdgcn_rcpf(1.0f + __expf(-v[e]));
  }
  const f32x4 t = quad_transpose(w, j);
  const int ch = (is_off ? co : co - p.n_off) + (j & 3);          // this lane's plane after the transpose
  const unsigned vo = valid ? (unsigned)(ch * HWb + (y * p.W + (x & ~3)) * 4) : kOOB;
  if (is_off) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, t), ho.off, vo, 0, 0);
  else __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, t), ho.msk, vo, 0, 0);
  return asum;
}

// Two horizontally adjacent pixels (x even) at once, for the Winograd kernel whose lanes own pixel pairs: one 8-byte store
// per channel, so a wave writes whole 128-byte lines of every plane (dword stores at an 8-byte lane stride left every line
// half written per instruction -- the 5.7 GB the large head writes made that its bottleneck).
__device__ __forceinline__ float dcn_head_store2(const Params& p, const HeadOut& ho, int b, int y, int x, int col_u, int lane_q,
                                                 const f32x4& v0, const f32x4& v1) {
  typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
  const int HWb = p.H * p.W * 4;
  const int co_u = col_u + p.co_off;
  const int co = co_u + lane_q;
  const int vo = (y * p.W + x) * 4 + lane_q * HWb;
  float asum = 0.0f;
  auto st2 = [&](const __amdgpu_buffer_rsrc_t& rs, float a, float c, int so) __attribute__((always_inline)) {
    const u32x2 d = {__builtin_bit_cast(unsigned, a), __builtin_bit_cast(unsigned, c)};
    __builtin_amdgcn_raw_buffer_store_b64(d, rs, vo, so, 0);
  };
  if (co_u < p.n_off) {
#pragma unroll
    for (int h2 = 0; h2 < 2; ++h2) {
      const int gt = (co >> 1) + h2, tap = gt % 9;
      const int ki = (tap * 11) >> 5, kj = tap - 3 * ki;
      float fy[2] = {0.0f, 0.0f}, fx[2] = {0.0f, 0.0f};
      if (p.flow) {
        const int ys = y - (ki << p.scale_shift), yy = ys >> p.scale_shift;
#pragma unroll
        for (int px = 0; px < 2; ++px) {
          const int xs = x + px - (kj << p.scale_shift), xx = xs >> p.scale_shift;
          const bool ok = (ys >= 0) & (xs >= 0) & (yy < p.fh) & (xx < p.fw);
          const float2 f = reinterpret_cast<const float2*>(p.flow)[(size_t)b * p.fh * p.fw + (ok ? yy * p.fw + xx : 0)];
          const float sc = ok ? (float)p.scale : 0.0f;
          fx[px] = f.x * sc;
          fy[px] = f.y * sc;
        }
      }
      asum += (fabsf(v0[2 * h2]) + fabsf(v0[2 * h2 + 1])) + (fabsf(v1[2 * h2]) + fabsf(v1[2 * h2 + 1]));
      st2(ho.off, v0[2 * h2] + fy[0], v1[2 * h2] + fy[1], (co_u + 2 * h2) * HWb);
      st2(ho.off, v0[2 * h2 + 1] + fx[0], v1[2 * h2 + 1] + fx[1], (co_u + 2 * h2 + 1) * HWb);
    }
  } else {
#pragma unroll
    for (int e = 0; e < 4; ++e)
      if (col_u + lane_q + e < p.Cout)
        st2(ho.msk, __builtin_amdgcn_rcpf(1.0f + __expf(-v0[e])), __builtin_amdgcn_rcpf(1.0f + __expf(-v1[e])),
            (co_u - p.n_off + e) * HWb);
  }
  return asum;
}

}  // namespace conv
}  // namespace c2m
