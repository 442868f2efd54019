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
