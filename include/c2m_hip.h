/*
 * c2m_hip.h -- C-ABI of libc2m_hip.so: the MI355X (gfx950) implementation of C2-Matching's restoration hot path.
 *
 * This is the drop-in boundary.  Plain pointers and sizes only: no torch / ATen types.  Every pointer is a DEVICE
 * pointer (HBM) unless stated otherwise; every launch is enqueued on `stream` (a hipStream_t passed as void*) and
 * returns without synchronising.  Every function returns a c2m_status (0 = ok); nothing prints.
 *
 * What each entry point replaces in the reference (paths relative to the upstream checkout):
 *
 *   c2m_feature_normalize_f32        F.normalize(feat.reshape(c,-1), dim=0)   mmsr/models/archs/corres_generation_arch.py:56-58
 *   c2m_feature_match_index_f32      sample_patches + feature_match_index     mmsr/models/archs/ref_map_util.py:4-23, 26-86
 *                                    (batched: the per-sample Python loop of corres_generation_arch.py:52 becomes grid.y)
 *   c2m_build_pre_offsets_f32        index_to_flow + 27 tensor_shift copies   mmsr/models/archs/corres_generation_arch.py:29-46, 69-109
 *                                                                             mmsr/models/archs/arch_util.py:291-315
 *   c2m_dcn_v2_forward_f32           dcn_v2_cuda_forward + im2col launcher    mmsr/models/archs/DCNv2/src/cuda/dcn_v2_cuda.cu:42-172
 *                                                                             mmsr/models/archs/DCNv2/src/cuda/dcn_v2_im2col_cuda.h:67-75
 *   c2m_dcn_v2_backward_f32          dcn_v2_cuda_backward + col2im launchers  mmsr/models/archs/DCNv2/src/cuda/dcn_v2_cuda.cu:206-335
 *                                                                             mmsr/models/archs/DCNv2/src/cuda/dcn_v2_im2col_cuda.h:77-95
 *   c2m_dcn_fuse_offsets_f32         chunk/cat/repeat/reorder/add/sigmoid     mmsr/models/archs/DCNv2/dcn_v2.py:229-245
 *
 * The reference's FFI for the DCN path is the pybind module `_ext` (DCNv2/src/vision.cpp:3-9).  The Python module
 * c2-matching_amd/_ext.py exports the same four names with the same argument lists and forwards to this ABI; see
 * INTEGRATION.md for the ctypes stub a maintainer of the reference would add.
 */
#ifndef C2M_HIP_H
#define C2M_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* c2m_stream_t; /* hipStream_t; NULL = the null stream */

typedef enum c2m_status {
  C2M_OK = 0,
  C2M_ERR_INVALID_ARG = 1,   /* null pointer, non-positive size, channels % groups != 0, ... */
  C2M_ERR_UNSUPPORTED = 2,   /* valid request that this build has no kernel for */
  C2M_ERR_WORKSPACE = 3,     /* workspace pointer null or too small */
  C2M_ERR_LAUNCH = 4,        /* hipLaunchKernel / hipGetLastError reported a failure */
  C2M_ERR_NO_DEVICE = 5      /* no gfx950 device visible to the HIP runtime */
} c2m_status;

int c2m_abi_version(void);                 /* bumped on any signature change; currently 3 (round 6: c2m_resblock3x3_*) */
const char* c2m_status_string(int status); /* static string, never NULL */
const char* c2m_last_hip_error(void);      /* hipGetErrorString of the last failing HIP call on this thread */
int c2m_device_arch(char* buf, int buflen);/* gcnArchName of the current device, e.g. "gfx950:sramecc+:xnack-" */

/*
 * Kernel timing for bench.py's roofline line: when enabled, every API call brackets its DOMINANT kernel (the MFMA
 * correlation sweep, the DCNv2 implicit-GEMM forward, ...) with hipEventRecord on the caller's stream.
 * c2m_profile_collect synchronises the recorded events, writes the elapsed milliseconds (oldest first) and the
 * kernel ids (C2M_KERNEL_*) and clears the list.  Off by default; costs two event records per call when on.
 */
enum { C2M_KERNEL_CORR_MFMA = 1, C2M_KERNEL_CORR_GENERIC = 2, C2M_KERNEL_DCN_FWD = 3, C2M_KERNEL_DCN_BWD_DATA = 4,
       C2M_KERNEL_DCN_BWD_WEIGHT = 5, C2M_KERNEL_CONV3X3 = 6 /* fp32-MFMA direct / Winograd kernels */,
       C2M_KERNEL_CONV3X3_SPLIT = 7 /* split kernel (C2M_CONV_SPLIT_F16X2 / C2M_CONV_SPLIT_BF16X3 / C2M_CONV_BF16) */,
       C2M_KERNEL_CONV3X3_WGRAD = 8,
       C2M_KERNEL_CORR_FILTER = 9 /* f16-pipe pre-filter sweep of the correlation (corr_filter.hip) */,
       C2M_KERNEL_CORR_RESOLVE = 10 /* exact fp32 re-score of the filter's candidates */ };
enum { C2M_ACT_NONE = 0, C2M_ACT_RELU = 1, C2M_ACT_LEAKY_RELU = 2 };   /* fused activations of the decoder-path entry points */
int c2m_profile_enable(int on);
int c2m_profile_collect(float* ms, int* kernel_id, int capacity, int* count);

/* ---------------------------------------------------------------------------------------------------------------
 * Correlation / index search
 * ------------------------------------------------------------------------------------------------------------- */

/* x, out: [B][C][HW] fp32.  out[b,c,p] = x[b,c,p] / max(||x[b,:,p]||_2, 1e-12).  In-place allowed. */
int c2m_feature_normalize_f32(c2m_stream_t stream, const float* x, int B, int C, int HW, float* out);
/* The same, and ss_out [B][HW] (or NULL) = sum over c of out[b,c,p]^2 as ONE fmaf chain, c ascending, over the stored floats: the
 * per-pixel sums of squares c2m_feature_match_index_pre_f32 needs for its patch norms (ref_map_util.py:63, :80), formed while the
 * values are in registers instead of by another pass over the map (round 6: two of the correlation's preparation launches). */
int c2m_feature_normalize_ss_f32(c2m_stream_t stream, const float* x, int B, int C, int HW, float* out, float* ss_out);

/* Bytes of scratch c2m_feature_match_index_f32 needs for these shapes (patch norms of both maps, duplicate-row table, and
 * the pre-filter's channels-last copies / f16 pieces / candidate lists, sized for C = 256). */
size_t c2m_feature_match_workspace_bytes(int B, int Hq, int Wq, int Hr, int Wr);
/* The same for maps of C channels: the pre-filter's per-channel scratch is sized by C (none where the filter has no kernel:
 * C other than 64 / 128 / 256) instead of the 256-channel upper bound -- a quarter of the bytes at C = 64, a few MB where only
 * the generic kernel can run.  c2m_feature_match_index_f32 accepts a workspace of either size. */
size_t c2m_feature_match_workspace_bytes_c(int B, int C, int Hq, int Wq, int Hr, int Wr);

/*
 * Diagnostics for the MFMA kernel's duplicate-row elimination: after c2m_feature_match_index_f32 the workspace holds,
 * at *byte_offset, int32 pairs [B][*x_tiles][2] = (from, to): ref pixel rows [from, to) of that (sample, x-tile) were
 * not swept because they repeat rows from-3 .. from-1 bit for bit (their patch rows can never win the lowest-index
 * tie rule of ref_map_util.py:74).  from == to: every row swept.  $C2M_CORR_DEDUP=0 disables the elimination.
 */
int c2m_feature_match_skip_table(int B, int Hq, int Wq, int Hr, int Wr, size_t* byte_offset, int* x_tiles);

/*
 * The MFMA path of c2m_feature_match_index_f32 has two implementations with identical results (indices AND values):
 *   exact sweep   every (query, ref patch) score on the fp32 matrix pipe (corr_argmax.hip);
 *   pre-filter    the same sweep on the f16 matrix pipe (two pieces per operand, three products: 3/16 of the matrix time)
 *                 keeps, per query, every candidate whose filter score lies within a RIGOROUS error band of the best one
 *                 (8.4e-5 |query patch| + 1e-6); the listed candidates are then re-scored with the oracle's exact fp32
 *                 chain and the first maximum taken (corr_filter.hip).  Needs is_norm, |x| < 3.99 everywhere and ref patch
 *                 norms >= 0.5 (channel-normalised features satisfy all three); anything else falls back to the exact
 *                 sweep on the device, without a host round trip.
 * mode: 1 pre-filter (default), 0 exact sweep only, -1 follow $C2M_CORR_FILTER (unset = 1).  Per CALLING THREAD (thread_local: a
 * DataParallel replica thread or another stream's host thread never sees a test's setting); measurement and tests.
 */
int c2m_feature_match_set_filter(int mode);

/*
 * The DCN offset/mask head epilogue (C2M_OUT_DCN_HEAD) of the split kernels has two store paths with identical results:
 *   1 (default, maps with W % 4 == 0)  16-byte planar stores after a 4 x 4 register transpose inside lane quads, pre-offsets
 *                                      from a per-row flow window held in registers (ds_bpermute look-ups);
 *   0                                  dword planar stores, one 8-byte flow load per (4 channels, pixel).
 * mode: 1 / 0, -1 follow $C2M_HEAD_QUAD (unset = 1).  Per CALLING THREAD (thread_local); measurement and tests.
 */
int c2m_conv3x3_set_head_stores(int mode);

/*
 * Diagnostics of the pre-filter: after c2m_feature_match_index_f32 took that path the workspace holds int32 [B][Hqp*Wqp]
 * candidate counts at *cnt_offset (-1 = every ref patch was re-scored), int32 [B][Hqp*Wqp][*slots] candidates at
 * *cand_offset (>= 0x40000000: "re-score the whole candidate set of lane (entry & 31)") and int32 flags at *flags_offset
 * ([0] != 0: the inputs were outside the filter's domain and the exact sweep produced the result).
 */
int c2m_feature_match_filter_tables(int B, int Hq, int Wq, int Hr, int Wr, size_t* cnt_offset, size_t* cand_offset,
                                    size_t* flags_offset, int* slots);

/*
 * feat_in [B][C][Hq][Wq], feat_ref [B][C][Hr][Wr] fp32 contiguous; for every sample b independently
 *   max_idx[b][qy][qx] = argmax_n corr(q, n),  n = ry * Wrp + rx row-major over the ref patch grid, lowest n on ties
 *   max_val[b][qy][qx] = that maximum (divided by the query patch norm + 1e-5 when norm_input)
 * with Hqp = (Hq - patch)/in_stride + 1 etc.  max_idx is int64 (torch.max indices), max_val fp32, both [B][Hqp][Wqp].
 * patch == 3, both strides == 1 and C in {64,128,256} run the MFMA sliding-window kernel; everything else runs
 * the generic kernel (same arithmetic, bit-identical results, much slower).  `force_generic` != 0 forces the latter.
 */
int c2m_feature_match_index_f32(c2m_stream_t stream, const float* feat_in, const float* feat_ref, int B, int C,
                                int Hq, int Wq, int Hr, int Wr, int patch, int in_stride, int ref_stride,
                                int is_norm, int norm_input, int force_generic, int64_t* max_idx, float* max_val,
                                void* workspace, size_t workspace_bytes);
/* The same with the per-pixel sums of squares of one or both maps handed in (ss_in_pre [B][Hq*Wq], ss_ref_pre [B][Hr*Wr]: what
 * c2m_feature_normalize_ss_f32 wrote next to the normalised maps; NULL = computed here).  They must be the canonical chain over
 * THESE maps' values (fmaf, c ascending) -- the patch norms, and with them the index map's bit-exactness, depend on it. */
int c2m_feature_match_index_pre_f32(c2m_stream_t stream, const float* feat_in, const float* feat_ref, int B, int C,
                                    int Hq, int Wq, int Hr, int Wr, int patch, int in_stride, int ref_stride,
                                    int is_norm, int norm_input, int force_generic, int64_t* max_idx, float* max_val,
                                    void* workspace, size_t workspace_bytes, const float* ss_in_pre, const float* ss_ref_pre);

/*
 * max_idx [B][h-2][w-2] int64 (h, w = feature-map size of BOTH maps)  ->
 *   off3 [B][9][h][w][2], off2 [B][9][2h][2w][2], off1 [B][9][4h][4w][2] fp32, last dim (x, y); any may be NULL.
 */
int c2m_build_pre_offsets_f32(c2m_stream_t stream, const int64_t* max_idx, int B, int h, int w, float* off3,
                              float* off2, float* off1);

/* ---------------------------------------------------------------------------------------------------------------
 * DCNv2 (modulated deformable convolution), fp32, NCHW contiguous
 *   input [B][C][H][W]  weight [Co][C][kh][kw]  bias [Co]
 *   offset [B][dg*2*kh*kw][Ho][Wo] (channel (g*kh*kw+k)*2 = dy, +1 = dx)   mask [B][dg*kh*kw][Ho][Wo]
 *   output / grad_output [B][Co][Ho][Wo],  Ho = (H + 2*ph - (dh*(kh-1)+1))/sh + 1
 * ------------------------------------------------------------------------------------------------------------- */
size_t c2m_dcn_v2_forward_workspace_bytes(int B, int C, int H, int W, int Co, int kh, int kw, int dg);
int c2m_dcn_v2_forward_f32(c2m_stream_t stream, const float* input, const float* weight, const float* bias,
                           const float* offset, const float* mask, int B, int C, int H, int W, int Co, int kh, int kw,
                           int sh, int sw, int ph, int pw, int dh, int dw, int dg, float* output, void* workspace,
                           size_t workspace_bytes);

/* Same operator with the implicit GEMM on bf16 MFMA (v_mfma_f32_32x32x16_bf16, fp32 accumulation): weights and the
 * blended column values are rounded to bf16 (RNE); tensors, sampling positions, bilinear blend and bias stay float32.
 * For callers that asked for reduced precision (the Python operator uses it under bf16 autocast, BASELINE config 5);
 * geometries without a bf16 kernel (groups of fewer than 16 channels that cannot be paired) compute in fp32.
 * Same workspace as c2m_dcn_v2_forward_f32. */
int c2m_dcn_v2_forward_bf16mma_f32(c2m_stream_t stream, const float* input, const float* weight, const float* bias,
                                   const float* offset, const float* mask, int B, int C, int H, int W, int Co, int kh,
                                   int kw, int sh, int sw, int ph, int pw, int dh, int dw, int dg, float* output,
                                   void* workspace, size_t workspace_bytes);

/*
 * Fused decoder path (inference): the same forward operator with the per-call preparation hoisted out of it.
 *   c2m_nchw_to_nhwc_bordered_f32   input [B][C][H][W] -> zero-bordered channels-last copy [B][H+3][W+3][C] (1 pixel
 *                                   top/left, 2 bottom/right).  It is what the forward kernel gathers from; pixel (0,0)
 *                                   sits at out + ((W+3) + 1) * C, so the copy doubles as a c2m_conv_src (pix_pitch C,
 *                                   row_pitch (W+3)*C, img_pitch (H+3)*(W+3)*C) for the offset convolutions.
 *   c2m_dcn_v2_relayout_f32         weight [Co][C][kh][kw] -> the kernel's A-operand layout (cache it while the weights
 *                                   do not change); c2m_dcn_v2_relayout_bytes == 0: geometry not on the channels-last path
 *                                   (8/16/32 channels per group and an even group count are).
 *   c2m_dcn_v2_forward_nhwc_f32     dcn_v2_cuda_forward (dcn_v2_cuda.cu:42-172) from those two; output planar
 *                                   [B][Co][Ho][Wo] (out_nhwc = 0) or channels-last with the given pitches (in floats)
 *                                   and an optional fused activation (C2M_ACT_*: the lrelu that follows every DynAgg,
 *                                   ref_restoration_arch.py:152-154).  input_grouped = 1 (C / dg == 8 only):
 *                                   input_bordered is group-major [B][dg][H+3][W+3][8] instead -- a 32-byte sample run
 *                                   then shares its 128-byte line with the neighbouring positions of the same group
 *                                   instead of with three other groups (c2m_conv3x3_desc.out2 writes this layout).
 */
int c2m_nchw_to_nhwc_bordered_f32(c2m_stream_t stream, const float* input, int B, int C, int H, int W, float* out);
size_t c2m_dcn_v2_relayout_bytes(int C, int Co, int kh, int kw, int dg);
int c2m_dcn_v2_relayout_f32(c2m_stream_t stream, const float* weight, int C, int Co, int kh, int kw, int dg, float* wt);
int c2m_dcn_v2_forward_nhwc_f32(c2m_stream_t stream, const float* input_bordered, const float* wt, const float* bias,
                                const float* offset, const float* mask, int B, int C, int H, int W, int Co, int kh, int kw,
                                int sh, int sw, int ph, int pw, int dh, int dw, int dg, float* output, int out_nhwc,
                                int out_pix_pitch, int out_row_pitch, long long out_img_pitch, int act, float slope,
                                int input_grouped);

/*
 * The same forward with the implicit GEMM on the F16 matrix pipe, fp32 result (the arithmetic of C2M_CONV_SPLIT_F16X2: the
 * blended column value c = x0 + 2^-11 x1' in two round-to-nearest f16 pieces, per-tensor-scaled weights S w = wA + w1, three
 * products per k step, one fp32 accumulator, times 1/S): 3/32 of the fp32 pipe's matrix time, error of the class of the fp32
 * accumulation chain.  Geometries with >= 16 channels per (virtual) group, i.e. every DynAgg layer of the restoration network
 * (c2m_dcn_v2_relayout_f16x2_bytes == 0 otherwise).  Domain: |mask * bilinear sample| < 65520 -- beyond it outputs are not
 * finite and `range_flag` (device int, may be NULL; never cleared here) is set to 1: the caller recomputes with
 * c2m_dcn_v2_forward_nhwc_f32, as for the convolutions (c2m_conv3x3_desc.range_flag).
 */
size_t c2m_dcn_v2_relayout_f16x2_bytes(int C, int Co, int kh, int kw, int dg);
int c2m_dcn_v2_relayout_f16x2(c2m_stream_t stream, const float* weight, int C, int Co, int kh, int kw, int dg, void* wt);
int c2m_dcn_v2_forward_nhwc_f16x2(c2m_stream_t stream, const float* input_bordered, const void* wt, const float* bias,
                                  const float* offset, const float* mask, int B, int C, int H, int W, int Co, int kh, int kw,
                                  int sh, int sw, int ph, int pw, int dh, int dw, int dg, float* output, int out_nhwc,
                                  int out_pix_pitch, int out_row_pitch, long long out_img_pitch, int act, float slope,
                                  int input_grouped, int* range_flag);

size_t c2m_dcn_v2_backward_workspace_bytes(int B, int C, int H, int W, int Co, int kh, int kw, int sh, int sw, int ph,
                                           int pw, int dh, int dw, int dg);
/* All gradients are OVERWRITTEN (the reference starts them from zeros, dcn_v2_cuda.cu:251-255).  grad_input may be NULL:
 * the col2im scatter is then skipped (C2-Matching warps frozen VGG features of the Ref image: that gradient is never used). */
int c2m_dcn_v2_backward_f32(c2m_stream_t stream, const float* input, const float* weight, const float* bias,
                            const float* offset, const float* mask, const float* grad_output, int B, int C, int H,
                            int W, int Co, int kh, int kw, int sh, int sw, int ph, int pw, int dh, int dw, int dg,
                            float* grad_input, float* grad_offset, float* grad_mask, float* grad_weight,
                            float* grad_bias, void* workspace, size_t workspace_bytes);

/*
 * conv_out [B][3*dg*K][H][W] (raw output of conv_offset_mask, K = kh*kw), pre_offset [B][K][H][W][2] (x, y) or NULL ->
 *   offset [B][2*dg*K][H][W] = cat(o1, o2) + interleaved (y, x) pre-offset repeated over the dg groups
 *   mask   [B][dg*K][H][W]   = sigmoid(third chunk)
 *   abs_sum (device double[C2M_ABS_SUM_SLOTS], may be NULL): the slots together accumulate sum |cat(o1,o2)| for the
 *   reference's "offset mean > 100" warning without a host sync in the hot path (caller zeroes them and adds them up;
 *   many slots so that ~10^5 workgroups do not serialise on one atomic).
 */
#define C2M_ABS_SUM_SLOTS 256
int c2m_dcn_fuse_offsets_f32(c2m_stream_t stream, const float* conv_out, const float* pre_offset, int B, int dg, int K,
                             int H, int W, float* offset, float* mask, double* abs_sum);

/* ---------------------------------------------------------------------------------------------------------------
 * 3x3 / stride 1 / pad 1 convolution, fp32, channels-last, fused epilogue
 *
 *   out = act( conv3x3( cat(src[0], src[1]) ) + bias ) + res1 + res2
 *
 * replaces, per call, one nn.Conv2d of the decoder (mmsr/models/archs/ref_restoration_arch.py:140-187,
 * arch_util.py:80-136) TOGETHER with the elementwise ops the reference runs around it: torch.cat of the two inputs
 * (:147,:151 ...), the bias add, ReLU / LeakyReLU, the residual adds, nn.PixelShuffle(2) (out_mode 1), and -- for the
 * DCN offset/mask head (out_mode 3) -- chunk/cat/sigmoid of mmsr/models/archs/DCNv2/dcn_v2.py:229-245 plus the pre-offset
 * construction of corres_generation_arch.py:29-46,69-109 (index_to_flow, 9 tensor_shift copies, x s, repeat over groups,
 * (x,y)->(y,x)), synthesised from the flow map of the arg-max indices instead of being read from [B,9,H,W,2] tensors.
 *
 * Sources are channels-last with explicit pitches (in floats), so a torch channels_last tensor, a channel slice of one,
 * or the zero-bordered copy made by c2m_nchw_to_nhwc_bordered_f32 can be passed without a copy.  Every source must bring
 * a multiple of 32 channels; pitches and base pointers must be multiples of 4 floats (16-byte LDS-DMA pieces).
 * ------------------------------------------------------------------------------------------------------------- */
typedef struct c2m_conv_src {
  const float* ptr;      /* pixel (0,0), channel 0, sample 0 */
  int C;                 /* channels taken from this source */
  int pix_pitch;         /* floats between horizontally adjacent pixels (= total channels of the tensor) */
  int row_pitch;         /* floats between rows */
  long long img_pitch;   /* floats between samples */
} c2m_conv_src;

enum { C2M_CONV_DIRECT = 0, C2M_CONV_WINOGRAD_F23X = 1, C2M_CONV_WINOGRAD_F43X = 2, C2M_CONV_SPLIT_BF16X3 = 3, C2M_CONV_BF16 = 4,
       C2M_CONV_SPLIT_F16X2 = 5, C2M_CONV_WINO_F16X2_F43Y = 6, C2M_CONV_WINO_F16X2_F23Y = 7 };
enum { C2M_OUT_NHWC = 0, C2M_OUT_NHWC_PIXEL_SHUFFLE2 = 1, C2M_OUT_NCHW = 2, C2M_OUT_DCN_HEAD = 3, C2M_OUT_NHWC_MAXPOOL2 = 4 };

typedef struct c2m_conv3x3_desc {
  int B, H, W, Cin, Cout;
  int nsrc;                /* 1 or 2; Cin = sum of src[].C */
  c2m_conv_src src[2];
  const float* wr;         /* weights re-laid-out by c2m_conv3x3_relayout_f32 (cache it while the weights do not change) */
  const float* bias;       /* [Cout] or NULL */
  int act;                 /* C2M_ACT_* (applied before the residual adds; ignored by C2M_OUT_DCN_HEAD) */
  float slope;             /* LeakyReLU negative slope */
  int out_mode;            /* C2M_OUT_* */
  float* out;              /* NHWC: pitches below.  PIXEL_SHUFFLE2: [B][2H][2W][Cout/4] with the pitches below.
                              NHWC_MAXPOOL2: [B][H/2][W/2][Cout] = MaxPool2d(2,2) of the activated output (C2M_CONV_WINOGRAD_F23X
                              only, even H; the conv -> ReLU -> pool steps of the VGG towers, vgg_arch.py:107-123).
                              NCHW: [B][Cout][H][W].  DCN_HEAD: offset [B][n_off][H][W] (raw + pre-offset, (dy,dx) pairs) */
  int out_pix_pitch, out_row_pitch;
  long long out_img_pitch;
  const float* res1;       /* NHWC only: tensors with out's geometry added after the activation, or NULL */
  const float* res2;
  float* mask_out;         /* DCN_HEAD: [B][Cout - n_off][H][W] = sigmoid(logits) */
  const float* flow;       /* DCN_HEAD: [B][fh][fw][2] (x,y) from c2m_index_to_flow_f32, or NULL for "no pre-offset" */
  int fh, fw;              /* = h-2, w-2 of the matched feature maps */
  int scale;               /* H / h: 1, 2 or 4 */
  int n_off;               /* offset channels = 2 * deformable_groups * 9 */
  double* abs_sum;         /* DCN_HEAD: C2M_ABS_SUM_SLOTS partial sums of |raw offset| (caller zeroes), or NULL */
  int algo;                /* C2M_CONV_DIRECT (0) or C2M_CONV_WINOGRAD_F23X: Winograd F(2,3) along x -- 1.5x fewer matrix
                              instructions; NHWC mode, Cout % 64 == 0, even W, channels % 16 == 0; `wr` must then come from
                              c2m_conv3x3_relayout_wino_f32.  fp32; differs from the direct kernel by transform rounding.
                              C2M_CONV_WINOGRAD_F43X (2): Winograd F(4,3) along x -- 2x fewer matrix instructions; NHWC mode only,
                              Cout % 64 == 0, W % 64 == 0, channels % 16 == 0, `wr` from c2m_conv3x3_relayout_wino4_f32; its
                              transforms put the result ~4x further from the exact value than the direct kernel (still
                              ~1e-6 relative): meant for the decoder, not for the extractors that feed the index search.
                              C2M_CONV_SPLIT_BF16X3 (3): fp32 result on the BF16 matrix pipe -- every operand is split exactly into
                              three bf16 pieces and six piece products are accumulated in fp32 (csrc/conv3x3_split.hip): 6/16 of
                              the fp32-MFMA time, error below an fp32 fmaf chain; any out_mode (C2M_OUT_NHWC_MAXPOOL2 included),
                              any H, W, channels % 16 == 0, no out2; `wr` from c2m_conv3x3_relayout_split_f32(pieces = 3).
                              C2M_CONV_BF16 (4): the same kernel with one round-to-nearest bf16 piece per operand -- a plain bf16
                              convolution with fp32 accumulation (bf16 inference, BASELINE configs[4]); pieces = 1.
                              C2M_CONV_SPLIT_F16X2 (5): fp32 result on the F16 matrix pipe with THREE products: activations split
                              x = x0 + 2^-11 x1' (round-to-nearest halves: |x - x0 - 2^-11 x1'| <= max(2^-22 |x|, 2^-36)), weights scaled
                              by the per-tensor power of two S that puts max |w| into [2^14, 2^15) and split the same way;
                              S w.x = wA.x0 + w1.x0 + (2^-11 wA).x1' (dropped: <= 2^-22 |w||x|), one accumulator, times 1/S in the
                              epilogue.  Error of the same class as the fp32 accumulation chain itself (tests/test_conv_gpu.py
                              holds it to the tolerance of the other fp32 kernels).  Domain: |x| < 65520 -- larger inputs give NaN
                              (never a silently wrong number); for such data use C2M_CONV_SPLIT_BF16X3 (full fp32 range).
                              pieces = 2 (the image carries 1/S behind it).
                              C2M_CONV_WINO_F16X2_F43Y (6) / _F23Y (7): the same f16 x 2 arithmetic behind a Winograd F(4,3) / F(2,3)
                              transform ALONG Y (csrc/conv3x3_wino16.hip): 1/2 (2/3) of the matrix instructions of (5).  C2M_OUT_NHWC
                              only, Cout % 64 == 0, channels % 16 == 0, any H, W; `wr` from c2m_conv3x3_relayout_split_f32 with
                              pieces = 2 | R << 4 (0x42 / 0x22).  The input transform runs in fp32 before the split, the weight
                              transform in float64 before theirs, the output transform in fp32: error against float64 below (5)'s
                              for F(2,3) and at the exact-fp32-MFMA kernel's level for F(4,3) (tests/test_conv_gpu.py holds both to
                              (5)'s tolerances).  Domain |x| < 26200 (F43) / 32760 (F23), reported through range_flag as for (5) */
  int cout_offset;         /* DCN_HEAD: this call computes head channels [cout_offset, cout_offset + Cout) of cout_total */
  int cout_total;          /* (weights / bias passed are those rows only); 0 = the whole head in one call.  Lets a 216-channel
                              head run as 192 channels on 64-wide tiles + 24 on a 32-wide tile instead of 256 padded ones */
  float* out2;             /* NHWC + C2M_CONV_DIRECT only, or NULL: a second copy of the output in the 8-channel group-major
                              layout out2[b*img + (co/8)*plane + y*row + x*8 + co%8] (pitches in floats, multiples of 4) --
                              what c2m_dcn_v2_forward_nhwc_f32(input_grouped = 1) gathers from.  Cout % 8 == 0 */
  int out2_row_pitch;
  long long out2_plane_pitch, out2_img_pitch;
  int* range_flag;         /* C2M_CONV_SPLIT_F16X2 only, or NULL: device int the kernel sets to 1 when an input activation lies
                              outside the flavour's domain (|x| >= 65520: the output then holds NaN).  Never cleared by the
                              kernel: the caller zeroes it, runs any number of convolutions and reads it once -- if set, the
                              results are to be recomputed with C2M_CONV_SPLIT_BF16X3 (what c2m_amd.ops.f16_range_guard and
                              the fused module forwards do, so that a drop-in never returns NaN where nn.Conv2d returns a
                              number: arch_util.py:80-136) */
  int io_flags;            /* C2M_CONV_BF16 + C2M_OUT_NHWC only, else 0: element types of the tensors of a bf16 (autocast)
                              forward, so that activations travel between the fused convolutions as bf16 (half the HBM bytes;
                              BASELINE configs[4]).  Bit 0: src[0] holds bf16 (nsrc = 1; pitches in ELEMENTS, multiples of 8,
                              the tile goes HBM -> LDS by DMA without passing registers); bit 1: `out` holds bf16; bit 2 / 3:
                              res1 / res2 hold bf16.  Pointers are passed as float* regardless.  Accumulation, bias,
                              activation and the residual adds stay fp32; one rounding at the store */
} c2m_conv3x3_desc;
#define C2M_IO_SRC_BF16 1
#define C2M_IO_OUT_BF16 2
#define C2M_IO_RES1_BF16 4
#define C2M_IO_RES2_BF16 8
#define C2M_IO_DWORD_STORES 16   /* split algorithms, C2M_OUT_NHWC_PIXEL_SHUFFLE2 / C2M_OUT_NCHW only (alone): take the epilogue's
                                    one-dword-per-lane stores instead of the 16-byte lane-swapped / quad-transposed ones -- the
                                    same bits through the other instruction sequence; per call (tests, measurement) */

size_t c2m_conv3x3_relayout_bytes(int Cin, int Cout);   /* 0 if the geometry is unsupported (Cin % 32 != 0) */
int c2m_conv3x3_relayout_f32(c2m_stream_t stream, const float* weight /* [Cout][Cin][3][3] */, int Cin, int Cout, float* wr);
size_t c2m_conv3x3_relayout_wino_bytes(int Cin, int Cout);   /* 0 if unsupported (Cin % 16, Cout % 64) */
int c2m_conv3x3_relayout_wino_f32(c2m_stream_t stream, const float* weight, int Cin, int Cout, float* wr);
size_t c2m_conv3x3_relayout_wino4_bytes(int Cin, int Cout);  /* 0 if unsupported (Cin % 16, Cout % 64) */
int c2m_conv3x3_relayout_wino4_f32(c2m_stream_t stream, const float* weight, int Cin, int Cout, float* wr);
size_t c2m_conv3x3_relayout_split_bytes(int Cin, int Cout, int pieces);   /* pieces 3, 1 or 2 (f16 x 2), or 2 | R << 4 (0x42, 0x22: images
                                                                             of C2M_CONV_WINO_F16X2_F43Y / _F23Y, Cout % 64 == 0); 0 if unsupported (Cin % 16) */
int c2m_conv3x3_relayout_split_f32(c2m_stream_t stream, const float* weight, int Cin, int Cout, int pieces, void* wr);
/* Many images per call (the per-forward refresh of a module's cached weight images from the parameters' current contents:
 * ~160 tensors of a RestorationNet in two launches instead of three per tensor).  `jobs` = DEVICE array [njobs][8] of int64:
 * {weight pointer, image pointer (sized by c2m_conv3x3_relayout_split_bytes), Cin, Cout, pieces | MT << 8 | dgrad << 16
 * (MT = 1 for Cout <= 32 else 2), image bytes / 2 (without the f16 x 2 tail), first block (running sum of
 * ceil(image elements / 256) over the jobs before), 0}; nblocks = that sum over all jobs; any_f16 != 0 if some job has
 * pieces = 2 (then one more launch reduces max |w| of those tensors for their scales). */
int c2m_conv3x3_relayout_split_multi(c2m_stream_t stream, const long long* jobs, int njobs, long long nblocks, int any_f16);
int c2m_conv3x3_nhwc_f32(c2m_stream_t stream, const c2m_conv3x3_desc* desc);

/*
 * Backward of the same convolution (stage-3 training, ref_restoration_model.py:192-269; the reference leaves it to cuDNN).
 *   data gradient:   dX = conv3x3(dY, W') with W'[ci][co][dy][dx] = W[co][ci][2-dy][2-dx] -- run c2m_conv3x3_nhwc_f32 with
 *                    Cin' = Cout, Cout' = Cin and `wr` from c2m_conv3x3_relayout_split_dgrad_f32 (same image format as
 *                    c2m_conv3x3_relayout_split_f32(Cout, Cin, pieces); size c2m_conv3x3_relayout_split_bytes(Cout, Cin, pieces));
 *   weight gradient: dW[co][ci][dy][dx] = sum_{b,y,x} dY[b][y][x][co] * X[b][y+dy-1][x+dx-1][ci] on channels-last tensors
 *                    (X = cat of one or two sources, each a multiple of 32 channels; pitches in floats, multiples of 4):
 *                    fp32 MFMA, contraction split over pixel segments, partial sums in `workspace`, summed without atomics
 *                    (deterministic).  grad_weight [Cout][Cin][3][3] is overwritten.
 */
int c2m_conv3x3_relayout_split_dgrad_f32(c2m_stream_t stream, const float* weight /* [Cout][Cin][3][3] */, int Cin, int Cout,
                                         int pieces, void* wr);
size_t c2m_conv3x3_wgrad_workspace_bytes(int B, int H, int W, int Cin, int Cout);
int c2m_conv3x3_wgrad_f32(c2m_stream_t stream, const c2m_conv_src* src, int nsrc, const float* grad_out, int g_pix_pitch,
                          int g_row_pitch, long long g_img_pitch, int B, int H, int W, int Cin, int Cout, float* grad_weight,
                          void* workspace, size_t workspace_bytes);

/*
 * First layer of an image tower -- 3 input channels, 64 output channels (vgg conv1_1, vgg_arch.py:107-123; conv_first,
 * ref_restoration_arch.py:30): image [B][3][H][W] planar fp32, weight [64][3][3][3], bias [64] or NULL;
 * mean / std [3] (both or neither): (image - mean[c]) / std[c] is applied first (vgg_arch.py:137-138), zero padding in the
 * normalised domain.  Output channels-last with the given pitches (floats) + optional activation; out2: optional
 * 8-channel group-major twin (see c2m_conv3x3_desc.out2).  One image (12 H W bytes) and eight output rows are addressed with
 * 32-bit byte offsets; C2M_ERR_UNSUPPORTED beyond 2^31 (an image of 13 000 x 13 000 pixels, rows of 10^6 pixels).
 */
int c2m_conv3x3_rgb64_f32(c2m_stream_t stream, const float* image, int B, int H, int W, const float* weight,
                          const float* bias, const float* mean, const float* std_, int act, float slope, float* out,
                          int out_pix_pitch, int out_row_pitch, long long out_img_pitch, float* out2, int out2_row_pitch,
                          long long out2_plane_pitch, long long out2_img_pitch);

/* max_idx [B][hq][wq] int64 -> flow [B][hq][wq][2] fp32 (x, y) = (idx % wq - x, idx / wq - y): index_to_flow of
 * corres_generation_arch.py:29-46 for the whole batch, without the zero padding (the consumer bounds-checks). */
int c2m_index_to_flow_f32(c2m_stream_t stream, const int64_t* max_idx, int B, int hq, int wq, float* flow);

/*
 * A whole ResidualBlockNoBN of the decoder bodies in ONE launch (csrc/conv3x3_resblock.hip):
 *
 *     out = x + conv2( relu( conv1(x) + bias1 ) ) + bias2  [+ res2]
 *
 * replaces mmsr/models/archs/arch_util.py:128-136 (`identity + conv2(relu(conv1(x)))`, res_scale 1) -- 96 of them behind
 * ref_restoration_arch.py:91-98,115-122,138-145,158,171,184 -- and, through res2, the stage skip `body(h) + x` of
 * ref_restoration_arch.py:153,166,179 on a body's last block.  C = 64 channels in and out, fp32 channels-last tensors with
 * explicit pitches (floats, multiples of 4; one sample addressed with 32-bit byte offsets), any H, W.  Arithmetic: the f16 x 2
 * flavour of c2m_conv3x3_nhwc_f32 (C2M_CONV_SPLIT_F16X2) for both convolutions -- wr1 / wr2 are the images
 * c2m_conv3x3_relayout_split_f32(pieces = 2) makes for conv1 / conv2 (the SAME cached images the two-launch path uses).  The
 * intermediate tensor never leaves the CU (a ring of f16 x 2 planes in LDS, sliding down 30-column strips) and the identity
 * is rebuilt from the two f16 pieces of x the kernel holds anyway (x~ = x0 + 2^-11 x1', |x - x~| <= 2^-22 |x|): results agree
 * with the two-launch path to that rounding, not bit for bit.  Domain |x|, |relu(conv1)| < 65520, reported through
 * range_flag like C2M_CONV_SPLIT_F16X2.  C2M_ERR_UNSUPPORTED for C != 64 or extents beyond 2^31 bytes per sample.
 */
typedef struct c2m_resblock3x3_desc {
  int B, H, W, C;           /* C = 64 */
  const float* x;           /* input = identity: pixel (0,0), channel 0, sample 0 */
  int x_pix_pitch, x_row_pitch;
  long long x_img_pitch;
  float* out;               /* may not alias x (a step reads halo rows other workgroups' steps write) */
  int out_pix_pitch, out_row_pitch;
  long long out_img_pitch;
  const float* res2;        /* NULL, or a tensor with out's geometry added to the result */
  const void* wr1;          /* conv1 / conv2 weights: c2m_conv3x3_relayout_split_f32(Cin 64, Cout 64, pieces 2) */
  const void* wr2;
  const float* bias1;       /* [64] or NULL */
  const float* bias2;
  int* range_flag;          /* as c2m_conv3x3_desc.range_flag, or NULL */
} c2m_resblock3x3_desc;
int c2m_resblock3x3_supported(int C, int H, int W);   /* 1 if c2m_resblock3x3_nhwc_f32 takes this shape */
int c2m_resblock3x3_nhwc_f32(c2m_stream_t stream, const c2m_resblock3x3_desc* d);

#ifdef __cplusplus
}
#endif
#endif /* C2M_HIP_H */
