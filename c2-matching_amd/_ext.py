"""`_ext` -- top-level module with the names the reference's pybind extension exports
(mmsr/models/archs/DCNv2/src/vision.cpp:3-9; dispatch and argument lists: src/dcn_v2.h:9-73), backed by the
gfx950 kernels of libc2m_hip.so through the C-ABI (include/c2m_hip.h).

The reference's ``dcn_v2.py`` does ``import _ext as _backend`` (dcn_v2.py:6): with this directory on ``sys.path`` the
reference file works unchanged on an MI355X.  Errors surface as RuntimeError like AT_ASSERTM/AT_ERROR did
(dcn_v2_cuda.cu:60-84).  There is no CPU path (the reference has none either: dcn_v2.h:38,72).
"""
from c2m_amd import C2MError, ops as _ops


def _check(input, weight, kernel_h, kernel_w):
    if not input.is_cuda:
        raise RuntimeError("Not implemented on the CPU")  # dcn_v2.h:38
    if weight.shape[2] != kernel_h or weight.shape[3] != kernel_w:
        raise RuntimeError(f"Input shape and kernel shape wont match: ({kernel_h} x {kernel_w} vs "
                           f"{weight.shape[2]} x {weight.shape[3]}).")
    if input.shape[1] != weight.shape[1]:
        raise RuntimeError(f"Input shape and kernel channels wont match: ({input.shape[1]} vs {weight.shape[1]}).")


def dcn_v2_forward(input, weight, bias, offset, mask, kernel_h, kernel_w, stride_h, stride_w, pad_h, pad_w,
                   dilation_h, dilation_w, deformable_group, bf16_mma=False):
    """-> output [B, Co, Ho, Wo] (new tensor).  `bf16_mma=True` (an extension over the reference's signature): GEMM on
    bf16 MFMA with fp32 accumulation, tensors stay float32."""
    _check(input, weight, kernel_h, kernel_w)
    try:
        return _ops.dcn_v2_forward(input, weight, bias, offset, mask, (stride_h, stride_w), (pad_h, pad_w),
                                   (dilation_h, dilation_w), deformable_group, bf16_mma=bf16_mma)
    except C2MError as e:
        raise RuntimeError(str(e)) from e


def dcn_v2_backward(input, weight, bias, offset, mask, grad_output, kernel_h, kernel_w, stride_h, stride_w, pad_h, pad_w,
                    dilation_h, dilation_w, deformable_group, need_input_grad=True):
    """-> [grad_input, grad_offset, grad_mask, grad_weight, grad_bias] (new tensors).  `need_input_grad=False` (an
    extension over the reference's signature) skips the col2im scatter and returns None in its place."""
    _check(input, weight, kernel_h, kernel_w)
    try:
        return list(_ops.dcn_v2_backward(input, weight, bias, offset, mask, grad_output, (stride_h, stride_w),
                                         (pad_h, pad_w), (dilation_h, dilation_w), deformable_group, need_input_grad))
    except C2MError as e:
        raise RuntimeError(str(e)) from e


def dcn_v2_psroi_pooling_forward(*args, **kwargs):
    # Exported by the reference (vision.cpp:7) but never reached by any C2-Matching arch: out of scope.
    raise NotImplementedError("deformable PS-ROI pooling is not part of the C2-Matching hot path")


def dcn_v2_psroi_pooling_backward(*args, **kwargs):
    raise NotImplementedError("deformable PS-ROI pooling is not part of the C2-Matching hot path")
