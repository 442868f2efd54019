"""Tensor-level entry points of the MI355X hot path (PyTorch is plumbing only: memory, streams).

Every function takes CUDA(=HIP) float32 tensors, enqueues hand-written gfx950 kernels from libc2m_hip.so on the
current torch stream and returns freshly allocated tensors.  CPU tensors are rejected: there is no fallback path.
"""
import torch

from . import _lib


def _dev_f32(t, name):
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise _lib.C2MError(f"{name} must be a tensor on the GPU (the HIP path has no CPU fallback)")
    if t.dtype != torch.float32:
        raise _lib.C2MError(f"{name} must be float32, got {t.dtype}")
    return t if t.is_contiguous() else t.contiguous()


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _pair(v):
    return (int(v), int(v)) if isinstance(v, int) else (int(v[0]), int(v[1]))


def feature_normalize(x):
    """x [B,C,H,W] or [C,H,W] -> per-pixel channel-normalised copy (corres_generation_arch.py:56-58)."""
    x = _dev_f32(x, "x")
    shp = x.shape
    xb = x.view(1, *shp) if x.dim() == 3 else x
    B, C = xb.shape[0], xb.shape[1]
    out = torch.empty_like(xb)
    with torch.cuda.device(x.device):
        _lib.check(_lib.lib().c2m_feature_normalize_f32(_stream(), xb.data_ptr(), B, C, xb.numel() // (B * C),
                                                        out.data_ptr()), "c2m_feature_normalize_f32")
    return out.view(shp)


def feature_match_index_batched(feat_in, feat_ref, patch_size=3, input_stride=1, ref_stride=1, is_norm=True,
                                norm_input=False, force_generic=False):
    """Batched ref_map_util.feature_match_index: feat_in [B,C,Hq,Wq], feat_ref [B,C,Hr,Wr] ->
    (max_idx int64 [B,Hqp,Wqp], max_val float32 [B,Hqp,Wqp])."""
    fi, fr = _dev_f32(feat_in, "feat_in"), _dev_f32(feat_ref, "feat_ref")
    if fi.dim() != 4 or fr.dim() != 4 or fi.shape[:2] != fr.shape[:2] or fi.device != fr.device:
        raise _lib.C2MError("feat_in / feat_ref must be [B,C,H,W] with equal B, C and device")
    B, C, Hq, Wq = fi.shape
    Hr, Wr = fr.shape[2:]
    p, si, sr = int(patch_size), int(input_stride), int(ref_stride)
    if min(Hq, Wq, Hr, Wr) < p:
        raise _lib.C2MError("feature maps smaller than the patch")
    Hqp, Wqp = (Hq - p) // si + 1, (Wq - p) // si + 1
    L = _lib.lib()
    with torch.cuda.device(fi.device):
        nbytes = L.c2m_feature_match_workspace_bytes(B, Hq, Wq, Hr, Wr)
        ws = torch.empty(nbytes, dtype=torch.uint8, device=fi.device)
        idx = torch.empty((B, Hqp, Wqp), dtype=torch.int64, device=fi.device)
        val = torch.empty((B, Hqp, Wqp), dtype=torch.float32, device=fi.device)
        _lib.check(L.c2m_feature_match_index_f32(_stream(), fi.data_ptr(), fr.data_ptr(), B, C, Hq, Wq, Hr, Wr, p, si,
                                                 sr, int(bool(is_norm)), int(bool(norm_input)), int(bool(force_generic)),
                                                 idx.data_ptr(), val.data_ptr(), ws.data_ptr(), nbytes),
                   "c2m_feature_match_index_f32")
    return idx, val


def build_pre_offsets(max_idx, h, w, scales=(1, 2, 4)):
    """max_idx int64 [B,h-2,w-2] -> tuple of pre-offset tensors [B,9,s*h,s*w,2] for s in scales (subset of 1,2,4)."""
    if not max_idx.is_cuda or max_idx.dtype != torch.int64:
        raise _lib.C2MError("max_idx must be an int64 GPU tensor")
    mi = max_idx.contiguous()
    B = mi.shape[0]
    if tuple(mi.shape[1:]) != (h - 2, w - 2):
        raise _lib.C2MError("max_idx must be [B, h-2, w-2]")
    if not set(scales) <= {1, 2, 4} or len(set(scales)) != len(tuple(scales)):
        raise _lib.C2MError(f"scales must be distinct members of (1, 2, 4), got {tuple(scales)}")
    outs = {s: torch.empty((B, 9, h * s, w * s, 2), dtype=torch.float32, device=mi.device) for s in scales}
    ptr = lambda s: outs[s].data_ptr() if s in outs else None  # noqa: E731
    with torch.cuda.device(mi.device):
        _lib.check(_lib.lib().c2m_build_pre_offsets_f32(_stream(), mi.data_ptr(), B, h, w, ptr(1), ptr(2), ptr(4)),
                   "c2m_build_pre_offsets_f32")
    return tuple(outs[s] for s in scales)


def _dcn_geom(inp, weight, stride, padding, dilation):
    B, C, H, W = inp.shape
    Co, Ck, kh, kw = weight.shape
    (sh, sw), (ph, pw), (dh, dw) = _pair(stride), _pair(padding), _pair(dilation)
    if Ck != C:
        raise _lib.C2MError(f"Input shape and kernel channels wont match: ({C} vs {Ck}).")
    Ho = (H + 2 * ph - (dh * (kh - 1) + 1)) // sh + 1
    Wo = (W + 2 * pw - (dw * (kw - 1) + 1)) // sw + 1
    return (B, C, H, W, Co, kh, kw, sh, sw, ph, pw, dh, dw), Ho, Wo


def dcn_v2_forward(inp, weight, bias, offset, mask, stride=1, padding=1, dilation=1, deformable_groups=1,
                   bf16_mma=False):
    """bf16_mma=True: the implicit GEMM runs on bf16 MFMA with fp32 accumulation (weights and blended samples rounded to
    bf16); tensors stay float32.  For callers that asked for reduced precision (bf16 autocast), not the default."""
    inp, weight, bias, offset, mask = (_dev_f32(t, n) for t, n in
                                       ((inp, "input"), (weight, "weight"), (bias, "bias"), (offset, "offset"), (mask, "mask")))
    g, Ho, Wo = _dcn_geom(inp, weight, stride, padding, dilation)
    B, C, H, W, Co, kh, kw = g[:7]
    dg = int(deformable_groups)
    if tuple(offset.shape) != (B, 2 * dg * kh * kw, Ho, Wo) or tuple(mask.shape) != (B, dg * kh * kw, Ho, Wo):
        raise _lib.C2MError("offset/mask shape does not match [B, 2*dg*kh*kw, Ho, Wo] / [B, dg*kh*kw, Ho, Wo]")
    L = _lib.lib()
    with torch.cuda.device(inp.device):
        nbytes = L.c2m_dcn_v2_forward_workspace_bytes(B, C, H, W, Co, kh, kw, dg)
        ws = torch.empty(max(nbytes, 1), dtype=torch.uint8, device=inp.device)
        out = torch.empty((B, Co, Ho, Wo), dtype=torch.float32, device=inp.device)
        fn = L.c2m_dcn_v2_forward_bf16mma_f32 if bf16_mma else L.c2m_dcn_v2_forward_f32
        _lib.check(fn(_stream(), inp.data_ptr(), weight.data_ptr(), bias.data_ptr(), offset.data_ptr(), mask.data_ptr(),
                      *g, dg, out.data_ptr(), ws.data_ptr(), nbytes), "c2m_dcn_v2_forward")
    return out


def dcn_v2_backward(inp, weight, bias, offset, mask, grad_output, stride=1, padding=1, dilation=1, deformable_groups=1,
                    need_input_grad=True):
    """-> (grad_input or None, grad_offset, grad_mask, grad_weight, grad_bias)."""
    inp, weight, bias, offset, mask, grad_output = (_dev_f32(t, n) for t, n in (
        (inp, "input"), (weight, "weight"), (bias, "bias"), (offset, "offset"), (mask, "mask"), (grad_output, "grad_output")))
    g, Ho, Wo = _dcn_geom(inp, weight, stride, padding, dilation)
    dg = int(deformable_groups)
    L = _lib.lib()
    with torch.cuda.device(inp.device):
        nbytes = L.c2m_dcn_v2_backward_workspace_bytes(*g, dg)
        ws = torch.empty(max(nbytes, 1), dtype=torch.uint8, device=inp.device)
        go, gm, gw, gb = (torch.empty_like(t) for t in (offset, mask, weight, bias))
        gi = torch.empty_like(inp) if need_input_grad else None
        _lib.check(L.c2m_dcn_v2_backward_f32(_stream(), inp.data_ptr(), weight.data_ptr(), bias.data_ptr(),
                                             offset.data_ptr(), mask.data_ptr(), grad_output.data_ptr(), *g, dg,
                                             gi.data_ptr() if gi is not None else None, go.data_ptr(), gm.data_ptr(),
                                             gw.data_ptr(), gb.data_ptr(),
                                             ws.data_ptr(), nbytes), "c2m_dcn_v2_backward_f32")
    return gi, go, gm, gw, gb


def dcn_fuse_offsets(conv_out, pre_offset, deformable_groups, kernel_taps, abs_sum=None):
    """conv_offset_mask output [B,3*dg*K,H,W] (+ pre_offset [B,K,H,W,2] or None) -> (offset [B,2*dg*K,H,W], mask
    [B,dg*K,H,W]) as dcn_v2.py:229-245 builds them; abs_sum (float64 GPU tensor of 256 slots, zeroed by the caller) accumulates sum|learned offset| across its slots."""
    conv_out = _dev_f32(conv_out, "conv_out")
    B, C3, H, W = conv_out.shape
    dg, K = int(deformable_groups), int(kernel_taps)
    if C3 != 3 * dg * K:
        raise _lib.C2MError("conv_out must have 3*dg*K channels")
    if pre_offset is not None:
        pre_offset = _dev_f32(pre_offset, "pre_offset")
        if tuple(pre_offset.shape) != (B, K, H, W, 2):
            raise _lib.C2MError("pre_offset must be [B, K, H, W, 2]")
    if abs_sum is not None and (abs_sum.dtype != torch.float64 or abs_sum.numel() < 256 or not abs_sum.is_cuda):
        raise _lib.C2MError("abs_sum must be a float64 GPU tensor with 256 slots (C2M_ABS_SUM_SLOTS)")
    offset = torch.empty((B, 2 * dg * K, H, W), dtype=torch.float32, device=conv_out.device)
    mask = torch.empty((B, dg * K, H, W), dtype=torch.float32, device=conv_out.device)
    with torch.cuda.device(conv_out.device):
        _lib.check(_lib.lib().c2m_dcn_fuse_offsets_f32(_stream(), conv_out.data_ptr(),
                                                       pre_offset.data_ptr() if pre_offset is not None else None,
                                                       B, dg, K, H, W, offset.data_ptr(), mask.data_ptr(),
                                                       abs_sum.data_ptr() if abs_sum is not None else None),
                   "c2m_dcn_fuse_offsets_f32")
    return offset, mask
