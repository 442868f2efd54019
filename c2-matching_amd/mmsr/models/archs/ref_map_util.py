"""MI355X drop-in for the reference's ``mmsr/models/archs/ref_map_util.py``.

Same names, argument meaning and return contract as ref_map_util.py:4-23 (``sample_patches``) and :26-86
(``feature_match_index``); the work happens in hand-written gfx950 kernels (csrc/corr_argmax.hip) reached through
the C-ABI of include/c2m_hip.h.  ``feature_match_index_batched`` is an additional entry point that removes the
per-sample Python loop of corres_generation_arch.py:52.

No unfolded patch tensor, no [Nr x Nq] score volume and no chunk loop exist on this path, so the reference's
``batch_size = int(1024**2 * 512 / (h*w))`` memory bound (ref_map_util.py:54-60) has no counterpart.
"""
import torch

from c2m_amd import ops as _ops


def sample_patches(inputs, patch_size=3, stride=1):
    """(c, h, w) -> (c, patch_size, patch_size, n_patches), patches row-major (ref_map_util.py:4-23).

    Kept for API compatibility only; feature_match_index never materialises patches."""
    c, h, w = inputs.shape
    nh, nw = (h - patch_size) // stride + 1, (w - patch_size) // stride + 1
    sc, sh, sw = inputs.stride()
    win = inputs.as_strided((c, patch_size, patch_size, nh, nw), (sc, sh, sw, sh * stride, sw * stride))
    return win.reshape(c, patch_size, patch_size, nh * nw)


def feature_match_index_batched(feat_input, feat_ref, patch_size=3, input_stride=1, ref_stride=1, is_norm=True,
                                norm_input=False):
    """feat_input (b, c, h, w), feat_ref (b, c, h', w') -> max_idx int64 (b, ho, wo), max_val float32 (b, ho, wo).
    Half / bf16 features (autocast) are matched in float32."""
    return _ops.feature_match_index_batched(feat_input.float(), feat_ref.float(), patch_size, input_stride, ref_stride,
                                            is_norm, norm_input)


def feature_match_index(feat_input, feat_ref, patch_size=3, input_stride=1, ref_stride=1, is_norm=True,
                        norm_input=False):
    """Patch matching between input and reference features (ref_map_util.py:26-86).

    Args and returns as the reference: feat_input (c, h, w), feat_ref (c, h', w') on the GPU ->
    max_idx int64 (ho, wo) = index of the best ref patch (row-major, lowest index on ties), max_val float32."""
    if feat_input.dim() != 3 or feat_ref.dim() != 3:
        raise ValueError("feature_match_index expects (c, h, w) tensors; use feature_match_index_batched for batches")
    idx, val = _ops.feature_match_index_batched(feat_input[None].float(), feat_ref[None].float(), patch_size,
                                                input_stride, ref_stride, is_norm, norm_input)
    return idx[0], val[0]


__all__ = ["sample_patches", "feature_match_index", "feature_match_index_batched"]
_ = torch  # torch is the tensor type of this module's API
