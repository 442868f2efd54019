"""On-device replacement of the reference dataset's PIL resize chain.

``RefCUFEDDataset`` builds every sample on the host with ``PIL.Image.resize(..., Image.BICUBIC)``: GT -> LR (/4) and
LR -> bicubic x4 for both the input and the Ref image (mmsr/data/ref_cufed_dataset.py:118-143), one image at a time.
``pil_bicubic_resize`` reproduces Pillow's 8-bit resampler BIT FOR BIT on whole batches of uint8 tensors, on whatever
device they live on: the same coefficient construction (Pillow ``Resample.c``: ``precompute_coeffs`` with the a = -0.5
bicubic kernel and support 2 * max(scale, 1), ``normalize_coeffs_8bpc`` fixed point with 22 fraction bits), the same two
passes (horizontal, then vertical) with the intermediate image rounded and clipped to uint8.  The integer arithmetic is
carried out in float64 matrix products: every operand is an integer below 2**31 and every sum stays below 2**53, so the
products and sums are exact.  Checked against the installed Pillow in tests/test_data_path.py.
"""
import math

import torch

_PRECISION_BITS = 32 - 8 - 2


def _bicubic(x):
    a = -0.5
    if x < 0.0:
        x = -x
    if x < 1.0:
        return ((a + 2.0) * x - (a + 3.0)) * x * x + 1
    if x < 2.0:
        return (((x - 5) * x + 8) * x - 4) * a
    return 0.0


def _coeff_matrix(in_size, out_size):
    """[out_size, in_size] float64 matrix of Pillow's fixed-point bicubic coefficients (integers)."""
    scale = in_size / out_size
    filterscale = max(scale, 1.0)
    support = 2.0 * filterscale
    ss = 1.0 / filterscale
    m = torch.zeros((out_size, in_size), dtype=torch.float64)
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        xmin = max(int(center - support + 0.5), 0)
        xmax = min(int(center + support + 0.5), in_size)
        w = [_bicubic((x + xmin - center + 0.5) * ss) for x in range(xmax - xmin)]
        ww = 0.0
        for v in w:
            ww += v
        for x, v in enumerate(w):
            k = v / ww if ww != 0.0 else v
            fixed = int(-0.5 + k * (1 << _PRECISION_BITS)) if k < 0 else int(0.5 + k * (1 << _PRECISION_BITS))
            m[xx, xmin + x] = float(fixed)
    return m


_cache = {}


def _coeffs(in_size, out_size, device):
    key = (in_size, out_size, str(device))
    if key not in _cache:
        _cache[key] = _coeff_matrix(in_size, out_size).to(device)
    return _cache[key]


def _pass(x, m):
    """x [..., n_in] float64 integers, m [n_out, n_in] -> clip8((2^21 + x . m^T) >> 22) as float64 integers."""
    acc = x @ m.t() + float(1 << (_PRECISION_BITS - 1))
    return torch.floor(acc / float(1 << _PRECISION_BITS)).clamp_(0.0, 255.0)


def pil_bicubic_resize(img, out_h, out_w):
    """img: uint8 tensor [..., H, W] (any leading dims: batch, channels) -> uint8 [..., out_h, out_w], identical to
    ``PIL.Image.fromarray(...).resize((out_w, out_h), Image.BICUBIC)`` applied per channel image."""
    if img.dtype != torch.uint8:
        raise TypeError('pil_bicubic_resize works on uint8 images (Pillow quantises to 8 bits between the two passes)')
    H, W = img.shape[-2:]
    x = img.to(torch.float64)
    if W != out_w:
        x = _pass(x, _coeffs(W, out_w, img.device))                       # horizontal
    if H != out_h:
        x = _pass(x.transpose(-1, -2), _coeffs(H, out_h, img.device)).transpose(-1, -2)   # vertical
    return x.to(torch.uint8)


def make_lq_and_up(img_gt_u8, scale=4):
    """uint8 GT batch [B,3,H,W] (H, W multiples of scale) -> (lq uint8 [B,3,H/s,W/s], up uint8 [B,3,H,W]): the
    ``img_in_lq`` / ``img_in_up`` (and ``img_ref_lq`` / ``img_ref_up``) construction of ref_cufed_dataset.py:118-130."""
    H, W = img_gt_u8.shape[-2:]
    lq = pil_bicubic_resize(img_gt_u8, H // scale, W // scale)
    return lq, pil_bicubic_resize(lq, H, W)


_ = math
