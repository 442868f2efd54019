"""On-device image metrics with the reference's definitions (``mmsr/utils/metrics.py``: ``psnr`` :34-66, ``ssim``
:69-143, ``bgr2ycbcr`` :146-168; ``tensor2img`` ``mmsr/utils/util.py:107-162``), SURVEY.md 8f row 4.

The reference pulls every SR image back to the host, converts it to a numpy HWC/BGR array and evaluates PSNR / PSNR_Y /
SSIM_Y there, one image at a time with an ``empty_cache`` in between (ref_restoration_model.py:295-351).  Here the same
quantities are reduced on the GPU for the whole batch (float64 accumulation) and only three scalars per image cross
PCIe.  Same names and argument meaning as the reference functions; inputs are torch tensors instead of ndarrays.
"""
import math

import torch
import torch.nn.functional as F


def tensor2img_device(t, min_max=(0, 1)):
    """[B,3,H,W] or [3,H,W] RGB tensor -> float tensor(s) [.., H, W, 3] in BGR order, values round(255 * clamp(t)):
    what ``tensor2img`` (util.py:107-162) hands to the metrics (its ``astype(uint8)`` result is discarded there, so the
    images stay float), without leaving the device."""
    x = t.detach().float().clamp(*min_max)
    x = (x - min_max[0]) / (min_max[1] - min_max[0])
    x = (x * 255.0).round()
    return x.flip(-3).movedim(-3, -1)


def _crop(img, crop_border):
    if crop_border:
        return img[..., crop_border:-crop_border, crop_border:-crop_border, :]
    return img


def _hwc(img):
    return img[..., None] if img.dim() == 2 else img


def psnr(img1, img2, crop_border=0):
    """PSNR of two images [.., H, W, C] (or [H, W]) with range [0, 255] (metrics.py:34-66).  Leading dims are a batch:
    returns a float64 tensor of that shape (a python float for a single image)."""
    assert img1.shape == img2.shape, f'Image shapes are differnet: {img1.shape}, {img2.shape}.'
    a, b = _crop(_hwc(img1), crop_border).double(), _crop(_hwc(img2), crop_border).double()
    mse = ((a - b) ** 2).mean(dim=(-3, -2, -1))
    out = torch.where(mse == 0, torch.full_like(mse, float('inf')), 20.0 * torch.log10(255.0 / torch.sqrt(mse)))
    return float(out) if out.dim() == 0 else out


def bgr2ycbcr(img, only_y=True):
    """MATLAB rgb2ycbcr on a BGR float image in [0, 1] (metrics.py:146-168).  Returns values in [0, 1]."""
    x = img.double() * 255.0
    if only_y:
        coef = torch.tensor([24.966, 128.553, 65.481], dtype=torch.float64, device=img.device)
        return ((x @ coef) / 255.0 + 16.0) / 255.0
    m = torch.tensor([[24.966, 112.0, -18.214], [128.553, -74.203, -93.786], [65.481, -37.797, 112.0]],
                     dtype=torch.float64, device=img.device)
    add = torch.tensor([16.0, 128.0, 128.0], dtype=torch.float64, device=img.device)
    return ((x @ m) / 255.0 + add) / 255.0


def _gauss_window(dev):
    x = torch.arange(11, dtype=torch.float64, device=dev) - 5.0
    k = torch.exp(-(x * x) / (2 * 1.5 * 1.5))
    k = k / k.sum()
    return torch.outer(k, k)[None, None]


def ssim(img1, img2, crop_border=0):
    """SSIM (metrics.py:69-143): 11x11 Gaussian window (sigma 1.5), 'valid' part only, mean over channels."""
    assert img1.shape == img2.shape, f'Image shapes are differnet: {img1.shape}, {img2.shape}.'
    a, b = _crop(_hwc(img1), crop_border).double(), _crop(_hwc(img2), crop_border).double()
    lead = a.shape[:-3]
    a = a.reshape(-1, *a.shape[-3:]).movedim(-1, 1).reshape(-1, 1, a.shape[-3], a.shape[-2])
    b = b.reshape(-1, *b.shape[-3:]).movedim(-1, 1).reshape(-1, 1, b.shape[-3], b.shape[-2])
    w = _gauss_window(a.device)
    c1, c2 = (0.01 * 255) ** 2, (0.03 * 255) ** 2
    mu1, mu2 = F.conv2d(a, w), F.conv2d(b, w)
    s11 = F.conv2d(a * a, w) - mu1 * mu1
    s22 = F.conv2d(b * b, w) - mu2 * mu2
    s12 = F.conv2d(a * b, w) - mu1 * mu2
    smap = ((2 * mu1 * mu2 + c1) * (2 * s12 + c2)) / ((mu1 * mu1 + mu2 * mu2 + c1) * (s11 + s22 + c2))
    per = smap.mean(dim=(-3, -2, -1))
    nch = img1.shape[-1] if img1.dim() > 2 else 1
    out = per.reshape(*lead, nch).mean(dim=-1) if lead else per.reshape(nch).mean()
    return float(out) if out.dim() == 0 else out


def validation_metrics(sr, gt, crop_border=4):
    """The three numbers of ``nondist_validation`` (ref_restoration_model.py:338-351) for a batch of RGB tensors
    [B,3,H,W] in [0,1]: dict of float64 tensors [B] 'psnr', 'psnr_y', 'ssim_y'.  crop_border defaults to the scale (4),
    the rule of options.py:56-57."""
    s, g = tensor2img_device(sr), tensor2img_device(gt)
    sy, gy = bgr2ycbcr(s / 255.0, only_y=True) * 255.0, bgr2ycbcr(g / 255.0, only_y=True) * 255.0
    return {'psnr': psnr(s, g, crop_border), 'psnr_y': psnr(sy[..., None], gy[..., None], crop_border),
            'ssim_y': ssim(sy[..., None], gy[..., None], crop_border)}


__all__ = ['psnr', 'ssim', 'bgr2ycbcr', 'tensor2img_device', 'validation_metrics']
_ = math
