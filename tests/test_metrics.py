"""mmsr.utils.metrics (device PSNR / PSNR_Y / SSIM_Y) against values produced by the reference's own
mmsr/utils/metrics.py (tests/golden/make_golden.py metrics).  Runs on CPU tensors here and on the GPU under -m gpu."""
import numpy as np
import pytest
import torch


def _check(device, golden_dir):
    from make_golden import metric_images
    from mmsr.utils import metrics
    gold = np.load(f"{golden_dir}/metrics_golden.npz")
    for k in range(3):
        a, b = (torch.from_numpy(x).to(device) for x in metric_images(k))
        assert abs(metrics.psnr(a, b, crop_border=4) - float(gold[f"psnr{k}"])) < 1e-4
        ay = metrics.bgr2ycbcr(a / 255.0, only_y=True) * 255.0
        by = metrics.bgr2ycbcr(b / 255.0, only_y=True) * 255.0
        assert abs(metrics.psnr(ay, by, crop_border=4) - float(gold[f"psnr_y{k}"])) < 1e-4
        assert abs(metrics.ssim(ay, by, crop_border=4) - float(gold[f"ssim_y{k}"])) < 1e-6
    # batched entry point == per-image calls
    a, b = (torch.from_numpy(x).to(device) for x in metric_images(0))
    rgb = lambda t: (t / 255.0).flip(-1).movedim(-1, 0)  # noqa: E731
    batch_sr = torch.stack([rgb(b), rgb(a)])
    batch_gt = torch.stack([rgb(a), rgb(a)])
    m = metrics.validation_metrics(batch_sr, batch_gt, crop_border=4)
    assert abs(float(m["psnr"][0]) - float(gold["psnr0"])) < 1e-4
    assert abs(float(m["psnr_y"][0]) - float(gold["psnr_y0"])) < 1e-4
    assert abs(float(m["ssim_y"][0]) - float(gold["ssim_y0"])) < 1e-6
    assert torch.isinf(m["psnr"][1]) and abs(float(m["ssim_y"][1]) - 1.0) < 1e-12


def test_metrics_match_reference_cpu(golden_dir):
    _check(torch.device("cpu"), golden_dir)


@pytest.mark.gpu
def test_metrics_match_reference_gpu(dev, golden_dir):
    _check(dev, golden_dir)
