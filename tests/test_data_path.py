"""mmsr.data.pil_bicubic against the installed Pillow (the library the reference's dataset calls,
mmsr/data/ref_cufed_dataset.py:118-130): bit-exact on uint8 images, CPU here and GPU under -m gpu."""
import numpy as np
import pytest
import torch


def _images():
    rng = np.random.default_rng(5)
    smooth = np.kron(rng.random((3, 20, 24)) * 255, np.ones((1, 8, 8))) + rng.random((3, 160, 192)) * 30
    return [np.clip(smooth, 0, 255).astype(np.uint8), (rng.random((3, 164, 100)) * 255).astype(np.uint8)]


def _check(device):
    Image = pytest.importorskip("PIL.Image")
    from mmsr.data import make_lq_and_up, pil_bicubic_resize
    for a in _images():
        C, H, W = a.shape
        pil = Image.fromarray(a.transpose(1, 2, 0))
        for (oh, ow) in ((H // 4, W // 4), (H * 2, W * 2), (H // 4, W), (37, 53)):
            want = np.array(pil.resize((ow, oh), Image.BICUBIC)).transpose(2, 0, 1)
            got = pil_bicubic_resize(torch.from_numpy(a).to(device), oh, ow).cpu().numpy()
            assert np.array_equal(got, want), (a.shape, oh, ow, int(np.abs(got.astype(int) - want.astype(int)).max()))
        lq_pil = pil.resize((W // 4, H // 4), Image.BICUBIC)
        up_pil = lq_pil.resize((W, H), Image.BICUBIC)
        lq, up = make_lq_and_up(torch.from_numpy(a)[None].to(device), 4)
        assert np.array_equal(lq[0].cpu().numpy(), np.array(lq_pil).transpose(2, 0, 1))
        assert np.array_equal(up[0].cpu().numpy(), np.array(up_pil).transpose(2, 0, 1))


def test_pil_bicubic_bit_exact_cpu():
    _check(torch.device("cpu"))


@pytest.mark.gpu
def test_pil_bicubic_bit_exact_gpu(dev):
    _check(dev)
