"""CPU: known-answer tests pinning the DCNv2 restatement (the reference has no CPU DCNv2 and no tests: SURVEY 8c)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

import c2m_oracle as oracle
import torch_port


def _rand(shape, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(shape, generator=g) * scale


def test_zero_offset_unit_mask_is_conv2d():
    B, C, H, W, Co, dg = 2, 16, 9, 11, 8, 4
    x, w, b = _rand((B, C, H, W), 1), _rand((Co, C, 3, 3), 2), _rand((Co,), 3)
    o = oracle.dcn_v2_forward(x.numpy(), w.numpy(), b.numpy(), np.zeros((B, 18 * dg, H, W), np.float32),
                              np.ones((B, 9 * dg, H, W), np.float32), deformable_groups=dg)
    np.testing.assert_allclose(o, F.conv2d(x, w, b, padding=1).numpy(), atol=5e-5)


def test_integer_offsets_are_a_shifted_gather():
    B, C, H, W, Co, dg = 1, 4, 8, 8, 3, 2
    x, w, b = _rand((B, C, H, W), 4), _rand((Co, C, 3, 3), 5), _rand((Co,), 6)
    off = np.zeros((B, 18 * dg, H, W), np.float32)
    off[:, 0::2] = 2.0   # dy
    off[:, 1::2] = -1.0  # dx
    o = oracle.dcn_v2_forward(x.numpy(), w.numpy(), b.numpy(), off, np.ones((B, 9 * dg, H, W), np.float32), deformable_groups=dg)
    shifted = torch.zeros_like(x)
    shifted[:, :, : H - 2, 1:] = x[:, :, 2:, : W - 1]  # sample (y+2, x-1), zero outside
    # interior only: at the border the deformed taps can land inside the image where the plain conv sees padding
    np.testing.assert_allclose(o[..., 1:-1, 1:-1], F.conv2d(shifted, w, b, padding=1).numpy()[..., 1:-1, 1:-1], atol=5e-5)


def test_out_of_range_offsets_give_bias_and_zero_grads():
    B, C, H, W, Co, dg = 1, 4, 6, 7, 3, 2
    x, w, b = _rand((B, C, H, W), 7), _rand((Co, C, 3, 3), 8), _rand((Co,), 9)
    off = np.full((B, 18 * dg, H, W), 1000.0, np.float32)
    m = np.ones((B, 9 * dg, H, W), np.float32)
    o = oracle.dcn_v2_forward(x.numpy(), w.numpy(), b.numpy(), off, m, deformable_groups=dg)
    np.testing.assert_array_equal(o, np.broadcast_to(b.numpy().reshape(1, Co, 1, 1), o.shape))
    gi, go, gm, gw, gb = oracle.dcn_v2_backward(x.numpy(), w.numpy(), b.numpy(), off, m, np.ones_like(o), deformable_groups=dg)
    assert not gi.any() and not go.any() and not gm.any() and not gw.any()
    np.testing.assert_allclose(gb, np.full(Co, H * W, np.float32))


def test_dcn_sep_init_state_offsets_equal_pre_offset():
    # re_init_dcn_offset state (ref_restoration_arch.py:42-49): conv_offset_mask == 0 -> offset = pre_offset, mask = 0.5
    B, C, H, W, Co, dg = 1, 8, 6, 6, 4, 2
    x, w, b = _rand((B, C, H, W), 10), _rand((Co, C, 3, 3), 11), _rand((Co,), 12)
    off = np.zeros((B, 18 * dg, H, W), np.float32)
    off[:, 1::2] = 1.0
    o = oracle.dcn_v2_forward(x.numpy(), w.numpy(), b.numpy(), off, np.full((B, 9 * dg, H, W), 0.5, np.float32), deformable_groups=dg)
    shifted = torch.zeros_like(x)
    shifted[:, :, :, : W - 1] = x[:, :, :, 1:]
    want = F.conv2d(0.5 * shifted, w, None, padding=1).numpy() + b.numpy().reshape(1, Co, 1, 1)
    np.testing.assert_allclose(o[..., 1:-1, 1:-1], want[..., 1:-1, 1:-1], atol=5e-5)


@pytest.mark.parametrize("cfg", [
    dict(B=2, C=16, H=9, W=11, Co=8, k=(3, 3), s=(1, 1), p=(1, 1), d=(1, 1), dg=4),
    dict(B=1, C=6, H=10, W=13, Co=5, k=(3, 2), s=(2, 1), p=(1, 2), d=(1, 2), dg=3),
])
def test_forward_and_backward_against_fp64_autograd(cfg):
    B, C, H, W, Co, dg = cfg["B"], cfg["C"], cfg["H"], cfg["W"], cfg["Co"], cfg["dg"]
    kh, kw = cfg["k"]
    Ho = (H + 2 * cfg["p"][0] - (cfg["d"][0] * (kh - 1) + 1)) // cfg["s"][0] + 1
    Wo = (W + 2 * cfg["p"][1] - (cfg["d"][1] * (kw - 1) + 1)) // cfg["s"][1] + 1
    K = kh * kw
    x, w, b = _rand((B, C, H, W), 20), _rand((Co, C, kh, kw), 21), _rand((Co,), 22)
    off, m = _rand((B, 2 * K * dg, Ho, Wo), 23, 2.5), torch.sigmoid(_rand((B, K * dg, Ho, Wo), 24))
    o = oracle.dcn_v2_forward(x.numpy(), w.numpy(), b.numpy(), off.numpy(), m.numpy(), cfg["s"], cfg["p"], cfg["d"], dg)
    leaves = [t.double().requires_grad_() for t in (x, w, b, off, m)]
    ref = torch_port.dcn_v2_reference(*leaves, cfg["s"], cfg["p"], cfg["d"], dg)
    np.testing.assert_allclose(o, ref.detach().numpy(), atol=5e-5)
    go = _rand(ref.shape, 25)
    ref.backward(go.double())
    grads = oracle.dcn_v2_backward(x.numpy(), w.numpy(), b.numpy(), off.numpy(), m.numpy(), go.numpy(), cfg["s"], cfg["p"], cfg["d"], dg)
    for got, leaf in zip(grads, (leaves[0], leaves[3], leaves[4], leaves[1], leaves[2])):
        np.testing.assert_allclose(got, leaf.grad.numpy(), atol=2e-4 * max(1.0, float(leaf.grad.abs().max())))


def test_torch_restatement_passes_gradcheck():
    t = lambda *s: torch.randn(*s, dtype=torch.double, requires_grad=True)  # noqa: E731
    off = (torch.rand(1, 36, 5, 6, dtype=torch.double) * 3 - 1.5 + 0.013).requires_grad_()
    assert torch.autograd.gradcheck(lambda *a: torch_port.dcn_v2_reference(*a, dg=2),
                                    [t(1, 4, 5, 6), t(3, 4, 3, 3), t(3), off, torch.rand(1, 18, 5, 6, dtype=torch.double, requires_grad=True)],
                                    eps=1e-6, atol=1e-5)


def test_conv_port_agrees_with_oracle_on_indices():
    import synth
    fi = oracle.feature_normalize(synth.gaussish((64, 16, 15), 31))
    fr = oracle.feature_normalize(synth.gaussish((64, 13, 18), 32))
    i1, v1 = oracle.feature_match_index(fi, fr, 3, 1, 1, True, True)
    i2, v2 = torch_port.feature_match_index_conv(torch.from_numpy(fi), torch.from_numpy(fr), 3, 1, 1, True, True, chunk_elems=50 * 13 * 14)
    assert np.array_equal(i1, i2.numpy())
    np.testing.assert_allclose(v1, v2.numpy(), atol=2e-6)
