"""GPU parity tests of the DCNv2 path (forward, backward, offset/mask assembly, Python operator).  All calls go through
the C-ABI.  DCNv2 is floating point with a different summation order than the oracle (MFMA k-order (g,tap,c) vs the
reference column order c*9+tap; cuBLAS order is unspecified anyway): tolerances are stated per test and are far inside
north_star's 1e-3 abs bound on SR pixels."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def env(dev):
    import c2m_amd
    import c2m_oracle as oracle
    import synth
    return c2m_amd.ops, oracle, synth


def _t(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


def _case(synth, B, C, H, W, Co, kh, kw, st, pd, dl, dg, seed, off_scale=3.0):
    Ho = (H + 2 * pd[0] - (dl[0] * (kh - 1) + 1)) // st[0] + 1
    Wo = (W + 2 * pd[1] - (dl[1] * (kw - 1) + 1)) // st[1] + 1
    K = kh * kw
    x = synth.gaussish((B, C, H, W), seed)
    w = (synth.gaussish((Co, C, kh, kw), seed + 1) * (1.0 / np.sqrt(C * K))).astype(np.float32)
    b = synth.gaussish((Co,), seed + 2)
    off = synth.gaussish((B, 2 * K * dg, Ho, Wo), seed + 3) * off_scale
    off[:, :, 0, :] += 40.0          # a band of far out-of-range samples
    off[:, 0::2, -1, :] = np.round(off[:, 0::2, -1, :])   # integer coordinates (zero-weight corners)
    msk = synth.uniform((B, K * dg, Ho, Wo), seed + 4, 0.0, 1.0)
    return x, w, b, off, msk


SHAPES = [
    # B, C, H, W, Co, kh, kw, stride, pad, dil, dg
    (2, 256, 12, 14, 256, 3, 3, (1, 1), (1, 1), (1, 1), 8),   # small DynAgg layer (ref_restoration_arch.py:77-85)
    (2, 128, 17, 19, 128, 3, 3, (1, 1), (1, 1), (1, 1), 8),   # medium (:101-109)
    (3, 64, 33, 21, 64, 3, 3, (1, 1), (1, 1), (1, 1), 8),     # large (:124-132), ragged pixel count
    (1, 8, 10, 13, 5, 3, 2, (2, 1), (1, 2), (1, 2), 2),       # strides / dilation / non-square kernel / Co not % 32
    (1, 16, 9, 9, 40, 1, 1, (1, 1), (0, 0), (1, 1), 4),       # 1x1 kernel, Co = 40
    (1, 48, 11, 9, 24, 3, 3, (1, 1), (1, 1), (1, 1), 6),      # 8-channel groups, dg % 4 != 0: channels-last path without group pairing
    (1, 24, 7, 8, 16, 3, 3, (1, 1), (1, 1), (1, 1), 3),       # odd group count: NCHW kernel
    (2, 64, 9, 10, 96, 3, 3, (2, 2), (1, 1), (1, 1), 2),      # 32-channel groups, Co = 96 (padded m-tiles), stride 2
]


@pytest.mark.parametrize("shape", SHAPES)
def test_forward_matches_oracle(env, dev, shape):
    ops, oracle, synth = env
    B, C, H, W, Co, kh, kw, st, pd, dl, dg = shape
    x, w, b, off, msk = _case(synth, B, C, H, W, Co, kh, kw, st, pd, dl, dg, 300)
    got = ops.dcn_v2_forward(_t(x, dev), _t(w, dev), _t(b, dev), _t(off, dev), _t(msk, dev), st, pd, dl, dg).cpu().numpy()
    want = oracle.dcn_v2_forward(x, w, b, off, msk, st, pd, dl, dg)
    assert got.shape == want.shape
    np.testing.assert_allclose(got, want, rtol=0, atol=2e-5 * max(1.0, float(np.abs(want).max())))


@pytest.mark.parametrize("shape", [SHAPES[2], SHAPES[5]])
def test_forward_nhwc_group_major_input(env, dev, shape):
    """Fused-path ABI with 8-channel deformable groups: gathering from the group-major twin of the bordered copy
    (BorderedNHWC.grouped8, what the producing convolution writes) is the same arithmetic as gathering from the
    channels-last copy -> bit-identical outputs, and both match the oracle."""
    ops, oracle, synth = env
    B, C, H, W, Co, kh, kw, st, pd, dl, dg = shape
    x, w, b, off, msk = _case(synth, B, C, H, W, Co, kh, kw, st, pd, dl, dg, 310)
    args = (_t(w, dev), _t(b, dev), _t(off, dev), _t(msk, dev), dg)
    bo = ops.BorderedNHWC(_t(x, dev))
    assert bo.grouped8 is None
    plain = ops.dcn_v2_forward_nhwc(bo, *args, nhwc_out=False)
    bo.grouped8 = bo.buf.view(B, H + 3, W + 3, C // 8, 8).permute(0, 3, 1, 2, 4).contiguous()
    grouped = ops.dcn_v2_forward_nhwc(bo, *args, nhwc_out=False)
    assert torch.equal(plain, grouped)
    want = oracle.dcn_v2_forward(x, w, b, off, msk, st, pd, dl, dg)
    np.testing.assert_allclose(grouped.cpu().numpy(), want, rtol=0, atol=2e-5 * max(1.0, float(np.abs(want).max())))
    cl = ops.dcn_v2_forward_nhwc(bo, *args, act=ops.ACT_LRELU, slope=0.1)   # channels-last output + fused lrelu
    np.testing.assert_allclose(cl.cpu().numpy(), np.where(want > 0, want, 0.1 * want), rtol=0,
                               atol=2e-5 * max(1.0, float(np.abs(want).max())))


@pytest.mark.parametrize("shape", [SHAPES[0], SHAPES[1], SHAPES[2], SHAPES[7]])
def test_forward_bf16_mma_matches_bf16_oracle(env, dev, shape):
    """bf16-MFMA variant (fp32 tensors; staged input, weights and blended samples rounded to bf16, fp32 accumulation)
    against the ORACLE evaluated with exactly those roundings (oracle.dcn_v2_forward_bf16).  The only freedom left is the
    fp32 rounding of a column value before it is rounded to bf16 (mask folded into the bilinear weights here, applied
    after the blend there): rare 1-bf16-ulp flips of single column entries.  Measured ~3e-5 relative in the 2-norm, i.e.
    200x below the distance to the fp32 operator (~6e-3), which is also bounded."""
    ops, oracle, synth = env
    B, C, H, W, Co, kh, kw, st, pd, dl, dg = shape
    x, w, b, off, msk = _case(synth, B, C, H, W, Co, kh, kw, st, pd, dl, dg, 300)
    args = [_t(a, dev) for a in (x, w, b, off, msk)]
    got = ops.dcn_v2_forward(*args, st, pd, dl, dg, bf16_mma=True).cpu().numpy()
    want16 = oracle.dcn_v2_forward_bf16(x, w, b, off, msk, st, pd, dl, dg)
    want32 = oracle.dcn_v2_forward(x, w, b, off, msk, st, pd, dl, dg)
    rel16 = float(np.linalg.norm(got - want16) / np.linalg.norm(want16))
    rel32 = float(np.linalg.norm(got - want32) / np.linalg.norm(want32))
    assert rel16 < 3e-4, f"bf16 MFMA forward vs bf16 oracle: {rel16}"
    assert float(np.abs(got - want16).max()) < 2e-3 * max(1.0, float(np.abs(want16).max()))
    assert rel32 < 6e-3, f"bf16 MFMA forward deviates {rel32} (2-norm, relative) from the fp32 oracle"
    assert rel32 > 3 * rel16   # it really ran the reduced-precision kernel and the bf16 oracle really models it
    # integer-valued data (exactly representable in bf16, integer sample positions): identical to the fp32 operator
    xi = torch.round(args[0] * 2).clamp(-8, 8)
    wi = torch.round(args[1] * 64).clamp(-4, 4)
    offi = torch.round(args[3])
    mski = torch.ones_like(args[4])
    a = ops.dcn_v2_forward(xi, wi, args[2], offi, mski, st, pd, dl, dg)
    c = ops.dcn_v2_forward(xi, wi, args[2], offi, mski, st, pd, dl, dg, bf16_mma=True)
    assert torch.equal(a, c)
    np.testing.assert_allclose(a.cpu().numpy(), oracle.dcn_v2_forward(xi.cpu().numpy(), wi.cpu().numpy(), b,
                                                                     offi.cpu().numpy(), mski.cpu().numpy(), st, pd, dl, dg),
                               rtol=0, atol=2e-5 * max(1.0, float(a.abs().max())))


def _dcn_ref64(x, w, b, off, msk, dg):
    """float64 DCNv2 forward (3x3, stride 1, pad 1) from torch primitives: bilinear sampling with zeros outside
    (dcn_v2_im2col_cuda.cu:25-60: corners outside the image contribute nothing) = grid_sample(zeros, align_corners)."""
    x, w, b, off, msk = (a.double() for a in (x, w, b, off, msk))
    B, C, H, W = x.shape
    cpg = C // dg
    ys, xs = torch.meshgrid(torch.arange(H, device=x.device, dtype=torch.float64),
                            torch.arange(W, device=x.device, dtype=torch.float64), indexing="ij")
    cols = torch.zeros((B, C, 9, H, W), dtype=torch.float64, device=x.device)
    for g in range(dg):
        xg = x[:, g * cpg:(g + 1) * cpg]
        for k in range(9):
            py = ys - 1 + k // 3 + off[:, (g * 9 + k) * 2]
            px = xs - 1 + k % 3 + off[:, (g * 9 + k) * 2 + 1]
            grid = torch.stack((2 * px / (W - 1) - 1, 2 * py / (H - 1) - 1), dim=-1)
            smp = F.grid_sample(xg, grid, mode="bilinear", padding_mode="zeros", align_corners=True)
            cols[:, g * cpg:(g + 1) * cpg, k] = smp * msk[:, g * 9 + k][:, None]
    return torch.einsum("ock,bckhw->bohw", w.reshape(w.shape[0], C, 9), cols) + b[None, :, None, None]


@pytest.mark.parametrize("shape", [SHAPES[0], SHAPES[1], SHAPES[2], (2, 32, 13, 15, 64, 3, 3, (1, 1), (1, 1), (1, 1), 2)])
def test_forward_nhwc_f16x2_matches_oracle(env, dev, shape):
    """The f16 x 2 implicit GEMM (fp32 result on the f16 matrix pipe: blended samples and scaled weights in two f16 pieces,
    three products per k step) against the oracle, at the fp32 kernel's tolerance, and no further from a float64 evaluation
    than the fp32-MFMA kernel is (x 1.5); channels-last output + activation; the cache refresh rebuilds its image."""
    ops, oracle, synth = env
    B, C, H, W, Co, kh, kw, st, pd, dl, dg = shape
    x, w, b, off, msk = _case(synth, B, C, H, W, Co, kh, kw, st, pd, dl, dg, 320)
    wt = torch.nn.Parameter(_t(w, dev))
    args = (wt, _t(b, dev), _t(off, dev), _t(msk, dev), dg)
    assert ops.dcn_f16x2_ok(wt, dg)
    bo = ops.BorderedNHWC(_t(x, dev))
    with torch.no_grad():
        f32 = ops.dcn_v2_forward_nhwc(bo, *args, nhwc_out=False, algo="fp32")
        f16 = ops.dcn_v2_forward_nhwc(bo, *args, nhwc_out=False, algo="f16x2")
    want = oracle.dcn_v2_forward(x, w, b, off, msk, st, pd, dl, dg)
    scale = max(1.0, float(np.abs(want).max()))
    np.testing.assert_allclose(f16.cpu().numpy(), want, rtol=0, atol=2e-5 * scale)
    want64 = _dcn_ref64(*[_t(a, dev) for a in (x, w, b, off, msk)], dg)
    assert float((want64 - _t(want, dev).double()).abs().max()) < 2e-5 * scale       # (the float64 restatement is the oracle's)
    e16 = float((f16.double() - want64).abs().max()), float((f16.double() - want64).pow(2).mean().sqrt())
    e32 = float((f32.double() - want64).abs().max()), float((f32.double() - want64).pow(2).mean().sqrt())
    assert e16[1] <= 1.5 * e32[1] and e16[0] <= 2.0 * e32[0], (e16, e32)
    assert not torch.equal(f16, f32)          # (it really ran the other arithmetic)
    with torch.no_grad():
        cl = ops.dcn_v2_forward_nhwc(bo, *args, act=ops.ACT_LRELU, slope=0.1, algo="f16x2")
    np.testing.assert_allclose(cl.cpu().numpy(), np.where(want > 0, want, 0.1 * want), rtol=0, atol=2e-5 * scale)
    if C == 8 * dg:                                     # group-major twin: the same arithmetic
        bo.grouped8 = bo.buf.view(B, H + 3, W + 3, C // 8, 8).permute(0, 3, 1, 2, 4).contiguous()
        with torch.no_grad():
            assert torch.equal(ops.dcn_v2_forward_nhwc(bo, *args, nhwc_out=False, algo="f16x2"), f16)
    # a .data write + refresh rebuilds the f16 x 2 image (scale included: the weights grow 1000-fold)
    wt.data.mul_(1000.0)
    ops.refresh_weight_caches([wt])
    with torch.no_grad():
        big = ops.dcn_v2_forward_nhwc(bo, wt, torch.zeros_like(args[1]), *args[2:], nhwc_out=False, algo="f16x2")
        ref = ops.dcn_v2_forward_nhwc(bo, wt, torch.zeros_like(args[1]), *args[2:], nhwc_out=False, algo="fp32")
    assert float((big - ref).abs().max()) < 2e-5 * max(1.0, float(ref.abs().max()))


def test_forward_nhwc_f16x2_reports_its_domain(env, dev):
    """|mask * sample| >= 65520 leaves the f16 x 2 domain: the outputs are not finite there and the device range flag is set
    (the fused forwards then recompute on the fp32 pipe: ops.f16_range_guard); in-range launches leave the flag alone."""
    ops, oracle, synth = env
    B, C, H, W, Co, kh, kw, st, pd, dl, dg = SHAPES[2]
    x, w, b, off, msk = _case(synth, B, C, H, W, Co, kh, kw, st, pd, dl, dg, 330)
    flag = ops._range_flag(torch.device(dev))
    flag.zero_()
    args = (_t(w, dev), _t(b, dev), _t(off, dev), _t(msk * 0 + 1, dev), dg)
    ok = ops.dcn_v2_forward_nhwc(ops.BorderedNHWC(_t(x, dev)), *args, nhwc_out=False, algo="f16x2")
    assert int(flag.item()) == 0 and bool(torch.isfinite(ok).all())
    xb = x.copy()
    xb[1, 5, 10:14, 8:12] = 3.0e5
    bad = ops.dcn_v2_forward_nhwc(ops.BorderedNHWC(_t(xb, dev)), *args, nhwc_out=False, algo="f16x2")
    assert int(flag.item()) == 1 and not bool(torch.isfinite(bad).all())
    full = ops.dcn_v2_forward_nhwc(ops.BorderedNHWC(_t(xb, dev)), *args, nhwc_out=False, algo="fp32")
    assert bool(torch.isfinite(full).all())
    flag.zero_()


def test_forward_zero_offset_is_conv2d(env, dev):
    ops, _, synth = env
    B, C, H, W, Co, dg = 2, 64, 20, 24, 64, 8
    x, w, b = _t(synth.gaussish((B, C, H, W), 1), dev), _t(synth.gaussish((Co, C, 3, 3), 2) * 0.05, dev), _t(synth.gaussish((Co,), 3), dev)
    got = ops.dcn_v2_forward(x, w, b, torch.zeros(B, 18 * dg, H, W, device=dev), torch.ones(B, 9 * dg, H, W, device=dev), 1, 1, 1, dg)
    want = F.conv2d(x.double(), w.double(), b.double(), padding=1).float()
    assert float((got - want).abs().max()) < 5e-5


@pytest.mark.parametrize("C,H", [(256, 160), (128, 320), (64, 640)])
def test_forward_full_size_properties(env, dev, C, H):
    """The three DynAgg layers at BASELINE config-3 size (one sample of the batch; the oracle would take minutes here):
    size-independent properties instead.  (1) zero offsets + unit mask == conv2d; (2) an integer offset field is the
    conv2d of the translated, zero-filled input (compared off the 1-pixel frame, where the conv's own padding taps --
    unlike the deformed ones -- never land inside the image); (3) the output is linear in the mask; (4) fractional
    offsets: the operator equals the bilinear mix of its four integer-offset neighbours."""
    ops, _, synth = env
    dg = 8
    g = torch.Generator(device=dev).manual_seed(7)
    x = torch.randn((1, C, H, H), generator=g, device=dev)
    w = torch.randn((C, C, 3, 3), generator=g, device=dev) * (1.0 / np.sqrt(9 * C))
    b = torch.randn((C,), generator=g, device=dev)
    zeros = torch.zeros((1, 18 * dg, H, H), device=dev)
    ones = torch.ones((1, 9 * dg, H, H), device=dev)
    ref = F.conv2d(x, w, b, padding=1)
    tol = 2e-4 * float(ref.abs().max())
    assert float((ops.dcn_v2_forward(x, w, b, zeros, ones, 1, 1, 1, dg) - ref).abs().max()) < tol

    def shifted(dy, dx):   # conv2d of the input translated by (dy, dx) with zero fill == every tap displaced by (dy, dx)
        xs = torch.zeros_like(x)
        ys, yd = (slice(dy, None), slice(0, H - dy)) if dy >= 0 else (slice(0, H + dy), slice(-dy, None))
        xs_, xd = (slice(dx, None), slice(0, H - dx)) if dx >= 0 else (slice(0, H + dx), slice(-dx, None))
        xs[:, :, yd, xd] = x[:, :, ys, xs_]
        return F.conv2d(xs, w, None, padding=1)

    def field(dy, dx):
        off = torch.empty_like(zeros)
        off[:, 0::2] = dy
        off[:, 1::2] = dx
        return off

    zb = torch.zeros_like(b)
    inner = (slice(None), slice(None), slice(1, -1), slice(1, -1))
    got = ops.dcn_v2_forward(x, w, zb, field(3.0, -5.0), ones, 1, 1, 1, dg)
    assert float((got - shifted(3, -5))[inner].abs().max()) < tol
    half = ops.dcn_v2_forward(x, w, zb, field(3.0, -5.0), 0.5 * ones, 1, 1, 1, dg)
    assert float((half - 0.5 * got).abs().max()) < tol
    frac = ops.dcn_v2_forward(x, w, zb, field(2.25, -4.5), ones, 1, 1, 1, dg)
    mix = 0.75 * 0.5 * (shifted(2, -5) + shifted(2, -4)) + 0.25 * 0.5 * (shifted(3, -5) + shifted(3, -4))
    assert float((frac - mix)[inner].abs().max()) < tol


@pytest.mark.parametrize("C,H", [(256, 160), (128, 320), (64, 640)])
def test_backward_full_size_finite_differences(env, dev, C, H):
    """grad_offset / grad_mask of the three DynAgg layers at config-3 size against central differences of the forward.
    An offset or mask element at pixel (y, x) only moves out[:, y, x], so the directional derivative of
    L = sum(grad_out * out) is a 'C'-term sum: fp32 is precise enough.  All sample points sit at half-integer positions,
    where the operator is exactly linear within +-0.25 (no kink inside the difference stencil)."""
    ops, _, synth = env
    dg, eps = 8, 0.25
    g = torch.Generator(device=dev).manual_seed(11)
    x = torch.randn((1, C, H, H), generator=g, device=dev)
    w = torch.randn((C, C, 3, 3), generator=g, device=dev) * (1.0 / np.sqrt(9 * C))
    b = torch.zeros((C,), device=dev)
    off = torch.randint(-6, 7, (1, 18 * dg, H, H), generator=g, device=dev).float() + 0.5
    msk = torch.rand((1, 9 * dg, H, H), generator=g, device=dev)
    go = torch.randn((1, C, H, H), generator=g, device=dev)
    _, g_off, g_msk, _, _ = ops.dcn_v2_backward(x, w, b, off, msk, go, 1, 1, 1, dg, need_input_grad=False)

    def local_loss(o_, m_, y, xx):
        out = ops.dcn_v2_forward(x, w, b, o_, m_, 1, 1, 1, dg)
        return float((out[0, :, y, xx].double() * go[0, :, y, xx].double()).sum())

    rs = np.random.RandomState(3)
    for _ in range(4):
        ch, y, xx = int(rs.randint(18 * dg)), int(rs.randint(8, H - 8)), int(rs.randint(8, H - 8))
        op, om = off.clone(), off.clone()
        op[0, ch, y, xx] += eps
        om[0, ch, y, xx] -= eps
        fd = (local_loss(op, msk, y, xx) - local_loss(om, msk, y, xx)) / (2 * eps)
        an = float(g_off[0, ch, y, xx])
        assert abs(fd - an) <= 2e-3 * max(1.0, abs(fd)), f"grad_offset[{ch},{y},{xx}]: analytic {an} vs finite difference {fd}"
    for _ in range(2):
        ch, y, xx = int(rs.randint(9 * dg)), int(rs.randint(H)), int(rs.randint(H))
        mp, mm = msk.clone(), msk.clone()
        mp[0, ch, y, xx] += eps
        mm[0, ch, y, xx] -= eps
        fd = (local_loss(off, mp, y, xx) - local_loss(off, mm, y, xx)) / (2 * eps)
        an = float(g_msk[0, ch, y, xx])
        assert abs(fd - an) <= 2e-3 * max(1.0, abs(fd)), f"grad_mask[{ch},{y},{xx}]: analytic {an} vs finite difference {fd}"


@pytest.mark.parametrize("C,H", [(256, 160), (128, 320), (64, 640)])
def test_backward_full_size_weight_grad_is_conv2d_grad(env, dev, C, H):
    """Zero offsets + unit mask: grad_weight / grad_bias (and grad_input, atomics path) of the config-3 layers must be the
    gradients of a plain 3x3 convolution (torch autograd on the same device, fp32)."""
    ops, _, synth = env
    dg = 8
    g = torch.Generator(device=dev).manual_seed(13)
    x = torch.randn((1, C, H, H), generator=g, device=dev, requires_grad=True)
    w = (torch.randn((C, C, 3, 3), generator=g, device=dev) * (1.0 / np.sqrt(9 * C))).requires_grad_(True)
    b = torch.zeros((C,), device=dev, requires_grad=True)
    go = torch.randn((1, C, H, H), generator=g, device=dev) * (1.0 / H)
    gx_ref, gw_ref, gb_ref = torch.autograd.grad(F.conv2d(x, w, b, padding=1), (x, w, b), go)
    zeros = torch.zeros((1, 18 * dg, H, H), device=dev)
    ones = torch.ones((1, 9 * dg, H, H), device=dev)
    gx, _, _, gw, gb = ops.dcn_v2_backward(x.detach(), w.detach(), b.detach(), zeros, ones, go, 1, 1, 1, dg)
    for name, a, r in (("grad_weight", gw, gw_ref), ("grad_bias", gb, gb_ref), ("grad_input", gx, gx_ref)):
        tol = 5e-4 * max(1e-3, float(r.abs().max()))
        assert float((a - r).abs().max()) <= tol, f"{name}: {float((a - r).abs().max())} > {tol}"
    _, _, _, gw2, gb2 = ops.dcn_v2_backward(x.detach(), w.detach(), b.detach(), zeros, ones, go, 1, 1, 1, dg, need_input_grad=False)
    assert float((gw2 - gw_ref).abs().max()) <= 5e-4 * max(1e-3, float(gw_ref.abs().max()))


@pytest.mark.parametrize("shape", [s for s in SHAPES if (s[1] // s[10]) % 4 == 0])
def test_backward_matches_oracle(env, dev, shape):
    ops, oracle, synth = env
    B, C, H, W, Co, kh, kw, st, pd, dl, dg = shape
    x, w, b, off, msk = _case(synth, B, C, H, W, Co, kh, kw, st, pd, dl, dg, 400, off_scale=2.0)
    out_shape = oracle.dcn_v2_forward(x, w, b, off, msk, st, pd, dl, dg).shape
    go = synth.gaussish(out_shape, 410)
    got = ops.dcn_v2_backward(_t(x, dev), _t(w, dev), _t(b, dev), _t(off, dev), _t(msk, dev), _t(go, dev), st, pd, dl, dg)
    want = oracle.dcn_v2_backward(x, w, b, off, msk, go, st, pd, dl, dg)
    for name, g_, w_ in zip(("grad_input", "grad_offset", "grad_mask", "grad_weight", "grad_bias"), got, want):
        g_ = g_.cpu().numpy()
        assert g_.shape == w_.shape, name
        tol = 1e-4 * max(1.0, float(np.abs(w_).max()))
        assert float(np.abs(g_ - w_).max()) <= tol, f"{name}: max err {np.abs(g_ - w_).max()} > {tol}"


def test_backward_is_overwriting_not_accumulating(env, dev):
    ops, _, synth = env
    B, C, H, W, Co, dg = 1, 32, 8, 9, 32, 8
    x, w, b, off, msk = _case(synth, B, C, H, W, Co, 3, 3, (1, 1), (1, 1), (1, 1), dg, 500)
    args = [_t(a, dev) for a in (x, w, b, off, msk)] + [_t(synth.gaussish((B, Co, H, W), 501), dev)]
    g1 = ops.dcn_v2_backward(*args, 1, 1, 1, dg)
    g2 = ops.dcn_v2_backward(*args, 1, 1, 1, dg)
    for a, b_ in zip(g1, g2):
        assert float((a - b_).abs().max()) <= 1e-5 * max(1.0, float(a.abs().max()))


@pytest.mark.parametrize("cfg", [
    # B, C, H, W, Co, stride, dg      -> channels per group
    (2, 64, 11, 13, 64, (1, 1), 8),    # 8: two whole groups per lane
    (2, 128, 9, 10, 128, (1, 1), 8),   # 16: one whole group per lane
    (1, 256, 7, 9, 256, (1, 1), 8),    # 32: half a group per lane, cross-half add
    (2, 64, 9, 10, 96, (2, 2), 2),     # 32, stride 2, Co = 96
    (1, 48, 8, 9, 24, (1, 1), 6),      # C % 32 != 0: stays on the atomic kernel
])
def test_backward_without_input_grad(env, dev, cfg):
    """grad_input is optional at the C-ABI (NULL pointer).  The other four gradients must not change -- although
    grad_offset / grad_mask then come from a different kernel (dcn_bwd_offmask_kernel: channels-last gathers, plain
    stores instead of atomics) -- and must still match the oracle."""
    ops, oracle, synth = env
    B, C, H, W, Co, st, dg = cfg
    x, w, b, off, msk = _case(synth, B, C, H, W, Co, 3, 3, st, (1, 1), (1, 1), dg, 520)
    Ho, Wo = off.shape[2:]
    go = synth.gaussish((B, Co, Ho, Wo), 521)
    args = [_t(a, dev) for a in (x, w, b, off, msk, go)]
    full = ops.dcn_v2_backward(*args, st, 1, 1, dg)
    # the outputs are torch.empty() blocks (recycled memory): the offset/mask kernel must write every element itself
    part = ops.dcn_v2_backward(*args, st, 1, 1, dg, need_input_grad=False)
    assert part[0] is None
    for a, b_ in zip(full[1:], part[1:]):
        assert float((a - b_).abs().max()) <= 1e-5 * max(1.0, float(a.abs().max()))
    want = oracle.dcn_v2_backward(x, w, b, off, msk, go, st, (1, 1), (1, 1), dg)
    for name, g_, w_ in zip(("grad_offset", "grad_mask"), part[1:3], want[1:3]):
        g_ = g_.cpu().numpy()
        tol = 1e-4 * max(1.0, float(np.abs(w_).max()))
        assert float(np.abs(g_ - w_).max()) <= tol, f"{name}: max err {np.abs(g_ - w_).max()} > {tol}"


def test_fuse_offsets_matches_reference_formula(env, dev):
    """dcn_v2.py:229-245 written with the reference's own tensor ops."""
    ops, _, synth = env
    B, dg, K, H, W = 2, 8, 9, 11, 13
    raw = _t(synth.gaussish((B, 3 * dg * K, H, W), 600), dev)
    pre = _t(np.round(synth.gaussish((B, K, H, W, 2), 601) * 5), dev)
    abs_sum = torch.zeros(256, dtype=torch.float64, device=dev)
    offset, mask = ops.dcn_fuse_offsets(raw, pre, dg, K, abs_sum)
    o1, o2, m = torch.chunk(raw, 3, dim=1)
    want_off = torch.cat((o1, o2), dim=1)
    rep = pre.repeat([1, dg, 1, 1, 1])
    reorder = torch.zeros_like(want_off)
    reorder[:, 0::2] = rep[..., 1]
    reorder[:, 1::2] = rep[..., 0]
    assert torch.equal(offset, want_off + reorder)
    assert float((mask - torch.sigmoid(m)).abs().max()) < 1e-6
    assert abs(float(abs_sum.sum()) / want_off.numel() - float(want_off.abs().mean())) < 1e-5
    off2, mask2 = ops.dcn_fuse_offsets(raw, None, dg, K)
    assert torch.equal(off2, want_off) and torch.equal(mask2, mask)


def test_python_operator_forward_backward(env, dev):
    """DCN_sep_pre_multi_offset through autograd == fp64 autograd of the independent torch restatement."""
    ops, oracle, synth = env
    import torch_port
    from mmsr.models.archs.DCNv2.dcn_v2 import DCN_sep_pre_multi_offset
    B, C, H, W, dg = 2, 32, 10, 12, 8
    layer = DCN_sep_pre_multi_offset(C, C, 3, stride=1, padding=1, dilation=1, deformable_groups=dg, extra_offset_mask=True).to(dev)
    assert float(layer.conv_offset_mask.weight.abs().max()) == 0.0  # zero-initialised head (dcn_v2.py:218-220)
    with torch.no_grad():
        layer.conv_offset_mask.weight.copy_(_t(synth.gaussish(tuple(layer.conv_offset_mask.weight.shape), 700) * 0.05, dev))
        layer.conv_offset_mask.bias.copy_(_t(synth.gaussish((216,), 701) * 0.1, dev))
    x = _t(synth.gaussish((B, C, H, W), 702), dev).requires_grad_()
    feat = _t(synth.gaussish((B, C, H, W), 703), dev).requires_grad_()
    pre = _t(np.round(synth.gaussish((B, 9, H, W, 2), 704) * 3), dev)
    out = layer([x, feat], pre)
    go = _t(synth.gaussish(tuple(out.shape), 705), dev)
    out.backward(go)

    # independent fp64 CPU restatement with stock torch ops
    xd, fd = x.detach().cpu().double().requires_grad_(), feat.detach().cpu().double().requires_grad_()
    wd, bd = layer.weight.detach().cpu().double().requires_grad_(), layer.bias.detach().cpu().double().requires_grad_()
    cw, cb = layer.conv_offset_mask.weight.detach().cpu().double().requires_grad_(), layer.conv_offset_mask.bias.detach().cpu().double().requires_grad_()
    raw = F.conv2d(fd, cw, cb, padding=1)
    o1, o2, m = torch.chunk(raw, 3, dim=1)
    off = torch.cat((o1, o2), dim=1)
    rep = pre.cpu().double().repeat([1, dg, 1, 1, 1])
    reorder = torch.zeros_like(off)
    reorder[:, 0::2] = rep[..., 1]
    reorder[:, 1::2] = rep[..., 0]
    ref = torch_port.dcn_v2_reference(xd, wd, bd, off + reorder, torch.sigmoid(m), dg=dg)
    ref.backward(go.cpu().double())
    assert float((out.detach().cpu().double() - ref.detach()).abs().max()) < 1e-4
    for name, got, want in (("x", x.grad, xd.grad), ("feat", feat.grad, fd.grad), ("weight", layer.weight.grad, wd.grad),
                            ("bias", layer.bias.grad, bd.grad), ("com.weight", layer.conv_offset_mask.weight.grad, cw.grad),
                            ("com.bias", layer.conv_offset_mask.bias.grad, cb.grad)):
        err = float((got.cpu().double() - want).abs().max())
        assert err <= 2e-4 * max(1.0, float(want.abs().max())), f"{name}: {err}"


def test_ext_module_contract(env, dev):
    import _ext
    x = torch.zeros(1, 4, 5, 5, device=dev)
    w = torch.zeros(2, 4, 3, 3, device=dev)
    with pytest.raises(RuntimeError):
        _ext.dcn_v2_forward(x.cpu(), w.cpu(), torch.zeros(2), torch.zeros(1, 18, 5, 5), torch.zeros(1, 9, 5, 5), 3, 3, 1, 1, 1, 1, 1, 1, 1)
    with pytest.raises(RuntimeError):  # kernel size mismatch (dcn_v2_cuda.cu:79-80)
        _ext.dcn_v2_forward(x, w, torch.zeros(2, device=dev), torch.zeros(1, 18, 5, 5, device=dev), torch.zeros(1, 9, 5, 5, device=dev), 5, 5, 1, 1, 1, 1, 1, 1, 1)
    with pytest.raises(NotImplementedError):
        _ext.dcn_v2_psroi_pooling_forward()
    out = _ext.dcn_v2_forward(x, w, torch.ones(2, device=dev), torch.zeros(1, 18, 5, 5, device=dev), torch.ones(1, 9, 5, 5, device=dev), 3, 3, 1, 1, 1, 1, 1, 1, 1)
    assert tuple(out.shape) == (1, 2, 5, 5) and float((out - 1).abs().max()) == 0.0
