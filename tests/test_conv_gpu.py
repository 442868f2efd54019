"""GPU parity tests of the fused channels-last 3x3 convolution (csrc/conv3x3.hip) through the C-ABI.

The operator is floating point; fp32 MFMA is an exact fmaf chain, so the only freedom against a reference convolution is
the summation order: tolerance 1e-5 * scale against F.conv2d evaluated in float64 (what VERDICT r1 item 7 asks for)."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ops(dev):
    import c2m_amd
    return c2m_amd.ops


def _cl(t):
    return t.contiguous(memory_format=torch.channels_last)


def _rand(shape, dev, seed, scale=1.0):
    g = torch.Generator(device=dev).manual_seed(seed)
    return torch.randn(shape, generator=g, device=dev) * scale


def _ref(xs, w, b, act, slope, res):
    y = F.conv2d(torch.cat([x.double() for x in xs], 1), w.double(), None if b is None else b.double(), padding=1)
    if act == 1:
        y = y.clamp_min(0)
    elif act == 2:
        y = torch.where(y > 0, y, y * slope)
    for r in res:
        y = y + r.double()
    return y


CASES = [
    # B, [Cin per source], Cout, H, W, act, n residuals
    (2, [64], 64, 40, 40, 1, 0),        # residual-block conv1 (arch_util.py:131) at configs[0] size: ragged x tiles (40 = 32 + 8)
    (2, [64], 64, 40, 40, 0, 1),        # conv2 + identity
    (1, [64], 64, 37, 45, 0, 2),        # odd sizes, two residuals (last block of a body + the stage skip)
    (2, [64, 256], 256, 20, 24, 2, 0),  # small_offset_conv1: cat(content, ref) -> 256, LeakyReLU (ref_restoration_arch.py:147-149)
    (1, [64, 128], 64, 16, 64, 2, 0),   # head_medium
    (1, [32], 3, 33, 31, 0, 0),         # tail_large.2: 32 -> 3
    (1, [64], 32, 36, 32, 2, 0),        # tail_large.0: 64 -> 32
    (1, [128], 216, 12, 40, 0, 0),      # a 216-channel head as a plain conv (Cout not a multiple of 64)
    (3, [320], 64, 9, 5, 2, 0),         # tiny map, 10 chunks from a single source
]


@pytest.mark.parametrize("case", CASES)
def test_conv3x3_matches_fp64_conv2d(ops, dev, case):
    B, cins, Cout, H, W, act, nres = case
    xs = [_cl(_rand((B, c, H, W), dev, 10 + k)) for k, c in enumerate(cins)]
    w = _rand((Cout, sum(cins), 3, 3), dev, 20, 1.0 / np.sqrt(9 * sum(cins)))
    b = _rand((Cout,), dev, 21)
    res = [_cl(_rand((B, Cout, H, W), dev, 30 + k)) for k in range(nres)]
    got = ops.conv3x3(xs, w, b, act=act, slope=0.1, res1=res[0] if nres > 0 else None, res2=res[1] if nres > 1 else None)
    want = _ref(xs, w, b, act, 0.1, res)
    assert got.shape == want.shape and got.is_contiguous(memory_format=torch.channels_last)
    err = float((got.double() - want).abs().max())
    assert err < 1e-5 * max(1.0, float(want.abs().max())), err


def test_conv3x3_source_may_be_a_channel_slice_and_output_a_view(ops, dev):
    """Sources are described by pitches: a channel slice of a wider channels-last tensor is read in place."""
    big = _cl(_rand((2, 128, 24, 40), dev, 1))
    x = big[:, 32:96]
    w, b = _rand((64, 64, 3, 3), dev, 2, 0.05), _rand((64,), dev, 3)
    got = ops.conv3x3(x, w, b)
    want = _ref([x], w, b, 0, 0.1, [])
    assert float((got.double() - want).abs().max()) < 1e-5 * float(want.abs().max())


def test_conv3x3_pixel_shuffle_and_nchw_outputs(ops, dev):
    x = _cl(_rand((2, 64, 24, 40), dev, 4))
    w, b = _rand((256, 64, 3, 3), dev, 5, 0.05), _rand((256,), dev, 6)
    got = ops.conv3x3(x, w, b, act=ops.ACT_LRELU, slope=0.1, out_mode="pixel_shuffle")
    want = F.leaky_relu(F.pixel_shuffle(F.conv2d(x.double(), w.double(), b.double(), padding=1), 2), 0.1)   # tail_small (:155-157)
    assert tuple(got.shape) == (2, 64, 48, 80)
    assert float((got.double() - want).abs().max()) < 1e-5 * float(want.abs().max())
    w3, b3 = _rand((3, 64, 3, 3), dev, 7, 0.05), _rand((3,), dev, 8)
    got = ops.conv3x3(x, w3, b3, out_mode="nchw")
    want = F.conv2d(x.double(), w3.double(), b3.double(), padding=1)
    assert got.is_contiguous() and float((got.double() - want).abs().max()) < 1e-5 * float(want.abs().max())


def test_conv3x3_full_size_body_conv(ops, dev):
    """BASELINE configs[2] shape of the dominant layer: 64 -> 64 at 640x640 (two samples), ReLU."""
    x = _cl(_rand((2, 64, 640, 640), dev, 40))
    w, b = _rand((64, 64, 3, 3), dev, 41, 0.04), _rand((64,), dev, 42)
    got = ops.conv3x3(x, w, b, act=ops.ACT_RELU)
    want = F.conv2d(x, w, b, padding=1).relu()     # stock fp32 conv (MIOpen may use Winograd: looser)
    assert float((got - want).abs().max()) < 2e-4 * float(want.abs().max())
    sub = (slice(0, 1), slice(None), slice(300, 340), slice(600, 640))
    want64 = F.conv2d(x[:1, :, 299:341, 599:640].double(), w.double(), b.double(), padding=1).relu()[:, :, 1:-1, 1:]
    assert float((got[sub].double() - want64).abs().max()) < 1e-5 * float(want64.abs().max())


def test_weight_cache_follows_the_version_counter(ops, dev):
    x = _cl(_rand((1, 32, 8, 8), dev, 50))
    w = torch.nn.Parameter(_rand((32, 32, 3, 3), dev, 51, 0.1))
    a = ops.conv3x3(x, w)
    with torch.no_grad():
        w.mul_(2.0)           # in-place update = optimiser step: the cached re-layout must not be reused
    b2 = ops.conv3x3(x, w)
    assert float((b2 - 2 * a).abs().max()) < 1e-5 * float(a.abs().max())


def test_dcn_head_matches_oracle_assembly(ops, dev):
    """out_mode DCN_HEAD against the reference formula dcn_v2.py:229-245 with pre-offsets built by the oracle from the
    same index map (corres_generation_arch.py:69-109): offsets/mask within conv tolerance, pre-offset part exact."""
    import c2m_oracle as oracle
    import synth
    B, C, dg = 2, 64, 8
    # w = 14: every scale on the direct kernel; w = 16: 32- / 64-pixel-wide maps at scales 2 / 4 take the Winograd kernel for the
    # 192-channel slice of the head (+ the direct kernel for the remaining 24 channels)
    for (h, w, s) in ((12, 14, 1), (12, 14, 2), (12, 14, 4), (12, 16, 1), (12, 16, 2), (12, 16, 4)):
        H, W = h * s, w * s
        feat = _cl(_rand((B, C, H, W), dev, 60 + s))
        wt, bs = _rand((3 * dg * 9, C, 3, 3), dev, 61, 0.02), _rand((3 * dg * 9,), dev, 62, 0.1)
        hp, wp = h - 2, w - 2
        idx = (synth.uniform((B, hp, wp), 63, 0.0, 1.0).astype(np.float64) * (hp * wp)).astype(np.int64) % (hp * wp)
        flow = ops.index_to_flow(torch.from_numpy(idx).to(dev))
        abs_sum = torch.zeros(256, dtype=torch.float64, device=dev)
        off, msk = ops.conv3x3_dcn_head(feat, wt, bs, dg, flow, s, abs_sum)
        raw = F.conv2d(feat.double(), wt.double(), bs.double(), padding=1)
        o1, o2, m = torch.chunk(raw, 3, dim=1)
        want_off = torch.cat((o1, o2), 1)
        pre = np.stack([oracle.build_pre_offsets(idx[b], h, w)[{1: 0, 2: 1, 4: 2}[s]] for b in range(B)])   # [B,9,H,W,2]
        pre_t = torch.from_numpy(pre).to(dev).double().flip(-1).permute(0, 1, 4, 2, 3).reshape(B, 18, H, W).repeat(1, dg, 1, 1)
        want_abs = float(want_off.abs().sum())
        want_off = want_off + pre_t
        tol = 2e-5 * max(1.0, float(raw.abs().max()))
        assert float((off.double() - want_off).abs().max()) < tol
        assert float((msk.double() - torch.sigmoid(m)).abs().max()) < 1e-6
        assert abs(float(abs_sum.sum()) - want_abs) < 1e-4 * want_abs
    off0, _ = ops.conv3x3_dcn_head(feat, wt, bs, dg, None, 4)
    assert float((off0.double() - torch.cat((o1, o2), 1)).abs().max()) < tol


def test_vgg_and_extractor_stacks_match_stock_torch(dev):
    """VGGFeatureExtractor / ContrasExtractorSep run their conv+ReLU stacks on the fused channels-last kernel under no_grad;
    with gradients enabled they run module by module on stock torch.  Same weights, same image: the taps agree to fp32
    rounding; tapped activations are interior views of zero-bordered channels-last buffers."""
    import warnings
    import c2m_amd
    from mmsr.models.archs.contras_extractor_arch import ContrasExtractorSep
    from mmsr.models.archs.vgg_arch import VGGFeatureExtractor
    with warnings.catch_warnings():
        warnings.simplefilter("ignore", RuntimeWarning)
        vgg = VGGFeatureExtractor(["relu1_1", "relu2_1", "relu3_1"], "vgg19").to(dev).eval()
        ext = ContrasExtractorSep().to(dev).eval()
    torch.manual_seed(3)
    for m in list(vgg.modules()) + list(ext.modules()):
        if isinstance(m, torch.nn.Conv2d):
            torch.nn.init.kaiming_normal_(m.weight)
            torch.nn.init.normal_(m.bias, std=0.05)
    img = torch.rand(2, 3, 72, 88, device=dev)
    img2 = torch.rand(2, 3, 72, 88, device=dev)
    with torch.enable_grad():
        want = vgg(img)
        want_e = ext(img, img2)
    with torch.no_grad():
        assert vgg._use_fused(img)
        got = vgg(img)
        got_e = ext(img, img2)
    for k in ("relu1_1", "relu2_1", "relu3_1"):
        assert got[k].shape == want[k].shape
        err = float((got[k] - want[k]).abs().max())
        assert err < 2e-4 * max(1.0, float(want[k].abs().max())), (k, err)   # stock path may use Winograd convolutions
        bo = c2m_amd.ops.bordered_of(got[k])
        assert bo is not None and c2m_amd.ops.BorderedNHWC(got[k]) is bo
        H, W = got[k].shape[2:]
        assert float(bo.buf[:, 0].abs().max()) == 0 and float(bo.buf[:, H + 1:].abs().max()) == 0
        assert float(bo.buf[:, :, 0].abs().max()) == 0 and float(bo.buf[:, :, W + 1:].abs().max()) == 0
        assert bo.grouped8 is None   # a bare extractor asks for no group-major twin
    # CorrespondenceGenerationArch's extractor also writes relu1_1 in the 8-channel group-major layout the large DynAgg
    # gathers from: the same values, re-arranged, with the same zero border
    vgg.grouped8_taps = ("relu1_1",)
    with torch.no_grad():
        got2 = vgg(img)
    bo = c2m_amd.ops.bordered_of(got2["relu1_1"])
    Bc, Hp, Wp, Cc = bo.buf.shape
    assert torch.equal(got2["relu1_1"], got["relu1_1"])
    assert torch.equal(bo.grouped8, bo.buf.view(Bc, Hp, Wp, Cc // 8, 8).permute(0, 3, 1, 2, 4).contiguous())
    assert c2m_amd.ops.bordered_of(got2["relu2_1"]).grouped8 is None
    for k in ("dense_features1", "dense_features2"):
        assert got_e[k].is_contiguous() and got_e[k].shape == want_e[k].shape
        err = float((got_e[k] - want_e[k].detach()).abs().max())
        assert err < 2e-4 * max(1.0, float(want_e[k].abs().max())), (k, err)


@pytest.mark.parametrize("B,C,Co,H,W", [(2, 64, 64, 12, 64), (1, 128, 128, 22, 96), (1, 64, 128, 8, 32)])
def test_conv3x3_fused_maxpool(ops, dev, B, C, Co, H, W):
    """conv + ReLU + MaxPool2d(2, 2) in the Winograd F(2,3) kernel's epilogue (the conv1_2 / conv2_2 -> pool steps of the VGG
    towers): identical to pooling the un-fused kernel's output (max is exact), and within conv tolerance of float64."""
    x = _cl(_rand((B, C, H, W), dev, 70))
    w = _rand((Co, C, 3, 3), dev, 71, 1.0 / np.sqrt(9 * C))
    b = _rand((Co,), dev, 72)
    got = ops.conv3x3(x, w, b, act=ops.ACT_RELU, out_mode="nhwc_pool2", algo="winograd")
    plain = ops.conv3x3(x, w, b, act=ops.ACT_RELU, algo="winograd")
    assert got.shape == (B, Co, H // 2, W // 2) and got.is_contiguous(memory_format=torch.channels_last)
    assert torch.equal(got, F.max_pool2d(plain, 2, 2))
    want = F.max_pool2d(_ref([x], w, b, 1, 0.0, []), 2, 2)
    assert float((got.double() - want).abs().max()) < 2e-5 * max(1.0, float(want.abs().max()))


WINO4_CASES = [
    # B, [Cin per source], Cout, H, W, act, n residuals     (W % 64 == 0, Cout % 64 == 0, channels % 16 == 0)
    (2, [64], 64, 12, 64, 1, 0),
    (1, [64], 64, 9, 128, 0, 2),         # ragged tile rows, two residuals
    (2, [64, 64], 64, 8, 64, 2, 0),      # two sources (head_large: cat(x, swapped))
    (1, [96], 128, 6, 64, 0, 1),         # 6 chunks of 16, two cout blocks
    (1, [64], 64, 40, 320, 1, 1),        # several tiles per workgroup stream
    (1, [64, 128], 128, 19, 192, 2, 0),  # medium_offset_conv1 geometry
]


@pytest.mark.parametrize("case", WINO4_CASES)
def test_conv3x3_winograd_f43_matches_fp64(ops, dev, case):
    """Winograd F(4,3)-along-x kernel (what the decoder's convolutions take on 64-pixel-tileable maps, ops.conv3x3(fast=True))
    against float64 conv2d: the transforms cost ~4x the rounding error of the direct kernel -> 2e-5 * scale, as for F(2,3)."""
    B, cins, Cout, H, W, act, nres = case
    xs = [_cl(_rand((B, c, H, W), dev, 10 + k)) for k, c in enumerate(cins)]
    w = _rand((Cout, sum(cins), 3, 3), dev, 20, 1.0 / np.sqrt(9 * sum(cins)))
    b = _rand((Cout,), dev, 21)
    res = [_cl(_rand((B, Cout, H, W), dev, 30 + k)) for k in range(nres)]
    kw = dict(act=act, slope=0.1, res1=res[0] if nres > 0 else None, res2=res[1] if nres > 1 else None)
    got = ops.conv3x3(xs, w, b, algo="winograd4", **kw)
    want = _ref(xs, w, b, act, 0.1, res)
    assert got.shape == want.shape and got.is_contiguous(memory_format=torch.channels_last)
    tol = 2e-5 * max(1.0, float(want.abs().max()))
    assert float((got.double() - want).abs().max()) < tol
    plain = ops.conv3x3(xs, w, b, **kw)                    # the automatic choice without fast=True: the F(2,3) kernel
    assert float((plain.double() - want).abs().max()) < tol


@pytest.mark.parametrize("B,H,W,act,norm", [(2, 40, 40, 1, True), (1, 37, 75, 2, False), (3, 8, 64, 0, True),
                                             (1, 131, 200, 1, True), (2, 5, 3, 1, False)])
def test_conv3x3_rgb64_first_layer(ops, dev, B, H, W, act, norm):
    """The 3 -> 64 first-layer kernel (vgg conv1_1 / conv_first): (image - mean) / std, zero padding in the normalised
    domain, conv + bias + activation, against float64 conv2d; plus the group-major twin and a bordered destination."""
    img = torch.rand((B, 3, H, W), generator=torch.Generator(device=dev).manual_seed(5), device=dev)
    w = _rand((64, 3, 3, 3), dev, 6, 1.0 / np.sqrt(27))
    b = _rand((64,), dev, 7)
    mean = torch.tensor([0.485, 0.456, 0.406], device=dev).view(1, 3, 1, 1)
    std = torch.tensor([0.229, 0.224, 0.225], device=dev).view(1, 3, 1, 1)
    xin = ((img - mean) / std) if norm else img
    want = _ref([xin], w, b, act, 0.1, [])
    got = ops.conv3x3_rgb64(img, w, b, act=act, slope=0.1, mean=mean if norm else None, std=std if norm else None)
    assert got.shape == want.shape and got.is_contiguous(memory_format=torch.channels_last)
    tol = 1e-5 * max(1.0, float(want.abs().max()))
    assert float((got.double() - want).abs().max()) < tol
    bo = ops._bordered_empty(B, 64, H, W, dev, grouped8=True)
    view = bo.interior()
    ops.conv3x3_rgb64(img, w, b, act=act, slope=0.1, mean=mean if norm else None, std=std if norm else None, out=view,
                      out2_grouped8=bo.grouped8)
    assert torch.equal(view, got)
    assert torch.equal(bo.grouped8, bo.buf.view(B, H + 3, W + 3, 8, 8).permute(0, 3, 1, 2, 4).contiguous())
    assert float(bo.buf[:, 0].abs().max()) == 0 and float(bo.buf[:, :, W + 1:].abs().max()) == 0


def test_conv3x3_rgb64_first_layer_at_full_size(ops, dev):
    """The first-layer kernel on a chip-filling launch (configs[2]'s own shape: B=16 at 640 x 640, 12 800 tiles on 768
    persistent workgroups, three per CU), every image against a stock convolution, repeated, plus the bordered destination
    with its group-major twin: the store-data hazard this kernel's round-5 rewrite ran into (128-bit buffer stores with a
    register soffset: lanes 12..15 / 28..31 wrong by O(1)) and anything of its class shows here, not on the
    small maps above."""
    B, H, W = 16, 640, 640
    img = torch.rand((B, 3, H, W), generator=torch.Generator(device=dev).manual_seed(15), device=dev)
    w = _rand((64, 3, 3, 3), dev, 16, 1.0 / np.sqrt(27))
    b = _rand((64,), dev, 17)
    mean = torch.tensor([0.485, 0.456, 0.406], device=dev).view(1, 3, 1, 1)
    std = torch.tensor([0.229, 0.224, 0.225], device=dev).view(1, 3, 1, 1)
    want = torch.cat([F.conv2d((img[i:i + 1] - mean) / std, w, b, padding=1).relu() for i in range(B)])
    tol = 1e-5 * max(1.0, float(want.abs().max()))
    first = None
    for rep in range(3):
        got = ops.conv3x3_rgb64(img, w, b, act=ops.ACT_RELU, mean=mean, std=std)
        assert float((got - want).abs().max()) < tol, rep
        first = got if first is None else first
        assert torch.equal(got, first), rep
    bo = ops._bordered_empty(B, 64, H, W, dev, grouped8=True)
    ops.conv3x3_rgb64(img, w, b, act=ops.ACT_RELU, mean=mean, std=std, out=bo.interior(), out2_grouped8=bo.grouped8)
    assert torch.equal(bo.interior(), first)
    assert torch.equal(bo.grouped8, bo.buf.view(B, H + 3, W + 3, 8, 8).permute(0, 3, 1, 2, 4).contiguous())


WINO_CASES = [
    # B, [Cin per source], Cout, H, W, act, n residuals     (W % 32 == 0, Cout % 64 == 0, channels % 16 == 0)
    (2, [64], 64, 12, 64, 1, 0),
    (1, [64], 64, 9, 128, 0, 2),
    (2, [64, 64], 64, 8, 64, 2, 0),      # two sources (head_large: cat(x, swapped))
    (1, [96], 128, 6, 64, 0, 1),         # 6 chunks of 16, two cout blocks
    (1, [64], 64, 40, 320, 1, 1),        # several tiles per workgroup stream
    (2, [64], 64, 21, 160, 1, 1),        # width % 64 != 0: the 32 x 8 tile variant (LR-scale layers), ragged tile rows
    (1, [64, 256], 256, 16, 96, 2, 0),   # small_offset_conv1 geometry on 32-wide tiles
]


@pytest.mark.parametrize("case", WINO_CASES)
def test_conv3x3_winograd_matches_fp64_and_direct(ops, dev, case):
    """Winograd F(2,3)-along-x kernel (what ops.conv3x3 picks for 64-wide-tileable maps) against float64 conv2d and against
    the direct kernel: same fp32 products up to the rounding of the input / weight transforms -> 2e-5 * scale."""
    B, cins, Cout, H, W, act, nres = case
    xs = [_cl(_rand((B, c, H, W), dev, 110 + k)) for k, c in enumerate(cins)]
    w = _rand((Cout, sum(cins), 3, 3), dev, 120, 1.0 / np.sqrt(9 * sum(cins)))
    b = _rand((Cout,), dev, 121)
    res = [_cl(_rand((B, Cout, H, W), dev, 130 + k)) for k in range(nres)]
    kw = dict(act=act, slope=0.1, res1=res[0] if nres > 0 else None, res2=res[1] if nres > 1 else None)
    got = ops.conv3x3(xs, w, b, algo="winograd", **kw)
    direct = ops.conv3x3(xs, w, b, algo="direct", **kw)
    want = _ref(xs, w, b, act, 0.1, res)
    scale = max(1.0, float(want.abs().max()))
    assert float((got.double() - want).abs().max()) < 2e-5 * scale
    assert float((got - direct).abs().max()) < 2e-5 * scale
    assert float((direct.double() - want).abs().max()) < 1e-5 * scale


# ---------------------------------------------------------------------------------------------------------------------
# split kernel (csrc/conv3x3_split.hip).  "split": fp32 operands split exactly into three bf16 pieces, six piece products per
# product sum on the bf16 matrix pipe.  "split16": two round-to-nearest f16 pieces per activation, per-tensor-scaled weights,
# three products on the f16 matrix pipe.  Both held to the DIRECT kernel's tolerance (1e-5 * scale against float64).
# ---------------------------------------------------------------------------------------------------------------------
SPLIT_ALGOS = ("split", "split16")
SPLIT_CASES = [c for c in CASES if c[2] % 4 == 0] + WINO_CASES + [   # (Cout = 3: planar output only, see the modes test)
    (1, [16], 64, 8, 32, 0, 0),          # one chunk: prologue only, no steady state
    (1, [32], 64, 3, 5, 1, 0),           # map smaller than a tile
    (2, [48, 16], 96, 17, 33, 2, 1),     # 16-channel source boundary inside the stream, Cout = 64 + 32 (ragged cout block)
    (1, [64], 64, 64, 96, 1, 1),         # 24 tiles: several workgroups with several tiles each
    (1, [32], 64, 384, 384, 1, 0),       # 576 tiles > 512 resident workgroups: persistent workgroups with TWO tiles each (the
    (1, [16], 32, 384, 392, 2, 1),       # next tile's first chunk is split during the last chunk of this one); one-chunk tiles
]


@pytest.mark.parametrize("algo", SPLIT_ALGOS)
@pytest.mark.parametrize("case", SPLIT_CASES)
def test_conv3x3_split_matches_fp64_conv2d(ops, dev, case, algo):
    B, cins, Cout, H, W, act, nres = case
    xs = [_cl(_rand((B, c, H, W), dev, 210 + k)) for k, c in enumerate(cins)]
    w = _rand((Cout, sum(cins), 3, 3), dev, 220, 1.0 / np.sqrt(9 * sum(cins)))
    b = _rand((Cout,), dev, 221)
    res = [_cl(_rand((B, Cout, H, W), dev, 230 + k)) for k in range(nres)]
    kw = dict(act=act, slope=0.1, res1=res[0] if nres > 0 else None, res2=res[1] if nres > 1 else None)
    got = ops.conv3x3(xs, w, b, algo=algo, **kw)
    want = _ref(xs, w, b, act, 0.1, res)
    assert got.shape == want.shape and got.is_contiguous(memory_format=torch.channels_last)
    scale = max(1.0, float(want.abs().max()))
    err = float((got.double() - want).abs().max())
    assert err < 1e-5 * scale, err
    # the automatic choice: f16 x 2 where a range guard / an explicit flavour covers its domain, bf16 x 3 for a bare call
    if algo == "split16" and ops._SPLIT16:
        with ops.conv_flavour("f16x2"):
            auto = ops.conv3x3(xs, w, b, fast=True, **kw)
        assert torch.equal(auto, got)
    elif algo == "split":
        auto = ops.conv3x3(xs, w, b, fast=True, **kw)
        assert torch.equal(auto, got)


def test_conv3x3_split_is_at_least_as_accurate_as_the_fp32_mfma_kernels(ops, dev):
    """The six piece products leave a dropped-term error below the rounding of an fp32 fmaf chain of the same length
    (K = 9 * 256): the split kernel's distance from float64 must not exceed the direct (exact-fp32-MFMA) kernel's by more
    than half (first hardware run: direct 1.06e-5, split 7.6e-6, Winograd F(2,3) 6.2e-6 on outputs of magnitude ~5)."""
    x = _cl(_rand((1, 256, 24, 64), dev, 300))
    w, b = _rand((256, 256, 3, 3), dev, 301, 1.0 / 48.0), _rand((256,), dev, 302)
    want = _ref([x], w, b, 0, 0.0, [])
    errs = {a: float((ops.conv3x3(x, w, b, algo=a).double() - want).abs().max()) for a in ("direct", "winograd", "split", "split16")}
    assert errs["split"] <= 1.5 * errs["direct"] + 1e-7, errs
    assert errs["split16"] <= 1.5 * errs["direct"] + 1e-7, errs
    # ... and in the root-mean-square sense too
    rms = {a: float((ops.conv3x3(x, w, b, algo=a).double() - want).pow(2).mean().sqrt()) for a in ("direct", "split", "split16")}
    assert rms["split"] <= 1.5 * rms["direct"] and rms["split16"] <= 1.5 * rms["direct"], rms


@pytest.mark.parametrize("xs_,ws_", [(1e-3, 1.0), (1e3, 1.0), (1.0, 1e-6), (1.0, 1e5), (3e-3, 2e-4), (250.0, 37.0)])
def test_conv3x3_split16_scales(ops, dev, xs_, ws_):
    """f16 x 2 flavour away from unit scale: activations of magnitude 1e-3 ... 1e3 (the low piece is stored times 2^11: no
    fp16 underflow down to 2^-36 absolute) and weights of any magnitude (per-tensor power-of-two scale) keep the RELATIVE
    accuracy of the unit-scale case."""
    x = _cl(_rand((1, 64, 16, 40), dev, 330)) * xs_
    w, b = _rand((64, 64, 3, 3), dev, 331, 1.0 / 24.0) * ws_, _rand((64,), dev, 332) * (xs_ * ws_)
    want = _ref([x], w, b, 0, 0.0, [])
    scale = float(want.abs().max())
    e16 = float((ops.conv3x3(x, w, b, algo="split16").double() - want).abs().max())
    ed = float((ops.conv3x3(x, w, b, algo="direct").double() - want).abs().max())
    assert e16 < 1e-5 * scale and e16 <= 1.5 * ed + 1e-7 * scale, (e16, ed, scale)


def test_conv3x3_split16_domain(ops, dev):
    """Outside |x| < 65520 the f16 x 2 flavour returns NaN where the oversized value is read (never a finite wrong number);
    the bf16 x 3 flavour covers the whole fp32 range.  Zero weights (the zero-initialised DCN heads): exactly the bias.
    A mixed-magnitude activation map (1e-6 ... 1e4 in one tensor) keeps the accuracy relative to the OUTPUT scale."""
    x = _cl(_rand((1, 32, 8, 32), dev, 333))
    w, b = _rand((32, 32, 3, 3), dev, 334, 0.06), _rand((32,), dev, 335)
    xb = x.clone()
    xb[0, 5, 4, 7] = 1.0e5
    got = ops.conv3x3(xb, w, b, algo="split16")
    assert bool(torch.isnan(got[0, :, 3:6, 6:9]).all()) and bool(torch.isfinite(got[0, :, :, 10:]).all())
    ok = ops.conv3x3(xb, w, b, algo="split")
    want = _ref([xb], w, b, 0, 0.0, [])
    assert float((ok.double() - want).abs().max()) < 1e-5 * float(want.abs().max())
    z = ops.conv3x3(x, torch.zeros_like(w), b, algo="split16")
    assert torch.equal(z, b.view(1, -1, 1, 1).expand_as(z))
    mag = 10.0 ** (_rand((1, 32, 8, 32), dev, 336) * 2.5 - 1.0).clamp(-6.0, 3.5)   # 1e-6 ... 3e3 (x itself reaches ~4)
    xm = _cl(x * mag)
    want = _ref([xm], w, b, 0, 0.0, [])
    e16 = float((ops.conv3x3(xm, w, b, algo="split16").double() - want).abs().max())
    ed = float((ops.conv3x3(xm, w, b, algo="direct").double() - want).abs().max())
    assert e16 < 1e-5 * float(want.abs().max()) and e16 <= 1.5 * ed + 1e-7 * float(want.abs().max()), (e16, ed)


# ---------------------------------------------------------------------------------------------------------------------
# Winograd F(4,3) / F(2,3) ALONG Y on the f16 x 2 pieces (csrc/conv3x3_wino16.hip): "wino16" / "wino16_f23".  Channels-last
# output, Cout % 64 == 0, any map size.  Opt-in kernels (round 5's go / no-go on VERDICT r4 item 1: DESIGN.md 6.7).  Both are
# held to 1e-5 * scale against float64.  F(2,3) also meets the split kernel's second criterion (<= 1.5 x the exact-fp32-MFMA
# "direct" kernel's distance from float64 -- it is in fact closer than the direct f16 x 2 kernel); F(4,3) does NOT (measured
# 2 - 2.5 x on 64-channel layers, where the fp32 chain itself is short and the transforms' constants dominate): it is held to
# <= 3 x, and that is one of the two reasons it is not the default.
# ---------------------------------------------------------------------------------------------------------------------
WINO16_ALGOS = ("wino16", "wino16_f23")


@pytest.fixture
def experimental(ops):
    """Round 6: both no-go kernels moved to csrc/experimental/ and out of the product library (`make EXPERIMENTAL=1` builds them)."""
    if not ops.experimental_built():
        pytest.skip("libc2m_hip.so built without EXPERIMENTAL=1 (csrc/experimental/ kernels are not in the product library)")
WINO16_VS_DIRECT = {"wino16": 3.0, "wino16_f23": 1.5}
WINO16_CASES = [c for c in CASES + WINO_CASES if c[2] % 64 == 0] + [
    (1, [16], 64, 8, 32, 0, 0),          # one chunk per tile: prologue / first-operand paths only
    (1, [32], 64, 3, 5, 1, 0),           # map smaller than a tile in both directions
    (2, [48, 16], 128, 17, 33, 2, 1),    # 16-channel source boundary inside the stream, two cout blocks, ragged tiles
    (1, [64], 64, 64, 96, 1, 1),         # several tiles per workgroup
    (1, [32], 64, 390, 392, 1, 0),       # > 256 workgroup slots: persistent workgroups, odd chunk count per stream position
    (1, [16], 64, 384, 400, 2, 1),       # one-chunk tiles in a long stream (the plane-buffer / ring parities walk through every phase)
    (1, [48], 64, 50, 61, 0, 2),         # three chunks per tile: odd number of units per tile (ring slot / plane buffer parity per tile)
]


@pytest.mark.parametrize("algo", WINO16_ALGOS)
@pytest.mark.parametrize("case", WINO16_CASES)
def test_conv3x3_wino16_matches_fp64_conv2d(experimental, ops, dev, case, algo):
    B, cins, Cout, H, W, act, nres = case
    xs = [_cl(_rand((B, c, H, W), dev, 410 + k)) for k, c in enumerate(cins)]
    w = _rand((Cout, sum(cins), 3, 3), dev, 420, 1.0 / np.sqrt(9 * sum(cins)))
    b = _rand((Cout,), dev, 421)
    res = [_cl(_rand((B, Cout, H, W), dev, 430 + k)) for k in range(nres)]
    kw = dict(act=act, slope=0.1, res1=res[0] if nres > 0 else None, res2=res[1] if nres > 1 else None)
    got = ops.conv3x3(xs, w, b, algo=algo, **kw)
    want = _ref(xs, w, b, act, 0.1, res)
    assert got.shape == want.shape and got.is_contiguous(memory_format=torch.channels_last)
    scale = max(1.0, float(want.abs().max()))
    err = float((got.double() - want).abs().max())
    assert err < 1e-5 * scale, err
    if algo == {7: "wino16", 8: "wino16_f23"}.get(ops._WINO16) and ops._SPLIT16:   # the automatic choice under a guard / explicit flavour
        with ops.conv_flavour("f16x2"):
            auto = ops.conv3x3(xs, w, b, fast=True, **kw)
        assert torch.equal(auto, got)


def test_conv3x3_wino16_is_as_accurate_as_the_fp32_mfma_kernel(experimental, ops, dev):
    """Same criterion as the direct f16 x 2 kernel's: distance from float64 <= 1.5 x the exact-fp32-MFMA kernel's, in the
    maximum and in the root-mean-square sense, on K = 9 * 256 products per output (N(0,1) inputs) and on non-negative
    (post-ReLU-like) inputs, where a Winograd transform's intermediate magnitudes are least favourable."""
    for kind in ("normal", "relu"):
        x = _rand((1, 256, 32, 60), dev, 440)
        if kind == "relu":
            x = x.clamp_min(0)
        x = _cl(x)
        w, b = _rand((256, 256, 3, 3), dev, 441, 1.0 / 48.0), _rand((256,), dev, 442)
        want = _ref([x], w, b, 0, 0.0, [])
        outs = {a: ops.conv3x3(x, w, b, algo=a).double() for a in ("direct", "split16") + WINO16_ALGOS}
        errs = {a: float((o - want).abs().max()) for a, o in outs.items()}
        rms = {a: float((o - want).pow(2).mean().sqrt()) for a, o in outs.items()}
        for a in WINO16_ALGOS:
            assert errs[a] <= WINO16_VS_DIRECT[a] * errs["direct"] + 1e-7, (kind, errs)
            assert rms[a] <= WINO16_VS_DIRECT[a] * rms["direct"], (kind, rms)
            assert errs[a] < 1e-5 * max(1.0, float(want.abs().max())), (kind, errs)
        assert errs["wino16_f23"] <= 1.25 * errs["split16"] + 1e-7 and rms["wino16_f23"] <= 1.1 * rms["split16"], (kind, errs, rms)


@pytest.mark.parametrize("algo", WINO16_ALGOS)
@pytest.mark.parametrize("xs_,ws_", [(1e-3, 1.0), (1e3, 1.0), (1.0, 1e-6), (1.0, 1e5), (3e-3, 2e-4), (250.0, 37.0)])
def test_conv3x3_wino16_scales(experimental, ops, dev, algo, xs_, ws_):
    """Away from unit scale (activations 1e-3 ... 1e3, weights of any magnitude) the RELATIVE accuracy of the unit-scale case holds."""
    x = _cl(_rand((1, 64, 16, 40), dev, 450)) * xs_
    w, b = _rand((64, 64, 3, 3), dev, 451, 1.0 / 24.0) * ws_, _rand((64,), dev, 452) * (xs_ * ws_)
    want = _ref([x], w, b, 0, 0.0, [])
    scale = float(want.abs().max())
    e16 = float((ops.conv3x3(x, w, b, algo=algo).double() - want).abs().max())
    ed = float((ops.conv3x3(x, w, b, algo="direct").double() - want).abs().max())
    assert e16 < 1e-5 * scale and e16 <= WINO16_VS_DIRECT[algo] * ed + 1e-7 * scale, (e16, ed, scale)


@pytest.mark.parametrize("algo", WINO16_ALGOS)
def test_conv3x3_wino16_domain_and_range_flag(experimental, ops, dev, algo):
    """Domain |x| < 26200 (F(4,3)) / 32760 (F(2,3)): beyond it the kernel raises the range flag (the guarded module forwards then
    recompute on bf16 x 3) -- below it, mixed magnitudes 1e-6 ... 3e3 keep the tolerance; zero weights give exactly the bias."""
    x = _cl(_rand((1, 32, 8, 32), dev, 460))
    w, b = _rand((64, 32, 3, 3), dev, 461, 0.06), _rand((64,), dev, 462)
    ops.range_flag_set(dev)            # (clear what an earlier test's out-of-domain launch may have reported)
    assert not ops.range_flag_set(dev)
    ops.conv3x3(x, w, b, algo=algo)
    assert not ops.range_flag_set(dev)
    xb = x.clone()
    xb[0, 5, 4, 7] = 4.0e4
    ops.conv3x3(_cl(xb), w, b, algo=algo)
    assert ops.range_flag_set(dev) and not ops.range_flag_set(dev)     # reported, then cleared
    z = ops.conv3x3(x, torch.zeros_like(w), b, algo=algo)
    assert torch.equal(z, b.view(1, -1, 1, 1).expand_as(z))
    mag = 10.0 ** (_rand((1, 32, 8, 32), dev, 463) * 2.5 - 1.0).clamp(-6.0, 3.5)
    xm = _cl(x * mag)
    want = _ref([xm], w, b, 0, 0.0, [])
    e16 = float((ops.conv3x3(xm, w, b, algo=algo).double() - want).abs().max())
    ed = float((ops.conv3x3(xm, w, b, algo="direct").double() - want).abs().max())
    assert e16 < 1e-5 * float(want.abs().max()) and e16 <= WINO16_VS_DIRECT[algo] * ed + 1e-7 * float(want.abs().max()), (e16, ed)


@pytest.mark.parametrize("algo", WINO16_ALGOS)
def test_conv3x3_wino16_full_size_body_conv(experimental, ops, dev, algo):
    """configs[2]'s own body layer: 64 -> 64 @640^2, B = 16 (chip-filling: 14 080 tiles on 256 workgroups, 55 tiles per stream),
    ReLU + residual; a sub-block of the LAST sample against float64, the whole tensor against the direct f16 x 2 kernel, and
    bit-identical results across repeated launches."""
    x = _cl(_rand((16, 64, 640, 640), dev, 470))
    w, b = _rand((64, 64, 3, 3), dev, 471, 1.0 / 24.0), _rand((64,), dev, 472)
    r = _cl(_rand((16, 64, 640, 640), dev, 473))
    got = ops.conv3x3(x, w, b, act=ops.ACT_RELU, res1=r, algo=algo)
    ref16 = ops.conv3x3(x, w, b, act=ops.ACT_RELU, res1=r, algo="split16")
    scale = float(ref16.abs().max())
    assert float((got - ref16).abs().max()) < 1e-5 * scale
    for rep in range(2):
        assert torch.equal(ops.conv3x3(x, w, b, act=ops.ACT_RELU, res1=r, algo=algo), got), rep
    sub = (slice(15, 16), slice(None), slice(570, 640), slice(560, 640))
    want = (F.conv2d(x[sub].double(), w.double(), b.double(), padding=1).relu() + r[sub].double())[:, :, 2:, 2:]
    assert float((got[sub][:, :, 2:, 2:].double() - want).abs().max()) < 1e-5 * float(want.abs().max())


@pytest.mark.parametrize("shape", [(2, 45, 70), (1, 64, 96), (3, 8, 32), (1, 200, 130)])
def test_conv3x3_residual_epilogue_in_place_and_edges(ops, dev, shape):
    """The straight-line residual epilogue of the split kernel (tiles inside the image: residual pieces requested in batches, stores
    without branches; edge tiles: the generic path): maps with interior AND edge tiles, one and two residuals, fp32 and bf16 tensors --
    against float64, and IN PLACE (`out` is the residual tensor itself: every lane must have read its pieces before it overwrites them),
    bit-identical to the out-of-place result."""
    B, H, W = shape
    x = _cl(_rand((B, 64, H, W), dev, 900 + H))
    w, b = _rand((64, 64, 3, 3), dev, 901, 1.0 / 24.0), _rand((64,), dev, 902)
    r1, r2 = _cl(_rand((B, 64, H, W), dev, 903)), _cl(_rand((B, 64, H, W), dev, 904))
    for algo in ("split16", "split"):
        for res in ((r1,), (r1, r2), (None, r2)):
            kw = dict(act=ops.ACT_RELU, res1=res[0], res2=res[1] if len(res) > 1 else None, algo=algo)
            got = ops.conv3x3(x, w, b, **kw)
            want = _ref([x], w, b, 1, 0.1, [r for r in res if r is not None])
            assert float((got.double() - want).abs().max()) < 1e-5 * max(1.0, float(want.abs().max())), (algo, len(res))
            tgt = (res[0] if res[0] is not None else res[1]).clone(memory_format=torch.preserve_format)
            kw2 = dict(kw)
            kw2["res1" if res[0] is not None else "res2"] = tgt
            out = ops.conv3x3(x, w, b, out=tgt, **kw2)
            assert out.data_ptr() == tgt.data_ptr() and torch.equal(tgt, got), (algo, len(res))
    xb, rb = x.to(torch.bfloat16), r1.to(torch.bfloat16)                 # configs[4]'s body: bf16 source, residual and output
    got = ops.conv3x3(xb, w, b, res1=rb, algo="bf16", out_dtype=torch.bfloat16)
    want = F.conv2d(xb.double(), w.to(torch.bfloat16).double(), b.double(), padding=1) + rb.double()
    assert float((got.double() - want).abs().max()) < 2.0 ** -7 * max(1.0, float(want.abs().max()))   # one bf16 rounding of the sum
    tgt = rb.clone(memory_format=torch.preserve_format)
    ops.conv3x3(xb, w, b, res1=tgt, out=tgt, algo="bf16")
    assert torch.equal(tgt, got)


@pytest.mark.parametrize("algo", SPLIT_ALGOS)
def test_conv3x3_split_output_modes(ops, dev, algo):
    """PixelShuffle(2), planar NCHW, ReLU + MaxPool2d(2, 2), a channel-slice source and a strided (bordered) destination."""
    x = _cl(_rand((2, 64, 24, 40), dev, 304))
    w, b = _rand((256, 64, 3, 3), dev, 305, 0.05), _rand((256,), dev, 306)
    got = ops.conv3x3(x, w, b, act=ops.ACT_LRELU, slope=0.1, out_mode="pixel_shuffle", algo=algo)
    want = F.leaky_relu(F.pixel_shuffle(F.conv2d(x.double(), w.double(), b.double(), padding=1), 2), 0.1)
    assert tuple(got.shape) == (2, 64, 48, 80)
    assert float((got.double() - want).abs().max()) < 1e-5 * float(want.abs().max())
    # ragged: 80 = 64 + 16 output channels (the lane-swapped 16-byte stores serve whole 32-channel tiles only), odd map
    xr = _cl(_rand((1, 32, 9, 37), dev, 313))
    wr_, br_ = _rand((80, 32, 3, 3), dev, 314, 0.05), _rand((80,), dev, 315)
    got = ops.conv3x3(xr, wr_, br_, act=ops.ACT_LRELU, slope=0.1, out_mode="pixel_shuffle", algo=algo)
    want = F.leaky_relu(F.pixel_shuffle(F.conv2d(xr.double(), wr_.double(), br_.double(), padding=1), 2), 0.1)
    assert tuple(got.shape) == (1, 20, 18, 74)
    assert float((got.double() - want).abs().max()) < 1e-5 * float(want.abs().max())
    w3, b3 = _rand((3, 64, 3, 3), dev, 307, 0.05), _rand((3,), dev, 308)
    got = ops.conv3x3(x, w3, b3, out_mode="nchw", algo=algo)
    want = F.conv2d(x.double(), w3.double(), b3.double(), padding=1)
    assert got.is_contiguous() and float((got.double() - want).abs().max()) < 1e-5 * float(want.abs().max())
    # planar output of many channels: W % 4 == 0 takes the quad-transposed 16-byte stores, W % 4 != 0 the dword stores
    for (Co, H, W) in ((72, 11, 36), (72, 11, 38), (256, 9, 64)):
        xn = _cl(_rand((2, 32, H, W), dev, 316))
        wn, bn = _rand((Co, 32, 3, 3), dev, 317, 0.05), _rand((Co,), dev, 318)
        got = ops.conv3x3(xn, wn, bn, act=ops.ACT_RELU, out_mode="nchw", algo=algo)
        want = F.conv2d(xn.double(), wn.double(), bn.double(), padding=1).relu()
        assert got.is_contiguous() and float((got.double() - want).abs().max()) < 1e-5 * float(want.abs().max()), (Co, H, W)
    for (B, C, Co, H, W) in ((2, 64, 64, 12, 64), (1, 128, 128, 22, 90), (1, 64, 32, 8, 6)):
        xp = _cl(_rand((B, C, H, W), dev, 310))
        wp, bp = _rand((Co, C, 3, 3), dev, 311, 1.0 / np.sqrt(9 * C)), _rand((Co,), dev, 312)
        got = ops.conv3x3(xp, wp, bp, act=ops.ACT_RELU, out_mode="nhwc_pool2", algo=algo)
        plain = ops.conv3x3(xp, wp, bp, act=ops.ACT_RELU, algo=algo)
        assert got.shape == (B, Co, H // 2, W // 2) and torch.equal(got, F.max_pool2d(plain, 2, 2))
    big = _cl(_rand((2, 128, 24, 40), dev, 313))
    xs = big[:, 32:96]
    w2, b2 = _rand((64, 64, 3, 3), dev, 314, 0.05), _rand((64,), dev, 315)
    bo = ops._bordered_empty(2, 64, 24, 40, dev)
    view = bo.interior()
    ops.conv3x3(xs, w2, b2, out=view, algo=algo)
    want = _ref([xs], w2, b2, 0, 0.1, [])
    assert float((view.double() - want).abs().max()) < 1e-5 * float(want.abs().max())
    assert float(bo.buf[:, 0].abs().max()) == 0 and float(bo.buf[:, :, 41:].abs().max()) == 0


def test_conv3x3_bf16_single_piece_is_a_bf16_convolution(ops, dev):
    """algo="bf16": one round-to-nearest bf16 piece per operand, fp32 accumulation -- equal (to fp32 summation error) to a
    float64 convolution of the bf16-rounded tensors, and ~2^-8 relative away from the fp32 result."""
    x = _cl(_rand((2, 64, 20, 48), dev, 320))
    w, b = _rand((64, 64, 3, 3), dev, 321, 1.0 / 24.0), _rand((64,), dev, 322)
    got = ops.conv3x3(x, w, b, act=ops.ACT_RELU, algo="bf16")
    xb, wb = x.bfloat16().double(), w.bfloat16().double()
    want = F.conv2d(xb, wb, b.double(), padding=1).relu()
    scale = float(want.abs().max())
    assert float((got.double() - want).abs().max()) < 1e-5 * scale
    full = F.conv2d(x.double(), w.double(), b.double(), padding=1).relu()
    rel = float((got.double() - full).norm() / full.norm())
    assert 1e-4 < rel < 1e-2, rel


@pytest.mark.parametrize("shape", [(2, 64, 20, 48), (1, 64, 37, 70), (3, 32, 8, 32), (1, 128, 19, 33), (1, 64, 64, 256)])
def test_conv3x3_bf16_tensors_in_and_out(ops, dev, shape):
    """c2m_conv3x3_desc.io_flags: bf16 source (tile by LDS-DMA into three plane buffers), bf16 output (lane-swapped 16-byte
    stores), bf16 / fp32 residuals in every combination -- against a float64 convolution of the SAME bf16-rounded operands:
    fp32 outputs to fp32 summation error, bf16 outputs to one bf16 rounding (2^-9 relative) of the fp32 result."""
    B, C, H, W = shape
    bf = torch.bfloat16
    x = _cl(_rand((B, C, H, W), dev, 330))
    w, b = _rand((C, C, 3, 3), dev, 331, 1.0 / 24.0), _rand((C,), dev, 332)
    r1, r2 = _cl(_rand((B, C, H, W), dev, 333)), _cl(_rand((B, C, H, W), dev, 334))
    xh = x.to(bf).contiguous(memory_format=torch.channels_last)
    r1h = r1.to(bf).contiguous(memory_format=torch.channels_last)
    wb = w.bfloat16().double()
    core = F.conv2d(xh.double(), wb, b.double(), padding=1)
    cases = [  # (source, act, res1, res2, out dtype)
        (xh, ops.ACT_RELU, None, None, None),
        (xh, ops.ACT_NONE, r1h, None, bf),
        (xh, ops.ACT_NONE, r1h, r2, None),
        (xh, ops.ACT_LRELU, r1, r2, bf),
        (x, ops.ACT_RELU, None, None, bf),
        (x, ops.ACT_NONE, r1h, None, None),
    ]
    for src, act, a1, a2, od in cases:
        got = ops.conv3x3(src, w, b, act=act, slope=0.1, res1=a1, res2=a2, algo="bf16", out_dtype=od)
        assert got.dtype == (od or torch.float32) and got.is_contiguous(memory_format=torch.channels_last)
        want = core if src is xh else F.conv2d(x.bfloat16().double(), wb, b.double(), padding=1)
        want = want.relu() if act == ops.ACT_RELU else F.leaky_relu(want, 0.1) if act == ops.ACT_LRELU else want
        for r in (a1, a2):
            if r is not None:
                want = want + r.double()
        scale = float(want.abs().max())
        err = (got.double() - want).abs()
        if od is None:
            assert float(err.max()) < 1e-5 * scale, (src.dtype, act, od)
        else:
            assert bool((err <= want.abs() * 2.0 ** -8 + 1e-5 * scale).all()), (src.dtype, act, od, float(err.max()))
            again = ops.conv3x3(src, w, b, act=act, slope=0.1, res1=a1, res2=a2, algo="bf16")   # fp32 out, rounded here
            assert torch.equal(again.to(bf), got)


def test_conv3x3_bf16_tensors_views_and_rejections(ops, dev):
    """bf16 tensors as channel slices / row-pitched views; what the boundary refuses (two sources, other kernels, other modes)."""
    bf = torch.bfloat16
    big = _cl(_rand((2, 128, 24, 40), dev, 340)).to(bf).contiguous(memory_format=torch.channels_last)
    xs = big[:, 32:96]
    w, b = _rand((64, 64, 3, 3), dev, 341, 0.05), _rand((64,), dev, 342)
    outbig = torch.zeros((2, 128, 24, 40), dtype=bf, device=dev).contiguous(memory_format=torch.channels_last)
    ops.conv3x3(xs, w, b, out=outbig[:, 64:], algo="bf16")
    want = F.conv2d(xs.double(), w.bfloat16().double(), b.double(), padding=1)
    err = (outbig[:, 64:].double() - want).abs()
    assert bool((err <= want.abs() * 2.0 ** -8 + 1e-5 * float(want.abs().max())).all())
    assert float(outbig[:, :64].abs().max()) == 0
    from c2m_amd._lib import C2MError
    with pytest.raises(C2MError):
        ops.conv3x3([xs, xs], _rand((64, 128, 3, 3), dev, 343), None, algo="bf16")
    with pytest.raises(C2MError):
        ops.conv3x3(xs, w, b, algo="split")
    with pytest.raises(C2MError):
        ops.conv3x3(xs, w, b, algo="bf16", out_mode="nchw")
    with pytest.raises(C2MError):
        ops.conv3x3(xs.float(), w, b, algo="split16", out_dtype=bf)


@pytest.mark.parametrize("algo", SPLIT_ALGOS)
def test_dcn_head_on_the_split_kernel(ops, dev, algo):
    """DCN offset/mask head epilogue of the split kernel (192 + 24 channels): same check as the fp32-MFMA kernels'."""
    import c2m_oracle as oracle
    import synth
    B, C, dg = 2, 64, 8
    # W % 4 != 0: dword planar stores; W % 4 == 0: the quad-transposed 16-byte stores with the per-row flow window (every
    # scale; ragged tiles in x and y; windows that cross 32-pixel tile boundaries)
    for (h, w, s) in ((12, 14, 1), (12, 16, 2), (12, 16, 4), (11, 16, 1), (9, 44, 1), (13, 22, 2), (7, 19, 4), (10, 40, 4)):
        H, W = h * s, w * s
        feat = _cl(_rand((B, C, H, W), dev, 360 + s))
        wt, bs = _rand((3 * dg * 9, C, 3, 3), dev, 361, 0.02), _rand((3 * dg * 9,), dev, 362, 0.1)
        hp, wp = h - 2, w - 2
        idx = (synth.uniform((B, hp, wp), 363, 0.0, 1.0).astype(np.float64) * (hp * wp)).astype(np.int64) % (hp * wp)
        flow = ops.index_to_flow(torch.from_numpy(idx).to(dev))
        abs_sum = torch.zeros(256, dtype=torch.float64, device=dev)
        off, msk = ops.conv3x3_dcn_head(feat, wt, bs, dg, flow, s, abs_sum, algo=algo)
        raw = F.conv2d(feat.double(), wt.double(), bs.double(), padding=1)
        o1, o2, m = torch.chunk(raw, 3, dim=1)
        want_off = torch.cat((o1, o2), 1)
        pre = np.stack([oracle.build_pre_offsets(idx[b], h, w)[{1: 0, 2: 1, 4: 2}[s]] for b in range(B)])
        pre_t = torch.from_numpy(pre).to(dev).double().flip(-1).permute(0, 1, 4, 2, 3).reshape(B, 18, H, W).repeat(1, dg, 1, 1)
        want_abs = float(want_off.abs().sum())
        tol = 1e-5 * max(1.0, float(raw.abs().max()))
        assert float((off.double() - (want_off + pre_t)).abs().max()) < tol
        assert float((msk.double() - torch.sigmoid(m)).abs().max()) < 1e-6
        assert abs(float(abs_sum.sum()) - want_abs) < 1e-4 * want_abs


@pytest.mark.parametrize("algo", ["split16", "split"])
def test_dcn_head_store_paths_are_bit_identical_at_full_size(ops, dev, algo):
    """The quad-transposed 16-byte stores + register flow window against the dword stores + per-pixel flow loads on a chip-filling
    launch (two resident workgroups per CU, ~59 M values per call, repeated): every offset and mask bit equal.  (A staged
    version of these stores was withdrawn in round 4 for a one-in-10^6 corruption that only full-size launches showed.)"""
    import synth
    B, C, dg, s, h = 8, 64, 8, 4, 160
    H = W = h * s
    feat = _cl(_rand((B, C, H, W), dev, 380))
    wt, bs = _rand((3 * dg * 9, C, 3, 3), dev, 381, 0.02), _rand((3 * dg * 9,), dev, 382, 0.1)
    hp = h - 2
    idx = (synth.uniform((B, hp, hp), 383, 0.0, 1.0).astype(np.float64) * (hp * hp)).astype(np.int64) % (hp * hp)
    flow = ops.index_to_flow(torch.from_numpy(idx).to(dev))
    with ops.head_store_mode(0):
        a0 = torch.zeros(256, dtype=torch.float64, device=dev)
        off0, msk0 = ops.conv3x3_dcn_head(feat, wt, bs, dg, flow, s, a0, algo=algo)
    for rep in range(3):
        with ops.head_store_mode(1):
            a1 = torch.zeros(256, dtype=torch.float64, device=dev)
            off1, msk1 = ops.conv3x3_dcn_head(feat, wt, bs, dg, flow, s, a1, algo=algo)
        assert torch.equal(off0, off1) and torch.equal(msk0, msk1), rep
        assert abs(float(a0.sum()) - float(a1.sum())) <= 1e-9 * float(a0.sum())
        del off1, msk1
    # the medium stage (scale 2, 128 -> 216): two sources
    f2a, f2b = _cl(_rand((B, 64, 320, 320), dev, 384)), _cl(_rand((B, 64, 320, 320), dev, 385))
    w2, b2 = _rand((3 * dg * 9, 128, 3, 3), dev, 386, 0.02), _rand((3 * dg * 9,), dev, 387, 0.1)
    with ops.head_store_mode(0):
        r0 = ops.conv3x3_dcn_head([f2a, f2b], w2, b2, dg, flow, 2, None, algo=algo)
    with ops.head_store_mode(1):
        r1 = ops.conv3x3_dcn_head([f2a, f2b], w2, b2, dg, flow, 2, None, algo=algo)
    assert torch.equal(r0[0], r1[0]) and torch.equal(r0[1], r1[1])


@pytest.mark.parametrize("algo", SPLIT_ALGOS)
def test_pixel_shuffle_and_planar_store_paths_are_bit_identical_at_full_size(ops, dev, algo):
    """the PixelShuffle epilogue's lane-swapped 16-byte stores and the planar (NCHW) epilogue's
    quad-transposed 16-byte stores against their one-dword-per-lane paths (conv3x3(dword_stores=True), per call) on
    chip-filling launches -- configs[2]'s own shapes at B=16: tail 64 -> 256 PixelShuffle @320^2 (105 M values per call,
    6 400 tiles on 512 resident workgroup slots: two workgroups per CU throughout) and the extractor's planar 256-channel
    output @160^2 plus a 64-channel planar map @640^2 -- repeated, both flavours: every bit equal.  (The staged stores
    withdrawn in round 4 were wrong once in 10^6 values and only on full CUs; small maps cannot see that class.)"""
    x = _cl(_rand((16, 64, 320, 320), dev, 390))
    w, b = _rand((256, 64, 3, 3), dev, 391, 0.05), _rand((256,), dev, 392)
    kw = dict(act=ops.ACT_LRELU, slope=0.1, out_mode="pixel_shuffle", algo=algo)
    ref = ops.conv3x3(x, w, b, dword_stores=True, **kw)
    assert tuple(ref.shape) == (16, 64, 640, 640) and bool(torch.isfinite(ref).all())
    for rep in range(3):
        got = ops.conv3x3(x, w, b, **kw)
        assert torch.equal(got, ref), ("pixel_shuffle", rep)
        del got
    del ref, x
    for (B, C, Co, H, W, seed) in ((16, 64, 256, 160, 160, 393), (16, 32, 64, 640, 640, 396)):
        xn = _cl(_rand((B, C, H, W), dev, seed))
        wn, bn = _rand((Co, C, 3, 3), dev, seed + 1, 0.05), _rand((Co,), dev, seed + 2)
        kw = dict(act=ops.ACT_RELU, out_mode="nchw", algo=algo)
        ref = ops.conv3x3(xn, wn, bn, dword_stores=True, **kw)
        assert ref.is_contiguous() and bool(torch.isfinite(ref).all())
        for rep in range(3):
            got = ops.conv3x3(xn, wn, bn, **kw)
            assert torch.equal(got, ref), ("nchw", Co, H, rep)
            del got
        # ... and the dword path itself is the convolution (a sub-block against float64)
        want = F.conv2d(xn[:1, :, :40, :72].double(), wn.double(), bn.double(), padding=1).relu()[:, :, :38, :70]
        assert float((ref[:1, :, :38, :70].double() - want).abs().max()) < 1e-5 * float(want.abs().max())
        del ref, xn


def test_weight_cache_refresh_follows_data_writes(ops, dev):
    """`w.data.copy_()` does not bump `_version`; ops.refresh_weight_caches() (called by the fused forwards once per
    call) rebuilds the cached weight images from the tensor's current contents."""
    x = _cl(_rand((1, 32, 8, 8), dev, 350))
    w = torch.nn.Parameter(_rand((32, 32, 3, 3), dev, 351, 0.1))
    for algo in ("direct", "split", "split16"):
        a = ops.conv3x3(x, w, algo=algo)
        v0 = w._version
        w.data.mul_(2.0)
        assert w._version == v0
        assert ops.refresh_weight_caches([w]) >= 1
        b2 = ops.conv3x3(x, w, algo=algo)
        assert float((b2 - 2 * a).abs().max()) < 1e-5 * float(a.abs().max())
        w.data.mul_(0.5)
        ops.refresh_weight_caches([w])


def test_multi_tensor_refresh_equals_a_fresh_relayout(ops, dev):
    """ops.refresh_weight_caches rebuilds all split-kernel images of its parameters in ONE multi-tensor call (device job
    table): after `.data` writes the refreshed images must be bit-identical to images built from scratch -- every flavour
    (bf16 x 3, f16 x 2 with its per-tensor scale, bf16), ragged cout blocks, a row-sliced DCN head, the data-gradient image."""
    P = torch.nn.Parameter
    w1, w2, w3 = P(_rand((64, 64, 3, 3), dev, 370, 0.05)), P(_rand((32, 48, 3, 3), dev, 371, 0.07)), P(_rand((256, 128, 3, 3), dev, 372, 0.03))
    wh, bh = P(_rand((216, 64, 3, 3), dev, 373, 0.02)), _rand((216,), dev, 374, 0.1)
    x1, x2, x3 = _cl(_rand((1, 64, 12, 40), dev, 375)), _cl(_rand((1, 48, 12, 40), dev, 376)), _cl(_rand((1, 128, 12, 40), dev, 377))
    g1 = _cl(_rand((1, 64, 12, 40), dev, 378))
    flow = torch.zeros(1, 10, 38, 2, device=dev)

    def run():
        outs = []
        for algo in ("split", "split16", "bf16"):
            outs += [ops.conv3x3(x1, w1, algo=algo), ops.conv3x3(x2, w2, algo=algo), ops.conv3x3(x3, w3, algo=algo)]
        outs += list(ops.conv3x3_dcn_head(x1, wh, bh, 8, flow, 1, algo="split16"))
        outs.append(ops.conv3x3_dgrad(g1, w1))
        return outs

    before = run()
    for k, w in enumerate((w1, w2, w3, wh)):
        w.data.mul_(1.7 + k).add_(_rand(tuple(w.shape), dev, 380 + k, 0.01))     # (no version bump)
    assert ops.refresh_weight_caches([w1, w2, w3, wh]) >= 11
    refreshed = run()
    ops.clear_weight_caches()
    fresh = run()
    for a, b_, c in zip(before, refreshed, fresh):
        assert torch.equal(b_, c)
        assert not torch.equal(a, b_)
    # a second round (same job table) after another write
    w2.data.neg_()
    ops.refresh_weight_caches([w1, w2, w3, wh])
    again = ops.conv3x3(x2, w2, algo="split16")
    ops.clear_weight_caches()
    assert torch.equal(again, ops.conv3x3(x2, w2, algo="split16"))


# ---------------------------------------------------------------------------------------------------------------------
# backward: data gradient (split kernel on rotated / transposed weights), weight gradient (csrc/conv3x3_wgrad.hip), the
# autograd Function around them -- against float64 autograd of F.conv2d
# ---------------------------------------------------------------------------------------------------------------------
GRAD_CASES = [
    # B, [Cin per source], Cout, H, W, act
    (2, [64], 64, 40, 40, 1),          # body conv1 at the stage-3 training size (LR 40), ReLU
    (2, [64], 64, 37, 45, 0),          # ragged segments
    (1, [64, 256], 256, 20, 24, 2),    # small_offset_conv1: two sources, LeakyReLU
    (1, [64], 216, 12, 40, 0),         # DCN offset/mask head: Cout not a multiple of 16 (padded data-gradient conv)
    (2, [64, 64], 64, 33, 70, 2),      # head_large geometry, three x segments
    (1, [32], 32, 8, 5, 0),            # one 32-channel block, map smaller than a segment
]


@pytest.mark.parametrize("case", GRAD_CASES)
def test_conv3x3_autograd_matches_fp64_conv2d_gradients(ops, dev, case):
    """gradients of act(conv3x3(cat(srcs)) + bias) w.r.t. every source, the weight and the bias against
    float64 autograd through F.conv2d, at 1e-5 * scale of each gradient."""
    B, cins, Cout, H, W, act = case
    xs = [_cl(_rand((B, c, H, W), dev, 410 + k)).requires_grad_(True) for k, c in enumerate(cins)]
    w = _rand((Cout, sum(cins), 3, 3), dev, 420, 1.0 / np.sqrt(9 * sum(cins))).requires_grad_(True)
    b = _rand((Cout,), dev, 421).requires_grad_(True)
    gy = _rand((B, Cout, H, W), dev, 422)
    y = ops.conv3x3_autograd(xs, w, b, act=act, slope=0.1)
    y.backward(gy)
    xs64 = [x.detach().double().requires_grad_(True) for x in xs]
    w64, b64 = w.detach().double().requires_grad_(True), b.detach().double().requires_grad_(True)
    y64 = F.conv2d(torch.cat(xs64, 1), w64, b64, padding=1)
    y64 = y64.clamp_min(0) if act == 1 else torch.where(y64 > 0, y64, y64 * 0.1) if act == 2 else y64
    y64.backward(gy.double())
    assert float((y.double() - y64).abs().max()) < 1e-5 * max(1.0, float(y64.abs().max()))
    for name, got, want in [("w", w.grad, w64.grad), ("b", b.grad, b64.grad)] + [(f"x{k}", x.grad, x64.grad) for k, (x, x64) in enumerate(zip(xs, xs64))]:
        scale = max(1e-6, float(want.abs().max()))
        err = float((got.double() - want).abs().max())
        assert err < 1e-5 * scale, (name, err, scale)


def test_conv3x3_wgrad_and_dgrad_entry_points(ops, dev):
    """The two backward entry points on their own, with a strided (bordered) source and a grad_out that is a channel-slice view."""
    B, C, Co, H, W = 2, 64, 64, 24, 40
    bo = ops._bordered_empty(B, C, H, W, dev)
    x = bo.interior()
    x.copy_(_rand((B, C, H, W), dev, 430))
    big = _cl(_rand((B, 2 * Co, H, W), dev, 431))
    g = big[:, Co:]
    w = _rand((Co, C, 3, 3), dev, 432, 0.05)
    gw = ops.conv3x3_wgrad([x], g, Co)
    dx = ops.conv3x3_dgrad(g, w)
    x64 = x.detach().double().contiguous().requires_grad_(True)
    w64 = w.double().requires_grad_(True)
    F.conv2d(x64, w64, None, padding=1).backward(g.double().contiguous())
    assert float((gw.double() - w64.grad).abs().max()) < 1e-5 * float(w64.grad.abs().max())
    assert float((dx.double() - x64.grad).abs().max()) < 1e-5 * float(x64.grad.abs().max())


def test_f16_range_guard_switches_a_module_to_bf16x3(ops, dev):
    """The default f16 x 2 flavour covers |activation| < 65520.  A launch that meets a larger value raises the device flag
    (c2m_conv3x3_desc.range_flag); a fused module forward then recomputes on bf16 x 3 (full fp32 range), warns once and
    keeps that flavour -- a drop-in never returns NaN where the reference's nn.Conv2d stack (arch_util.py:80-136,
    vgg_arch.py:107-145) returns a number."""
    import warnings
    from mmsr.models.archs.vgg_arch import VGGFeatureExtractor
    # (1) the flag itself, at the ops level
    x = _cl(_rand((1, 32, 8, 32), dev, 401))
    w, b = _rand((32, 32, 3, 3), dev, 402, 0.06), _rand((32,), dev, 403)
    flag = ops._range_flag(x.device)
    flag.zero_()
    ops.conv3x3(x, w, b, algo="split16")
    assert int(flag.item()) == 0
    xb = x.clone()
    xb[0, 3, 2, 9] = -7.0e4
    ops.conv3x3(xb, w, b, algo="split16")
    assert int(flag.item()) == 1
    # (2) a VGG tower fed an un-normalised 0..255-style image scaled far out of range
    with warnings.catch_warnings():
        warnings.simplefilter("ignore", RuntimeWarning)
        vgg = VGGFeatureExtractor(["relu1_1", "relu2_1", "relu3_1"], "vgg19", use_input_norm=False).to(dev).eval()
    torch.manual_seed(5)
    for m in vgg.modules():
        if isinstance(m, torch.nn.Conv2d):
            torch.nn.init.kaiming_normal_(m.weight)
            torch.nn.init.normal_(m.bias, std=0.05)
    img = torch.rand(1, 3, 40, 48, device=dev) * 6.0e4          # relu1_1 activations reach ~1e5 .. 1e6
    ref = {}
    vd = VGGFeatureExtractor(["relu1_1", "relu2_1", "relu3_1"], "vgg19", use_input_norm=False).double().eval()
    vd.load_state_dict({k: v.double().cpu() for k, v in vgg.state_dict().items()})
    with torch.no_grad():
        ref = vd(img.double().cpu())
        assert vgg._use_fused(img) and not getattr(vgg, "_c2m_conv_bf16x3", False)
        with pytest.warns(RuntimeWarning, match="bf16 x 3"):
            got = vgg(img)
        assert vgg._c2m_conv_bf16x3 is True
        for k, want in ref.items():
            assert bool(torch.isfinite(got[k]).all()), k
            scale = float(want.abs().max())
            assert float((got[k].double().cpu() - want).abs().max()) <= 1e-5 * scale, k
        assert max(float(v.abs().max()) for v in ref.values()) > 65520.0     # the input really left the f16 x 2 domain
        with warnings.catch_warnings():
            warnings.simplefilter("error")                        # pinned: no second warning, no f16 x 2 launch
            again = vgg(img)
        assert all(torch.equal(again[k], got[k]) for k in got)
        small = vgg(torch.rand(1, 3, 40, 48, device=dev))         # stays on bf16 x 3 (finite either way)
        assert all(bool(torch.isfinite(v).all()) for v in small.values())


def test_loader_matrix_wave_kernel_opt_in(experimental, dev):
    """csrc/conv3x3_pc.hip ($C2M_CONV_PC=1: the f16 x 2 arithmetic with loader and matrix waves, DESIGN.md 6.10 -- measured
    slower than the split kernel, kept opt-in): ragged, multi-chunk and chip-filling 64-cout layers against float64 and the
    fp32-MFMA kernel, bit-identical repeats.  A process of its own: the switch is read once per process."""
    import subprocess, sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, C2M_CONV_PC="1", C2M_CONV_PC_MINPIX="0")
    r = subprocess.run([sys.executable, os.path.join(repo, "scripts", "diag_pc.py")], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "ALL OK" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]


# ---------------------------------------------------------------------------------------------------------------------
# Round 6: a whole ResidualBlockNoBN (arch_util.py:80-136) in ONE launch (csrc/experimental/conv3x3_resblock.hip; VERDICT r5
# item 1).  Built, right on hardware, measured 6 - 7 % slower than two launches of the split kernel: a recorded
# no-go that lives in the experimental library (`make EXPERIMENTAL=1`, $C2M_LIB); these tests skip on the product library.
# ---------------------------------------------------------------------------------------------------------------------
RESBLOCK_CASES = [(1, 8, 30, 0), (1, 6, 30, 0), (1, 5, 7, 1), (2, 19, 61, 0), (3, 40, 95, 1), (1, 24, 32, 1), (2, 330, 210, 1), (16, 64, 64, 0)]


def _resblock_inputs(dev, B, H, W, seed):
    x = _cl(_rand((B, 64, H, W), dev, seed))
    w1 = _rand((64, 64, 3, 3), dev, seed + 1, 1.0 / 24.0)
    w2 = _rand((64, 64, 3, 3), dev, seed + 2, 1.0 / 24.0)
    b1, b2 = _rand((64,), dev, seed + 3, 0.1), _rand((64,), dev, seed + 4, 0.1)
    return x, w1, b1, w2, b2


@pytest.mark.parametrize("case", RESBLOCK_CASES)
def test_fused_residual_block_matches_the_two_launch_path(experimental, ops, dev, case):
    """out = x + conv2(relu(conv1(x))) (+ res2) in one launch against (a) float64 conv2d -- the split kernel's tolerances: 1e-5 *
    scale and <= 1.5 x the two-launch path's own distance -- and (b) the two-launch f16 x 2 path itself: the only difference is
    the identity rebuilt from x's two f16 pieces, <= 2^-22 |x| (asserted at 3 ulp of the output scale).  Ragged maps, maps smaller
    than a strip / a step, several strips and steps, a priming step inside a strip (B = 16: 256 workgroup ranges)."""
    B, H, W, with_res2 = case
    x, w1, b1, w2, b2 = _resblock_inputs(dev, B, H, W, 700 + H)
    r2 = _cl(_rand((B, 64, H, W), dev, 777)) if with_res2 else None
    with ops.conv_flavour("f16x2"):
        got = ops.resblock3x3(x, w1, b1, w2, b2, res2=r2)
        t = ops.conv3x3(x, w1, b1, act=ops.ACT_RELU, algo="split16")
        two = ops.conv3x3(t, w2, b2, res1=x, res2=r2, algo="split16")
    xd = x.double()
    want = xd + F.conv2d(F.relu(F.conv2d(xd, w1.double(), b1.double(), padding=1)), w2.double(), b2.double(), padding=1)
    if r2 is not None:
        want = want + r2.double()
    scale = max(1.0, float(want.abs().max()))
    e_f, e_2 = float((got.double() - want).abs().max()), float((two.double() - want).abs().max())
    assert got.is_contiguous(memory_format=torch.channels_last) and bool(torch.isfinite(got).all())
    assert e_f < 1e-5 * scale and e_f <= 1.5 * e_2 + 1e-7 * scale, (e_f, e_2)
    assert float((got - two).abs().max()) <= 3 * 2.0 ** -23 * scale


def test_fused_residual_block_full_size_and_domain(experimental, ops, dev):
    """configs[2]'s own body layer (64 -> 64 @640^2, B = 16: every workgroup range starts inside a strip) against the two-launch
    path, bit-stable over repeats; an activation beyond the f16 x 2 domain raises the same range flag as the split kernel."""
    x, w1, b1, w2, b2 = _resblock_inputs(dev, 16, 640, 640, 900)
    with ops.conv_flavour("f16x2"):
        a = ops.resblock3x3(x, w1, b1, w2, b2)
        b = ops.resblock3x3(x, w1, b1, w2, b2)
        t = ops.conv3x3(x, w1, b1, act=ops.ACT_RELU, algo="split16")
        two = ops.conv3x3(t, w2, b2, res1=x, algo="split16")
        assert torch.equal(a, b)
        assert float((a - two).abs().max()) <= 3 * 2.0 ** -23 * max(1.0, float(two.abs().max()))
        assert not ops.range_flag_set(dev)
        xb = x[:1, :, :40, :40].clone(memory_format=torch.channels_last)
        xb[0, 5, 7, 9] = 7.0e4
        ops.resblock3x3(xb, w1, b1, w2, b2)
        assert ops.range_flag_set(dev)
