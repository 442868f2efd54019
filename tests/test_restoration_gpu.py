"""GPU: the callers on either side of the kernels -- RestorationNet against the golden SR tensor produced by the
reference's own Python (with the C oracle standing in for its CUDA-only DCNv2), the full extractor -> correspondence ->
restoration chain, and one stage-3 training step."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _fill(module, prefix):
    from make_golden import fill_parameters
    fill_parameters(module, prefix)


def test_restoration_net_matches_reference_golden(dev, golden_dir):
    """BASELINE config 1 size (LR 40x40 -> SR 160x160).  Tolerance: north_star's 1e-3 abs on SR pixels; measured error
    is ~1e-6 (asserted at 5e-5 so that a wrong tap / group / corner cannot hide: the restoration residual is ~1e-2)."""
    from make_golden import restoration_inputs
    from mmsr.models.archs.ref_restoration_arch import RestorationNet
    gold = np.load(f"{golden_dir}/restoration_golden.npz")
    net = RestorationNet(ngf=64, n_blocks=16, groups=8).eval()
    _fill(net, "net_g.")
    net = net.to(dev)
    lr, pre, feats = restoration_inputs(1, 40, 40)
    taps = {}
    for stage in ("small", "medium", "large"):
        getattr(net.dyn_agg_restore, f"{stage}_dyn_agg").register_forward_hook(
            lambda m, i, o, stage=stage: taps.__setitem__(stage, o.detach().cpu().numpy()[..., ::5, ::5]))
    t = lambda a: torch.from_numpy(a).to(dev)  # noqa: E731
    with torch.no_grad():
        sr = net(t(lr), {k: t(v) for k, v in pre.items()}, {k: t(v) for k, v in feats.items()})
    for stage in ("small", "medium", "large"):
        err = float(np.abs(taps[stage] - gold[f"dyn_agg_{stage}"]).max())
        assert err < 5e-5, f"DynAgg {stage}: {err}"
    err = float(np.abs(sr.cpu().numpy() - gold["sr"]).max())
    assert err < 5e-5, f"SR max abs err {err}"
    # the fused inference path (channels-last conv3x3 / DCN-head / DCNv2 kernels, pre-offsets synthesised from the index map
    # the fixture's pre-offset tensors were built from) against the same golden
    from make_golden import restoration_index_map
    from mmsr.models.archs.corres_generation_arch import PreOffsets
    lazy = PreOffsets(t(restoration_index_map(1, 40, 40)), 40, 40)
    taps.clear()
    with torch.no_grad():
        assert net._use_fused(t(lr), lazy, {k: t(v) for k, v in feats.items()})
        sr_f = net(t(lr), lazy, {k: t(v) for k, v in feats.items()})
    for stage in ("small", "medium", "large"):   # hook output on this path = lrelu(DynAgg): undo it (slope 0.1, exact sign)
        got = np.where(taps[stage] > 0, taps[stage], taps[stage] / np.float32(0.1))
        err = float(np.abs(got - gold[f"dyn_agg_{stage}"]).max())
        assert err < 5e-5, f"fused DynAgg {stage}: {err}"
    err = float(np.abs(sr_f.cpu().numpy() - gold["sr"]).max())
    assert err < 5e-5, f"fused path SR max abs err {err}"
    assert np.array_equal(lazy["relu2_1"].cpu().numpy(), pre["relu2_1"])   # the lazy dict still yields the reference's tensors


def _build_chain(dev):
    from mmsr.models.archs.contras_extractor_arch import ContrasExtractorSep
    from mmsr.models.archs.corres_generation_arch import CorrespondenceGenerationArch
    from mmsr.models.archs.ref_restoration_arch import RestorationNet
    ext = ContrasExtractorSep().eval()
    mp = CorrespondenceGenerationArch(3, 1, ["relu1_1", "relu2_1", "relu3_1"], "vgg19").eval()
    g = RestorationNet(64, 16, 8).eval()
    torch.manual_seed(7)
    for m in list(ext.modules()) + list(mp.modules()):
        if isinstance(m, torch.nn.Conv2d):
            torch.nn.init.kaiming_normal_(m.weight)
            torch.nn.init.normal_(m.bias, std=0.05)
    _fill(g, "net_g.")
    return ext.to(dev), mp.to(dev), g.to(dev)


def test_full_chain_config1_matches_cpu_chain_built_from_oracle_ops(dev):
    """LR 40x40, Ref 64x64 zero-padded to 160x160 (BASELINE configs[0]).  The plain convolutions run once on the GPU;
    their outputs feed BOTH the HIP hot path and the CPU oracle ops, so the comparison isolates the hot path:
    index map / pre-offsets bit-exact, SR within 1e-3 abs (measured ~1e-6)."""
    import c2m_oracle as oracle
    import synth
    ext, mp, g = _build_chain(dev)
    lr = synth.uniform((1, 3, 40, 40), 2000, 0.0, 1.0)
    up = torch.nn.functional.interpolate(torch.from_numpy(lr), scale_factor=4, mode="bicubic", align_corners=False).clamp(0, 1)
    ref = np.zeros((1, 3, 160, 160), np.float32)
    ref[:, :, :64, :64] = synth.uniform((1, 3, 64, 64), 2001, 0.0, 1.0)   # test-time zero padding (ref_cufed_dataset.py:107-114)
    with torch.no_grad():
        feats = ext(up.to(dev), torch.from_numpy(ref).to(dev))
        pre, ref_feat = mp(feats, torch.from_numpy(ref).to(dev))
        sr = g(torch.from_numpy(lr).to(dev), pre, ref_feat)
    assert tuple(sr.shape) == (1, 3, 160, 160) and bool(torch.isfinite(sr).all())
    f1 = oracle.feature_normalize(feats["dense_features1"][0].cpu().numpy())
    f2 = oracle.feature_normalize(feats["dense_features2"][0].cpu().numpy())
    idx, _ = oracle.feature_match_index(f1, f2, 3, 1, 1, True, True)
    o3, o2, o1 = oracle.build_pre_offsets(idx, 40, 40)
    assert np.array_equal(pre["relu3_1"][0].cpu().numpy(), o3)
    assert np.array_equal(pre["relu2_1"][0].cpu().numpy(), o2)
    assert np.array_equal(pre["relu1_1"][0].cpu().numpy(), o1)
    # zero-padded ref -> exact ties; the padded region's first patch wins them (lowest index), never a later copy
    assert int(idx.max()) < 38 * 38


def test_full_chain_under_bf16_autocast(dev):
    """bf16 autocast at configs[0] size: the plain convolutions compute in bf16; the hot path takes their outputs as
    float32 (custom_fwd(cast_inputs=float32) / .float()).  PSNR against a synthetic ground truth, computed with the
    reference's validation rules, stays within the 0.02 dB budget BASELINE configs[4] states (the full-size version of this
    test is test_cfg5_320_bf16_autocast_psnr_budget)."""
    from mmsr.utils import metrics
    ext, mp, g = _build_chain(dev)
    gt = _smooth_gt(1, 160, 5).to(dev)
    lq = torch.nn.functional.interpolate(gt, scale_factor=0.25, mode="bicubic", align_corners=False).clamp(0, 1)
    up = torch.nn.functional.interpolate(lq, scale_factor=4, mode="bicubic", align_corners=False).clamp(0, 1)
    ref = torch.zeros((1, 3, 160, 160), device=dev)
    ref[:, :, :64, :64] = _smooth_gt(1, 64, 6).to(dev)

    def run():
        feats = ext(up, ref)
        pre, ref_feat = mp(feats, ref)
        return g(lq, pre, ref_feat)

    with torch.no_grad():
        sr32 = run()
        with torch.autocast("cuda", dtype=torch.bfloat16):
            sr16 = run()
    assert torch.isfinite(sr16).all()
    p32 = float(metrics.validation_metrics(sr32, gt, 4)["psnr"][0])
    p16 = float(metrics.validation_metrics(sr16.float(), gt, 4)["psnr"][0])
    assert 15.0 < p32 < 60.0 and abs(p16 - p32) <= 0.02, (p32, p16)


def test_stage3_training_step_runs_and_learns(dev):
    """One GPU, the reference's stage-3 MSE settings at a tiny size: loss is finite and decreases, every net_g parameter
    that the reference optimises receives a gradient (DCNv2 backward included)."""
    import synth
    from mmsr.models.ref_restoration_model import RefRestorationModel
    opt = {"dist": False, "gpu_ids": [0], "is_train": True, "path": {},
           "network_g": {"type": "RestorationNet", "ngf": 64, "n_blocks": 2, "groups": 8},
           "network_map": {"type": "CorrespondenceGenerationArch", "patch_size": 3, "stride": 1,
                           "vgg_layer_list": ["relu1_1", "relu2_1", "relu3_1"], "vgg_type": "vgg19"},
           "network_extractor": {"type": "ContrasExtractorSep"},
           "train": {"lr_g": 1e-3, "lr_offset": 1e-3, "lr_relu2_offset": 1e-4, "lr_relu3_offset": 1e-5,
                     "weight_decay_g": 0, "beta_g": [0.9, 0.999], "pixel_weight": 1.0}}
    torch.manual_seed(3)
    model = RefRestorationModel(opt)
    B, h = 2, 16
    gt = torch.from_numpy(synth.uniform((B, 3, 4 * h, 4 * h), 3000, 0.0, 1.0))
    lq = torch.nn.functional.interpolate(gt, scale_factor=0.25, mode="bicubic", align_corners=False).clamp(0, 1)
    up = torch.nn.functional.interpolate(lq, scale_factor=4, mode="bicubic", align_corners=False).clamp(0, 1)
    ref = torch.from_numpy(synth.uniform((B, 3, 4 * h, 4 * h), 3001, 0.0, 1.0))
    model.feed_data({"img_in_lq": lq, "img_ref": ref, "img_in": gt, "img_in_up": up})
    losses = []
    for step in range(1, 7):
        model.optimize_parameters(step)
        losses.append(float(model.log_dict["l_g_pix"]))
        if step == 1:
            missing = [n for n, p in model.net_g.named_parameters() if p.grad is None]
            assert not missing, missing
            nz = {n: float(p.grad.abs().max()) for n, p in model.net_g.named_parameters() if "dyn_agg" in n}
            assert all(np.isfinite(v) for v in nz.values())
            assert nz["dyn_agg_restore.small_dyn_agg.weight"] > 0 and nz["dyn_agg_restore.large_dyn_agg.conv_offset_mask.weight"] > 0
    assert all(np.isfinite(losses)) and losses[-1] < losses[0], losses
    out = model.test()
    assert tuple(out.shape) == (B, 3, 4 * h, 4 * h)


# ---------------------------------------------------------------------------------------------------------------------
# BASELINE-size chains (configs[2], [3], [4]).  The DynAgg sites are checked against the oracle through MODULE-level
# hooks (inputs [ref_feat, offset_feat] + pre_offset -> output), so the checks do not depend on how the operator is
# fused internally: the expected value is conv_offset_mask (torch fp64 on the CPU) -> the offset/mask assembly of
# dcn_v2.py:229-245 in numpy -> oracle.dcn_v2_forward.
# ---------------------------------------------------------------------------------------------------------------------
@torch.no_grad()
def _dynagg_expected(oracle, module, ref_feat, offset_feat, pre_offset, b):
    """Oracle value of one DynAgg call for sample b: (out [Co,H,W], offset, mask)."""
    dg, K = module.deformable_groups, 9
    head = module.conv_offset_mask
    raw = torch.nn.functional.conv2d(offset_feat[b:b + 1].double().cpu(), head.weight.double().cpu(),
                                     head.bias.double().cpu(), padding=1)[0].float().numpy()
    o1, o2, m = raw[:dg * K], raw[dg * K:2 * dg * K], raw[2 * dg * K:]
    offset = np.concatenate([o1, o2], 0)                                   # [2*dg*K, H, W]
    pre = pre_offset[b].cpu().numpy()                                      # [9, H, W, 2] (x, y)
    pre_yx = np.stack([pre[..., 1], pre[..., 0]], 1).reshape(2 * K, *pre.shape[1:3])   # (y, x) interleaved per tap
    offset = offset + np.tile(pre_yx, (dg, 1, 1))
    mask = 1.0 / (1.0 + np.exp(-m.astype(np.float64)))
    want = oracle.dcn_v2_forward(ref_feat[b:b + 1].cpu().numpy(), module.weight.detach().cpu().numpy(),
                                 module.bias.detach().cpu().numpy(), offset[None], mask[None].astype(np.float32),
                                 (1, 1), (1, 1), (1, 1), dg)[0]
    return want, offset, mask.astype(np.float32)


def _hook_dynagg(net, store, keep_grad=False):
    for stage in ("small", "medium", "large"):
        mod = getattr(net.dyn_agg_restore, f"{stage}_dyn_agg")

        def hook(m, args, out, stage=stage):
            (x, pre) = args
            fused = not torch.is_tensor(pre)          # FusedPreOffset: x[0] is a BorderedNHWC, out = lrelu(DynAgg)
            ref = x[0].interior() if fused else x[0]
            store[stage] = {"ref": ref.detach(), "feat": x[1].detach(), "pre": None if fused else pre.detach(),
                            "out": out.detach(), "lrelu": pre.lrelu_slope if fused else None}
            if keep_grad and out.requires_grad:
                out.register_hook(lambda g, stage=stage: store[stage].__setitem__("gout", g.detach()))
        mod.register_forward_hook(hook)


def _synthetic_pairs(B, h, dev, seed):
    """configs[2]-[4] style inputs: LR h x h, bicubic x4 upsampled LR, a 500x500 Ref zero-padded to the 4h canvas
    (test-time rule of ref_cufed_dataset.py:107-114; for h = 40 the Ref is 64x64 as in configs[0])."""
    import synth
    lq = torch.from_numpy(synth.uniform((B, 3, h, h), seed, 0.0, 1.0))
    up = torch.nn.functional.interpolate(lq, scale_factor=4, mode="bicubic", align_corners=False).clamp(0, 1)
    v = min(500, 4 * h) if h >= 125 else 64
    ref = np.zeros((B, 3, 4 * h, 4 * h), np.float32)
    ref[:, :, :v, :v] = synth.uniform((B, 3, v, v), seed + 1, 0.0, 1.0)
    return lq.to(dev), up.to(dev), torch.from_numpy(ref).to(dev)


def test_convolution_arithmetics_agree_end_to_end(dev):
    """The three convolution arithmetics behind the same forward -- f16 x 2 (three products, the inference default), bf16 x 3
    (six products) and the fp32-MFMA kernels -- on a configs[2]-shaped pair (LR 160, Ref 500 padded to 640): the extractor
    features agree to fp32 rounding, the index maps except for near-ties, and the decoder, handed the SAME pre-offsets and
    reference features, produces the same SR image to 2e-5 (north_star's bound is 1e-3)."""
    import c2m_amd
    ops = c2m_amd.ops
    ext, mp, g = _build_chain(dev)
    lq, up, ref = _synthetic_pairs(1, 160, dev, 4600)
    keep = (ops._SPLIT16, ops._SPLIT)
    out = {}
    try:
        for name, (s16, spl) in (("f16x2", (True, "all")), ("bf16x3", (False, "all")), ("fp32", (False, "0"))):
            ops._SPLIT16, ops._SPLIT = s16, spl
            with torch.no_grad():
                feats = ext(up, ref)
                pre, ref_feat = mp(feats, ref)
                if name == "f16x2":
                    pre0, ref0 = pre, ref_feat
                out[name] = {"f1": feats["dense_features1"].clone(), "idx": pre.max_idx.clone(), "sr_own": g(lq, pre, ref_feat),
                             "sr_same_inputs": g(lq, pre0, ref0)}
    finally:
        ops._SPLIT16, ops._SPLIT = keep
    base = out["fp32"]
    for name in ("f16x2", "bf16x3"):
        o = out[name]
        fscale = float(base["f1"].abs().max())
        assert float((o["f1"] - base["f1"]).abs().max()) < 2e-5 * fscale, name
        flips = int((o["idx"] != base["idx"]).sum())
        assert flips <= 8, (name, flips)                     # fp32 near-ties of 24 964 queries
        assert float((o["sr_same_inputs"] - base["sr_same_inputs"]).abs().max()) < 2e-5, name
    assert bool(torch.isfinite(out["f16x2"]["sr_own"]).all())


def _check_sample_against_oracle(oracle, g, feats, idx, pre, taps, b):
    """sample b of a configs[2]-shaped batch against the oracle: index map on row slices (bit-exact), the pre-offset maps
    of all three scales (bit-exact), the three DynAgg outputs at 160 / 320 / 640 (1e-4 * scale)"""
    f1 = oracle.feature_normalize(feats["dense_features1"][b].cpu().numpy())
    f2 = oracle.feature_normalize(feats["dense_features2"][b].cpu().numpy())
    hidx = idx[b].cpu().numpy()
    for rows in ((0, 3), (77, 80), (155, 158)):
        oi, _ = oracle.feature_match_index(f1, f2, 3, 1, 1, True, True, qrows=rows)
        assert np.array_equal(hidx[rows[0]:rows[1]], oi[rows[0]:rows[1]])
    o3, o2, o1 = oracle.build_pre_offsets(hidx, 160, 160)
    assert np.array_equal(pre["relu3_1"][b].cpu().numpy(), o3)
    assert np.array_equal(pre["relu2_1"][b].cpu().numpy(), o2)
    assert np.array_equal(pre["relu1_1"][b].cpu().numpy(), o1)
    for stage, key in (("small", "relu3_1"), ("medium", "relu2_1"), ("large", "relu1_1")):
        t = taps[stage]
        mod = getattr(g.dyn_agg_restore, f"{stage}_dyn_agg")
        want, _, _ = _dynagg_expected(oracle, mod, t["ref"], t["feat"], pre[key], b)
        if t["lrelu"] is not None:
            want = np.where(want > 0, want, want * np.float32(t["lrelu"]))
        got = t["out"][b].cpu().numpy()
        err = float(np.abs(got - want).max())
        assert err < 1e-4 * max(1.0, float(np.abs(want).max())), f"DynAgg {stage} at LR 160, sample {b}: {err}"


def test_cfg3_full_forward_at_batch16(dev):
    """BASELINE configs[2] AT ITS STATED BATCH OF 16: the full restoration forward (extractor ->
    correlation -> pre-offsets -> VGG taps -> RestorationNet, fused path, default arithmetic) on 16 pairs of LR 160x160 /
    Ref 500x500 (padded to 640x640); the LAST sample (15: the far end of every batch stride) is checked against the
    oracle -- index-map row slices and pre-offsets bit-exact, the three DynAgg outputs within 1e-4 * scale -- and its SR
    image equals the one the same pair gives in a batch of one (north_star's 1e-3)."""
    import c2m_oracle as oracle
    ext, mp, g = _build_chain(dev)
    lq, up, ref = _synthetic_pairs(16, 160, dev, 4100)
    taps = {}
    _hook_dynagg(g, taps)
    with torch.no_grad():
        feats = ext(up, ref)
        pre, ref_feat = mp(feats, ref)
        assert g._use_fused(lq, pre, ref_feat)
        sr = g(lq, pre, ref_feat)
        idx, _ = mp.match(feats)
    assert tuple(sr.shape) == (16, 3, 640, 640) and bool(torch.isfinite(sr).all())
    _check_sample_against_oracle(oracle, g, feats, idx, pre, taps, 15)
    with torch.no_grad():
        feats1 = ext(up[15:], ref[15:])
        pre1, ref_feat1 = mp(feats1, ref[15:])
        sr1 = g(lq[15:], pre1, ref_feat1)
    assert float((sr1[0] - sr[15]).abs().max()) < 1e-3


def test_image_boundary_parity_full_size_pair_vs_cpu_chain(dev):
    """North_star's tolerance at the IMAGE boundary as an assertion, one full-size configs[2] pair:
    the HIP path (extractor -> correlation -> pre-offsets -> VGG taps -> RestorationNet, default f16 x 2 arithmetic) against
    oracle/cpu_chain.py -- stock torch-CPU convolutions, the reference's conv2d-filter correlation (ref_map_util.py:26-86), the C
    oracle for pre-offsets and DCNv2.
      * index map: the two extractor implementations (hand-written f16 x 2 pieces vs oneDNN) may resolve an fp32 near-tie
        differently; every flip must BE one (float64 margin of the two candidates < 1e-6; measured ~1e-7), and there are at
        most 4 of 24 964 (measured 0 - 2);
      * SR image: given the GPU's index map the CPU decoder agrees to <= 1e-3 on EVERY pixel of the 640 x 640 image (measured
        ~1.2e-7): a regression that moved the decoder at full size only would fail here, not just show up in a bench line.
    ~15 s of host time on the GPU box's cores."""
    import c2m_oracle as oracle
    import cpu_chain
    oracle.set_num_threads(min(64, __import__("os").cpu_count() or 1))
    ext, mp, g = _build_chain(dev)
    lq, up, ref = _synthetic_pairs(3, 160, dev, 4700)
    with torch.no_grad():
        feats = ext(up, ref)
        pre, ref_feat = mp(feats, ref)
        assert g._use_fused(lq, pre, ref_feat)
        sr = g(lq, pre, ref_feat)
    b = 2                                                  # (not sample 0: batch strides are part of what is checked)
    idx_gpu = pre.max_idx[b:b + 1].cpu().numpy()
    sr_cpu, idx_cpu, f = cpu_chain.full_forward_cpu(ext, mp, g, lq[b:b + 1], up[b:b + 1], ref[b:b + 1], True, None, cond_idx=idx_gpu)
    flips = cpu_chain.mismatch_margins(f["dense_features1"][0], f["dense_features2"][0], idx_gpu[0], idx_cpu[0])
    assert len(flips) <= 4, flips
    assert all(abs(m[3]) < 1e-6 for m in flips), flips
    d = (sr[b].cpu() - f["sr_given_idx"][0]).abs()
    assert tuple(d.shape) == (3, 640, 640)
    assert float(d.max()) <= 1e-3, float(d.max())
    assert float(d.max()) <= 5e-5, f"decoder drifted from the CPU chain: {float(d.max())} (north_star allows 1e-3; measured 1.2e-7)"
    if not flips:                                          # same index map -> the unconditional images agree too
        assert float((sr[b].cpu() - sr_cpu[0]).abs().max()) <= 1e-3


def _random_restoration_inputs(h, dev, seed):
    """RestorationNet inputs of LR size h x h without running the extractor: a random LR image, a lazy PreOffsets over a random
    (valid) index map, N(0,1)-ish Ref features of the three tap shapes."""
    import synth
    from mmsr.models.archs.corres_generation_arch import PreOffsets
    g_ = torch.Generator(device="cpu").manual_seed(seed)
    lq = torch.from_numpy(synth.uniform((1, 3, h, h), seed, 0.0, 1.0)).to(dev)
    idx = torch.randint(0, (h - 2) * (h - 2), (1, h - 2, h - 2), generator=g_, dtype=torch.int64).to(dev)
    gen = torch.Generator(device=dev).manual_seed(seed + 1)
    feats = {k: torch.randn((1, c, s_ * h, s_ * h), generator=gen, device=dev) for k, c, s_ in
             (("relu3_1", 256, 1), ("relu2_1", 128, 2), ("relu1_1", 64, 4))}
    return lq, PreOffsets(idx, h, h), feats


def test_fused_path_size_limit_and_warning(dev):
    """the fused path's size limit is the 32-bit byte offset inside one sample's planar offset planes
    (ref_restoration_arch.py `_use_fused`): LR <= 482 x 482 at 8 deformable groups (round 5 stopped at 394 because it counted the
    mask planes into the same range).  (1) LR 400 x 400 -- beyond the old limit, the size class of WR-SR references -- stays on
    the hand-written kernels and agrees with the module-by-module path (stock convolutions + the NCHW DCNv2 operator, same
    pre-offsets) to north_star's 1e-3 (asserted at 1e-4: both are fp32-equivalent).  (2) Beyond the limit the net declines
    the fused path WITH a warning, once."""
    import warnings
    from mmsr.models.archs.ref_restoration_arch import RestorationNet
    torch.manual_seed(5)
    net = RestorationNet(64, 16, 8).eval().to(dev)
    for stage in ("small", "medium", "large"):   # live offset heads
        torch.nn.init.normal_(getattr(net.dyn_agg_restore, f"{stage}_dyn_agg").conv_offset_mask.weight, std=0.01)
    lq, pre, feats = _random_restoration_inputs(400, dev, 9100)
    with torch.no_grad():
        with warnings.catch_warnings():
            warnings.simplefilter("error")                       # no size warning at LR 400
            assert net._use_fused(lq, pre, feats)
            sr = net(lq, pre, feats)
        net.allow_fused = False
        want = net(lq, pre, feats)
        net.allow_fused = True
    assert tuple(sr.shape) == (1, 3, 1600, 1600) and bool(torch.isfinite(sr).all())
    err = float((sr - want).abs().max())
    assert err < 1e-4, err
    big = torch.zeros((1, 3, 484, 484), device=dev)
    tiny = {k: v[:, :, :8, :8] for k, v in feats.items()}       # (_use_fused looks at dtypes / channel counts only)
    with torch.no_grad():                                        # (the fused path is the no_grad path)
        with pytest.warns(RuntimeWarning, match="module by module"):
            assert not net._use_fused(big, pre, tiny)
        with warnings.catch_warnings():
            warnings.simplefilter("error")                       # once per module
            assert not net._use_fused(big, pre, tiny)
        assert net._use_fused(torch.zeros((1, 3, 482, 482), device=dev), pre, tiny)


def test_cfg3_chain_160_batch2(dev):
    """BASELINE configs[2] shape at B=2: extractor -> correlation/index map -> pre-offsets -> VGG taps -> RestorationNet at
    LR 160x160 / Ref 500x500 padded to 640x640.  Sample 1 (not 0: batch indexing) is checked against the oracle: index
    map on row slices (bit-exact), pre-offset maps at all three scales (bit-exact), the three DynAgg outputs at
    160/320/640 (1e-4 * scale), SR finite and batch-independent."""
    import c2m_oracle as oracle
    ext, mp, g = _build_chain(dev)
    lq, up, ref = _synthetic_pairs(2, 160, dev, 4000)
    taps = {}
    _hook_dynagg(g, taps)
    with torch.no_grad():
        feats = ext(up, ref)
        pre, ref_feat = mp(feats, ref)
        sr = g(lq, pre, ref_feat)
        idx, _ = mp.match(feats)
    assert tuple(sr.shape) == (2, 3, 640, 640) and bool(torch.isfinite(sr).all())
    b = 1
    f1 = oracle.feature_normalize(feats["dense_features1"][b].cpu().numpy())
    f2 = oracle.feature_normalize(feats["dense_features2"][b].cpu().numpy())
    hidx = idx[b].cpu().numpy()
    for rows in ((0, 3), (77, 80), (155, 158)):
        oi, _ = oracle.feature_match_index(f1, f2, 3, 1, 1, True, True, qrows=rows)
        assert np.array_equal(hidx[rows[0]:rows[1]], oi[rows[0]:rows[1]])
    o3, o2, o1 = oracle.build_pre_offsets(hidx, 160, 160)
    assert np.array_equal(pre["relu3_1"][b].cpu().numpy(), o3)
    assert np.array_equal(pre["relu2_1"][b].cpu().numpy(), o2)
    assert np.array_equal(pre["relu1_1"][b].cpu().numpy(), o1)
    with torch.no_grad():
        assert g._use_fused(lq, pre, ref_feat)   # this chain runs the fused channels-last path
    for stage, key in (("small", "relu3_1"), ("medium", "relu2_1"), ("large", "relu1_1")):
        t = taps[stage]
        mod = getattr(g.dyn_agg_restore, f"{stage}_dyn_agg")
        want, _, _ = _dynagg_expected(oracle, mod, t["ref"], t["feat"], pre[key], b)
        if t["lrelu"] is not None:
            want = np.where(want > 0, want, want * np.float32(t["lrelu"]))
        got = t["out"][b].cpu().numpy()
        err = float(np.abs(got - want).max())
        assert err < 1e-4 * max(1.0, float(np.abs(want).max())), f"DynAgg {stage} at LR 160: {err}"
    # the autograd-capable module-by-module path (stock convolutions, materialised pre-offset tensors) gives the same image
    with torch.no_grad():
        sr_stock = g(lq, {k: pre[k] for k in pre}, ref_feat)
    assert float((sr_stock - sr).abs().max()) < 1e-3   # north_star's bound on SR pixels
    # samples of a batch are independent: sample 1 alone gives the same SR image
    with torch.no_grad():
        feats1 = ext(up[1:], ref[1:])
        pre1, ref_feat1 = mp(feats1, ref[1:])
        sr1 = g(lq[1:], pre1, ref_feat1)
    assert float((sr1[0] - sr[1]).abs().max()) < 1e-3   # north_star's bound; conv algorithms may differ with B


def test_cfg4_per_rank_training_step(dev):
    """BASELINE configs[3], the slice one rank runs: 4 pairs, GT 160x160 -> LR 40x40, Ref 160x160, n_blocks=16, L1 loss,
    the reference's four Adam groups.  The DCNv2 weight / bias gradients of the three DynAgg layers equal the oracle's
    backward on the tensors captured at the module boundary (1e-4 * scale); every parameter gets a gradient; the loss goes
    down."""
    import c2m_oracle as oracle
    import synth
    from mmsr.models.ref_restoration_model import RefRestorationModel
    opt = {"dist": False, "gpu_ids": [0], "is_train": True, "path": {},
           "network_g": {"type": "RestorationNet", "ngf": 64, "n_blocks": 16, "groups": 8},
           "network_map": {"type": "CorrespondenceGenerationArch", "patch_size": 3, "stride": 1,
                           "vgg_layer_list": ["relu1_1", "relu2_1", "relu3_1"], "vgg_type": "vgg19"},
           "network_extractor": {"type": "ContrasExtractorSep"},
           "train": {"lr_g": 1e-4, "lr_offset": 1e-4, "lr_relu2_offset": 1e-5, "lr_relu3_offset": 1e-6,
                     "weight_decay_g": 0, "beta_g": [0.9, 0.999], "pixel_weight": 1.0}}
    torch.manual_seed(11)
    model = RefRestorationModel(opt)
    _fill(model.net_g, "net_g.")
    B, h = 4, 40
    gt = torch.from_numpy(synth.uniform((B, 3, 4 * h, 4 * h), 5000, 0.0, 1.0))
    lq = torch.nn.functional.interpolate(gt, scale_factor=0.25, mode="bicubic", align_corners=False).clamp(0, 1)
    up = torch.nn.functional.interpolate(lq, scale_factor=4, mode="bicubic", align_corners=False).clamp(0, 1)
    ref = torch.from_numpy(synth.uniform((B, 3, 4 * h, 4 * h), 5001, 0.0, 1.0))
    model.feed_data({"img_in_lq": lq, "img_ref": ref, "img_in": gt, "img_in_up": up})
    taps = {}
    _hook_dynagg(model.net_g, taps, keep_grad=True)
    # one forward/backward by hand (same calls as optimize_parameters, without the optimiser step) for the gradient check
    model._correspondence()
    out = model.net_g(model.img_in_lq, model.pre_offset, model.img_ref_feat)
    loss = model.cri_pix(out, model.gt)
    model.optimizer_g.zero_grad()
    loss.backward()
    missing = [n for n, p in model.net_g.named_parameters() if p.grad is None]
    assert not missing, missing
    for stage in ("small", "medium", "large"):
        t = taps[stage]
        mod = getattr(model.net_g.dyn_agg_restore, f"{stage}_dyn_agg")
        gw = np.zeros(tuple(mod.weight.shape), np.float32)
        gb = np.zeros(tuple(mod.bias.shape), np.float32)
        for b in range(B):
            _, offset, mask = _dynagg_expected(oracle, mod, t["ref"], t["feat"], t["pre"], b)
            g = oracle.dcn_v2_backward(t["ref"][b:b + 1].cpu().numpy(), mod.weight.detach().cpu().numpy(),
                                       mod.bias.detach().cpu().numpy(), offset[None], mask[None],
                                       t["gout"][b:b + 1].cpu().numpy(), (1, 1), (1, 1), (1, 1), mod.deformable_groups)
            gw += g[3]
            gb += g[4]
        for name, got, want in (("weight", mod.weight.grad, gw), ("bias", mod.bias.grad, gb)):
            err = float(np.abs(got.cpu().numpy() - want).max())
            assert err < 1e-4 * max(float(np.abs(want).max()), 1e-6) + 1e-9, f"{stage}_dyn_agg.{name}.grad: {err} vs scale {np.abs(want).max()}"
    losses = []
    for step in range(1, 6):
        model.optimize_parameters(step)
        losses.append(float(model.log_dict["l_g_pix"]))
    assert all(np.isfinite(losses)) and losses[-1] < losses[0], losses


def _smooth_gt(B, H, seed):
    """Synthetic ground truth with natural-image-like spectrum: a coarse random field upsampled bicubically plus a little
    fine texture, in [0, 1] -- so that a x4 restoration has a finite, realistic PSNR (~25-35 dB)."""
    import synth
    coarse = torch.from_numpy(synth.uniform((B, 3, H // 16, H // 16), seed, 0.0, 1.0))
    img = torch.nn.functional.interpolate(coarse, size=(H, H), mode="bicubic", align_corners=False)
    img = img + 0.03 * torch.from_numpy(synth.gaussish((B, 3, H, H), seed + 1))
    return img.clamp(0, 1)


def test_cfg5_320_bf16_autocast_psnr_budget(dev):
    """BASELINE configs[4]: CUFED5-shape inference at LR 320x320 / Ref 500x500 (zero-padded to 1280x1280) under bf16
    autocast.  Criterion as the config states it: PSNR against the ground truth, computed as the reference's validation
    does (tensor2img + metrics.psnr on [0,255] images, crop_border = scale = 4, also on Y -- ref_restoration_model.py:
    338-347, options.py:56-57), must stay within 0.02 dB of the fp32 run.  The fp32 index map is also checked against the
    oracle on row slices at this (largest) size."""
    import c2m_oracle as oracle
    from mmsr.utils import metrics
    ext, mp, g = _build_chain(dev)
    H = 1280
    gt = _smooth_gt(1, H, 6000).to(dev)
    lq = torch.nn.functional.interpolate(gt, scale_factor=0.25, mode="bicubic", align_corners=False).clamp(0, 1)
    up = torch.nn.functional.interpolate(lq, scale_factor=4, mode="bicubic", align_corners=False).clamp(0, 1)
    ref = torch.zeros((1, 3, H, H), device=dev)
    ref[:, :, :500, :500] = _smooth_gt(1, 512, 6002)[:, :, :500, :500].to(dev)

    def run():
        feats = ext(up, ref)
        pre, ref_feat = mp(feats, ref)
        return g(lq, pre, ref_feat), feats

    with torch.no_grad():
        sr32, feats = run()
        idx, _ = mp.match(feats)
        with torch.autocast("cuda", dtype=torch.bfloat16):
            sr16, _ = run()
    assert tuple(sr32.shape) == (1, 3, H, H) and torch.isfinite(sr16).all() and torch.isfinite(sr32).all()
    f1 = oracle.feature_normalize(feats["dense_features1"][0].cpu().numpy())
    f2 = oracle.feature_normalize(feats["dense_features2"][0].cpu().numpy())
    hidx = idx[0].cpu().numpy()
    assert hidx.shape == (318, 318)
    for rows in ((0, 2), (200, 202), (316, 318)):
        oi, _ = oracle.feature_match_index(f1, f2, 3, 1, 1, True, True, qrows=rows)
        assert np.array_equal(hidx[rows[0]:rows[1]], oi[rows[0]:rows[1]])
    m32 = metrics.validation_metrics(sr32, gt, crop_border=4)
    m16 = metrics.validation_metrics(sr16.float(), gt, crop_border=4)
    p32, p16 = float(m32["psnr"][0]), float(m16["psnr"][0])
    y32, y16 = float(m32["psnr_y"][0]), float(m16["psnr_y"][0])
    assert 15.0 < p32 < 60.0, p32   # a real restoration quality, not a degenerate image
    assert abs(p16 - p32) <= 0.02, f"PSNR fp32 {p32:.4f} dB vs bf16 {p16:.4f} dB"
    assert abs(y16 - y32) <= 0.02, f"PSNR_Y fp32 {y32:.4f} dB vs bf16 {y16:.4f} dB"


def test_cfg5_dynagg_layers_vs_oracle_and_bf16_residual(dev):
    """BASELINE configs[4] shapes (LR 320x320: DynAgg layers at 320^2 x 256, 640^2 x 128, 1280^2 x 64), VERDICT r2 3(b):
    (1) the three DynAgg outputs of the fp32 forward against the oracle DCNv2 on the FULL maps, from the offsets / masks
    the head kernel produces for the same inputs (2e-5 * scale, the DCNv2 tests' tolerance);
    (2) the bf16-autocast decoder (single-piece bf16 convolutions, same index map and Ref features) against the fp32 one on
    the restoration RESIDUAL sr - bilinear(lq), which is what the network computes: relative 2-norm error within 3e-2 (bf16
    has 8 mantissa bits: ~4e-3 per rounding, accumulated over ~100 layers), where a PSNR-vs-GT budget alone would pass with a
    badly wrong residual."""
    import c2m_amd
    import c2m_oracle as oracle
    ops = c2m_amd.ops
    ext, mp, g = _build_chain(dev)
    H = 1280
    gt = _smooth_gt(1, H, 6100).to(dev)
    lq = torch.nn.functional.interpolate(gt, scale_factor=0.25, mode="bicubic", align_corners=False).clamp(0, 1)
    up = torch.nn.functional.interpolate(lq, scale_factor=4, mode="bicubic", align_corners=False).clamp(0, 1)
    ref = torch.zeros((1, 3, H, H), device=dev)
    ref[:, :, :500, :500] = _smooth_gt(1, 512, 6102)[:, :, :500, :500].to(dev)
    for stage in ("small", "medium", "large"):   # live offset heads (the golden fill zeroes nothing, but make it explicit)
        head = getattr(g.dyn_agg_restore, f"{stage}_dyn_agg").conv_offset_mask
        assert float(head.weight.abs().max()) > 0
    cap = {}
    hooks = [getattr(g.dyn_agg_restore, f"{s}_dyn_agg").register_forward_hook(
        lambda m, i, o, s=s: cap.__setitem__(s, (i[0][0], i[0][1], i[1], o))) for s in ("small", "medium", "large")]
    with torch.no_grad():
        feats = ext(up, ref)
        pre, ref_feat = mp(feats, ref)
        assert g._use_fused(lq, pre, ref_feat)
        sr32 = g(lq, pre, ref_feat)
        cap32 = dict(cap)
        with torch.autocast("cuda", dtype=torch.bfloat16):
            assert g._use_fused(lq, pre, ref_feat)
            sr16 = g(lq, pre, ref_feat)
    for h_ in hooks:
        h_.remove()
    oracle.set_num_threads(64)
    for stage in ("small", "medium", "large"):
        refb, feat, fpre, out = cap32[stage]
        m = getattr(g.dyn_agg_restore, f"{stage}_dyn_agg")
        with torch.no_grad():
            off, msk = ops.conv3x3_dcn_head(feat, m.conv_offset_mask.weight, m.conv_offset_mask.bias, 8, fpre.flow, fpre.scale)
        x = refb.interior().contiguous().cpu().numpy()
        want = oracle.dcn_v2_forward(x, m.weight.detach().cpu().numpy(), m.bias.detach().cpu().numpy(), off.cpu().numpy(),
                                     msk.cpu().numpy(), (1, 1), (1, 1), (1, 1), 8)
        got = out.contiguous().cpu().numpy()
        got = np.where(got > 0, got, got / np.float32(0.1))      # the fused path folds LeakyReLU(0.1) into the warp
        err = float(np.abs(got - want).max())
        assert err < 2e-5 * max(1.0, float(np.abs(want).max())), (stage, err)
        del x, want, got, off, msk
    base = torch.nn.functional.interpolate(lq, None, 4, "bilinear", False)
    r32, r16 = sr32 - base, sr16.float() - base
    rel = float((r16 - r32).norm() / r32.norm())
    assert float(r32.abs().max()) > 1e-3            # the residual is not degenerate
    assert rel < 3e-2, rel


def _train_opt(dist_on):
    return {"dist": dist_on, "gpu_ids": [0], "is_train": True, "path": {},
            "network_g": {"type": "RestorationNet", "ngf": 64, "n_blocks": 2, "groups": 8},
            "network_map": {"type": "CorrespondenceGenerationArch", "patch_size": 3, "stride": 1,
                            "vgg_layer_list": ["relu1_1", "relu2_1", "relu3_1"], "vgg_type": "vgg19"},
            "network_extractor": {"type": "ContrasExtractorSep"},
            "train": {"lr_g": 1e-4, "lr_offset": 1e-4, "lr_relu2_offset": 1e-5, "lr_relu3_offset": 1e-6,
                      "weight_decay_g": 0, "beta_g": [0.9, 0.999], "pixel_weight": 1.0}}


def _train_batch(B=2, h=16, seed=4000):
    import synth
    gt = torch.from_numpy(synth.uniform((B, 3, 4 * h, 4 * h), seed, 0.0, 1.0))
    lq = torch.nn.functional.interpolate(gt, scale_factor=0.25, mode="bicubic", align_corners=False).clamp(0, 1)
    up = torch.nn.functional.interpolate(lq, scale_factor=4, mode="bicubic", align_corners=False).clamp(0, 1)
    ref = torch.from_numpy(synth.uniform((B, 3, 4 * h, 4 * h), seed + 1, 0.0, 1.0))
    return {"img_in_lq": lq, "img_ref": ref, "img_in": gt, "img_in_up": up}


def test_ddp_nccl_wrapped_net_g_step_and_test_match_the_bare_module(dev):
    """VERDICT r2 item 2 / ADVICE r2 (high): the REAL net_g (DCNv2 autograd function, live offset heads) wrapped by
    model_to_device in DistributedDataParallel over RCCL (world size 1 on this box, gradient_as_bucket_view=True), fed the
    lazy PreOffsets dict through DDP's input scatter: one optimize_parameters() step gives the bare module's gradients, and
    test() -- the fused inference path behind the wrapper -- the bare module's output."""
    import socket
    import torch.distributed as dist
    from mmsr.models.base_model import unwrap
    from mmsr.models.ref_restoration_model import RefRestorationModel
    with socket.socket() as s_:
        s_.bind(("127.0.0.1", 0))
        port = s_.getsockname()[1]
    import mmsr.models.archs.ref_restoration_arch as arch
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1, device_id=dev)
    arch._TRAIN_KERNELS = "1"   # the hand-written (deterministic) convolution kernels at this small size too: bare and wrapped
    try:                        # runs must then agree to rounding (two MIOpen runs of one net need not)
        torch.manual_seed(11)
        bare = RefRestorationModel(_train_opt(False))
        for stage in ("small", "medium", "large"):
            torch.nn.init.normal_(getattr(unwrap(bare.net_g).dyn_agg_restore, f"{stage}_dyn_agg").conv_offset_mask.weight, std=0.01)
        wrapped = RefRestorationModel(_train_opt(True))
        assert isinstance(wrapped.net_g, torch.nn.parallel.DistributedDataParallel)
        for dst, src in ((wrapped.net_g.module, bare.net_g), (wrapped.net_map, bare.net_map), (wrapped.net_extractor, bare.net_extractor)):
            dst.load_state_dict(unwrap(src).state_dict())
        data = _train_batch()
        for m in (bare, wrapped):
            m.feed_data(data)
        out_b, out_w = bare.test(), wrapped.test()
        assert float((out_b - out_w).abs().max()) < 1e-6
        bare.optimize_parameters(1)
        wrapped.optimize_parameters(1)
        torch.cuda.synchronize()
        assert abs(float(wrapped.log_dict["l_g_pix"]) - float(bare.log_dict["l_g_pix"])) < 1e-6
        n = 0
        for (k, pb), (_, pw) in zip(unwrap(bare.net_g).named_parameters(), wrapped.net_g.module.named_parameters()):
            if pb.grad is None:
                assert pw.grad is None, k
                continue
            scale = max(1e-12, float(pb.grad.abs().max()))
            assert float((pb.grad - pw.grad).abs().max()) <= 1e-5 * scale, k    # (DCNv2 weight grads use fp32 atomics)
            n += 1
        assert n > 50
    finally:
        arch._TRAIN_KERNELS = "auto"
        dist.destroy_process_group()


def test_hip_graph_training_step_matches_the_eager_step(dev):
    """train.hip_graph: the captured-and-replayed training step (2 eager warm-up steps, capture, replays; static input buffers
    refilled by feed_data) walks the same trajectory as the eager loop: per-step losses and the parameters after 6 steps on 6
    different batches agree to the run-to-run noise of the fp32 atomics in the DCNv2 weight gradients."""
    from mmsr.models.base_model import unwrap
    from mmsr.models.ref_restoration_model import RefRestorationModel
    import mmsr.models.archs.ref_restoration_arch as arch
    arch._TRAIN_KERNELS = "1"   # hand-written convolution kernels (deterministic) in both runs
    try:
        torch.manual_seed(12)
        eager = RefRestorationModel(_train_opt(False))
        for stage in ("small", "medium", "large"):
            torch.nn.init.normal_(getattr(unwrap(eager.net_g).dyn_agg_restore, f"{stage}_dyn_agg").conv_offset_mask.weight, std=0.01)
        opt = _train_opt(False)
        opt["train"]["hip_graph"] = True
        graphed = RefRestorationModel(opt)
        assert graphed._graph_on and not eager._graph_on
        for dst, src in ((graphed.net_g, eager.net_g), (graphed.net_map, eager.net_map), (graphed.net_extractor, eager.net_extractor)):
            dst.load_state_dict(src.state_dict())
        losses = []
        for step in range(1, 7):
            data = _train_batch(B=2, h=16, seed=4200 + 10 * step)
            for m in (eager, graphed):
                m.feed_data(data)
                m.optimize_parameters(step)
            torch.cuda.synchronize()
            losses.append((float(eager.log_dict["l_g_pix"]), float(graphed.log_dict["l_g_pix"])))
        assert graphed._graph is not None
        for le, lg in losses:
            assert abs(le - lg) <= 1e-4 * abs(le), losses
        assert len({round(le, 6) for le, _ in losses}) > 3          # the batches really differed
        # parameters: Adam normalises every gradient entry, so entries whose gradient is atomics-noise can walk apart by up to
        # 2 * lr per step between ANY two runs; the bulk must coincide
        worst, moved, same = 0.0, 0, 0
        for (k, pe), (_, pg) in zip(eager.net_g.named_parameters(), graphed.net_g.named_parameters()):
            d = (pe.detach() - pg.detach()).abs()
            worst = max(worst, float(d.max()))
            moved += int((d > 1e-5).sum())
            same += d.numel()
        assert worst <= 2 * 6 * 1e-4 + 1e-6, worst
        assert moved < 0.02 * same, (moved, same)
        # a new geometry drops the graph and captures again
        graphed.feed_data(_train_batch(B=1, h=16, seed=4300))
        assert graphed._graph is None
        for step in range(7, 11):
            graphed.optimize_parameters(step)
        torch.cuda.synchronize()
        assert graphed._graph is not None and bool(torch.isfinite(graphed.log_dict["l_g_pix"]))
        sr = graphed.test()                                         # the fused inference path still runs beside the graph
        assert bool(torch.isfinite(sr).all())
        # ... and a validation between two replays (test() re-points `output`, the reference's loop then deletes it) leaves the
        # next replay's result where the trainer looks for it: `output` is the captured step's tensor again
        captured = graphed._graph_output
        del graphed.output
        graphed.optimize_parameters(11)
        torch.cuda.synchronize()
        assert graphed.output is captured and graphed.output.requires_grad and bool(torch.isfinite(graphed.output).all())
    finally:
        arch._TRAIN_KERNELS = "auto"


def test_dataparallel_scatter_keeps_the_fused_path(dev):
    """nn.DataParallel (the reference's default without a launcher, base_model.py:73-74) rebuilds dict inputs per replica:
    PreOffsets must come out as PreOffsets (sliced index map), so the replica still takes the fused path."""
    ext, mp, g = _build_chain(dev)
    data = _train_batch(B=2, h=24, seed=4100)
    lq, up, ref = (data[k].to(dev) for k in ("img_in_lq", "img_in_up", "img_ref"))
    dp = torch.nn.DataParallel(g, device_ids=[0])
    with torch.no_grad():
        feats = ext(up, ref)
        pre, ref_feat = mp(feats, ref)
        want = g(lq, pre, ref_feat)
        got = dp(lq, pre, ref_feat)
    assert list(pre.keys()) == ["max_idx"]             # nothing was materialised on the way
    assert float((got - want).abs().max()) < 1e-6


def test_training_path_on_hand_written_kernels_matches_stock_modules(dev):
    """Gradients enabled: RestorationNet runs its 3x3 convolutions forward and backward on the hand-written kernels
    (ops.conv3x3_autograd).  Same weights, same inputs: output and every parameter gradient agree with the stock
    module-by-module path (allow_fused = False: MIOpen convolutions) to the accuracy of the latter."""
    import c2m_amd
    ext, mp, g = _build_chain(dev)
    for stage in ("small", "medium", "large"):
        torch.nn.init.normal_(getattr(g.dyn_agg_restore, f"{stage}_dyn_agg").conv_offset_mask.weight, std=0.01)
    data = _train_batch(B=2, h=24, seed=4200)
    lq, up, ref, gt = (data[k].to(dev) for k in ("img_in_lq", "img_in_up", "img_ref", "img_in"))
    with torch.no_grad():
        feats = ext(up, ref)
        pre, ref_feat = mp(feats, ref)

    import mmsr.models.archs.ref_restoration_arch as arch
    arch._TRAIN_KERNELS = "1"     # (auto would pick the stock modules at this small size)

    def run(fused):
        g.allow_fused = fused
        g.zero_grad(set_to_none=True)
        assert g._use_train_kernels(lq, ref_feat) == fused
        c2m_amd.profile_enable(True)
        c2m_amd.profile_collect()
        out = g(lq, pre, ref_feat)
        (out - gt).abs().mean().backward()
        torch.cuda.synchronize()
        names = {n for n, _ in c2m_amd.profile_collect(capacity=65536)}
        c2m_amd.profile_enable(False)
        return out.detach(), {k: p.grad.detach().clone() for k, p in g.named_parameters() if p.grad is not None}, names

    try:
        out_s, grads_s, names_s = run(False)
        out_k, grads_k, names_k = run(True)
    finally:
        arch._TRAIN_KERNELS = "auto"
        g.allow_fused = True
    assert {"conv3x3_split", "conv3x3_wgrad"} <= names_k and "conv3x3_wgrad" not in names_s
    assert float((out_k - out_s).abs().max()) < 1e-4
    assert grads_k.keys() == grads_s.keys() and len(grads_k) > 200
    worst = 0.0
    for k in grads_s:
        scale = max(1e-9, float(grads_s[k].abs().max()))
        worst = max(worst, float((grads_k[k] - grads_s[k]).abs().max()) / scale)
    # the hand-written backward is held to 1e-5 * scale against float64 per operator (tests/test_conv_gpu.py); the stock path
    # (MIOpen Winograd forward / backward, a different summation order through ~100 layers) sits 5e-3 away on its worst parameter
    assert worst < 1e-2, worst
