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
    """BASELINE config 5 runs inference under bf16 autocast.  The plain convolutions then compute in bf16; the hot path
    takes their outputs as float32 (custom_fwd(cast_inputs=float32) / .float()) and the SR image must stay close to the
    fp32 one: PSNR of the difference far above the 0.02 dB budget the config allows on a ~30 dB restoration."""
    ext, mp, g = _build_chain(dev)
    gen = torch.Generator(device=dev).manual_seed(5)
    lq = torch.rand((1, 3, 40, 40), generator=gen, device=dev)
    up = torch.nn.functional.interpolate(lq, scale_factor=4, mode="bicubic", align_corners=False)
    ref = torch.zeros((1, 3, 160, 160), device=dev)
    ref[:, :, :64, :64] = torch.rand((1, 3, 64, 64), generator=gen, device=dev)

    def run():
        feats = ext(up, ref)
        pre, ref_feat = mp(feats, ref)
        return g(lq, pre, ref_feat)

    with torch.no_grad():
        sr32 = run()
        with torch.autocast("cuda", dtype=torch.bfloat16):
            sr16 = run()
    assert torch.isfinite(sr16).all()
    mse = float(((sr16.float() - sr32) ** 2).mean())
    assert mse < 1e-3, f"bf16-autocast SR deviates from fp32: mse {mse}"   # >= 30 dB on a [0, 1] signal


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
