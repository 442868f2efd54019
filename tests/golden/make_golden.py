#!/usr/bin/env python3
"""Generates tests/golden/*.npz by running the REFERENCE's own Python on deterministic inputs.

Runs only in the build container (needs /root/reference).  Nothing of the reference is copied: this script imports
`mmsr/models/archs/ref_map_util.py` by path and -- with tiny stand-ins for the third-party packages that are not
installed here (mmcv.scandir, torchvision's VGG layer layout) -- `corres_generation_arch.py`, runs them on CPU and
stores inputs' seeds + the outputs.  The fixtures are data (arrays), committed next to this script.

    python tests/golden/make_golden.py [small|full160|cfg5|metrics|all]     (default: small = the round-1 fixtures)
"""
import importlib.util
import os
import sys
import types

import numpy as np
import torch
import torch.nn as nn

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(REPO, "oracle"))
import synth  # noqa: E402
import c2m_oracle as oracle  # noqa: E402  (only for the deterministic feature normalisation of the INPUTS)


def load_ref_map_util():
    spec = importlib.util.spec_from_file_location("ref_map_util_reference", f"{REF}/mmsr/models/archs/ref_map_util.py")
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def stub_third_party():
    """Stand-ins for packages the reference imports but the container lacks (mmcv, torchvision, cv2)."""
    mmcv = types.ModuleType("mmcv")
    mmcv.scandir = lambda d: sorted(os.listdir(d))
    runner = types.ModuleType("mmcv.runner")
    runner.master_only = lambda f: f
    runner.get_dist_info = lambda: (0, 1)
    runner.get_time_str = lambda: "t"
    runner.init_dist = lambda *a, **k: None
    mmcv.runner = runner
    sys.modules.update({"mmcv": mmcv, "mmcv.runner": runner, "cv2": types.ModuleType("cv2")})
    cfg = {"vgg16": [64, 64, "M", 128, 128, "M", 256, 256, 256, "M", 512, 512, 512, "M", 512, 512, 512, "M"],
           "vgg19": [64, 64, "M", 128, 128, "M", 256, 256, 256, 256, "M", 512, 512, 512, 512, "M", 512, 512, 512, 512, "M"]}

    def ctor(name):
        def make(pretrained=False):
            layers, c = [], 3
            for v in cfg[name]:
                if v == "M":
                    layers.append(nn.MaxPool2d(2, 2))
                else:
                    layers += [nn.Conv2d(c, v, 3, padding=1), nn.ReLU(inplace=True)]
                    c = v
            m = nn.Module()
            m.features = nn.Sequential(*layers)
            return m
        return make
    tv, tvm, tvv, tvu = (types.ModuleType(n) for n in ("torchvision", "torchvision.models", "torchvision.models.vgg", "torchvision.utils"))
    tvv.vgg16, tvv.vgg19 = ctor("vgg16"), ctor("vgg19")
    tv.models, tvm.vgg, tv.utils, tvu.make_grid = tvm, tvv, tvu, None
    sys.modules.update({"torchvision": tv, "torchvision.models": tvm, "torchvision.models.vgg": tvv, "torchvision.utils": tvu})
    # `_ext` (the reference's CUDA-only pybind module, DCNv2/src/vision.cpp:3-9): the reference has no CPU DCNv2, so
    # the stand-in forwards to the C oracle.  Golden SR outputs are therefore "reference Python + oracle DCNv2".
    ext = types.ModuleType("_ext")

    def dcn_v2_forward(inp, weight, bias, offset, mask, kh, kw, sh, sw, ph, pw, dh, dw, dg):
        o = oracle.dcn_v2_forward(inp.detach().numpy(), weight.detach().numpy(), bias.detach().numpy(),
                                  offset.detach().numpy(), mask.detach().numpy(), (sh, sw), (ph, pw), (dh, dw), dg)
        return torch.from_numpy(o)

    def dcn_v2_backward(inp, weight, bias, offset, mask, grad_out, kh, kw, sh, sw, ph, pw, dh, dw, dg):
        g = oracle.dcn_v2_backward(inp.detach().numpy(), weight.detach().numpy(), bias.detach().numpy(),
                                   offset.detach().numpy(), mask.detach().numpy(), grad_out.detach().numpy(),
                                   (sh, sw), (ph, pw), (dh, dw), dg)
        return [torch.from_numpy(a) for a in g]
    ext.dcn_v2_forward, ext.dcn_v2_backward = dcn_v2_forward, dcn_v2_backward
    sys.modules["_ext"] = ext


def norm_feat(shape, seed):
    return oracle.feature_normalize(synth.gaussish(shape, seed))


def top2_gap(rmu, fi, fr, patch, s_in, s_ref):
    """fp64 margin between best and second-best score per query (to classify any future mismatch)."""
    import torch.nn.functional as F
    q = F.unfold(torch.from_numpy(fi)[None].double(), patch, stride=s_in)[0]
    r = F.unfold(torch.from_numpy(fr)[None].double(), patch, stride=s_ref)[0]
    r = r / (r.norm(dim=0) + 1e-5)
    t = (r.t() @ q).topk(2, dim=0).values
    return float((t[0] - t[1]).min())


def corr_cases():
    # name, C, (hq,wq), (hr,wr), patch, in_stride, ref_stride, builder
    return [
        ("c256_10v16", 256, (10, 10), (16, 16), 3, 1, 1, None),
        ("c256_20v20", 256, (20, 20), (20, 20), 3, 1, 1, None),
        ("c256_40v40", 256, (40, 40), (40, 40), 3, 1, 1, None),
        ("c64_19v35x33", 64, (19, 23), (35, 33), 3, 1, 1, None),
        ("c256_21v21_s2", 256, (21, 21), (21, 21), 3, 2, 2, None),
        ("c32_12v14_p5", 32, (12, 12), (14, 14), 5, 1, 1, None),
        ("tie_halfcopy", 256, (12, 12), (12, 20), 3, 1, 1, "halfcopy"),
        ("tie_const", 256, (12, 12), (16, 16), 3, 1, 1, "const"),
        ("tie_zero", 64, (8, 8), (9, 9), 3, 1, 1, "zero"),
    ]


def build_inputs(name, C, hq, hr, builder, seed):
    fi = norm_feat((C,) + hq, seed)
    fr = norm_feat((C,) + hr, seed + 1)
    if builder == "halfcopy":      # right half of the ref = copy of the left half -> exact ties, lowest index wins
        fr[:, :, hr[1] // 2:] = fr[:, :, :hr[1] // 2]
    elif builder == "const":       # a constant block (like zero-padded test images after a conv stack)
        fr[:, 6:, :] = fr[:, 6:7, 0:1]
        fi[:, 7:, :] = fr[:, 6:7, 0:1]   # queries inside the same constant block: many exactly-equal maxima
    elif builder == "zero":
        fr[:] = 0.0
    return np.ascontiguousarray(fi), np.ascontiguousarray(fr)


def fill_parameters(module, prefix):
    """Deterministic, name-keyed parameter values (shared with the tests): conv weights ~0.02*g, offset heads 0.01*g
    (non-zero so the learned offsets are live, unlike the zero-initialised reference state), biases 0.01*g."""
    sd = module.state_dict()
    for name, t in sd.items():
        if name.endswith(("mean", "std")):
            continue
        g = synth.gaussish(tuple(t.shape), synth.name_seed(prefix + name))
        scale = 0.01 if ("conv_offset_mask" in name or name.endswith("bias")) else 0.02
        sd[name] = torch.from_numpy((g * scale).astype(np.float32))
    module.load_state_dict(sd)


def restoration_index_map(B, h, w):
    """The hashed arg-max index map [B, h-2, w-2] the restoration fixture's pre-offsets are built from."""
    hp, wp = h - 2, w - 2
    return (synth.uniform((B, hp, wp), 1001, 0.0, 1.0).astype(np.float64) * (hp * wp)).astype(np.int64) % (hp * wp)


def restoration_inputs(B, h, w):
    """LR image, pre-offset dict and ref-feature dict for RestorationNet at LR size h x w."""
    lr = synth.uniform((B, 3, h, w), 1000, 0.0, 1.0)
    idx = restoration_index_map(B, h, w)
    offs = [oracle.build_pre_offsets(idx[b], h, w) for b in range(B)]
    pre = {"relu3_1": np.stack([o[0] for o in offs]), "relu2_1": np.stack([o[1] for o in offs]),
           "relu1_1": np.stack([o[2] for o in offs])}
    feats = {"relu3_1": np.maximum(synth.gaussish((B, 256, h, w), 1002), 0),
             "relu2_1": np.maximum(synth.gaussish((B, 128, 2 * h, 2 * w), 1003), 0),
             "relu1_1": np.maximum(synth.gaussish((B, 64, 4 * h, 4 * w), 1004), 0)}
    return lr, pre, feats


def full160_inputs(b):
    """Pair b of the BASELINE configs[1]/[2] shape: 256 x 160 x 160 normalised features; the ref map carries the constant
    band a zero-padded 500x500 Ref leaves beyond 125/160 (ref_cufed_dataset.py:107-114) and a block of queries matches
    that band exactly (thousands of exactly-equal maxima).  At this size the reference runs its TWO-chunk path
    (batch_size = int(1024**2 * 512 / (160*160)) = 20971 < 24964 patches, ref_map_util.py:54-76)."""
    C, h = 256, 160
    fi = oracle.feature_normalize(synth.gaussish((C, h, h), 91 + 10 * b))
    raw = synth.gaussish((C, h, h), 92 + 10 * b)
    raw[:, 125:, :] = raw[:, 125:126, 125:126]
    raw[:, :, 125:] = raw[:, 125:126, 125:126]
    fr = oracle.feature_normalize(raw)
    fi[:, 150:, 150:] = fr[:, 130:131, 130:131]
    return np.ascontiguousarray(fi), np.ascontiguousarray(fr)


FULL160_PAIRS = 16
NEAR_TIE_MARGIN = 2e-6   # fp32 rounding uncertainty of a 2304-term dot product of this magnitude (sqrt(n) * 2^-24 * sum|a.b|)


def score_fp64(fi, fr, y, x, n):
    """Normalised correlation of query patch (y, x) with ref patch n in float64 (ref_map_util.py:62-69 in exact-ish arithmetic)."""
    wrp = fr.shape[2] - 2
    q = torch.from_numpy(np.ascontiguousarray(fi[:, y:y + 3, x:x + 3])).double().reshape(-1)
    ry, rx = divmod(int(n), wrp)
    r = torch.from_numpy(np.ascontiguousarray(fr[:, ry:ry + 3, rx:rx + 3])).double().reshape(-1)
    return float((r / (r.norm() + 1e-5) * q).sum())


def near_ties(b, fi, fr, ref_idx):
    """Queries where the reference's fp32 arg-max is NOT determined beyond fp32 rounding: the canonical-order oracle picks
    another candidate and the two scores differ by less than NEAR_TIE_MARGIN in float64.  Rows: (pair, y, x, reference
    index, other index, fp64 score(other) - fp64 score(reference) in units of 1e-12).  A disagreement with a LARGER margin
    would be a real defect and aborts the generation."""
    oi, _ = oracle.feature_match_index(fi, fr, 3, 1, 1, True, True)
    rows = []
    for (y, x) in np.argwhere(oi != ref_idx):
        gap = score_fp64(fi, fr, y, x, oi[y, x]) - score_fp64(fi, fr, y, x, ref_idx[y, x])
        assert abs(gap) < NEAR_TIE_MARGIN, (b, y, x, gap)
        rows.append((b, int(y), int(x), int(ref_idx[y, x]), int(oi[y, x]), int(round(gap * 1e12))))
        print("near tie: pair", b, "query", (int(y), int(x)), "reference", int(ref_idx[y, x]), "oracle", int(oi[y, x]),
              "fp64 gap", gap, flush=True)
    return rows


def make_full160():
    """Reference index maps (uint16: Nr = 24964 < 65536) of 16 distinct full-size pairs + values of pair 0 + the list of
    fp32-indeterminate near-ties (see near_ties)."""
    rmu = load_ref_map_util()
    torch.set_num_threads(os.cpu_count())
    out, ties = {}, []
    path = os.path.join(HERE, "corr_full160_golden.npz")
    have = dict(np.load(path)) if (os.path.exists(path) and "--reuse-reference" in sys.argv) else {}
    for b in range(FULL160_PAIRS):
        fi, fr = full160_inputs(b)
        if f"idx{b}" in have:   # reference output from an earlier run of this script (30 s per pair); only re-classify
            out[f"idx{b}"] = have[f"idx{b}"]
            if b == 0:
                out["val0"] = have["val0"]
        else:
            idx, val = rmu.feature_match_index(torch.from_numpy(fi), torch.from_numpy(fr), patch_size=3, input_stride=1,
                                               ref_stride=1, is_norm=True, norm_input=True)
            assert int(idx.max()) < 65536
            out[f"idx{b}"] = idx.numpy().astype(np.uint16)
            if b == 0:
                out["val0"] = val.numpy().astype(np.float32)
        ties += near_ties(b, fi, fr, out[f"idx{b}"].astype(np.int64))
        print("full160 pair", b, "idx range", int(out[f"idx{b}"].min()), int(out[f"idx{b}"].max()), flush=True)
    out["near_ties"] = np.array(ties, np.int64).reshape(-1, 6)
    np.savez_compressed(path, **out)


def check_against_reference_golden(got, ref_idx, ties, b, what):
    """Shared by the tests: `got` (index map of pair b) equals the reference's map everywhere except at the listed
    fp32-indeterminate near-ties, where it must be one of the two fp64-equivalent candidates."""
    ref_idx = ref_idx.astype(np.int64)
    bad = np.argwhere(got != ref_idx)
    allowed = {(int(t[1]), int(t[2])): int(t[4]) for t in ties if int(t[0]) == b}
    for (y, x) in bad:
        assert (int(y), int(x)) in allowed and int(got[y, x]) == allowed[(int(y), int(x))], \
            f"{what}: pair {b} query ({y},{x}): got {got[y, x]}, reference {ref_idx[y, x]} (not a listed near-tie)"
    return len(bad)


CFG5_ROWS = ((0, 6), (157, 163), (314, 320))   # query pixel-row slices (each gives 4 query patch rows)


def cfg5_inputs():
    """BASELINE configs[4] shape: 256 x 320 x 320 features, 500x500 Ref zero-padded to 1280x1280 -> valid 125/320."""
    C, h = 256, 320
    fi = oracle.feature_normalize(synth.gaussish((C, h, h), 591))
    raw = synth.gaussish((C, h, h), 592)
    raw[:, 125:, :] = raw[:, 125:126, 125:126]
    raw[:, :, 125:] = raw[:, 125:126, 125:126]
    fr = oracle.feature_normalize(raw)
    fi[:, 316:, 300:] = fr[:, 200:201, 200:201]
    return np.ascontiguousarray(fi), np.ascontiguousarray(fr)


def make_cfg5():
    """Reference index maps for three slices of query rows at the configs[4] feature size (Nq = Nr = 101124): the
    reference function run on feat_input[:, r0:r1] against the FULL ref map (one row slice costs ~1/50 of the pair)."""
    rmu = load_ref_map_util()
    torch.set_num_threads(os.cpu_count())
    fi, fr = cfg5_inputs()
    out = {}
    for (r0, r1) in CFG5_ROWS:
        idx, val = rmu.feature_match_index(torch.from_numpy(np.ascontiguousarray(fi[:, r0:r1])), torch.from_numpy(fr),
                                           patch_size=3, input_stride=1, ref_stride=1, is_norm=True, norm_input=True)
        out[f"idx_{r0}"] = idx.numpy().astype(np.int64)
        out[f"val_{r0}"] = val.numpy().astype(np.float32)
        print("cfg5 rows", r0, r1, "idx range", int(idx.min()), int(idx.max()), flush=True)
    np.savez_compressed(os.path.join(HERE, "corr_cfg5_golden.npz"), **out)


def metric_images(k):
    """Two float images in the layout tensor2img produces (H x W x 3, BGR, integer-valued floats in [0, 255])."""
    a = np.round(synth.uniform((37 + k, 41, 3), 7000 + k, 0.0, 255.0)).astype(np.float32)
    noise = synth.gaussish((37 + k, 41, 3), 7100 + k) * (2.0 + 3.0 * k)
    b = np.clip(np.round(a + noise), 0, 255).astype(np.float32)
    return a, b


def make_metrics():
    """PSNR / PSNR_Y / SSIM_Y of the reference's mmsr/utils/metrics.py (psnr :34-66, ssim :69-143, bgr2ycbcr :146-168) on
    synthetic image pairs, computed exactly as nondist_validation does (ref_restoration_model.py:338-351).  cv2 is absent
    here: `filter2D(img, -1, window)[5:-5, 5:-5]` -- the only cv2 call whose result is used away from the image border --
    is stood in for by scipy's correlate (the 11x11 Gaussian window is symmetric; the crop removes every border pixel)."""
    import scipy.ndimage
    cv2 = types.ModuleType("cv2")

    def gk(n, sigma):
        x = np.arange(n, dtype=np.float64) - (n - 1) / 2
        k = np.exp(-(x * x) / (2 * sigma * sigma))
        return (k / k.sum()).reshape(n, 1)
    cv2.getGaussianKernel = gk

    def filter2d(img, ddepth, window):   # the reference hands over (H, W, 1) arrays after its crop (metrics.py:135-136)
        flat = img.reshape(img.shape[0], img.shape[1])
        return scipy.ndimage.correlate(flat, window, mode="reflect").reshape(img.shape)
    cv2.filter2D = filter2d
    sys.modules["cv2"] = cv2
    spec = importlib.util.spec_from_file_location("metrics_reference", f"{REF}/mmsr/utils/metrics.py")
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    out = {}
    for k in range(3):
        a, b = metric_images(k)
        crop = 4
        out[f"psnr{k}"] = np.float64(m.psnr(a.copy(), b.copy(), crop_border=crop))
        ay = m.bgr2ycbcr(a.copy() / 255., only_y=True)
        by = m.bgr2ycbcr(b.copy() / 255., only_y=True)
        out[f"psnr_y{k}"] = np.float64(m.psnr(ay * 255, by * 255, crop_border=crop))
        out[f"ssim_y{k}"] = np.float64(m.ssim(ay * 255, by * 255, crop_border=crop))
        print("metrics", k, out[f"psnr{k}"], out[f"psnr_y{k}"], out[f"ssim_y{k}"])
    np.savez_compressed(os.path.join(HERE, "metrics_golden.npz"), **out)


SAMPLER_CASES = [(11, 3, 4), (7, 100, 2), (64, 1, 8), (5, 2, 3)]   # (len(dataset), ratio, world)


def make_sampler():
    """Index lists of the reference's DistIterSampler (mmsr/data/data_sampler.py:8-69) for every rank of a few (dataset size,
    ratio, world size) cases at epochs 0 and 5 -- the class is plain torch and imports as it is."""
    spec = importlib.util.spec_from_file_location("data_sampler_reference", f"{REF}/mmsr/data/data_sampler.py")
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    out = {}
    for (n, ratio, world) in SAMPLER_CASES:
        for epoch in (0, 5):
            for rank in range(world):
                smp = m.DistIterSampler(list(range(n)), num_replicas=world, rank=rank, ratio=ratio)
                smp.set_epoch(epoch)
                out[f"n{n}_r{ratio}_w{world}_e{epoch}_rank{rank}"] = np.array(list(iter(smp)), np.int64)
    np.savez_compressed(os.path.join(HERE, "sampler_golden.npz"), **out)
    print("sampler:", len(out), "index lists")


def main():
    torch.manual_seed(0)
    torch.set_num_threads(4)
    rmu = load_ref_map_util()
    out = {}
    for k, (name, C, hq, hr, patch, s_in, s_ref, builder) in enumerate(corr_cases()):
        fi, fr = build_inputs(name, C, hq, hr, builder, 100 + 10 * k)
        for norm_input in (False, True):
            idx, val = rmu.feature_match_index(torch.from_numpy(fi), torch.from_numpy(fr), patch_size=patch,
                                               input_stride=s_in, ref_stride=s_ref, is_norm=True, norm_input=norm_input)
            out[f"{name}/idx"] = idx.numpy().astype(np.int64)
            out[f"{name}/val_ni{int(norm_input)}"] = val.numpy().astype(np.float32)
        idx0, val0 = rmu.feature_match_index(torch.from_numpy(fi), torch.from_numpy(fr), patch_size=patch,
                                             input_stride=s_in, ref_stride=s_ref, is_norm=False, norm_input=False)
        out[f"{name}/idx_nonorm"] = idx0.numpy().astype(np.int64)
        out[f"{name}/val_nonorm"] = val0.numpy().astype(np.float32)
        out[f"{name}/gap"] = np.float64(top2_gap(rmu, fi, fr, patch, s_in, s_ref))
        out[f"{name}/meta"] = np.array([C, hq[0], hq[1], hr[0], hr[1], patch, s_in, s_ref, 100 + 10 * k], np.int64)
        print(name, "gap", out[f"{name}/gap"], "idx range", idx.min().item(), idx.max().item())
    # keep the smallest case's inputs as a guard on the generator itself
    fi, fr = build_inputs("c256_10v16", 256, (10, 10), (16, 16), None, 100)
    out["c256_10v16/feat_in"], out["c256_10v16/feat_ref"] = fi, fr
    np.savez_compressed(os.path.join(HERE, "corr_golden.npz"), **out)

    # ---- CorrespondenceGenerationArch.forward: normalise -> match -> index_to_flow -> 27 shifts (pre_offset dict)
    stub_third_party()
    sys.path.insert(0, REF)
    from mmsr.models.archs.corres_generation_arch import CorrespondenceGenerationArch  # noqa: E402
    net = CorrespondenceGenerationArch(patch_size=3, stride=1, vgg_layer_list=["relu1_1", "relu2_1", "relu3_1"], vgg_type="vgg19").eval()
    B, h, w = 2, 12, 14
    f1 = synth.gaussish((B, 256, h, w), 900)
    f2 = synth.gaussish((B, 256, h, w), 901)
    f2[1, :, :, 9:] = 0.0   # zero-padded region in sample 1: exact ties
    img = synth.uniform((B, 3, 4 * h, 4 * w), 902, 0.0, 1.0)
    with torch.no_grad():
        pre, _ = net({"dense_features1": torch.from_numpy(f1), "dense_features2": torch.from_numpy(f2)}, torch.from_numpy(img))
    po = {"meta": np.array([B, 256, h, w, 900, 901], np.int64)}
    for k, v in pre.items():
        po[k] = v.numpy().astype(np.float32)
        print(k, v.shape)
    np.savez_compressed(os.path.join(HERE, "pre_offset_golden.npz"), **po)

    # ---- RestorationNet.forward (ref_restoration_arch.py:51-65) at BASELINE config-1 size: LR 40x40 -> SR 160x160.
    # Inputs are synthetic at the net's interface (LR image, integer-valued pre-offsets from a hashed index map, ref
    # features) so that the fixture does not depend on how a VGG conv rounds on a particular CPU.
    from mmsr.models.archs.ref_restoration_arch import RestorationNet  # noqa: E402
    net_g = RestorationNet(ngf=64, n_blocks=16, groups=8).eval()
    fill_parameters(net_g, "net_g.")
    lr, pre, feats = restoration_inputs(1, 40, 40)
    import json
    from mmsr.models.archs.contras_extractor_arch import ContrasExtractorSep  # noqa: E402
    keys = {"net_g": {k: list(v.shape) for k, v in net_g.state_dict().items()},
            "net_map": {k: list(v.shape) for k, v in net.state_dict().items()},
            "net_extractor": {k: list(v.shape) for k, v in ContrasExtractorSep().state_dict().items()}}
    json.dump(keys, open(os.path.join(HERE, "state_dict_keys.json"), "w"), indent=0)
    taps = {}
    for stage in ("small", "medium", "large"):  # outputs of the three DynAgg (DCNv2) sites, spatially subsampled
        getattr(net_g.dyn_agg_restore, f"{stage}_dyn_agg").register_forward_hook(
            lambda m, i, o, stage=stage: taps.__setitem__(stage, o.detach().numpy()[..., ::5, ::5].copy()))
    with torch.no_grad():
        sr = net_g(torch.from_numpy(lr), {k: torch.from_numpy(v) for k, v in pre.items()},
                   {k: torch.from_numpy(v) for k, v in feats.items()})
    print("sr", sr.shape, float(sr.abs().mean()), float(sr.std()), {k: float(np.abs(v).mean()) for k, v in taps.items()})
    np.savez_compressed(os.path.join(HERE, "restoration_golden.npz"), sr=sr.numpy().astype(np.float32),
                        meta=np.array([1, 40, 40], np.int64), **{f"dyn_agg_{k}": v.astype(np.float32) for k, v in taps.items()})


if __name__ == "__main__":
    what = sys.argv[1] if len(sys.argv) > 1 else "small"
    if what in ("small", "all"):
        main()
    if what in ("full160", "all"):
        make_full160()
    if what in ("cfg5", "all"):
        make_cfg5()
    if what in ("metrics", "all"):
        make_metrics()
    if what in ("sampler", "all"):
        make_sampler()
