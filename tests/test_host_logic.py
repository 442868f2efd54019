"""CPU: host-side logic -- registries, checkpoint key layout (drop-in contract), the data-parallel
wrapper on a world_size-2 gloo group."""
import json
import os
import sys

import pytest
import torch
import torch.nn as nn

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_arch_registry_builds_the_yaml_blocks():
    import mmsr.models.networks as networks
    opt = {"network_g": {"type": "RestorationNet", "ngf": 64, "n_blocks": 2, "groups": 8},
           "network_map": {"type": "CorrespondenceGenerationArch", "patch_size": 3, "stride": 1,
                           "vgg_layer_list": ["relu1_1", "relu2_1", "relu3_1"], "vgg_type": "vgg19"},
           "network_extractor": {"type": "ContrasExtractorSep"}}
    g, m, e = networks.define_net_g(opt), networks.define_net_map(opt), networks.define_net_extractor(opt)
    assert type(g).__name__ == "RestorationNet" and type(m).__name__ == "CorrespondenceGenerationArch"
    assert type(e).__name__ == "ContrasExtractorSep"
    with pytest.raises(ValueError):
        networks.dynamical_instantiation(networks._arch_modules, "NoSuchArch", {})


def test_state_dict_layout_matches_the_reference(golden_dir):
    want = json.load(open(f"{golden_dir}/state_dict_keys.json"))
    from mmsr.models.archs.contras_extractor_arch import ContrasExtractorSep
    from mmsr.models.archs.corres_generation_arch import CorrespondenceGenerationArch
    from mmsr.models.archs.ref_restoration_arch import RestorationNet
    nets = {"net_g": RestorationNet(64, 16, 8),
            "net_map": CorrespondenceGenerationArch(3, 1, ["relu1_1", "relu2_1", "relu3_1"], "vgg19"),
            "net_extractor": ContrasExtractorSep()}
    for name, net in nets.items():
        got = {k: list(v.shape) for k, v in net.state_dict().items()}
        assert list(got.keys()) == list(want[name].keys()), name
        assert got == want[name], name
    assert sum(p.numel() for p in nets["net_g"].parameters()) == 8865547
    # re_init_dcn_offset state (ref_restoration_arch.py:42-49)
    for stage in ("small", "medium", "large"):
        head = getattr(nets["net_g"].dyn_agg_restore, f"{stage}_dyn_agg").conv_offset_mask
        assert float(head.weight.abs().max()) == 0.0 and float(head.bias.abs().max()) == 0.0
    assert not any(p.requires_grad for p in nets["net_map"].parameters())


def test_tensor_shift_and_sample_patches_semantics():
    from mmsr.models.archs.arch_util import tensor_shift
    from mmsr.models.archs.ref_map_util import sample_patches
    x = torch.arange(2 * 4 * 5 * 2, dtype=torch.float32).view(2, 4, 5, 2)
    y = tensor_shift(x, (1, 2))
    assert torch.equal(y[:, 1:, 2:], x[:, :3, :3]) and float(y[:, 0].abs().max()) == 0 and float(y[:, :, :2].abs().max()) == 0
    f = torch.randn(3, 6, 7)
    p = sample_patches(f, 3, 1)
    assert tuple(p.shape) == (3, 3, 3, 4 * 5)
    want = f.unfold(1, 3, 1).unfold(2, 3, 1).reshape(3, -1, 3, 3).permute(0, 2, 3, 1)
    assert torch.equal(p, want)
    p2 = sample_patches(f, 3, 2)
    assert torch.equal(p2, f.unfold(1, 3, 2).unfold(2, 3, 2).reshape(3, -1, 3, 3).permute(0, 2, 3, 1))


def _ddp_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    sys.path.insert(0, os.path.join(REPO, "c2-matching_amd"))
    import torch.distributed as dist
    from torch.nn.parallel import DistributedDataParallel
    from mmsr.models.base_model import BaseModel, unwrap
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        model = BaseModel({"dist": True, "gpu_ids": None, "is_train": True})
        torch.manual_seed(0)
        net = nn.Sequential(nn.Conv2d(3, 4, 3, padding=1), nn.ReLU(), nn.Conv2d(4, 3, 3, padding=1))
        frozen = nn.Conv2d(3, 3, 1)
        for p in frozen.parameters():
            p.requires_grad = False
        wrapped = model.model_to_device(net)
        assert isinstance(wrapped, DistributedDataParallel)
        assert not isinstance(model.model_to_device(frozen), DistributedDataParallel)       # frozen net stays bare
        assert not isinstance(model.model_to_device(nn.Conv2d(3, 3, 1), receives_gradients=False), DistributedDataParallel)
        g = torch.Generator().manual_seed(123)
        data = torch.randn(4, 3, 8, 8, generator=g)
        target = torch.randn(4, 3, 8, 8, generator=g)
        shard = slice(rank * 2, rank * 2 + 2)                                             # batch sharded by rank
        loss = nn.functional.l1_loss(wrapped(data[shard]), target[shard])
        loss.backward()
        grads = [p.grad.clone() for p in unwrap(wrapped).parameters()]
        q.put((rank, [g_.numpy() for g_ in grads]))
    finally:
        dist.destroy_process_group()


def test_ddp_wrapper_gloo_world2_matches_single_process():
    import numpy as np
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + os.getpid() % 2000
    procs = [ctx.Process(target=_ddp_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    # single-process gradient on the concatenated batch
    torch.manual_seed(0)
    net = nn.Sequential(nn.Conv2d(3, 4, 3, padding=1), nn.ReLU(), nn.Conv2d(4, 3, 3, padding=1))
    g = torch.Generator().manual_seed(123)
    data = torch.randn(4, 3, 8, 8, generator=g)
    target = torch.randn(4, 3, 8, 8, generator=g)
    nn.functional.l1_loss(net(data), target).backward()
    for r in range(2):
        for a, p in zip(got[r], net.parameters()):
            np.testing.assert_allclose(a, p.grad.numpy(), atol=1e-6)


def test_optimizer_groups_follow_parameter_names():
    from mmsr.models.ref_restoration_model import RefRestorationModel
    opt = {"dist": False, "gpu_ids": None, "is_train": True, "path": {},
           "network_g": {"type": "RestorationNet", "ngf": 64, "n_blocks": 1, "groups": 8},
           "network_map": {"type": "CorrespondenceGenerationArch", "patch_size": 3, "stride": 1,
                           "vgg_layer_list": ["relu1_1", "relu2_1", "relu3_1"], "vgg_type": "vgg19"},
           "network_extractor": {"type": "ContrasExtractorSep"},
           "train": {"lr_g": 1e-4, "lr_offset": 1e-4, "lr_relu2_offset": 1e-5, "lr_relu3_offset": 1e-6,
                     "weight_decay_g": 0, "beta_g": [0.9, 0.999], "pixel_weight": 1.0}}
    m = RefRestorationModel(opt)
    lrs = [g["lr"] for g in m.optimizer_g.param_groups]
    assert lrs == [1e-4, 1e-4, 1e-6, 1e-5]
    n = [len(g["params"]) for g in m.optimizer_g.param_groups]
    # 'offset' in name: {small,medium,large}_offset_conv{1,2} and *_dyn_agg.conv_offset_mask, weight+bias each
    assert n[1] == 6 and n[2] == 6 and n[3] == 6 and sum(n) == len(list(m.net_g.parameters()))
    assert "network_g" in opt and opt["network_g"]["type"] == "RestorationNet"  # caller's dict untouched


def test_bench_flop_model_matches_kernel_tiling():
    """bench.py prices the correlation kernel by the MFMA work it issues; the formula hard-codes the kernel's tiling
    (14 x 14 query patches per workgroup, 28 ref patch columns per x-tile, 8 waves): keep the two in sync."""
    import importlib.util
    import os
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = open(os.path.join(root, "c2-matching_amd", "csrc", "corr_argmax.hip")).read()
    consts = {k: int(v) for k, v in re.findall(r"constexpr int (TQ|WT|WP|NWAVE) = (\d+);", src)}
    assert consts == {"TQ": 16, "WT": 32, "WP": 28, "NWAVE": 8}
    spec = importlib.util.spec_from_file_location("bench", os.path.join(root, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    B, C, h = 16, 256, 160
    tiles = (-(-(h - 2) // (consts["TQ"] - 2))) ** 2
    steps = (-(-(h - 2) // consts["WP"])) * h
    want = B * tiles * (steps + 1) * consts["NWAVE"] * (C // 2) * (2 * 32 * 32 * 2)
    assert bench.corr_executed_flops(B, C, h) == want == 9286793035776


def test_bench_power_probe_parses_rocm_smi_and_never_raises(monkeypatch, tmp_path):
    """bench.py's `power_probe` (socket power / shader clock while the step runs back to back): the rocm-smi JSON is parsed as the tool
    prints it on the MI355X boxes, idle samples (< 600 W) are dropped when busy ones exist, and an unreadable tool yields {"error": ...}
    instead of an exception -- the probe must never cost a bench line."""
    import importlib.util
    import os
    import subprocess
    import time
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("bench", os.path.join(root, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    fake = tmp_path / "rocm-smi"
    fake.write_text("#!/bin/sh\n")
    fake.chmod(0o755)
    monkeypatch.setattr("shutil.which", lambda name: str(fake))
    seq = iter([(241.0, 95), (1398.0, 1641), (1402.0, 1650), (1400.0, 1633)] + [(1400.0, 1640)] * 1000)

    class R:
        def __init__(self, out):
            self.stdout = out

    def run_ok(cmd, **kw):
        if "-M" in cmd:
            return R('WARNING: something\n{"card0": {"Max Graphics Package Power (W)": "1400.0"}}')
        w, c = next(seq)
        return R('{"card0": {"sclk clock speed:": "(%dMhz)", "sclk clock level:": "1", "Current Socket Graphics Package Power (W)": "%.1f"}}' % (c, w))
    monkeypatch.setattr(subprocess, "run", run_ok)
    it = [0]
    got = bench.power_probe(lambda: time.sleep(0.01), lambda: None, it, seconds=0.3)
    assert got["cap_w"] == 1400.0 and 1390 <= got["socket_w"] <= 1402 and 1600 <= got["sclk_mhz"] <= 1700 and got["samples"] >= 3, got

    def run_bad(cmd, **kw):
        raise OSError("no such tool")
    monkeypatch.setattr(subprocess, "run", run_bad)
    got = bench.power_probe(lambda: time.sleep(0.01), lambda: None, it, seconds=0.1)
    assert "error" in got, got


def test_bench_dcn_roofline_prices_both_gemm_arithmetics():
    """bench.py's DCNv2 rows: the fp32 GEMM executes the algorithmic flops on the fp32 matrix pipe; the f16 x 2 GEMM executes
    three products per k step on the f16 pipe -- `frac` is always executed / that pipe's dense peak, and `frac_vs_fp32_pipe`
    keeps the figure north_star's DCNv2 bar was written for (algorithmic / 157.3 TF)."""
    import importlib.util
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("bench", os.path.join(root, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    B, C, H, ms = 16, 64, 640, 4.9
    flops = B * 2.0 * C * 9 * C * H * H
    a = bench.dcn_roofline("large", B, C, C, H, ms, 20)
    b = bench.dcn_roofline("large", B, C, C, H, ms, 20, f16x2=True)
    assert a["algorithmic_flops_per_launch"] == b["algorithmic_flops_per_launch"] == flops
    assert a["executed_flops_per_launch"] == flops and b["executed_flops_per_launch"] == 3 * flops
    assert a["peak"] == bench.FP32_MATRIX_PEAK_TFLOPS and b["peak"] == bench.BF16_MATRIX_PEAK_TFLOPS
    tf = flops / (ms * 1e-3) / 1e12
    assert abs(a["frac"] - tf / 157.3) < 1e-3 and abs(b["frac"] - 3 * tf / 2500.0) < 1e-3
    assert a["frac_vs_fp32_pipe"] == b["frac_vs_fp32_pipe"] and abs(b["frac_vs_fp32_pipe"] - tf / 157.3) < 1e-3
    assert 0.0 < b["frac"] < 1.0


def test_dcn_modules_are_copyable_and_picklable():
    """the deferred offset-mean watch must not make copy.deepcopy(net) / torch.save(net) fail."""
    import copy
    import io
    import torch
    from mmsr.models.archs.DCNv2.dcn_v2 import DCN_sep_pre_multi_offset, _OffsetMeanWatch
    m = DCN_sep_pre_multi_offset(16, 16, 3, stride=1, padding=1, deformable_groups=2)
    m._watch._pending = (object(), object(), 1)   # as after a forward: un-picklable scratch
    c = copy.deepcopy(m)
    assert isinstance(c._watch, _OffsetMeanWatch) and c._watch._pending is None and c._watch is not m._watch
    buf = io.BytesIO()
    torch.save(m, buf)
    buf.seek(0)
    r = torch.load(buf, weights_only=False)
    assert r._watch._pending is None and torch.equal(r.weight, m.weight)


def test_dcn_geometry_is_validated_at_construction():
    import pytest
    from mmsr.models.archs.DCNv2.dcn_v2 import DCNv2
    with pytest.raises(NotImplementedError):
        DCNv2(9, 8, 3, 1, 1, deformable_groups=3)     # 3 channels per group: no kernel
    with pytest.raises(ValueError):
        DCNv2(10, 8, 3, 1, 1, deformable_groups=3)
    DCNv2(12, 8, 3, 1, 1, deformable_groups=2)        # 6 per group: forward only (warning), constructs


def test_vgg_without_torchvision_says_weights_are_random(monkeypatch):
    """never fall back to random VGG weights silently."""
    import importlib.util
    import pytest
    if importlib.util.find_spec('torchvision') is not None:
        pytest.skip('torchvision present: the ImageNet weights are loaded as in the reference')
    monkeypatch.delenv('C2M_VGG_WEIGHTS', raising=False)
    from mmsr.models.archs.vgg_arch import VGGFeatureExtractor
    with pytest.warns(RuntimeWarning, match='RANDOM weights'):
        VGGFeatureExtractor(['relu1_1'], 'vgg19')


def _val_worker(rank, world, port, q):
    import os
    import sys
    import torch
    import torch.distributed as dist
    here = os.path.dirname(os.path.abspath(__file__))
    for p_ in (os.path.join(os.path.dirname(here), "c2-matching_amd"), os.path.join(here, "golden")):
        sys.path.insert(0, p_)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from mmsr.models.ref_restoration_model import RefRestorationModel

    class Fake(RefRestorationModel):   # validation logic only: no nets, SR = GT + a per-item perturbation
        def __init__(self):
            self.opt = {'dist': world > 1, 'scale': 4}
            self.device = torch.device('cpu')
            self.rank = rank
            self.is_train = False

        def feed_data(self, data):
            self.gt = data['img_in']
            self.img_in_lq = data['img_in_lq']
            self.k = data['k']

        def test(self):
            self.output = (self.gt + 0.01 * (self.k + 1) * torch.sin(40 * self.gt)).clamp(0, 1)
            return self.output

    g = torch.Generator().manual_seed(0)
    items = [{'img_in': torch.rand(1, 3, 40, 44, generator=g), 'img_in_lq': torch.zeros(1), 'k': k} for k in range(5)]
    res = Fake().validation(items, 0, None, False)
    q.put((rank, res))
    dist.destroy_process_group()


def test_distributed_validation_equals_single_process():
    """dist_validation (broken in the reference, sr_model.py:160-162) shards the loader over the ranks
    and all-reduces the metric sums: world-2 gloo result == single-process result on every rank."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    out = {}
    for world, port in ((1, 29731), (2, 29732)):
        q = ctx.Queue()
        procs = [ctx.Process(target=_val_worker, args=(r, world, port, q)) for r in range(world)]
        [p.start() for p in procs]
        got = [q.get(timeout=120) for _ in range(world)]
        [p.join(60) for p in procs]
        out[world] = dict(got)
    single = out[1][0]
    assert single['count'] == 5 and 20 < single['psnr'] < 60
    for r in (0, 1):
        for k in ('psnr', 'psnr_y', 'ssim_y'):
            assert abs(out[2][r][k] - single[k]) < 1e-9, (r, k)
        assert out[2][r]['count'] == 5


def test_weight_cache_entries_belong_to_one_tensor_object():
    """ops._WeightCache: an entry is only ever returned for the tensor object it was made for (a new parameter that reuses
    a dead one's id / address / version must not inherit its re-laid-out weights) and it dies with that tensor."""
    import gc
    import torch
    from c2m_amd import ops
    c = ops._WeightCache()
    w = torch.nn.Parameter(torch.zeros(2, 2, 3, 3))
    c._store(("slot",), ("key",), w, "relayout-of-w")
    assert c._lookup(("slot",), ("key",), w) == "relayout-of-w"
    assert c._lookup(("slot",), ("other key",), w) is None            # version / address changed
    twin = torch.nn.Parameter(torch.zeros(2, 2, 3, 3))
    assert c._lookup(("slot",), ("key",), twin) is None               # same key, different object
    del w
    gc.collect()
    assert len(c._d) == 0


def test_pool_fusion_predicate_and_grouped_twin_validation():
    """Host-side decisions of ops.vgg_stack_forward / conv3x3 that do not need a GPU: which MaxPool2d a convolution epilogue
    may absorb, and the shape contract of the group-major twin buffer."""
    import pytest
    import torch
    from c2m_amd import ops, _lib
    assert ops._pool_is_2x2(torch.nn.MaxPool2d(2, 2))
    assert ops._pool_is_2x2(torch.nn.MaxPool2d(kernel_size=(2, 2), stride=(2, 2), padding=0))
    assert not ops._pool_is_2x2(torch.nn.MaxPool2d(3, 2, 1))
    assert not ops._pool_is_2x2(torch.nn.MaxPool2d(2, 1))
    assert not ops._pool_is_2x2(torch.nn.MaxPool2d(2, 2, ceil_mode=True))
    dev = torch.device("cpu")
    assert ops._grouped8_args(None, 1, 64, 4, 4, dev) == (None, 0, 0, 0)
    good = torch.zeros(2, 8, 7, 9, 8)
    ptr, row, plane, img = ops._grouped8_args(good, 2, 64, 4, 6, dev)
    assert (row, plane, img) == (9 * 8, 7 * 9 * 8, 8 * 7 * 9 * 8) and ptr == good.data_ptr() + (9 + 1) * 8 * 4
    for bad in (torch.zeros(2, 8, 7, 9, 4), torch.zeros(2, 8, 7, 9, 8, dtype=torch.float64), torch.zeros(2, 7, 8, 9, 8)):
        with pytest.raises(_lib.C2MError):
            ops._grouped8_args(bad, 2, 64, 4, 6, dev)


def test_conv_algo_selection_is_shape_driven():
    """ops._wino_ok / _wino4_ok: F(2,3) needs 64-channel output tiles, 16-channel sources and whole 32-pixel tiles; F(4,3)
    additionally whole 64-pixel tiles and a plain channels-last output (it is what conv3x3(fast=True) may pick)."""
    import torch
    from c2m_amd import ops
    w64 = torch.zeros(64, 64, 3, 3)
    src = lambda c, w: [torch.zeros(1, c, 8, w)]   # noqa: E731
    assert ops._wino_ok(src(64, 160), w64, "nhwc", 160) and not ops._wino4_ok(src(64, 160), w64, "nhwc", 160)
    assert ops._wino4_ok(src(64, 640), w64, "nhwc", 640)
    assert not ops._wino4_ok(src(64, 640), w64, "nhwc_pool2", 640) and ops._wino_ok(src(64, 640), w64, "nhwc_pool2", 640)
    assert not ops._wino_ok(src(64, 40), w64, "nhwc", 40)                         # ragged width
    assert not ops._wino_ok(src(64, 64), torch.zeros(32, 64, 3, 3), "nhwc", 64)   # 32 output channels
    assert not ops._wino_ok(src(64, 64), w64, "nchw", 64)
    assert not ops._wino_ok([torch.zeros(1, 24, 8, 64), torch.zeros(1, 40, 8, 64)], w64, "nhwc", 64)   # 24-channel source
    # split-bf16 kernel: any map size, 16-channel chunks; policy from $C2M_CONV_SPLIT ("all": every call, "1": fast=True only)
    old = ops._SPLIT
    try:
        ops._SPLIT = "all"
        assert ops._split_ok(src(64, 37), w64, False) and ops._split_ok(src(48, 5), torch.zeros(24, 48, 3, 3), False)
        assert not ops._split_ok(src(24, 64), torch.zeros(64, 24, 3, 3), True)       # 24 channels: not a whole chunk
        assert not ops._split_ok(src(32, 64), w64, True)                             # sources do not add up to the weight's Cin
        ops._SPLIT = "1"
        assert ops._split_ok(src(64, 37), w64, True) and not ops._split_ok(src(64, 37), w64, False)
        ops._SPLIT = "0"
        assert not ops._split_ok(src(64, 37), w64, True)
    finally:
        ops._SPLIT = old


def test_cpu_chain_mismatch_margins_scores_both_picks():
    """oracle/cpu_chain.mismatch_margins (bench.py's near-tie diagnostic): for every query where two index maps disagree it
    reports score(pick_a) - score(pick_b) in float64 with ref_map_util.py:52-76's normalisation -- zero for an exact
    duplicate patch, and equal to the oracle's score difference otherwise."""
    import os
    import sys
    import numpy as np
    import torch
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
    import cpu_chain
    g = torch.Generator().manual_seed(3)
    f1, f2 = torch.randn(8, 9, 9, generator=g), torch.randn(8, 9, 9, generator=g)
    f2[:, 4:7, 4:7] = f2[:, 0:3, 0:3]            # ref patch (4, 4) duplicates ref patch (0, 0)
    wq = 7
    a = np.zeros((7, 7), dtype=np.int64)
    b = a.copy()
    b[2, 3] = 4 * wq + 4                          # the duplicate: an exact tie
    b[5, 1] = 3 * wq + 2                          # some other patch
    out = cpu_chain.mismatch_margins(f1, f2, a, b)
    assert [o[0] for o in out] == [(2, 3), (5, 1)] and out[0][1:3] == (0, 4 * wq + 4)
    assert abs(out[0][3]) < 1e-12
    n1 = torch.nn.functional.normalize(f1.reshape(8, -1), dim=0).view(8, 9, 9).double().numpy()
    n2 = torch.nn.functional.normalize(f2.reshape(8, -1), dim=0).view(8, 9, 9).double().numpy()
    sc = lambda ry, rx: float((n1[:, 5:8, 1:4] * n2[:, ry:ry + 3, rx:rx + 3]).sum() /   # noqa: E731
                              (np.sqrt((n2[:, ry:ry + 3, rx:rx + 3] ** 2).sum()) + 1e-5))
    assert abs(out[1][3] - (sc(0, 0) - sc(3, 2))) < 1e-12


def test_pre_offsets_survive_the_ddp_and_dataparallel_input_scatter():
    """RefRestorationModel hands `pre_offset` to a wrapped net_g as a positional input; DistributedDataParallel and
    DataParallel rebuild dict inputs as type(obj)(pairs) after moving / slicing every value.  The
    rebuilt object must still be a PreOffsets carrying the index map, and nothing may be materialised on the way (no
    kernel can run here: there is no GPU, so a materialisation attempt would raise)."""
    import copy
    import pickle
    from torch.distributed.utils import _recursive_to
    from torch.nn.parallel.scatter_gather import scatter_kwargs  # noqa: F401  (import check only)
    from mmsr.models.archs.corres_generation_arch import PreOffsets
    idx = torch.arange(2 * 6 * 7, dtype=torch.int64).view(2, 6, 7)
    pre = PreOffsets(idx, 8, 9)
    x = torch.zeros(2, 3, 8, 9)
    (moved,) = _recursive_to((x, pre, {"relu1_1": x}), torch.device("cpu"), False)
    assert type(moved[1]) is PreOffsets and moved[1].max_idx is idx and (moved[1].h, moved[1].w) == (8, 9)
    assert list(moved[1].keys()) == ["max_idx"]                      # still lazy
    assert "relu3_1" in pre and "relu1_1" in pre and "nope" not in pre
    # DataParallel's scatter_map on a dict: type(obj)(pairs) per replica with the values sliced along dim 0
    halves = [PreOffsets([("max_idx", part)]) for part in idx.chunk(2, 0)]
    assert [tuple(h_.max_idx.shape) for h_ in halves] == [(1, 6, 7)] * 2 and (halves[0].h, halves[0].w) == (8, 9)
    for clone in (copy.deepcopy(pre), pickle.loads(pickle.dumps(pre))):
        assert type(clone) is PreOffsets and torch.equal(clone.max_idx, idx) and (clone.h, clone.w) == (8, 9)
    with pytest.raises(TypeError):
        PreOffsets({"relu3_1": x})
    with pytest.raises(KeyError):
        pre["relu9_9"]


def test_weight_cache_refresh_follows_writes_through_data():
    """`p.data.copy_()` leaves `p._version` alone, so the (data_ptr, _version) key of the weight caches
    cannot see it.  refresh() re-runs the stored re-layout of the live entries -- of the given parameters only -- from the
    tensors' current contents; entries whose key is stale anyway are left to get()."""
    import torch
    from c2m_amd import ops
    c = ops._WeightCache()
    w1, w2 = torch.nn.Parameter(torch.zeros(2, 2, 3, 3)), torch.nn.Parameter(torch.ones(2, 2, 3, 3))
    calls = []

    def redo(w, buf):
        buf.copy_(w.detach().flatten()[:4] * 2)
        calls.append(id(w))
    for slot, w in (("a", w1), ("b", w2)):
        c._store((slot,), (w.data_ptr(), w._version), w, torch.zeros(4), redo=redo)
    v0 = w1._version
    w1.data.fill_(3.0)
    assert w1._version == v0                                         # the blind spot
    assert c.refresh({id(w1)}) == 1 and calls == [id(w1)]
    assert torch.equal(c._lookup(("a",), (w1.data_ptr(), w1._version), w1), torch.full((4,), 6.0))
    assert c.refresh() == 2
    with torch.no_grad():
        w2.add_(1.0)                                                 # a real in-place update: key is stale, get() handles it
    assert c.refresh({id(w2)}) == 0
    c.clear()
    assert c.refresh() == 0


def test_training_state_round_trips_like_the_reference(tmp_path):
    """BaseModel.save_training_state / resume_training (mmsr/models/base_model.py:267-307): `<iter>.state` with the keys epoch /
    iter / optimizers / schedulers, master rank only, nothing for iter -1; resuming restores the Adam moments and the scheduler
    position and refuses a state with the wrong number of optimizers."""
    import torch
    from mmsr.models.base_model import BaseModel
    opt = {'path': {'training_state': str(tmp_path)}, 'is_train': True}

    def make():
        m = BaseModel(opt)
        net = torch.nn.Linear(4, 3)
        torch.manual_seed(0)
        o = torch.optim.Adam(net.parameters(), lr=1e-2)
        m.optimizers.append(o)
        m.schedulers.append(torch.optim.lr_scheduler.MultiStepLR(o, [2, 5], 0.5))
        return m, net
    m, net = make()
    for _ in range(3):
        net(torch.ones(2, 4)).sum().backward()
        m.optimizers[0].step()
        m.schedulers[0].step()
    m.save_training_state(epoch=7, current_iter=-1)
    assert not list(tmp_path.iterdir())
    m.save_training_state(epoch=7, current_iter=300)
    state = torch.load(str(tmp_path / '300.state'))
    assert set(state) == {'epoch', 'iter', 'optimizers', 'schedulers'} and state['epoch'] == 7 and state['iter'] == 300
    m2, net2 = make()
    m2.resume_training(state)
    assert m2.schedulers[0].last_epoch == 3 and m2.optimizers[0].param_groups[0]['lr'] == m.optimizers[0].param_groups[0]['lr']
    s1, s2 = m.optimizers[0].state_dict()['state'], m2.optimizers[0].state_dict()['state']
    assert all(torch.equal(s1[k]['exp_avg'], s2[k]['exp_avg']) and torch.equal(s1[k]['exp_avg_sq'], s2[k]['exp_avg_sq']) for k in s1)
    m2.optimizers.append(m2.optimizers[0])
    with pytest.raises(AssertionError, match='Wrong lengths of optimizers'):
        m2.resume_training(state)
    m.rank = 1                                     # not the master: nothing is written
    m.save_training_state(epoch=8, current_iter=400)
    assert not (tmp_path / '400.state').exists()
