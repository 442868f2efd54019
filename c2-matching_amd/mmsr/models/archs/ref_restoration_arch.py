"""``RestorationNet`` (ref_restoration_arch.py:30-65) with the reference's parameter names (checkpoint keys listed in
SURVEY.md A.3).  The three ``*_dyn_agg`` sites are DCN_sep_pre_multi_offset layers (:77-85, :101-109, :124-132) and run
on the gfx950 DCNv2 kernels.

Two execution paths over the SAME modules / parameters:

* autograd (training, or any call with gradients enabled): the reference's composition with every 3x3 convolution that has
  at least 32 input channels on the hand-written kernels FORWARD AND BACKWARD (``ops.conv3x3_autograd``: split-bf16 forward,
  the same kernel on rotated / transposed weights for the data gradient, an fp32-MFMA weight-gradient kernel), DynAgg through
  ``_DCNv2`` with the hand-written forward/backward kernels; ``allow_fused = False`` (or hooks on inner modules) falls back to
  the stock modules (MIOpen);
* fused inference (``torch.no_grad()``, fp32 on the GPU, ``pre_offset`` produced by this package's
  ``CorrespondenceGenerationArch``): the whole net runs channels-last on the hand-written gfx950 kernels --
  csrc/conv3x3.hip computes act(conv(cat(a, b)) + bias) + residuals in one launch per convolution (no cat / bias /
  ReLU / add / PixelShuffle kernels, no layout changes between layers), the DCN offset/mask heads synthesise the
  pre-offsets from the arg-max flow map, and the DCNv2 forward reads the zero-bordered channels-last copy of the Ref
  feature that the offset convolutions already use and writes channels-last with its LeakyReLU folded in.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

import mmsr.models.archs.arch_util as arch_util
from c2m_amd import ops as _ops
from mmsr.models.archs.DCNv2.dcn_v2 import DCN_sep_pre_multi_offset as DynAgg
from mmsr.models.archs.DCNv2.dcn_v2 import FusedPreOffset, use_conv_kernels


class ContentExtractor(nn.Module):

    def __init__(self, in_nc=3, out_nc=3, nf=64, n_blocks=16):
        super(ContentExtractor, self).__init__()
        self.conv_first = nn.Conv2d(in_nc, nf, 3, 1, 1)
        self.body = arch_util.make_layer(arch_util.ResidualBlockNoBN, n_blocks, nf=nf)
        self.lrelu = nn.LeakyReLU(negative_slope=0.1, inplace=True)
        arch_util.default_init_weights([self.conv_first], 0.1)

    def forward(self, x):
        return self.body(self.lrelu(self.conv_first(x)))

    def forward_train(self, x):
        """With gradients: conv_first (3 input channels) on the stock module, the 16 residual blocks on the hand-written
        kernels (channels-last)."""
        f = self.lrelu(self.conv_first(x)).contiguous(memory_format=torch.channels_last)
        return _train_body(self.body, f)

    def forward_fused(self, x):
        """x [B,3,h,w] (any layout) -> content feature, channels-last.  A 3 -> 64 conv_first runs on the first-layer
        kernel; any other width sees the image zero-padded to 32 channels (the generic kernel's chunk size; the weights
        are padded to match)."""
        B, C, H, W = x.shape
        if tuple(self.conv_first.weight.shape) == (64, 3, 3, 3):
            f = _ops.conv3x3_rgb64(x, self.conv_first.weight, self.conv_first.bias, act=_ops.ACT_LRELU, slope=0.1)
            return _fused_body(self.body, f)
        x32 = torch.zeros((B, 32, H, W), dtype=torch.float32, device=x.device).contiguous(memory_format=torch.channels_last)
        x32[:, :C] = x
        f = _ops.conv3x3(x32, self.conv_first.weight, self.conv_first.bias, act=_ops.ACT_LRELU, slope=0.1)
        return _fused_body(self.body, f)


import os as _os

#: $C2M_TRAIN_KERNELS: "1" always run the training step's convolutions on the hand-written kernels, "0" never (stock
#: modules = MIOpen), "auto" (default) from TRAIN_KERNELS_MIN_PIXELS LR pixels per batch on
_TRAIN_KERNELS = _os.environ.get("C2M_TRAIN_KERNELS", "auto")
TRAIN_KERNELS_MIN_PIXELS = 4 * 96 * 96


def _train_kernels_wanted(x):
    if _TRAIN_KERNELS in ("0", "1"):
        return _TRAIN_KERNELS == "1"
    return x.shape[0] * x.shape[2] * x.shape[3] >= TRAIN_KERNELS_MIN_PIXELS


def _has_hooks(module):
    return any(m._forward_hooks or m._forward_pre_hooks for m in module.modules())


def _fusable_body(body):
    return all(isinstance(b, arch_util.ResidualBlockNoBN) and b.res_scale == 1 for b in body)


def _conv_t(m, srcs, act=_ops.ACT_NONE, slope=0.1):
    """act(m(cat(srcs))) WITH gradients: the hand-written forward / data-gradient / weight-gradient kernels
    (ops.conv3x3_autograd) where the shapes allow, the stock module otherwise (3-channel first / last convolutions)."""
    if isinstance(m, nn.Conv2d) and m.kernel_size == (3, 3) and m.stride == (1, 1) and m.padding == (1, 1) and \
            _ops.conv3x3_autograd_ok(srcs, m.weight):
        return _ops.conv3x3_autograd(srcs, m.weight, m.bias, act, slope)
    y = m(torch.cat(list(srcs), 1) if len(srcs) > 1 else srcs[0])
    return F.relu(y) if act == _ops.ACT_RELU else F.leaky_relu(y, slope) if act == _ops.ACT_LRELU else y


def _train_body(body, f, skip=None):
    """16 x (x + conv2(relu(conv1(x)))) under autograd on the hand-written kernels (arch_util.py:128-136)."""
    for blk in body:
        t = _conv_t(blk.conv1, [f], _ops.ACT_RELU)
        f = _conv_t(blk.conv2, [t]) + f
    return f if skip is None else f + skip


def _fused_body(body, f, skip=None):
    """16 x (x + conv2(relu(conv1(x)))) (arch_util.py:128-136), one launch per block on large maps (ops.resblock3x3), two
    otherwise; `skip` (the stage input,
    ref_restoration_arch.py:153,166,179 `h = body(h) + x`) rides on the last block's epilogue.  fast=True: decoder
    convolutions may take the Winograd F(4,3) kernel (ops.conv3x3) where the map is a whole number of 64-pixel tiles wide."""
    n = len(body)
    if _BF16_IO and _ops._SPLIT != "0" and _ops.bf16_autocast() and blk_bf16_ok(body):   # (bf16 tensors need the split kernel's bf16 flavour)
        # bf16 autocast (BASELINE configs[4]): the residual stream and the block-internal tensor travel as bf16 -- half the
        # HBM bytes of these launches, the tile goes HBM -> LDS without passing registers (c2m_conv3x3_desc.io_flags).  Sums
        # (bias, residuals) are taken in fp32 inside the kernel; the body's result leaves as fp32.  (torch's own autocast
        # keeps every conv output in bf16 as well: arch_util.py:128-136 under autocast.)
        bf = torch.bfloat16
        for k, blk in enumerate(body):
            t = _ops.conv3x3(f, blk.conv1.weight, blk.conv1.bias, act=_ops.ACT_RELU, fast=True, out_dtype=bf)
            f = _ops.conv3x3(t, blk.conv2.weight, blk.conv2.bias, res1=f, res2=skip if k == n - 1 else None, fast=True,
                             out_dtype=None if k == n - 1 else bf)
        return f
    if _ops.resblock3x3_wanted(f) and all(_ops.resblock3x3_ok(f, blk.conv1.weight, blk.conv2.weight) for blk in body):
        # one launch per block (csrc/conv3x3_resblock.hip): the intermediate tensor stays in LDS, the identity comes from the f16
        # pieces of x the kernel holds anyway -- 2 tensor passes per block instead of 5
        for k, blk in enumerate(body):
            f = _ops.resblock3x3(f, blk.conv1.weight, blk.conv1.bias, blk.conv2.weight, blk.conv2.bias,
                                 res2=skip if k == n - 1 else None)
        return f
    for k, blk in enumerate(body):
        t = _ops.conv3x3(f, blk.conv1.weight, blk.conv1.bias, act=_ops.ACT_RELU, fast=True)
        f = _ops.conv3x3(t, blk.conv2.weight, blk.conv2.bias, res1=f, res2=skip if k == n - 1 else None, fast=True)
    return f


# $C2M_BF16_IO=0: under bf16 autocast keep fp32 tensors between the fused convolutions (the round-3 behaviour)
_BF16_IO = _os.environ.get("C2M_BF16_IO", "1") != "0"


def blk_bf16_ok(body):
    return all(b.conv1.weight.shape[0] % 16 == 0 and b.conv1.weight.shape[1] % 16 == 0 and b.conv2.weight.shape[0] % 16 == 0 for b in body)


class DynamicAggregationRestoration(nn.Module):
    """Three coarse-to-fine stages (relu3_1 / relu2_1 / relu1_1 reference features): offset features from
    cat(content, ref) -> DynAgg warp of the ref feature -> fuse -> 16 residual blocks -> upsample."""

    _STAGES = (('small', 'relu3_1', 256), ('medium', 'relu2_1', 128), ('large', 'relu1_1', 64))

    def __init__(self, ngf=64, n_blocks=16, groups=8):
        super(DynamicAggregationRestoration, self).__init__()
        def up2():  # conv -> pixel shuffle x2 -> lrelu
            return nn.Sequential(nn.Conv2d(ngf, ngf * 4, kernel_size=3, stride=1, padding=1), nn.PixelShuffle(2),
                                 nn.LeakyReLU(0.1, True))

        tails = {'small': up2, 'medium': up2,
                 'large': lambda: nn.Sequential(nn.Conv2d(ngf, ngf // 2, kernel_size=3, stride=1, padding=1),
                                                nn.LeakyReLU(0.1, True),
                                                nn.Conv2d(ngf // 2, 3, kernel_size=3, stride=1, padding=1))}
        for name, _, ch in self._STAGES:  # registration order = the reference's checkpoint key order
            setattr(self, f'{name}_offset_conv1', nn.Conv2d(ngf + ch, ch, 3, 1, 1, bias=True))  # concat for diff
            setattr(self, f'{name}_offset_conv2', nn.Conv2d(ch, ch, 3, 1, 1, bias=True))
            setattr(self, f'{name}_dyn_agg', DynAgg(ch, ch, 3, stride=1, padding=1, dilation=1,
                                                   deformable_groups=groups, extra_offset_mask=True))
            setattr(self, f'head_{name}', nn.Sequential(nn.Conv2d(ngf + ch, ngf, kernel_size=3, stride=1, padding=1),
                                                        nn.LeakyReLU(0.1, True)))
            setattr(self, f'body_{name}', arch_util.make_layer(arch_util.ResidualBlockNoBN, n_blocks, nf=ngf))
            setattr(self, f'tail_{name}', tails[name]())
        self.lrelu = nn.LeakyReLU(negative_slope=0.1, inplace=True)

    def _stage(self, name, x, ref_feat, pre_offset):
        offset_feat = self.lrelu(getattr(self, f'{name}_offset_conv1')(torch.cat([x, ref_feat], 1)))
        offset_feat = self.lrelu(getattr(self, f'{name}_offset_conv2')(offset_feat))
        swapped = self.lrelu(getattr(self, f'{name}_dyn_agg')([ref_feat, offset_feat], pre_offset))
        h = getattr(self, f'head_{name}')(torch.cat([x, swapped], 1))
        h = getattr(self, f'body_{name}')(h) + x
        return getattr(self, f'tail_{name}')(h)

    def forward(self, x, pre_offset, img_ref_feat):
        for name, key, _ in self._STAGES:
            x = self._stage(name, x, img_ref_feat[key], pre_offset[key])
        return x

    def _stage_train(self, name, x, ref_feat, pre_offset):
        """_stage() with every 3x3 convolution on the differentiable hand-written kernels (channels-last activations); the
        DynAgg module keeps the reference's interface (NCHW tensors through _DCNv2), its offset/mask head included."""
        lre = _ops.ACT_LRELU
        ref_cl = _ops._as_nhwc(ref_feat)
        of = _conv_t(getattr(self, f'{name}_offset_conv1'), [x, ref_cl], lre)
        of = _conv_t(getattr(self, f'{name}_offset_conv2'), [of], lre)
        swapped = self.lrelu(getattr(self, f'{name}_dyn_agg')([ref_feat, of], pre_offset))
        h = _conv_t(getattr(self, f'head_{name}')[0], [x, _ops._as_nhwc(swapped)], lre)
        h = _train_body(getattr(self, f'body_{name}'), h, skip=x)
        tail = getattr(self, f'tail_{name}')
        if name == 'large':
            return tail[2](_conv_t(tail[0], [h], lre))          # 32 -> 3: stock module
        return tail[2](tail[1](_conv_t(tail[0], [h])))          # conv -> PixelShuffle(2) -> lrelu

    def forward_train(self, x, pre_offset, img_ref_feat):
        for name, key, _ in self._STAGES:
            x = self._stage_train(name, x, img_ref_feat[key], pre_offset[key])
        return x

    def _stage_fused(self, name, x, ref_feat, flow, scale):
        lrelu = dict(act=_ops.ACT_LRELU, slope=0.1, fast=True)
        ref = _ops.BorderedNHWC(ref_feat)          # one copy serves the offset conv (as a source) and the DCN gathers
        c1, c2 = getattr(self, f'{name}_offset_conv1'), getattr(self, f'{name}_offset_conv2')
        of = _ops.conv3x3([x, ref.interior()], c1.weight, c1.bias, **lrelu)
        of = _ops.conv3x3(of, c2.weight, c2.bias, **lrelu)
        # (a module call, so forward hooks see the DynAgg boundary on this path too)
        swapped = getattr(self, f'{name}_dyn_agg')([ref, of], FusedPreOffset(flow, scale, lrelu_slope=0.1))
        hd = getattr(self, f'head_{name}')[0]
        h = _ops.conv3x3([x, swapped], hd.weight, hd.bias, **lrelu)
        h = _fused_body(getattr(self, f'body_{name}'), h, skip=x)
        tail = getattr(self, f'tail_{name}')
        if name == 'large':
            t = _ops.conv3x3(h, tail[0].weight, tail[0].bias, **lrelu)
            return _ops.conv3x3(t, tail[2].weight, tail[2].bias, out_mode='nchw')
        # conv -> PixelShuffle(2) -> lrelu == conv -> lrelu -> PixelShuffle(2) (elementwise), done in the epilogue
        return _ops.conv3x3(h, tail[0].weight, tail[0].bias, out_mode='pixel_shuffle', **lrelu)

    def forward_fused(self, x, flow, img_ref_feat):
        for (name, key, _), scale in zip(self._STAGES, (1, 2, 4)):
            x = self._stage_fused(name, x, img_ref_feat[key], flow, scale)
        return x

    def fusable(self):
        return all(_fusable_body(getattr(self, f'body_{n}')) and getattr(self, f'{n}_dyn_agg').deformable_groups == 8
                   for n, _, _ in self._STAGES)

    def has_inner_hooks(self):
        """Forward (pre-)hooks registered on anything but the three DynAgg modules (whose boundary the fused path keeps):
        the fused path calls ops.conv3x3 on the sub-modules' parameters directly, so such hooks would silently not fire."""
        dyn = {id(getattr(self, f'{n}_dyn_agg')) for n, _, _ in self._STAGES}
        return any(id(m) not in dyn and m is not self and (m._forward_hooks or m._forward_pre_hooks) for m in self.modules())


class RestorationNet(nn.Module):

    def __init__(self, ngf=64, n_blocks=16, groups=8):
        super(RestorationNet, self).__init__()
        self.content_extractor = ContentExtractor(in_nc=3, out_nc=3, nf=ngf, n_blocks=n_blocks)
        self.dyn_agg_restore = DynamicAggregationRestoration(ngf, n_blocks, groups)
        arch_util.srntt_init_weights(self, init_type='normal', init_gain=0.02)
        self.re_init_dcn_offset()

    def re_init_dcn_offset(self):
        for name, _, _ in DynamicAggregationRestoration._STAGES:
            head = getattr(self.dyn_agg_restore, f'{name}_dyn_agg').conv_offset_mask
            head.weight.data.zero_()
            head.bias.data.zero_()

    #: set to False on an instance (or the class) to keep every forward on the module-by-module path (debugging, feature
    #: capture); forward hooks registered on inner modules switch the fused path off by themselves
    allow_fused = True

    def _use_fused(self, x, pre_offset, img_ref_feat):
        from mmsr.models.archs.corres_generation_arch import PreOffsets
        if not self.allow_fused:
            return False
        if torch.is_grad_enabled() or not isinstance(pre_offset, PreOffsets) or not x.is_cuda or x.dtype != torch.float32:
            return False
        if torch.is_autocast_enabled('cuda') and not _ops.bf16_autocast():
            return False   # (fp16 autocast: stock modules.  bf16 autocast -- BASELINE configs[4] -- stays on the fused path:
                           # its convolutions then run the single-piece bf16 flavour of the split kernel, ops.conv3x3)
        ok_feats = all(img_ref_feat[k].dtype == torch.float32 and img_ref_feat[k].shape[1] == c
                       for _, k, c in DynamicAggregationRestoration._STAGES)
        if not (ok_feats and _fusable_body(self.content_extractor.body) and self.dyn_agg_restore.fusable()):
            return False
        if self.dyn_agg_restore.has_inner_hooks() or _has_hooks(self.content_extractor):
            return False   # forward hooks on inner modules only fire on the module-by-module path
        # size limit of the fused kernels: 32-bit byte offsets inside ONE sample's planar offset planes [2*9*dg][4h][4w] between the
        # largest DCN head and the DCNv2 kernel that reads them (csrc/conv3x3.hip C2M_OUT_DCN_HEAD, dcn_v2.hip use_nhwc) --
        # LR <= 482 x 482 with 8 deformable groups (round 5 counted the mask planes in as well: 394).  Larger inputs run
        # module by module (correct, on stock convolution kernels) instead of raising in the middle of the forward -- with a
        # warning, once per module, so that nobody benchmarks the slow path unknowingly.
        h, w = x.shape[2:]
        dg = self.dyn_agg_restore.large_dyn_agg.deformable_groups
        if 18 * dg * (4 * h) * (4 * w) * 4 < 2 ** 31 - 1:
            return True
        if not getattr(self, "_c2m_warned_size", False):
            import warnings
            warnings.warn(f"RestorationNet: LR {h}x{w} is beyond the fused gfx950 path's 32-bit plane offsets (LR <= ~482x482 at "
                          f"{dg} deformable groups); this forward runs module by module on stock convolution kernels", RuntimeWarning)
            self._c2m_warned_size = True
        return False

    def _use_train_kernels(self, x, img_ref_feat):
        """Gradients enabled (stage-3 training): the decoder's 3x3 convolutions run forward AND backward on the hand-written
        kernels (ops.conv3x3_autograd) for fp32 GPU inputs outside autocast; `allow_fused = False` or forward hooks on inner
        modules keep the stock module-by-module path."""
        if not (self.allow_fused and torch.is_grad_enabled() and x.is_cuda and x.dtype == torch.float32):
            return False
        if not _train_kernels_wanted(x):
            return False
        if torch.is_autocast_enabled('cuda') or self.dyn_agg_restore.has_inner_hooks() or _has_hooks(self.content_extractor):
            return False
        return all(img_ref_feat[k].dtype == torch.float32 and img_ref_feat[k].shape[1] == c
                   for _, k, c in DynamicAggregationRestoration._STAGES) and \
            _fusable_body(self.content_extractor.body) and self.dyn_agg_restore.fusable()

    def forward(self, x, pre_offset, img_ref_feat):
        """x: LR image [B,3,h,w]; pre_offset / img_ref_feat: dicts keyed relu3_1 / relu2_1 / relu1_1."""
        base = F.interpolate(x, None, 4, 'bilinear', False)
        if self._use_fused(x, pre_offset, img_ref_feat):
            def fused():
                _ops.refresh_weight_caches(self)   # cached weight images follow writes through .data (no version bump)
                content_feat = self.content_extractor.forward_fused(x)
                return self.dyn_agg_restore.forward_fused(content_feat, pre_offset.flow, img_ref_feat)
            # (the f16 x 2 convolution flavour's domain is |activation| < 65520: checked on the device, bf16 x 3 re-run if left)
            return _ops.f16_range_guard(self, fused, x.device) + base
        train_kernels = self._use_train_kernels(x, img_ref_feat)
        with use_conv_kernels(train_kernels):   # the DynAgg heads follow this net's choice (thread-local, no module state)
            if train_kernels:
                content_feat = self.content_extractor.forward_train(x)
                return self.dyn_agg_restore.forward_train(content_feat, pre_offset, img_ref_feat) + base
            content_feat = self.content_extractor(x)
            return self.dyn_agg_restore(content_feat, pre_offset, img_ref_feat) + base
