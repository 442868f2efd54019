"""``RestorationNet`` (ref_restoration_arch.py:30-65) with the reference's parameter names (checkpoint keys listed in
SURVEY.md A.3).  The three ``*_dyn_agg`` sites are DCN_sep_pre_multi_offset layers (:77-85, :101-109, :124-132) and run
on the gfx950 DCNv2 kernels; the plain 3x3 convolutions / residual blocks / pixel shuffles stay stock torch (MIOpen)."""
import torch
import torch.nn as nn
import torch.nn.functional as F

import mmsr.models.archs.arch_util as arch_util
from mmsr.models.archs.DCNv2.dcn_v2 import DCN_sep_pre_multi_offset as DynAgg


class ContentExtractor(nn.Module):

    def __init__(self, in_nc=3, out_nc=3, nf=64, n_blocks=16):
        super(ContentExtractor, self).__init__()
        self.conv_first = nn.Conv2d(in_nc, nf, 3, 1, 1)
        self.body = arch_util.make_layer(arch_util.ResidualBlockNoBN, n_blocks, nf=nf)
        self.lrelu = nn.LeakyReLU(negative_slope=0.1, inplace=True)
        arch_util.default_init_weights([self.conv_first], 0.1)

    def forward(self, x):
        return self.body(self.lrelu(self.conv_first(x)))


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

    def forward(self, x, pre_offset, img_ref_feat):
        """x: LR image [B,3,h,w]; pre_offset / img_ref_feat: dicts keyed relu3_1 / relu2_1 / relu1_1."""
        base = F.interpolate(x, None, 4, 'bilinear', False)
        content_feat = self.content_extractor(x)
        return self.dyn_agg_restore(content_feat, pre_offset, img_ref_feat) + base
