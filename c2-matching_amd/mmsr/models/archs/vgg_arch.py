"""VGG feature taps with the reference's module/parameter naming (``vgg_arch.py:59-145``): ``vgg_net.conv1_1`` ...,
buffers ``mean`` / ``std``.  When torchvision is importable the ImageNet weights are loaded exactly as the reference
does (``vgg_arch.py:104-105``); any failure of that load propagates, as in the reference.  Without torchvision the same
layer stack is built locally and its weights come from ``$C2M_VGG_WEIGHTS/<vgg_type>.pth`` (a torchvision state dict);
if that is absent too the weights are RANDOM and a warning says so loudly -- only synthetic benchmarks and parity tests,
which overwrite them with seeded values, may run in that state."""
import logging
import os
import warnings
from collections import OrderedDict

import torch
import torch.nn as nn

logger = logging.getLogger('base')

_CFG = {
    'vgg11': [64, 'M', 128, 'M', 256, 256, 'M', 512, 512, 'M', 512, 512, 'M'],
    'vgg13': [64, 64, 'M', 128, 128, 'M', 256, 256, 'M', 512, 512, 'M', 512, 512, 'M'],
    'vgg16': [64, 64, 'M', 128, 128, 'M', 256, 256, 256, 'M', 512, 512, 512, 'M', 512, 512, 512, 'M'],
    'vgg19': [64, 64, 'M', 128, 128, 'M', 256, 256, 256, 256, 'M', 512, 512, 512, 512, 'M', 512, 512, 512, 512, 'M'],
}


def layer_names(vgg_type):
    """conv{b}_{i}, relu{b}_{i}, pool{b} in torchvision's `features` order (NAMES table, vgg_arch.py:7-40)."""
    names, block, idx = [], 1, 1
    for v in _CFG[vgg_type]:
        if v == 'M':
            names.append(f'pool{block}')
            block, idx = block + 1, 1
        else:
            names += [f'conv{block}_{idx}', f'relu{block}_{idx}']
            idx += 1
    return names


NAMES = {k: layer_names(k) for k in _CFG}


def build_vgg_features(vgg_type, upto, pretrained=True):
    """OrderedDict name -> layer for layers [0, upto] of torchvision's vgg `features` stack."""
    names = NAMES[vgg_type][:upto + 1]
    tv_layers, local_state = None, None
    if pretrained:
        try:
            import torchvision.models.vgg as tv_vgg
        except ImportError:
            tv_vgg = None
        if tv_vgg is not None:
            # a failed download / changed torchvision API raises here, exactly as in the reference
            enum = getattr(tv_vgg, f'{vgg_type.upper()}_Weights', None)
            net = getattr(tv_vgg, vgg_type)(weights=enum.IMAGENET1K_V1) if enum is not None else \
                getattr(tv_vgg, vgg_type)(pretrained=True)
            tv_layers = list(net.features[:upto + 1])
        else:
            path = os.path.join(os.environ.get('C2M_VGG_WEIGHTS', ''), f'{vgg_type}.pth')
            if os.environ.get('C2M_VGG_WEIGHTS') and os.path.exists(path):
                local_state = torch.load(path, map_location='cpu')
            else:
                msg = (f'torchvision is not installed and $C2M_VGG_WEIGHTS/{vgg_type}.pth was not found: the {vgg_type} '
                       'feature stack starts from RANDOM weights.  Load a checkpoint (or pass pretrained=False) before '
                       'using it on real images.')
                logger.warning(msg)
                warnings.warn(msg, RuntimeWarning, stacklevel=2)
    out, c_in, pos = OrderedDict(), 3, 0
    for v in _CFG[vgg_type]:
        n_here = 1 if v == 'M' else 2
        if pos >= len(names):
            break
        if v == 'M':
            out[names[pos]] = tv_layers[pos] if tv_layers else nn.MaxPool2d(kernel_size=2, stride=2)
        else:
            out[names[pos]] = tv_layers[pos] if tv_layers else nn.Conv2d(c_in, v, kernel_size=3, padding=1)
            if pos + 1 < len(names):
                out[names[pos + 1]] = tv_layers[pos + 1] if tv_layers else nn.ReLU(inplace=True)
            c_in = v
        pos += n_here
    if local_state is not None:   # torchvision key layout: features.<position>.{weight,bias}
        for k, name in enumerate(names):
            if isinstance(out[name], nn.Conv2d):
                out[name].weight.data.copy_(local_state[f'features.{k}.weight'])
                out[name].bias.data.copy_(local_state[f'features.{k}.bias'])
    return out


def _bf16_autocast():
    from c2m_amd import ops as _ops
    return _ops.bf16_autocast()


class VGGFeatureExtractor(nn.Module):
    """Returns {layer_name: feature} for the requested taps (vgg_arch.py:59-145)."""

    def __init__(self, layer_name_list, vgg_type='vgg19', use_input_norm=True, requires_grad=False,
                 remove_pooling=False, pooling_stride=2):
        super().__init__()
        if 'bn' in vgg_type:
            raise NotImplementedError('batch-norm VGG variants are not used by C2-Matching')
        self.layer_name_list = layer_name_list
        self.use_input_norm = use_input_norm
        self.names = NAMES[vgg_type]
        last = max(self.names.index(n) for n in layer_name_list)
        stack = OrderedDict()
        for name, layer in build_vgg_features(vgg_type, last).items():
            if 'pool' in name:
                if remove_pooling:
                    continue
                layer = nn.MaxPool2d(kernel_size=2, stride=pooling_stride)
            stack[name] = layer
        self.vgg_net = nn.Sequential(stack)
        if not requires_grad:
            for p in self.parameters():
                p.requires_grad = False
        if use_input_norm:  # statistics for images in [0, 1]
            self.register_buffer('mean', torch.Tensor([0.485, 0.456, 0.406]).view(1, 3, 1, 1))
            self.register_buffer('std', torch.Tensor([0.229, 0.224, 0.225]).view(1, 3, 1, 1))

    def _use_fused(self, x):
        """Inference on the gfx950 channels-last kernels: no autograd, fp32 image on the GPU, the plain torchvision stack
        (3x3 / stride 1 / pad 1 convolutions, ReLUs, 2x2 max-pools) and post-ReLU taps only."""
        def ok(m):
            if isinstance(m, nn.Conv2d):
                return (m.kernel_size == (3, 3) and m.stride == (1, 1) and m.padding == (1, 1) and m.groups == 1 and
                        (m.in_channels == 3 or m.in_channels % 32 == 0))
            if isinstance(m, nn.MaxPool2d):
                return m.kernel_size in (2, (2, 2)) and m.stride in (2, (2, 2))
            return isinstance(m, nn.ReLU)
        return (not torch.is_grad_enabled() and x.is_cuda and x.dtype == torch.float32 and x.dim() == 4 and x.shape[1] == 3 and
                (not torch.is_autocast_enabled('cuda') or _bf16_autocast()) and all(ok(m) for m in self.vgg_net._modules.values()) and
                all(n.startswith('relu') for n in self.layer_name_list))

    def forward(self, x):
        if self._use_fused(x):
            # inference on the gfx950 channels-last kernels: conv + ReLU in one launch, tapped activations written straight
            # into the zero-bordered channels-last buffers the DCNv2 warps gather from (no clone, no layout copy)
            from c2m_amd import ops as _ops
            return _ops.f16_range_guard(self, lambda: _ops.vgg_stack_forward(
                self.vgg_net._modules, x, taps=self.layer_name_list, mean=self.mean if self.use_input_norm else None,
                std=self.std if self.use_input_norm else None, grouped8_taps=getattr(self, 'grouped8_taps', ()),
                fast=getattr(self, 'fast_conv', False)), x.device)
        if self.use_input_norm:
            x = (x - self.mean) / self.std
        taps = {}
        for name, layer in self.vgg_net._modules.items():
            x = layer(x)
            if name in self.layer_name_list:
                taps[name] = x.clone()  # the next ReLU is in-place
        return taps
