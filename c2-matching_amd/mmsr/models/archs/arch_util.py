"""The pieces of the reference's ``arch_util.py`` that the restoration hot path touches: ``tensor_shift``
(arch_util.py:291-315), ``ResidualBlockNoBN`` (:80-136), ``make_layer`` (:63-77) and the two weight initialisers
(:12-61).  Plain torch modules -- the decoder convolutions run on MIOpen."""
import torch
import torch.nn as nn
import torch.nn.init as init
from torch.nn.modules.batchnorm import _BatchNorm


def srntt_init_weights(net, init_type='normal', init_gain=0.02):
    """N(0, gain) (or xavier / kaiming / orthogonal) on every Conv/Linear weight, zero bias (arch_util.py:12-34)."""
    fillers = {
        'normal': lambda w: init.normal_(w, 0.0, init_gain),
        'xavier': lambda w: init.xavier_normal_(w, gain=init_gain),
        'kaiming': lambda w: init.kaiming_normal_(w, a=0, mode='fan_in'),
        'orthogonal': lambda w: init.orthogonal_(w, gain=init_gain),
    }
    if init_type not in fillers:
        raise NotImplementedError(f'initialization method [{init_type}] is not implemented')

    def visit(m):
        kind = type(m).__name__
        if hasattr(m, 'weight') and ('Conv' in kind or 'Linear' in kind):
            fillers[init_type](m.weight.data)
            if getattr(m, 'bias', None) is not None:
                init.constant_(m.bias.data, 0.0)
        elif 'BatchNorm2d' in kind:
            init.normal_(m.weight.data, 1.0, init_gain)
            init.constant_(m.bias.data, 0.0)

    net.apply(visit)


def default_init_weights(module_list, scale=1):
    """Kaiming-normal (fan_in) scaled by `scale`, zero bias; BatchNorm to (1, 0) (arch_util.py:37-61)."""
    for root in (module_list if isinstance(module_list, list) else [module_list]):
        for m in root.modules():
            if isinstance(m, (nn.Conv2d, nn.Linear)):
                init.kaiming_normal_(m.weight, a=0, mode='fan_in')
                m.weight.data *= scale
                if m.bias is not None:
                    m.bias.data.zero_()
            elif isinstance(m, _BatchNorm):
                init.constant_(m.weight, 1)
                init.constant_(m.bias.data, 0.0)


def make_layer(basic_block, n_basic_blocks, **kwarg):
    return nn.Sequential(*[basic_block(**kwarg) for _ in range(n_basic_blocks)])


class ResidualBlockNoBN(nn.Module):
    """x + res_scale * conv2(relu(conv1(x))), 3x3 convs, no normalisation (arch_util.py:80-136; the spectral-norm
    variants are not used by C2-Matching's restoration net and are not provided)."""

    def __init__(self, nf=64, res_scale=1, pytorch_init=False):
        super().__init__()
        self.res_scale = res_scale
        self.conv1 = nn.Conv2d(nf, nf, 3, 1, 1, bias=True)
        self.conv2 = nn.Conv2d(nf, nf, 3, 1, 1, bias=True)
        self.relu = nn.ReLU(inplace=True)
        if not pytorch_init:
            default_init_weights([self.conv1, self.conv2], 0.1)

    def forward(self, x):
        return x + self.conv2(self.relu(self.conv1(x))) * self.res_scale


def tensor_shift(x, shift=(2, 2), fill_val=0):
    """[b, h, w, c] shifted down/right by `shift` pixels, vacated border = fill_val (arch_util.py:291-315).
    Kept for API compatibility; CorrespondenceGenerationArch builds all 27 shifted maps in one kernel instead."""
    sh, sw = shift
    if sh < 0 or sw < 0:
        raise NotImplementedError
    _, h, w, _ = x.size()
    out = torch.full_like(x, fill_val)
    out[:, sh:, sw:, :] = x[:, :h - sh, :w - sw, :]
    return out
