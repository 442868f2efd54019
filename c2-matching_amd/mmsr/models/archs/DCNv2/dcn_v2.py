"""MI355X drop-in for the reference's ``mmsr/models/archs/DCNv2/dcn_v2.py`` (operator signatures of dcn_v2.py:16-253).

Module path, class names, constructor arguments, parameter / sub-module names (= checkpoint keys: ``weight``, ``bias``,
``conv_offset_mask.{weight,bias}``) and initialisation follow the reference.  The compute goes through ``_ext``
(this package's C-ABI binding) instead of the reference's CUDA extension.  Differences that are deliberate:

* offset/mask assembly of the ``*_sep*`` modules (chunk, cat, repeat over groups, (x,y)->(y,x) interleave, add,
  sigmoid -- dcn_v2.py:229-245) is one fused kernel (c2m_dcn_fuse_offsets_f32) instead of ~6 elementwise passes;
* the "offset mean > 100" warning (dcn_v2.py:247-250) no longer blocks the stream: the mean is reduced on the device and
  read back asynchronously; the warning is emitted at a later call once the value has arrived;
* under ``torch.autocast`` (BASELINE config 5: bf16 inference) the operators take their inputs as float32 and compute in
  float32 (the reference's extension reads ``.data<float>()`` and would misread half tensors): the surrounding
  convolutions run in bf16, the sampling positions and the warp do not.
"""
import logging
import math

import _ext as _backend
import threading

import torch
from torch import nn
from torch.autograd import Function
from torch.amp import custom_bwd, custom_fwd
from torch.autograd.function import once_differentiable
from torch.nn.modules.utils import _pair

from c2m_amd import ops as _ops

logger = logging.getLogger('base')
_ABS_SLOTS = 256  # C2M_ABS_SUM_SLOTS of include/c2m_hip.h


class _DCNv2(Function):
    """autograd wrapper with the reference's argument order (dcn_v2.py:16-50)."""

    @staticmethod
    @custom_fwd(device_type='cuda', cast_inputs=torch.float32)
    def forward(ctx, input, offset, mask, weight, bias, stride, padding, dilation, deformable_groups, bf16_mma=False):
        ctx.geom = (_pair(weight.shape[2:4]), _pair(stride), _pair(padding), _pair(dilation), int(deformable_groups))
        (kh, kw), (sh, sw), (ph, pw), (dh, dw), dg = ctx.geom
        output = _backend.dcn_v2_forward(input, weight, bias, offset, mask, kh, kw, sh, sw, ph, pw, dh, dw, dg,
                                         bf16_mma=bool(bf16_mma))
        ctx.save_for_backward(input, offset, mask, weight, bias)
        return output

    @staticmethod
    @once_differentiable
    @custom_bwd(device_type='cuda')
    def backward(ctx, grad_output):
        input, offset, mask, weight, bias = ctx.saved_tensors
        (kh, kw), (sh, sw), (ph, pw), (dh, dw), dg = ctx.geom
        g_in, g_off, g_mask, g_w, g_b = _backend.dcn_v2_backward(input, weight, bias, offset, mask,
                                                                 grad_output.contiguous(), kh, kw, sh, sw, ph, pw,
                                                                 dh, dw, dg,
                                                                 need_input_grad=ctx.needs_input_grad[0])
        return g_in, g_off, g_mask, g_w, g_b, None, None, None, None, None


dcn_v2_conv = _DCNv2.apply


def _bf16_autocast():
    """True inside `torch.autocast('cuda', dtype=torch.bfloat16)`: the caller asked for reduced precision, so the DCNv2
    GEMM may run on bf16 MFMA (sampling positions, bilinear blend and accumulation stay float32)."""
    return torch.is_autocast_enabled('cuda') and torch.get_autocast_dtype('cuda') == torch.bfloat16


class _FusedOffsets(Function):
    """conv_offset_mask output (+ pre-offset) -> (offset, mask) in one kernel; backward is two elementwise torch ops."""

    @staticmethod
    @custom_fwd(device_type='cuda', cast_inputs=torch.float32)
    def forward(ctx, conv_out, pre_offset, deformable_groups, taps, abs_sum_bits):
        # abs_sum travels as an int64 view of the float64 accumulator: custom_fwd would down-cast a float64 argument
        abs_sum = None if abs_sum_bits is None else abs_sum_bits.view(torch.float64)
        offset, mask = _ops.dcn_fuse_offsets(conv_out, pre_offset, deformable_groups, taps, abs_sum)
        ctx.save_for_backward(mask)
        return offset, mask

    @staticmethod
    @once_differentiable
    @custom_bwd(device_type='cuda')
    def backward(ctx, g_offset, g_mask):
        (mask,) = ctx.saved_tensors
        return torch.cat((g_offset, g_mask * mask * (1 - mask)), dim=1), None, None, None, None


class FusedPreOffset:
    """Argument of the fused inference path of ``DCN_sep_pre_multi_offset`` in place of the ``pre_offset`` tensor: the
    un-padded flow map of the arg-max indices (``c2m_amd.ops.index_to_flow``, [B, h-2, w-2, 2]) + the scale of this layer
    (H / h = 1, 2, 4); the offset/mask head kernel synthesises the [B, 9, H, W, 2] pre-offsets from it on the fly.
    ``lrelu_slope``: fold the LeakyReLU that follows every DynAgg (ref_restoration_arch.py:152-154) into the output."""

    def __init__(self, flow, scale, lrelu_slope=None):
        self.flow, self.scale, self.lrelu_slope = flow, int(scale), lrelu_slope


class _OffsetMeanWatch:
    """Deferred version of the reference's `offset_mean > 100` check: no host sync inside forward."""

    def __init__(self):
        self._pending = None

    # the in-flight read-back (pinned buffer + event) is per-process scratch: modules stay picklable / deep-copyable
    # (torch.save(net), copy.deepcopy(net), DataParallel replicas) like the reference's
    def __getstate__(self):
        return {}

    def __setstate__(self, state):
        self._pending = None

    def __deepcopy__(self, memo):
        return _OffsetMeanWatch()

    def poll(self):
        pending = self._pending   # read once: DataParallel replicas share this object across threads
        if pending is not None and not torch.cuda.is_current_stream_capturing():
            host, event, numel = pending
            if event.query():
                mean = float(host.sum()) / numel
                if mean > 100:
                    logger.warning('Offset mean is {}, larger than 100.'.format(mean))
                if self._pending is pending:
                    self._pending = None

    def push(self, abs_sum, numel):
        if torch.cuda.is_current_stream_capturing():
            return   # (a hipGraph capture of the training step: no pinned allocation / event query inside it)
        if self._pending is None:
            host = torch.empty(_ABS_SLOTS, dtype=torch.float64, pin_memory=True)
            host.copy_(abs_sum, non_blocking=True)
            event = torch.cuda.Event()
            event.record()
            self._pending = (host, event, numel)


def _check_geometry(in_channels, out_channels, deformable_groups):
    """The gfx950 kernels cover a subset of the geometries the reference's generic CUDA kernels accept; say so when the
    module is built instead of at the first forward / backward (C2M_ERR_UNSUPPORTED)."""
    if in_channels % deformable_groups != 0:
        raise ValueError(f'in_channels ({in_channels}) must be divisible by deformable_groups ({deformable_groups})')
    cpg = in_channels // deformable_groups
    if cpg % 2 != 0:
        raise NotImplementedError(f'DCNv2 on MI355X needs an even number of channels per deformable group, got '
                                  f'{in_channels}/{deformable_groups} = {cpg}')
    if cpg % 4 != 0 or out_channels > 256:
        logger.warning(f'DCNv2({in_channels}, {out_channels}, deformable_groups={deformable_groups}): forward is '
                       'supported, BACKWARD is not (needs channels-per-group % 4 == 0 and out_channels <= 256); '
                       'training this layer will raise.')


class DCNv2(nn.Module):
    """Modulated deformable convolution with externally supplied offset and mask (dcn_v2.py:56-95)."""

    def __init__(self, in_channels, out_channels, kernel_size, stride, padding, dilation=1, deformable_groups=1):
        super(DCNv2, self).__init__()
        self.in_channels = in_channels
        self.out_channels = out_channels
        self.kernel_size = _pair(kernel_size)
        self.stride = _pair(stride)
        self.padding = _pair(padding)
        self.dilation = _pair(dilation)
        self.deformable_groups = deformable_groups
        _check_geometry(in_channels, out_channels, deformable_groups)
        self.weight = nn.Parameter(torch.Tensor(out_channels, in_channels, *self.kernel_size))
        self.bias = nn.Parameter(torch.Tensor(out_channels))
        self.reset_parameters()

    def reset_parameters(self):
        fan_in = self.in_channels * self.kernel_size[0] * self.kernel_size[1]
        bound = 1. / math.sqrt(fan_in)
        self.weight.data.uniform_(-bound, bound)
        self.bias.data.zero_()

    @property
    def _taps(self):
        return self.kernel_size[0] * self.kernel_size[1]

    def _conv(self, x, offset, mask):
        return dcn_v2_conv(x, offset, mask, self.weight, self.bias, self.stride, self.padding, self.dilation,
                           self.deformable_groups, _bf16_autocast())

    def forward(self, input, offset, mask):
        assert 2 * self.deformable_groups * self._taps == offset.shape[1]
        assert self.deformable_groups * self._taps == mask.shape[1]
        return self._conv(input, offset, mask)


_kernel_choice = threading.local()


class use_conv_kernels:
    """`with use_conv_kernels(True / False): ...` -- whether the offset/mask heads of the DCN modules called inside the block
    (on this thread) run the hand-written autograd kernels (c2m_amd.ops.conv3x3_autograd) or the stock nn.Conv2d."""

    def __init__(self, on):
        self.on = bool(on)

    def __enter__(self):
        self.prev = getattr(_kernel_choice, 'on', None)
        _kernel_choice.on = self.on
        return self

    def __exit__(self, *exc):
        if self.prev is None:
            del _kernel_choice.on
        else:
            _kernel_choice.on = self.prev
        return False


class _SelfOffsetDCN(DCNv2):
    """Shared machinery of DCN / DCN_sep / DCN_sep_pre_multi_offset: a `conv_offset_mask` head (zero-initialised,
    dg*3*kh*kw channels, same kernel/stride/padding as the main conv -- dcn_v2.py:112-124) feeding the fused
    offset/mask assembly."""

    def __init__(self, in_channels, out_channels, kernel_size, stride, padding, dilation=1, deformable_groups=1):
        super(_SelfOffsetDCN, self).__init__(in_channels, out_channels, kernel_size, stride, padding, dilation,
                                             deformable_groups)
        self.conv_offset_mask = nn.Conv2d(self.in_channels, self.deformable_groups * 3 * self._taps,
                                          kernel_size=self.kernel_size, stride=self.stride, padding=self.padding,
                                          bias=True)
        self.init_offset()
        self._watch = _OffsetMeanWatch()

    #: Default policy for the offset/mask head under autograd: False = the stock nn.Conv2d (double backward works, MIOpen is
    #: the faster choice on small maps); True = the hand-written forward / backward kernels.  A parent that has measured
    #: the choice for its sizes (RestorationNet: $C2M_TRAIN_KERNELS / TRAIN_KERNELS_MIN_PIXELS) overrides it for the calls it
    #: makes with `use_conv_kernels(...)` -- a thread-local context, not a mutation of this (shared) module.
    allow_conv_kernels = False

    def init_offset(self):
        self.conv_offset_mask.weight.data.zero_()
        self.conv_offset_mask.bias.data.zero_()

    def _offset_and_mask(self, feat, pre_offset=None, watch=False):
        head = self.conv_offset_mask
        if (getattr(_kernel_choice, 'on', self.allow_conv_kernels) and torch.is_grad_enabled() and feat.is_cuda and feat.dtype == torch.float32 and not torch.is_autocast_enabled('cuda') and
                head.kernel_size == (3, 3) and head.stride == (1, 1) and head.padding == (1, 1) and
                not head._forward_hooks and not head._forward_pre_hooks and _ops.conv3x3_autograd_ok([feat], head.weight)):
            raw = _ops.conv3x3_autograd([feat], head.weight, head.bias)   # hand-written forward / backward kernels (training)
        else:
            raw = head(feat)
        abs_sum = None
        if watch:
            self._watch.poll()
            abs_sum = torch.zeros(_ABS_SLOTS, dtype=torch.float64, device=raw.device)
        offset, mask = _FusedOffsets.apply(raw, pre_offset, self.deformable_groups, self._taps,
                                           None if abs_sum is None else abs_sum.view(torch.int64))
        if watch:
            self._watch.push(abs_sum, offset.numel())
        return offset, mask


class DCN(_SelfOffsetDCN):
    """Offsets and mask predicted from the input itself (dcn_v2.py:98-133)."""

    def forward(self, input):
        offset, mask = self._offset_and_mask(input)
        return self._conv(input, offset, mask)


class DCN_sep(_SelfOffsetDCN):
    '''Use other features to generate offsets and masks (dcn_v2.py:136-184).'''

    def __init__(self, in_channels, out_channels, kernel_size, stride, padding, dilation=1, deformable_groups=1,
                 extra_offset_mask=True):
        super(DCN_sep, self).__init__(in_channels, out_channels, kernel_size, stride, padding, dilation,
                                      deformable_groups)
        self.extra_offset_mask = extra_offset_mask

    def forward(self, x):
        feat = x
        if self.extra_offset_mask:
            x, feat = x[0], x[1]   # x = [input, features]
        offset, mask = self._offset_and_mask(feat, watch=True)
        return self._conv(x, offset, mask)


class DCN_sep_pre_multi_offset(_SelfOffsetDCN):
    '''
    Use other features to generate offsets and masks.

    Intialized the offset with precomputed non-local offset (dcn_v2.py:187-253).
    '''

    def __init__(self, in_channels, out_channels, kernel_size, stride, padding, dilation=1, deformable_groups=1,
                 extra_offset_mask=True):
        super(DCN_sep_pre_multi_offset, self).__init__(in_channels, out_channels, kernel_size, stride, padding,
                                                       dilation, deformable_groups)
        self.extra_offset_mask = extra_offset_mask

    def forward(self, x, pre_offset):
        '''
        Args:
            pre_offset: precomputed_offset. Size: [b, 9, h, w, 2], last dim (x, y)
        '''
        if isinstance(pre_offset, FusedPreOffset):
            return self._forward_fused(x[0], x[1], pre_offset)
        feat = x
        if self.extra_offset_mask:
            x, feat = x[0], x[1]   # x = [input, features]
        offset, mask = self._offset_and_mask(feat, pre_offset, watch=True)
        return self._conv(x, offset, mask)

    @torch.no_grad()
    def _forward_fused(self, ref, feat, pre):
        """Inference on the channels-last kernels.  ref: c2m_amd.ops.BorderedNHWC of the feature to warp (shared with the
        caller's offset convolutions); feat: channels-last offset feature; -> channels-last output.  One launch for the
        head (conv + bias + pre-offset synthesis + sigmoid + the warning's |offset| sum), one for the warp."""
        if self.kernel_size != (3, 3) or self.stride != (1, 1) or self.padding != (1, 1) or self.dilation != (1, 1):
            raise NotImplementedError('fused DynAgg path: 3x3 / stride 1 / pad 1 only')
        head = self.conv_offset_mask
        self._watch.poll()
        abs_sum = torch.zeros(_ABS_SLOTS, dtype=torch.float64, device=feat.device)
        offset, mask = _ops.conv3x3_dcn_head(feat, head.weight, head.bias, self.deformable_groups, pre.flow, pre.scale, abs_sum)
        self._watch.push(abs_sum, offset.numel())
        act = _ops.ACT_NONE if pre.lrelu_slope is None else _ops.ACT_LRELU
        return _ops.dcn_v2_forward_nhwc(ref, self.weight, self.bias, offset, mask, self.deformable_groups, act=act,
                                        slope=pre.lrelu_slope or 0.0)


class _NotOnThisPath(nn.Module):
    def __init__(self, *args, **kwargs):
        super().__init__()
        raise NotImplementedError("deformable PS-ROI pooling (dcn_v2.py:256-411) is never instantiated by C2-Matching "
                                  "and is not part of the MI355X hot path")


class DCNv2Pooling(_NotOnThisPath):
    pass


class DCNPooling(_NotOnThisPath):
    pass


def dcn_v2_pooling(*args, **kwargs):
    raise NotImplementedError("deformable PS-ROI pooling is not part of the MI355X hot path")
