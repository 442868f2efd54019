"""MI355X drop-in for ``CorrespondenceGenerationArch`` (corres_generation_arch.py:14-117).

Same constructor kwargs (YAML ``network_map`` block) and the same ``forward(dense_features, img_ref_hr) ->
(pre_offset, img_ref_feat)`` contract.  What changes underneath:

* the per-sample Python loop (:52) is gone: channel normalisation (:56-58), 3x3 patch matching (:60-67) and the
  index->flow->27 shifted/upsampled offset maps (:69-104) are three batched kernel launches;
* the index map never leaves the GPU and no intermediate flow / shifted tensors are allocated.
"""
import logging

import torch
import torch.nn as nn

from c2m_amd import ops as _ops
from mmsr.models.archs.vgg_arch import VGGFeatureExtractor

logger = logging.getLogger('base')


class PreOffsets(dict):
    """The ``pre_offset`` dict of the reference (keys relu3_1 / relu2_1 / relu1_1 -> [B, 9, s*h, s*w, 2] float32, last dim
    (x, y), s = 1 / 2 / 4; corres_generation_arch.py:69-109), built LAZILY from the arg-max index map.

    What the dict STORES is the index map (key ``max_idx``, int64 [B, h-2, w-2]) plus whichever of the three tensors have
    been asked for so far.  ``pre_offset[key]``, ``key in pre_offset`` and ``.get(key)`` behave like the reference's dict
    (the tensor is produced by one kernel launch on first access and then kept); iteration / ``items()`` show the stored
    entries only.  That is what makes the object safe to hand to a wrapped ``net_g``: DistributedDataParallel and
    DataParallel rebuild dict inputs as ``type(obj)(pairs)`` after moving / slicing every VALUE along dim 0
    (torch.distributed.utils._recursive_to, nn.parallel.scatter_gather.scatter) -- here the pairs are ``max_idx`` (batch
    along dim 0, so a DataParallel replica receives its slice) and nothing is materialised on the way; ``h, w`` are
    recovered from the map (3x3 patches at stride 1: h = hq + 2).

    The fused decoder path of ``RestorationNet`` never indexes the scales: it hands ``flow`` (index_to_flow of the whole
    batch, [B, h-2, w-2, 2]) to the DCN offset/mask head kernel, which synthesises the shifted / up-scaled / repeated
    offsets on the fly -- the three tensors (59 + 236 + 944 MB at batch 16, LR 160) are then never materialised."""

    _SCALE = {'relu3_1': 1, 'relu2_1': 2, 'relu1_1': 4}

    def __init__(self, max_idx, h=None, w=None):
        super().__init__()
        if isinstance(max_idx, torch.Tensor):
            dict.__setitem__(self, 'max_idx', max_idx)
        else:   # a mapping / an iterable of (key, value) pairs: the rebuild after a DDP / DataParallel input scatter
            for k, v in dict(max_idx).items():
                dict.__setitem__(self, k, v)
            if not dict.__contains__(self, 'max_idx'):
                raise TypeError("PreOffsets needs the arg-max index map (a tensor, or a mapping with key 'max_idx')")
        mi = dict.__getitem__(self, 'max_idx')
        self.h = int(h) if h is not None else mi.shape[-2] + 2
        self.w = int(w) if w is not None else mi.shape[-1] + 2
        self._flow = None

    @property
    def max_idx(self):
        return dict.__getitem__(self, 'max_idx')

    @property
    def flow(self):
        if self._flow is None:
            self._flow = _ops.index_to_flow(self.max_idx)
        return self._flow

    def __missing__(self, key):
        if key not in self._SCALE:
            raise KeyError(key)
        (t,) = _ops.build_pre_offsets(self.max_idx, self.h, self.w, scales=(self._SCALE[key],))
        dict.__setitem__(self, key, t)
        return t

    def __contains__(self, key):
        return key in self._SCALE or dict.__contains__(self, key)

    def get(self, key, default=None):
        return self[key] if key in self else default

    def __reduce__(self):   # pickling / copy.deepcopy: the stored entries travel, the cached flow map does not
        return (type(self), (dict(self.items()), self.h, self.w))


class CorrespondenceGenerationArch(nn.Module):

    def __init__(self, patch_size=3, stride=1, vgg_layer_list=['relu3_1', 'relu2_1', 'relu1_1'], vgg_type='vgg19'):
        super(CorrespondenceGenerationArch, self).__init__()
        self.patch_size = patch_size
        self.stride = stride
        self.vgg_layer_list = vgg_layer_list
        self.vgg = VGGFeatureExtractor(layer_name_list=vgg_layer_list, vgg_type=vgg_type)
        # fused inference path only: relu1_1 (64 channels) is what the restoration net's large DynAgg warps with 8
        # deformable groups (ref_restoration_arch.py:73-76, 176-180), i.e. 8-channel groups -- its producer also writes the
        # group-major copy that DCNv2 geometry gathers from (ops.BorderedNHWC.grouped8).  Unused by any other consumer.
        self.vgg.grouped8_taps = ('relu1_1',)
        # these taps are warped by DCNv2, they do not feed the index search: their convolutions may run on the split-bf16
        # kernel (fp32-accurate on the bf16 matrix pipe, c2m_amd.ops.conv3x3(fast=True)) like the decoder's
        self.vgg.fast_conv = True

    def index_to_flow(self, max_idx):
        """(h, w) int64 index map of ONE sample -> [1, h+2, w+2, 2] flow (x, y), zero-padded bottom/right
        (corres_generation_arch.py:29-46).  API compatibility; forward() uses the batched kernel."""
        h, w = max_idx.shape
        (flow,) = _ops.build_pre_offsets(max_idx[None].contiguous(), h + 2, w + 2, scales=(1,))
        return flow[:, 0]

    @torch.no_grad()
    def match(self, dense_features):
        """-> (max_idx int64 [B, h-2, w-2], max_val float32 [B, h-2, w-2]) for the whole batch."""
        # (.float(): under autocast the extractor hands over bf16 features; matching runs in float32)
        # (with_sumsq: the normalisation kernel also leaves the per-pixel sums of squares the matcher's patch norms need)
        feat_in = _ops.feature_normalize(dense_features['dense_features1'].float(), with_sumsq=True)
        feat_ref = _ops.feature_normalize(dense_features['dense_features2'].float(), with_sumsq=True)
        return _ops.feature_match_index_batched(feat_in, feat_ref, self.patch_size, self.stride, self.stride,
                                                is_norm=True, norm_input=True)

    def forward(self, dense_features, img_ref_hr):
        if self.patch_size != 3 or self.stride != 1:
            # the 9-shift construction of the reference (:69-104) is tied to 3x3 patches at stride 1
            raise NotImplementedError('pre-offset generation is defined for patch_size=3, stride=1')
        h, w = dense_features['dense_features1'].shape[2:]
        max_idx, _ = self.match(dense_features)
        # size: [b, 9, h, w, 2], the order of the last dim: [x, y] -- materialised on first access, see PreOffsets
        pre_offset = PreOffsets(max_idx, h, w)
        img_ref_feat = self.vgg(img_ref_hr)
        return pre_offset, img_ref_feat
