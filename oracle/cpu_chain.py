"""CPU restatement of the WHOLE restoration forward (extractor -> correlation / index map -> pre-offsets -> VGG taps ->
RestorationNet with the three DynAgg warps) for ONE batch, built from the oracle ops.

TEST INFRASTRUCTURE / CPU BASELINE ONLY (tests/, __graft_entry__.smoke(), bench.py's cpu_baseline leg): nothing under
c2-matching_amd/ imports this file.  It is "the reference's PyTorch-CPU path" as far as one exists: the plain
convolutions are the same stock torch modules the reference uses (on the CPU), the correlation is the reference's
conv2d-filter algorithm (oracle/torch_port.py, ref_map_util.py:26-86), the pre-offset construction and DCNv2 -- which
the reference cannot run on a CPU at all (DCNv2/src/dcn_v2.h:38,72 throw) -- are the C oracle.
"""
import copy
import time
import types

import numpy as np
import torch
import torch.nn.functional as F

import c2m_oracle as oracle
import torch_port


def _dynagg_forward_cpu(self, x, pre_offset):
    """DCN_sep_pre_multi_offset.forward (dcn_v2.py:222-253) on CPU tensors through the oracle."""
    inp, feat = x[0], x[1]
    dg, K = self.deformable_groups, self.kernel_size[0] * self.kernel_size[1]
    out = self.conv_offset_mask(feat)
    o1, o2, mask = torch.chunk(out, 3, dim=1)
    offset = torch.cat((o1, o2), dim=1)
    pre = pre_offset.flip(-1)                                    # (x, y) -> (y, x)            dcn_v2.py:236-240
    pre = pre.permute(0, 1, 4, 2, 3).reshape(pre.shape[0], 2 * K, pre.shape[2], pre.shape[3]).repeat(1, dg, 1, 1)
    offset = offset + pre
    mask = torch.sigmoid(mask)
    o = oracle.dcn_v2_forward(inp.numpy(), self.weight.detach().numpy(), self.bias.detach().numpy(), offset.numpy(),
                              mask.numpy(), self.stride, self.padding, self.dilation, dg)
    return torch.from_numpy(o)


def cpu_copy(net_g):
    """Deep copy of a RestorationNet on the CPU whose DynAgg modules compute through the oracle."""
    g = copy.deepcopy(net_g).cpu().eval()
    for stage in ("small", "medium", "large"):
        m = getattr(g.dyn_agg_restore, f"{stage}_dyn_agg")
        m.forward = types.MethodType(_dynagg_forward_cpu, m)
    return g


def mismatch_margins(feats1, feats2, idx_a, idx_b):
    """For every query where two index maps of ONE pair disagree: float64 score of both picks on the given (CPU) features,
    as ref_map_util.py:52-76 scores them (ref patch divided by its norm + 1e-5).  -> list of (query, pick_a, pick_b,
    score_a - score_b): |margin| at fp32 rounding level (<~ 5e-6) = a near-tie either implementation of
    the extractor convolutions may resolve either way."""
    C, h, w = feats1.shape
    f1 = F.normalize(feats1.reshape(C, -1), dim=0).view(C, h, w).double().numpy()
    f2 = F.normalize(feats2.reshape(C, -1), dim=0).view(C, h, w).double().numpy()
    wq = w - 2
    out = []
    for q in np.argwhere(np.asarray(idx_a) != np.asarray(idx_b)):
        qy, qx = int(q[0]), int(q[1])
        patch = f1[:, qy:qy + 3, qx:qx + 3]
        sc = []
        for n in (int(idx_a[qy, qx]), int(idx_b[qy, qx])):
            ry, rx = divmod(n, wq)
            r = f2[:, ry:ry + 3, rx:rx + 3]
            sc.append(float((patch * r).sum() / (np.sqrt((r * r).sum()) + 1e-5)))
        out.append(((qy, qx), int(idx_a[qy, qx]), int(idx_b[qy, qx]), sc[0] - sc[1]))
    return out


@torch.no_grad()
def correspondence_cpu(feats1, feats2, use_conv_algorithm=True):
    """dense features [B,C,h,w] x2 -> (max_idx int64 [B,h-2,w-2], pre_offset dict) as corres_generation_arch.py:48-117."""
    B, _, h, w = feats1.shape
    idxs, offs = [], []
    for b in range(B):
        f1 = F.normalize(feats1[b].reshape(feats1.shape[1], -1), dim=0).view(feats1.shape[1], h, w)
        f2 = F.normalize(feats2[b].reshape(feats2.shape[1], -1), dim=0).view(feats2.shape[1], h, w)
        if use_conv_algorithm:
            idx, _ = torch_port.feature_match_index_conv(f1, f2, 3, 1, 1, True, True)
            idx = idx.numpy().astype(np.int64)
        else:
            idx, _ = oracle.feature_match_index(f1.numpy(), f2.numpy(), 3, 1, 1, True, True)
        idxs.append(idx)
        offs.append(oracle.build_pre_offsets(idx, h, w))
    pre = {"relu3_1": torch.from_numpy(np.stack([o[0] for o in offs])),
           "relu2_1": torch.from_numpy(np.stack([o[1] for o in offs])),
           "relu1_1": torch.from_numpy(np.stack([o[2] for o in offs]))}
    return np.stack(idxs), pre


@torch.no_grad()
def _pre_offsets_of(idx_map, h, w):
    offs = [oracle.build_pre_offsets(np.asarray(idx_map[b], dtype=np.int64), h, w) for b in range(len(idx_map))]
    return {"relu3_1": torch.from_numpy(np.stack([o[0] for o in offs])),
            "relu2_1": torch.from_numpy(np.stack([o[1] for o in offs])),
            "relu1_1": torch.from_numpy(np.stack([o[2] for o in offs]))}


@torch.no_grad()
def full_forward_cpu(ext, mp, net_g, lq, up, ref, use_conv_algorithm=True, timings=None, idx_for_offsets=None, cond_idx=None):
    """-> (sr [B,3,4h,4w], max_idx, feats dict).  ext / mp / net_g are the (GPU or CPU) modules whose weights to use.
    idx_for_offsets: build the pre-offsets from this index map [B,h-2,w-2] instead of the CPU one (isolates the decoder
    from fp32 near-tie flips between two convolution implementations of the extractor); the CPU map is still returned.
    cond_idx: AFTER the timed forward, also evaluate the decoder with the pre-offsets of this index map (only if it differs
    from the CPU map -- otherwise the result is the same tensor) -> feats["sr_given_idx"]."""
    ext_c, mp_c, g_c = copy.deepcopy(ext).cpu().eval(), copy.deepcopy(mp).cpu().eval(), cpu_copy(net_g)
    lq, up, ref = lq.detach().float().cpu(), up.detach().float().cpu(), ref.detach().float().cpu()
    t0 = time.perf_counter()
    feats = ext_c(up, ref)
    t1 = time.perf_counter()
    idx, pre = correspondence_cpu(feats["dense_features1"], feats["dense_features2"], use_conv_algorithm)
    h, w = feats["dense_features1"].shape[2:]
    if idx_for_offsets is not None:
        pre = _pre_offsets_of(idx_for_offsets, h, w)
    ref_feat = mp_c.vgg(ref)
    t2 = time.perf_counter()
    sr = g_c(lq, pre, ref_feat)
    t3 = time.perf_counter()
    if timings is not None:
        timings.update({"extractor_s": t1 - t0, "correspondence_s": t2 - t1, "restoration_s": t3 - t2})
    if cond_idx is not None:
        same = np.array_equal(np.asarray(cond_idx, dtype=np.int64), np.asarray(idx, dtype=np.int64))
        feats = dict(feats)
        feats["sr_given_idx"] = sr if same else g_c(lq, _pre_offsets_of(cond_idx, h, w), ref_feat)
    return sr, idx, feats
