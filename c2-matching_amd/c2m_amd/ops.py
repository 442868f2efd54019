"""Tensor-level entry points of the MI355X hot path (PyTorch is plumbing only: memory, streams).

Every function takes CUDA(=HIP) float32 tensors, enqueues hand-written gfx950 kernels from libc2m_hip.so on the
current torch stream and returns freshly allocated tensors.  CPU tensors are rejected: there is no fallback path.
"""
import torch

from . import _lib


def _dev_f32(t, name):
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise _lib.C2MError(f"{name} must be a tensor on the GPU (the HIP path has no CPU fallback)")
    if t.dtype != torch.float32:
        raise _lib.C2MError(f"{name} must be float32, got {t.dtype}")
    return t if t.is_contiguous() else t.contiguous()


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _pair(v):
    return (int(v), int(v)) if isinstance(v, int) else (int(v[0]), int(v[1]))


class NormalizedFeatures(torch.Tensor):
    """What `feature_normalize(x, with_sumsq=True)` returns: the normalised map (an ordinary tensor to every consumer) that also
    carries `.c2m_sumsq` -- its per-pixel sums of squares [B, H*W], formed by the normalisation kernel while the values were in
    registers.  `feature_match_index_batched` hands them to the library instead of launching a pass over each map; anything
    derived from the tensor (a slice, a copy, arithmetic) is a plain tensor again and simply loses the shortcut."""

    @staticmethod
    def wrap(t, ss):
        r = t.as_subclass(NormalizedFeatures)
        r.c2m_sumsq = ss
        r.c2m_version = r._version      # (an in-place write to the map bumps _version: the sums are then stale and are not used)
        return r

    @classmethod
    def __torch_function__(cls, func, types, args=(), kwargs=None):
        with torch._C.DisableTorchFunctionSubclass():
            return func(*args, **(kwargs or {}))


def feature_normalize(x, with_sumsq=False):
    """x [B,C,H,W] or [C,H,W] -> per-pixel channel-normalised copy (corres_generation_arch.py:56-58).  with_sumsq: the result is a
    NormalizedFeatures (same values; carries the per-pixel sums of squares for feature_match_index_batched)."""
    x = _dev_f32(x, "x")
    shp = x.shape
    xb = x.view(1, *shp) if x.dim() == 3 else x
    B, C = xb.shape[0], xb.shape[1]
    out = torch.empty_like(xb)
    HW = xb.numel() // (B * C)
    ss = torch.empty((B, HW), dtype=torch.float32, device=x.device) if with_sumsq else None
    with torch.cuda.device(x.device):
        _lib.check(_lib.lib().c2m_feature_normalize_ss_f32(_stream(), xb.data_ptr(), B, C, HW, out.data_ptr(),
                                                           ss.data_ptr() if ss is not None else None), "c2m_feature_normalize_ss_f32")
    out = out.view(shp)
    return NormalizedFeatures.wrap(out, ss) if with_sumsq else out


def _pre_sumsq(t, B, HW):
    """The sums of squares a NormalizedFeatures carries, if `t` still IS that very tensor (same storage, shape, contiguous)."""
    ss = getattr(t, "c2m_sumsq", None) if isinstance(t, NormalizedFeatures) else None
    if ss is None or not t.is_contiguous() or tuple(ss.shape) != (B, HW) or ss.device != t.device or t._version != getattr(t, "c2m_version", -1):
        return None
    return ss


def feature_match_index_batched(feat_in, feat_ref, patch_size=3, input_stride=1, ref_stride=1, is_norm=True,
                                norm_input=False, force_generic=False, return_skip=False):
    """Batched ref_map_util.feature_match_index: feat_in [B,C,Hq,Wq], feat_ref [B,C,Hr,Wr] ->
    (max_idx int64 [B,Hqp,Wqp], max_val float32 [B,Hqp,Wqp]).  return_skip=True appends the MFMA kernel's duplicate-row
    table, int32 [B, x_tiles, 2] = (from, to) per (sample, ref x-tile) (c2m_feature_match_skip_table; diagnostics)."""
    fi, fr = _dev_f32(feat_in, "feat_in"), _dev_f32(feat_ref, "feat_ref")
    if fi.dim() != 4 or fr.dim() != 4 or fi.shape[:2] != fr.shape[:2] or fi.device != fr.device:
        raise _lib.C2MError("feat_in / feat_ref must be [B,C,H,W] with equal B, C and device")
    B, C, Hq, Wq = fi.shape
    Hr, Wr = fr.shape[2:]
    p, si, sr = int(patch_size), int(input_stride), int(ref_stride)
    if min(Hq, Wq, Hr, Wr) < p:
        raise _lib.C2MError("feature maps smaller than the patch")
    Hqp, Wqp = (Hq - p) // si + 1, (Wq - p) // si + 1
    L = _lib.lib()
    with torch.cuda.device(fi.device):
        nbytes = L.c2m_feature_match_workspace_bytes_c(B, C, Hq, Wq, Hr, Wr)   # (sized by the maps' channels: ADVICE r4)
        ws = torch.empty(nbytes, dtype=torch.uint8, device=fi.device)
        idx = torch.empty((B, Hqp, Wqp), dtype=torch.int64, device=fi.device)
        val = torch.empty((B, Hqp, Wqp), dtype=torch.float32, device=fi.device)
        ss_i, ss_r = _pre_sumsq(feat_in, B, Hq * Wq), _pre_sumsq(feat_ref, B, Hr * Wr)
        with _apply_switch("filter"):
            _lib.check(L.c2m_feature_match_index_pre_f32(_stream(), fi.data_ptr(), fr.data_ptr(), B, C, Hq, Wq, Hr, Wr, p, si,
                                                         sr, int(bool(is_norm)), int(bool(norm_input)), int(bool(force_generic)),
                                                         idx.data_ptr(), val.data_ptr(), ws.data_ptr(), nbytes,
                                                         ss_i.data_ptr() if ss_i is not None else None,
                                                         ss_r.data_ptr() if ss_r is not None else None),
                       "c2m_feature_match_index_pre_f32")
        mfma = (not force_generic) and p == 3 and si == 1 and sr == 1 and C in (64, 128, 256)   # the C-ABI's own dispatch rule
        if return_skip or _corr_diag.enabled:
            # diagnostics only (bench.py's swept-row count, the dedup tests): the duplicate-row table exists only when the
            # MFMA kernel ran; nothing is kept otherwise, so the workspace dies with the call and no module state is
            # touched on the product path (DataParallel replicas run this function concurrently)
            table = _skip_table(ws, (B, Hq, Wq, Hr, Wr)) if mfma else None
            if _corr_diag.enabled:
                _corr_diag.table = table
                _corr_diag.filter = _filter_tables(ws, (B, Hq, Wq, Hr, Wr), Hqp * Wqp) if mfma else None
            if return_skip:
                if table is None:
                    raise _lib.C2MError("return_skip: the duplicate-row table exists only on the MFMA kernel's path "
                                        "(patch 3, strides 1, C in 64/128/256, force_generic=False)")
                return idx, val, table
    return idx, val


class _CorrDiag:
    """Opt-in diagnostics of the correlation launch: `with ops.record_corr_skip_table(): ...` keeps the duplicate-row table
    of the most recent MFMA launch inside the block (a small int32 tensor, not the workspace) and the pre-filter's
    candidate counts / flags."""
    enabled = False
    table = None
    filter = None


_corr_diag = _CorrDiag()


class record_corr_skip_table:
    def __enter__(self):
        _corr_diag.enabled, _corr_diag.table = True, None
        return self

    def __exit__(self, *exc):
        _corr_diag.enabled = False
        return False


def _skip_table(ws, shp):
    import ctypes
    off, nxt = ctypes.c_size_t(0), ctypes.c_int(0)
    _lib.check(_lib.lib().c2m_feature_match_skip_table(*shp, ctypes.byref(off), ctypes.byref(nxt)),
               "c2m_feature_match_skip_table")
    B = shp[0]
    return ws[off.value:off.value + 8 * B * nxt.value].view(torch.int32).view(B, nxt.value, 2).clone()


def _filter_tables(ws, shp, nq):
    import ctypes
    c, k, f, slots = ctypes.c_size_t(0), ctypes.c_size_t(0), ctypes.c_size_t(0), ctypes.c_int(0)
    _lib.check(_lib.lib().c2m_feature_match_filter_tables(*shp, ctypes.byref(c), ctypes.byref(k), ctypes.byref(f),
                                                          ctypes.byref(slots)), "c2m_feature_match_filter_tables")
    B, K = shp[0], slots.value
    return {"cnt": ws[c.value:c.value + 4 * B * nq].view(torch.int32).view(B, nq).clone(),
            "cand": ws[k.value:k.value + 4 * B * nq * K].view(torch.int32).view(B, nq, K).clone(),
            "flags": ws[f.value:f.value + 32].view(torch.int32).clone()}


def last_corr_filter_tables():
    """Pre-filter diagnostics recorded under `record_corr_skip_table()`: {'cnt': int32 [B, Nq] candidates per query (-1 =
    every ref patch re-scored), 'cand': int32 [B, Nq, slots], 'flags': int32 [8] ([0] != 0: the exact sweep produced the
    result)} -- meaningful only if the most recent launch took the pre-filter path (c2m_feature_match_filter_tables)."""
    if _corr_diag.filter is None:
        raise _lib.C2MError("no MFMA correlation launch was recorded (use `with ops.record_corr_skip_table():`)")
    return _corr_diag.filter


# The library's two A/B switches (c2m_feature_match_set_filter, c2m_conv3x3_set_head_stores) are thread_local: a launch sees the
# CALLING thread's setting.  The context managers below therefore also record the mode process-wide, and the op wrappers
# re-apply it on whichever thread launches (under nn.DataParallel the replicas launch from worker threads, where a
# `with ops.corr_filter_mode(0):` of the main thread used to have no effect -- an A/B pass would have measured the default path
# without noticing).  Nesting / concurrent use from several threads with DIFFERENT modes is not supported (last writer wins).
_switch_override = {"filter": None, "head": None}


class _apply_switch:
    """Re-apply a recorded process-wide override to this thread's C-ABI switch for the duration of one launch."""

    def __init__(self, which):
        self.which = which
        self.mode = _switch_override[which]

    def _set(self, mode):
        L = _lib.lib()
        fn = L.c2m_feature_match_set_filter if self.which == "filter" else L.c2m_conv3x3_set_head_stores
        _lib.check(fn(mode), "c2m set switch")

    def __enter__(self):
        if self.mode is not None:
            self._set(self.mode)
        return self

    def __exit__(self, *exc):
        if self.mode is not None:
            self._set(-1)
        return False


class corr_filter_mode:
    """`with ops.corr_filter_mode(0): ...` -- exact fp32 sweep only; (1): pre-filter + exact re-score (default); results are
    identical (c2m_feature_match_set_filter).  Applies to every thread's launches through this module while the block is open
    (measurement / tests)."""

    def __init__(self, mode):
        self.mode = int(mode)
        if self.mode not in (0, 1):
            raise _lib.C2MError("corr_filter_mode: 0 or 1")

    def __enter__(self):
        self.prev = _switch_override["filter"]
        _switch_override["filter"] = self.mode
        return self

    def __exit__(self, *exc):
        _switch_override["filter"] = self.prev
        return False


class head_store_mode:
    """`with ops.head_store_mode(0): ...` -- the DCN head epilogue's dword planar stores; (1): 16-byte stores through the quad
    transpose + flow window (default where W % 4 == 0); identical results (c2m_conv3x3_set_head_stores).  Applies to every
    thread's launches through this module while the block is open."""

    def __init__(self, mode):
        self.mode = int(mode)
        if self.mode not in (0, 1):
            raise _lib.C2MError("head_store_mode: 0 or 1")

    def __enter__(self):
        self.prev = _switch_override["head"]
        _switch_override["head"] = self.mode
        return self

    def __exit__(self, *exc):
        _switch_override["head"] = self.prev
        return False


def last_corr_skip_table():
    """Duplicate-row table recorded under `record_corr_skip_table()`: int32 [B, x_tiles, 2] = (from, to), ref rows
    [from, to) of that (sample, x-tile) were not swept (c2m_feature_match_skip_table).  Diagnostics / bench only."""
    if _corr_diag.table is None:
        raise _lib.C2MError("no MFMA correlation launch was recorded (use `with ops.record_corr_skip_table():`)")
    return _corr_diag.table


def build_pre_offsets(max_idx, h, w, scales=(1, 2, 4)):
    """max_idx int64 [B,h-2,w-2] -> tuple of pre-offset tensors [B,9,s*h,s*w,2] for s in scales (subset of 1,2,4)."""
    if not max_idx.is_cuda or max_idx.dtype != torch.int64:
        raise _lib.C2MError("max_idx must be an int64 GPU tensor")
    mi = max_idx.contiguous()
    B = mi.shape[0]
    if tuple(mi.shape[1:]) != (h - 2, w - 2):
        raise _lib.C2MError("max_idx must be [B, h-2, w-2]")
    if not set(scales) <= {1, 2, 4} or len(set(scales)) != len(tuple(scales)):
        raise _lib.C2MError(f"scales must be distinct members of (1, 2, 4), got {tuple(scales)}")
    outs = {s: torch.empty((B, 9, h * s, w * s, 2), dtype=torch.float32, device=mi.device) for s in scales}
    ptr = lambda s: outs[s].data_ptr() if s in outs else None  # noqa: E731
    with torch.cuda.device(mi.device):
        _lib.check(_lib.lib().c2m_build_pre_offsets_f32(_stream(), mi.data_ptr(), B, h, w, ptr(1), ptr(2), ptr(4)),
                   "c2m_build_pre_offsets_f32")
    return tuple(outs[s] for s in scales)


def _dcn_geom(inp, weight, stride, padding, dilation):
    B, C, H, W = inp.shape
    Co, Ck, kh, kw = weight.shape
    (sh, sw), (ph, pw), (dh, dw) = _pair(stride), _pair(padding), _pair(dilation)
    if Ck != C:
        raise _lib.C2MError(f"Input shape and kernel channels wont match: ({C} vs {Ck}).")
    Ho = (H + 2 * ph - (dh * (kh - 1) + 1)) // sh + 1
    Wo = (W + 2 * pw - (dw * (kw - 1) + 1)) // sw + 1
    return (B, C, H, W, Co, kh, kw, sh, sw, ph, pw, dh, dw), Ho, Wo


def dcn_v2_forward(inp, weight, bias, offset, mask, stride=1, padding=1, dilation=1, deformable_groups=1,
                   bf16_mma=False):
    """bf16_mma=True: the implicit GEMM runs on bf16 MFMA with fp32 accumulation (weights and blended samples rounded to
    bf16); tensors stay float32.  For callers that asked for reduced precision (bf16 autocast), not the default."""
    inp, weight, bias, offset, mask = (_dev_f32(t, n) for t, n in
                                       ((inp, "input"), (weight, "weight"), (bias, "bias"), (offset, "offset"), (mask, "mask")))
    g, Ho, Wo = _dcn_geom(inp, weight, stride, padding, dilation)
    B, C, H, W, Co, kh, kw = g[:7]
    dg = int(deformable_groups)
    if tuple(offset.shape) != (B, 2 * dg * kh * kw, Ho, Wo) or tuple(mask.shape) != (B, dg * kh * kw, Ho, Wo):
        raise _lib.C2MError("offset/mask shape does not match [B, 2*dg*kh*kw, Ho, Wo] / [B, dg*kh*kw, Ho, Wo]")
    L = _lib.lib()
    with torch.cuda.device(inp.device):
        nbytes = L.c2m_dcn_v2_forward_workspace_bytes(B, C, H, W, Co, kh, kw, dg)
        ws = torch.empty(max(nbytes, 1), dtype=torch.uint8, device=inp.device)
        out = torch.empty((B, Co, Ho, Wo), dtype=torch.float32, device=inp.device)
        fn = L.c2m_dcn_v2_forward_bf16mma_f32 if bf16_mma else L.c2m_dcn_v2_forward_f32
        _lib.check(fn(_stream(), inp.data_ptr(), weight.data_ptr(), bias.data_ptr(), offset.data_ptr(), mask.data_ptr(),
                      *g, dg, out.data_ptr(), ws.data_ptr(), nbytes), "c2m_dcn_v2_forward")
    return out


def dcn_v2_backward(inp, weight, bias, offset, mask, grad_output, stride=1, padding=1, dilation=1, deformable_groups=1,
                    need_input_grad=True):
    """-> (grad_input or None, grad_offset, grad_mask, grad_weight, grad_bias)."""
    inp, weight, bias, offset, mask, grad_output = (_dev_f32(t, n) for t, n in (
        (inp, "input"), (weight, "weight"), (bias, "bias"), (offset, "offset"), (mask, "mask"), (grad_output, "grad_output")))
    g, Ho, Wo = _dcn_geom(inp, weight, stride, padding, dilation)
    dg = int(deformable_groups)
    L = _lib.lib()
    with torch.cuda.device(inp.device):
        nbytes = L.c2m_dcn_v2_backward_workspace_bytes(*g, dg)
        ws = torch.empty(max(nbytes, 1), dtype=torch.uint8, device=inp.device)
        go, gm, gw, gb = (torch.empty_like(t) for t in (offset, mask, weight, bias))
        gi = torch.empty_like(inp) if need_input_grad else None
        _lib.check(L.c2m_dcn_v2_backward_f32(_stream(), inp.data_ptr(), weight.data_ptr(), bias.data_ptr(),
                                             offset.data_ptr(), mask.data_ptr(), grad_output.data_ptr(), *g, dg,
                                             gi.data_ptr() if gi is not None else None, go.data_ptr(), gm.data_ptr(),
                                             gw.data_ptr(), gb.data_ptr(),
                                             ws.data_ptr(), nbytes), "c2m_dcn_v2_backward_f32")
    return gi, go, gm, gw, gb


def dcn_fuse_offsets(conv_out, pre_offset, deformable_groups, kernel_taps, abs_sum=None):
    """conv_offset_mask output [B,3*dg*K,H,W] (+ pre_offset [B,K,H,W,2] or None) -> (offset [B,2*dg*K,H,W], mask
    [B,dg*K,H,W]) as dcn_v2.py:229-245 builds them; abs_sum (float64 GPU tensor of 256 slots, zeroed by the caller) accumulates sum|learned offset| across its slots."""
    conv_out = _dev_f32(conv_out, "conv_out")
    B, C3, H, W = conv_out.shape
    dg, K = int(deformable_groups), int(kernel_taps)
    if C3 != 3 * dg * K:
        raise _lib.C2MError("conv_out must have 3*dg*K channels")
    if pre_offset is not None:
        pre_offset = _dev_f32(pre_offset, "pre_offset")
        if tuple(pre_offset.shape) != (B, K, H, W, 2):
            raise _lib.C2MError("pre_offset must be [B, K, H, W, 2]")
    if abs_sum is not None and (abs_sum.dtype != torch.float64 or abs_sum.numel() < 256 or not abs_sum.is_cuda):
        raise _lib.C2MError("abs_sum must be a float64 GPU tensor with 256 slots (C2M_ABS_SUM_SLOTS)")
    offset = torch.empty((B, 2 * dg * K, H, W), dtype=torch.float32, device=conv_out.device)
    mask = torch.empty((B, dg * K, H, W), dtype=torch.float32, device=conv_out.device)
    with torch.cuda.device(conv_out.device):
        _lib.check(_lib.lib().c2m_dcn_fuse_offsets_f32(_stream(), conv_out.data_ptr(),
                                                       pre_offset.data_ptr() if pre_offset is not None else None,
                                                       B, dg, K, H, W, offset.data_ptr(), mask.data_ptr(),
                                                       abs_sum.data_ptr() if abs_sum is not None else None),
                   "c2m_dcn_fuse_offsets_f32")
    return offset, mask


# ---------------------------------------------------------------------------------------------------------------------
# 3x3 convolution, channels-last, fused epilogue (csrc/conv3x3.hip)
# ---------------------------------------------------------------------------------------------------------------------
ACT_NONE, ACT_RELU, ACT_LRELU = 0, 1, 2
class _ConvFlops:
    """Opt-in FLOP accounting of the conv3x3 launches (bench.py's roofline line): off on the product path -- nothing is
    counted and no module state is written unless a caller enabled it with count_conv_flops(True)."""
    enabled = False
    algorithmic = 0.0    # direct-convolution flops 2*Cout*9*Cin*H*W*B
    executed = 0.0       # flops the matrix instructions actually perform (F(2,3): 2/3, F(4,3): 1/2 of the direct count)
    by_algo = None       # {kernel family: [launches, algorithmic, executed]}

    @classmethod
    def add(cls, family, algo, execd):
        if cls.enabled:
            cls.algorithmic += algo
            cls.executed += execd
            e = cls.by_algo.setdefault(family, [0, 0.0, 0.0])
            e[0] += 1
            e[1] += algo
            e[2] += execd


def count_conv_flops(on=True):
    """Enable / disable the accounting and reset the counters."""
    _ConvFlops.enabled = bool(on)
    _ConvFlops.algorithmic = _ConvFlops.executed = 0.0
    _ConvFlops.by_algo = {}


def conv_flops_of_last_steps(reset=True, executed=False):
    """2*Cout*9*Cin*H*W*B summed over the conv3x3 calls since the last reset (needs count_conv_flops(True)); executed=True:
    the flops the matrix instructions really perform (Winograd F(2,3) launches execute 2/3 of the direct count)."""
    v = _ConvFlops.executed if executed else _ConvFlops.algorithmic
    if reset:
        _ConvFlops.algorithmic = _ConvFlops.executed = 0.0
        _ConvFlops.by_algo = {}
    return v


def conv_flops_by_family():
    return {k: list(v) for k, v in (_ConvFlops.by_algo or {}).items()}


class _WeightCache:
    """Re-laid-out conv weights, keyed by the parameter tensor and its version counter: inference re-uses them across
    calls; an optimiser step (in-place update -> new _version) or load_state_dict invalidates them.  An entry belongs to
    one tensor OBJECT (weak reference): a new parameter that happens to reuse a dead one's id, address and version never
    hits its entry, and entries die with their tensor.

    Writes through ``param.data`` (the reference's own init_offset / default_init_weights idiom, EMA updates, manual
    copies) do NOT bump ``_version``: ``refresh()`` re-runs the re-layout of the live entries from the tensors' current
    contents (a few microseconds of GPU time per entry, no host sync) and the fused inference entry points call it once
    per forward for the parameters of their module (refresh_weight_caches), so a cached image is never older than the
    forward that uses it; ``clear()`` drops everything (clear_weight_caches)."""

    def __init__(self):
        import threading
        self._d = {}
        self._tables = {}    # refresh job tables: frozenset(param ids) | None -> (signature, device int64 [n, 8], nblocks, any_f16)
        # nn.DataParallel runs the fused forwards of its replicas on one Python thread per GPU: every access to the two
        # dicts goes through this lock (re-entrant: refresh() -> redo() -> _relayout() stays on one thread)
        self._lock = threading.RLock()

    def _lookup(self, slot, key, weight):
        with self._lock:
            hit = self._d.get(slot)
        if hit is not None and hit[0] == key and hit[2]() is weight:
            return hit[1]
        return None

    def _store(self, slot, key, weight, value, redo=None):
        import weakref
        d, lock = self._d, self._lock

        def _drop(ref, slot=slot):
            with lock:
                cur = d.get(slot)
                if cur is not None and cur[2] is ref:
                    del d[slot]
        with lock:
            self._d[slot] = (key, value, weakref.ref(weight, _drop), redo)

    def clear(self):
        with self._lock:
            self._d.clear()
            self._tables.clear()

    _SPLIT_PIECES = {3: 3, 4: 1, 5: 3, 6: 2, 7: 0x42, 8: 0x22}     # cache kind -> pieces code of the split-kernel image (7 / 8: Winograd F(4,3) / F(2,3) along y on f16 x 2)

    def refresh(self, param_ids=None, kinds=None):
        with self._lock:
            return self._refresh_locked(param_ids, kinds)

    def _refresh_locked(self, param_ids=None, kinds=None):
        """Re-run the re-layout of the live entries (of the tensors whose id() is in param_ids, or all) from the tensors'
        current contents.  The split-kernel images -- nearly all of them -- go through ONE multi-tensor call
        (c2m_conv3x3_relayout_split_multi: a device job table, rebuilt only when the set of images changes); the rest
        (fp32-MFMA kernels' layouts, padded input channels) one call per image.  kinds: restrict to these cache kinds (an
        inference forward has no use for the autograd path's images, and vice versa)."""
        n = 0
        jobs, sig, dev = [], [], None
        for slot, (key, value, ref, redo) in list(self._d.items()):
            w = ref()
            if w is None or redo is None or (param_ids is not None and id(w) not in param_ids):
                continue
            if key[0] != w.data_ptr() or key[1] != w._version:
                continue        # stale by key: the next get() rebuilds it anyway
            rows, kind = (slot[1], slot[2]) if isinstance(slot, tuple) and len(slot) == 3 else (None, None)
            if kinds is not None and kind is not None and kind not in kinds:
                continue
            if kind in self._SPLIT_PIECES and len(key) == 5 and key[3] is None and w.is_contiguous() and (dev is None or dev == w.device):
                dev = w.device
                Co, Ci = (rows[1] - rows[0] if rows is not None else w.shape[0]), w.shape[1]
                cin_k, cout_k = (Co, Ci) if kind == 5 else (Ci, Co)      # the data-gradient conv swaps the roles
                pieces = self._SPLIT_PIECES[kind]
                nbytes = _lib.lib().c2m_conv3x3_relayout_split_bytes(cin_k, cout_k, pieces) - (256 if (pieces & 15) == 2 else 0)
                wptr = w.data_ptr() + (rows[0] * Ci * 9 * 4 if rows is not None else 0)
                jobs.append((wptr, value.data_ptr(), cin_k, cout_k, pieces | ((1 if cout_k <= 32 else 2) << 8) | ((1 if kind == 5 else 0) << 16),
                             nbytes // 2))
                sig.append((slot, wptr, value.data_ptr()))
            else:
                redo(w, value)
            n += 1
        if jobs:
            tkey = (None if param_ids is None else frozenset(param_ids), None if kinds is None else frozenset(kinds))
            sig = tuple(sig)
            tab = self._tables.get(tkey)
            if tab is None or tab[0] != sig:
                if torch.cuda.is_current_stream_capturing():
                    raise _lib.C2MError("weight-cache refresh: the set of cached weight images changed during a hipGraph capture (its "
                                        "job table would need a host-to-device copy); run one eager forward before capturing")
                rowsl, first = [], 0
                for (wptr, iptr, ci, co, flags, elems) in jobs:
                    rowsl.append([wptr, iptr, ci, co, flags, elems, first, 0])
                    first += (elems + 255) // 256
                host = torch.tensor(rowsl, dtype=torch.int64).pin_memory()
                tab = (sig, host.to(dev, non_blocking=True), first, int(any((j[4] & 0xf) == 2 for j in jobs)), host)
                if len(self._tables) >= 64:      # (tables of parameter sets that no longer exist: a handful of KiB each)
                    self._tables.clear()
                self._tables[tkey] = tab
            with torch.cuda.device(dev):
                _lib.check(_lib.lib().c2m_conv3x3_relayout_split_multi(_stream(), tab[1].data_ptr(), len(jobs), tab[2], tab[3]),
                           "c2m_conv3x3_relayout_split_multi")
        return n

    @staticmethod
    def _relayout(weight, rows, pad_cin_to, wino, wr=None):
        w = weight.detach()
        if rows is not None:
            w = w[rows[0]:rows[1]]
        if w.dtype != torch.float32 or not w.is_cuda or w.dim() != 4 or tuple(w.shape[2:]) != (3, 3):
            raise _lib.C2MError("conv3x3: weight must be a float32 GPU tensor [Cout, Cin, 3, 3]")
        Co, Ci = w.shape[:2]
        if pad_cin_to is not None and Ci < pad_cin_to:
            w = torch.nn.functional.pad(w, (0, 0, 0, 0, 0, pad_cin_to - Ci))
            Ci = pad_cin_to
        w = w.contiguous()
        L = _lib.lib()
        wino = int(wino)
        if wino in (3, 4, 5, 6, 7, 8):   # split images (3 exact bf16 pieces / 1 rounded piece per weight; 6: scaled f16 x 2; 7 / 8: its Winograd-along-y images); 5: of the data-gradient conv
            pieces = 1 if wino == 4 else 2 if wino == 6 else 0x42 if wino == 7 else 0x22 if wino == 8 else 3
            nbytes = L.c2m_conv3x3_relayout_split_bytes(Co, Ci, pieces) if wino == 5 else L.c2m_conv3x3_relayout_split_bytes(Ci, Co, pieces)
        else:
            nbytes = (L.c2m_conv3x3_relayout_bytes, L.c2m_conv3x3_relayout_wino_bytes, L.c2m_conv3x3_relayout_wino4_bytes)[wino](Ci, Co)
        if nbytes == 0:
            raise _lib.C2MError(f"conv3x3: unsupported channel counts Cin={Ci}, Cout={Co}" + (" for this kernel" if wino else
                                " (input channels must be a multiple of 32)"))
        if wr is None:
            wr = torch.empty((nbytes + 3) // 4, dtype=torch.float32, device=w.device)
        with torch.cuda.device(w.device):
            if wino == 5:
                _lib.check(L.c2m_conv3x3_relayout_split_dgrad_f32(_stream(), w.data_ptr(), Ci, Co, pieces, wr.data_ptr()),
                           "c2m_conv3x3_relayout_split_dgrad_f32")
            elif wino in (3, 4, 6, 7, 8):
                _lib.check(L.c2m_conv3x3_relayout_split_f32(_stream(), w.data_ptr(), Ci, Co, pieces, wr.data_ptr()),
                           "c2m_conv3x3_relayout_split_f32")
            else:
                fn = (L.c2m_conv3x3_relayout_f32, L.c2m_conv3x3_relayout_wino_f32, L.c2m_conv3x3_relayout_wino4_f32)[wino]
                _lib.check(fn(_stream(), w.data_ptr(), Ci, Co, wr.data_ptr()), "c2m_conv3x3_relayout")
        return wr

    def get(self, weight, pad_cin_to=None, rows=None, wino=0):   # wino: 0 direct, 1 F(2,3), 2 F(4,3), 3 split-bf16x3, 4 bf16, 5 dgrad, 6 split-f16x2
        key = (weight.data_ptr(), weight._version, tuple(weight.shape), pad_cin_to, weight.device.index)
        slot = (id(weight), rows, wino)
        hit = self._lookup(slot, key, weight)
        if hit is not None:
            return hit
        wr = self._relayout(weight, rows, pad_cin_to, wino)
        self._store(slot, key, weight, wr,
                    redo=lambda w, buf, rows=rows, pad=pad_cin_to, wino=wino: self._relayout(w, rows, pad, wino, buf))
        return wr


_wcache = _WeightCache()


def _nhwc_src(t, name, bf16_ok=False):
    """channels_last float32 GPU tensor (logical [B,C,H,W], channel stride 1) or a channel slice of one -> ConvSrc.
    bf16_ok: bfloat16 tensors pass too (pitches are in elements either way; c2m_conv3x3_desc.io_flags names the types)."""
    if not t.is_cuda or t.dim() != 4 or not (t.dtype == torch.float32 or (bf16_ok and t.dtype == torch.bfloat16)):
        raise _lib.C2MError(f"{name} must be a 4-D float32 GPU tensor" + (" (or bfloat16)" if bf16_ok else ""))
    sb, sc, sh, sw = t.stride()
    if sc != 1:
        raise _lib.C2MError(f"{name} must be channels-last (stride 1 along C); got strides {t.stride()}")
    return _lib.ConvSrc(t.data_ptr(), t.shape[1], sw, sh, sb)


def empty_nhwc(B, C, H, W, device, dtype=torch.float32):
    return torch.empty((B, C, H, W), dtype=dtype, device=device, memory_format=torch.channels_last)


import os as _os

_WINO = _os.environ.get("C2M_CONV_WINO", "1") != "0"
_WINO4 = _os.environ.get("C2M_CONV_WINO4", "1") != "0"
# C2M_CONV_SPLIT: "all" (default) -- every convolution the split-bf16 kernel supports runs on it (csrc/conv3x3_split.hip:
# fp32-accurate, 6 bf16 MFMAs per fp32 product sum), the extractor towers that feed the index search included (measured on
# configs[2]: 1 near-tie flip of 24 964 queries against the CPU chain, 2 with the fp32-MFMA kernels); "1" -- only calls
# with fast=True (decoder, DCN heads, VGG taps of the Ref); "0" -- never (fp32-MFMA direct / Winograd kernels only)
_SPLIT = _os.environ.get("C2M_CONV_SPLIT", "all")
# C2M_CONV_SPLIT16: "1" (default) -- where the split kernel is chosen automatically, inference runs its f16 x 2 flavour
# (three f16 products per fp32 product sum instead of six bf16 ones; error of the class of an fp32 accumulation chain,
# domain |x| < 65520: include/c2m_hip.h C2M_CONV_SPLIT_F16X2); "0" -- the bf16 x 3 flavour (full fp32 range) everywhere.
# The autograd path (conv3x3_autograd: gradients can be tiny) always runs bf16 x 3.
_SPLIT16 = _os.environ.get("C2M_CONV_SPLIT16", "1") != "0"
# C2M_CONV_WINO16: "0" (default) off; "43" / "23" -- where the f16 x 2 flavour would run a channels-last layer with Cout % 64 == 0 it
# runs behind a Winograd F(4,3) / F(2,3) transform ALONG Y (csrc/conv3x3_wino16.hip: 1/2 / 2/3 of the matrix instructions).  Measured
# (round 5): F(4,3) is 6 % faster than the direct f16 x 2 kernel on 64 -> 64 @640^2 and 2-2.5x further from float64
# than the exact-fp32 kernel on 64-channel layers; F(2,3) is more accurate than the direct kernel and 12 % slower -- the family is
# bound by vector-memory traffic, not by the matrix pipe.  Not the default; algo="wino16" / "wino16_f23" select it per call.
# Round 6: the kernel lives in csrc/experimental/ and is only in a library built with `make EXPERIMENTAL=1` (experimental_built()).
_WINO16 = {"43": 7, "23": 8}.get(_os.environ.get("C2M_CONV_WINO16", "0"), 0)


def experimental_built():
    """True if libc2m_hip.so was built with `make EXPERIMENTAL=1` (csrc/experimental/: the Winograd-along-y and loader /
    matrix-wave convolution kernels, both measured no-gos kept for their numbers -- DESIGN.md 6.7, 6.10).  Probed once by a
    one-tile launch: the product library answers algo "wino16" with C2M_ERR_UNSUPPORTED."""
    v = getattr(experimental_built, "_v", None)
    if v is None:
        x = torch.zeros((1, 16, 4, 32), device="cuda").contiguous(memory_format=torch.channels_last)
        try:
            conv3x3(x, torch.zeros((64, 16, 3, 3), device="cuda"), algo="wino16_f23")
            v = True
        except _lib.C2MError:
            v = False
        experimental_built._v = v
    return v
# C2M_DCN_F16X2: "1" (default) -- the DCNv2 forward's implicit GEMM follows the convolutions onto the f16 x 2 arithmetic
_DCN_F16X2 = _os.environ.get("C2M_DCN_F16X2", "1") != "0"
# internal kernel ids (= weight-cache kinds; 5 is the data-gradient image of the bf16 x 3 kernel) -> c2m_conv3x3_desc.algo
ALGO_IDS = {"direct": 0, "winograd": 1, "winograd4": 2, "split": 3, "bf16": 4, "split16": 6, "wino16": 7, "wino16_f23": 8}
_DESC_ALGO = (0, 1, 2, 3, 4, 3, 5, 6, 7)
_FAMILY = ("direct", "winograd_f23", "winograd_f43", "split_bf16x3", "bf16", "split_bf16x3", "split_f16x2", "wino16_f43y", "wino16_f23y")
# matrix flops actually executed per algorithmic (direct-convolution) flop, and the pipe they run on (Winograd along y on
# f16 x 2: three products x 6/12 (4/6) of the taps x 32/30 of the columns -- two MFMA columns of a tile are halo only)
_EXEC_FACTOR = (1.0, 2.0 / 3.0, 0.5, 6.0, 1.0, 6.0, 3.0, 3.0 * 0.5 * 32.0 / 30.0, 3.0 * (2.0 / 3.0) * 32.0 / 30.0)


# ---- f16 x 2 is the DEFAULT arithmetic of fp32 inference, and its domain is |activation| < 65520: outside it the kernel
# writes NaN where nn.Conv2d (arch_util.py:80-136) returns a number.  Every f16 x 2 launch therefore reports out-of-range
# inputs into a per-device flag (c2m_conv3x3_desc.range_flag), and the fused module forwards run under `f16_range_guard`: the
# flag is zeroed before, read once after (one 4-byte read-back per module forward), and if it is set the forward is repeated
# on the bf16 x 3 flavour (full fp32 exponent range) -- the module is then pinned to that flavour.
import threading as _threading

_tls = _threading.local()
_range_flags = {}
_range_lock = _threading.Lock()


def _device_flag(dev):
    """The per-device int32 flag (explicit f16 x 2 calls outside any guard report into it; range_flag_set() reads it)."""
    key = (dev.type, dev.index if dev.index is not None else torch.cuda.current_device())
    t = _range_flags.get(key)
    if t is None:
        with _range_lock:
            t = _range_flags.get(key)
            if t is None:
                t = _range_flags[key] = torch.zeros(1, dtype=torch.int32, device=dev)
    return t


def _range_flag(dev):
    """The int32 device flag f16 x 2 launches report into: the enclosing f16_range_guard's OWN flag (one per guard invocation,
    so that concurrent guards on other threads / streams of the device cannot zero each other's report), else -- explicit
    algo="split16" / "f16x2" calls outside any guard -- one per device, which the caller may read with range_flag_set()."""
    g = getattr(_tls, "guard_flag", None)
    if g is not None and g.device == (dev if dev.index is not None else torch.device(dev.type, torch.cuda.current_device())):
        return g
    return _device_flag(dev)


def _split16_now():
    """f16 x 2 (True) or bf16 x 3 (False) for an fp32 split-kernel call of this thread right now."""
    ov = getattr(_tls, "flavour", None)
    return _SPLIT16 if ov is None else ov


def _f16x2_auto():
    """What algo=None resolves to in conv3x3 / conv3x3_dcn_head / dcn_v2_forward_nhwc: the f16 x 2 arithmetic only where
    somebody reads the range flag afterwards -- inside an f16_range_guard (the fused module forwards) -- or where the caller
    asked for it by name (`with ops.conv_flavour("f16x2")`, then the flag is the caller's to read: range_flag_set()).  A bare
    direct call gets the full-range arithmetic (bf16 x 3 convolutions, fp32-MFMA DCNv2): never a silent NaN."""
    ov = getattr(_tls, "flavour", None)
    if ov is not None:
        return ov
    if getattr(_tls, "guarded", False):
        return _SPLIT16
    if _SPLIT16 and not _f16x2_auto.warned and not torch.cuda.is_current_stream_capturing():
        #  say it once: $C2M_CONV_SPLIT16=1 does NOT make a bare call f16 x 2 -- about twice the matrix work
        _f16x2_auto.warned = True
        import warnings
        warnings.warn("c2m_amd.ops: a conv3x3 / conv3x3_dcn_head / dcn_v2_forward_nhwc call outside `f16_range_guard` and without "
                      "`ops.conv_flavour(...)` runs the FULL-RANGE arithmetic (bf16 x 3 convolutions, fp32-MFMA DCNv2: ~2x the matrix "
                      "work of the f16 x 2 default of the fused module forwards).  For f16 x 2 wrap the calls in "
                      "`with ops.conv_flavour('f16x2'):` and poll `ops.range_flag_set(device)` (domain |x| < 65520); "
                      "`with ops.conv_flavour('bf16x3'):` silences this.", RuntimeWarning, stacklevel=3)
    return False


_f16x2_auto.warned = False


def range_flag_set(device, clear=True):
    """True if an f16 x 2 launch OUTSIDE any guard reported an out-of-domain activation on `device` (a torch.device, a string
    or an index) since the last clear (one 4-byte read-back).  Always the per-device flag: called from code that runs under an
    f16_range_guard it neither sees nor zeroes the guard's own per-invocation flag (ADVICE r5 -- clearing that one would make
    the guard return f16 x 2 overflow garbage instead of re-running on bf16 x 3)."""
    f = _device_flag(torch.device("cuda", device) if isinstance(device, int) else torch.device(device))
    v = int(f.item()) != 0
    if clear and v:
        f.zero_()
    return v


class conv_flavour:
    """`with ops.conv_flavour("bf16x3"): ...` / ("f16x2"): the fp32 split-kernel flavour of this thread's conv3x3 calls inside
    the block (overrides $C2M_CONV_SPLIT16)."""

    def __init__(self, name):
        if name not in ("f16x2", "bf16x3"):
            raise _lib.C2MError("conv_flavour: 'f16x2' or 'bf16x3'")
        self.val = name == "f16x2"

    def __enter__(self):
        self.prev = getattr(_tls, "flavour", None)
        _tls.flavour = self.val
        return self

    def __exit__(self, *exc):
        _tls.flavour = self.prev
        return False


def f16_range_guard(owner, fn, device):
    """Run `fn()` (a fused forward made of conv3x3 calls) so that it never returns f16 x 2 overflow garbage: see above.
    `owner` (an nn.Module or any object) remembers the pinned flavour in `owner._c2m_conv_bf16x3`."""
    if getattr(_tls, "guarded", False):      # an enclosing guard (a parent module's forward) checks the flag for all of us
        return fn()
    # (under bf16 autocast the convolutions run the bf16 flavour and never touch the flag; the DCNv2 forwards -- fp32 under
    # autocast, as the reference's custom_fwd(cast_inputs=float32) makes them -- still run f16 x 2 and do)
    if getattr(owner, "_c2m_conv_bf16x3", False) or not _split16_now():
        if getattr(owner, "_c2m_conv_bf16x3", False):
            with conv_flavour("bf16x3"):
                return fn()
        return fn()
    if torch.cuda.is_current_stream_capturing():
        # the range check reads a flag back, which a hipGraph capture cannot do: captured forwards run the full-range flavour
        with conv_flavour("bf16x3"):
            return fn()
    flag = torch.zeros(1, dtype=torch.int32, device=device)   # this invocation's own flag (see _range_flag)
    _tls.guarded, _tls.guard_flag = True, flag
    try:
        out = fn()
    finally:
        _tls.guarded, _tls.guard_flag = False, None
    if int(flag.item()) == 0:
        return out
    import warnings
    warnings.warn(f"{type(owner).__name__}: an activation left the f16 x 2 convolution flavour's domain (|x| >= 65520); the forward "
                  "was recomputed on the bf16 x 3 flavour (full fp32 range), which this module now keeps", RuntimeWarning)
    try:
        owner._c2m_conv_bf16x3 = True
    except Exception:  # noqa: BLE001 -- an owner without attributes: just recompute
        pass
    with conv_flavour("bf16x3"):
        return fn()


def _wino_ok(srcs, weight, out_mode, W):
    """Winograd F(2,3)-along-x kernel: channels-last output, 64-wide cout tiles, whole 32-pixel tiles along x."""
    return (_WINO and out_mode in ("nhwc", "nhwc_pool2") and weight.shape[0] % 64 == 0 and W % 32 == 0 and
            all(s.shape[1] % 16 == 0 for s in srcs) and sum(s.shape[1] for s in srcs) == weight.shape[1])


def _wino4_ok(srcs, weight, out_mode, W):
    """Winograd F(4,3)-along-x kernel: as F(2,3) but plain channels-last output and whole 64-pixel tiles along x."""
    return _WINO4 and W % 64 == 0 and out_mode == "nhwc" and _wino_ok(srcs, weight, out_mode, W)


def bf16_autocast():
    """True inside `torch.autocast('cuda', dtype=torch.bfloat16)`: the caller asked for reduced precision (BASELINE
    configs[4]) -- the convolutions then run the split kernel with ONE round-to-nearest bf16 piece per operand (algo
    "bf16": bf16 products, fp32 accumulation, fp32 tensors in and out), 1/6 of the matrix work of the fp32 flavour."""
    return torch.is_autocast_enabled('cuda') and torch.get_autocast_dtype('cuda') == torch.bfloat16


def _split_ok(srcs, weight, fast):
    """Split-bf16 kernel: any map size / output mode; 16-channel chunks; sources add up to the weight's input channels."""
    return ((_SPLIT == "all" or (fast and _SPLIT != "0")) and all(s.shape[1] % 16 == 0 for s in srcs) and
            sum(s.shape[1] for s in srcs) == weight.shape[1])


def conv3x3(srcs, weight, bias=None, act=ACT_NONE, slope=0.1, res1=None, res2=None, out_mode="nhwc", out=None, algo=None,
            out2_grouped8=None, fast=False, out_dtype=None, dword_stores=False):
    """out = act(conv3x3(cat(srcs, dim=1)) + bias) + res1 + res2 on channels-last tensors, one kernel.

    bf16 tensors ("bf16" kernel and "nhwc" mode only -- what a bf16-autocast forward keeps its activations in between fused
    convolutions, BASELINE configs[4]): a single bfloat16 source, bfloat16 residuals and `out_dtype=torch.bfloat16` (or a
    bfloat16 `out`) are accepted in any combination with float32 ones; sums stay fp32, one rounding at the store.

    srcs: one or two channels_last tensors [B,Ci,H,W] (each Ci % 32 == 0; a single source with fewer input channels than
    the (zero-padded) weight is not accepted -- pad the tensor).  out_mode: "nhwc" -> channels_last [B,Cout,H,W];
    "pixel_shuffle" -> channels_last [B,Cout/4,2H,2W] (= PixelShuffle(2) of the conv output); "nchw" -> contiguous;
    "nhwc_pool2" -> channels_last [B,Cout,H/2,W/2] = MaxPool2d(2, 2) of the activated output (Winograd F(2,3) shapes only).
    out2_grouped8: a zero-bordered group-major buffer [B,Cout/8,H+3,W+3,8] that receives a second copy of the output
    (what the DCNv2 kernel gathers 8-channel groups from); "nhwc" mode on the direct kernel only.
    algo: None (auto), "direct", "winograd", "winograd4", "split" (fp32-accurate on the bf16 matrix pipe: three exact bf16
    pieces per operand, six MFMAs per product sum, csrc/conv3x3_split.hip), "split16" (fp32-accurate on the f16 matrix pipe:
    two round-to-nearest f16 pieces per activation, per-tensor-scaled weights, THREE MFMAs per product sum; |x| < 65520),
    "bf16" (one rounded piece: a bf16 convolution with fp32 accumulation).  Auto: the split kernel wherever it applies (any
    map size / mode, channels % 16 == 0) -- its f16 x 2 flavour inside an f16_range_guard (the fused module forwards) or under
    `ops.conv_flavour("f16x2")`, unless $C2M_CONV_SPLIT16=0; a bare direct call gets bf16 x 3 (full range: _f16x2_auto).
    dword_stores: split kernels, "pixel_shuffle" / "nchw" modes: this call's epilogue stores one dword per lane instead of
    the 16-byte lane-swapped / quad-transposed pieces (c2m_hip.h C2M_IO_DWORD_STORES: same bits, the other instruction
    sequence -- tests and measurement; per call, nothing process-wide).
    $C2M_CONV_SPLIT=1 restricts it to calls with fast=True (the decoder; the extractor towers that feed the index search
    then stay on the fp32-MFMA kernels), $C2M_CONV_SPLIT=0 restores the round-2 choice everywhere: Winograd F(4,3) with
    fast=True / F(2,3) where the shapes allow, else direct."""
    srcs = list(srcs) if isinstance(srcs, (list, tuple)) else [srcs]
    B, _, H, W = srcs[0].shape
    Cin = sum(s.shape[1] for s in srcs)
    Cout = weight.shape[0]
    dev = srcs[0].device
    if algo is None:
        reduced = bf16_autocast()
        if (out2_grouped8 is None and (_split_ok(srcs, weight, fast) or (reduced and _split_ok(srcs, weight, True))) and (out_mode != "nhwc_pool2" or (H % 2 == 0 and W % 2 == 0))
                and (out_mode not in ("nhwc", "nhwc_pool2") or Cout % 4 == 0)):   # channels-last stores are 16-byte vectors
            wino = 4 if reduced else 6 if _f16x2_auto() else 3
            if (wino == 6 and _WINO16 and out_mode == "nhwc" and Cout % 64 == 0 and out is None and out_dtype in (None, torch.float32)
                    and all(t_ is None or t_.dtype == torch.float32 for t_ in (srcs[0], res1, res2))):
                wino = _WINO16
        else:
            wino = 2 if (fast and _wino4_ok(srcs, weight, out_mode, W)) else 1 if _wino_ok(srcs, weight, out_mode, W) else 0
    else:
        wino = ALGO_IDS[algo]
    if out2_grouped8 is not None:
        if algo not in (None, "direct") or out_mode != "nhwc":
            raise _lib.C2MError("out2_grouped8 needs the direct kernel in nhwc mode")
        wino = 0
    wr = _wcache.get(weight, pad_cin_to=Cin if weight.shape[1] < Cin else None, wino=wino)
    d = _lib.Conv3x3Desc()
    d.algo = _DESC_ALGO[wino]
    d.B, d.H, d.W, d.Cin, d.Cout, d.nsrc = B, H, W, Cin, Cout, len(srcs)
    for k, s in enumerate(srcs):
        if tuple(s.shape[2:]) != (H, W) or s.shape[0] != B:
            raise _lib.C2MError("conv3x3: sources must share B, H, W")
        d.src[k] = _nhwc_src(s, f"src{k}", bf16_ok=True)
    io = 1 if srcs[0].dtype == torch.bfloat16 else 0
    if any(s.dtype == torch.bfloat16 for s in srcs[1:]) or (io and len(srcs) != 1):
        raise _lib.C2MError("conv3x3: a bfloat16 source must be the only source")
    if out is not None:
        out_dtype = out.dtype
    if out_dtype == torch.bfloat16:
        io |= 2
    elif out_dtype not in (None, torch.float32):
        raise _lib.C2MError("conv3x3: out_dtype is float32 or bfloat16")
    for bit, r in ((4, res1), (8, res2)):
        if r is not None and r.dtype == torch.bfloat16:
            io |= bit
    if io and (wino != 4 or out_mode != "nhwc"):
        raise _lib.C2MError("conv3x3: bfloat16 tensors need the bf16 kernel (algo='bf16' / bf16 autocast) in nhwc mode")
    if dword_stores:
        if io or wino not in (3, 4, 6) or out_mode not in ("pixel_shuffle", "nchw"):
            raise _lib.C2MError("conv3x3: dword_stores selects a store path of the split kernels' pixel_shuffle / nchw epilogues")
        io = 16
    d.io_flags = io
    d.wr = wr.data_ptr()
    if bias is not None:
        bias = _dev_f32(bias.detach(), "bias")
    d.bias = bias.data_ptr() if bias is not None else None
    d.act, d.slope = int(act), float(slope)
    if out_mode == "nhwc":
        if out is None:
            out = empty_nhwc(B, Cout, H, W, dev, torch.bfloat16 if io & 2 else torch.float32)
        d.out_mode = 0
        o = _nhwc_src(out, "out", bf16_ok=True)
        d.out_pix_pitch, d.out_row_pitch, d.out_img_pitch = o.pix_pitch, o.row_pitch, o.img_pitch
        for name, r in (("res1", res1), ("res2", res2)):
            if r is not None:
                rs = _nhwc_src(r, name, bf16_ok=True)
                if (rs.pix_pitch, rs.row_pitch, rs.img_pitch) != (o.pix_pitch, o.row_pitch, o.img_pitch) or r.shape != out.shape:
                    raise _lib.C2MError(f"{name} must have the geometry of the output")
                setattr(d, name, r.data_ptr())
        if out2_grouped8 is not None:
            d.out2, d.out2_row_pitch, d.out2_plane_pitch, d.out2_img_pitch = _grouped8_args(out2_grouped8, B, Cout, H, W, dev)
    elif out_mode == "pixel_shuffle":
        out = empty_nhwc(B, Cout // 4, 2 * H, 2 * W, dev)
        d.out_mode = 1
        o = _nhwc_src(out, "out")
        d.out_pix_pitch, d.out_row_pitch, d.out_img_pitch = o.pix_pitch, o.row_pitch, o.img_pitch
    elif out_mode == "nhwc_pool2":
        if wino not in (1, 3, 4, 6) or H % 2 != 0 or W % 2 != 0:
            raise _lib.C2MError("conv3x3: the pooled epilogue needs the Winograd F(2,3) or the split-bf16 kernel and even H, W")
        out = empty_nhwc(B, Cout, H // 2, W // 2, dev)
        d.out_mode = 4
        o = _nhwc_src(out, "out")
        d.out_pix_pitch, d.out_row_pitch, d.out_img_pitch = o.pix_pitch, o.row_pitch, o.img_pitch
    elif out_mode == "nchw":
        out = torch.empty((B, Cout, H, W), dtype=torch.float32, device=dev)
        d.out_mode = 2
    else:
        raise _lib.C2MError(f"unknown out_mode {out_mode}")
    d.out = out.data_ptr()
    if wino in (6, 7, 8):
        d.range_flag = _range_flag(dev).data_ptr()
    with torch.cuda.device(dev):
        _lib.check(_lib.lib().c2m_conv3x3_nhwc_f32(_stream(), d), "c2m_conv3x3_nhwc_f32")
    if _ConvFlops.enabled:
        _ConvFlops.add(_FAMILY[wino], 2.0 * Cout * 9 * Cin * H * W * B, 2.0 * Cout * 9 * Cin * H * W * B * _EXEC_FACTOR[wino])
    return out


# C2M_RESBLOCK: "0" (default) -- a ResidualBlockNoBN (arch_util.py:80-136) is two conv3x3 launches; "1" -- the fused inference
# bodies run it as ONE launch (csrc/experimental/conv3x3_resblock.hip: the intermediate tensor stays in LDS) on maps of at least
# C2M_RESBLOCK_MINPIX pixels, IF the library was built with `make EXPERIMENTAL=1`.  Round 6 built and measured it:
# right on its first run on hardware, equal to the two-launch path to 1 ulp -- and 6 - 7 % SLOWER at 640^2 / 320^2, 25 % at 160^2
# (on this chip a wave's vector-ALU instructions are paid for in matrix-pipe time whether or not another wave
# runs beside it, and the fused form needs 11 % more MFMAs plus more vector work per MFMA).  A recorded no-go, kept for its numbers.
_RESBLOCK = _os.environ.get("C2M_RESBLOCK", "0")
_RESBLOCK_MINPIX = int(_os.environ.get("C2M_RESBLOCK_MINPIX", str(300 * 300)))


def resblock3x3_ok(x, w1, w2, res2=None):
    """True if `resblock3x3` takes these tensors: fp32 channels-last GPU activations, two [64, 64, 3, 3] weights."""
    return (x.is_cuda and x.dtype == torch.float32 and x.dim() == 4 and x.shape[1] == 64 and x.stride(1) == 1 and
            tuple(w1.shape) == (64, 64, 3, 3) and tuple(w2.shape) == (64, 64, 3, 3) and
            (res2 is None or (res2.dtype == torch.float32 and res2.shape == x.shape and res2.stride() == x.stride())))


def resblock3x3_wanted(x):
    """The policy of the fused bodies (ref_restoration_arch._fused_body): f16 x 2 active for this thread's auto calls, map large enough."""
    if _RESBLOCK != "1" or _SPLIT == "0" or not _f16x2_auto():
        return False
    return x.shape[2] * x.shape[3] >= _RESBLOCK_MINPIX and bool(_lib.lib().c2m_resblock3x3_supported(64, x.shape[2], x.shape[3]))


def resblock3x3(x, w1, b1, w2, b2, res2=None, out=None):
    """out = x + conv2(relu(conv1(x) + b1)) + b2 (+ res2): a ResidualBlockNoBN (arch_util.py:128-136, res_scale 1) in ONE launch
    on the f16 x 2 arithmetic (c2m_resblock3x3_nhwc_f32: the intermediate tensor stays in LDS, x is read once, the identity is
    rebuilt from x's two f16 pieces -- equal to two conv3x3(algo="split16") launches up to that rounding, |d| <= 2^-22 |x|).
    Needs a library built with `make EXPERIMENTAL=1` (C2MError "unsupported" otherwise): a measured no-go, see _RESBLOCK above.
    x: channels_last [B,64,H,W] fp32; domain as algo="split16" (range flag of the enclosing guard / the device)."""
    if not resblock3x3_ok(x, w1, w2, res2):
        raise _lib.C2MError("resblock3x3: channels_last float32 GPU tensor [B,64,H,W], weights [64,64,3,3]")
    B, _, H, W = x.shape
    dev = x.device
    wr1, wr2 = _wcache.get(w1, wino=6), _wcache.get(w2, wino=6)
    if out is None:
        out = empty_nhwc(B, 64, H, W, dev)
    if out.data_ptr() == x.data_ptr():
        raise _lib.C2MError("resblock3x3: out may not alias x")
    xs, o = _nhwc_src(x, "x"), _nhwc_src(out, "out")
    d = _lib.ResBlockDesc()
    d.B, d.H, d.W, d.C = B, H, W, 64
    d.x, d.x_pix_pitch, d.x_row_pitch, d.x_img_pitch = xs.ptr, xs.pix_pitch, xs.row_pitch, xs.img_pitch
    d.out, d.out_pix_pitch, d.out_row_pitch, d.out_img_pitch = out.data_ptr(), o.pix_pitch, o.row_pitch, o.img_pitch
    if res2 is not None:
        r = _nhwc_src(res2, "res2")
        if (r.pix_pitch, r.row_pitch, r.img_pitch) != (o.pix_pitch, o.row_pitch, o.img_pitch):
            raise _lib.C2MError("resblock3x3: res2 must have the geometry of the output")
        d.res2 = res2.data_ptr()
    d.wr1, d.wr2 = wr1.data_ptr(), wr2.data_ptr()
    b1 = _dev_f32(b1.detach(), "bias1") if b1 is not None else None
    b2 = _dev_f32(b2.detach(), "bias2") if b2 is not None else None
    d.bias1 = b1.data_ptr() if b1 is not None else None
    d.bias2 = b2.data_ptr() if b2 is not None else None
    d.range_flag = _range_flag(dev).data_ptr()
    with torch.cuda.device(dev):
        _lib.check(_lib.lib().c2m_resblock3x3_nhwc_f32(_stream(), d), "c2m_resblock3x3_nhwc_f32")
    if _ConvFlops.enabled:
        alg = 2 * 2.0 * 64 * 9 * 64 * H * W * B
        steps = B * ((W + 29) // 30) * ((H + 2 + 7) // 8)          # strips x steps of 8 rows x 32 MFMA columns, both convolutions
        _ConvFlops.add("resblock_split_f16x2", alg, steps * 2 * 3.0 * 2.0 * 64 * 9 * 64 * 256)
    return out


def conv3x3_dgrad(grad_out, weight):
    """Data gradient of conv3x3: grad_out channels-last [B,Cout,H,W], weight [Cout,Cin,3,3] -> dX channels-last [B,Cin,H,W]
    = conv3x3(grad_out, W') with W'[ci][co][dy][dx] = W[co][ci][2-dy][2-dx], on the split-bf16 kernel (fp32-accurate)."""
    Cout, Cin = weight.shape[:2]
    B, Cg, H, W = grad_out.shape
    if Cg != Cout or Cout % 16 != 0 or Cin % 4 != 0:
        raise _lib.C2MError("conv3x3_dgrad: grad_out channels must equal weight.shape[0] (a multiple of 16); Cin % 4 == 0")
    wr = _wcache.get(weight, wino=5)
    out = empty_nhwc(B, Cin, H, W, grad_out.device)
    d = _lib.Conv3x3Desc()
    d.algo = 3
    d.B, d.H, d.W, d.Cin, d.Cout, d.nsrc = B, H, W, Cout, Cin, 1
    d.src[0] = _nhwc_src(grad_out, "grad_out")
    d.wr = wr.data_ptr()
    d.out_mode = 0
    o = _nhwc_src(out, "out")
    d.out, d.out_pix_pitch, d.out_row_pitch, d.out_img_pitch = out.data_ptr(), o.pix_pitch, o.row_pitch, o.img_pitch
    with torch.cuda.device(grad_out.device):
        _lib.check(_lib.lib().c2m_conv3x3_nhwc_f32(_stream(), d), "c2m_conv3x3_nhwc_f32 (dgrad)")
    if _ConvFlops.enabled:
        f = 2.0 * Cout * 9 * Cin * H * W * B
        _ConvFlops.add("dgrad_split_bf16x3", f, f * 6.0)
    return out


def conv3x3_wgrad(srcs, grad_out, Cout):
    """Weight gradient of conv3x3(cat(srcs)): srcs / grad_out channels-last -> grad_weight [Cout, Cin, 3, 3] (fp32 MFMA over
    pixel segments, deterministic two-stage reduction; csrc/conv3x3_wgrad.hip)."""
    srcs = list(srcs) if isinstance(srcs, (list, tuple)) else [srcs]
    B, _, H, W = srcs[0].shape
    Cin = sum(s_.shape[1] for s_ in srcs)
    if tuple(grad_out.shape) != (B, Cout, H, W):
        raise _lib.C2MError("conv3x3_wgrad: grad_out must be [B, Cout, H, W]")
    dev = grad_out.device
    L = _lib.lib()
    arr = (_lib.ConvSrc * 2)()
    for k, s_ in enumerate(srcs):
        arr[k] = _nhwc_src(s_, f"src{k}")
    g = _nhwc_src(grad_out, "grad_out")
    gw = torch.empty((Cout, Cin, 3, 3), dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        nbytes = L.c2m_conv3x3_wgrad_workspace_bytes(B, H, W, Cin, Cout)
        ws = torch.empty(max(nbytes, 4) // 4, dtype=torch.float32, device=dev)
        _lib.check(L.c2m_conv3x3_wgrad_f32(_stream(), arr, len(srcs), grad_out.data_ptr(), g.pix_pitch, g.row_pitch, g.img_pitch,
                                           B, H, W, Cin, Cout, gw.data_ptr(), ws.data_ptr(), nbytes), "c2m_conv3x3_wgrad_f32")
    return gw


def _as_nhwc(t):
    """float32 channels-last view / copy of a [B,C,H,W] tensor (what the kernels' pitch descriptors need)."""
    t = t.float() if t.dtype != torch.float32 else t
    return t if t.stride(1) == 1 else t.contiguous(memory_format=torch.channels_last)


class _Conv3x3Fn(torch.autograd.Function):
    """Differentiable act(conv3x3(cat(srcs)) + bias) on the hand-written kernels: forward = the split-bf16 kernel, backward =
    the same kernel on rotated / transposed weights (data gradient) + the fp32-MFMA weight-gradient kernel + two elementwise
    torch ops (activation mask, bias sum).  Stage-3 training (ref_restoration_model.py:192-269) runs its decoder through this
    instead of cuDNN/MIOpen convolutions."""

    @staticmethod
    def forward(ctx, weight, bias, act, slope, *srcs):
        srcs = [_as_nhwc(s_) for s_ in srcs]
        out = conv3x3(srcs, weight, bias, act=act, slope=slope, algo="split")
        ctx.act, ctx.slope, ctx.nsrc = int(act), float(slope), len(srcs)
        ctx.save_for_backward(weight, *(srcs + ([out] if act != ACT_NONE else [])))
        return out

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, grad_out):
        saved = ctx.saved_tensors
        weight, srcs = saved[0], list(saved[1:1 + ctx.nsrc])
        g = _as_nhwc(grad_out)
        if ctx.act != ACT_NONE:
            out = saved[1 + ctx.nsrc]
            g = g * torch.where(out > 0, 1.0, ctx.slope if ctx.act == ACT_LRELU else 0.0)
            g = _as_nhwc(g)
        need = ctx.needs_input_grad
        gw = conv3x3_wgrad(srcs, g, weight.shape[0]) if need[0] else None
        gb = g.sum(dim=(0, 2, 3)) if need[1] else None
        gsrc = [None] * ctx.nsrc
        if any(need[4:4 + ctx.nsrc]):
            Co = weight.shape[0]
            if Co % 16 != 0:   # the data-gradient convolution sweeps dY in 16-channel chunks: zero-pad (216-channel DCN heads)
                pad = 16 - Co % 16
                dx = conv3x3_dgrad(_as_nhwc(torch.nn.functional.pad(g, (0, 0, 0, 0, 0, pad))),
                                   torch.nn.functional.pad(weight.detach(), (0, 0, 0, 0, 0, 0, 0, pad)))
            else:
                dx = conv3x3_dgrad(g, weight)
            c0 = 0
            for k, s_ in enumerate(srcs):
                if need[4 + k]:
                    gsrc[k] = dx[:, c0:c0 + s_.shape[1]]
                c0 += s_.shape[1]
        return (gw, gb, None, None, *gsrc)


def conv3x3_autograd(srcs, weight, bias=None, act=ACT_NONE, slope=0.1):
    """act(conv3x3(cat(srcs)) + bias) with gradients (channels-last float32 in and out; sources a multiple of 32 channels,
    Cout a multiple of 16).  See _Conv3x3Fn."""
    srcs = list(srcs) if isinstance(srcs, (list, tuple)) else [srcs]
    return _Conv3x3Fn.apply(weight, bias, int(act), float(slope), *srcs)


def conv3x3_autograd_ok(srcs, weight):
    """Shapes the differentiable path supports: 32-channel source blocks (weight-gradient kernel) and 16-byte channels-last
    output vectors (Cout % 4 == 0; the data-gradient convolution pads Cout to a multiple of 16 itself)."""
    srcs = list(srcs) if isinstance(srcs, (list, tuple)) else [srcs]
    return (all(s_.is_cuda and s_.dim() == 4 and s_.shape[1] % 32 == 0 for s_ in srcs) and weight.shape[0] % 4 == 0 and
            sum(s_.shape[1] for s_ in srcs) == weight.shape[1] and tuple(weight.shape[2:]) == (3, 3) and _SPLIT != "0")


def _grouped8_args(g2, B, Cout, H, W, dev):
    if g2 is None:
        return None, 0, 0, 0
    if (g2.dtype != torch.float32 or not g2.is_contiguous() or Cout % 8 != 0 or
            tuple(g2.shape) != (B, Cout // 8, H + 3, W + 3, 8) or g2.device != dev):
        raise _lib.C2MError("out2_grouped8 must be a contiguous float32 [B, Cout/8, H+3, W+3, 8] buffer")
    # image pixel (0, 0) = bordered pixel (1, 1); pitches in floats
    return g2.data_ptr() + ((W + 3) + 1) * 8 * 4, (W + 3) * 8, (H + 3) * (W + 3) * 8, (Cout // 8) * (H + 3) * (W + 3) * 8


def conv3x3_rgb64(image, weight, bias=None, act=ACT_NONE, slope=0.1, mean=None, std=None, out=None, out2_grouped8=None):
    """First layer of an image tower (3 -> 64 channels; vgg conv1_1, conv_first) as its own im2col kernel:
    out = act(conv3x3((image - mean) / std) + bias), channels_last [B,64,H,W].  image: [B,3,H,W] (any layout; read as
    contiguous NCHW); mean / std: [1,3,1,1] buffers of the extractor or None.  out / out2_grouped8 as in conv3x3."""
    x = image.float().contiguous()
    if x.dim() != 4 or x.shape[1] != 3 or not x.is_cuda:
        raise _lib.C2MError("conv3x3_rgb64: image must be a GPU tensor [B,3,H,W]")
    if tuple(weight.shape) != (64, 3, 3, 3):
        raise _lib.C2MError("conv3x3_rgb64: weight must be [64,3,3,3]")
    B, _, H, W = x.shape
    dev = x.device
    if (mean is None) != (std is None):
        raise _lib.C2MError("conv3x3_rgb64: mean and std go together")
    if 12 * H * W >= 2 ** 31 and out2_grouped8 is None:
        # the first-layer kernel addresses one image with 32-bit buffer offsets (12 bytes per pixel: H * W < 178 956 971, e.g.
        # 13 377 x 13 377); beyond that the layer runs on the generic kernel over a 32-channel zero-padded copy (what
        # ContentExtractor.forward_fused does for first layers that are not 3 -> 64) instead of raising
        xn = x if mean is None else (x - mean.detach().reshape(1, 3, 1, 1).float()) / std.detach().reshape(1, 3, 1, 1).float()
        x32 = torch.zeros((B, 32, H, W), dtype=torch.float32, device=dev).contiguous(memory_format=torch.channels_last)
        x32[:, :3] = xn
        with conv_flavour("bf16x3"):   # (explicitly full range: no guard needed, no f16 x 2 domain question)
            return conv3x3(x32, weight, bias, act=act, slope=slope, out=out)
    w = _dev_f32(weight.detach(), "weight")
    bias = _dev_f32(bias.detach(), "bias") if bias is not None else None
    if mean is not None:
        mean, std = _dev_f32(mean.detach().reshape(3), "mean"), _dev_f32(std.detach().reshape(3), "std")
    if out is None:
        out = empty_nhwc(B, 64, H, W, dev)
    o = _nhwc_src(out, "out")
    if tuple(out.shape) != (B, 64, H, W):
        raise _lib.C2MError("conv3x3_rgb64: out must be [B,64,H,W]")
    o2, o2_row, o2_plane, o2_img = _grouped8_args(out2_grouped8, B, 64, H, W, dev)
    with torch.cuda.device(dev):
        _lib.check(_lib.lib().c2m_conv3x3_rgb64_f32(
            _stream(), x.data_ptr(), B, H, W, w.data_ptr(), bias.data_ptr() if bias is not None else None,
            mean.data_ptr() if mean is not None else None, std.data_ptr() if std is not None else None, int(act),
            float(slope), out.data_ptr(), o.pix_pitch, o.row_pitch, o.img_pitch, o2, o2_row, o2_plane, o2_img),
            "c2m_conv3x3_rgb64_f32")
    if _ConvFlops.enabled:
        _ConvFlops.add("rgb_first_layer", 2.0 * 64 * 27 * H * W * B, 2.0 * 64 * 28 * H * W * B)
    return out


def index_to_flow(max_idx):
    """max_idx int64 [B,hq,wq] -> flow float32 [B,hq,wq,2] (x, y), un-padded (corres_generation_arch.py:29-46)."""
    if not max_idx.is_cuda or max_idx.dtype != torch.int64 or max_idx.dim() != 3:
        raise _lib.C2MError("max_idx must be an int64 GPU tensor [B, hq, wq]")
    mi = max_idx.contiguous()
    B, hq, wq = mi.shape
    flow = torch.empty((B, hq, wq, 2), dtype=torch.float32, device=mi.device)
    with torch.cuda.device(mi.device):
        _lib.check(_lib.lib().c2m_index_to_flow_f32(_stream(), mi.data_ptr(), B, hq, wq, flow.data_ptr()), "c2m_index_to_flow_f32")
    return flow


def conv3x3_dcn_head(srcs, weight, bias, deformable_groups, flow=None, scale=1, abs_sum=None, algo=None):
    """The DCN offset/mask head (conv_offset_mask of DCN_sep_pre_multi_offset, dcn_v2.py:229-245) fused with the
    pre-offset construction: -> (offset [B,2*dg*9,H,W], mask [B,dg*9,H,W]) planar, ready for dcn_v2_forward.
    flow: index_to_flow(max_idx) of the matched LR features (or None: no pre-offset); scale = H / h (1, 2, 4).
    The head's channels are computed in slices of 64-channel tiles + one 32-wide remainder (216 = 192 + 24) so that
    no padded tile is multiplied."""
    srcs = list(srcs) if isinstance(srcs, (list, tuple)) else [srcs]
    B, _, H, W = srcs[0].shape
    Cin = sum(s.shape[1] for s in srcs)
    dg = int(deformable_groups)
    Cout = weight.shape[0]
    if Cout != 3 * dg * 9:
        raise _lib.C2MError("conv3x3_dcn_head: weight must have 3*dg*9 output channels")
    dev = srcs[0].device
    bias = _dev_f32(bias.detach(), "bias")
    offset = torch.empty((B, 2 * dg * 9, H, W), dtype=torch.float32, device=dev)
    mask = torch.empty((B, dg * 9, H, W), dtype=torch.float32, device=dev)
    if flow is not None:
        if flow.dtype != torch.float32 or not flow.is_cuda or flow.dim() != 4 or flow.shape[0] != B or flow.shape[3] != 2:
            raise _lib.C2MError("flow must be float32 [B, fh, fw, 2] on the GPU")
        flow = flow.contiguous()
    if abs_sum is not None and (abs_sum.dtype != torch.float64 or abs_sum.numel() < 256 or not abs_sum.is_cuda):
        raise _lib.C2MError("abs_sum must be a float64 GPU tensor with 256 slots (C2M_ABS_SUM_SLOTS)")
    split = (Cout // 64) * 64
    slices = [(0, Cout)] if (split == 0 or Cout - split > 32 or split == Cout) else [(0, split), (split, Cout)]
    use_split = algo in ("split", "bf16", "split16") or (algo is None and _SPLIT != "0" and all(s_.shape[1] % 16 == 0 for s_ in srcs))
    split_id = (4 if (algo == "bf16" or (algo is None and bf16_autocast())) else
                6 if (algo == "split16" or (algo is None and _f16x2_auto())) else 3)
    fam = []
    for (c0, c1) in slices:
        # split-bf16 kernel (any shape); else 64-channel-tileable slices on whole 32-pixel tiles take the Winograd F(2,3)
        # kernel (1.5x fewer matrix instructions), the rest the direct kernel
        wino = split_id if use_split else int(_WINO and algo is None and (c1 - c0) % 64 == 0 and W % 32 == 0 and
                                       all(s_.shape[1] % 16 == 0 for s_ in srcs))
        fam.append(wino)
        wr = _wcache.get(weight, rows=(c0, c1), wino=wino)
        d = _lib.Conv3x3Desc()
        d.algo = _DESC_ALGO[wino]
        d.B, d.H, d.W, d.Cin, d.Cout, d.nsrc = B, H, W, Cin, c1 - c0, len(srcs)
        for k, s in enumerate(srcs):
            d.src[k] = _nhwc_src(s, f"src{k}")
        d.wr = wr.data_ptr()
        d.bias = bias.data_ptr() + 4 * c0
        d.out_mode = 3
        d.out, d.mask_out, d.n_off, d.scale = offset.data_ptr(), mask.data_ptr(), 2 * dg * 9, int(scale)
        d.cout_offset, d.cout_total = c0, Cout
        if flow is not None:
            d.flow, d.fh, d.fw = flow.data_ptr(), flow.shape[1], flow.shape[2]
        if abs_sum is not None:
            d.abs_sum = abs_sum.data_ptr()
        if wino == 6:
            d.range_flag = _range_flag(dev).data_ptr()
        with torch.cuda.device(dev), _apply_switch("head"):
            _lib.check(_lib.lib().c2m_conv3x3_nhwc_f32(_stream(), d), "c2m_conv3x3_nhwc_f32")
    if _ConvFlops.enabled:
        for (c0, c1), w_ in zip(slices, fam):
            _ConvFlops.add("dcn_head_" + _FAMILY[w_], 2.0 * (c1 - c0) * 9 * Cin * H * W * B,
                           2.0 * (c1 - c0) * 9 * Cin * H * W * B * _EXEC_FACTOR[w_])
    return offset, mask


# ---------------------------------------------------------------------------------------------------------------------
# DCNv2 forward on the fused decoder path: bordered channels-last input shared with the convolutions, cached weights,
# channels-last output with the activation folded in
# ---------------------------------------------------------------------------------------------------------------------
class BorderedNHWC:
    """Zero-bordered channels-last copy [B][H+3][W+3][C] of an NCHW feature map (what the DCNv2 kernels gather from).
    `grouped8`: optional twin in the 8-channel group-major layout [B][C/8][H+3][W+3][8], written by the producing
    convolution's epilogue; DCNv2 layers with 8 channels per deformable group gather from it instead."""
    grouped8 = None

    def __new__(cls, x=None):
        pre = bordered_of(x) if x is not None else None   # already the interior view of a bordered buffer: no copy
        return pre if pre is not None else super().__new__(cls)

    def __init__(self, x):
        if bordered_of(x) is self:
            return
        x = _dev_f32(x, "x")
        self.B, self.C, self.H, self.W = x.shape
        self.buf = torch.empty((self.B, self.H + 3, self.W + 3, self.C), dtype=torch.float32, device=x.device)
        with torch.cuda.device(x.device):
            _lib.check(_lib.lib().c2m_nchw_to_nhwc_bordered_f32(_stream(), x.data_ptr(), self.B, self.C, self.H, self.W,
                                                               self.buf.data_ptr()), "c2m_nchw_to_nhwc_bordered_f32")

    def interior(self):
        """Logical [B,C,H,W] channels-last view of the image inside the border (a conv3x3 source)."""
        return self.buf[:, 1:self.H + 1, 1:self.W + 1, :].permute(0, 3, 1, 2)


class _DcnWeightCache(_WeightCache):
    @staticmethod
    def _relayout_dcn(weight, dg, wt=None):
        w = _dev_f32(weight.detach(), "weight")
        Co, C, kh, kw = w.shape
        L = _lib.lib()
        nbytes = L.c2m_dcn_v2_relayout_bytes(C, Co, kh, kw, dg)
        if nbytes == 0:
            raise _lib.C2MError("dcn_v2_forward_nhwc: geometry is not on the channels-last path")
        if wt is None:
            wt = torch.empty(nbytes // 4, dtype=torch.float32, device=w.device)
        with torch.cuda.device(w.device):
            _lib.check(L.c2m_dcn_v2_relayout_f32(_stream(), w.data_ptr(), C, Co, kh, kw, dg, wt.data_ptr()), "c2m_dcn_v2_relayout_f32")
        return wt

    @staticmethod
    def _relayout_dcn16(weight, dg, wt=None):
        w = _dev_f32(weight.detach(), "weight")
        Co, C, kh, kw = w.shape
        L = _lib.lib()
        nbytes = L.c2m_dcn_v2_relayout_f16x2_bytes(C, Co, kh, kw, dg)
        if nbytes == 0:
            raise _lib.C2MError("dcn_v2_forward_nhwc: geometry has no f16 x 2 kernel (needs >= 16 channels per (virtual) group)")
        if wt is None:
            wt = torch.empty(nbytes // 4, dtype=torch.float32, device=w.device)
        with torch.cuda.device(w.device):
            _lib.check(L.c2m_dcn_v2_relayout_f16x2(_stream(), w.data_ptr(), C, Co, kh, kw, dg, wt.data_ptr()), "c2m_dcn_v2_relayout_f16x2")
        return wt

    def get(self, weight, dg, f16x2=False):
        key = (weight.data_ptr(), weight._version, tuple(weight.shape), dg, weight.device.index)
        slot = (id(weight), None, "dcn16") if f16x2 else id(weight)
        hit = self._lookup(slot, key, weight)
        if hit is not None:
            return hit
        fn = self._relayout_dcn16 if f16x2 else self._relayout_dcn
        wt = fn(weight, dg)
        self._store(slot, key, weight, wt, redo=lambda w, buf, dg=dg, fn=fn: fn(w, dg, buf))
        return wt


_dcn_wcache = _DcnWeightCache()


def refresh_weight_caches(module_or_params=None, all_kinds=False):
    """Rebuild the cached weight images (conv3x3 re-layouts, DCNv2 re-layouts) of the given module's / iterable's
    parameters -- or of every live entry -- from the tensors' CURRENT contents; unless all_kinds, only the images the
    CURRENT path multiplies with (no-grad forward: the f16 x 2 / bf16 / bf16 x 3 forward images of the active flavour and the
    fp32-MFMA layouts, not the autograd path's data-gradient images).  The fused inference entry points call
    this once per forward: in-place writes through ``param.data`` do not bump the version counter the caches are keyed
    on, so without it such an update would keep computing with the old weights.  Returns the number of
    images rebuilt.  No host synchronisation."""
    if _os.environ.get("C2M_WEIGHT_REFRESH", "1") == "0":   # measurement only: what the per-forward refresh costs
        return 0
    ids = None
    if module_or_params is not None:
        params = module_or_params.parameters() if hasattr(module_or_params, "parameters") else module_or_params
        ids = {id(p) for p in params}
    kinds = None
    if not all_kinds and not torch.is_grad_enabled():
        kinds = {0, 1, 2, 4} if bf16_autocast() else ({0, 1, 2, 6, 7, 8} if _f16x2_auto() else {0, 1, 2, 3})
    return _wcache.refresh(ids, kinds) + _dcn_wcache.refresh(ids)


def clear_weight_caches():
    """Drop every cached weight image (they are rebuilt on the next use)."""
    _wcache.clear()
    _dcn_wcache.clear()


def dcn_f16x2_ok(weight, deformable_groups):
    Co, C, kh, kw = weight.shape
    return _lib.lib().c2m_dcn_v2_relayout_f16x2_bytes(C, Co, kh, kw, int(deformable_groups)) != 0


def dcn_v2_forward_nhwc(inp_bordered, weight, bias, offset, mask, deformable_groups, act=ACT_NONE, slope=0.1,
                        nhwc_out=True, algo=None):
    """3x3 / stride 1 / pad 1 DCNv2 forward from a BorderedNHWC input; planar offset / mask as dcn_v2_forward takes them.
    -> channels_last [B,Co,H,W] (nhwc_out) or contiguous NCHW, with the activation applied.
    algo: "fp32" (implicit GEMM on the fp32 matrix pipe), "f16x2" (fp32 result on the f16 pipe, three products per k step:
    the convolutions' f16 x 2 arithmetic, domain |sample| < 65520 reported through the same range flag), None: f16 x 2
    wherever the convolutions of this thread run it by default (_f16x2_auto: inside an f16_range_guard -- the fused module
    forwards -- or under `ops.conv_flavour("f16x2")`; $C2M_DCN_F16X2=0 keeps fp32) and the geometry has that kernel; a bare
    direct call runs fp32."""
    if not isinstance(inp_bordered, BorderedNHWC):
        raise _lib.C2MError("inp_bordered must be a BorderedNHWC")
    B, C, H, W = inp_bordered.B, inp_bordered.C, inp_bordered.H, inp_bordered.W
    Co = weight.shape[0]
    dg = int(deformable_groups)
    offset, mask, bias = _dev_f32(offset, "offset"), _dev_f32(mask, "mask"), _dev_f32(bias.detach(), "bias")
    if tuple(offset.shape) != (B, 2 * dg * 9, H, W) or tuple(mask.shape) != (B, dg * 9, H, W):
        raise _lib.C2MError("offset/mask shape does not match [B, 2*dg*9, H, W] / [B, dg*9, H, W]")
    if algo is None:
        f16 = _DCN_F16X2 and _SPLIT != "0" and _f16x2_auto() and dcn_f16x2_ok(weight, dg)
    elif algo in ("fp32", "f16x2"):
        f16 = algo == "f16x2"
    else:
        raise _lib.C2MError("dcn_v2_forward_nhwc: algo is None, 'fp32' or 'f16x2'")
    wt = _dcn_wcache.get(weight, dg, f16x2=f16)
    dev = offset.device
    if nhwc_out:
        out = empty_nhwc(B, Co, H, W, dev)
        o = _nhwc_src(out, "out")
        pitches = (1, o.pix_pitch, o.row_pitch, o.img_pitch)
    else:
        out = torch.empty((B, Co, H, W), dtype=torch.float32, device=dev)
        pitches = (0, 0, 0, 0)
    grouped = inp_bordered.grouped8 is not None and C == 8 * dg
    src = inp_bordered.grouped8 if grouped else inp_bordered.buf
    with torch.cuda.device(dev):
        if f16:
            _lib.check(_lib.lib().c2m_dcn_v2_forward_nhwc_f16x2(_stream(), src.data_ptr(), wt.data_ptr(), bias.data_ptr(),
                                                               offset.data_ptr(), mask.data_ptr(), B, C, H, W, Co, 3, 3, 1, 1, 1,
                                                               1, 1, 1, dg, out.data_ptr(), *pitches, int(act), float(slope),
                                                               int(grouped), _range_flag(dev).data_ptr()),
                       "c2m_dcn_v2_forward_nhwc_f16x2")
        else:
            _lib.check(_lib.lib().c2m_dcn_v2_forward_nhwc_f32(_stream(), src.data_ptr(), wt.data_ptr(), bias.data_ptr(),
                                                             offset.data_ptr(), mask.data_ptr(), B, C, H, W, Co, 3, 3, 1, 1, 1, 1,
                                                             1, 1, dg, out.data_ptr(), *pitches, int(act), float(slope),
                                                             int(grouped)),
                       "c2m_dcn_v2_forward_nhwc_f32")
    return out


# ---------------------------------------------------------------------------------------------------------------------
# VGG-style feature stacks (conv3x3 + ReLU + 2x2 max-pool) on the channels-last kernels
# ---------------------------------------------------------------------------------------------------------------------
def _bordered_empty(B, C, H, W, device, grouped8=False):
    """BorderedNHWC whose border is zero and whose interior is uninitialised (to be written by a conv epilogue).
    grouped8: also allocate the group-major twin (same border)."""
    o = BorderedNHWC.__new__(BorderedNHWC)
    o.B, o.C, o.H, o.W = B, C, H, W
    o.buf = torch.empty((B, H + 3, W + 3, C), dtype=torch.float32, device=device)
    o.buf[:, 0].zero_()
    o.buf[:, H + 1:].zero_()
    o.buf[:, 1:H + 1, 0].zero_()
    o.buf[:, 1:H + 1, W + 1:].zero_()
    if grouped8:
        if C % 8 != 0:
            raise _lib.C2MError("a group-major twin needs C % 8 == 0")
        g = torch.empty((B, C // 8, H + 3, W + 3, 8), dtype=torch.float32, device=device)
        g[:, :, 0].zero_()
        g[:, :, H + 1:].zero_()
        g[:, :, 1:H + 1, 0].zero_()
        g[:, :, 1:H + 1, W + 1:].zero_()
        o.grouped8 = g
    return o


def bordered_of(t):
    """The BorderedNHWC a tensor is the interior view of (set by vgg_stack_forward), else None."""
    return getattr(t, "_c2m_bordered", None)


def _pool_is_2x2(m):
    two = lambda v: v in (2, (2, 2))   # noqa: E731
    return two(m.kernel_size) and two(m.stride) and m.padding in (0, (0, 0)) and m.dilation in (1, (1, 1)) and not m.ceil_mode


def vgg_stack_forward(layers, x, taps=(), mean=None, std=None, last_nchw=False, grouped8_taps=(), fast=False):
    """Run an ordered {name: nn.Conv2d(3x3, pad 1) | nn.ReLU | nn.MaxPool2d(2, 2)} stack (torchvision's vgg `features`
    layout, mmsr/models/archs/vgg_arch.py:107-123) on the fused channels-last convolution: every conv + its ReLU is one
    launch.  x: [B,3,H,W] image; (x - mean) / std is applied while the image is widened to the kernel's 32-channel chunk.
    Returns {tap name: tensor}; tapped activations are written by the conv epilogue straight into zero-bordered
    channels-last buffers (logical NCHW views of their interiors are returned: the DCNv2 gathers and the offset
    convolutions of the decoder read those buffers in place).  last_nchw: the final layer's output as a contiguous NCHW
    tensor under the key of that layer (the correlation kernels read planar features).  grouped8_taps: taps that also get
    the group-major twin (BorderedNHWC.grouped8) a DCNv2 layer with 8-channel deformable groups gathers from.  fast: as in
    conv3x3 (the split-bf16 kernel for every layer but the first): the VGG taps of the Ref do, the extractor towers that feed
    the index search do not."""
    names = list(layers.keys())
    B, C, H, W = x.shape
    dev = x.device
    # cached weight images follow in-place writes through .data (which the version counter does not see)
    refresh_weight_caches([p_ for layer_ in layers.values() for p_ in layer_.parameters()])
    first = layers[names[0]]
    rgb64 = C == 3 and isinstance(first, torch.nn.Conv2d) and tuple(first.weight.shape) == (64, 3, 3, 3)
    if rgb64:
        cur = x      # the first layer reads the planar image itself and normalises it while staging
    else:
        cur = torch.zeros((B, 32, H, W), dtype=torch.float32, device=dev).contiguous(memory_format=torch.channels_last)
        xin = x.float()
        if mean is not None:
            xin = (xin - mean) / std
        cur[:, :C] = xin
    out, k = {}, 0

    def conv(k_, src, layer_, **kw):
        if k_ == 0 and rgb64:
            kw.pop("algo", None)
            return conv3x3_rgb64(src, layer_.weight, layer_.bias, mean=mean, std=std, **kw)
        return conv3x3(src, layer_.weight, layer_.bias, fast=fast, **kw)

    while k < len(names):
        name, layer = names[k], layers[names[k]]
        if isinstance(layer, torch.nn.Conv2d):
            if layer.kernel_size != (3, 3) or layer.stride != (1, 1) or layer.padding != (1, 1) or layer.groups != 1:
                raise _lib.C2MError(f"vgg_stack_forward: {name} is not a 3x3 / stride 1 / pad 1 convolution")
            relu = k + 1 < len(names) and isinstance(layers[names[k + 1]], torch.nn.ReLU)
            tap_name = names[k + 1] if (relu and names[k + 1] in taps) else (name if name in taps and not relu else None)
            if name in taps and relu:
                raise _lib.C2MError("tapping a conv output that is followed by an in-place ReLU is not supported")
            last = (k + (2 if relu else 1)) >= len(names)
            Bc, _, Hc, Wc = cur.shape
            nxt = layers[names[k + 2]] if (relu and k + 2 < len(names)) else None
            pool = (isinstance(nxt, torch.nn.MaxPool2d) and _pool_is_2x2(nxt) and tap_name is None and not (k == 0 and rgb64) and
                    Hc % 2 == 0 and Wc % 2 == 0 and
                    (_wino_ok([cur], layer.weight, "nhwc_pool2", Wc) or _split_ok([cur], layer.weight, fast)))
            if last and last_nchw:
                cur = conv3x3(cur, layer.weight, layer.bias, act=ACT_RELU if relu else ACT_NONE, out_mode="nchw", fast=fast)
                out[names[k + 1] if relu else name] = cur
            elif tap_name is not None:
                bo = _bordered_empty(Bc, layer.out_channels, Hc, Wc, dev, grouped8=tap_name in grouped8_taps)
                view = bo.interior()
                conv(k, cur, layer, act=ACT_RELU if relu else ACT_NONE, out=view, out2_grouped8=bo.grouped8)
                view._c2m_bordered = bo
                out[tap_name] = view
                cur = view
            elif pool:   # conv -> ReLU -> MaxPool2d(2, 2) in one launch: only the pooled map is written
                cur = conv3x3(cur, layer.weight, layer.bias, act=ACT_RELU, out_mode="nhwc_pool2", fast=fast)
                k += 1   # (the pool layer)
            else:
                cur = conv(k, cur, layer, act=ACT_RELU if relu else ACT_NONE)
            k += 2 if relu else 1
        elif isinstance(layer, torch.nn.MaxPool2d):
            cur = torch.nn.functional.max_pool2d(cur, layer.kernel_size, layer.stride, layer.padding)
            k += 1
        elif isinstance(layer, torch.nn.ReLU):
            cur = torch.relu(cur)
            k += 1
        else:
            raise _lib.C2MError(f"vgg_stack_forward: unsupported layer {name}: {type(layer).__name__}")
    return out
