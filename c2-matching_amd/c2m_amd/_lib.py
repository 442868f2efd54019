"""ctypes loader for libc2m_hip.so (the C-ABI of include/c2m_hip.h).

There is NO fallback: if the shared library is missing, was built for another ABI version, or a call returns a
non-zero status, this module raises.  torch must be imported first so that the library binds to the HIP runtime
already loaded by PyTorch-ROCm (same libamdhip64.so.7 soname) and shares its device pointers and streams.
"""
import ctypes
import os

import torch  # noqa: F401  (loads libamdhip64 before our library resolves it)

_HERE = os.path.dirname(os.path.abspath(__file__))
# ($C2M_LIB: another build of the same library -- kernel A/B measurements; the product path is the in-tree build)
LIB_PATH = os.environ.get("C2M_LIB") or os.path.join(os.path.dirname(_HERE), "csrc", "libc2m_hip.so")
ABI_VERSION = 3

_vp, _i, _sz = ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t
_lib = None


class ConvSrc(ctypes.Structure):
    """c2m_conv_src of include/c2m_hip.h"""
    _fields_ = [("ptr", _vp), ("C", _i), ("pix_pitch", _i), ("row_pitch", _i), ("img_pitch", ctypes.c_longlong)]


class Conv3x3Desc(ctypes.Structure):
    """c2m_conv3x3_desc of include/c2m_hip.h"""
    _fields_ = [("B", _i), ("H", _i), ("W", _i), ("Cin", _i), ("Cout", _i), ("nsrc", _i), ("src", ConvSrc * 2),
                ("wr", _vp), ("bias", _vp), ("act", _i), ("slope", ctypes.c_float), ("out_mode", _i), ("out", _vp),
                ("out_pix_pitch", _i), ("out_row_pitch", _i), ("out_img_pitch", ctypes.c_longlong), ("res1", _vp),
                ("res2", _vp), ("mask_out", _vp), ("flow", _vp), ("fh", _i), ("fw", _i), ("scale", _i), ("n_off", _i),
                ("abs_sum", _vp), ("algo", _i), ("cout_offset", _i), ("cout_total", _i),
                ("out2", _vp), ("out2_row_pitch", _i), ("out2_plane_pitch", ctypes.c_longlong),
                ("out2_img_pitch", ctypes.c_longlong), ("range_flag", _vp), ("io_flags", ctypes.c_int)]


class ResBlockDesc(ctypes.Structure):
    """c2m_resblock3x3_desc of include/c2m_hip.h"""
    _fields_ = [("B", _i), ("H", _i), ("W", _i), ("C", _i), ("x", _vp), ("x_pix_pitch", _i), ("x_row_pitch", _i),
                ("x_img_pitch", ctypes.c_longlong), ("out", _vp), ("out_pix_pitch", _i), ("out_row_pitch", _i),
                ("out_img_pitch", ctypes.c_longlong), ("res2", _vp), ("wr1", _vp), ("wr2", _vp), ("bias1", _vp), ("bias2", _vp),
                ("range_flag", _vp)]


class C2MError(RuntimeError):
    pass


def _declare(L):
    L.c2m_abi_version.restype = _i
    L.c2m_status_string.restype = ctypes.c_char_p
    L.c2m_status_string.argtypes = [_i]
    L.c2m_last_hip_error.restype = ctypes.c_char_p
    L.c2m_device_arch.argtypes = [ctypes.c_char_p, _i]
    L.c2m_profile_enable.argtypes = [_i]
    L.c2m_profile_collect.argtypes = [ctypes.POINTER(ctypes.c_float), ctypes.POINTER(_i), _i, ctypes.POINTER(_i)]
    L.c2m_feature_normalize_f32.argtypes = [_vp, _vp, _i, _i, _i, _vp]
    L.c2m_feature_match_workspace_bytes.restype = _sz
    L.c2m_feature_match_workspace_bytes.argtypes = [_i] * 5
    L.c2m_feature_match_workspace_bytes_c.restype = _sz
    L.c2m_feature_match_workspace_bytes_c.argtypes = [_i] * 6
    L.c2m_feature_match_skip_table.argtypes = [_i] * 5 + [ctypes.POINTER(ctypes.c_size_t), ctypes.POINTER(ctypes.c_int)]
    L.c2m_feature_match_index_f32.argtypes = [_vp, _vp, _vp] + [_i] * 12 + [_vp, _vp, _vp, _sz]
    L.c2m_feature_match_index_pre_f32.argtypes = [_vp, _vp, _vp] + [_i] * 12 + [_vp, _vp, _vp, _sz, _vp, _vp]
    L.c2m_feature_match_index_pre_f32.restype = _i
    L.c2m_feature_normalize_ss_f32.argtypes = [_vp, _vp, _i, _i, _i, _vp, _vp]
    L.c2m_feature_normalize_ss_f32.restype = _i
    L.c2m_feature_match_set_filter.argtypes = [_i]
    L.c2m_conv3x3_set_head_stores.argtypes = [_i]
    L.c2m_feature_match_filter_tables.argtypes = [_i] * 5 + [ctypes.POINTER(ctypes.c_size_t)] * 3 + [ctypes.POINTER(ctypes.c_int)]
    L.c2m_build_pre_offsets_f32.argtypes = [_vp, _vp, _i, _i, _i, _vp, _vp, _vp]
    for name in ("c2m_dcn_v2_forward_workspace_bytes",):
        getattr(L, name).restype = _sz
        getattr(L, name).argtypes = [_i] * 8
    L.c2m_dcn_v2_backward_workspace_bytes.restype = _sz
    L.c2m_dcn_v2_backward_workspace_bytes.argtypes = [_i] * 14
    L.c2m_dcn_v2_forward_f32.argtypes = [_vp] * 6 + [_i] * 14 + [_vp, _vp, _sz]
    L.c2m_dcn_v2_forward_bf16mma_f32.argtypes = [_vp] * 6 + [_i] * 14 + [_vp, _vp, _sz]
    L.c2m_dcn_v2_backward_f32.argtypes = [_vp] * 7 + [_i] * 14 + [_vp] * 5 + [_vp, _sz]
    L.c2m_dcn_fuse_offsets_f32.argtypes = [_vp, _vp, _vp, _i, _i, _i, _i, _i, _vp, _vp, _vp]
    L.c2m_nchw_to_nhwc_bordered_f32.argtypes = [_vp, _vp, _i, _i, _i, _i, _vp]
    L.c2m_dcn_v2_relayout_bytes.restype = _sz
    L.c2m_dcn_v2_relayout_bytes.argtypes = [_i] * 5
    L.c2m_dcn_v2_relayout_f32.argtypes = [_vp, _vp, _i, _i, _i, _i, _i, _vp]
    L.c2m_conv3x3_rgb64_f32.argtypes = [_vp, _vp, _i, _i, _i, _vp, _vp, _vp, _vp, _i, ctypes.c_float, _vp, _i, _i,
                                        ctypes.c_longlong, _vp, _i, ctypes.c_longlong, ctypes.c_longlong]
    L.c2m_dcn_v2_forward_nhwc_f32.argtypes = [_vp] * 6 + [_i] * 14 + [_vp, _i, _i, _i, ctypes.c_longlong, _i, ctypes.c_float, _i]
    L.c2m_dcn_v2_relayout_f16x2_bytes.restype = _sz
    L.c2m_dcn_v2_relayout_f16x2_bytes.argtypes = [_i] * 5
    L.c2m_dcn_v2_relayout_f16x2.argtypes = [_vp, _vp, _i, _i, _i, _i, _i, _vp]
    L.c2m_dcn_v2_forward_nhwc_f16x2.argtypes = [_vp] * 6 + [_i] * 14 + [_vp, _i, _i, _i, ctypes.c_longlong, _i, ctypes.c_float, _i, _vp]
    L.c2m_conv3x3_relayout_bytes.restype = _sz
    L.c2m_conv3x3_relayout_bytes.argtypes = [_i, _i]
    L.c2m_conv3x3_relayout_f32.argtypes = [_vp, _vp, _i, _i, _vp]
    L.c2m_conv3x3_relayout_wino_bytes.restype = _sz
    L.c2m_conv3x3_relayout_wino_bytes.argtypes = [_i, _i]
    L.c2m_conv3x3_relayout_wino_f32.argtypes = [_vp, _vp, _i, _i, _vp]
    L.c2m_conv3x3_relayout_wino4_bytes.restype = _sz
    L.c2m_conv3x3_relayout_wino4_bytes.argtypes = [_i, _i]
    L.c2m_conv3x3_relayout_wino4_f32.argtypes = [_vp, _vp, _i, _i, _vp]
    L.c2m_conv3x3_relayout_split_bytes.restype = _sz
    L.c2m_conv3x3_relayout_split_bytes.argtypes = [_i, _i, _i]
    L.c2m_conv3x3_relayout_split_f32.argtypes = [_vp, _vp, _i, _i, _i, _vp]
    L.c2m_conv3x3_relayout_split_dgrad_f32.argtypes = [_vp, _vp, _i, _i, _i, _vp]
    L.c2m_conv3x3_relayout_split_multi.argtypes = [_vp, _vp, _i, ctypes.c_longlong, _i]
    L.c2m_conv3x3_wgrad_workspace_bytes.restype = _sz
    L.c2m_conv3x3_wgrad_workspace_bytes.argtypes = [_i] * 5
    L.c2m_conv3x3_wgrad_f32.argtypes = [_vp, ctypes.POINTER(ConvSrc), _i, _vp, _i, _i, ctypes.c_longlong] + [_i] * 5 + [_vp, _vp, _sz]
    L.c2m_conv3x3_nhwc_f32.argtypes = [_vp, ctypes.POINTER(Conv3x3Desc)]
    L.c2m_resblock3x3_nhwc_f32.argtypes = [_vp, ctypes.POINTER(ResBlockDesc)]
    L.c2m_resblock3x3_nhwc_f32.restype = _i
    L.c2m_resblock3x3_supported.argtypes = [_i, _i, _i]
    L.c2m_resblock3x3_supported.restype = _i
    L.c2m_index_to_flow_f32.argtypes = [_vp, _vp, _i, _i, _i, _vp]


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise C2MError(f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                           "(hipcc --offload-arch=gfx950).  There is no CPU or PyTorch fallback for this path.")
        L = ctypes.CDLL(LIB_PATH)
        _declare(L)
        v = L.c2m_abi_version()
        if v != ABI_VERSION:
            raise C2MError(f"libc2m_hip.so ABI {v} != expected {ABI_VERSION}")
        _lib = L
    return _lib


def check(status, what):
    if status != 0:
        L = lib()
        msg = L.c2m_status_string(status).decode()
        hip = L.c2m_last_hip_error().decode() if status == 4 else ""
        raise C2MError(f"{what}: {msg}" + (f" [{hip}]" if hip else ""))


def device_arch():
    buf = ctypes.create_string_buffer(256)
    check(lib().c2m_device_arch(buf, 256), "c2m_device_arch")
    return buf.value.decode()


KERNEL_NAMES = {1: "corr_argmax_mfma", 2: "corr_argmax_generic", 3: "dcn_v2_forward", 4: "dcn_v2_backward_data",
                5: "dcn_v2_backward_weight", 6: "conv3x3_mfma", 7: "conv3x3_split", 8: "conv3x3_wgrad", 9: "corr_filter",
                10: "corr_resolve"}


def profile_enable(on=True):
    check(lib().c2m_profile_enable(int(bool(on))), "c2m_profile_enable")


def profile_collect(capacity=512):
    """-> list of (kernel name, milliseconds) for every dominant-kernel launch since the last collect."""
    ms = (ctypes.c_float * capacity)()
    ids = (_i * capacity)()
    n = _i(0)
    check(lib().c2m_profile_collect(ms, ids, capacity, ctypes.byref(n)), "c2m_profile_collect")
    return [(KERNEL_NAMES.get(ids[k], str(ids[k])), float(ms[k])) for k in range(n.value)]
