"""CPU: the C-ABI library exists, loads, and exports every symbol include/c2m_hip.h declares (no compute calls)."""
import ctypes
import os
import re

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    hdr = open(os.path.join(REPO, "include", "c2m_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    return sorted(set(re.findall(r"\b(c2m_[a-z0-9_]+)\s*\(", hdr)))


def test_header_declares_the_expected_entry_points():
    names = _declared()
    for must in ("c2m_feature_match_index_f32", "c2m_feature_normalize_f32", "c2m_build_pre_offsets_f32",
                 "c2m_dcn_v2_forward_f32", "c2m_dcn_v2_forward_bf16mma_f32", "c2m_dcn_v2_backward_f32", "c2m_dcn_fuse_offsets_f32", "c2m_abi_version"):
        assert must in names


def test_library_exports_every_declared_symbol():
    import c2m_amd
    assert os.path.exists(c2m_amd.LIB_PATH), "run __graft_entry__.build() first"
    lib = ctypes.CDLL(c2m_amd.LIB_PATH)
    missing = [n for n in _declared() if not hasattr(lib, n)]
    assert not missing, missing
    assert lib.c2m_abi_version() == 3
    lib.c2m_status_string.restype = ctypes.c_char_p
    assert lib.c2m_status_string(0) == b"ok" and b"workspace" in lib.c2m_status_string(3)


def test_product_path_never_touches_the_oracle():
    pkg = os.path.join(REPO, "c2-matching_amd")
    for root, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                src = open(os.path.join(root, f)).read()
                assert "c2m_oracle" not in src and "torch_port" not in src and "oracle/" not in src.replace("the oracle", ""), os.path.join(root, f)


def test_cpu_tensors_are_rejected_not_emulated():
    import pytest
    import torch
    import c2m_amd
    with pytest.raises(c2m_amd.C2MError):
        c2m_amd.ops.feature_match_index_batched(torch.zeros(1, 4, 8, 8), torch.zeros(1, 4, 8, 8))
    with pytest.raises(c2m_amd.C2MError):
        c2m_amd.ops.dcn_v2_forward(torch.zeros(1, 4, 5, 5), torch.zeros(2, 4, 3, 3), torch.zeros(2), torch.zeros(1, 18, 5, 5), torch.zeros(1, 9, 5, 5))


def test_correlation_workspace_is_sized_by_the_maps_channels():
    """the pre-filter's per-channel scratch follows C (none where the filter has no kernel); the C-less query keeps
    returning the 256-channel upper bound, which every layout fits (no GPU needed: pure size arithmetic in the library)."""
    import c2m_amd
    L = c2m_amd._lib.lib()
    shp = (16, 160, 160, 160, 160)
    upper = L.c2m_feature_match_workspace_bytes(*shp)
    by_c = {c: L.c2m_feature_match_workspace_bytes_c(shp[0], c, *shp[1:]) for c in (32, 64, 128, 256)}
    assert by_c[256] == upper and by_c[64] < by_c[128] < by_c[256]
    assert by_c[32] < 64 << 20 < by_c[64]                 # no filter kernel for C = 32: tables only
    per_c = (by_c[256] - by_c[128]) / 128                 # bytes per channel of the four per-channel blocks
    assert abs((by_c[128] - by_c[64]) / 64 - per_c) <= 1024 and abs(by_c[64] - by_c[32] - 64 * per_c) <= 4096
    assert L.c2m_feature_match_workspace_bytes_c(0, 256, 160, 160, 160, 160) == 0
