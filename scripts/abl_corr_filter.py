#!/usr/bin/env python3
"""Kernel time of the correlation's filter sweep alone (HIP events, c2m_profile_*), for A/B builds ($C2M_LIB) and the
compile-time ablations of corr_filter.hip (C2M_CORRF_ABL).  usage: abl_corr_filter.py [B] [size]"""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "c2-matching_amd"))
import torch
import c2m_amd

B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
N = int(sys.argv[2]) if len(sys.argv) > 2 else 160
g = torch.Generator(device="cuda").manual_seed(1)
fi = torch.nn.functional.normalize(torch.randn((B, 256, N, N), generator=g, device="cuda"), dim=1)
fr = torch.nn.functional.normalize(torch.randn((B, 256, N, N), generator=g, device="cuda"), dim=1)
c2m_amd.ops.feature_match_index_batched(fi, fr, 3, 1, 1, True, True)
c2m_amd.profile_enable(True); c2m_amd.profile_collect()
for _ in range(5):
    c2m_amd.ops.feature_match_index_batched(fi, fr, 3, 1, 1, True, True)
torch.cuda.synchronize()
rows = c2m_amd.profile_collect()
c2m_amd.profile_enable(False)
by = {}
for nm, ms in rows:
    by.setdefault(nm, []).append(ms)
print({k: round(sum(v) / len(v), 3) for k, v in by.items()}, "lib", os.environ.get("C2M_LIB", "in-tree"))
