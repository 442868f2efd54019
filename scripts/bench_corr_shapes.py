#!/usr/bin/env python3
"""Kernel time of the correlation/arg-max launch for arbitrary query / ref map shapes (HIP events via c2m_profile_*).
usage: bench_corr_shapes.py B C Hq Wq Hr Wr [B C Hq Wq Hr Wr ...]"""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "c2-matching_amd"))
import torch
import c2m_amd

def run(B, C, Hq, Wq, Hr, Wr, reps=3):
    g = torch.Generator(device="cuda").manual_seed(1)
    fi = torch.nn.functional.normalize(torch.randn((B, C, Hq, Wq), generator=g, device="cuda"), dim=1)
    fr = torch.nn.functional.normalize(torch.randn((B, C, Hr, Wr), generator=g, device="cuda"), dim=1)
    c2m_amd.ops.feature_match_index_batched(fi, fr, 3, 1, 1, True, True)
    c2m_amd.profile_enable(True); c2m_amd.profile_collect()
    for _ in range(reps):
        c2m_amd.ops.feature_match_index_batched(fi, fr, 3, 1, 1, True, True)
    torch.cuda.synchronize()
    ms = [m for (nm, m) in c2m_amd.profile_collect() if nm == "corr_argmax_mfma"]
    c2m_amd.profile_enable(False)
    return sum(ms) / len(ms)

a = [int(x) for x in sys.argv[1:]]
for i in range(0, len(a), 6):
    print(a[i:i + 6], "kernel ms %.3f" % run(*a[i:i + 6]))
