#!/usr/bin/env python3
"""Kernel time of the 3 -> 64 first-layer kernel (HIP events, c2m_profile_*) for A/B builds ($C2M_LIB) and the compile-time
ablations of conv3x3_c3_kernel (C2M_C3_ABL).  usage: abl_c3.py [B] [size]"""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "c2-matching_amd"))
import torch
import c2m_amd
from c2m_amd import ops

B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
N = int(sys.argv[2]) if len(sys.argv) > 2 else 640
TWIN = len(sys.argv) > 3 and sys.argv[3] == "twin"   # bordered destination + 8-channel group-major twin (what the VGG tap writes)
g = torch.Generator(device="cuda").manual_seed(1)
img = torch.rand((B, 3, N, N), generator=g, device="cuda")
w = torch.randn((64, 3, 3, 3), generator=g, device="cuda") * 0.2
b = torch.randn((64,), generator=g, device="cuda")
mean = torch.tensor([0.485, 0.456, 0.406], device="cuda").view(1, 3, 1, 1)
std = torch.tensor([0.229, 0.224, 0.225], device="cuda").view(1, 3, 1, 1)
out = ops.conv3x3_rgb64(img, w, b, act=1, mean=mean, std=std)
want = torch.nn.functional.conv2d(((img - mean) / std)[:1], w, b, padding=1).relu()
err = float((out[:1] - want).abs().max())
kw = {"out": out}
if TWIN:
    bo = ops._bordered_empty(B, 64, N, N, "cuda", grouped8=True)
    kw = {"out": bo.interior(), "out2_grouped8": bo.grouped8}
    ops.conv3x3_rgb64(img, w, b, act=1, mean=mean, std=std, **kw)
    err = max(err, float((bo.interior()[:1] - want).abs().max()))
c2m_amd.profile_enable(True); c2m_amd.profile_collect()
for _ in range(10):
    ops.conv3x3_rgb64(img, w, b, act=1, mean=mean, std=std, **kw)
torch.cuda.synchronize()
ms = [m for (nm, m) in c2m_amd.profile_collect()]
c2m_amd.profile_enable(False)
ms.sort()
gb = B * 64 * N * N * 4 / 1e9
print({"ms_median": round(ms[len(ms) // 2], 4), "ms_min": round(ms[0], 4), "write_GBs": round(gb / ms[len(ms) // 2] * 1e3, 1),
       "max_err_vs_torch": err, "twin": TWIN, "lib": os.environ.get("C2M_LIB", "in-tree")})
