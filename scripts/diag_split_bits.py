#!/usr/bin/env python3
"""SHA-256 of f16 x 2 convolution outputs over shapes, activation scales and epilogue modes -- run under two builds of the library
($C2M_LIB) and diff the lines: a change to the split that claims bit-identical results (C2M_SPLIT_MIX) must print the same hashes."""
import hashlib, os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "c2-matching_amd"))
import torch
from c2m_amd import ops

dev = torch.device("cuda:0")


def h(t):
    return hashlib.sha256(t.detach().contiguous().cpu().numpy().tobytes()).hexdigest()[:16]


for (B, Ci, Co, H, W) in [(1, 64, 64, 33, 47), (2, 64, 64, 160, 160), (1, 128, 64, 96, 80), (1, 256, 256, 40, 40), (1, 64, 216, 64, 64), (4, 64, 64, 320, 320)]:
    for xs in (1.0, 1e-3, 3e-6, 1e-9, 2.0e4):
        g = torch.Generator(device=dev).manual_seed(B * 1000 + Ci + H)
        x = (torch.randn((B, Ci, H, W), generator=g, device=dev) * xs).contiguous(memory_format=torch.channels_last)
        w = torch.randn((Co, Ci, 3, 3), generator=g, device=dev) * 0.03
        b = torch.randn((Co,), generator=g, device=dev) * 0.1
        y = ops.conv3x3(x, w, b, act=ops.ACT_RELU, algo="split16")
        r = ops.conv3x3(x, w, b, res1=y, algo="split16") if Ci == Co else y
        torch.cuda.synchronize()
        print({"shape": (B, Ci, Co, H, W), "x_scale": xs, "relu": h(y), "res": h(r), "finite": bool(torch.isfinite(r).all())}, flush=True)
