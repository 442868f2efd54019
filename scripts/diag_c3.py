#!/usr/bin/env python3
"""Where does the 3 -> 64 first-layer kernel differ from torch?  (debug aid)"""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "c2-matching_amd"))
import torch
from c2m_amd import ops
for (B, H, W, norm) in [(3, 8, 64, True), (1, 8, 64, False), (2, 40, 40, True), (1, 37, 75, False), (1, 131, 200, True), (2, 5, 3, True), (16, 640, 640, True), (4, 1280, 1280, True), (1, 3000, 3000, True)]:   # last: one image's output > 2^31 bytes
    g = torch.Generator(device="cuda").manual_seed(5)
    img = torch.rand((B, 3, H, W), generator=g, device="cuda")
    w = torch.randn((64, 3, 3, 3), generator=g, device="cuda") * 0.2
    b = torch.randn((64,), generator=g, device="cuda")
    mean = torch.tensor([0.485, 0.456, 0.406], device="cuda").view(1, 3, 1, 1)
    std = torch.tensor([0.229, 0.224, 0.225], device="cuda").view(1, 3, 1, 1)
    got = ops.conv3x3_rgb64(img, w, b, act=0, mean=mean if norm else None, std=std if norm else None)
    bo = ops._bordered_empty(B, 64, H, W, "cuda", grouped8=True)
    ops.conv3x3_rgb64(img, w, b, act=0, mean=mean if norm else None, std=std if norm else None, out=bo.interior(), out2_grouped8=bo.grouped8)
    twin_ok = torch.equal(bo.interior(), got) and torch.equal(bo.grouped8, bo.buf.view(B, H + 3, W + 3, 8, 8).permute(0, 3, 1, 2, 4).contiguous())
    want = torch.cat([torch.nn.functional.conv2d(((img[i:i + 1] - mean) / std) if norm else img[i:i + 1], w, b, padding=1) for i in range(B)])
    d = (got - want).abs()
    print("  bordered + group-major twin identical:", twin_ok)
    print((B, H, W, norm), "max err", float(d.max()))
    if float(d.max()) > 1e-4:
        badpix = (d.amax(dim=1) > 1e-4)
        print("  bad pixels per image:", badpix.flatten(1).sum(1).tolist(), "of", H * W)
        ys, xs = torch.nonzero(badpix[0], as_tuple=True)
        print("  image 0 bad rows:", sorted(set(ys.tolist()))[:20], "cols:", sorted(set(xs.tolist()))[:70])
        print("  bad channels:", torch.nonzero(d.amax(dim=(0, 2, 3)) > 1e-4).flatten().tolist()[:64])
