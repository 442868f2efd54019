#!/usr/bin/env python3
"""Debug: the DCN head / planar outputs with and without the four-pixel staging (subprocess per setting: the switch is read once)."""
import os, subprocess, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CODE = r'''
import sys, torch
sys.path.insert(0, "%s/c2-matching_amd")
from c2m_amd import ops
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(1)
out = {}
for (B, Cin, H, W, scale, fmode) in ((1, 64, 160, 160, 4, "rand"), (1, 64, 160, 160, 4, "none"), (1, 64, 320, 320, 4, "rand")):
    x = torch.randn((B, Cin, H, W), generator=g, device=dev).contiguous(memory_format=torch.channels_last)
    w = torch.randn((216, Cin, 3, 3), generator=g, device=dev) * 0.02
    b = torch.randn((216,), generator=g, device=dev) * 0.1
    fh, fw = H // scale - 2, W // scale - 2
    flow = torch.randint(-20, 20, (B, fh, fw, 2), generator=g, device=dev).float()
    if fmode == "zero": flow = flow * 0
    if fmode == "const": flow = flow * 0 + 7.0
    if fmode == "none": flow = None
    ab = torch.zeros(256, dtype=torch.float64, device=dev)
    off, msk = ops.conv3x3_dcn_head(x, w, b, 8, flow, scale, ab)
    out[(B, Cin, H, W, scale, fmode, "off")] = off.cpu(); out[(B, Cin, H, W, scale, fmode, "msk")] = msk.cpu()
    continue
    w3 = torch.randn((3, 32, 3, 3), generator=g, device=dev) * 0.05
    x3 = torch.randn((B, 32, H, W), generator=g, device=dev).contiguous(memory_format=torch.channels_last)
    out[(B, Cin, H, W, "nchw3")] = ops.conv3x3(x3, w3, None, out_mode="nchw").cpu()
torch.save(out, sys.argv[1])
''' % REPO
import torch
res = {}
for wide in ("0", "1"):
    f = f"/tmp/dbg_wide_{wide}.pt"
    subprocess.check_call([sys.executable, "-c", CODE, f], env=dict(os.environ, C2M_HEAD_WIDE=wide))
    res[wide] = torch.load(f)
for k, wide in [(k, w) for w in ("1",) for k in res["0"]]:
    a, b = res["0"][k], res[wide][k]
    d = (a.double() - b.double()).abs()
    print("wide", wide, k, "max diff", float(d.max()), "n diff", int((d > 0).sum()), "of", d.numel())
    if d.numel() > 1 and float(d.max()) > 0:
        idx = (d > 0).nonzero()
        print("   first mismatches", idx[:6].tolist(), "channels with mismatch", sorted(set(idx[:, 1].tolist()))[:40])
        print("   ys", sorted(set(idx[:, 2].tolist())), "xs", sorted(set(idx[:, 3].tolist())))
        for i in idx[:12].tolist():
            bb, c, y, x = i
            print("   ", i, "old", float(a[bb, c, y, x]), "new", float(b[bb, c, y, x]), "old nbrs c-1/c+1", float(a[bb, c - 1, y, x]), float(a[bb, c + 1, y, x]),
                  "old x-16", float(a[bb, c, y, x - 16]), "old raw diff", float(b[bb, c, y, x] - a[bb, c, y, x]))
