import os, sys, torch
import torch.nn.functional as F
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "c2-matching_amd"))
import c2m_amd
ops = c2m_amd.ops
dev = torch.device("cuda:0")
torch.manual_seed(0)
for (B, C, Co, H, W) in ((1, 32, 32, 8, 40), (1, 64, 64, 40, 40), (2, 64, 64, 64, 64)):
    x = torch.randn(B, C, H, W, device=dev).contiguous(memory_format=torch.channels_last)
    w = torch.randn(Co, C, 3, 3, device=dev) * 0.05
    b = torch.randn(Co, device=dev)
    # poison LDS-visible state: run a conv on ones first so stale LDS data is non-zero
    ops.conv3x3(torch.ones_like(x) * 100, w, b)
    got = ops.conv3x3(x, w, b)
    want = F.conv2d(x.double(), w.double(), b.double(), padding=1).float()
    err = (got - want).abs()
    print((B, C, Co, H, W), "max err", float(err.max()), "interior max", float(err[:, :, 1:-1, 1:-1].max()))
    e2 = err.amax(dim=(0, 1))
    print("rows with err>1e-3:", (e2.amax(dim=1) > 1e-3).nonzero().flatten().tolist()[:20])
    print("cols with err>1e-3:", (e2.amax(dim=0) > 1e-3).nonzero().flatten().tolist()[:20])
    print("chan with err>1e-3:", (err.amax(dim=(0, 2, 3)) > 1e-3).nonzero().flatten().tolist()[:20])
