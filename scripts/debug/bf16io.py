"""Debug aid: bf16-tensor convolutions (c2m_conv3x3_desc.io_flags) against float64, error maps per case."""
import os, sys, traceback
import torch, torch.nn.functional as F
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(R, "c2-matching_amd"))
from c2m_amd import ops
dev = torch.device("cuda:0")
bf = torch.bfloat16
torch.manual_seed(1)
def cl(t): return t.contiguous(memory_format=torch.channels_last)
for (B, C, H, W) in ((1, 16, 8, 32), (1, 32, 8, 32), (1, 64, 8, 32), (2, 64, 20, 48)):
    x = cl(torch.randn(B, C, H, W, device=dev)); w = torch.randn(C, C, 3, 3, device=dev) / 24; b = torch.randn(C, device=dev)
    r1 = cl(torch.randn(B, C, H, W, device=dev))
    xh = cl(x.to(bf)); r1h = cl(r1.to(bf))
    wb = w.bfloat16().double()
    for name, src, a1, od in (("src16", xh, None, None), ("out16", x, None, bf), ("res16", x, r1h, None), ("all16", xh, r1h, bf)):
        try:
            got = ops.conv3x3(src, w, b, res1=a1, algo="bf16", out_dtype=od)
            torch.cuda.synchronize()
            want = F.conv2d(src.double(), wb, b.double(), padding=1) if src is xh else F.conv2d(x.bfloat16().double(), wb, b.double(), padding=1)
            if a1 is not None: want = want + a1.double()
            err = (got.double() - want).abs()
            bad = err > (want.abs() * 2.0 ** -8 + 1e-4)
            print((B, C, H, W), name, "max err %.3e" % float(err.max()), "bad", int(bad.sum()), "of", bad.numel(), flush=True)
            if int(bad.sum()):
                idx = bad.nonzero()
                print("   first bad (b,c,y,x):", idx[:6].tolist(), " bad channels:", sorted(set(idx[:, 1].tolist()))[:20],
                      " bad rows:", sorted(set(idx[:, 2].tolist()))[:12], " bad cols:", sorted(set(idx[:, 3].tolist()))[:40])
                print("   got", got[tuple(idx[0].tolist())].item(), "want", want[tuple(idx[0].tolist())].item())
        except Exception:
            traceback.print_exc()
