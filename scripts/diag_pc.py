#!/usr/bin/env python3
"""The loader / matrix-wave convolution kernel (csrc/conv3x3_pc.hip, $C2M_CONV_PC=1) against float64 and against the fp32-MFMA
direct kernel, on ragged and full-size maps.  Run with C2M_CONV_PC=1 C2M_CONV_PC_MINPIX=0."""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "c2-matching_amd"))
import torch
import torch.nn.functional as F
from c2m_amd import ops

assert os.environ.get("C2M_CONV_PC") == "1", "set C2M_CONV_PC=1 C2M_CONV_PC_MINPIX=0"
dev = "cuda"
def rnd(shape, seed, scale=1.0):
    return torch.randn(shape, generator=torch.Generator(device=dev).manual_seed(seed), device=dev) * scale
def cl(t):
    return t.contiguous(memory_format=torch.channels_last)

ok = True
CASES = [(1, 64, 16, 32, 0, 0), (1, 64, 16, 32, 1, 1), (2, 64, 40, 75, 1, 2), (1, 64, 33, 64, 2, 0), (3, 32, 17, 31, 1, 1), (1, 128, 50, 96, 1, 0),
         (2, 64, 160, 160, 1, 1), (16, 64, 640, 640, 1, 1), (16, 64, 640, 640, 0, 0)]
for (B, Cin, H, W, act, nres) in CASES:
    x = cl(rnd((B, Cin, H, W), 1))
    w = rnd((64, Cin, 3, 3), 2, (Cin * 9) ** -0.5)
    b = rnd((64,), 3)
    res = [cl(rnd((B, 64, H, W), 10 + k)) for k in range(nres)]
    kw = dict(act=act, slope=0.1, res1=res[0] if nres > 0 else None, res2=res[1] if nres > 1 else None)
    with ops.conv_flavour("f16x2"):
        got = ops.conv3x3([x], w, b, algo="split16", **kw)
    ref32 = ops.conv3x3([x], w, b, algo="direct", **kw)
    nb = min(B, 2)
    y = F.conv2d(x[:nb].double(), w.double(), b.double(), padding=1)
    y = y.relu() if act == 1 else (F.leaky_relu(y, 0.1) if act == 2 else y)
    for r in res:
        y = y + r[:nb].double()
    e64 = float((got[:nb].double() - y).abs().max())
    e32 = float((ref32[:nb].double() - y).abs().max())
    d = float((got - ref32).abs().max())
    tol = 1e-5 * max(1.0, float(y.abs().max()))
    good = e64 < tol and d < 2 * tol and bool(torch.isfinite(got).all())
    ok &= good
    print((B, Cin, H, W, act, nres), "err vs fp64 %.3e (direct fp32 kernel %.3e)  max |pc - direct| %.3e  %s" % (e64, e32, d, "OK" if good else "MISMATCH"))
    if not good:
        bad = (got - ref32).abs().amax(dim=1) > 2 * tol
        ys, xs = torch.nonzero(bad[0], as_tuple=True)
        print("   bad pixels per image", bad.flatten(1).sum(1).tolist()[:8], "rows", sorted(set(ys.tolist()))[:24], "cols", sorted(set(xs.tolist()))[:40])
        print("   bad channels", torch.nonzero((got - ref32).abs().amax(dim=(0, 2, 3)) > 2 * tol).flatten().tolist()[:64])
# repeatability on the chip-filling case
x = cl(rnd((16, 64, 640, 640), 5)); w = rnd((64, 64, 3, 3), 6, 1 / 24.0); b = rnd((64,), 7)
with ops.conv_flavour("f16x2"):
    first = ops.conv3x3([x], w, b, algo="split16", act=1)
    for rep in range(3):
        again = ops.conv3x3([x], w, b, algo="split16", act=1)
        same = torch.equal(first, again)
        ok &= same
        print("repeat", rep, "bit-identical" if same else "DIFFERS")
print("ALL OK" if ok else "FAILED")
sys.exit(0 if ok else 1)
