#!/usr/bin/env python3
"""Fused residual block (csrc/conv3x3_resblock.hip) against the two-launch path and float64, then timings.
usage: diag_resblock.py [check] [time]"""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "c2-matching_amd"))
import torch
import torch.nn.functional as F
import c2m_amd
from c2m_amd import ops

dev = torch.device("cuda:0")


def make(B, H, W, seed, wscale=0.04):
    g = torch.Generator(device=dev).manual_seed(seed)
    x = torch.randn((B, 64, H, W), generator=g, device=dev).contiguous(memory_format=torch.channels_last)
    w1 = torch.randn((64, 64, 3, 3), generator=g, device=dev) * wscale
    w2 = torch.randn((64, 64, 3, 3), generator=g, device=dev) * wscale
    b1 = torch.randn((64,), generator=g, device=dev) * 0.1
    b2 = torch.randn((64,), generator=g, device=dev) * 0.1
    return x, w1, b1, w2, b2


def two_launch(x, w1, b1, w2, b2, res2=None):
    t = ops.conv3x3(x, w1, b1, act=ops.ACT_RELU, algo="split16")
    return ops.conv3x3(t, w2, b2, res1=x, res2=res2, algo="split16")


def ref64(x, w1, b1, w2, b2, res2=None):
    xd = x.double()
    t = F.relu(F.conv2d(xd, w1.double(), b1.double(), padding=1))
    y = xd + F.conv2d(t, w2.double(), b2.double(), padding=1)
    return y if res2 is None else y + res2.double()


def check():
    ok = True
    for (B, H, W, with_res2) in [(1, 8, 30, False), (1, 6, 30, False), (1, 16, 30, False), (1, 24, 32, True), (2, 19, 61, False), (1, 5, 7, True),
                                 (3, 40, 95, True), (1, 160, 160, False), (2, 330, 210, True), (16, 64, 64, False)]:
        x, w1, b1, w2, b2 = make(B, H, W, 100 + H + W)
        r2 = torch.randn_like(x) if with_res2 else None
        got = ops.resblock3x3(x, w1, b1, w2, b2, res2=r2)
        torch.cuda.synchronize()
        want = ref64(x, w1, b1, w2, b2, r2)
        two = two_launch(x, w1, b1, w2, b2, r2)
        scale = float(want.abs().max())
        e_f = float((got.double() - want).abs().max())
        e_2 = float((two.double() - want).abs().max())
        d = float((got - two).abs().max())
        bad = not (e_f < 1e-5 * scale and e_f <= 1.5 * e_2 + 1e-6 * scale)
        ok = ok and not bad
        print({"shape": (B, H, W), "res2": with_res2, "err_fused_vs_f64": e_f, "err_two_launch_vs_f64": e_2, "fused_vs_two": d, "scale": scale,
               "finite": bool(torch.isfinite(got).all()), "BAD": bad}, flush=True)
        if bad:
            dd = (got.double() - want).abs()
            idx = torch.nonzero(dd > 1e-5 * scale)
            print("  first bad (b, c, y, x):", idx[:6].tolist(), "count", idx.shape[0], "rows", sorted(set(idx[:, 2].tolist()))[:20],
                  "cols", sorted(set(idx[:, 3].tolist()))[:40], flush=True)
    a = ops.resblock3x3(*make(2, 50, 70, 5))
    b = ops.resblock3x3(*make(2, 50, 70, 5))
    print({"repeat_bit_identical": bool(torch.equal(a, b))})
    print("ALL OK" if ok and torch.equal(a, b) else "FAILED")


def timed(fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    c2m_amd.profile_enable(True); c2m_amd.profile_collect()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    ms = [t for (n, t) in c2m_amd.profile_collect() if n.startswith("conv3x3")]
    c2m_amd.profile_enable(False)
    return sum(ms) / iters


def time_(sizes=(640, 320, 160)):
    for hw in sizes:
        x, w1, b1, w2, b2 = make(16, hw, hw, 7)
        out = torch.empty_like(x)
        f = timed(lambda: ops.resblock3x3(x, w1, b1, w2, b2, out=out))
        t = timed(lambda: two_launch(x, w1, b1, w2, b2))
        fl = 2 * 2.0 * 64 * 9 * 64 * hw * hw * 16
        print({"hw": hw, "fused_ms": round(f, 4), "two_launch_ms": round(t, 4), "ratio": round(f / t, 3), "fused_alg_tflops": round(fl / f / 1e9, 1)}, flush=True)


if __name__ == "__main__":
    what = sys.argv[1:] or ["check", "time"]
    with ops.conv_flavour("f16x2"):
        if "check" in what:
            check()
        if "time" in what:
            time_()
        if "time640" in what:
            time_((640,))
