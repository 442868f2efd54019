#!/usr/bin/env python3
"""The pair stream (c2m_conv3x3_pair_nhwc_f32: a residual block's two convolutions as one tile stream per XCD) against two launches of
the split kernel: bit identity, then timings.  usage: diag_pair.py [check] [time] [soak]"""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "c2-matching_amd"))
import torch
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
    return ops.conv3x3(t, w2, b2, res1=x, res2=res2, algo="split16"), t


def check():
    ok = True
    for (B, H, W, with_res2) in [(8, 330, 64, False), (8, 200, 210, True), (16, 101, 250, True), (8, 336, 33, False), (16, 160, 160, False),
                                 (8, 330, 210, True), (24, 97, 130, False), (16, 320, 320, False), (16, 640, 640, True)]:
        if not c2m_amd._lib.lib().c2m_conv3x3_pair_supported(B, 64, H, W):
            print({"shape": (B, H, W), "skipped": "too small for the safe lag"}); continue
        x, w1, b1, w2, b2 = make(B, H, W, 100 + H + W)
        r2 = torch.randn_like(x) if with_res2 else None
        t = torch.full_like(x, float("nan"))
        got = ops.conv3x3_pair(x, w1, b1, w2, b2, res2=r2, t=t, check=True)
        want, tw = two_launch(x, w1, b1, w2, b2, r2)
        same, same_t = bool(torch.equal(got, want)), bool(torch.equal(t, tw))
        ok = ok and same and same_t
        print({"shape": (B, H, W), "res2": with_res2, "out_bit_identical": same, "t_bit_identical": same_t,
               "max_diff": float((got - want).abs().max())}, flush=True)
        if not same:
            dd = (got - want).abs()
            idx = torch.nonzero(dd > 0)
            print("  differing (b, c, y, x):", idx[:6].tolist(), "count", idx.shape[0], "samples", sorted(set(idx[:, 0].tolist())),
                  "rows", sorted(set(idx[:, 2].tolist()))[:24], flush=True)
    print("ALL OK" if ok else "FAILED")


def soak(n=30):
    x, w1, b1, w2, b2 = make(16, 640, 640, 7)
    want, _ = two_launch(x, w1, b1, w2, b2)
    bad = 0
    t = torch.empty_like(x)
    for i in range(n):
        t.fill_(float("nan"))
        got = ops.conv3x3_pair(x, w1, b1, w2, b2, t=t, check=True)
        bad += 0 if torch.equal(got, want) else 1
    print({"soak_runs": n, "mismatching_runs": bad})


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


def time_():
    for hw in (640, 320, 160):
        x, w1, b1, w2, b2 = make(16, hw, hw, 7)
        out, t = torch.empty_like(x), torch.empty_like(x)
        f = timed(lambda: ops.conv3x3_pair(x, w1, b1, w2, b2, out=out, t=t))
        tl = timed(lambda: two_launch(x, w1, b1, w2, b2))
        print({"hw": hw, "pair_ms": round(f, 4), "two_launch_ms": round(tl, 4), "ratio": round(f / tl, 3)}, flush=True)


if __name__ == "__main__":
    what = sys.argv[1:] or ["check", "time"]
    with ops.conv_flavour("f16x2"):
        if "check" in what:
            check()
        if "soak" in what:
            soak()
        if "time" in what:
            time_()
