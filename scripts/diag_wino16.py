"""Diagnostics for csrc/conv3x3_wino16.hip: per-case error against float64 conv2d and WHERE the wrong values sit (row inside
the tile, column inside the tile, cout, sample) -- for bring-up of the kernel; `python scripts/diag_wino16.py [f23]`."""
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "c2-matching_amd"))
import c2m_amd  # noqa: E402

ops = c2m_amd.ops
dev = torch.device("cuda:0")


def rnd(shape, seed, scale=1.0):
    g = torch.Generator(device=dev).manual_seed(seed)
    return torch.randn(shape, generator=g, device=dev) * scale


def run(algo, B, cins, Cout, H, W, act=0, nres=0, R=4):
    xs = [rnd((B, c, H, W), 10 + k).contiguous(memory_format=torch.channels_last) for k, c in enumerate(cins)]
    w = rnd((Cout, sum(cins), 3, 3), 20, 1.0 / np.sqrt(9 * sum(cins)))
    b = rnd((Cout,), 21)
    res = [rnd((B, Cout, H, W), 30 + k).contiguous(memory_format=torch.channels_last) for k in range(nres)]
    want = F.conv2d(torch.cat([x.double() for x in xs], 1), w.double(), b.double(), padding=1)
    if act == 1:
        want = want.clamp_min(0)
    for r in res:
        want = want + r.double()
    got = ops.conv3x3(xs, w, b, act=act, res1=res[0] if nres > 0 else None, res2=res[1] if nres > 1 else None, algo=algo)
    torch.cuda.synchronize()
    err = (got.double() - want).abs()
    scale = max(1.0, float(want.abs().max()))
    bad = err > 1e-5 * scale
    bad = bad | ~torch.isfinite(got)
    line = f"{algo} B{B} {cins}->{Cout} {H}x{W} act{act} res{nres}: max err {float(err.nan_to_num(1e9).max()):.3e} (tol {1e-5 * scale:.1e}) bad {int(bad.sum())}/{bad.numel()}"
    if bad.any():
        idx = bad.nonzero()
        TH = 4 * R
        def hist(v, n):
            return np.bincount(v.cpu().numpy(), minlength=n).tolist()
        line += (f"\n   by sample {hist(idx[:, 0], B)}\n   by cout%64 {hist(idx[:, 1] % 64, 64)}\n   by y%{TH} {hist(idx[:, 2] % TH, TH)}"
                 f"\n   by x%30 {hist(idx[:, 3] % 30, 30)}\n   by tile_y {hist(idx[:, 2] // TH, (H + TH - 1) // TH)}\n   by tile_x {hist(idx[:, 3] // 30, (W + 29) // 30)}"
                 f"\n   first bad: {idx[:5].tolist()} got {[float(got[tuple(i)]) for i in idx[:5]]} want {[float(want[tuple(i)]) for i in idx[:5]]}")
    print(line, flush=True)
    return not bad.any()


if __name__ == "__main__":
    f23 = "f23" in sys.argv[1:]
    only = [int(a) for a in sys.argv[1:] if a.isdigit()]
    algo, R = ("wino16_f23", 2) if f23 else ("wino16", 4)
    ok = True
    for k, case in enumerate([(1, [16], 64, 16, 30), (1, [16], 64, 8, 32), (1, [64], 64, 16, 30), (1, [64], 64, 40, 40), (2, [64], 64, 37, 45, 1, 2),
                 (1, [64, 128], 64, 16, 64, 0, 0), (1, [48], 128, 50, 61, 0, 1), (2, [64], 64, 160, 160, 1, 1), (1, [32], 64, 390, 392, 1, 0)]):
        if only and k not in only:
            continue
        ok = run(algo, *case, R=R) and ok
    print("ALL OK" if ok else "FAILURES")
