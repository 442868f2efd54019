#!/usr/bin/env python3
"""Go / no-go measurement for a FUSED ResidualBlockNoBN (arch_util.py:80-136: x + conv2(relu(conv1(x)))) on the f16 x 2 split
kernel -- VERDICT r5 "next round" item 1: timing-only upper bounds from the compile-time masks that exist ($C2M_SPLIT_ABL,
csrc/conv3x3_split.hip; results of masked runs are WRONG by construction, only their time is read).

One process = one mask (the mask is read once per process).  Per call it times, with HIP events on the launch stream,
  conv1 : relu(conv(x))           at H x W, H*1.0667 x W (30 of 32 MFMA columns valid) and H*1.328 x W (34 x 10 / 32 x 8 halo recompute)
  conv2 : conv(t) + x  and  conv(t)   at H x W and H*1.0667 x W
so that the caller can compose
  today's pair                      = conv1[0](H) + conv2+res[0](H)
  mask upper bound (review's recipe) = conv1[64: one store per tile](H) + conv2[6: no halo loads, no split](H)
  tile-fused estimate                = conv1[64](1.328 H) + conv2[6](H)      (+ the residual, if it cannot come from LDS)
  sliding-window estimate            = conv1[64](1.0667 H) + conv2[6](1.0667 H)
usage: abl_resblock.py [--hw 640] [--batch 16] [--iters 20] [--persample]"""
import argparse
import json
import os
import sys

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "c2-matching_amd"))
import c2m_amd  # noqa: E402

ops = c2m_amd.ops


def timed(fn, iters):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    c2m_amd.profile_enable(True)
    c2m_amd.profile_collect()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    ms = [t for (n, t) in c2m_amd.profile_collect() if n.startswith("conv3x3")]
    c2m_amd.profile_enable(False)
    return sum(ms) / iters


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--hw", type=int, default=640)
    ap.add_argument("--batch", type=int, default=16)
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--persample", action="store_true", help="also: the pair launched per group of 1 / 2 / 4 samples (does the "
                    "intermediate tensor stay in the 256 MB Infinity Cache?)")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    abl = int(os.environ.get("C2M_SPLIT_ABL", "0"))
    B, W = a.batch, a.hw
    w1 = torch.randn(64, 64, 3, 3, device=dev) * 0.02
    w2 = torch.randn(64, 64, 3, 3, device=dev) * 0.02
    b1 = torch.randn(64, device=dev) * 0.1
    res = {"abl": abl, "hw": a.hw, "batch": B}
    with ops.conv_flavour("f16x2"):
        for tag, scale in (("1.000", 1.0), ("1.067", 32.0 / 30.0), ("1.328", 340.0 / 256.0)):
            H = int(round(a.hw * scale / 8.0)) * 8
            x = torch.randn(B, 64, H, W, device=dev).contiguous(memory_format=torch.channels_last)
            t = ops.empty_nhwc(B, 64, H, W, dev)
            y = ops.empty_nhwc(B, 64, H, W, dev)
            res[f"conv1_relu_H{tag}"] = round(timed(lambda: ops.conv3x3(x, w1, b1, act=ops.ACT_RELU, out=t, algo="split16"), a.iters), 4)
            if scale < 1.2:
                res[f"conv2_res_H{tag}"] = round(timed(lambda: ops.conv3x3(t, w2, b1, res1=x, out=y, algo="split16"), a.iters), 4)
                res[f"conv2_nores_H{tag}"] = round(timed(lambda: ops.conv3x3(t, w2, b1, out=y, algo="split16"), a.iters), 4)
            if a.persample and scale == 1.0 and abl == 0:
                for g in (1, 2, 4):
                    def pair():
                        for b0 in range(0, B, g):
                            ops.conv3x3(x[b0:b0 + g], w1, b1, act=ops.ACT_RELU, out=t[b0:b0 + g], algo="split16")
                            ops.conv3x3(t[b0:b0 + g], w2, b1, res1=x[b0:b0 + g], out=y[b0:b0 + g], algo="split16")
                    res[f"pair_groups_of_{g}"] = round(timed(pair, max(4, a.iters // 4)), 4)
            del x, t, y
    print(json.dumps(res), flush=True)


if __name__ == "__main__":
    main()
