#!/usr/bin/env python3
"""Micro-benchmark of the DCNv2 kernels at BASELINE shapes (not the driver's bench: see bench.py).

forward : the three DynAgg layers at LR 160x160, B=16 (configs[2]) -- 30.2 GFLOP per sample per layer
backward: the same layers at LR 40x40, B=4 per GPU (configs[3])
Offsets = patch-match style pre-offsets (piecewise constant, long range) + small learned offsets, as in the model."""
import argparse
import json
import os
import sys

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "c2-matching_amd"))
import c2m_amd  # noqa: E402

ops = c2m_amd.ops
PEAK = 157.3


def make_inputs(B, C, H, dg, dev, seed, lr, flow="random"):
    g = torch.Generator(device=dev).manual_seed(seed)
    x = torch.relu(torch.randn((B, C, H, H), generator=g, device=dev))
    w = torch.randn((C, C, 3, 3), generator=g, device=dev) * 0.02
    b = torch.zeros(C, device=dev)
    hp = lr - 2
    if flow == "random":   # what random-noise images give the matcher: every pixel points somewhere else
        idx = torch.randint(0, hp * hp, (B, hp, hp), generator=g, device=dev)
    else:                  # "blocks": 8x8-pixel regions share one displacement (coherent flow, as on natural images)
        nb = (hp + 7) // 8
        disp = torch.randint(-hp, hp, (B, 2, nb, nb), generator=g, device=dev)
        disp = disp.repeat_interleave(8, 2).repeat_interleave(8, 3)[:, :, :hp, :hp]
        yy, xx = torch.meshgrid(torch.arange(hp, device=dev), torch.arange(hp, device=dev), indexing="ij")
        ty = (yy[None] + disp[:, 0]).clamp(0, hp - 1)
        tx = (xx[None] + disp[:, 1]).clamp(0, hp - 1)
        idx = (ty * hp + tx).contiguous()
    s = H // lr
    (pre,) = ops.build_pre_offsets(idx, lr, lr, scales=(s,))
    raw = torch.randn((B, 3 * dg * 9, H, H), generator=g, device=dev) * 0.5
    off, msk = ops.dcn_fuse_offsets(raw, pre, dg, 9)
    return x, w, b, off, msk


def timed(fn, iters, name):
    fn()
    torch.cuda.synchronize()
    c2m_amd.profile_enable(True)
    c2m_amd.profile_collect()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    recs = c2m_amd.profile_collect()
    c2m_amd.profile_enable(False)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    out = {"call_ms": [e0.elapsed_time(e1) / iters]}   # whole operator call: staging copy, weight re-layout, kernel
    for k, ms in recs:
        out.setdefault(k, []).append(ms)
    return {k: sum(v) / len(v) for k, v in out.items()}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=3)
    ap.add_argument("--batch", type=int, default=16)
    ap.add_argument("--lr", type=int, default=160)
    ap.add_argument("--bwd-batch", type=int, default=4)
    ap.add_argument("--bwd-lr", type=int, default=40)
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    res = {"forward": [], "backward": []}
    for flow in ("random", "blocks"):
        for name, C, s in (("small", 256, 1), ("medium", 128, 2), ("large", 64, 4)):
            H = args.lr * s
            x, w, b, off, msk = make_inputs(args.batch, C, H, 8, dev, 1, args.lr, flow)
            t = timed(lambda: ops.dcn_v2_forward(x, w, b, off, msk, 1, 1, 1, 8), args.iters, name)
            flops = 2.0 * C * 9 * C * H * H * args.batch
            ms = t["dcn_v2_forward"]
            tb = timed(lambda: ops.dcn_v2_forward(x, w, b, off, msk, 1, 1, 1, 8, bf16_mma=True), args.iters, name)
            res["forward"].append({"layer": name, "flow": flow, "C": C, "H": H, "B": args.batch, "ms": ms, "call_ms": t["call_ms"],
                                   "bf16_mma_ms": tb["dcn_v2_forward"],
                                   "tflops": flops / ms / 1e9, "frac_fp32_mfma_peak": flops / ms / 1e9 / PEAK})
            del x, w, b, off, msk
            torch.cuda.empty_cache()
    for flow, name, C, s in [(f, *l) for f in ("random", "blocks") for l in (("small", 256, 1), ("medium", 128, 2), ("large", 64, 4))]:
        H = args.bwd_lr * s
        x, w, b, off, msk = make_inputs(args.bwd_batch, C, H, 8, dev, 2, args.bwd_lr, flow)
        go = torch.randn_like(x)
        t = timed(lambda: ops.dcn_v2_backward(x, w, b, off, msk, go, 1, 1, 1, 8), args.iters, name)
        t2 = timed(lambda: ops.dcn_v2_backward(x, w, b, off, msk, go, 1, 1, 1, 8, need_input_grad=False), args.iters, name)
        flops = 2.0 * C * 9 * C * H * H * args.bwd_batch
        res["backward"].append({"layer": name, "flow": flow, "C": C, "H": H, "B": args.bwd_batch,
                                "data_ms": t.get("dcn_v2_backward_data"),
                                "data_ms_no_grad_input": t2.get("dcn_v2_backward_data"), "call_ms_no_grad_input": t2["call_ms"],
                                "weight_ms": t.get("dcn_v2_backward_weight"),
                                "data_tflops": flops / t["dcn_v2_backward_data"] / 1e9,
                                "weight_tflops": flops / t["dcn_v2_backward_weight"] / 1e9})
    print(json.dumps(res))


if __name__ == "__main__":
    main()
