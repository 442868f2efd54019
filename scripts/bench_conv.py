#!/usr/bin/env python3
"""Kernel-level timing of the fused channels-last conv3x3 (csrc/conv3x3.hip) at the decoder's shapes (B=16, LR 160)."""
import argparse
import json
import os
import sys

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "c2-matching_amd"))
import c2m_amd  # noqa: E402

ops = c2m_amd.ops
SHAPES = [  # name, [cin per source], cout, hw, mode
    ("body 64->64 @160", [64], 64, 160, "nhwc"), ("body 64->64 @320", [64], 64, 320, "nhwc"),
    ("body 64->64 @640", [64], 64, 640, "nhwc"), ("body+res 64->64 @640", [64], 64, 640, "nhwc_res"),
    ("body+res 64->64 @160", [64], 64, 160, "nhwc_res"), ("small_offset_conv1 320->256 @160", [64, 256], 256, 160, "nhwc"),
    ("small_offset_conv2 256->256 @160", [256], 256, 160, "nhwc"), ("head_small 320->64 @160", [64, 256], 64, 160, "nhwc"),
    ("tail_small 64->256 ps @160", [64], 256, 160, "pixel_shuffle"), ("medium_offset_conv1 192->128 @320", [64, 128], 128, 320, "nhwc"),
    ("large_offset_conv1 128->64 @640", [64, 64], 64, 640, "nhwc"), ("dcn head 64->216 @640", [64], 216, 640, "head"),
    ("dcn head 256->216 @160", [256], 216, 160, "head"), ("tail_large.0 64->32 @640", [64], 32, 640, "nhwc"),
    ("tail_large.2 32->3 @640", [32], 3, 640, "nchw"),
    ("body 64->64 @1280", [64], 64, 1280, "nhwc"), ("body+res 64->64 @1280", [64], 64, 1280, "nhwc_res"),
]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=16)
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--only", type=str, default="")
    ap.add_argument("--fast", action="store_true", help="fast=True: the decoder's setting (split-bf16 kernel; with C2M_CONV_SPLIT=0 the "
                                                        "Winograd F(4,3) kernel where the map allows)")
    ap.add_argument("--data", type=str, default="randn", help="randn | zeros | ones (power / clock experiments: the matrix pipe draws less on constant data)")
    ap.add_argument("--algo", type=str, default=None, help="force direct / winograd / winograd4 / split / bf16")
    ap.add_argument("--io16", action="store_true", help="bf16 tensors in and out (single-source nhwc shapes; implies --algo bf16)")
    args = ap.parse_args()
    if args.io16:
        args.algo = "bf16"
    dev = torch.device("cuda:0")
    B = args.batch
    out = []
    if args.algo is None:
        ops.conv_flavour("f16x2" if ops._SPLIT16 else "bf16x3").__enter__()   # what the guarded module forwards run (ops._f16x2_auto)
    for name, cins, co, hw, mode in SHAPES:
        if args.only and args.only not in name:
            continue
        if "@1280" in name and (B > 4 or not args.only):
            continue
        if args.io16 and (len(cins) != 1 or not mode.startswith("nhwc")):
            continue
        xs = [torch.randn(B, c, hw, hw, device=dev).contiguous(memory_format=torch.channels_last) for c in cins]
        w = torch.randn(co, sum(cins), 3, 3, device=dev) * 0.02
        b = torch.randn(co, device=dev)
        if args.data != "randn":
            fill = 0.0 if args.data == "zeros" else 1.0
            xs = [x.fill_(fill) for x in xs]
            w.fill_(fill * 0.02)
        if args.io16:
            xs = [x.to(torch.bfloat16).contiguous(memory_format=torch.channels_last) for x in xs]
        od = torch.bfloat16 if args.io16 else None
        flow = torch.zeros(B, hw // (hw // 160) - 2, hw // (hw // 160) - 2, 2, device=dev)

        def run():
            if mode == "head":
                return ops.conv3x3_dcn_head(xs, w, b, 8, flow, hw // 160, algo=args.algo)
            if mode == "nhwc_res":   # second conv of a ResidualBlockNoBN: no activation, + identity
                return ops.conv3x3(xs, w, b, act=ops.ACT_NONE, res1=xs[0], fast=args.fast, algo=args.algo, out_dtype=od)
            return ops.conv3x3(xs, w, b, act=ops.ACT_RELU, out_mode=mode, fast=args.fast, algo=args.algo, out_dtype=od)
        c2m_amd.profile_enable(True)
        c2m_amd.profile_collect()
        try:
            for _ in range(2):
                run()
            torch.cuda.synchronize()
            c2m_amd.profile_collect()
            for _ in range(args.iters):
                run()
        except c2m_amd.C2MError as e:
            print({"layer": name, "skipped": str(e)}, flush=True)
            c2m_amd.profile_collect()
            c2m_amd.profile_enable(False)
            continue
        torch.cuda.synchronize()
        ms = [t for (n, t) in c2m_amd.profile_collect() if n in ("conv3x3_mfma", "conv3x3_split")]
        c2m_amd.profile_enable(False)
        ms = sum(ms) / args.iters    # per call (a DCN head is two launches)
        fl = 2.0 * co * 9 * sum(cins) * hw * hw * B
        out.append({"layer": name, "ms": round(ms, 3), "tflops": round(fl / ms / 1e9, 1), "frac_fp32_mfma_peak": round(fl / ms / 1e9 / 157.3, 3)})
        print(out[-1], flush=True)
        del xs
    print(json.dumps(out))


if __name__ == "__main__":
    main()
