#!/usr/bin/env python3
"""DCNv2 forward through the fused path's entry point (channels-last bordered input, planar offsets) at the three DynAgg
layers of configs[2] (B=16, LR 160): fp32-MFMA GEMM vs the f16 x 2 GEMM.  Inputs as scripts/bench_dcn.py makes them."""
import argparse
import json
import os
import sys

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "c2-matching_amd"))
sys.path.insert(0, os.path.join(REPO, "scripts"))
import c2m_amd  # noqa: E402
import bench_dcn  # noqa: E402

ops = c2m_amd.ops


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=16)
    ap.add_argument("--lr", type=int, default=160)
    ap.add_argument("--iters", type=int, default=5)
    ap.add_argument("--flow", default="random")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    out = []
    for name, C, s in (("small", 256, 1), ("medium", 128, 2), ("large", 64, 4)):
        H = a.lr * s
        x, w, b, off, msk = bench_dcn.make_inputs(a.batch, C, H, 8, dev, 11, a.lr, a.flow)
        bo = ops.BorderedNHWC(x)
        if C == 64:
            bo.grouped8 = bo.buf.view(a.batch, H + 3, H + 3, C // 8, 8).permute(0, 3, 1, 2, 4).contiguous()
        for algo in ("fp32", "f16x2"):
            fn = lambda: ops.dcn_v2_forward_nhwc(bo, w, b, off, msk, 8, act=ops.ACT_LRELU, slope=0.1, algo=algo)  # noqa: E731
            fn()
            torch.cuda.synchronize()
            c2m_amd.profile_enable(True)
            c2m_amd.profile_collect()
            for _ in range(a.iters):
                fn()
            torch.cuda.synchronize()
            ms = [t for (n, t) in c2m_amd.profile_collect() if n == "dcn_v2_forward"]
            c2m_amd.profile_enable(False)
            fl = a.batch * 2.0 * C * 9 * C * H * H
            r = {"layer": name, "algo": algo, "ms": round(sum(ms) / len(ms), 3), "alg_tflops": round(fl / (sum(ms) / len(ms)) / 1e9, 1)}
            out.append(r)
            print(r, flush=True)
        del x, off, msk, bo
    print(json.dumps(out))


if __name__ == "__main__":
    main()
