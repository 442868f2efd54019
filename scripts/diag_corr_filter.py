#!/usr/bin/env python3
"""A/B of the two implementations of the correlation's MFMA path on one GPU: pre-filter + exact re-score vs exact sweep.
Prints, per case: equality of index maps / values, the pre-filter's flag, the histogram of candidates per query, kernel
times of both (c2m_profile_*: filter sweep, re-score, exact sweep) and the end-to-end time of the call."""
import json
import os
import sys
import time

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(REPO, "c2-matching_amd"), REPO]
import c2m_amd  # noqa: E402
from c2m_amd import ops, _lib  # noqa: E402
import bench  # noqa: E402


def run(fi, fr, mode, iters):
    with ops.corr_filter_mode(mode), ops.record_corr_skip_table():
        idx, val = ops.feature_match_index_batched(fi, fr, 3, 1, 1, True, True)
        tab = ops.last_corr_filter_tables()
        torch.cuda.synchronize()
        _lib.profile_enable(True)
        t0 = time.perf_counter()
        for _ in range(iters):
            ops.feature_match_index_batched(fi, fr, 3, 1, 1, True, True)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / iters * 1e3
        prof = _lib.profile_collect(4096)
        _lib.profile_enable(False)
    k = {}
    for name, ms in prof:
        k.setdefault(name, []).append(ms)
    return idx, val, tab, dt, {n: round(sum(v) / len(v), 3) for n, v in k.items()}


def case(name, fi, fr, iters=5):
    fi, fr = ops.feature_normalize(fi), ops.feature_normalize(fr)
    i1, v1, tab, t1, k1 = run(fi, fr, 1, iters)
    i0, v0, _, t0, k0 = run(fi, fr, 0, iters)
    cnt = tab["cnt"].cpu().numpy().ravel()
    hist = {str(c): int(n) for c, n in zip(*np.unique(np.clip(cnt, -1, 9), return_counts=True))}
    scan = int((tab["cand"] >= 0x40000000).sum())
    print(json.dumps({"case": name, "idx_equal": bool(torch.equal(i1, i0)), "idx_mismatches": int((i1 != i0).sum()),
                      "val_equal": bool(torch.equal(v1, v0)), "flag": int(tab["flags"][0]), "cnt_hist": hist, "lane_scans": scan,
                      "ms_call_filter": round(t1, 3), "ms_call_exact": round(t0, 3), "kernels_filter": k1, "kernels_exact": k0}),
          flush=True)


def main():
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(3)
    for (B, C, h) in ((2, 64, 40), (2, 256, 40), (1, 128, 90)):
        case(f"randn B{B} C{C} {h}x{h}", torch.randn((B, C, h, h), generator=g, device=dev), torch.randn((B, C, h, h), generator=g, device=dev))
    fi, fr = bench.synth_features(16, 256, 160, 125, dev, 1234)
    case("configs[1] synthetic features B16 C256 160x160 (band beyond 125)", fi, fr, iters=3)
    ext, mp, net = bench.build_models(dev)
    lq, up, ref = bench.synth_images(16, 160, dev, 1234)
    with torch.no_grad():
        f = ext(up, ref)
    case("configs[2] extractor features B16 C256 160x160", f["dense_features1"].float(), f["dense_features2"].float(), iters=3)
    if "--lr320" in sys.argv:
        lq, up, ref = bench.synth_images(4, 320, dev, 1234)
        with torch.no_grad():
            f = ext(up, ref)
        case("configs[4] extractor features B4 C256 320x320", f["dense_features1"].float(), f["dense_features2"].float(), iters=2)


if __name__ == "__main__":
    main()
