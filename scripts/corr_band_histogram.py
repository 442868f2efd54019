#!/usr/bin/env python3
"""How many ref candidates lie within an error band of the best score, per query?  (measure the 16-bit
pre-filter instead of arguing it.)

For a sample of query patches the full score row against every ref patch is computed in float64 from the extractor features
(the scores of ref_map_util.py:52-76: q . r / (|r| + 1e-5), q and r per-pixel channel-normalised), and the number of
candidates with score >= max - band counted for a ladder of bands.  Runs on the CPU (extractor through stock torch modules);
features: (a) bench.py's synthetic pairs (U(0,1) LR bicubic x4 / U(0,1) 500x500 Ref zero-padded), (b) `_smooth_gt` images
(natural-image-like spectrum) -- both with the bench's seeded random extractor weights (no checkpoints offline).

    python scripts/corr_band_histogram.py [--lr 160] [--queries 1500] [--out profiles/r04_corr_band_histogram.json]
"""
import argparse
import json
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(REPO, "c2-matching_amd"), os.path.join(REPO, "tests", "golden"), REPO):
    if p not in sys.path:
        sys.path.insert(0, p)

BANDS = [0.0, 1e-7, 1e-6, 1e-5, 3e-5, 1e-4, 2.5e-4, 5e-4, 1e-3, 3e-3, 1e-2]


def smooth_gt(B, H, seed):
    import synth
    coarse = torch.from_numpy(synth.uniform((B, 3, H // 16, H // 16), seed, 0.0, 1.0))
    img = F.interpolate(coarse, size=(H, H), mode="bicubic", align_corners=False)
    img = img + 0.03 * torch.from_numpy(synth.gaussish((B, 3, H, H), seed + 1))
    return img.clamp(0, 1)


def features(kind, h, seed):
    import bench
    ext, _, _ = bench.build_models("cpu")
    H = 4 * h
    if kind == "bench":
        _, up, ref = bench.synth_images(1, h, "cpu", seed)
    else:
        gt = smooth_gt(1, H, 6000 + seed)
        lq = F.interpolate(gt, scale_factor=0.25, mode="bicubic", align_corners=False).clamp(0, 1)
        up = F.interpolate(lq, scale_factor=4, mode="bicubic", align_corners=False).clamp(0, 1)
        ref = torch.zeros((1, 3, H, H))
        v = min(H, 500)
        ref[:, :, :v, :v] = smooth_gt(1, 512, 6002 + seed)[:, :, :v, :v]
    with torch.no_grad():
        f = ext(up, ref)
    f1 = F.normalize(f["dense_features1"][0], dim=0).double()
    f2 = F.normalize(f["dense_features2"][0], dim=0).double()
    return f1, f2


def histogram(f1, f2, nq, seed):
    C, h, w = f1.shape
    q_unf = F.unfold(f1[None], 3)[0].T                      # [Nq, C*9]
    r_unf = F.unfold(f2[None], 3)[0]                        # [C*9, Nr]
    r_unf = r_unf / (r_unf.norm(dim=0, keepdim=True) + 1e-5)
    g = np.random.default_rng(seed)
    sel = torch.from_numpy(g.choice(q_unf.shape[0], size=min(nq, q_unf.shape[0]), replace=False))
    qn = q_unf[sel].norm(dim=1)
    counts = {b: [] for b in BANDS}
    gaps = []
    for i0 in range(0, len(sel), 256):
        s = q_unf[sel[i0:i0 + 256]] @ r_unf                # float64 [256, Nr]
        top = s.max(dim=1, keepdim=True).values
        for b in BANDS:
            counts[b].append((s >= top - b).sum(dim=1))
        t2 = torch.topk(s, 2, dim=1).values
        gaps.append(t2[:, 0] - t2[:, 1])
    gaps = torch.cat(gaps)
    out = {"queries": int(len(sel)), "candidates": int(r_unf.shape[1]), "mean_query_patch_norm": float(qn.mean()),
           "score_max_mean": None, "top2_gap_quantiles": {str(q): float(torch.quantile(gaps, q)) for q in (0.001, 0.01, 0.05, 0.5)},
           "bands": {}}
    for b in BANDS:
        c = torch.cat(counts[b]).double()
        out["bands"][f"{b:g}"] = {"mean_in_band": float(c.mean()), "max_in_band": int(c.max()),
                                  "frac_queries_with_2_or_more": float((c >= 2).double().mean()),
                                  "frac_queries_with_3_or_more": float((c >= 3).double().mean()),
                                  "frac_queries_with_more_than_64": float((c > 64).double().mean())}
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--lr", type=int, default=160)
    ap.add_argument("--queries", type=int, default=1500)
    ap.add_argument("--out", default=None)
    a = ap.parse_args()
    torch.manual_seed(0)
    res = {"what": "per-query count of ref candidates with float64 score >= max - band (scores as ref_map_util.py:52-76 on "
                   "channel-normalised extractor features, |q_patch| ~ 3); band of the f16 x 2 pre-filter = 2 * rigorous bound "
                   "~ 8e-5 * |q_patch| ~ 2.5e-4", "lr": a.lr}
    for kind in ("bench", "smooth"):
        f1, f2 = features(kind, a.lr, 1234)
        res[kind] = histogram(f1, f2, a.queries, 7)
        print(kind, json.dumps(res[kind]["bands"], indent=None), flush=True)
        print(kind, "gap quantiles", res[kind]["top2_gap_quantiles"], flush=True)
    if a.out:
        with open(a.out, "w") as fh:
            json.dump(res, fh, indent=1)


if __name__ == "__main__":
    main()
