#!/usr/bin/env python3
"""Diagnostic: where do the reference features of bench.py's configs[2] input repeat?  Prints, for sample 0, the kernel's
skip table, the pixel columns that equal their left neighbour in every row/channel and the pixel rows that equal the row
above over the full width."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "c2-matching_amd"))
import torch
import bench
from c2m_amd import ops
import torch.nn.functional as F

dev = torch.device("cuda:0")
h = int(sys.argv[1]) if len(sys.argv) > 1 else 160
ext, mp, net = bench.build_models(dev)
lq, up, ref = bench.synth_images(2, h, dev, 1234)
with torch.no_grad():
    feats = ext(up, ref)
    n1 = F.normalize(feats["dense_features1"], dim=1)
    n2 = F.normalize(feats["dense_features2"], dim=1)
    idx, val, tab = ops.feature_match_index_batched(n1, n2, 3, 1, 1, True, True, return_skip=True)
print("skip table sample 0:", tab[0].tolist())
f = n2[0].view(torch.int32)
coleq = (f[:, :, 1:] == f[:, :, :-1]).all(0).all(0).cpu().numpy()
roweq = (f[:, 1:, :] == f[:, :-1, :]).all(0).all(1).cpu().numpy()
print("columns equal to their left neighbour:", [i + 1 for i, e in enumerate(coleq) if e])
print("rows equal to the row above:", [i + 1 for i, e in enumerate(roweq) if e])
