#!/usr/bin/env python3
"""What stock MIOpen gives the decoder's 3x3 convolutions (fp32), NCHW vs channels_last: the number a hand-written conv has to beat."""
import json
import time

import torch
import torch.nn.functional as F

dev = torch.device("cuda:0")
res = []
for (ci, co, hh) in ((64, 64, 160), (64, 64, 320), (64, 64, 640), (320, 256, 160), (64, 216, 640), (128, 64, 640), (64, 256, 320)):
    for fmt in ("nchw", "nhwc"):
        for bench in (False, True):
            torch.backends.cudnn.benchmark = bench
            x = torch.randn(16, ci, hh, hh, device=dev)
            w = torch.randn(co, ci, 3, 3, device=dev) * 0.02
            b = torch.zeros(co, device=dev)
            if fmt == "nhwc":
                x = x.contiguous(memory_format=torch.channels_last)
                w = w.contiguous(memory_format=torch.channels_last)
            for _ in range(3):
                y = F.conv2d(x, w, b, padding=1)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            n = 10
            for _ in range(n):
                y = F.conv2d(x, w, b, padding=1)
            torch.cuda.synchronize()
            ms = (time.perf_counter() - t0) / n * 1e3
            fl = 16 * 2.0 * ci * co * 9 * hh * hh
            res.append({"cin": ci, "cout": co, "hw": hh, "fmt": fmt, "benchmark": bench, "ms": round(ms, 3), "tflops": round(fl / ms / 1e9, 1)})
            print(res[-1], flush=True)
            del x, w, y
print(json.dumps(res))
