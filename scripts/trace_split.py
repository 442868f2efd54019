#!/usr/bin/env python3
"""Timeline of conv3x3_split_kernel (f16 x 2, 64 -> 64) from the -DC2M_SPLIT_TRACE build (csrc/conv3x3_split.hip, ABL & 1024): every wave
stamps s_memtime just before and just after each unit-end barrier and after the epilogue, for two consecutive tiles.
  C2M_LIB=build_exp/trace/libc2m_hip.so C2M_SPLIT_ABL=1024 python scripts/trace_split.py [H] [B]
Prints, in shader cycles (means over all waves of all workgroups): per unit of a tile the time from the previous barrier's release to this
wave's arrival at the next (its own work: 36 MFMAs = 1 152 pipe cycles, operand reads, split rounds, loads) and the time it then waits
in the barrier; the epilogue; the tile period; and, for one CU, the interleaved event list of its two workgroups."""
import os, struct, sys, tempfile
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "c2-matching_amd"))
os.environ.setdefault("C2M_SPLIT_ABL", "1024")
import numpy as np
import torch
from c2m_amd import ops

H = int(sys.argv[1]) if len(sys.argv) > 1 else 640
B = int(sys.argv[2]) if len(sys.argv) > 2 else 16
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(1)
x = torch.randn((B, 64, H, H), generator=g, device=dev).contiguous(memory_format=torch.channels_last)
w = torch.randn((64, 64, 3, 3), generator=g, device=dev) * 0.04
b = torch.randn((64,), generator=g, device=dev) * 0.1
r = torch.randn_like(x)


def run(with_res):
    for _ in range(3):
        y = ops.conv3x3(x, w, b, act=ops.ACT_NONE if with_res else ops.ACT_RELU, res1=r if with_res else None, algo="split16")
    torch.cuda.synchronize()
    fn = tempfile.mktemp(suffix=".trace")
    os.environ["C2M_SPLIT_TRACE_FILE"] = fn
    y = ops.conv3x3(x, w, b, act=ops.ACT_NONE if with_res else ops.ACT_RELU, res1=r if with_res else None, algo="split16")
    torch.cuda.synchronize()
    del os.environ["C2M_SPLIT_TRACE_FILE"]
    raw = open(fn, "rb").read()
    os.unlink(fn)
    grid, tpw, res, hh = struct.unpack("4i", raw[:16])
    t = np.frombuffer(raw[16:16 + grid * 4 * 64 * 4], dtype=np.uint32).reshape(grid, 4, 64).astype(np.int64)
    return grid, tpw, t, y


def analyse(name, grid, tpw, t):
    hw = t[:, 0, 63]
    xcc = t[:, 0, 62] & 15
    cu = (hw >> 8) & 15
    sh = (hw >> 12) & 1
    se = (hw >> 13) & 7
    key = xcc * 1000 + se * 100 + sh * 20 + cu
    ev = t[:, :, :52].reshape(grid, 4, 2, 26)          # [wg][wave][tile][event]
    ok = (ev[:, :, :, :25] > 0).all(axis=(1, 2, 3))
    ev = ev[ok]
    print(f"=== {name}: grid {grid}, tiles per workgroup {tpw}, traced workgroups {int(ok.sum())}, distinct CUs {len(set(key.tolist()))}")
    A = ev[..., 0:24:2]      # before barrier of unit u
    Bf = ev[..., 1:24:2]     # after
    E = ev[..., 24]
    d32 = lambda a: (a + (1 << 32)) % (1 << 32)       # 32-bit wrap
    # tile 1 has a previous event (tile 0's E); use tile 1 for "start"
    work = np.empty(A.shape[:-1] + (12,), dtype=np.int64)
    work[..., 1:] = d32(A[..., 1:] - Bf[..., :-1])
    work[:, :, 1, 0] = d32(A[:, :, 1, 0] - E[:, :, 0])
    work[:, :, 0, 0] = work[:, :, 1, 0]
    wait = d32(Bf - A)
    epi = d32(E - Bf[..., 11])
    period = d32(E[:, :, 1] - E[:, :, 0])
    print("unit:            " + " ".join(f"{u:6d}" for u in range(12)) + "   | epilogue   tile period")
    print("own work (mean): " + " ".join(f"{v:6.0f}" for v in work[:, :, 1].mean(axis=(0, 1))) + f"   | {epi.mean():8.0f}   {period.mean():8.0f}")
    print("barrier wait:    " + " ".join(f"{v:6.0f}" for v in wait[:, :, 1].mean(axis=(0, 1))))
    print("own work (p90):  " + " ".join(f"{v:6.0f}" for v in np.percentile(work[:, :, 1], 90, axis=(0, 1))) + f"   | {np.percentile(epi, 90):8.0f}   {np.percentile(period, 90):8.0f}")
    tot_work, tot_wait = work[:, :, 1].sum(axis=-1).mean(), wait[:, :, 1].sum(axis=-1).mean()
    print(f"per tile and wave: own work {tot_work:.0f} + barrier waits {tot_wait:.0f} + epilogue {epi[:, :, 1].mean():.0f} = {tot_work + tot_wait + epi[:, :, 1].mean():.0f}"
          f"  (12 units x 36 MFMAs x 32 cycles = 13 824 pipe cycles per wave; two waves share a pipe)")
    # per-wave spread inside a workgroup: who arrives last at the barriers?
    last = (A[:, :, 1, :] == A[:, :, 1, :].max(axis=1, keepdims=True)).mean(axis=(0, 2))
    print("share of barriers at which wave w arrives last: " + " ".join(f"w{w}: {v:.2f}" for w, v in enumerate(last)))
    # one CU's two workgroups, interleaved
    keys = key[ok]
    for k in sorted(set(keys.tolist())):
        idx = np.nonzero(keys == k)[0]
        if len(idx) >= 2:
            a, bq = idx[0], idx[1]
            t0 = min(ev[a, 0, 0, 0], ev[bq, 0, 0, 0])
            rows = []
            for which, i in (("A", a), ("B", bq)):
                for tl in range(2):
                    for u in range(12):
                        rows.append((int(d32(ev[i, 0, tl, 2 * u] - t0)), f"{which} tile{tl} unit{u:2d} arrives"))
                        rows.append((int(d32(ev[i, 0, tl, 2 * u + 1] - t0)), f"{which} tile{tl} unit{u:2d} released"))
                    rows.append((int(d32(ev[i, 0, tl, 24] - t0)), f"{which} tile{tl} epilogue done"))
            rows.sort()
            print(f"--- CU key {k}: wave 0 of its two workgroups (cycles from the first event)")
            for tt, s in rows:
                if "released" in s or "epilogue" in s:
                    print(f"   {tt:8d}  {s}")
            break


for with_res in (False, True):
    grid, tpw, t, y = run(with_res)
    analyse("body+res" if with_res else "body (ReLU)", grid, tpw, t)
want = torch.nn.functional.conv2d(x[:1].double(), w.double(), b.double(), padding=1) + r[:1].double()
print("max |err| vs float64 of the traced build (body+res, image 0):", float((y[:1].double() - want).abs().max()))
