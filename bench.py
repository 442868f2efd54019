#!/usr/bin/env python3
"""bench.py -- LR-Ref image pairs / second of the MI355X hot path on synthetic 160x160-LR / 500x500-Ref pairs.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload corr|restore]

One "step" = one pass of the hot path over one batch of B=16 pairs per GPU (BASELINE.json configs[1]:
"Batch-16 160x160 LR / 500x500 Ref, correlation+index-map only"): channel-normalise both 256x160x160 feature maps,
3x3-patch correlation + arg-max index map, pre-offset maps at the three scales.  Inputs are resident in HBM before the
timed region.  N>1: launched by torch.distributed.run, one rank per GPU, every rank its own batch (weak scaling, no
data-path collective: pairs are independent -- SURVEY.md 8e); the barrier / max-over-ranks timing uses RCCL.

Prints ONE JSON line (rank 0) with the contract fields plus `roofline` (dominant kernel, timed with HIP events on the
launch stream through c2m_profile_*) and `cpu_baseline` (the reference's algorithm on PyTorch-CPU, bounded sample).
"""
import argparse
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
for _p in (os.path.join(REPO, "c2-matching_amd"),):
    if _p not in sys.path:
        sys.path.insert(0, _p)

import torch  # noqa: E402

FP32_MATRIX_PEAK_TFLOPS = 157.3  # /opt/skills/guides/MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense
HBM_PEAK_GBS = 8000.0


def synth_features(B, C, h, valid, dev, seed):
    """N(0,1) features; the ref map carries the constant band a zero-padded 500x500 Ref leaves beyond 125/160
    (ref_cufed_dataset.py:107-114) -> exact ties exist, as in the reference's test-time data."""
    g = torch.Generator(device=dev).manual_seed(seed)
    fin = torch.randn((B, C, h, h), generator=g, device=dev, dtype=torch.float32)
    fref = torch.randn((B, C, h, h), generator=g, device=dev, dtype=torch.float32)
    if valid < h:
        const = fref[:, :, valid:valid + 1, valid:valid + 1].clone()
        fref[:, :, valid:, :] = const
        fref[:, :, :, valid:] = const
    return fin, fref


def corr_step(ops, fin, fref, h):
    n1 = ops.feature_normalize(fin)
    n2 = ops.feature_normalize(fref)
    idx, val = ops.feature_match_index_batched(n1, n2, 3, 1, 1, True, True)
    offs = ops.build_pre_offsets(idx, h, h)
    return idx, val, offs


def corr_executed_flops(B, C, h):
    """MFMA flops the sweep actually issues (tiles incl. halo / quantisation), per launch."""
    tiles = ((h - 2 + 13) // 14) ** 2
    steps = ((h - 2 + 27) // 28) * h
    return B * tiles * (steps + 1) * 8 * (C // 2) * (2 * 32 * 32 * 2)


def cpu_baseline(h, C, threads=None, budget_s=12.0):
    """Reference algorithm (ref patches as conv2d filters, chunked running arg-max -- ref_map_util.py:26-86 as restated in
    oracle/torch_port.py) on PyTorch-CPU.  Bounded sample: `rows` query pixel rows of ONE h x h pair against the full ref map
    (cost is linear in query rows, pairs/s = (rows-2)/(h-2) / seconds).  A 48-row probe sizes the sample so that it runs
    for roughly `budget_s` seconds (at most the whole pair)."""
    sys.path.insert(0, os.path.join(REPO, "oracle"))
    import torch_port
    import c2m_oracle
    threads = threads or os.cpu_count()
    torch.set_num_threads(threads)
    g = torch.Generator().manual_seed(1234)
    fi_full = torch.nn.functional.normalize(torch.randn((C, h, h), generator=g), dim=0)
    fr = torch.nn.functional.normalize(torch.randn((C, h, h), generator=g), dim=0)
    torch_port.feature_match_index_conv(fi_full[:, :6], fr[:, :12], 3, 1, 1, True, True)  # warm-up

    def run(rows):
        t0 = time.perf_counter()
        torch_port.feature_match_index_conv(fi_full[:, :rows].contiguous(), fr, 3, 1, 1, True, True)
        return time.perf_counter() - t0

    rows = min(48, h)
    dt = run(rows)
    want = int(min(h, max(rows, 2 + (rows - 2) * budget_s / max(dt, 1e-3))))
    if want > rows + 8:
        rows = want
        dt = run(rows)
    frac = (rows - 2) / (h - 2)
    # one host thread, 8 query rows (SURVEY.md 8d asks for a 1-thread figure beside the all-core one)
    torch.set_num_threads(1)
    t0 = time.perf_counter()
    torch_port.feature_match_index_conv(fi_full[:, :8].contiguous(), fr, 3, 1, 1, True, True)
    dt_1 = time.perf_counter() - t0
    torch.set_num_threads(threads)
    # the C oracle (pixel-level restructuring, OpenMP) on a 48-row sample, for context
    c2m_oracle.set_num_threads(min(threads, 32))
    t0 = time.perf_counter()
    c2m_oracle.feature_match_index(fi_full[:, :48].contiguous().numpy(), fr.numpy(), 3, 1, 1, True, True)
    dt_c = time.perf_counter() - t0
    return {"value": frac / dt, "unit": "pairs/s", "cores": threads, "kind": "port",
            "sample": f"{rows} of {h} query rows of one {h}x{h}x{C} pair vs full ref map, PyTorch-CPU conv2d+max "
                      f"(reference algorithm, oracle/torch_port.py), {dt:.2f}s; linear extrapolation to a pair",
            "one_thread_pairs_per_s": (6 / (h - 2)) / dt_1,
            "oracle_c_openmp_pairs_per_s": (46 / (h - 2)) / dt_c, "oracle_c_threads": min(threads, 32)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=16, help="pairs per GPU per step")
    ap.add_argument("--lr", type=int, default=160, help="LR size = feature-map size")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} needs torch.distributed.run with {args.gpus} ranks (WORLD_SIZE={world})")
    assert torch.cuda.is_available(), "bench.py needs an MI355X"
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group("nccl", device_id=dev)  # RCCL over xGMI; used for the barrier + max only

    import c2m_amd
    ops = c2m_amd.ops
    B, C, h = args.batch, 256, args.lr
    valid = (500 * h) // 640  # 500x500 Ref inside the 640x640 padded canvas, at feature scale
    fin, fref = synth_features(B, C, h, valid, dev, 1234 + rank)

    def sync():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        out = corr_step(ops, fin, fref, h)
    c2m_amd.profile_enable(True)
    c2m_amd.profile_collect()
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = corr_step(ops, fin, fref, h)
    sync()
    dt = time.perf_counter() - t0
    kern = [ms for (name, ms) in c2m_amd.profile_collect() if name == "corr_argmax_mfma"]
    c2m_amd.profile_enable(False)
    if dist is not None:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    assert int(out[0].min()) >= 0 and int(out[0].max()) < (h - 2) ** 2

    if rank == 0:
        pairs = B * world * args.steps
        kms = sum(kern) / max(len(kern), 1)
        exec_flops = corr_executed_flops(B, C, h)
        algo_flops = B * 2.0 * ((h - 2) ** 2) ** 2 * C * 9      # 2*Nq*Nr*C*9 per pair
        achieved = exec_flops / (kms * 1e-3) / 1e12 if kms > 0 else 0.0
        traffic = None
        tfile = os.path.join(REPO, "profiles", "corr_pmc_traffic.json")
        if os.path.exists(tfile) and (B, h) == (16, 160):
            traffic = json.load(open(tfile)).get("hbm_bytes_per_launch")
        line = {
            "metric": "LR-Ref image pairs/sec (160x160 LR, 500x500 Ref, 4x SR)",
            "value": pairs / dt, "unit": "pairs/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"configs[1]: batch-{B} {h}x{h} LR / 500x500 Ref (zero-padded to {4*h}), "
                                   "feature normalise + 3x3 correlation/arg-max index map + pre-offset maps, "
                                   f"{B} pairs per GPU per step", "feature_channels": C, "parallelism": f"dp{world} (batch-sharded, no collective)"},
            "roofline": {"bound": "mfma", "kernel": "corr_argmax_mfma_kernel<256>", "achieved": achieved,
                         "peak": FP32_MATRIX_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": achieved / FP32_MATRIX_PEAK_TFLOPS,
                         "traffic": traffic, "kernel_ms": kms, "launches_timed": len(kern),
                         "flops_counted": "executed fp32 MFMA flops per launch (pixel-level restructuring incl. halo)",
                         "executed_flops_per_launch": exec_flops,
                         "algorithmic_flops_per_launch": algo_flops,
                         "algorithmic_equiv_tflops": algo_flops / (kms * 1e-3) / 1e12 if kms > 0 else 0.0},
        }
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(h, C)
        print(json.dumps(line))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
