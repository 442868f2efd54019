#!/usr/bin/env python3
"""bench.py -- LR-Ref image pairs / second of the MI355X restoration path on synthetic 160x160-LR / 500x500-Ref pairs.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload restore|corr|train] [--lr 160] [--dtype f32|bf16]

Default workload = BASELINE.json configs[2], the configuration the metric is quoted on: "Batch-16 full restoration forward
(correlation + DCNv2 warp + decoder), 1xMI355X, fp32".  One "step" = one pass over one batch of B=16 pairs per GPU:
ContrasExtractorSep on the bicubic-upsampled LR and the zero-padded Ref -> channel normalise -> 3x3 patch correlation /
arg-max index map -> pre-offsets -> VGG19 taps of the Ref -> RestorationNet (content extractor, three DynAgg = DCNv2 warps,
3 x 16 residual blocks, up-sampling tails) -> SR image [16,3,640,640].  Images and weights are resident in HBM before the
timed region; weights are seeded random (no checkpoints offline), the offset heads are live (N(0, 0.01)).

Other workloads: `--workload corr` = configs[1] (normalise + correlation + pre-offsets on synthetic features);
`--workload train` = configs[3] (stage-3 MSE training step, 4 pairs per GPU, GT 160x160: forward, L1 loss, backward incl. the
three DCNv2 backward passes, Adam; with N > 1 net_g's gradients are all-reduced by DDP over RCCL);
`--lr 320 --dtype bf16` = configs[4] (CUFED5-shape inference under bf16 autocast; batch defaults to 4 per GPU).

N > 1: one rank per GPU.  Started as a plain command, bench.py re-executes itself under `python -m torch.distributed.run
--nnodes=1 --nproc-per-node N --master-addr 127.0.0.1` (the reference's launcher logic is mmsr/train.py:24-45); started by
torch.distributed.run it reads RANK / LOCAL_RANK / WORLD_SIZE.  Every rank works on its own batch (weak scaling, no
data-path collective: pairs are independent -- SURVEY.md 8e); barrier / max-over-ranks timing use RCCL.

Prints ONE JSON line (rank 0): the contract fields, `stage_ms`, `roofline` (the hand-written kernel family with the largest
share of the step) + `roofline_kernels` (every hand-written hot kernel: HIP events on the launch stream through
c2m_profile_*; `frac` = EXECUTED matrix flops / peak of the pipe they run on, always <= 1; algorithmic-equivalent rates in
their own keys), `configs1_corr_only` (the round-1 headline, kept as a sub-record) and `cpu_baseline` (the same forward on
the host cores through oracle/cpu_chain.py, outside the timed region, with the GPU-vs-CPU parity of three pairs of the
timed batch).
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
for _p in (os.path.join(REPO, "c2-matching_amd"),):
    if _p not in sys.path:
        sys.path.insert(0, _p)

FP32_MATRIX_PEAK_TFLOPS = 157.3   # /opt/skills/guides/MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense
BF16_MATRIX_PEAK_TFLOPS = 2500.0  # same guide: v_mfma_f32_32x32x16_bf16, dense (no sparsity)
# what the 16-bit matrix pipe SUSTAINS (back-to-back MFMAs on all 1024 SIMDs, nothing else): it depends on the operand values
# (power).  Random-valued operand sets that change from one MFMA to the next -- what a convolution feeds it: 1440 - 1630
# (scripts/ubench/mfma_power_share.hip, profiles/r05_ubench_mfma_power_share.log) -> 1530 prices the matrix
# floor.  Constant non-zero operands: 1780 (profiles/r03_ubench_mfma_bf16_rate.log), kept as a second key.
BF16_SUSTAINED_TFLOPS = 1530.0
BF16_SUSTAINED_CONST_OPERANDS_TFLOPS = 1780.0
HBM_PEAK_GBS = 8000.0
HBM_ACHIEVABLE_GBS = 6300.0       # same guide: what a streaming kernel gets
PARITY_SR_TOL = 1e-3              # BASELINE.json north_star: "SR pixels within 1e-3 abs fp32"
PARITY_FLIP_MARGIN = 1e-6         # an index-map flip against the CPU chain must be an fp32 near-tie (float64 margin below this)
METRIC = "LR-Ref image pairs/sec (160x160 LR, 500x500 Ref, 4x SR)"


def _respawn_under_torchrun(args):
    """`python bench.py --gpus N` as a plain command: start N ranks of this very command line on this node."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # the host driver only supports dmabuf IPC (RCCL needs it)
    raise SystemExit(subprocess.call(cmd, env=env))


# ---------------------------------------------------------------------------------------------------------------------
# synthetic inputs
# ---------------------------------------------------------------------------------------------------------------------
def synth_features(B, C, h, valid, dev, seed):
    """N(0,1) features; the ref map carries the constant band a zero-padded 500x500 Ref leaves beyond 125/160
    (ref_cufed_dataset.py:107-114) -> exact ties exist, as in the reference's test-time data."""
    import torch
    g = torch.Generator(device=dev).manual_seed(seed)
    fin = torch.randn((B, C, h, h), generator=g, device=dev, dtype=torch.float32)
    fref = torch.randn((B, C, h, h), generator=g, device=dev, dtype=torch.float32)
    if valid < h:
        const = fref[:, :, valid:valid + 1, valid:valid + 1].clone()
        fref[:, :, valid:, :] = const
        fref[:, :, :, valid:] = const
    return fin, fref


def synth_images(B, h, dev, seed):
    """img_in_lq ~ U(0,1) [B,3,h,h]; img_in_up = bicubic x4; img_ref ~ U(0,1) 500x500 zero-padded to 4h."""
    import torch
    g = torch.Generator(device=dev).manual_seed(seed)
    lq = torch.rand((B, 3, h, h), generator=g, device=dev)
    up = torch.nn.functional.interpolate(lq, scale_factor=4, mode="bicubic", align_corners=False).clamp(0, 1)
    ref = torch.zeros((B, 3, 4 * h, 4 * h), device=dev)
    v = min(4 * h, 500)
    ref[:, :, :v, :v] = torch.rand((B, 3, v, v), generator=g, device=dev)
    return lq, up, ref


def build_models(dev, seed=1234):
    import warnings
    import torch
    from mmsr.models.archs.contras_extractor_arch import ContrasExtractorSep
    from mmsr.models.archs.corres_generation_arch import CorrespondenceGenerationArch
    from mmsr.models.archs.ref_restoration_arch import RestorationNet
    torch.manual_seed(seed)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore", RuntimeWarning)   # "VGG weights are RANDOM": intended here (synthetic benchmark)
        ext = ContrasExtractorSep().eval()
        mp = CorrespondenceGenerationArch(3, 1, ["relu1_1", "relu2_1", "relu3_1"], "vgg19").eval()
    net = RestorationNet(64, 16, 8).eval()
    for m in list(ext.modules()) + list(mp.modules()):
        if isinstance(m, torch.nn.Conv2d):
            torch.nn.init.kaiming_normal_(m.weight)
    for stage in ("small", "medium", "large"):   # live offsets (N(0, 0.01)) instead of the zero-initialised heads
        torch.nn.init.normal_(getattr(net.dyn_agg_restore, f"{stage}_dyn_agg").conv_offset_mask.weight, std=0.01)
    return ext.to(dev), mp.to(dev), net.to(dev)


# ---------------------------------------------------------------------------------------------------------------------
# work models
# ---------------------------------------------------------------------------------------------------------------------
def corr_executed_flops(B, C, h, swept_rows=None):
    """MFMA flops the correlation sweep actually issues (tiles incl. halo / quantisation), per launch.  swept_rows = ref
    pixel rows swept, summed over samples and x-tiles (less than B * x_tiles * h when duplicate rows were eliminated)."""
    tiles = ((h - 2 + 13) // 14) ** 2
    if swept_rows is None:
        swept_rows = B * ((h - 2 + 27) // 28) * h
    return tiles * (swept_rows + B) * 8 * (C // 2) * (2 * 32 * 32 * 2)


def corr_swept_rows(h):
    """(rows swept, rows of the full sweep) of the most recent recorded correlation launch, from the kernel's own skip table."""
    from c2m_amd import ops
    tab = ops.last_corr_skip_table().cpu()
    full = tab.shape[0] * tab.shape[1] * h
    return int(full - (tab[..., 1] - tab[..., 0]).sum().item()), full


def _tf(flops, ms):
    return flops / (ms * 1e-3) / 1e12 if ms > 0 else 0.0


def _rnd(x, n=4):
    return None if x is None else round(float(x), n)


def corr_roofline(B, C, h, kms, n, pmc, swept=None):
    """Exact fp32-MFMA sweep (csrc/corr_argmax.hip; the pre-filter's fall-back and `c2m_feature_match_set_filter(0)`)."""
    exec_flops = corr_executed_flops(B, C, h, swept[0] if swept else None)
    algo = B * 2.0 * ((h - 2) ** 2) ** 2 * C * 9             # 2*Nq*Nr*C*9 per pair
    return {"k": "corr_exact_sweep", "bound": "mfma", "pipe": "fp32 MFMA", "kernel": f"corr_argmax_mfma_kernel<{C}>", "achieved": _rnd(_tf(exec_flops, kms), 2),
            "peak": FP32_MATRIX_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": _rnd(_tf(exec_flops, kms) / FP32_MATRIX_PEAK_TFLOPS),
            "traffic": pmc.get("corr_hbm_bytes_per_launch"), "kernel_ms": _rnd(kms, 3), "launches_timed": n,
            "executed_flops_per_launch": exec_flops, "algorithmic_flops_per_launch": algo,
            "algorithmic_frac": _rnd(_tf(algo, kms) / FP32_MATRIX_PEAK_TFLOPS),
            "ref_rows_swept": swept[0] if swept else None, "ref_rows_full_sweep": swept[1] if swept else None}


def corr_filter_roofline(B, C, h, kern, pmc, swept=None):
    """The correlation on its default path (csrc/corr_filter.hip): the sliding-window sweep on the f16 matrix pipe (three piece
    products per k step: 3 * C/16 MFMAs of 32x32x16 per wave and ref row) + the exact fp32 re-score of the listed candidates."""
    fk, rk = kern.get("corr_filter", []), kern.get("corr_resolve", [])
    fms = sum(fk) / max(len(fk), 1)
    tiles = ((h - 2 + 13) // 14) ** 2
    rows = swept[0] if swept else B * ((h - 2 + 27) // 28) * h
    execd = tiles * (rows + B) * 8 * (3 * C // 16) * (2 * 32 * 32 * 16)
    algo = B * 2.0 * ((h - 2) ** 2) ** 2 * C * 9
    return {"k": "corr_filter", "bound": "mfma", "pipe": "f16 MFMA", "kernel": f"corr_filter_kernel<{C}> (+ corr_resolve: exact fp32 re-score)",
            "achieved": _rnd(_tf(execd, fms), 1), "peak": BF16_MATRIX_PEAK_TFLOPS, "unit": "TFLOP/s",
            "frac": _rnd(_tf(execd, fms) / BF16_MATRIX_PEAK_TFLOPS), "traffic": pmc.get("corr_filter_hbm_bytes_per_launch"),
            "kernel_ms": _rnd(fms, 3), "launches_timed": len(fk), "resolve_ms": _rnd(sum(rk) / max(len(rk), 1), 3),
            "executed_flops_per_launch": execd, "algorithmic_flops_per_launch": algo,
            # SURVEY 8d counts the reference's conv2d formulation (2*Nq*Nr*C*9); the diagonal-sum restructuring issues 8.6x fewer
            # products and skips duplicate ref rows, so the 8d figure over the f16 peak can exceed 1 -- no work is skipped
            # (bit-exact full maps), fewer products are needed
            "algorithmic_frac": _rnd(_tf(algo, fms) / BF16_MATRIX_PEAK_TFLOPS),
            "ref_rows_swept": swept[0] if swept else None, "ref_rows_full_sweep": swept[1] if swept else None}


def dcn_roofline(name, B, C, Co, H, kms, n, traffic=None, src=None, f16x2=False):
    """2*Co*(9C)*H*W flops per sample.  fp32 GEMM: executed = algorithmic on the fp32 matrix pipe; f16 x 2 GEMM
    (the default beside the f16 x 2 convolutions): three products per k step on the f16 pipe -- `frac` = executed / 2.5 PF, and
    `frac_vs_fp32_pipe` = algorithmic / 157.3 TF, the figure north_star's ">= 0.50 on DCNv2 forward" was written for."""
    flops = B * 2.0 * Co * 9 * C * H * H
    execd = 3.0 * flops if f16x2 else flops
    peak = BF16_MATRIX_PEAK_TFLOPS if f16x2 else FP32_MATRIX_PEAK_TFLOPS
    return {"k": f"dcn_{name}", "bound": "mfma", "pipe": "f16 MFMA x3" if f16x2 else "fp32 MFMA", "kernel": f"dcn_v2_forward[{name}: C={C}, {H}x{H}]",
            "achieved": _rnd(_tf(execd, kms), 2), "peak": peak, "unit": "TFLOP/s", "frac": _rnd(_tf(execd, kms) / peak),
            "frac_vs_fp32_pipe": _rnd(_tf(flops, kms) / FP32_MATRIX_PEAK_TFLOPS), "algorithmic_frac": _rnd(_tf(flops, kms) / peak),
            "traffic": traffic, "kernel_ms": _rnd(kms, 3), "launches_timed": n, "algorithmic_flops_per_launch": flops,
            "executed_flops_per_launch": execd}


def _is_split_family(k):
    return "split" in k or k == "bf16" or k.endswith("_bf16")


def conv_rooflines(kern, fam, steps, pmc):
    """The 3x3 convolutions of one step, split by the matrix pipe they run on.  fam = ops.conv_flops_by_family() of the
    timed steps: {family: [launches, algorithmic flops, executed matrix flops]}.  (What the kernels are: profiles/bench_notes.md.)"""
    out = []
    for kid, pipe, peak in (("conv3x3_split", "f16 / bf16 MFMA", BF16_MATRIX_PEAK_TFLOPS), ("conv3x3_mfma", "fp32 MFMA", FP32_MATRIX_PEAK_TFLOPS)):
        ms = kern.get(kid, [])
        if not ms:
            continue
        mine = {k: v for k, v in fam.items() if _is_split_family(k) == (kid == "conv3x3_split")}
        algo = sum(v[1] for v in mine.values()) / steps
        execd = sum(v[2] for v in mine.values()) / steps
        kms = sum(ms) / steps
        e = {"k": kid, "bound": "mfma", "pipe": pipe, "kernel": f"{kid}_kernel family (all launches of one step)",
             "achieved": _rnd(_tf(execd, kms), 1), "peak": peak, "unit": "TFLOP/s", "frac": _rnd(_tf(execd, kms) / peak),
             "kernel_ms": _rnd(kms, 2), "launches_timed": len(ms), "algorithmic_flops_per_launch": algo,
             "executed_flops_per_launch": execd, "algorithmic_equiv_tflops": _rnd(_tf(algo, kms), 1),
             # `frac` = EXECUTED matrix flops / dense peak (f16 x 2: three products per algorithmic product);
             # `algorithmic_frac` = SURVEY 8d's algorithmic flops / the same peak -- the figure 8d's recipe gives
             "algorithmic_frac": _rnd(_tf(algo, kms) / peak),
             "traffic": pmc.get(f"{kid}_hbm_bytes_per_step")}
        if kid == "conv3x3_split":
            e["frac_of_sustained_rate"] = _rnd(_tf(execd, kms) / BF16_SUSTAINED_TFLOPS)
            e["sustained_tflops"] = {"changing_random_operands": BF16_SUSTAINED_TFLOPS, "constant_operands": BF16_SUSTAINED_CONST_OPERANDS_TFLOPS}
            # what actually caps it, measured with rocm-smi while the body layer runs back to back (DESIGN.md 6.14,
            # profiles/r06_power_and_clock_by_kernel.txt): the socket sits at its 1 400 W cap and the shader clock at 1.64 of 2.4 GHz --
            # `peak` above assumes 2.4 GHz; the line's own `power_probe` reads the same two numbers over the whole step
            e["limited_by"] = "socket power cap (1400 W) -> shader clock ~1.64 of 2.4 GHz while this family runs; see power_probe, DESIGN.md 6.14"
            # the family has TWO roofs: the matrix pipe at the rate it sustains on non-zero operands and HBM at
            # the ~6.3 TB/s it delivers; `sum` = what a kernel whose memory and matrix phases do not overlap at all would take
            if e["traffic"]:
                t_mfma = execd / (BF16_SUSTAINED_TFLOPS * 1e12) * 1e3
                t_hbm = e["traffic"] / (HBM_ACHIEVABLE_GBS * 1e9) * 1e3
                e["floors_ms"] = {"mfma_at_sustained_rate": _rnd(t_mfma, 1), "hbm_at_6.3TBs": _rnd(t_hbm, 1), "max": _rnd(max(t_mfma, t_hbm), 1),
                                  "sum": _rnd(t_mfma + t_hbm, 1), "frac_of_max": _rnd(max(t_mfma, t_hbm) / kms)}
        out.append(e)
    return out


# ---------------------------------------------------------------------------------------------------------------------
# CPU baseline (outside the timed region, rank 0, N=1 only)
# ---------------------------------------------------------------------------------------------------------------------
def host_cpu_info():
    model, phys = "unknown", set()
    try:
        pid = cid = None
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name") and model == "unknown":
                model = line.split(":", 1)[1].strip()
            elif line.startswith("physical id"):
                pid = line.split(":", 1)[1].strip()
            elif line.startswith("core id"):
                cid = line.split(":", 1)[1].strip()
            elif not line.strip():
                if pid is not None and cid is not None:
                    phys.add((pid, cid))
                pid = cid = None
    except OSError:
        pass
    logical = os.cpu_count() or 1
    return model, (len(phys) or max(1, logical // 2)), logical


def cpu_baseline_restore(ext, mp, net, lq, up, ref, sr_gpu, idx_gpu, one_thread=True):
    """The same forward on the host cores (oracle/cpu_chain.py: stock torch-CPU convolutions, the reference's conv2d-filter
    correlation algorithm ref_map_util.py:26-86, oracle pre-offsets and oracle DCNv2 -- the reference has no CPU DCNv2).
    Warm-up run + three timed runs (median), each on a different pair of the timed batch (first, middle, last) so that the
    parity of the GPU forward is checked on three pairs: index map vs the CPU chain's own, SR vs the CPU chain run with ITS OWN
    index map (unconditional) and -- for every sampled pair, outside the timed part -- with the GPU's index map (isolates the
    decoder from near-tie flips: `sr_max_abs_diff_given_gpu_index_map`; a second decoder pass only where the maps differ)."""
    import torch
    sys.path.insert(0, os.path.join(REPO, "oracle"))
    import cpu_chain
    import c2m_oracle
    model, phys, logical = host_cpu_info()
    threads = phys
    torch.set_num_threads(threads)
    c2m_oracle.set_num_threads(min(threads, 64))
    B = lq.shape[0]
    pairs = sorted({0, B // 2, B - 1})

    def run(b, cond=True):
        tm = {}
        sr, idx, feats = cpu_chain.full_forward_cpu(ext, mp, net, lq[b:b + 1], up[b:b + 1], ref[b:b + 1], True, tm,
                                                    cond_idx=idx_gpu[b:b + 1] if cond else None)
        return sum(tm.values()), tm, sr, idx, feats   # (the timed part: extractor + correspondence + restoration)

    run(0, cond=False)   # warm-up (page in, thread pools)
    times, stage, parity = [], [], []
    for b in pairs:
        dt, tm, sr_cpu, idx_cpu, feats = run(b)
        times.append(dt)
        stage.append(tm)
        d = (sr_gpu[b].cpu() - sr_cpu[0]).abs()
        mg = cpu_chain.mismatch_margins(feats["dense_features1"][0], feats["dense_features2"][0], idx_gpu[b], idx_cpu[0])
        parity.append({"pair": b, "index_map_flips": len(mg), "of_queries": int(idx_gpu[b].size),
                       "max_fp64_margin_of_flips": float(f"{max((abs(m[3]) for m in mg), default=0.0):.3g}"),
                       "sr_max_abs_diff": float(f"{float(d.max()):.3g}"), "sr_pixels_over_1e-3": int((d > 1e-3).sum()),
                       "sr_pixels": int(d.numel()),
                       "sr_max_abs_diff_given_gpu_index_map": float(f"{float((sr_gpu[b].cpu() - feats['sr_given_idx'][0]).abs().max()):.3g}")})
    med = sorted(times)[len(times) // 2]
    out = {"value": _rnd(1.0 / med, 4), "unit": "pairs/s", "cores": threads, "kind": "port",
           "cpu_model": model, "physical_cores": phys,
           "sample": f"pairs {pairs} of the step's {B}, whole forward each (PyTorch-CPU + C oracle), after a warm-up run; median of "
                     f"{', '.join(f'{t:.1f}' for t in times)} s",
           "stage_s": {k: _rnd(v, 2) for k, v in stage[times.index(med)].items()},
           "parity_gpu_vs_cpu": parity}   # index maps at the IMAGE boundary (two conv implementations): profiles/bench_notes.md
    if one_thread:
        out["one_thread"] = cpu_one_thread(ext, mp, net, lq, up, ref)
    return out


def cpu_one_thread(ext, mp, net, lq, up, ref):
    """1-thread figure on bounded samples of the same pair (a whole configs[2] pair takes minutes on one core): extractor +
    VGG taps on the full images, correlation on a slice of query rows (cost is linear in query rows), RestorationNet on the
    top-left quarter of the LR image (convolution / DCNv2 cost is linear in pixels) -- each scaled to the full pair."""
    import copy
    import torch
    import torch.nn.functional as F
    import cpu_chain
    import c2m_oracle
    import torch_port
    torch.set_num_threads(1)
    c2m_oracle.set_num_threads(1)
    ext_c, mp_c = copy.deepcopy(ext).cpu().eval(), copy.deepcopy(mp).cpu().eval()
    g_c = cpu_chain.cpu_copy(net)
    l1, u1, r1 = lq[:1].cpu(), up[:1].cpu(), ref[:1].cpu()
    h = l1.shape[2]
    with torch.no_grad():
        t0 = time.perf_counter()
        feats = ext_c(u1, r1)
        ref_feat = mp_c.vgg(r1)
        t_ext = time.perf_counter() - t0
        f1 = F.normalize(feats["dense_features1"][0], dim=0)
        f2 = F.normalize(feats["dense_features2"][0], dim=0)
        rows = 6                                              # 4 query-patch rows of h-2
        t0 = time.perf_counter()
        torch_port.feature_match_index_conv(f1[:, :rows].contiguous(), f2, 3, 1, 1, True, True)
        t_corr = (time.perf_counter() - t0) * (h - 2) / (rows - 2)
        q = h // 4                                             # 1/16 of the pixels
        idx = torch.zeros((q - 2, q - 2), dtype=torch.int64)
        offs = c2m_oracle.build_pre_offsets(idx.numpy(), q, q)
        pre = {"relu3_1": torch.from_numpy(offs[0])[None], "relu2_1": torch.from_numpy(offs[1])[None],
               "relu1_1": torch.from_numpy(offs[2])[None]}
        rf = {"relu3_1": ref_feat["relu3_1"][:, :, :q, :q].contiguous(), "relu2_1": ref_feat["relu2_1"][:, :, :2 * q, :2 * q].contiguous(),
              "relu1_1": ref_feat["relu1_1"][:, :, :4 * q, :4 * q].contiguous()}
        t0 = time.perf_counter()
        g_c(l1[:, :, :q, :q].contiguous(), pre, rf)
        t_rest = (time.perf_counter() - t0) * 16.0
    total = t_ext + t_corr + t_rest
    return {"value": _rnd(1.0 / total, 5), "cores": 1,
            "sample": f"extractor + VGG taps in full {t_ext:.1f} s, correlation 4 of {h - 2} query rows x {(h - 2) / 4:.1f} = {t_corr:.1f} s, "
                      f"RestorationNet 1/16 of the pixels x 16 = {t_rest:.1f} s"}


def power_probe(step, sync, it, seconds=3.0):
    """Socket power / shader clock (rocm-smi) while `step` runs back to back for ~`seconds` s: {"socket_w": mean, "socket_w_max", "cap_w",
    "sclk_mhz": mean, "samples"} or {"error": ...}.  A helper thread polls rocm-smi; the main thread keeps the queue full."""
    import re, shutil, subprocess, threading
    exe = shutil.which("rocm-smi") or "/opt/rocm/bin/rocm-smi"
    if not os.path.exists(exe):
        return {"error": "rocm-smi not found"}
    samples, stop = [], threading.Event()

    def poll():
        while not stop.is_set():
            try:
                out = subprocess.run([exe, "-P", "-c", "--json"], capture_output=True, text=True, timeout=10).stdout
                d = next(iter(json.loads(out[out.index("{"):]).values()))
                w = [float(v) for k, v in d.items() if "Package Power" in k]
                c = [int(re.sub(r"\D", "", v)) for k, v in d.items() if k.startswith("sclk clock speed")]
                if w and c:
                    samples.append((w[0], c[0]))
            except Exception:  # noqa: BLE001
                return
    try:
        th = threading.Thread(target=poll, daemon=True)
        t0 = time.perf_counter()
        th.start()
        while time.perf_counter() - t0 < seconds:
            it[0] = 0
            step()
            sync()
        stop.set()
        th.join(timeout=15)
        busy = [x for x in samples if x[0] > 600.0] or samples
        if not busy:
            return {"error": "no sample"}
        cap = None
        try:
            out = subprocess.run([exe, "-M", "--json"], capture_output=True, text=True, timeout=10).stdout
            cap = [float(v) for v in next(iter(json.loads(out[out.index("{"):]).values())).values()][0]
        except Exception:  # noqa: BLE001
            pass
        return {"socket_w": _rnd(sum(a for a, _ in busy) / len(busy), 0), "socket_w_max": max(a for a, _ in busy), "cap_w": cap,
                "sclk_mhz": int(sum(b for _, b in busy) / len(busy)), "samples": len(busy), "what": "rocm-smi while the step runs back to back"}
    except Exception as e:  # noqa: BLE001
        stop.set()
        return {"error": repr(e)}


def cpu_baseline_corr(h, C, budget_s=12.0):
    """configs[1] alone: reference algorithm on PyTorch-CPU, a bounded slice of query rows of one pair."""
    import torch
    sys.path.insert(0, os.path.join(REPO, "oracle"))
    import torch_port
    model, phys, logical = host_cpu_info()
    torch.set_num_threads(phys)
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
    want = int(min(h, max(rows, 2 + (rows - 2) * budget_s / 3 / max(dt, 1e-3))))
    rows = max(rows, want)
    ts = sorted(run(rows) for _ in range(3))
    dt = ts[1]
    frac = (rows - 2) / (h - 2)
    return {"value": frac / dt, "unit": "pairs/s", "cores": phys, "kind": "port", "cpu_model": model, "physical_cores": phys,
            "logical_cpus": logical,
            "sample": f"{rows} of {h} query rows of one {h}x{h}x{C} pair vs full ref map, PyTorch-CPU conv2d+max (reference "
                      f"algorithm, oracle/torch_port.py), median of 3 ({dt:.2f} s); linear extrapolation to a pair"}


# ---------------------------------------------------------------------------------------------------------------------
def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=None, help="pairs per GPU per step (default 16; 4 for --lr 320; --workload train: "
                                                             "--global-batch / world size)")
    ap.add_argument("--global-batch", type=int, default=32,
                    help="--workload train: pairs per step over ALL ranks (BASELINE configs[3]: 32); each rank takes global / world, "
                         "which must divide (mmsr/data/__init__.py:70-73)")
    ap.add_argument("--lr", type=int, default=None, help="LR size = feature-map size (default 160; 40 for --workload train)")
    ap.add_argument("--workload", choices=("restore", "corr", "train"), default="restore")
    ap.add_argument("--graph", choices=("0", "1"), default="0",
                    help="train workload: capture the training step into a hipGraph (train.hip_graph; one process only).  Measured on "
                         "configs[3]: 33.1 ms against 31.7 ms eager -- the step is GPU-bound, not launch-bound -- hence off by default")
    ap.add_argument("--no-alt", action="store_true", help="skip the extra passes on the other convolution arithmetics")
    ap.add_argument("--dtype", choices=("f32", "bf16"), default="f32", help="bf16: inference under torch.autocast(bfloat16) (configs[4])")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--experimental", action="store_true",
                    help="also time the step on the Winograd-along-y kernels of csrc/experimental/ (a library built with `make "
                         "EXPERIMENTAL=1`; round 5's recorded no-go, DESIGN.md 6.7)")
    ap.add_argument("--plumbing-only", action="store_true",
                    help="launch-path check without a GPU workload: respawn under torch.distributed.run, read RANK / LOCAL_RANK / "
                         "WORLD_SIZE, init the process group ($C2M_BENCH_BACKEND, default nccl = RCCL; gloo on a CPU-only host), "
                         "barrier + the MAX all-reduce the timing uses, print one JSON line, exit (tests/test_data_path.py)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        _respawn_under_torchrun(args)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus != world and not (args.gpus == 1 and world == 1):
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")

    import torch
    backend = os.environ.get("C2M_BENCH_BACKEND", "nccl")   # "nccl" IS RCCL on ROCm; "gloo" only with --plumbing-only
    on_gpu = not (args.plumbing_only and backend == "gloo")
    if on_gpu:
        assert torch.cuda.is_available(), "bench.py needs an MI355X"
        torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank) if on_gpu else torch.device("cpu")
    dist = None
    if world > 1 or os.environ.get("C2M_BENCH_FORCE_DIST") == "1":   # (the env var: exercise the RCCL path on a 1-GPU box)
        import torch.distributed as dist
        if not dist.is_initialized():
            if "MASTER_ADDR" not in os.environ:
                os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", "29533"
                os.environ.setdefault("RANK", "0")
                os.environ.setdefault("WORLD_SIZE", "1")
            if on_gpu:
                dist.init_process_group(backend, device_id=dev)  # RCCL over xGMI
            else:
                dist.init_process_group(backend)
    rccl_world = dist.get_world_size() if dist is not None else None
    if args.plumbing_only:
        # what every timed region does around its steps: barrier, then MAX over ranks of the per-rank time
        t = torch.tensor([float(rank + 1)], device=dev, dtype=torch.float64)
        if dist is not None:
            dist.barrier()
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
        if rank == 0:
            print(json.dumps({"plumbing": "ok", "backend": backend, "n_gpus": args.gpus, "world_size": world,
                              "process_group_world": rccl_world, "max_over_ranks": float(t.item()),
                              "local_rank": local_rank, "master": os.environ.get("MASTER_ADDR")}))
        if dist is not None:
            dist.destroy_process_group()
        return

    import c2m_amd
    ops = c2m_amd.ops
    train = args.workload == "train"
    h = args.lr or (40 if train else 160)
    if train and args.batch is None:
        from mmsr.data import per_rank_batch_size
        B = per_rank_batch_size(args.global_batch, world)     # asserts global % world == 0 as the reference's dataloader does
    else:
        B = args.batch or (4 if h >= 320 else 16)
    C = 256

    def sync():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    def timed(step_fn):
        for _ in range(args.warmup):
            step_fn()
        ops.count_conv_flops(True)   # (re)start the opt-in conv FLOP accounting: only the timed steps count
        c2m_amd.profile_enable(True)
        c2m_amd.profile_collect()
        sync()
        t0 = time.perf_counter()
        with ops.record_corr_skip_table():
            for _ in range(args.steps):
                out = step_fn()
        sync()
        dt = time.perf_counter() - t0
        timed.families = ops.conv_flops_by_family()
        ops.count_conv_flops(False)
        prof = c2m_amd.profile_collect(capacity=65536)
        c2m_amd.profile_enable(False)
        timed.rank_step_ms = None
        if dist is not None:
            # the line's time is the MAX over ranks (driver contract); the MIN travels with it so that a scaling loss can be
            # attributed: min ~ max -> every rank slowed down (shared resource), min << max -> one straggler
            t = torch.tensor([dt, -dt], device=dev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            timed.rank_step_ms = {"min_over_ranks": _rnd(-float(t[1].item()) / args.steps * 1e3, 3),
                                  "max_over_ranks": _rnd(float(t[0].item()) / args.steps * 1e3, 3), "this_rank": _rnd(dt / args.steps * 1e3, 3)}
            dt = float(t[0].item())
        kern = {}
        for name, ms in prof:
            kern.setdefault(name, []).append(ms)
        return dt, kern, out

    def finish(line):
        try:   # peak of torch's allocator on this rank (inputs, weights, activations and the correlation's workspaces all live there)
            line["hbm_peak_gb"] = round(torch.cuda.max_memory_allocated() / 2 ** 30, 2)
        except Exception:  # noqa: BLE001
            pass
        if rank == 0:
            print(json.dumps(line))
        if dist is not None:
            dist.destroy_process_group()

    # HBM bytes from the PMC passes kept under profiles/ (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate runs of this very
    # command; gfx950 corrections applied by scripts/summarize_step.py).  STATIC: measured once per round on the default
    # workload, not during this run -- `traffic_source` says so in the line
    pmc = {}
    tfile = os.path.join(REPO, "profiles", "step_pmc_traffic.json")
    if os.path.exists(tfile) and (B, h, args.workload, args.dtype) == (16, 160, "restore", "f32"):
        pmc = json.load(open(tfile))
        pmc["source"] = ("profiles/step_pmc_traffic.json (static: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of "
                         f"`python bench.py`, {pmc.get('measured_at', 'see profiles/README.md')}; not re-measured in this run)")
    base = {"unit": "pairs/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "data": "synthetic", "rccl_world_size": rccl_world}
    timed.rank_step_ms = None

    # ---- configs[3]: stage-3 training step ---------------------------------------------------------------------------
    if train:
        import warnings
        from mmsr.models.base_model import unwrap
        from mmsr.models.ref_restoration_model import RefRestorationModel
        opt = {"dist": dist is not None, "gpu_ids": [local_rank], "is_train": True, "path": {},
               "network_g": {"type": "RestorationNet", "ngf": 64, "n_blocks": 16, "groups": 8},
               "network_map": {"type": "CorrespondenceGenerationArch", "patch_size": 3, "stride": 1,
                               "vgg_layer_list": ["relu1_1", "relu2_1", "relu3_1"], "vgg_type": "vgg19"},
               "network_extractor": {"type": "ContrasExtractorSep"},
               "train": {"lr_g": 1e-4, "lr_offset": 1e-4, "lr_relu2_offset": 1e-5, "lr_relu3_offset": 1e-6,
                         "weight_decay_g": 0, "beta_g": [0.9, 0.999], "pixel_weight": 1.0,
                         "hip_graph": dist is None and args.graph == "1"}}
        if args.warmup < 4 and opt["train"]["hip_graph"]:
            raise SystemExit("--workload train with the hipGraph step needs --warmup >= 4 (2 eager steps + capture + 1 replay)")
        torch.manual_seed(10)     # same initial weights on every rank (DDP broadcasts rank 0's anyway)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore", RuntimeWarning)
            model = RefRestorationModel(opt)
        # which synthetic pairs this rank trains on: the reference's rank partition (DistIterSampler, data_sampler.py:50-63) over
        # a 4096-pair synthetic set -- every rank draws the same epoch permutation and keeps its stride-by-rank share
        from mmsr.data import DistIterSampler
        smp = DistIterSampler(range(4096), num_replicas=world, rank=rank, ratio=1)
        mine = [i for _, i in zip(range(B), iter(smp))]

        def pair(i):
            g = torch.Generator().manual_seed(100000 + i)
            return torch.rand((3, 4 * h, 4 * h), generator=g), torch.rand((3, 4 * h, 4 * h), generator=g)
        pairs = [pair(i) for i in mine]
        gt = torch.stack([a for a, _ in pairs])
        lq = torch.nn.functional.interpolate(gt, scale_factor=0.25, mode="bicubic", align_corners=False).clamp(0, 1)
        up = torch.nn.functional.interpolate(lq, scale_factor=4, mode="bicubic", align_corners=False).clamp(0, 1)
        model.feed_data({"img_in_lq": lq, "img_ref": torch.stack([r for _, r in pairs]), "img_in": gt, "img_in_up": up})
        it = [0]

        def train_step():
            it[0] += 1
            model.optimize_parameters(it[0])
            return model.log_dict["l_g_pix"]

        dt, kern, loss = timed(train_step)
        grad_bytes = sum(p.numel() * 4 for p in unwrap(model.net_g).parameters() if p.requires_grad)
        line = dict(base, metric=METRIC + " -- stage-3 training step", value=_rnd(B * world * args.steps / dt, 3),
                    ms_per_step=_rnd(dt / args.steps * 1e3, 3), dtype="f32", scaling="strong" if args.batch is None else "weak",
                    config={"workload": f"configs[3]: stage-3 MSE training step (extractor + correspondence under no_grad, RestorationNet "
                                        f"forward, L1 loss, backward incl. three DCNv2 backward passes, Adam with the reference's four "
                                        f"parameter groups), {B} pairs per GPU, GT {4*h}x{4*h} (LR {h}x{h}, Ref {4*h}x{4*h})",
                            "global_batch": B * world,
                            "parallelism": f"dp{world}" + (" (DDP over RCCL: one all-reduce of net_g's gradients per step, bucketed, "
                                                          "overlapped with backward)" if dist is not None else " (single process)")},
                    gradient_allreduce_bytes_per_step_per_gpu=grad_bytes if dist is not None else 0,
                    net_g_gradient_bytes=grad_bytes, loss=float(loss), hip_graph=bool(opt["train"]["hip_graph"]),
                    train_kernels=os.environ.get("C2M_TRAIN_KERNELS", "auto"), pairs_of_this_rank=mine[:4] + ["..."],
                    c2m_kernel_ms_per_step={k: _rnd(sum(v) / args.steps, 3) for k, v in kern.items()},
                    rank_step_ms=timed.rank_step_ms)
        return finish(line)

    # ---- configs[1] leg: correlation only on synthetic features (sub-record of the default line) --------------------
    valid = min(h, 125)   # 500x500 Ref inside the zero-padded canvas, at feature scale (/4)
    fin, fref = synth_features(B, C, h, valid, dev, 1234 + rank)

    def corr_step():
        n1 = ops.feature_normalize(fin)
        n2 = ops.feature_normalize(fref)
        idx, val = ops.feature_match_index_batched(n1, n2, 3, 1, 1, True, True)
        return idx, val, ops.build_pre_offsets(idx, h, h)

    if args.workload == "corr":
        dt, kern, out = timed(corr_step)
        assert int(out[0].min()) >= 0 and int(out[0].max()) < (h - 2) ** 2
        swept = corr_swept_rows(h)
        ck = kern.get("corr_argmax_mfma", [])
        filt = bool(kern.get("corr_filter"))
        line = dict(base, metric=METRIC + " -- correlation + index map only", value=B * world * args.steps / dt,
                    ms_per_step=dt / args.steps * 1e3, dtype="f32",
                    config={"workload": f"configs[1]: batch-{B} {h}x{h} LR / 500x500 Ref (zero-padded to {4*h}), feature "
                                        "normalise + 3x3 correlation/arg-max index map + pre-offset maps (NO DCNv2 / decoder)",
                            "feature_channels": C, "parallelism": f"dp{world} (batch-sharded, no collective)"},
                    roofline=corr_filter_roofline(B, C, h, kern, pmc, swept) if filt else
                    corr_roofline(B, C, h, sum(ck) / max(len(ck), 1), len(ck), pmc, swept))
        line["c2m_kernel_ms_per_step"] = {k: round(sum(v) / args.steps, 4) for k, v in kern.items()}
        line["rank_step_ms"] = timed.rank_step_ms
        if world == 1 and not args.no_cpu_baseline and rank == 0:
            line["cpu_baseline"] = cpu_baseline_corr(h, C)
        return finish(line)

    # ---- configs[2] / configs[4]: full restoration forward -----------------------------------------------------------
    ext, mp, net = build_models(dev)
    lq, up, ref = synth_images(B, h, dev, 1234 + rank)
    ev = [[torch.cuda.Event(enable_timing=True) for _ in range(4)] for _ in range(args.steps + args.warmup)]
    it = [0]
    last = {}
    bf16 = args.dtype == "bf16"

    class _StepOwner:   # (what RefRestorationModel is to its test(): remembers a pinned full-range flavour)
        pass
    step_owner = _StepOwner()

    @torch.no_grad()
    def restore_step():
        e = ev[it[0]]
        it[0] += 1

        def whole():   # RefRestorationModel.test(): extractor -> correspondence + VGG taps -> RestorationNet
            with torch.autocast("cuda", dtype=torch.bfloat16, enabled=bf16):
                e[0].record()
                feats = ext(up, ref)
                e[1].record()
                pre, ref_feat = mp(feats, ref)
                e[2].record()
                sr_ = net(lq, pre, ref_feat)
                e[3].record()
            last["pre"] = pre
            return sr_
        # one f16 x 2 range check per step, as RefRestorationModel.test() does it (the module forwards' guards nest inside and
        # skip their own read-backs): $C2M_BENCH_STEP_GUARD=0 restores one check per module (three host syncs per step)
        if os.environ.get("C2M_BENCH_STEP_GUARD", "1") != "0":
            return ops.f16_range_guard(step_owner, whole, dev)
        return whole()

    dt, kern, sr = timed(restore_step)
    rank_ms = timed.rank_step_ms
    pre_timed = last["pre"]     # (the extra passes on other arithmetics below overwrite last["pre"])
    fam = timed.families
    swept = corr_swept_rows(h)   # of the timed steps' correlation launch (same inputs every step)
    assert tuple(sr.shape) == (B, 3, 4 * h, 4 * h) and bool(torch.isfinite(sr).all())
    stage = {"extractor": 0.0, "correspondence": 0.0, "restoration": 0.0}
    for e in ev[args.warmup:]:
        for k, name in enumerate(stage):
            stage[name] += e[k].elapsed_time(e[k + 1]) / args.steps

    sub = None
    if rank == 0 and dist is None and h == 160 and not bf16:   # configs[1] sub-record (short: 3 steps)
        steps_keep, args.steps = args.steps, 3
        sdt, skern, _ = timed(corr_step)
        args.steps = steps_keep
        sk = skern.get("corr_argmax_mfma", [])
        sub = {"workload": "configs[1]: normalise + correlation/arg-max + pre-offsets on synthetic features (round-1 headline)",
               "pairs_per_s": B * 3 / sdt, "ms_per_step": sdt / 3 * 1e3, "corr_kernel_ms": sum(sk) / max(len(sk), 1)}

    line = None
    if rank == 0:
        rl = []
        ck = kern.get("corr_argmax_mfma", [])
        if kern.get("corr_filter"):
            rl.append(corr_filter_roofline(B, C, h, kern, pmc, swept))
        elif ck:
            rl.append(corr_roofline(B, C, h, sum(ck) / len(ck), len(ck), pmc, swept))
        dk = kern.get("dcn_v2_forward", [])
        layers = (("small", 256, h), ("medium", 128, 2 * h), ("large", 64, 4 * h))
        if dk and len(dk) % 3 == 0:   # launch order inside a step: small, medium, large (ref_restoration_arch.py:152-180)
            dtraf = sorted(pmc.get("dcn_v2_forward_hbm_bytes_per_launch", {}).values())   # small < medium < large
            from c2m_amd import ops as _o
            d16 = bool(_o._DCN_F16X2 and _o._SPLIT16 and _o._SPLIT != "0")
            for k, (lname, ch, hh) in enumerate(layers):
                mine = dk[k::3]
                rl.append(dcn_roofline(lname, B, ch, ch, hh, sum(mine) / len(mine), len(mine),
                                       dtraf[k] if len(dtraf) == 3 else None, pmc.get("source"), f16x2=d16))
            tot = sum(dk) / (len(dk) // 3)
            flops = sum(B * 2.0 * ch * 9 * ch * hh * hh for _, ch, hh in layers)
            dpeak, dmul = (BF16_MATRIX_PEAK_TFLOPS, 3.0) if d16 else (FP32_MATRIX_PEAK_TFLOPS, 1.0)
            rl.append({"k": "dcn_all_three", "bound": "mfma", "pipe": "f16 MFMA x3" if d16 else "fp32 MFMA", "kernel": "dcn_v2_forward[all three DynAgg layers of one step]",
                       "achieved": _rnd(_tf(dmul * flops, tot), 2), "peak": dpeak, "unit": "TFLOP/s",
                       "frac": _rnd(_tf(dmul * flops, tot) / dpeak), "frac_vs_fp32_pipe": _rnd(_tf(flops, tot) / FP32_MATRIX_PEAK_TFLOPS),
                       "traffic": None, "kernel_ms": _rnd(tot, 3),
                       "launches_timed": len(dk), "algorithmic_flops_per_launch": flops})
        rl += conv_rooflines(kern, fam, args.steps, pmc)
        dominant = max((r for r in rl if "all three" not in r["kernel"]), key=lambda r: r["kernel_ms"]) if rl else None
        cfg = (f"configs[4]: CUFED5-shape inference, LR {h}x{h} / Ref 500x500 zero-padded to {4*h}x{4*h}, bf16 autocast, {B} pairs per GPU per step"
               if bf16 else
               f"configs[2]: batch-{B} full restoration forward (extractor + correlation/index map + VGG taps + RestorationNet with 3 DCNv2 "
               f"warps + decoder), LR {h}x{h}, Ref 500x500 zero-padded to {4*h}x{4*h}, SR {4*h}x{4*h}, fp32")
        # compact per-kernel table: [ms per step, frac of the pipe's dense peak, pipe]
        # (4th entry: correlation -- the re-score's ms; DCNv2 -- algorithmic flops / fp32 matrix peak, north_star's DCNv2 figure)
        table = {r["k"]: [r["kernel_ms"], r["frac"], r["pipe"]] + ([r["resolve_ms"]] if "resolve_ms" in r else []) +
                 ([r["frac_vs_fp32_pipe"]] if "frac_vs_fp32_pipe" in r else []) for r in rl}
        line = dict(base, metric=METRIC, value=_rnd(B * world * args.steps / dt, 3), ms_per_step=_rnd(dt / args.steps * 1e3, 3),
                    dtype="bf16" if bf16 else "f32",
                    config={"workload": cfg, "parallelism": f"dp{world} (batch-sharded, no collective)"},
                    stage_ms={k: _rnd(v, 2) for k, v in stage.items()}, kernel_table=table, roofline=dominant,
                    c2m_kernel_ms_per_step={k: _rnd(sum(v) / args.steps, 3) for k, v in kern.items()},
                    configs1_corr_only=sub, notes="profiles/bench_notes.md", rank_step_ms=rank_ms)
        if not bf16:
            try:   # measured, on this GPU, in this run: distance of every convolution arithmetic from a float64 convolution
                g_ = torch.Generator(device=dev).manual_seed(77)
                xe = torch.randn((1, 256, 24, 64), generator=g_, device=dev).contiguous(memory_format=torch.channels_last)
                we = torch.randn((256, 256, 3, 3), generator=g_, device=dev) / 48.0
                be = torch.randn((256,), generator=g_, device=dev)
                want = torch.nn.functional.conv2d(xe.double(), we.double(), be.double(), padding=1)
                chk = {}   # [max abs, rms] error on a 256 -> 256 layer (K = 2304, outputs ~5)
                from c2m_amd import ops as _o
                arith = [("fp32_mfma", "direct"), ("f16x2", "split16"), ("bf16x3", "split")]
                if args.experimental and _o.experimental_built():
                    arith += [("wino16_f43y", "wino16"), ("wino16_f23y", "wino16_f23")]
                for name, algo in arith:
                    d_ = (_o.conv3x3(xe, we, be, algo=algo).double() - want)
                    chk[name] = [float(f"{float(d_.abs().max()):.3g}"), float(f"{float(d_.pow(2).mean().sqrt()):.3g}")]
                line["conv_error_vs_fp64"] = chk
            except Exception as e:  # noqa: BLE001 -- a diagnostic, never the reason a bench line is lost
                line["conv_error_vs_fp64"] = {"error": repr(e)}
            # what the step is bound by, read off the part itself (outside the timed region): socket power and shader clock sampled by
            # rocm-smi while the same step runs back to back for a few seconds.  The dominant convolution family sits at the socket's
            # power cap; the whole step averages a little under it.  Never the reason a bench line is lost.
            if not args.no_alt and world == 1 and rank == 0:   # (one process only: rocm-smi lists every card, the parser reads the first)
                line["power_probe"] = power_probe(restore_step, sync, it)
            if not args.no_alt:
                # the same step on the other arithmetics (outside the timed region of `value`): strict fp32-MFMA convolutions +
                # exact fp32 correlation sweep (no 16-bit pipe anywhere), and the bf16 x 3 convolution flavour
                from c2m_amd import ops as _ops

                def rerun(n_):
                    it[0] = 0
                    restore_step()
                    sync()
                    t0 = time.perf_counter()
                    for _ in range(n_):
                        it[0] = 0
                        restore_step()
                    sync()
                    return B * world * n_ / (time.perf_counter() - t0)
                n_alt = min(args.steps, 5)
                keep = (_ops._SPLIT16, _ops._SPLIT)
                try:
                    _ops._SPLIT16, _ops._SPLIT = False, keep[1]
                    line["value_bf16x3_convolutions"] = _rnd(rerun(n_alt), 2)
                    _ops._SPLIT16, _ops._SPLIT = False, "0"
                    with _ops.corr_filter_mode(0):
                        line["value_strict_fp32_mfma"] = _rnd(rerun(n_alt), 2)
                    _ops._SPLIT16, _ops._SPLIT = keep
                    with _ops.corr_filter_mode(0):
                        line["value_exact_corr_sweep"] = _rnd(rerun(n_alt), 2)
                    # round 5's go / no-go on Winograd-along-y over the f16 x 2 pieces (csrc/experimental/conv3x3_wino16.hip): the
                    # same step with every eligible channels-last layer on the F(4,3) / F(2,3) kernel -- only with --experimental
                    # on a library built with `make EXPERIMENTAL=1` (the no-go is recorded: DESIGN.md 6.7)
                    if args.experimental and _ops.experimental_built():
                        keep_w = _ops._WINO16
                        try:
                            _ops._WINO16 = 7
                            line["value_wino16_f43y_convolutions"] = _rnd(rerun(n_alt), 2)
                            _ops._WINO16 = 8
                            line["value_wino16_f23y_convolutions"] = _rnd(rerun(n_alt), 2)
                        finally:
                            _ops._WINO16 = keep_w
                finally:
                    _ops._SPLIT16, _ops._SPLIT = keep
        if world == 1 and not args.no_cpu_baseline and not bf16:
            idx_gpu = pre_timed.max_idx.cpu().numpy()
            line["cpu_baseline"] = cpu_baseline_restore(ext, mp, net, lq, up, ref, sr, idx_gpu)
            # north_star's tolerance as an ASSERTION of the run: given the GPU's own index map the SR image must
            # agree with the CPU chain to 1e-3 (measured ~1e-7), and an index-map flip is only acceptable as an fp32 near-tie
            # (float64 margin of the two candidates < 1e-6; measured ~1e-7).  The line is printed either way; the exit code says it.
            par = line["cpu_baseline"]["parity_gpu_vs_cpu"]
            bad = [q for q in par if q["sr_max_abs_diff_given_gpu_index_map"] > PARITY_SR_TOL or q["max_fp64_margin_of_flips"] > PARITY_FLIP_MARGIN]
            line["parity_ok"] = not bad
            if bad:
                finish(line)
                raise SystemExit(f"bench.py: parity violated on pairs {[q['pair'] for q in bad]} (SR given the GPU index map > {PARITY_SR_TOL} "
                                 f"or an index flip with float64 margin > {PARITY_FLIP_MARGIN}): {bad}")
    finish(line)


if __name__ == "__main__":
    main()
