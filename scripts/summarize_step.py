#!/usr/bin/env python3
"""Condense the rocprofv3 output of `scripts/gpu_run.sh <src> prof pmc` (gpurun_out/<src>/) into the summaries kept under profiles/.

  summarize_step.py <tag> [steps_in_pmc_runs] [src]       e.g. summarize_step.py r04_final 4 r04_final
writes profiles/<tag>_step_kernel_stats.csv   kernel stats of `bench.py --steps 3 --warmup 2` (c2m names shortened)
       profiles/<tag>_pmc_counters.json       per kernel: every PMC counter collected (separate passes merged)
       profiles/step_pmc_traffic.json         HBM bytes per launch / per step + MFMA-busy fractions, read by bench.py
and copies the logs (bench_*.log, pytest_gpu.log, smoke.log) as profiles/<tag>_*.
HBM bytes = (2 x FETCH_SIZE + WRITE_SIZE) KiB per /opt/skills/guides/MI355X_MICROARCH.md: gfx950 reports half of the
streamed read bytes; FETCH_SIZE and WRITE_SIZE come from their own passes.
"""
import csv, glob, json, os, re, shutil, sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]
nsteps = int(sys.argv[2]) if len(sys.argv) > 2 else 4      # bench.py --steps 2 --warmup 2 in the PMC passes
SRC = os.path.join(REPO, "gpurun_out", sys.argv[3] if len(sys.argv) > 3 else "final")
csv.field_size_limit(1 << 30)


def short(name):
    if "corr_filter_kernel" in name and "c2m::" not in name:      # (rocprofv3 leaves this template's name mangled)
        return "c2m::corrf::corr_filter_kernel"
    m = re.search(r"(c2m::[A-Za-z0-9_:]+(<[^>(]*>)?)", name)
    return m.group(1) if m else name[:100]


for f in glob.glob(os.path.join(SRC, "prof_step", "**", "*kernel_stats.csv"), recursive=True):
    rows = list(csv.reader(open(f)))
    with open(os.path.join(REPO, "profiles", f"{tag}_step_kernel_stats.csv"), "w", newline="") as o:
        w = csv.writer(o)
        w.writerow(rows[0])
        for r in rows[1:]:
            w.writerow([short(r[0])] + r[1:])

acc = {}
for d in ("pmc_mfma", "pmc_fetch", "pmc_write"):
    for fn in glob.glob(os.path.join(SRC, d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(fn)):
            s = short(r["Kernel_Name"])
            if not s.startswith("c2m::"):
                continue
            a = acc.setdefault(s, {}).setdefault(r["Counter_Name"], {})
            a[r["Dispatch_Id"]] = a.get(r["Dispatch_Id"], 0.0) + float(r["Counter_Value"])   # summed over the XCD rows
summary = {k: {c: {"launches": len(v), "mean_per_launch": sum(v.values()) / len(v), "sum": sum(v.values())} for c, v in cs.items()}
           for k, cs in acc.items()}
json.dump(summary, open(os.path.join(REPO, "profiles", f"{tag}_pmc_counters.json"), "w"), indent=1, sort_keys=True)


def hbm(k, per="mean_per_launch"):
    c = summary[k]
    if "FETCH_SIZE" not in c:
        return None
    return (2.0 * c["FETCH_SIZE"][per] + c.get("WRITE_SIZE", {}).get(per, 0.0)) * 1024.0


busy = {}
for k, c in summary.items():
    if "SQ_VALU_MFMA_BUSY_CYCLES" in c and "GRBM_GUI_ACTIVE" in c and c["GRBM_GUI_ACTIVE"]["sum"] > 0:
        busy[k] = (c["SQ_VALU_MFMA_BUSY_CYCLES"]["sum"] / 1024.0) / (c["GRBM_GUI_ACTIVE"]["sum"] / 8.0)
conv_split = [k for k in summary if "conv3x3_split_kernel" in k]
conv_mfma = [k for k in summary if "conv3x3" in k and "relayout" not in k and "split" not in k and "wgrad" not in k]
dcn = [k for k in summary if "dcn_fwd" in k]
corr = [k for k in summary if "corr_argmax_mfma_kernel" in k]
corrf = [k for k in summary if "corr_filter_kernel" in k]
out = {
    "workload": "bench.py default (configs[2], B=16, LR 160), rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes",
    "measured_at": f"{tag} (scripts/gpu_run.sh, stages prof + pmc)",
    "conv3x3_split_hbm_bytes_per_step": sum(hbm(k, "sum") or 0.0 for k in conv_split) / nsteps,
    "conv3x3_mfma_hbm_bytes_per_step": sum(hbm(k, "sum") or 0.0 for k in conv_mfma) / nsteps,
    "dcn_v2_forward_hbm_bytes_per_step": sum(hbm(k, "sum") or 0.0 for k in dcn) / nsteps,
    "dcn_v2_forward_hbm_bytes_per_launch": {k: hbm(k) for k in dcn},
    "corr_hbm_bytes_per_launch": hbm(corr[0]) if corr else None,
    "corr_filter_hbm_bytes_per_launch": hbm(corrf[0]) if corrf else None,
    "corr_filter_mfma_busy_fraction": busy.get(corrf[0]) if corrf else None,
    "corr_mfma_busy_fraction": busy.get(corr[0]) if corr else None,
    "mfma_busy_fraction": busy,
    "note": "bytes = (2 x FETCH_SIZE + WRITE_SIZE) KiB per MI355X_MICROARCH.md (gfx950 reports half of the streamed read bytes; "
            "LDS-DMA reads uncalibrated: upper bound). mfma_busy_fraction = SQ_VALU_MFMA_BUSY_CYCLES / 1024 SIMDs / "
            "(GRBM_GUI_ACTIVE / 8 XCDs), summed over the launches of the run.",
}
json.dump(out, open(os.path.join(REPO, "profiles", "step_pmc_traffic.json"), "w"), indent=1)
for name in ("bench_default", "bench_corr", "bench_conv", "bench_dcn", "bench_train", "bench_train_stock", "bench_train_b4", "bench_train_b4_stock",
             "bench_train_kernels", "bench_train_rccl_1rank", "bench_cfg5_bf16", "bench_cfg5_f32", "pytest_gpu", "smoke", "diag_corr_filter"):
    src = os.path.join(SRC, name + ".log")
    if os.path.exists(src):
        shutil.copy(src, os.path.join(REPO, "profiles", f"{tag}_{name}.log"))
print(json.dumps({k: round(v, 3) for k, v in busy.items()}, indent=1))
print({k: v for k, v in out.items() if k.endswith("per_step") or k.startswith("corr_")})
