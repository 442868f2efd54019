#!/usr/bin/env python3
"""Condense rocprofv3 outputs (gpurun_out/) into the small summaries kept under profiles/.

  summarize_pmc.py <tag>      e.g. r01_final2
writes profiles/<tag>_corr_kernel_stats.csv   (c2m kernels of the --kernel-trace --stats run of bench.py)
       profiles/<tag>_dcn_kernel_stats.csv    (same for scripts/bench_dcn.py)
       profiles/<tag>_corr_pmc_counters.json  (per-kernel mean of every PMC counter collected, separate passes merged)
       profiles/corr_pmc_traffic.json         (HBM bytes per launch of the correlation kernel, read by bench.py)
"""
import csv, glob, json, os, re, sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(REPO, "gpurun_out")
tag = sys.argv[1]
csv.field_size_limit(1 << 30)


def short(name):
    m = re.search(r"(c2m::[A-Za-z0-9_:]+(<[^>(]*>)?)", name)
    return m.group(1) if m else None


def stats(src, dst):
    if not os.path.exists(src):
        return
    rows = list(csv.reader(open(src)))
    with open(dst, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(rows[0])
        for r in rows[1:]:
            s = short(r[0])
            if s:
                w.writerow([s] + r[1:])


stats(os.path.join(OUT, "prof_corr", "corr_kernel_stats.csv"), os.path.join(REPO, "profiles", f"{tag}_corr_kernel_stats.csv"))
stats(os.path.join(OUT, "prof_dcn", "dcn_kernel_stats.csv"), os.path.join(REPO, "profiles", f"{tag}_dcn_kernel_stats.csv"))

acc = {}
for d in ("pmc_fetch", "pmc_write", "pmc_mfma"):
    for fn in glob.glob(os.path.join(OUT, d, "*counter_collection.csv")):
        for r in csv.DictReader(open(fn)):
            s = short(r["Kernel_Name"])
            if not s:
                continue
            a = acc.setdefault(s, {}).setdefault(r["Counter_Name"], {})
            a.setdefault(r["Dispatch_Id"], 0.0)
            a[r["Dispatch_Id"]] += float(r["Counter_Value"])   # summed over XCD rows of one dispatch
summary = {k: {c: {"launches": len(v), "mean_per_launch": sum(v.values()) / len(v)} for c, v in cs.items()} for k, cs in acc.items()}
json.dump(summary, open(os.path.join(REPO, "profiles", f"{tag}_corr_pmc_counters.json"), "w"), indent=1, sort_keys=True)

ck = [k for k in summary if "corr_argmax_mfma_kernel" in k]
if ck and "FETCH_SIZE" in summary[ck[0]]:
    f = summary[ck[0]]["FETCH_SIZE"]["mean_per_launch"]
    wv = summary[ck[0]].get("WRITE_SIZE", {}).get("mean_per_launch", 0.0)
    c = summary[ck[0]]
    util = None
    if "SQ_VALU_MFMA_BUSY_CYCLES" in c and "GRBM_GUI_ACTIVE" in c:
        util = (c["SQ_VALU_MFMA_BUSY_CYCLES"]["mean_per_launch"] / 1024.0) / (c["GRBM_GUI_ACTIVE"]["mean_per_launch"] / 8.0)
    json.dump({"kernel": ck[0], "workload": "B=16, 160x160x256 feature maps", "FETCH_SIZE_KB": f, "WRITE_SIZE_KB": wv,
               "hbm_bytes_per_launch": (2.0 * f + wv) * 1024.0, "mfma_busy_fraction": util,
               "note": "separate --pmc passes (FETCH_SIZE, WRITE_SIZE); FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 "
                       "reports 1/2 of streamed read bytes; the LDS-DMA reads are uncalibrated, so this is an upper bound). "
                       "Compulsory bytes = 16 x 52.73 MB = 843.7 MB. mfma_busy_fraction = SQ_VALU_MFMA_BUSY_CYCLES / 1024 SIMDs "
                       "/ (GRBM_GUI_ACTIVE / 8 XCDs)."},
              open(os.path.join(REPO, "profiles", "corr_pmc_traffic.json"), "w"), indent=1)
print(json.dumps({k: {c: round(v["mean_per_launch"], 1) for c, v in cs.items()} for k, cs in summary.items() if "corr_argmax" in k}, indent=1))

# DCNv2 kernels: MFMA busy fraction per kernel from the bench_dcn.py counter pass
dacc = {}
for fn in glob.glob(os.path.join(OUT, "pmc_dcn", "*counter_collection.csv")):
    for r in csv.DictReader(open(fn)):
        sname = short(r["Kernel_Name"])
        if not sname or "dcn" not in sname:
            continue
        a = dacc.setdefault(sname, {}).setdefault(r["Counter_Name"], {})
        a[r["Dispatch_Id"]] = a.get(r["Dispatch_Id"], 0.0) + float(r["Counter_Value"])
if dacc:
    dsum = {}
    for k, cs in dacc.items():
        d = {c: {"launches": len(v), "mean_per_launch": sum(v.values()) / len(v)} for c, v in cs.items()}
        if "SQ_VALU_MFMA_BUSY_CYCLES" in d and "GRBM_GUI_ACTIVE" in d:
            d["mfma_busy_fraction"] = (d["SQ_VALU_MFMA_BUSY_CYCLES"]["mean_per_launch"] / 1024.0) / (d["GRBM_GUI_ACTIVE"]["mean_per_launch"] / 8.0)
        if "SQ_WAIT_ANY" in d and "SQ_WAVE_CYCLES" in d:
            d["wait_fraction_of_wave_cycles"] = d["SQ_WAIT_ANY"]["mean_per_launch"] / d["SQ_WAVE_CYCLES"]["mean_per_launch"]
        dsum[k] = d
    json.dump(dsum, open(os.path.join(REPO, "profiles", f"{tag}_dcn_pmc_counters.json"), "w"), indent=1, sort_keys=True)
    print(json.dumps({k: {"mfma_busy": round(v.get("mfma_busy_fraction", -1), 3), "wait": round(v.get("wait_fraction_of_wave_cycles", -1), 3)} for k, v in dsum.items()}, indent=1))
