#!/usr/bin/env python3
"""Per-kernel means of the rocprofv3 --pmc counter CSVs under a directory: python scripts/pmc_kernel.py <dir> [name filter]"""
import csv, glob, os, re, sys
csv.field_size_limit(1 << 30)
d, flt = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "c2m::")
acc = {}
for fn in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(fn)):
        k = r["Kernel_Name"]
        if flt not in k:
            continue
        m = re.search(r"(c2m::[A-Za-z0-9_:]+(<[^>(]*>)?)", k)
        k = m.group(1) if m else k[:60]
        a = acc.setdefault(k, {}).setdefault(r["Counter_Name"], {})
        a[r["Dispatch_Id"]] = a.get(r["Dispatch_Id"], 0.0) + float(r["Counter_Value"])
for k, cs in acc.items():
    print(k)
    for c, v in sorted(cs.items()):
        print(f"   {c:34s} launches {len(v):4d}  mean {sum(v.values()) / len(v):16.1f}")
