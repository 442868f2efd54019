#!/usr/bin/env python3
"""Condense scripts/power_probe.sh output (rocm-smi JSON samples per labelled run) into one line per sample set."""
import json, re, sys
cur, rows = None, {}
for line in open(sys.argv[1]):
    line = line.strip()
    if line.startswith("=== "):
        cur = line[4:]; rows[cur] = []
    elif line.startswith('{"card0"') and cur:
        d = json.loads(line)["card0"]
        rows[cur].append((float(d["Current Socket Graphics Package Power (W)"]), int(re.sub(r"\D", "", d["sclk clock speed:"])),
                          float(d["Temperature (Sensor junction) (C)"])))
for k, v in rows.items():
    print(k)
    print("   samples (power W @ sclk MHz): " + "  ".join(f"{a:.0f}@{b}" for a, b, _ in v))
    busy = [x for x in v if x[0] > 600]
    if busy:
        print(f"   busy samples {len(busy)}: power mean {sum(a for a, _, _ in busy) / len(busy):.0f} W (max {max(a for a, _, _ in busy):.0f}), "
              f"sclk mean {sum(b for _, b, _ in busy) / len(busy):.0f} MHz (min {min(b for _, b, _ in busy)}, max {max(b for _, b, _ in busy)}), junction <= {max(c for _, _, c in busy):.0f} C")
