#!/usr/bin/env python
"""Condense a rocprofv3 counter_collection.csv into one row per (kernel, counter): launches, mean per launch.
Usage: pmc_summarize.py counters.csv out.csv [kernel-name substring ...]   (no filter: all kernels of this library + xp_cast)"""
import collections
import csv
import sys

src, dst, keep = sys.argv[1], sys.argv[2], sys.argv[3:] or ["gemm256_kernel", "gemm_kernel", "cast_kernel", "attn_", "ln_", "splitk"]
agg = collections.defaultdict(lambda: [0, 0.0])
for r in csv.DictReader(open(src)):
    name = r["Kernel_Name"]
    if not any(k in name for k in keep):
        continue
    short = name.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:60]
    key = (short, r["Grid_Size"], r["Counter_Name"])
    agg[key][0] += 1
    agg[key][1] += float(r["Counter_Value"])
with open(dst, "w", newline="") as f:
    w = csv.writer(f)
    w.writerow(["kernel", "grid_size", "counter", "launches", "mean_per_launch"])
    for (k, g, c), (n, v) in sorted(agg.items()):
        w.writerow([k, g, c, n, f"{v / n:.6g}"])
        print(f"{k:50s} grid {g:>9s} {c:28s} n={n:3d} mean={v / n:.6g}")
