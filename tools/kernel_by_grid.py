#!/usr/bin/env python
"""One row per (kernel, grid size) from a rocprofv3 --kernel-trace CSV: launches, median / mean / min duration.

The 256-wide GEMM family serves every video-tower shape with one kernel per operand layout, so the per-kernel averages of
`--stats` mix four shapes and, in the two-chain forward, half- and full-batch launches.  The grid size separates them: at cfg #2
the NT kernel runs 888 workgroups for fc1 (444 per half-batch chain), 666 / 333 for qkv, 222 / 111 for out-proj and fc2 -- so
`bench.py`'s `roofline.kernel_ms` (HIP events around the fc1 launches) can be checked against a rocprof duration.

    python tools/kernel_by_grid.py <kernel_trace.csv> [name substring ...]  > profiles/rNN_kernel_by_grid.txt
"""
import collections
import csv
import statistics
import sys


def col(row, *names):
    for n in names:
        if n in row and row[n] != "":
            return row[n]
    raise KeyError(names)


def summarize(path, keep=()):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        name = col(r, "Kernel_Name", "Name")
        if keep and not any(k in name for k in keep):
            continue
        gx = int(col(r, "Grid_Size_X", "Grid_Size"))
        wx = int(r.get("Workgroup_Size_X") or r.get("Workgroup_Size") or 1)
        gy, gz = int(r.get("Grid_Size_Y") or 1), int(r.get("Grid_Size_Z") or 1)
        wy, wz = int(r.get("Workgroup_Size_Y") or 1), int(r.get("Workgroup_Size_Z") or 1)
        wgs = (gx // max(wx, 1)) * (gy // max(wy, 1)) * (gz // max(wz, 1))
        dur = (int(col(r, "End_Timestamp")) - int(col(r, "Start_Timestamp"))) / 1e3          # us
        short = name.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
        agg[(short[:64], wgs, wx * wy * wz)].append(dur)
    rows = []
    for (k, wgs, wsz), d in agg.items():
        rows.append((sum(d), k, wgs, wsz, len(d), statistics.median(d), sum(d) / len(d), min(d)))
    rows.sort(reverse=True)
    return rows


def main():
    if len(sys.argv) < 2:
        raise SystemExit(__doc__)
    rows = summarize(sys.argv[1], sys.argv[2:])
    print(f"{'kernel':64s} {'workgroups':>10s} {'threads':>7s} {'launches':>8s} {'median_us':>10s} {'mean_us':>9s} {'min_us':>8s} {'total_ms':>9s}")
    for tot, k, wgs, wsz, n, med, mean, mn in rows:
        print(f"{k:64s} {wgs:10d} {wsz:7d} {n:8d} {med:10.1f} {mean:9.1f} {mn:8.1f} {tot / 1e3:9.2f}")


if __name__ == "__main__":
    main()
