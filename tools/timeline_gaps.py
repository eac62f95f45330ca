#!/usr/bin/env python
"""Idle time of the GPU inside the timed steps of a rocprofv3 --kernel-trace run of bench.py: union of kernel intervals over all
streams vs wall time between the first and the last kernel of the steady-state window, the largest gaps and what follows them.
Usage: timeline_gaps.py kernel_trace.csv [n_last_kernels_fraction]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", r.get("Stream_Id", "?"))) for r in rows))
# steady state: the window between the 40 % and 80 % marks of the run's adamw launches (a few whole steps)
ad = [e for e in ev if "adamw_kernel" in e[2]]
lo, hi = ad[len(ad) * 4 // 10][1], ad[len(ad) * 8 // 10][1]
win = [e for e in ev if lo <= e[0] and e[1] <= hi]
nsteps = sum(1 for e in win if "adamw_kernel" in e[2]) / 2
busy, cur_s, cur_e, gaps = 0, None, None, []
for s, e, name, q in win:
    if cur_e is None or s > cur_e:
        if cur_e is not None:
            busy += cur_e - cur_s
            gaps.append((s - cur_e, name))
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
wall = win[-1][1] - win[0][0]
print(f"window: {nsteps:.1f} steps, wall {wall / 1e6:.3f} ms ({wall / 1e6 / nsteps:.3f} ms/step), GPU busy (union of kernels) {busy / 1e6:.3f} ms "
      f"= {100 * busy / wall:.1f} %, idle {(wall - busy) / 1e6 / nsteps:.3f} ms/step in {len(gaps) / nsteps:.0f} gaps/step")
import collections
by = collections.defaultdict(lambda: [0, 0])
for g, name in gaps:
    k = name.replace("(anonymous namespace)::", "").replace("void ", "")[:60]
    by[k][0] += 1; by[k][1] += g
print("idle time by the kernel that FOLLOWS the gap (us per step):")
for k, (n, t) in sorted(by.items(), key=lambda kv: -kv[1][1])[:14]:
    print(f"  {t / 1e3 / nsteps:8.1f} us  {n / nsteps:6.1f} gaps  avg {t / n / 1e3:5.2f} us  {k}")
