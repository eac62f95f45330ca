#!/usr/bin/env python
"""ONE steady-state training step of a rocprofv3 --kernel-trace run (tools/step_only.py or bench.py --profile-run) as a timeline:
one line per kernel -- start offset from the step's first kernel (us), duration, queue, workgroups, short name -- plus a summary of
the step's phases as the trace shows them (first / last kernel of the video tower's forward layers, the backward layers, the optimizer).

    python tools/step_timeline.py <kernel_trace.csv> [--step K] [--min-us D] > profiles/rNN_step_timeline.txt

The step boundaries are the ends of the LAST adamw_kernel launch of each step (a step has 3 such launches, or 2 + late ones)."""
import argparse
import csv


def short(name):
    n = name.replace("(anonymous namespace)::", "").replace("void ", "")
    if n.startswith("_ZN12_GLOBAL__N_1"):
        n = n[len("_ZN12_GLOBAL__N_1"):].lstrip("0123456789")
    return n.split("(")[0][:44]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("trace")
    ap.add_argument("--step", type=int, default=-3, help="which step (index into the steps found; negative from the end)")
    ap.add_argument("--min-us", type=float, default=0.0, help="hide kernels shorter than this")
    a = ap.parse_args()
    ev = []
    for r in csv.DictReader(open(a.trace)):
        gx, wx = int(r["Grid_Size_X"]), int(r["Workgroup_Size_X"])
        gy, gz = int(r.get("Grid_Size_Y") or 1), int(r.get("Grid_Size_Z") or 1)
        wy, wz = int(r.get("Workgroup_Size_Y") or 1), int(r.get("Workgroup_Size_Z") or 1)
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"]), r.get("Queue_Id", "?"),
                   (gx // wx) * (gy // max(wy, 1)) * (gz // max(wz, 1))))
    ev.sort()
    # step boundary: the sqnorm_partials launches open the optimizer of a step; the step ends with the last adamw of that group
    ad = [i for i, e in enumerate(ev) if e[2].startswith("adamw_kernel")]
    groups, cur = [], [ad[0]]
    for i in ad[1:]:
        if ev[i][0] - ev[cur[-1]][1] > 3_000_000:        # > 3 ms apart: another step
            groups.append(cur); cur = [i]
        else:
            cur.append(i)
    groups.append(cur)
    ends = [max(ev[i][1] for i in g) for g in groups]
    k = a.step if a.step >= 0 else len(ends) + a.step
    lo, hi = ends[k - 1], ends[k]
    win = [e for e in ev if e[0] >= lo - 2_000_000 and e[0] < hi]       # (late adamw launches of the previous step overlap this one)
    win = [e for e in win if e[1] > lo or not e[2].startswith("adamw")]
    win = [e for e in win if e[0] >= lo or e[1] > lo]
    t0 = min(e[0] for e in win if e[0] >= lo)
    qs = {q: i for i, q in enumerate(sorted({e[3] for e in win}))}
    print(f"# step {k} of {len(ends)}: {(hi - lo) / 1e6:.3f} ms between optimizer ends; {len(win)} kernels; queues {qs}")
    busy = {q: 0 for q in qs}
    for s, e, n, q, wg in win:
        busy[q] += e - s
        if (e - s) / 1e3 >= a.min_us:
            print(f"{(s - t0) / 1e3:9.1f} {(e - s) / 1e3:7.1f}  q{qs[q]} {'  ' * qs[q]}{wg:5d} {n}")
    print("# busy per queue (ms):", {f"q{qs[q]}": round(b / 1e6, 3) for q, b in busy.items()})


if __name__ == "__main__":
    main()
