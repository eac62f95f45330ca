#!/bin/bash
# Usage (GPU box): bash tools/pmc.sh <tag> "<counters space separated>" <python script + args>
# Collects hardware counters in their own run (no trace domains besides kernel-trace).
TAG=$1; CTRS=$2; shift 2
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pmc_$TAG
( cd $REPO && timeout 600 rocprofv3 --kernel-trace --pmc $CTRS --output-format csv -d /tmp/pmc_$TAG -o $TAG -- python "$@" ) > $OUT/run.log 2>&1
tail -3 $OUT/run.log
find /tmp/pmc_$TAG -name "*counter_collection.csv" -exec cp {} $OUT/counters.csv \;
find /tmp/pmc_$TAG -name "*kernel_trace.csv" -exec cp {} $OUT/kernel_trace.csv \;
python3 - <<PY
import csv, collections, os
rows = list(csv.DictReader(open("$OUT/counters.csv")))
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for r in rows:
    k = r["Kernel_Name"][:70]
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
    cnt[(k, r["Counter_Name"])] += 1
# durations of the SAME (counter-collecting) run: a profiled pass is slower than an un-profiled one, so counter / wall ratios
# (e.g. the shader clock GRBM_GUI_ACTIVE / 8 XCDs / duration) must use these, not the kernel-trace-only numbers
dur = collections.defaultdict(list)
if os.path.isfile("$OUT/kernel_trace.csv"):
    for r in csv.DictReader(open("$OUT/kernel_trace.csv")):
        dur[r["Kernel_Name"][:70]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, d in agg.items():
    us = sorted(dur.get(k, []))
    med = us[len(us) // 2] if us else float("nan")
    print(k, f"  [median duration in this run {med:.1f} us over {len(us)} launches]")
    for c, v in d.items():
        per = v / cnt[(k, c)]
        extra = f"  -> {per / 8 / med / 1e3:.3f} GHz (per-launch / 8 XCDs / median duration)" if c == "GRBM_GUI_ACTIVE" and us else ""
        print(f"    {c:32s} total={v:.4g}  per_launch={per:.4g}  launches={cnt[(k, c)]}{extra}")
PY
