#!/bin/bash
# (historical run script: XPRETRAIN_WGRAD_PRIORITY and XPRETRAIN_FWD_CHAINS were experiment switches of that moment -- the weight-gradient stream now has
# the default priority and there are exactly two forward chains; the tree no longer reads them)
R=${GRAFT_REPO_ROOT:-.}
cd $R; O=$R/gpurun_out/r04w; mkdir -p $O
export TMPDIR=/tmp
for mode in forced plain; do
rm -rf /tmp/prof_fc
if [ $mode = forced ]; then export XPRETRAIN_BENCH_FORCE_COLLECTIVES=1; else unset XPRETRAIN_BENCH_FORCE_COLLECTIVES; fi
XPRETRAIN_WGRAD_PRIORITY=normal XPRETRAIN_FWD_SPLIT_STREAM=side timeout 400 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_fc -o p -- python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/prof_$mode.log 2>&1
grep "^{" $O/prof_$mode.log | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$mode under rocprof', d['value'], d['ms_per_step'])"
f=$(find /tmp/prof_fc -name "*kernel_trace.csv" | head -1)
python tools/timeline_gaps.py $f | tee $O/${mode}2_timeline_gaps.txt | head -12
python - "$f" > $O/${mode}2_one_step_timeline.txt <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")[:48], r.get("Queue_Id", "?")) for r in rows)
ad = [i for i, e in enumerate(ev) if "adamw_kernel" in e[2]]
i0, i1 = ad[len(ad) // 2 - 2], ad[len(ad) // 2]      # one step: between two optimizer launches
t0 = ev[i0][1]
last_end = {}
for s, e, n, q in ev[i0 + 1:i1 + 1]:
    gap = s - last_end.get(q, t0)
    print(f"{(s - t0) / 1e3:9.1f} us  q{q:>3s}  dur {(e - s) / 1e3:7.1f}  gap on this queue {gap / 1e3:8.1f}  {n}")
    last_end[q] = e
PY
done
