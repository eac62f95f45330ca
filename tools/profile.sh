#!/bin/bash
# Usage (on the GPU box, via gpurun):  bash tools/profile.sh <tag> <script.py> [args...]
# Runs the script under rocprofv3 --kernel-trace --stats and copies the per-kernel summary to gpurun_out/<tag>/.
set -u
TAG=${1:-prof}; shift || true
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
rm -rf /tmp/rp_$TAG
( cd $REPO && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rp_$TAG -o $TAG -- python "$@" ) > $OUT/run.log 2>&1
find /tmp/rp_$TAG -name "*kernel_stats*.csv" -exec cp {} $OUT/kernel_stats.csv \;
grep -v "^W20\|^E20\|amdgpu.ids" $OUT/run.log | tail -4
python3 - <<PY
import csv
rows = list(csv.DictReader(open("$OUT/kernel_stats.csv")))
print(f"{'calls':>6} {'avg_us':>9} {'tot_ms':>8} {'%':>6}  kernel")
for r in rows[:28]:
    print(f"{int(r['Calls']):6d} {float(r['AverageNs'])/1e3:9.1f} {float(r['TotalDurationNs'])/1e6:8.2f} {float(r['Percentage']):6.2f}  {r['Name'][:110]}")
PY
