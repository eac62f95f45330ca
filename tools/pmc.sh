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
python3 - <<PY
import csv, collections
rows = list(csv.DictReader(open("$OUT/counters.csv")))
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for r in rows:
    k = r["Kernel_Name"][:70]
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
    cnt[(k, r["Counter_Name"])] += 1
for k, d in agg.items():
    print(k)
    for c, v in d.items():
        print(f"    {c:32s} total={v:.4g}  per_launch={v / cnt[(k, c)]:.4g}  launches={cnt[(k, c)]}")
PY
