#!/bin/bash
# Usage (on the GPU box, via gpurun):  bash tools/profile.sh <tag> [bench args...]
# Runs bench.py under rocprofv3 --kernel-trace --stats and copies the per-kernel summary to gpurun_out/<tag>/.
set -u
TAG=${1:-prof}; shift || true
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/rp_$TAG
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rp_$TAG -o $TAG -- \
    python $REPO/bench.py --no-cpu-baseline "$@" > $OUT/bench_under_rocprof.log 2>&1
find /tmp/rp_$TAG -name "*kernel_stats*.csv" -exec cp {} $OUT/kernel_stats.csv \;
find /tmp/rp_$TAG -name "*domain_stats*.csv" -exec cp {} $OUT/domain_stats.csv \;
ls -la /tmp/rp_$TAG/* | head -20
tail -2 $OUT/bench_under_rocprof.log
head -40 $OUT/kernel_stats.csv
