#!/bin/bash
# rocprofv3 --kernel-trace --stats of the default bench command on the final tree (the summary the roofline numbers are read against)
R=${GRAFT_REPO_ROOT:-.}
cd $R; O=$R/gpurun_out/r04zz; mkdir -p $O
export TMPDIR=/tmp
rm -rf /tmp/prof_b
timeout 100 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_b -o p -- python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_under_rocprof.log 2>&1
grep "^{" $O/bench_under_rocprof.log > $O/bench_under_rocprof.json
cp $(find /tmp/prof_b -name "*kernel_stats.csv" | head -1) $O/bench_kernel_stats.csv
head -5 $O/bench_kernel_stats.csv | cut -c1-160
