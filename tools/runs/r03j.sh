#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r03j; mkdir -p $O
export TMPDIR=/tmp
rm -rf /tmp/rp_tl
( timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/rp_tl -o tl -- python bench.py --steps 10 --warmup 3 --no-cpu-baseline ) > $O/tl_run.log 2>&1
F=$(find /tmp/rp_tl -name "*kernel_trace.csv" | head -1)
python tools/timeline_gaps.py $F | tee $O/timeline_gaps.txt
python bench.py --steps 10 --warmup 3 --no-cpu-baseline --frames 8 --res 448 2>&1 | grep "^{" > $O/cfg3_bench.json; cut -c1-400 $O/cfg3_bench.json
python bench.py --steps 10 --warmup 3 --no-cpu-baseline --frames 32 2>&1 | grep "^{" > $O/cfg4_bench.json; cut -c1-400 $O/cfg4_bench.json
bash tools/profile.sh r03j_cfg3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --frames 8 --res 448 > $O/cfg3_kernels.txt 2>&1
bash tools/profile.sh r03j_cfg4 bench.py --steps 5 --warmup 2 --no-cpu-baseline --frames 32 > $O/cfg4_kernels.txt 2>&1
