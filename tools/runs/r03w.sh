#!/bin/bash
R=${GRAFT_REPO_ROOT:-.}
cd $R; O=$R/gpurun_out/r03w; mkdir -p $O
export TMPDIR=/tmp
prof() { ( cd $1; rm -rf /tmp/rp_$2; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rp_$2 -o $2 -- python tools/step_only.py 10 2>&1 | grep step_only; find /tmp/rp_$2 -name "*kernel_stats*.csv" -exec cp {} $O/$2_kernel_stats.csv \; ) }
( cd $R/_ab_r2 && python tools/step_only.py 20 ); ( cd $R && python tools/step_only.py 20 )
prof $R/_ab_r2 r2
prof $R head
