#!/bin/bash
# round 4, call 2: full GPU suite on the new defaults, split-batch probe, vendor kernel names, RCCL-footprint contention with a CU budget
R=${GRAFT_REPO_ROOT:-.}
cd $R; O=$R/gpurun_out/r04b; mkdir -p $O
export TMPDIR=/tmp
( timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -4 ) > $O/pytest_gpu.txt; cat $O/pytest_gpu.txt
timeout 300 python tools/split_batch_probe.py --steps 10 2>&1 | grep -v amdgpu.ids | tee $O/split_batch_probe.txt
XPRETRAIN_WGRAD_STREAM=0 timeout 300 python tools/split_batch_probe.py --steps 10 2>&1 | grep -v amdgpu.ids | grep "training step" | sed 's/^/[wgrad stream off] /' | tee -a $O/split_batch_probe.txt
for b in 256 240 224; do XPRETRAIN_CU_BUDGET=$b timeout 200 python tools/contention_probe.py 10 fat 16,32 2>&1 | grep -v amdgpu.ids | tee -a $O/contention_cu_budget.txt; done
rm -rf /tmp/rp_v; ( timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rp_v -o v -- python tools/vendor_instep.py --steps 2 --order after ) > $O/vendor_rocprof.log 2>&1
find /tmp/rp_v -name "*kernel_stats*.csv" -exec cp {} $O/vendor_kernel_stats.csv \;
python3 - <<PY
import csv
rows = list(csv.DictReader(open("$O/vendor_kernel_stats.csv")))
for r in rows[:40]:
    print(f"{int(r['Calls']):6d} {float(r['AverageNs'])/1e3:9.1f} {float(r['TotalDurationNs'])/1e6:8.2f}  {r['Name'][:160]}")
PY
