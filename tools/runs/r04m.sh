#!/bin/bash
# round 4, call 14: multi-group persistent attention forward (R > 208) -- tests, configs[3] bench + kernel table against the 7-wave kernel
R=${GRAFT_REPO_ROOT:-.}
cd $R; O=$R/gpurun_out/r04m; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_attention_gpu.py tests/test_determinism_gpu.py -x -q > $O/pytest_part.log 2>&1; echo "pytest exit code $?" | tee $O/pytest_part.txt; grep -E "passed|failed|error" $O/pytest_part.log | tail -3 | tee -a $O/pytest_part.txt; grep -E "^(FAILED|ERROR)|Error|assert" $O/pytest_part.log | head -20
for v in 1 0; do XPRETRAIN_ATTN_FWD3=$v timeout 300 python bench.py --no-cpu-baseline --frames 8 --res 448 --steps 8 2>&1 | grep "^{" > $O/bench_cfg3_fwd3_$v.json; python -c "
import json; d=json.load(open('$O/bench_cfg3_fwd3_$v.json')); print('cfg3 XPRETRAIN_ATTN_FWD3=$v', d['value'], d['ms_per_step'], d['vit_forward_train_mode_ms'])"; done
for v in 1 0; do XPRETRAIN_ATTN_FWD3=$v timeout 300 python bench.py --no-cpu-baseline --frames 8 --res 448 --steps 8 2>&1 | grep "^{" > $O/bench_cfg3_fwd3_${v}_b.json; python -c "
import json; d=json.load(open('$O/bench_cfg3_fwd3_${v}_b.json')); print('cfg3 XPRETRAIN_ATTN_FWD3=$v (2nd)', d['value'], d['ms_per_step'], d['vit_forward_train_mode_ms'])"; done
rm -rf /tmp/rp_c3; ( timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rp_c3 -o c3 -- python bench.py --no-cpu-baseline --frames 8 --res 448 --steps 4 --warmup 2 ) > $O/cfg3_rocprof.log 2>&1
find /tmp/rp_c3 -name "*kernel_stats*.csv" -exec cp {} $O/cfg3_kernel_stats.csv \;
python3 - <<PY
import csv
rows = list(csv.DictReader(open("$O/cfg3_kernel_stats.csv")))
print(f"{'calls':>6} {'avg_us':>9} {'tot_ms':>8} {'%':>6}  kernel")
for r in rows[:14]:
    print(f"{int(r['Calls']):6d} {float(r['AverageNs'])/1e3:9.1f} {float(r['TotalDurationNs'])/1e6:8.2f} {float(r['Percentage']):6.2f}  {r['Name'][:100]}")
PY
