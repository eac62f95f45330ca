#!/bin/bash
# round 4, call 7: full GPU suite on the final tree (configs[4] teacher-forced on all layers), default bench with the in-step roofline timing,
# rocprofv3 --kernel-trace --stats of the same bench command (the summary the roofline's kernel_ms must agree with)
R=${GRAFT_REPO_ROOT:-.}
cd $R; O=$R/gpurun_out/r04g; mkdir -p $O
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "pytest exit code $?" | tee $O/pytest_gpu.txt; grep -E "passed|failed|error" $O/pytest_gpu.log | tail -3 | tee -a $O/pytest_gpu.txt; grep -E "^(FAILED|ERROR)|Error|assert" $O/pytest_gpu.log | head -20
timeout 300 python bench.py --steps 20 --warmup 5 2>&1 | grep "^{" > $O/bench_default.json; python -c "
import json; d=json.load(open('$O/bench_default.json')); r=d['roofline']; print('bench', d['value'], d['ms_per_step'], d['vit_forward_train_mode_ms'], d['vit_forward_frac_of_bf16_peak'], 'roofline', r['frac'], r['kernel_ms'], r['kernel_ms_isolated'], r['frac_isolated'], r['kernel_ms_source'][:40])"
rm -rf /tmp/rp_b; ( timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rp_b -o b -- python bench.py --steps 10 --warmup 3 --no-cpu-baseline ) > $O/bench_rocprof.log 2>&1
find /tmp/rp_b -name "*kernel_stats*.csv" -exec cp {} $O/bench_kernel_stats.csv \;
grep "^{" $O/bench_rocprof.log | head -1 > $O/bench_under_rocprof.json
python3 - <<PY
import csv
rows = list(csv.DictReader(open("$O/bench_kernel_stats.csv")))
print(f"{'calls':>6} {'avg_us':>9} {'tot_ms':>8} {'%':>6}  kernel")
for r in rows[:24]:
    print(f"{int(r['Calls']):6d} {float(r['AverageNs'])/1e3:9.1f} {float(r['TotalDurationNs'])/1e6:8.2f} {float(r['Percentage']):6.2f}  {r['Name'][:110]}")
PY
