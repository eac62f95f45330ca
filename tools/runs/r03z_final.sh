#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r03z; mkdir -p $O
timeout 1500 python -m pytest tests/ -x -q -m gpu 2>&1 | tail -4 > $O/tests.txt; cat $O/tests.txt
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
python bench.py 2>&1 | grep "^{" > $O/bench_default.json; python -c "
import json; d=json.load(open('$O/bench_default.json'))
print({k: d[k] for k in ('value','ms_per_step','vit_forward_ms','vit_forward_train_mode_ms','vit_forward_frac_of_bf16_peak','step_tflops_per_gpu','host_cpu_ms_per_step')})
print(d['roofline']['frac'], d['roofline']['kernel_ms'], d['roofline']['traffic'], d['roofline_bwd']['frac'], d['roofline_bwd']['traffic'], d['cpu_baseline']['value'])"
bash tools/profile.sh r03z_bench bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_kernel_table.txt 2>&1; head -12 $O/bench_kernel_table.txt | cut -c1-150
python tools/bench_kernels.py all > $O/bench_kernels.txt 2>&1
