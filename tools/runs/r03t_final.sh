#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r03t; mkdir -p $O
python bench.py 2>&1 | grep "^{" > $O/bench_default.json; python -c "
import json; d=json.load(open('$O/bench_default.json'))
print({k: d[k] for k in ('value','ms_per_step','vit_forward_ms','vit_forward_train_mode_ms','vit_forward_frac_of_bf16_peak','step_tflops_per_gpu','host_cpu_ms_per_step')})
print(d['roofline']['frac'], d['roofline']['kernel_ms'], d['roofline']['traffic'], d['roofline_bwd']['frac'], d['cpu_baseline']['value'])"
bash tools/profile.sh r03t_bench bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_kernel_table.txt 2>&1; head -24 $O/bench_kernel_table.txt | cut -c1-160
python tools/bench_kernels.py all > $O/bench_kernels.txt 2>&1
