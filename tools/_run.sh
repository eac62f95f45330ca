mkdir -p gpurun_out/r02 gpurun_out/determinism
timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | tail -4 > gpurun_out/r02/pytest_gpu_full.txt; cat gpurun_out/r02/pytest_gpu_full.txt
timeout 600 python tools/determinism_hunt.py --setting asis --iters 50 --mode none > gpurun_out/determinism/loop50_asis.txt 2>&1; tail -1 gpurun_out/determinism/loop50_asis.txt
timeout 600 python bench.py > gpurun_out/r02/bench_default_b.json 2>/dev/null; python -c "
import json
d=json.load(open('gpurun_out/r02/bench_default_b.json')); print({k:d[k] for k in ('value','ms_per_step','host_enqueue_ms_per_step','vit_forward_ms','vit_forward_frac_of_bf16_peak')}); print(d['roofline']); print(d['cpu_baseline']['value'])"
