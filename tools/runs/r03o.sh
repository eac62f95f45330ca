#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r03o; mkdir -p $O
timeout 2400 python -m pytest tests/test_gemm_gpu.py tests/test_model_gpu.py tests/test_layernorm_gpu.py -x -q > $O/pytest.txt 2>&1; tail -4 $O/pytest.txt
for i in 1 2; do python bench.py --steps 20 --warmup 3 --no-cpu-baseline 2>&1 | grep "^{" | tee $O/bench_$i.json | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['vit_forward_ms'], d['vit_forward_train_mode_ms'], d['roofline']['frac'])"; done
