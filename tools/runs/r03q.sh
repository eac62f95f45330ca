#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r03q; mkdir -p $O
timeout 1500 python -m pytest tests/test_layernorm_gpu.py tests/test_gemm_gpu.py tests/test_model_gpu.py tests/test_dp_sim_gpu.py tests/test_distributed_gpu.py tests/test_fullsize_gpu.py -x -q > $O/pytest1.txt 2>&1; tail -4 $O/pytest1.txt
timeout 2400 python -m pytest tests/test_fullsize_parity_gpu.py -x -q -s > $O/pytest2.txt 2>&1; grep -v "maxrel=" $O/pytest2.txt | grep "full_cfg\|passed\|failed\|fp32 mode" | cut -c1-330
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep smoke
python bench.py --steps 20 --warmup 3 --no-cpu-baseline 2>&1 | grep "^{" | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('bench', d['value'], d['ms_per_step'], d['vit_forward_ms'], d['vit_forward_train_mode_ms'])"
XPRETRAIN_PROXY_FP32=0 python bench.py --steps 20 --warmup 3 --no-cpu-baseline 2>&1 | grep "^{" | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('bench without side rows', d['value'], d['ms_per_step'], d['vit_forward_ms'], d['vit_forward_train_mode_ms'])"
