#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r03m; mkdir -p $O
timeout 1500 python -m pytest tests/test_layernorm_gpu.py tests/test_attention_gpu.py tests/test_model_gpu.py tests/test_determinism_gpu.py tests/test_fullsize_gpu.py -x -q > $O/pytest.txt 2>&1; tail -4 $O/pytest.txt
for i in 1 2; do
  python bench.py --steps 20 --warmup 3 --no-cpu-baseline 2>&1 | grep "^{" | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('default        ', d['value'], d['ms_per_step'], d['vit_forward_ms'])"
  XPRETRAIN_ATTN_FWD3=0 python bench.py --steps 20 --warmup 3 --no-cpu-baseline 2>&1 | grep "^{" | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('old attn fwd   ', d['value'], d['ms_per_step'], d['vit_forward_ms'])"
  XPRETRAIN_GEMM256_PERSIST=0 python bench.py --steps 20 --warmup 3 --no-cpu-baseline 2>&1 | grep "^{" | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('no persist gemm', d['value'], d['ms_per_step'], d['vit_forward_ms'])"
  XPRETRAIN_GEMM256_MT1=4 python bench.py --steps 20 --warmup 3 --no-cpu-baseline 2>&1 | grep "^{" | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('256-row tiles  ', d['value'], d['ms_per_step'], d['vit_forward_ms'])"
done | tee $O/bench_ab.txt
