#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r03i; mkdir -p $O
timeout 900 python -m pytest tests/test_attention_gpu.py -x -q > $O/pytest_attn.txt 2>&1; tail -4 $O/pytest_attn.txt
for i in 1 2; do
  XPRETRAIN_ATTN_FWD3=0 python tools/bench_kernels.py attn 2>&1 | grep "attn fwd  B8 H12" | sed 's/^/old: /'
  python tools/bench_kernels.py attn 2>&1 | grep "attn fwd  B8 H12" | sed 's/^/new: /'
done | tee $O/bench_attn.txt
XPRETRAIN_ATTN_FWD3=0 python tools/fwd_only.py 10 12 224 infer 2>&1 | grep vit | sed 's/^/old: /' | tee $O/fwd.txt
python tools/fwd_only.py 10 12 224 infer 2>&1 | grep vit | sed 's/^/new: /' | tee -a $O/fwd.txt
python tools/fwd_only.py 10 32 224 infer 2>&1 | grep vit | sed 's/^/new T=32: /' | tee -a $O/fwd.txt
