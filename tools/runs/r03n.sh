#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r03n; mkdir -p $O
run() { python bench.py --steps 20 --warmup 3 --no-cpu-baseline 2>&1 | grep "^{" | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1', d['value'], d['ms_per_step'], d['vit_forward_ms'], d['roofline']['kernel_ms'], d['roofline_bwd']['kernel_ms'])"; }
for i in 1 2 3; do
  run "fwd224 dx224"
  XPRETRAIN_GEMM256_MT1_NS=4 run "fwd224 dx256"
  XPRETRAIN_GEMM256_MT1=4 XPRETRAIN_GEMM256_MT1_NS=3 run "fwd256 dx224"
  XPRETRAIN_GEMM256_MT1=4 run "fwd256 dx256"
done | tee $O/bench_ab_tiles.txt
