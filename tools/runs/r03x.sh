#!/bin/bash
R=${GRAFT_REPO_ROOT:-.}
cd $R; O=$R/gpurun_out/r03x; mkdir -p $O
run() { ( cd $1 && python bench.py --no-cpu-baseline --steps 20 2>&1 | grep "^{" > $O/$2.json; python -c "
import json; d=json.load(open('$O/$2.json')); print('$2', d['value'], d['ms_per_step'], d.get('vit_forward_ms'), d.get('vit_forward_train_mode_ms'))" ); }
for i in 1 2; do
run $R/_ab_r2 r2_$i
XPRETRAIN_GEMM256_STAGED=0 run $R m0_$i
XPRETRAIN_GEMM256_STAGED=0 XPRETRAIN_GEMM256_PERSIST=0 run $R m0_nopersist_$i
XPRETRAIN_GEMM256_STAGED=1 run $R m1_$i
XPRETRAIN_GEMM256_STAGED=2 run $R m2_$i
XPRETRAIN_GEMM256_STAGED=4 run $R m4_$i
XPRETRAIN_GEMM256_STAGED=3 run $R m3_$i
XPRETRAIN_GEMM256_STAGED=7 run $R m7_$i
done
XPRETRAIN_GEMM256_STAGED=7 timeout 600 python -m pytest tests/test_gemm_gpu.py tests/test_model_gpu.py -x -q 2>&1 | tail -5
