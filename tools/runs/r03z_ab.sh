#!/bin/bash
R=${GRAFT_REPO_ROOT:-.}
cd $R; O=$R/gpurun_out/r03z_ab; mkdir -p $O
run() { python bench.py --no-cpu-baseline --steps 20 2>&1 | grep "^{" > $O/$1.json; python -c "
import json; d=json.load(open('$O/$1.json')); print('$1', d['value'], d['ms_per_step'], d.get('vit_forward_ms'), d.get('vit_forward_train_mode_ms'))"; }
for i in 1 2 3; do
run default_$i
XPRETRAIN_ATTN_FWD3=0 run nofwd3_$i
XPRETRAIN_GEMM_NO_XCD=1 run noxcd_$i
XPRETRAIN_GEMM256_STAGED=3 run mask3_$i
done
