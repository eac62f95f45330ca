#!/bin/bash
R=${GRAFT_REPO_ROOT:-.}
cd $R; O=$R/gpurun_out/r03v; mkdir -p $O
run() { ( cd $1 && python bench.py --no-cpu-baseline --steps 20 2>&1 | grep "^{" > $O/$2.json; python -c "
import json; d=json.load(open('$O/$2.json')); print('$2', d['value'], d['ms_per_step'], d.get('vit_forward_ms'))" ); }
for i in 1 2; do
run $R/_ab_r2 r2_$i
XPRETRAIN_OVERLAP_TEXT=0 run $R/_ab_r2 r2_nooverlap_$i
run $R head_$i
XPRETRAIN_OVERLAP_TEXT=0 run $R head_nooverlap_$i
XPRETRAIN_GEMM256_PERSIST=0 run $R head_nopersist_$i
XPRETRAIN_ATTN_FWD3=0 run $R head_nofwd3_$i
XPRETRAIN_GEMM256_PERSIST=0 XPRETRAIN_ATTN_FWD3=0 XPRETRAIN_PROXY_FP32=0 run $R head_alloff_$i
done
