#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r03u; mkdir -p $O
for i in 1 2; do
for v in 1 0; do XPRETRAIN_LN_BWD_SIDE=$v python bench.py --no-cpu-baseline --steps 20 2>&1 | grep "^{" > $O/bench_$v.json; python -c "
import json; d=json.load(open('$O/bench_$v.json')); print('side=$v', d['value'], d['ms_per_step'], d['vit_forward_ms'])"; done; done
timeout 900 python -m pytest tests/test_layernorm_gpu.py tests/test_model_gpu.py tests/test_fullsize_parity_gpu.py tests/test_fullsize_gpu.py tests/test_determinism_gpu.py -x -q 2>&1 | tail -15 > $O/tests.txt; cat $O/tests.txt
grep -h "grad\|loss" gpurun_out/parity_log.txt | tail -80 > $O/parity_tail.txt
