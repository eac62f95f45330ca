#!/bin/bash
# round 4, call 4: full GPU suite (exit code recorded), LayerNorm forward half-wave kernel A/B (isolated + in-step), final default bench
R=${GRAFT_REPO_ROOT:-.}
cd $R; O=$R/gpurun_out/r04d; mkdir -p $O
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "pytest exit code $?" | tee $O/pytest_gpu.txt; grep -E "passed|failed|error" $O/pytest_gpu.log | tail -3 | tee -a $O/pytest_gpu.txt; grep -E "^(FAILED|ERROR)|Error|assert" $O/pytest_gpu.log | head -20
for v in 1 0; do XPRETRAIN_LN_HALFWAVE=$v timeout 200 python tools/bench_kernels.py all 2>&1 | grep -i "layernorm" | sed "s/^/[XPRETRAIN_LN_HALFWAVE=$v] /" | tee -a $O/ln_ab.txt; done
timeout 900 python tools/instep_ab.py --rounds 3 --steps 20 --out $O/instep_ab_ln.txt default ln0:XPRETRAIN_LN_HALFWAVE=0 2>&1 | tail -4
timeout 300 python bench.py --steps 20 --warmup 5 2>&1 | grep "^{" > $O/bench_default.json; python -c "
import json; d=json.load(open('$O/bench_default.json')); print('bench', d['value'], d['ms_per_step'], d['vit_forward_train_mode_ms'], d['vit_forward_ms'], d['roofline']['frac'], d['roofline']['kernel_ms'], d['roofline']['traffic'], d['roofline_bwd']['traffic']); print(d.get('cpu_baseline'))"
