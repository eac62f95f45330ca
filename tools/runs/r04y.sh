#!/bin/bash
# round 4, final validation after the forward chains and the data-parallel fixes: full GPU suite, smoke(), default bench, and the same
# bench through a one-rank RCCL group
R=${GRAFT_REPO_ROOT:-.}
cd $R; O=$R/gpurun_out/r04y; mkdir -p $O
export TMPDIR=/tmp
timeout 1800 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "pytest exit code $?" | tee $O/pytest_gpu.txt; grep -E "passed|failed|error" $O/pytest_gpu.log | tail -3 | tee -a $O/pytest_gpu.txt; grep -E "^(FAILED|ERROR)|Error|assert" $O/pytest_gpu.log | head -20
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('__SMOKE_OK__')" 2>&1 | grep -v amdgpu.ids | tail -4 | tee $O/smoke.txt
timeout 300 python bench.py --steps 20 --warmup 5 2>&1 | grep "^{" > $O/bench_default.json; python -c "
import json; d=json.load(open('$O/bench_default.json')); r=d['roofline']; print('bench', d['value'], d['ms_per_step'], d['vit_forward_train_mode_ms'], d['vit_forward_ms'], 'roofline', r['frac'], r['kernel_ms'], r['traffic'], d['roofline_bwd']['traffic'], d['cpu_baseline']['value'], d['config']['second_chain_stream'])"
XPRETRAIN_BENCH_FORCE_COLLECTIVES=1 timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>&1 | grep "^{" > $O/bench_forced_one_rank_collectives.json; python -c "
import json; d=json.load(open('$O/bench_forced_one_rank_collectives.json')); print('forced', d['value'], d['ms_per_step'], d['config']['forced_one_rank_collectives'])"
