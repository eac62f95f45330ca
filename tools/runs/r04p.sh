#!/bin/bash
# round 4, last call: quick parity subset, PMC passes of the GEMM family on the FINAL sources (stamp for roofline.traffic), final default bench
R=${GRAFT_REPO_ROOT:-.}
cd $R; O=$R/gpurun_out/r04p; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gemm_gpu.py tests/test_model_gpu.py tests/test_layernorm_gpu.py tests/test_fullsize_gpu.py -x -q > $O/pytest_part.log 2>&1; echo "pytest exit code $?" | tee $O/pytest_part.txt; grep -E "passed|failed|error" $O/pytest_part.log | tail -3 | tee -a $O/pytest_part.txt; grep -E "^(FAILED|ERROR)|Error|assert" $O/pytest_part.log | head -20
bash tools/pmc_gemm256.sh r04p > $O/pmc_gemm256.log 2>&1; tail -1 $O/pmc_gemm256.log | cut -c1-200
cp $R/gpurun_out/r04p_pmc_gemm256.json $R/profiles/r04p_pmc_gemm256.json
timeout 300 python bench.py --steps 20 --warmup 5 2>&1 | grep "^{" > $O/bench_default.json; python -c "
import json; d=json.load(open('$O/bench_default.json')); r=d['roofline']; print('bench', d['value'], d['ms_per_step'], d['vit_forward_train_mode_ms'], d['vit_forward_ms'], 'roofline', r['frac'], r['kernel_ms'], r['traffic'], d['roofline_bwd']['traffic'], d['cpu_baseline']['value'])"
