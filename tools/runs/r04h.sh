#!/bin/bash
# round 4, call 8: patch-gather tests, inference-forward A/B of the gather, PMC passes on the final GEMM sources, final bench line
R=${GRAFT_REPO_ROOT:-.}
cd $R; O=$R/gpurun_out/r04h; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_embed_loss_gpu.py tests/test_model_gpu.py tests/test_gemm_gpu.py -x -q > $O/pytest_part.log 2>&1; echo "pytest exit code $?" | tee $O/pytest_part.txt; grep -E "passed|failed|error" $O/pytest_part.log | tail -3 | tee -a $O/pytest_part.txt; grep -E "^(FAILED|ERROR)|Error|assert" $O/pytest_part.log | head -20
timeout 600 python tools/instep_ab.py --rounds 2 --steps 10 --out $O/instep_ab_patch_gather.txt default nogather:XPRETRAIN_PATCH_GATHER=0 2>&1 | tail -4
bash tools/pmc_gemm256.sh r04h > $O/pmc_gemm256.log 2>&1; tail -2 $O/pmc_gemm256.log | cut -c1-300
cp $R/gpurun_out/r04h_pmc_gemm256.json $R/profiles/r04h_pmc_gemm256.json
timeout 300 python bench.py --steps 20 --warmup 5 2>&1 | grep "^{" > $O/bench_default.json; python -c "
import json; d=json.load(open('$O/bench_default.json')); r=d['roofline']; print('bench', d['value'], d['ms_per_step'], d['vit_forward_train_mode_ms'], d['vit_forward_ms'], 'roofline', r['frac'], r['kernel_ms'], r['traffic'], d['roofline_bwd']['traffic'])"
