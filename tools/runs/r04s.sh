#!/bin/bash
# round 4: the video tower's forward as two half-batch chains (functional.ForwardSplit): bit-identity test, interleaved step A/B,
# PMC passes of the GEMM family on the new sources (family rule: from 96 workgroups), default bench
R=${GRAFT_REPO_ROOT:-.}
cd $R; O=$R/gpurun_out/r04s; mkdir -p $O
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_model_gpu.py tests/test_gemm_gpu.py -x -q > $O/pytest_part.log 2>&1; echo "pytest exit code $?" | tee $O/pytest_part.txt; grep -E "passed|failed|error" $O/pytest_part.log | tail -3 | tee -a $O/pytest_part.txt; grep -E "^(FAILED|ERROR)|Error|assert" $O/pytest_part.log | head -20
timeout 900 python tools/instep_ab.py --rounds 3 --steps 20 --out $O/ab_fwd_split.txt split:XPRETRAIN_FWD_SPLIT=1 one:XPRETRAIN_FWD_SPLIT=0 2>&1 | tail -4
bash tools/pmc_gemm256.sh r04s > $O/pmc_gemm256.log 2>&1; tail -1 $O/pmc_gemm256.log | cut -c1-200
cp $R/gpurun_out/r04s_pmc_gemm256.json $R/profiles/r04s_pmc_gemm256.json
timeout 300 python bench.py --steps 20 --warmup 5 2>&1 | grep "^{" > $O/bench_default.json; python -c "
import json; d=json.load(open('$O/bench_default.json')); r=d['roofline']; print('bench', d['value'], d['ms_per_step'], d['vit_forward_train_mode_ms'], d['vit_forward_ms'], 'roofline', r['frac'], r['kernel_ms'], r['kernel_ms_half_batch_launch_beside_the_other_chain'], r['traffic'], d['roofline_bwd']['traffic'], d['cpu_baseline']['value'])"
