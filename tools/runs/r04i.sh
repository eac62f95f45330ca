#!/bin/bash
# round 4, call 9: attention forward with parked outputs -- tests, isolated timing, SQ counters, in-step A/B against the previous build
R=${GRAFT_REPO_ROOT:-.}
cd $R; O=$R/gpurun_out/r04i; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_attention_gpu.py tests/test_model_gpu.py tests/test_embed_loss_gpu.py tests/test_determinism_gpu.py tests/test_fullsize_gpu.py -x -q > $O/pytest_part.log 2>&1; echo "pytest exit code $?" | tee $O/pytest_part.txt; grep -E "passed|failed|error" $O/pytest_part.log | tail -3 | tee -a $O/pytest_part.txt; grep -E "^(FAILED|ERROR)|Error|assert" $O/pytest_part.log | head -20
timeout 200 python tools/bench_kernels.py all 2>&1 | grep -i "attn" | tee $O/attn_isolated.txt
( cd _ab_prev && timeout 200 python tools/bench_kernels.py all 2>&1 | grep -i "attn" | sed 's/^/[previous build] /' ) | tee -a $O/attn_isolated.txt
timeout 900 python tools/instep_ab.py --rounds 3 --steps 20 --out $O/instep_ab_attn.txt default prev@_ab_prev 2>&1 | tail -4
bash tools/pmc.sh r04i/pmc_attn "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT" tools/attn_probe.py 8 > $O/pmc_sq_attention.txt 2>&1; grep -A9 "attn_fwd3" $O/pmc_sq_attention.txt | head -12
