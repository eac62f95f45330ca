#!/bin/bash
# round 4, call 10: re-run of the model / embedding tests, in-step A/B of two scheduling knobs (priority of the weight-gradient stream,
# a cap on the split-K factor of the small dW GEMMs)
R=${GRAFT_REPO_ROOT:-.}
cd $R; O=$R/gpurun_out/r04j; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_model_gpu.py tests/test_embed_loss_gpu.py -x -q > $O/pytest_part.log 2>&1; echo "pytest exit code $?" | tee $O/pytest_part.txt; grep -E "passed|failed|error" $O/pytest_part.log | tail -3 | tee -a $O/pytest_part.txt; grep -E "^(FAILED|ERROR)|Error|assert" $O/pytest_part.log | head -20
timeout 1200 python tools/instep_ab.py --rounds 2 --steps 20 --out $O/instep_ab_knobs.txt default low:XPRETRAIN_WGRAD_PRIO=low high:XPRETRAIN_WGRAD_PRIO=high split14:XPRETRAIN_DW_MAX_SPLIT=14 split9:XPRETRAIN_DW_MAX_SPLIT=9 2>&1 | tail -7
