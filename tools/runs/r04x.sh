#!/bin/bash
# (historical run script: XPRETRAIN_WGRAD_PRIORITY and XPRETRAIN_FWD_CHAINS were experiment switches of that moment -- the weight-gradient stream now has
# the default priority and there are exactly two forward chains; the tree no longer reads them)
R=${GRAFT_REPO_ROOT:-.}
cd $R; O=$R/gpurun_out/r04x; mkdir -p $O
timeout 600 python -m pytest tests/test_distributed_gpu.py -x -q > $O/pytest_part.log 2>&1; echo "pytest exit code $?"; grep -E "passed|failed|error" $O/pytest_part.log | tail -3; grep -E "^(FAILED|ERROR)|Error|assert" $O/pytest_part.log | head -20
for m in plain forced; do timeout 300 python tools/step_phases.py 20 $m 2>&1 | grep -v Warn | grep -E "mode|^  " | tee -a $O/step_phases2.txt; done
