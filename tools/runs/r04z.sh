#!/bin/bash
R=${GRAFT_REPO_ROOT:-.}
cd $R; O=$R/gpurun_out/r04y; mkdir -p $O
timeout 1800 python -m pytest tests/test_fullsize_parity_gpu.py tests/test_determinism_gpu.py -x -q > $O/pytest_gpu2.log 2>&1; echo "pytest exit code $?" | tee $O/pytest_gpu2.txt; grep -E "passed|failed|error" $O/pytest_gpu2.log | tail -3 | tee -a $O/pytest_gpu2.txt; grep -E "^(FAILED|ERROR)|Error|assert" $O/pytest_gpu2.log | head -20
