#!/bin/bash
R=${GRAFT_REPO_ROOT:-.}
cd $R; O=$R/gpurun_out/r05i; mkdir -p $O
export TMPDIR=/tmp
rm -f gpurun_out/parity_log.txt
timeout 900 python -m pytest tests/test_fullsize_parity_gpu.py -x -q -k "bench_batch_8" > $O/pytest.log 2>&1; echo "pytest exit $?"; tail -3 $O/pytest.log
grep "gradient, teacher" gpurun_out/parity_log.txt | grep text_model > $O/text_grads.txt; wc -l $O/text_grads.txt
