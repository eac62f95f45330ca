#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r03g; mkdir -p $O
timeout 1500 python -m pytest tests/test_fullsize_parity_gpu.py -x -q -s -k "batch_8" > $O/b8.txt 2>&1; grep -v "maxrel=" $O/b8.txt | tail -12
timeout 900 python -m pytest tests/test_model_gpu.py -x -q -s > $O/model.txt 2>&1; grep "cfg1:\|cfg1 worst\|tiny\b.*loss\|passed\|failed\|worst" $O/model.txt | grep -v maxrel | tail -12
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep smoke | tee $O/smoke.txt
cat gpurun_out/fullsize_parity_full_cfg2_b8.json
