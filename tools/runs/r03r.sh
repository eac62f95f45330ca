#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r03r; mkdir -p $O
timeout 2400 python -m pytest tests/test_fullsize_parity_gpu.py tests/test_model_gpu.py -q -s > $O/pytest2.txt 2>&1; grep -v "maxrel=" $O/pytest2.txt | grep "full_cfg\|passed\|failed\|fp32 mode\|cfg1\|tiny:" | cut -c1-300
