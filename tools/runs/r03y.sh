#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r03y; mkdir -p $O
timeout 600 python -m pytest tests/test_gemm_gpu.py tests/test_layernorm_gpu.py tests/test_model_gpu.py -x -q 2>&1 | tail -4
bash tools/pmc_gemm256.sh r03y > $O/pmc.log 2>&1; tail -3 $O/pmc.log
