#!/bin/bash
R=${GRAFT_REPO_ROOT:-.}
cd $R
( cd _ab_lines && timeout 200 python -m pytest tests/test_gemm_gpu.py -x -q 2>&1 | tail -3 )
python tools/instep_ab.py --rounds 2 --out gpurun_out/r03_lines_ab.txt default direct:XPRETRAIN_GEMM256_STAGED=0 lines@_ab_lines:XPRETRAIN_GEMM256_STAGED=0 direct224:XPRETRAIN_GEMM256_MT1=3 lines224@_ab_lines:XPRETRAIN_GEMM256_MT1=3
