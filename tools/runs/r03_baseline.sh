#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
bash tools/pmc.sh r03a_grbm "GRBM_GUI_ACTIVE GRBM_COUNT" tools/gemm256_probe.py 6 > gpurun_out/r03a_grbm.txt 2>&1
grep -A3 "gemm256\|cast_kernelIDF" gpurun_out/r03a_grbm.txt | head -40
python tools/gemm_trace256.py 2>&1 | tail -6
