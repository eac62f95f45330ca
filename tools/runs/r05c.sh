#!/bin/bash
R=${GRAFT_REPO_ROOT:-.}
cd $R; O=$R/gpurun_out/r05c; mkdir -p $O
export TMPDIR=/tmp
XPRETRAIN_GEMM256W=0 timeout 200 python tools/gemm_timeline.py fc1 out qkv fc2 2>&1 | grep -v amdgpu.ids | tee $O/timeline.txt
