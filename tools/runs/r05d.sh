#!/bin/bash
# phase stagger: isolated timeline + sustained timing at several delays
R=${GRAFT_REPO_ROOT:-.}
cd $R; O=$R/gpurun_out/r05d; mkdir -p $O
export TMPDIR=/tmp
export XPRETRAIN_GEMM256W=0
for d in 0 6 9 12 15; do
  echo "== PHASE_US=$d" | tee -a $O/phase.txt
  XPRETRAIN_GEMM256_PHASE_US=$d timeout 120 python tools/bench_kernels.py gemmfwd 2>&1 | grep "gemm fwd" | tee -a $O/phase.txt
done
XPRETRAIN_GEMM256_PHASE_US=12 timeout 200 python tools/gemm_timeline.py fc1 qkv 2>&1 | grep -v amdgpu.ids | cut -c1-1200 | tee $O/timeline_phase12.txt
