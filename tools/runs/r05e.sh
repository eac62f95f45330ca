#!/bin/bash
# persistent tile loop + phase stagger in the 8-wave kernel: parity, sustained timing, timeline
R=${GRAFT_REPO_ROOT:-.}
cd $R; O=$R/gpurun_out/r05e; mkdir -p $O
export TMPDIR=/tmp
export XPRETRAIN_GEMM256W=0
timeout 400 python -m pytest tests/test_gemm_gpu.py -x -q > $O/pytest_gemm.log 2>&1; echo "pytest gemm (non-persistent) exit $?"; tail -2 $O/pytest_gemm.log
XPRETRAIN_GEMM256_PERSIST=1 XPRETRAIN_GEMM256_PHASE_US=10 XPRETRAIN_CU_BUDGET=16 timeout 400 python -m pytest tests/test_gemm_gpu.py -x -q > $O/pytest_gemm_p.log 2>&1; echo "pytest gemm (persistent, 16 CUs) exit $?"; tail -2 $O/pytest_gemm_p.log
XPRETRAIN_GEMM256_PERSIST=1 XPRETRAIN_GEMM256_PHASE_US=10 timeout 400 python -m pytest tests/test_gemm_gpu.py -x -q -k "token_count or gemm256" > $O/pytest_gemm_p2.log 2>&1; echo "pytest gemm (persistent) exit $?"; tail -2 $O/pytest_gemm_p2.log
for v in "0 0" "1 0" "1 6" "1 9" "1 12" "1 15"; do
  set -- $v
  echo "== PERSIST=$1 PHASE_US=$2" | tee -a $O/phase.txt
  XPRETRAIN_GEMM256_PERSIST=$1 XPRETRAIN_GEMM256_PHASE_US=$2 timeout 120 python tools/bench_kernels.py gemmfwd 2>&1 | grep "gemm fwd" | tee -a $O/phase.txt
done
XPRETRAIN_GEMM256_PERSIST=1 XPRETRAIN_GEMM256_PHASE_US=10 timeout 200 python tools/gemm_timeline.py fc1 qkv 2>&1 | grep -v amdgpu.ids | cut -c1-1500 | tee $O/timeline_p10.txt
