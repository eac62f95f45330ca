#!/bin/bash
R=${GRAFT_REPO_ROOT:-.}
cd $R; O=$R/gpurun_out/r05f; mkdir -p $O
export TMPDIR=/tmp
export XPRETRAIN_GEMM256W=0
for v in "0 0 2" "1 12 2" "1 6 4" "1 4 4" "1 3 8" "1 5 4" "0 0 2"; do
  set -- $v
  echo "== PERSIST=$1 PHASE_US=$2 GROUPS=$3" | tee -a $O/phase.txt
  XPRETRAIN_GEMM256_PERSIST=$1 XPRETRAIN_GEMM256_PHASE_US=$2 XPRETRAIN_GEMM256_PHASE_GROUPS=$3 timeout 120 python tools/bench_kernels.py gemmfwd 2>&1 | grep "gemm fwd" | tee -a $O/phase.txt
done
timeout 900 python tools/instep_ab.py --rounds 2 --steps 20 --out $O/ab_persist.txt base p2:XPRETRAIN_GEMM256_PERSIST=1,XPRETRAIN_GEMM256_PHASE_US=12 p4:XPRETRAIN_GEMM256_PERSIST=1,XPRETRAIN_GEMM256_PHASE_US=5,XPRETRAIN_GEMM256_PHASE_GROUPS=4 base1:XPRETRAIN_FWD_SPLIT=0 p4one:XPRETRAIN_GEMM256_PERSIST=1,XPRETRAIN_GEMM256_PHASE_US=5,XPRETRAIN_GEMM256_PHASE_GROUPS=4,XPRETRAIN_FWD_SPLIT=0 2>&1 | tail -7
