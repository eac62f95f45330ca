#!/bin/bash
# round 5, call 1: the four-wave 128x128-wave-tile forward loop (gemm256w.hip): parity, isolated timing with ablations, in-step A/B
R=${GRAFT_REPO_ROOT:-.}
cd $R; O=$R/gpurun_out/r05a; mkdir -p $O
export TMPDIR=/tmp
timeout 400 python -m pytest tests/test_gemm_gpu.py -x -q > $O/pytest_gemm.log 2>&1; echo "pytest gemm (W=1) exit $?"; tail -3 $O/pytest_gemm.log
XPRETRAIN_GEMM256W=2 timeout 300 python -m pytest tests/test_gemm_gpu.py -x -q -k "gemm256" > $O/pytest_gemm_bar2.log 2>&1; echo "pytest gemm (W=2) exit $?"; tail -3 $O/pytest_gemm_bar2.log
for r in 1 2; do
for v in 0 1 2 11 12; do
  echo "== round $r GEMM256W=$v" >> $O/bench_gemmfwd.txt
  XPRETRAIN_GEMM256W=$v timeout 120 python tools/bench_kernels.py gemmfwd 2>&1 | grep "gemm fwd" >> $O/bench_gemmfwd.txt
done
echo "== round $r GEMM256W=1 COLGROUPS=1" >> $O/bench_gemmfwd.txt
XPRETRAIN_GEMM256_COLGROUPS=1 timeout 120 python tools/bench_kernels.py gemmfwd 2>&1 | grep "gemm fwd" >> $O/bench_gemmfwd.txt
echo "== round $r GEMM256W=0 COLGROUPS=1" >> $O/bench_gemmfwd.txt
XPRETRAIN_GEMM256W=0 XPRETRAIN_GEMM256_COLGROUPS=1 timeout 120 python tools/bench_kernels.py gemmfwd 2>&1 | grep "gemm fwd" >> $O/bench_gemmfwd.txt
done
cat $O/bench_gemmfwd.txt
for v in 1 2 0; do echo "== trace GEMM256W=$v"; XPRETRAIN_GEMM256W=$v XPRETRAIN_GEMM256=2 timeout 120 python tools/gemm_trace.py 2>&1 | grep "^=="; done | tee $O/trace.txt
timeout 900 python tools/instep_ab.py --rounds 2 --steps 20 --out $O/ab_w.txt w1:XPRETRAIN_GEMM256W=1 w0:XPRETRAIN_GEMM256W=0 w2:XPRETRAIN_GEMM256W=2 2>&1 | tail -5
