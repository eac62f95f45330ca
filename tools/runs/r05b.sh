#!/bin/bash
# round 5, call 2: where the energy of the 8-wave forward kernel goes (ablations: time == energy at the power cap) + power probe
R=${GRAFT_REPO_ROOT:-.}
cd $R; O=$R/gpurun_out/r05b; mkdir -p $O
export TMPDIR=/tmp
export XPRETRAIN_GEMM256W=0
for r in 1 2; do
for v in 0 1 2 3 4; do
  echo "== round $r ABL=$v" >> $O/abl.txt
  XPRETRAIN_GEMM256_ABL=$v timeout 120 python tools/bench_kernels.py gemmfwd 2>&1 | grep "gemm fwd" >> $O/abl.txt
done
done
cat $O/abl.txt
which rocm-smi amd-smi; rocm-smi --showpower --showclocks --json 2>&1 | head -c 600; echo
timeout 120 python tools/power_probe.py 3 2>&1 | tail -4 | tee $O/power.txt
XPRETRAIN_GEMM256_ABL=3 timeout 120 python tools/power_probe.py 3 2>&1 | tail -4 | tee $O/power_abl3.txt
