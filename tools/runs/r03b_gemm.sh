#!/bin/bash
# round 3, step b: direct epilogue + shape-aware tile height.  Correctness first, then A/B of the tile height, then slot trace.
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r03b; mkdir -p $O
timeout 900 python -m pytest tests/test_gemm_gpu.py -x -q > $O/pytest_gemm.txt 2>&1; tail -5 $O/pytest_gemm.txt
for i in 1 2; do
  XPRETRAIN_GEMM256_MT1=4 python tools/bench_kernels.py gemm 2>&1 | grep -v colsum > $O/bench_mt4_$i.txt
  python tools/bench_kernels.py gemm 2>&1 | grep -v colsum > $O/bench_auto_$i.txt
done
paste -d'|' $O/bench_mt4_1.txt $O/bench_auto_1.txt | cut -c1-200
echo ---- second round; paste -d'|' $O/bench_mt4_2.txt $O/bench_auto_2.txt | cut -c1-200
XPRETRAIN_GEMM256_MT1=4 python tools/gemm_trace256.py > $O/trace_mt4.txt 2>&1; grep -v amdgpu.ids $O/trace_mt4.txt
python tools/gemm_trace256.py > $O/trace_auto.txt 2>&1; grep -v amdgpu.ids $O/trace_auto.txt
python tools/fwd_only.py 10 12 224 both 2>&1 | grep -v amdgpu.ids | tee $O/fwd.txt
