#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r03c; mkdir -p $O
XPRETRAIN_GEMM256_MI32=1 timeout 900 python -m pytest tests/test_gemm_gpu.py -x -q > $O/pytest_gemm_mi32.txt 2>&1; tail -5 $O/pytest_gemm_mi32.txt
for i in 1 2; do
  XPRETRAIN_GEMM256_MI32=1 python tools/bench_kernels.py gemm 2>&1 | grep "fwd" > $O/bench_mi32_$i.txt
  XPRETRAIN_GEMM256_MT1=4 python tools/bench_kernels.py gemm 2>&1 | grep "fwd" > $O/bench_mt4_$i.txt
  python tools/bench_kernels.py gemm 2>&1 | grep "fwd" > $O/bench_auto_$i.txt
done
for i in 1 2; do echo "--- round $i: mi32 | 16x16 256 | 16x16 224"; paste -d'|' $O/bench_mi32_$i.txt $O/bench_mt4_$i.txt $O/bench_auto_$i.txt | sed 's/gemm fwd //g; s/M=18848 //g' | cut -c1-220; done
XPRETRAIN_GEMM256_MI32=1 python tools/gemm_trace256.py > $O/trace_mi32.txt 2>&1; grep -v amdgpu.ids $O/trace_mi32.txt
XPRETRAIN_GEMM256_MI32=1 python tools/fwd_only.py 10 12 224 both 2>&1 | grep -v amdgpu.ids | tee $O/fwd_mi32.txt
python tools/fwd_only.py 10 12 224 both 2>&1 | grep -v amdgpu.ids | tee $O/fwd_auto.txt
