#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r03d; mkdir -p $O
timeout 900 python -m pytest tests/test_gemm_gpu.py -x -q > $O/pytest_gemm.txt 2>&1; tail -5 $O/pytest_gemm.txt
for i in 1 2; do
  XPRETRAIN_GEMM256_PERSIST=0 python tools/bench_kernels.py gemm 2>&1 | grep "fwd" > $O/bench_nop_$i.txt
  python tools/bench_kernels.py gemm 2>&1 | grep "fwd" > $O/bench_p_$i.txt
done
for i in 1 2; do echo "--- round $i: one tile per workgroup | persistent"; paste -d'|' $O/bench_nop_$i.txt $O/bench_p_$i.txt | sed 's/gemm fwd //g; s/M=18848 //g' | cut -c1-220; done
XPRETRAIN_GEMM256_PERSIST=0 python tools/fwd_only.py 10 12 224 both 2>&1 | grep -v amdgpu.ids | tee $O/fwd_nop.txt
python tools/fwd_only.py 10 12 224 both 2>&1 | grep -v amdgpu.ids | tee $O/fwd_p.txt
