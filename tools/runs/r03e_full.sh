#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r03e; mkdir -p $O
timeout 2400 python -m pytest tests/ -x -q -m gpu > $O/pytest_gpu.txt 2>&1; tail -8 $O/pytest_gpu.txt
python bench.py --steps 20 --warmup 3 --no-cpu-baseline 2>&1 | grep -v amdgpu.ids | tail -1 | tee $O/bench.json
