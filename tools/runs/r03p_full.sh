#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r03p; mkdir -p $O
timeout 2400 python -m pytest tests/ -x -q -m gpu > $O/pytest_gpu.txt 2>&1; tail -5 $O/pytest_gpu.txt
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; echo "smoke rc=$?"
python bench.py 2>&1 | grep "^{" | tee $O/bench_default.json | cut -c1-250
