#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r03s; mkdir -p $O
bash tools/pmc_gemm256.sh r03k > gpurun_out/r03k_pmc_run.txt 2>&1; tail -1 gpurun_out/r03k_pmc_run.txt | cut -c1-120
timeout 2400 python -m pytest tests/ -x -q -m gpu > $O/pytest_gpu.txt 2>&1; tail -3 $O/pytest_gpu.txt
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; echo "smoke rc=$?"
