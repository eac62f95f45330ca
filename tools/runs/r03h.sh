#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r03h; mkdir -p $O
timeout 2400 python -m pytest tests/test_distributed_gpu.py tests/test_model_gpu.py tests/test_fullsize_parity_gpu.py tests/test_dp_sim_gpu.py -x -q > $O/pytest.txt 2>&1; tail -6 $O/pytest.txt
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; echo "smoke rc=$?"; grep smoke $O/smoke.txt
python tools/contention_probe.py 10 2>&1 | grep -v amdgpu.ids | tee $O/contention.txt
