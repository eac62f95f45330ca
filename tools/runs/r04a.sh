#!/bin/bash
# round 4, call 1: correctness of the new scheduling switches, vendor-library in-step calibration, in-step A/B of the dW changes,
# RCCL-footprint contention probe, per-kernel tables of the step with and without the weight-gradient stream
R=${GRAFT_REPO_ROOT:-.}
cd $R; O=$R/gpurun_out/r04a; mkdir -p $O
export TMPDIR=/tmp
( timeout 600 python -m pytest tests/test_gemm_gpu.py tests/test_model_gpu.py -x -q 2>&1 | tail -5 ) > $O/pytest_default.txt
( XPRETRAIN_WGRAD_STREAM=1 XPRETRAIN_DW_CHUNK_MAJOR=0 timeout 600 python -m pytest tests/test_model_gpu.py tests/test_fullsize_gpu.py tests/test_determinism_gpu.py -x -q 2>&1 | tail -5 ) > $O/pytest_wgrad_stream.txt
cat $O/pytest_default.txt $O/pytest_wgrad_stream.txt
timeout 400 python tools/vendor_instep.py --steps 6 --order both --out $O/vendor_instep.txt 2>&1 | grep -v amdgpu.ids | tail -45
timeout 900 python tools/instep_ab.py --rounds 2 --steps 20 --out $O/instep_ab.txt \
   default cm0:XPRETRAIN_DW_CHUNK_MAJOR=0 ws1:XPRETRAIN_WGRAD_STREAM=1 ws1cm0:XPRETRAIN_WGRAD_STREAM=1,XPRETRAIN_DW_CHUNK_MAJOR=0 2>&1 | tail -8
timeout 300 python tools/contention_probe.py 10 2>&1 | grep -v amdgpu.ids | tee $O/contention_probe.txt
bash tools/profile.sh r04a/step_default tools/step_only.py 10 > $O/kernel_table_default.txt 2>&1
XPRETRAIN_WGRAD_STREAM=1 bash tools/profile.sh r04a/step_ws1 tools/step_only.py 10 > $O/kernel_table_ws1.txt 2>&1
tail -32 $O/kernel_table_default.txt
