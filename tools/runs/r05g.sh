#!/bin/bash
# round 5: new GPU tests (prefetch loader, lazy logits), the step in the reference loop's stream environment (bench.py --prefetch)
R=${GRAFT_REPO_ROOT:-.}
cd $R; O=$R/gpurun_out/r05g; mkdir -p $O
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_prefetch_gpu.py tests/test_model_gpu.py tests/test_gemm_gpu.py tests/test_distributed_gpu.py -x -q > $O/pytest.log 2>&1; echo "pytest exit $?"; tail -3 $O/pytest.log
timeout 1200 python tools/instep_ab.py --rounds 2 --steps 20 --out $O/ab_prefetch.txt base pf1/--prefetch=1 pf1u8/--prefetch=1/--prefetch-dtype=uint8 pf2/--prefetch=2 fc:XPRETRAIN_BENCH_FORCE_COLLECTIVES=1 fc_pf1:XPRETRAIN_BENCH_FORCE_COLLECTIVES=1/--prefetch=1 fc_pf2:XPRETRAIN_BENCH_FORCE_COLLECTIVES=1/--prefetch=2 2>&1 | tail -9
