#!/bin/bash
R=${GRAFT_REPO_ROOT:-.}
cd $R; O=$R/gpurun_out/r05h; mkdir -p $O
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_prefetch_gpu.py -x -q > $O/pytest.log 2>&1; echo "pytest exit $?"; tail -3 $O/pytest.log
timeout 1200 python tools/instep_ab.py --rounds 2 --steps 20 --out $O/ab_prefetch.txt base pf1/--prefetch=1 pf2/--prefetch=2 fc:XPRETRAIN_BENCH_FORCE_COLLECTIVES=1 fc_pf1:XPRETRAIN_BENCH_FORCE_COLLECTIVES=1/--prefetch=1 fc_pf2:XPRETRAIN_BENCH_FORCE_COLLECTIVES=1/--prefetch=2 fc_pf3u8:XPRETRAIN_BENCH_FORCE_COLLECTIVES=1/--prefetch=3/--prefetch-dtype=uint8 2>&1 | tail -9
