#!/bin/bash
# (historical run script: XPRETRAIN_WGRAD_PRIORITY and XPRETRAIN_FWD_CHAINS were experiment switches of that moment -- the weight-gradient stream now has
# the default priority and there are exactly two forward chains; the tree no longer reads them)
R=${GRAFT_REPO_ROOT:-.}
cd $R; O=$R/gpurun_out/r04u; mkdir -p $O
timeout 300 python -m pytest tests/test_model_gpu.py -x -q -k "two_half" 2>&1 | tail -2
timeout 1200 python tools/instep_ab.py --rounds 3 --steps 20 --out $O/ab_split_stream.txt side:XPRETRAIN_FWD_SPLIT_STREAM=side own:XPRETRAIN_FWD_SPLIT_STREAM=own 2>&1 | tail -4
PROBE_EXTRA_STREAMS=1,2 timeout 600 python tools/contention_probe.py 10 fat 32 2>&1 | grep -v Warn | tee $O/streams_side_shared.txt | tail -5
