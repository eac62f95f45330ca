#!/bin/bash
# (historical run script: XPRETRAIN_WGRAD_PRIORITY and XPRETRAIN_FWD_CHAINS were experiment switches of that moment -- the weight-gradient stream now has
# the default priority and there are exactly two forward chains; the tree no longer reads them)
R=${GRAFT_REPO_ROOT:-.}
cd $R; O=$R/gpurun_out/r04v; mkdir -p $O
FC=XPRETRAIN_BENCH_FORCE_COLLECTIVES=1
timeout 1700 python tools/instep_ab.py --rounds 1 --steps 20 --out $O/ab_forced_matrix2.txt \
  O_plain_default \
  N_text:$FC,XPRETRAIN_WGRAD_STREAM=0,XPRETRAIN_FWD_SPLIT=0 \
  I_text_split:$FC,XPRETRAIN_WGRAD_STREAM=0,XPRETRAIN_FWD_SPLIT_STREAM=own \
  J_all_wgrad_normal_prio:$FC,XPRETRAIN_WGRAD_PRIORITY=normal,XPRETRAIN_FWD_SPLIT_STREAM=own \
  K_wgrad_normal_prio_only:$FC,XPRETRAIN_WGRAD_PRIORITY=normal,XPRETRAIN_OVERLAP_TEXT=0,XPRETRAIN_FWD_SPLIT=0 \
  P_all_side_normal_prio:$FC,XPRETRAIN_WGRAD_PRIORITY=normal,XPRETRAIN_FWD_SPLIT_STREAM=side \
  L_all_q2:$FC,GPU_MAX_HW_QUEUES=2 \
  M_all_q3:$FC,GPU_MAX_HW_QUEUES=3 \
  Q_all_q1:$FC,GPU_MAX_HW_QUEUES=1 2>&1 | tail -11
