#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
bash tools/pmc_gemm256.sh r03k > gpurun_out/r03k_pmc_run.txt 2>&1
tail -5 gpurun_out/r03k_pmc_run.txt | cut -c1-300
bash tools/pmc.sh r03k_attn_sq "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VALU" tools/bench_kernels.py attn > gpurun_out/r03k_attn_sq.txt 2>&1
grep -A9 "attn_fwd3\|attn_bwd_dq\|attn_bwd_dkv" gpurun_out/r03k_attn_sq.txt | head -40
