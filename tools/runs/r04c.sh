#!/bin/bash
# round 4, call 3: the full GPU suite on the one-family build (exit code recorded), PMC passes of the GEMM / attention kernels on the
# final sources, default bench line (incl. cpu_baseline), kernel table of the step, configs[3] / configs[4] bench lines
R=${GRAFT_REPO_ROOT:-.}
cd $R; O=$R/gpurun_out/r04c; mkdir -p $O
export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "pytest exit code $?" | tee $O/pytest_gpu.txt; grep -E "passed|failed|error" $O/pytest_gpu.log | tail -3 | tee -a $O/pytest_gpu.txt; grep -E "^(FAILED|ERROR)|Error|assert" $O/pytest_gpu.log | head -20
timeout 300 python bench.py --steps 20 --warmup 5 2>&1 | grep "^{" > $O/bench_default.json; python -c "
import json; d=json.load(open('$O/bench_default.json')); print('bench', d['value'], d['ms_per_step'], d['vit_forward_train_mode_ms'], d['vit_forward_ms'], d['roofline']['frac'], d['roofline']['kernel_ms'], d['roofline']['traffic_source'][:60]); print(d.get('cpu_baseline'))"
bash tools/pmc_gemm256.sh r04c > $O/pmc_gemm256.log 2>&1; tail -3 $O/pmc_gemm256.log
bash tools/pmc.sh r04c/pmc_attn "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT" tools/attn_probe.py 8 > $O/pmc_sq_attention.txt 2>&1; tail -5 $O/pmc_sq_attention.txt
bash tools/profile.sh r04c/step tools/step_only.py 10 > $O/kernel_table_step.txt 2>&1; tail -30 $O/kernel_table_step.txt
timeout 300 python bench.py --no-cpu-baseline --frames 8 --res 448 --steps 10 2>&1 | grep "^{" > $O/bench_cfg3.json; python -c "
import json; d=json.load(open('$O/bench_cfg3.json')); print('cfg3', d['value'], d['ms_per_step'])"
timeout 300 python bench.py --no-cpu-baseline --frames 32 --steps 10 2>&1 | grep "^{" > $O/bench_cfg4.json; python -c "
import json; d=json.load(open('$O/bench_cfg4.json')); print('cfg4', d['value'], d['ms_per_step'])"
timeout 200 python tools/bench_kernels.py all 2>&1 | grep -v amdgpu.ids > $O/bench_kernels.txt; tail -25 $O/bench_kernels.txt
