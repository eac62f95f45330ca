#!/bin/bash
# One runner for every GPU-box call (replaces the per-call scripts of rounds 3-5 under tools/runs/; what each round ran is listed
# in profiles/README.md next to the files it produced).  Run through gpurun from the repository root, tasks chained with ';':
#
#   gpurun -- 'bash tools/gpu_run.sh r05z tests; bash tools/gpu_run.sh r05z bench --steps 20; bash tools/gpu_run.sh r05z profile'
#
#   tests   [pytest args]          python -m pytest tests -m gpu -q [args]            -> gpurun_out/<tag>/pytest.log
#   bench   [bench.py args]        the bench line                                     -> gpurun_out/<tag>/bench.json
#   ab      <rounds> <variant...>  tools/instep_ab.py, interleaved whole-step A/B     -> gpurun_out/<tag>/ab.txt
#   kernels [gemm|gemmfwd|ln|attn] tools/bench_kernels.py (sustained clocks)          -> gpurun_out/<tag>/kernels.txt
#   profile [bench.py args]        rocprofv3 --kernel-trace --stats of bench.py: per-kernel stats + one row per (kernel, grid)
#                                                                                     -> gpurun_out/<tag>/{kernel_stats.csv,kernel_by_grid.txt,bench_under_rocprof.json}
#   pmc                            PMC passes of the 256-wide GEMM family (tools/pmc_gemm256.sh)  -> gpurun_out/<tag>_pmc_gemm256.json
#   pmcfwd                         MFMA-busy fraction of the video tower's training-mode forward  -> gpurun_out/<tag>_pmc_vit_forward.json
#   pmcattn                        SQ / HBM counters of the attention kernels                    -> gpurun_out/<tag>/pmc_sq_attention.txt
#   timeline [shapes]              per-tile timeline of single GEMM launches (tools/gemm_timeline.py) -> gpurun_out/<tag>/timeline.txt
#   steptl  [args]                one steady-state step as a per-kernel timeline (tools/step_timeline.py)  -> gpurun_out/<tag>/step_timeline.txt
#   vendor  [args]                 in-step calibration against the vendor library (tools/vendor_instep.py) -> gpurun_out/<tag>/vendor_instep.txt
set -u
TAG=${1:?tag}; TASK=${2:?task}; shift 2
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R; O=$R/gpurun_out/$TAG; mkdir -p $O
export TMPDIR=/tmp
case $TASK in
  tests)
    timeout 1500 python -m pytest tests -m gpu -q "$@" > $O/pytest.log 2>&1; echo "pytest exit $?"; tail -4 $O/pytest.log ;;
  bench)
    timeout 900 python bench.py "$@" 2> $O/bench.err | grep "^{" > $O/bench.json
    python3 -c "
import json; d=json.load(open('$O/bench.json')); r=d['roofline']; b=d['roofline_bwd']
print('bench', d['value'], 'pairs/s', d['ms_per_step'], 'ms/step; vit fwd', d['vit_forward_train_mode_ms'], d['vit_forward_ms'], '; roofline', r['frac'], r['kernel_ms'], '; bwd', b['frac'], b['kernel_ms'], '; cpu', d.get('cpu_baseline', {}).get('value'))" ;;
  ab)
    ROUNDS=$1; shift
    timeout 3000 python tools/instep_ab.py --rounds $ROUNDS --steps 20 --out $O/ab.txt "$@" 2>&1 | tail -$(( $# + 2 )) ;;
  kernels)
    timeout 600 python tools/bench_kernels.py "${1:-all}" 2>&1 | grep -v amdgpu.ids | tee $O/kernels.txt ;;
  profile)
    rm -rf /tmp/prof_$TAG
    timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$TAG -o p -- python bench.py --steps 10 --warmup 3 --profile-run "$@" > $O/bench_under_rocprof.log 2>&1
    grep "^{" $O/bench_under_rocprof.log > $O/bench_under_rocprof.json
    cp $(find /tmp/prof_$TAG -name "*kernel_stats.csv" | head -1) $O/kernel_stats.csv
    python3 tools/kernel_by_grid.py $(find /tmp/prof_$TAG -name "*kernel_trace.csv" | head -1) > $O/kernel_by_grid.txt
    grep -E "^kernel|gemm256_kernel<false, false> +888" $O/kernel_by_grid.txt | cut -c1-150
    python3 -c "
import json; d=json.load(open('$O/bench_under_rocprof.json')); print('the same run: roofline.kernel_ms (HIP events, median of the in-step fc1 launches) =', d['roofline']['kernel_ms'], 'ms;', d['ms_per_step'], 'ms/step under the profiler')" ;;
  pmc)
    bash tools/pmc_gemm256.sh $TAG > $O/pmc.log 2>&1; tail -1 $O/pmc.log | cut -c1-300 ;;
  pmcfwd)
    # aggregate MFMA-busy fraction of the video tower's training-mode forward (one PMC pass, tools/pmc_vit_forward.py)
    rm -rf /tmp/pmcfwd_$TAG
    ( cd $R && timeout 600 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d /tmp/pmcfwd_$TAG -o f -- python tools/fwd_only.py 5 12 224 train ) > $O/pmcfwd.log 2>&1
    python3 tools/pmc_vit_forward.py $(find /tmp/pmcfwd_$TAG -name "*counter_collection.csv" | head -1) $R/gpurun_out/${TAG}_pmc_vit_forward.json $TAG | head -12 ;;
  pmcattn)
    # SQ counters of the attention kernels (forward persistent kernel + fused backward), two passes -> gpurun_out/<tag>/pmc_sq_attention.txt
    { bash tools/pmc.sh ${TAG}_pa1 "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT" tools/attn_bwd_probe.py 5 fused colsum;
      bash tools/pmc.sh ${TAG}_pa2 "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR GRBM_GUI_ACTIVE" tools/attn_bwd_probe.py 5 fused colsum;
      bash tools/pmc.sh ${TAG}_pa3 "FETCH_SIZE" tools/attn_bwd_probe.py 5 fused colsum; bash tools/pmc.sh ${TAG}_pa4 "WRITE_SIZE" tools/attn_bwd_probe.py 5 fused colsum; } 2>&1 | grep -v "^W20\|amdgpu.ids" | grep -A10 "attn_bwd5_kernel\|attn_fwd3_kernel" > $O/pmc_sq_attention.txt
    head -30 $O/pmc_sq_attention.txt ;;
  timeline)
    timeout 300 python tools/gemm_timeline.py "$@" 2>&1 | grep -v amdgpu.ids | cut -c1-1500 | tee $O/timeline.txt ;;
  steptl)
    # one steady-state step as a per-kernel timeline (tools/step_timeline.py) -> gpurun_out/<tag>/step_timeline.txt (+ the raw trace, gzipped)
    rm -rf /tmp/tl_$TAG
    timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/tl_$TAG -o t -- python tools/step_only.py 8 > $O/steptl.log 2>&1
    TR=$(find /tmp/tl_$TAG -name "*kernel_trace.csv" | head -1)
    python3 tools/step_timeline.py $TR "$@" > $O/step_timeline.txt; gzip -c $TR > $O/kernel_trace.csv.gz
    head -1 $O/step_timeline.txt; tail -1 $O/step_timeline.txt; tail -1 $O/steptl.log ;;
  vendor)
    timeout 900 python tools/vendor_instep.py --out $O/vendor_instep.txt "$@" 2>&1 | grep -v amdgpu.ids | tail -48 ;;
  *) echo "unknown task $TASK"; exit 2 ;;
esac
