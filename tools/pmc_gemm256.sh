#!/bin/bash
# Usage (GPU box): bash tools/pmc_gemm256.sh <tag>     -> gpurun_out/<tag>_pmc_{sq,fetch,write}_gemm256.csv (small summaries)
# Three SEPARATE rocprofv3 passes (kernel-trace + pmc only, as the microarch guide prescribes): SQ activity, FETCH_SIZE, WRITE_SIZE.
TAG=${1:-r02}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
run() {   # name, counters...
  local NAME=$1; shift
  rm -rf /tmp/pmc_$NAME
  ( cd $REPO && timeout 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d /tmp/pmc_$NAME -o $NAME -- python tools/gemm256_probe.py 4 ) > $OUT/${TAG}_pmc_${NAME}.log 2>&1
  local F=$(find /tmp/pmc_$NAME -name "*counter_collection.csv" | head -1)
  python3 $REPO/tools/pmc_summarize.py $F $OUT/${TAG}_pmc_${NAME}_gemm256.csv gemm256 cast_kernel > $OUT/${TAG}_pmc_${NAME}.txt
  tail -n 40 $OUT/${TAG}_pmc_${NAME}.txt
}
run sq SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS
run fetch FETCH_SIZE
run write WRITE_SIZE
run tcc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum
python3 $REPO/tools/pmc_gemm256_json.py $TAG $OUT $OUT/${TAG}_pmc_gemm256.json
