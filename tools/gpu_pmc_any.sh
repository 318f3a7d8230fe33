#!/bin/bash
# Runs on the GPU box (via gpurun): separate PMC passes (never combined with other trace domains) of any tools/ driver.
# usage: tools/gpu_pmc_any.sh <tag> <script.py> [args...]      (env PMC_SETS="sq sq2 lds fetch write tcc tcp" to select)
set -o pipefail
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=$1; shift
SCRIPT=$1; shift
ARGS="$@"
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp
SETS=${PMC_SETS:-"sq sq2 lds tcp"}
declare -A C
C[sq]="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE"
C[sq2]="SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU"
C[lds]="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_LEVEL_WAVES"
C[fetch]="FETCH_SIZE"
C[write]="WRITE_SIZE"
C[tcc]="TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum"
C[tcp]="TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum"
for name in $SETS; do
  rocprofv3 --pmc ${C[$name]} --kernel-trace --output-format csv -d $OUT/pmc_$name -- python3 $R/tools/$SCRIPT $ARGS > $OUT/pmc_$name.log 2>&1 || { echo "pmc $name failed"; tail -3 $OUT/pmc_$name.log; }
done
python3 $R/tools/summarize_prof.py $OUT > $OUT/summary.txt 2>&1
grep -v "at::native\|rocclr\|^void  " $OUT/summary.txt | cut -c1-${COLS:-420}
