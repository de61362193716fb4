#!/bin/bash
# rocprofv3 evidence for one bench.py configuration (run on the GPU box through gpurun):
#   bash tools/profile_round.sh <tag> <name> <kernel-filter> <bench.py args...>
#   pass 0: --kernel-trace --stats           -> per-kernel durations (per grid size)
#   pass 1-5: --pmc only (no tracing domain) -> SQ instruction mix / waits / LDS, TCC fetch + write bytes, GRBM
# The text summary goes to gpurun_out/<tag>/<name>_summary.txt; copy what is to be kept into profiles/.
set -u
TAG=$1; NAME=$2; FILT=$3; shift 3
BENCH="python bench.py --cpu-budget 0 $*"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/$TAG
mkdir -p $OUT
SUM=$OUT/${NAME}_summary.txt
echo "# command: rocprofv3 ... -- $BENCH" > $SUM
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/trace_$NAME -o t -- $BENCH > $OUT/${NAME}_trace.log 2>&1
grep '^{' $OUT/${NAME}_trace.log | python tools/legs.py >> $SUM 2>/dev/null
python tools/rocpd_summary.py $OUT/trace_$NAME/t_results.db "$FILT" >> $SUM
i=0
for CTRS in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD" \
            "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_SCA" \
            "FETCH_SIZE TCC_HIT_sum" "WRITE_SIZE TCC_MISS_sum" "GRBM_GUI_ACTIVE GRBM_COUNT"; do
  i=$((i+1))
  timeout 600 rocprofv3 --pmc $CTRS -d $OUT/pmc_${NAME}_$i -o p -- $BENCH > $OUT/${NAME}_pmc$i.log 2>&1
  echo "" >> $SUM
  echo "# --pmc $CTRS" >> $SUM
  python tools/rocpd_summary.py $OUT/pmc_${NAME}_$i/p_results.db "$FILT" | sed -n '/# PMC counters/,$p' >> $SUM
done
rm -rf $OUT/trace_$NAME $OUT/pmc_${NAME}_[0-9]
cat $SUM
