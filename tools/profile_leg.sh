#!/bin/bash
# rocprofv3 evidence for ONE leg replayed by tools/replay_leg.py (run on the GPU box through gpurun):
#   bash tools/profile_leg.sh <tag> <leg> <kernel-filter>
# pass 0: --kernel-trace --stats; passes 1-4: --pmc only (never combined with a trace domain).  Summary: gpurun_out/<tag>/<leg>_summary.txt
set -u
TAG=$1; LEG=$2; FILT=$3
CMD="python tools/replay_leg.py run $LEG"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/$TAG; mkdir -p $OUT
SUM=$OUT/${LEG}_summary.txt
echo "# command: rocprofv3 ... -- $CMD" > $SUM
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/trace_$LEG -o t -- $CMD > $OUT/${LEG}_trace.log 2>&1
grep '^{' $OUT/${LEG}_trace.log >> $SUM
python tools/rocpd_summary.py $OUT/trace_$LEG/t_results.db "$FILT" >> $SUM
i=0
for CTRS in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD" \
            "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_SCA" \
            "FETCH_SIZE TCC_HIT_sum" "WRITE_SIZE TCC_MISS_sum"; do
  i=$((i+1))
  timeout 600 rocprofv3 --pmc $CTRS -d $OUT/pmc_${LEG}_$i -o p -- $CMD > $OUT/${LEG}_pmc$i.log 2>&1
  echo "" >> $SUM; echo "# --pmc $CTRS" >> $SUM
  python tools/rocpd_summary.py $OUT/pmc_${LEG}_$i/p_results.db "$FILT" | sed -n '/# PMC counters/,$p' >> $SUM
done
rm -rf $OUT/trace_$LEG $OUT/pmc_${LEG}_[0-9]
tail -n 60 $SUM
