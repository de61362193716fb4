#!/bin/bash
# Collect the rocprofv3 evidence for bench.py's dominant kernel (run on the GPU box through gpurun):
#   pass 0: --kernel-trace --stats            -> per-kernel durations
#   pass 1-4: --pmc only (no other tracing)   -> SQ instruction mix / waits / LDS, TCC fetch + write bytes
# Summaries are written as text under gpurun_out/<tag>/ ; copy the ones to keep into profiles/.
set -u
TAG=${1:-prof}
shift || true
BENCH="python bench.py --steps 5 --warmup 1 --cpu-budget 0 $*"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/$TAG
mkdir -p $OUT
echo "# command: $BENCH" > $OUT/summary.txt
rocprofv3 --kernel-trace --stats -d $OUT/trace -o t -- $BENCH > $OUT/trace.log 2>&1
grep '^{' $OUT/trace.log | python tools/benchline.py "bench-under-rocprof:" >> $OUT/summary.txt
python tools/rocpd_summary.py $OUT/trace/t_results.db fixpoint >> $OUT/summary.txt
i=0
for CTRS in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD" \
            "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_SCA" \
            "FETCH_SIZE TCC_HIT_sum" "WRITE_SIZE TCC_MISS_sum" "GRBM_GUI_ACTIVE GRBM_COUNT"; do
  i=$((i+1))
  rocprofv3 --pmc $CTRS -d $OUT/pmc$i -o p -- $BENCH > $OUT/pmc$i.log 2>&1
  echo "" >> $OUT/summary.txt
  echo "# --pmc $CTRS" >> $OUT/summary.txt
  python tools/rocpd_summary.py $OUT/pmc$i/p_results.db fixpoint >> $OUT/summary.txt
done
rm -rf $OUT/trace $OUT/pmc[0-9]   # keep the text, drop the databases
cat $OUT/summary.txt
