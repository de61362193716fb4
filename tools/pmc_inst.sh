cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/${1:-r04pmc}; mkdir -p $OUT; LEG=${2:-frontier}; FILT=${3:-neqfix}
CMD="python tools/replay_leg.py run $LEG"
: > $OUT/${LEG}_inst_summary.txt
i=0
for CTRS in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VALU SQ_WAVES" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_BUSY_CYCLES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_BRANCH SQ_LDS_BANK_CONFLICT" "GRBM_GUI_ACTIVE GRBM_COUNT"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $CTRS -d $OUT/pmc_i_$i -o p -- $CMD > $OUT/inst_pmc$i.log 2>&1
  echo "# --pmc $CTRS" >> $OUT/${LEG}_inst_summary.txt
  timeout 60 python tools/rocpd_summary.py $OUT/pmc_i_$i/p_results.db $FILT 2>&1 | sed -n '/# PMC counters/,$p' >> $OUT/${LEG}_inst_summary.txt
done
rm -rf $OUT/pmc_i_[0-9]
cat $OUT/${LEG}_inst_summary.txt
