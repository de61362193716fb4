#!/bin/bash
# SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE of the config-3 launch with and without the bank-aware table order (option big_bank): one --pmc pass each
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for b in 1 0; do
  OUT=gpurun_out/c3bank_$b; mkdir -p $OUT
  PCP_SET_OPTIONS=big_bank=$b timeout 600 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU -d $OUT/pmc -o p -- python tools/replay_leg.py run c3 > $OUT/log.txt 2>&1
  echo "== big_bank=$b"; grep '^{' $OUT/log.txt | head -2
  python tools/rocpd_summary.py $OUT/pmc/p_results.db bigfix | sed -n '/# PMC counters/,$p'
  rm -rf $OUT/pmc
done
