cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r04a; mkdir -p $OUT
rocprofv3 -L > $OUT/counters.txt 2>&1
grep -i -o "SQC_[A-Z_0-9]*\|SQ_IFETCH[A-Z_0-9]*\|SQ_INST_CACHE[A-Z_0-9]*" $OUT/counters.txt | sort -u > $OUT/counter_names.txt
CMD="python tools/replay_leg.py run frontier"
i=0
for CTRS in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VALU SQ_WAVES"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $CTRS -d $OUT/pmc_ic_$i -o p -- $CMD > $OUT/ic_pmc$i.log 2>&1
  echo "# --pmc $CTRS" >> $OUT/icache_summary.txt
  python tools/rocpd_summary.py $OUT/pmc_ic_$i/p_results.db neqfix >> $OUT/icache_summary.txt 2>&1
done
rm -rf $OUT/pmc_ic_[0-9]
