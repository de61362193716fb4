# VALU / SALU / LDS wave-instructions of the headline launch by phase: the launch with phases switched off (neq_debug 3 = staging only,
# 1 = staging + status scan, 0 = everything).  bash tools/pmc_phases.sh <tag>
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/${1:-r05phases}; mkdir -p $OUT
: > $OUT/phases_summary.txt
for D in 3 2 1 0; do
  echo "## neq_debug=$D" >> $OUT/phases_summary.txt
  PCP_OPTS=neq_debug=$D timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_WAVE_CYCLES SQ_WAVES -d $OUT/pmc_$D -o p -- python tools/replay_leg.py run frontier > $OUT/pmc_$D.log 2>&1
  timeout 60 python tools/rocpd_summary.py $OUT/pmc_$D/p_results.db neqfix 2>&1 | sed -n '/# PMC counters/,$p' >> $OUT/phases_summary.txt
  rm -rf $OUT/pmc_$D
done
awk '/^## neq_debug/{d=$0} /grid=512  dispatches=[56]/{k=1;print d; next} /^kernel/{k=0} k&&/INSTS_VALU|INSTS_SALU|INSTS_LDS|WAVE_CYCLES|ACTIVE_INST/{print}' $OUT/phases_summary.txt
