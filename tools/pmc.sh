#!/bin/bash
# Extra rocprofv3 PMC passes on bench.py (run on the GPU box):  bash tools/pmc.sh <tag> "<counters pass 1>" "<counters pass 2>" ...
# Each pass is --pmc only (no tracing domains).  BENCH_ARGS adds bench.py options.
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/$TAG; mkdir -p $OUT; : > $OUT/summary.txt
i=0
for CTRS in "$@"; do
  i=$((i+1))
  timeout 150 rocprofv3 --pmc $CTRS -d $OUT/pmc$i -o p -- python bench.py --steps 5 --warmup 1 --cpu-budget 0 ${BENCH_ARGS:-} > $OUT/pmc$i.log 2>&1
  echo "# --pmc $CTRS" >> $OUT/summary.txt
  python tools/rocpd_summary.py $OUT/pmc$i/p_results.db fixpoint_kernelILi16ELb0ELb1ELb1 >> $OUT/summary.txt 2>&1
done
rm -rf $OUT/pmc[0-9]*/
grep -v "^ *[0-9]* .*fixpoint\|^ calls\|^# kernel stats\|^$" $OUT/summary.txt
