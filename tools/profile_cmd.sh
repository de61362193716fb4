#!/bin/bash
# rocprofv3 --kernel-trace --stats summary of an arbitrary command (run on the GPU box):  bash tools/profile_cmd.sh <tag> <command...>
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/$TAG; mkdir -p $OUT
echo "# command: $*" > $OUT/summary.txt
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/trace -o t -- "$@" > $OUT/trace.log 2>&1
grep -v "amdgpu.ids\|^W2\|^E2\|^I2" $OUT/trace.log | tail -3 >> $OUT/summary.txt
python tools/rocpd_summary.py $OUT/trace/t_results.db "" >> $OUT/summary.txt
rm -rf $OUT/trace
cat $OUT/summary.txt
