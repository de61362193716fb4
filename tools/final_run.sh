# One gpurun call at the end of a round:  bash tools/final_run.sh <tag> [notests]      (LITE=1 NEQ_LEGS="frontier cells search neqforest": only the legs whose kernels changed)
# GPU suite, the bench as the driver runs it, the two search benches, then rocprofv3 evidence per leg (tools/profile_leg.sh: kernel trace +
# four PMC passes around tools/replay_leg.py) and a kernel trace of the bench command itself.  Everything lands in gpurun_out/<tag>/.
set -u
T=$1
mkdir -p gpurun_out/$T
if [ "${2:-}" != "notests" ]; then
  timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/$T/gputests.log 2>&1; echo "pytest exit $?" >> gpurun_out/$T/gputests.log
fi
timeout 600 python bench.py > gpurun_out/$T/bench.json 2> gpurun_out/$T/bench.err
cp gpurun_out/bench_legs.json gpurun_out/$T/bench_legs.json 2>/dev/null
timeout 200 python bench.py --mode search > gpurun_out/$T/search_interval.json 2> gpurun_out/$T/search_interval.err
timeout 200 python bench.py --mode search --engine worklist > gpurun_out/$T/search_worklist.json 2> gpurun_out/$T/search_worklist.err
timeout 200 python bench.py --mode search --engine worklist --cells > gpurun_out/$T/search_worklist_cells.json 2> gpurun_out/$T/search_worklist_cells.err
timeout 200 python bench.py --mode search --domains set > gpurun_out/$T/search_set.json 2> gpurun_out/$T/search_set.err
timeout 300 python bench.py --legs none --cpu-budget 0 --c5-single > gpurun_out/$T/bench_c5_single.json 2> gpurun_out/$T/bench_c5_single.err
timeout 400 bash tools/profile_cmd.sh $T/benchcmd python bench.py --legs none --cpu-budget 0 > gpurun_out/$T/prof_benchcmd.log 2>&1
[ -z "${LITE:-}" ] && timeout 300 python tools/replay_leg.py save deep500 deep3000 > gpurun_out/$T/save.log 2>&1
for L in ${NEQ_LEGS:-frontier cells deep500 deep3000 mix mixh search neqforest}; do timeout 600 bash tools/profile_leg.sh $T $L neqfix > gpurun_out/$T/prof_$L.log 2>&1; done
PCP_FOREST_8K=1 timeout 600 bash tools/profile_leg.sh $T/forest8k neqforest neqfix > gpurun_out/$T/prof_neqforest8k.log 2>&1
[ -z "${LITE:-}" ] && timeout 300 bash tools/pmc_phases.sh $T/phases > gpurun_out/$T/prof_phases.log 2>&1
timeout 120 python tools/box_probe.py > gpurun_out/$T/box_probe.json 2> /dev/null
(cd tools/micro && for D in 0 10000 20000; do timeout 60 ./stream_probe 16384 $D 5; done; timeout 60 ./stage_probe 16384 20000 5) > gpurun_out/$T/stream_probe.txt 2>&1
[ -z "${LITE:-}" ] && NEQ_CONFIGS='[{}, {"neq_stagger": 6000}, {"neq_stagger": 12000}, {"neq_debug": 32768}, {"neq_debug": 3}, {"neq_debug": 32771}, {"neq_debug": 1}, {"neq_debug": 2}]' timeout 300 python tools/neq_probe.py frontier > gpurun_out/$T/neq_probe.txt 2>&1
if [ -z "${LITE:-}" ]; then
timeout 500 bash tools/profile_leg.sh $T c3 bigfix > gpurun_out/$T/prof_c3.log 2>&1
timeout 500 bash tools/profile_leg.sh $T c4 smallfix > gpurun_out/$T/prof_c4.log 2>&1
timeout 500 bash tools/profile_leg.sh $T f4 formfix > gpurun_out/$T/prof_f4.log 2>&1
timeout 500 bash tools/profile_leg.sh $T explicit fixpoint > gpurun_out/$T/prof_explicit.log 2>&1
timeout 500 bash tools/profile_leg.sh $T setforest "setdfs" > gpurun_out/$T/prof_setforest.log 2>&1
fi
# leftovers of a profile pass that hit its timeout (rocpd databases are hundreds of MB: gpurun copies back at most 64 MiB)
find gpurun_out/$T -type d \( -name "trace_*" -o -name "pmc_*" -o -name trace \) -prune -exec rm -rf {} + 2>/dev/null
find gpurun_out/$T -type f -size +4M -exec rm -f {} + 2>/dev/null
du -sh gpurun_out/$T
grep -h "passed\|failed" gpurun_out/$T/gputests.log 2>/dev/null | tail -2
tail -c 600 gpurun_out/$T/bench.json
