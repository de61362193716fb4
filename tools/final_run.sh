set -u
T=$1
mkdir -p gpurun_out/$T
timeout 900 python -m pytest tests -m gpu -q > gpurun_out/$T/gputests.log 2>&1; echo "pytest exit $?" >> gpurun_out/$T/gputests.log
timeout 400 python bench.py > gpurun_out/$T/bench.json 2> gpurun_out/$T/bench.err
cp gpurun_out/bench_legs.json gpurun_out/$T/bench_legs.json 2>/dev/null
timeout 200 python bench.py --mode search > gpurun_out/$T/search_interval.json 2> gpurun_out/$T/search_interval.err
timeout 200 python bench.py --mode search --domains set > gpurun_out/$T/search_set.json 2> gpurun_out/$T/search_set.err
timeout 300 python tools/replay_leg.py save deep500 deep3000 > gpurun_out/$T/save.log 2>&1
for L in frontier deep500 deep3000 search neqforest; do timeout 500 bash tools/profile_leg.sh $T $L neqfix > gpurun_out/$T/prof_$L.log 2>&1; done
timeout 500 bash tools/profile_leg.sh $T c3 bigfix > gpurun_out/$T/prof_c3.log 2>&1
timeout 500 bash tools/profile_leg.sh $T c4 fixpoint > gpurun_out/$T/prof_c4.log 2>&1
timeout 500 bash tools/profile_leg.sh $T setforest "setdfs" > gpurun_out/$T/prof_setforest.log 2>&1
grep -h "passed\|failed" gpurun_out/$T/gputests.log | tail -2
