cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/ablate
python tools/build_variant.py 512 gpurun_out/ablate/lib512.so 2>/dev/null
PCP_HIP_LIB=$PWD/gpurun_out/ablate/lib512.so python bench.py --cpu-budget 0 --nodes 4096 --steps 2 --warmup 1 | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('steps3 (hard words per launch):', d['config']['filter_steps_per_step_per_gpu']-0, d['config'])"
rm -f gpurun_out/ablate/*.so
