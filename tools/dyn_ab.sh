# A/B of the tile tickets (option neq_dynamic = tiles per workgroup taken by the fixed stride before it draws; 0 = never) on one box
mkdir -p gpurun_out/$1
NEQ_CONFIGS='[{"neq_dynamic": 2}, {"neq_dynamic": 0}, {"neq_dynamic": 2}, {"neq_dynamic": 0}, {"neq_dynamic": 1}]' timeout 300 python tools/neq_probe.py frontier 2>&1 | grep -v amdgpu > gpurun_out/$1/probe.txt
timeout 200 python tools/wide_batch.py neq_dynamic=0 2>&1 | grep -v amdgpu > gpurun_out/$1/wide_static.txt
timeout 200 python tools/wide_batch.py neq_dynamic=2 2>&1 | grep -v amdgpu > gpurun_out/$1/wide_ticket2.txt
timeout 200 python tools/wide_batch.py neq_dynamic=1 2>&1 | grep -v amdgpu > gpurun_out/$1/wide_ticket1.txt
[ -n "${MIX:-}" ] && MIX_CONFIGS='[{"neq_dynamic": 2}, {"neq_dynamic": 1}, {"neq_dynamic": 0}, {"neq_dynamic": 1, "nodes_per_block": 4}, {"neq_dynamic": 0, "nodes_per_block": 4}]' timeout 300 python tools/mix_tiles.py 2>&1 | grep -v amdgpu > gpurun_out/$1/mix.txt
for f in probe wide_static wide_ticket2 wide_ticket1 mix; do echo "== $f"; cat gpurun_out/$1/$f.txt 2>/dev/null; done
