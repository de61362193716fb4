#!/bin/bash
# Ablation builds of the sweep (profiling only): which part of the loop costs what.  Run on the GPU box.
# PCP_ABLATE bits: 1 no LDS reads, 2 no arithmetic, 8 no live-word I/O, 16 skip the sweep's cold part, 32 skip the rounds.
# CMD='python tools/deep_frontier.py 3000' ABS='0 16 32 48' bash tools/ablate.sh   runs another command per build.
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/ablate
for ab in ${ABS:-0 1 2 3 8 16 32 48}; do
  python tools/build_variant.py $ab gpurun_out/ablate/lib$ab.so 2>/dev/null
done
# the frontier is generated with the real library first and cached by bench.py? no: each run regenerates it with the ablated
# kernel, so only compare kernel_ms (the ablated statuses are meaningless).
for ab in ${ABS:-0 1 2 3 8 16 32 48}; do
  if [ -n "${CMD:-}" ]; then echo "ablate=$ab: $(PCP_HIP_LIB=$PWD/gpurun_out/ablate/lib$ab.so $CMD 2>&1 | tail -1)"
  else PCP_HIP_LIB=$PWD/gpurun_out/ablate/lib$ab.so python bench.py --steps 10 --warmup 2 --cpu-budget 0 "$@" 2>&1 | python tools/benchline.py ablate=$ab; fi
done
rm -f gpurun_out/ablate/*.so
