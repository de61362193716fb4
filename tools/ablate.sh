#!/bin/bash
# Ablation builds of the sweep (profiling only): which part of the loop costs what.  Run on the GPU box.
# PCP_ABLATE bits: 1 no LDS reads, 2 no arithmetic, 4 no record stream, 8 no live-word I/O.
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/ablate
for ab in 0 1 2 3 4 8 12 15; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC -DPCP_ABLATE=$ab pcp_amd/csrc/pcp_api.hip pcp_amd/csrc/pcp_kernels.hip -o gpurun_out/ablate/lib$ab.so 2>/dev/null
done
# the frontier is generated with the real library first and cached by bench.py? no: each run regenerates it with the ablated
# kernel, so only compare kernel_ms (the ablated statuses are meaningless).
for ab in 0 1 2 3 4 8 12 15; do
  PCP_HIP_LIB=$PWD/gpurun_out/ablate/lib$ab.so python bench.py --steps 10 --warmup 2 --cpu-budget 0 "$@" 2>&1 | python tools/benchline.py ablate=$ab
done
rm -f gpurun_out/ablate/*.so
