cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/ablate
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC -DPCP_ABLATE=320 pcp_amd/csrc/pcp_api.hip pcp_amd/csrc/pcp_kernels.hip -o gpurun_out/ablate/lib320.so 2>/dev/null
PCP_HIP_LIB=$PWD/gpurun_out/ablate/lib320.so python tools/phase_bench.py "$@" 2>&1 | tail -1
rm -f gpurun_out/ablate/*.so
