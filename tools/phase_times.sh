cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/ablate
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC -DPCP_ABLATE=64 pcp_amd/csrc/pcp_api.hip pcp_amd/csrc/pcp_kernels.hip -o gpurun_out/ablate/lib64.so 2>/dev/null
PCP_HIP_LIB=$PWD/gpurun_out/ablate/lib64.so python tools/phase_times.py 2>&1 | grep -v amdgpu | tail -3
for wl in 0; do PCP_WORD_LEVEL=$wl PCP_HIP_LIB=$PWD/gpurun_out/ablate/lib64.so python tools/phase_times.py 2>&1 | grep -v amdgpu | tail -3; done
rm -f gpurun_out/ablate/*.so
