cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/ablate
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC -DPCP_ABLATE=128 pcp_amd/csrc/pcp_api.hip pcp_amd/csrc/pcp_kernels.hip -o gpurun_out/ablate/lib128.so 2>/dev/null
PCP_HIP_LIB=$PWD/gpurun_out/ablate/lib128.so python ${SEG_TOOL:-tools/seg_times.py} 2>&1 | tail -7
rm -f gpurun_out/ablate/*.so
