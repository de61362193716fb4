cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/ablate
python tools/build_variant.py 64 gpurun_out/ablate/lib64.so 2>/dev/null
PCP_HIP_LIB=$PWD/gpurun_out/ablate/lib64.so python tools/c3_phases.py "$@" 2>&1 | grep -v amdgpu | tail -3
rm -f gpurun_out/ablate/*.so
