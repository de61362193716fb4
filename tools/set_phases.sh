cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/ablate
python tools/build_variant.py 2048 gpurun_out/ablate/lib2048.so 2>/dev/null
PCP_HIP_LIB=$PWD/gpurun_out/ablate/lib2048.so python tools/set_phases.py 2>&1 | grep -v amdgpu | tail -3
rm -f gpurun_out/ablate/*.so
