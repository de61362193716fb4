cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/ablate
python tools/build_variant.py 1024 gpurun_out/ablate/lib1024.so 2>/dev/null
PCP_HIP_LIB=$PWD/gpurun_out/ablate/lib1024.so python tools/round_hist.py 2>&1 | grep -v amdgpu | tail -6
rm -f gpurun_out/ablate/*.so
