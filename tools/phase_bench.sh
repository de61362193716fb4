cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/ablate
python tools/build_variant.py 320 gpurun_out/ablate/lib320.so 2>/dev/null
PCP_HIP_LIB=$PWD/gpurun_out/ablate/lib320.so python tools/phase_bench.py "$@" 2>&1 | tail -1
rm -f gpurun_out/ablate/*.so
