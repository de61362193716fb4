cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/ablate
python tools/build_variant.py 128 gpurun_out/ablate/lib128.so 2>/dev/null
PCP_HIP_LIB=$PWD/gpurun_out/ablate/lib128.so python ${SEG_TOOL:-tools/seg_times.py} 2>&1 | tail -7
rm -f gpurun_out/ablate/*.so
