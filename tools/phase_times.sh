cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/ablate
for ab in ${ABLS:-64 320}; do
python tools/build_variant.py $ab gpurun_out/ablate/lib$ab.so 2>/dev/null
for act in ${ACTS:-implicit explicit}; do PCP_ACTIVE=$act PCP_HIP_LIB=$PWD/gpurun_out/ablate/lib$ab.so python tools/phase_times.py "$@" 2>&1 | grep -v amdgpu | tail -3; done
done
rm -f gpurun_out/ablate/*.so
