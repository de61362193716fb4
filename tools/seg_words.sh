cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/ablate
for ab in ${ABL:-144 128}; do
python tools/build_variant.py $ab gpurun_out/ablate/lib$ab.so 2>/dev/null
echo "ablate $ab: $(PCP_HIP_LIB=$PWD/gpurun_out/ablate/lib$ab.so python tools/seg_words.py 2>&1 | grep -v amdgpu | tail -2 | tr "\n" " ")"
done
rm -f gpurun_out/ablate/*.so
