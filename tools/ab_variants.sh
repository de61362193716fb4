for rep in 1 2; do
for v in base nt sc1 ntsc; do
  lib=pcp_amd/libpcp_hip_$v.so; [ $v == base ] && lib=pcp_amd/libpcp_hip.so
  echo "== $v rep $rep"
  PCP_HIP_LIB=$lib NEQ_CONFIGS='[{}]' timeout 200 python tools/neq_probe.py frontier 2>&1 | grep -v amdgpu
done; done
