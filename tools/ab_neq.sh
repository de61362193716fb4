#!/bin/bash
# A/B of library variants on ONE box:  bash tools/ab_neq.sh <lib suffixes...>   ("" = the product library)
for rep in 1 2; do
  for v in "$@"; do
    lib=pcp_amd/libpcp_hip${v:+_$v}.so
    [ "$v" == "base" ] && lib=pcp_amd/libpcp_hip.so
    echo "== $v ($lib) rep $rep"
    PCP_HIP_LIB=$lib timeout 200 python tools/neq_probe.py frontier deep500 deep3000 2>&1 | grep -v amdgpu | head -8
  done
done
