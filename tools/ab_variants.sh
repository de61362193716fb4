#!/bin/bash
# A/B of library variants on ONE box: bash tools/ab_variants.sh <variant names...>  (base = the product library; variants from tools/build_neq_variant.py)
for rep in 1 2; do
for v in "$@"; do
  lib=pcp_amd/libpcp_hip_$v.so; [ $v == base ] && lib=pcp_amd/libpcp_hip.so
  echo "== $v rep $rep"
  PCP_HIP_LIB=$lib NEQ_CONFIGS='[{}]' timeout 200 python tools/neq_probe.py frontier 2>&1 | grep -v amdgpu
done; done
