#!/bin/bash
# A/B of two builds of the library on the headline launch, alternating on ONE box (run through gpurun):
#   bash tools/ab_headline.sh <tag> <variant.so> [rounds]
# prints the median HIP-event kernel time of `bench.py --legs none --cpu-budget 0` per run; PCP_HIP_LIB selects the build.
set -u
TAG=$1; VAR=$2; R=${3:-3}
OUT=gpurun_out/$TAG; mkdir -p $OUT
for i in $(seq 1 $R); do
  for which in default variant; do
    if [ $which = variant ]; then export PCP_HIP_LIB=$PWD/$VAR; else unset PCP_HIP_LIB; fi
    timeout 200 python bench.py --legs none --cpu-budget 0 --steps 20 > $OUT/ab_${which}_$i.json 2> $OUT/ab_${which}_$i.err
    python - "$OUT/ab_${which}_$i.json" "$which" "$i" <<'PY'
import json, sys
for l in open(sys.argv[1]):
    if l.startswith('{') and '"metric"' in l:
        o = json.loads(l); k = o["config"]["kernel_ms_per_launch"]
        print(f"{sys.argv[2]:8s} run {sys.argv[3]}: kernel_ms min {k['min']:.4f} median {k['median']:.4f} max {k['max']:.4f}", flush=True)
PY
  done
done
