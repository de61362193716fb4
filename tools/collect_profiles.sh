#!/bin/bash
# Copy what a tools/final_run.sh pass left under gpurun_out/<tag>/ into profiles/ under the round's names, and derive the JSON files bench.py
# reads:   bash tools/collect_profiles.sh <tag> <rNN>        (run here, after the gpurun call has merged its output)
set -u
T=gpurun_out/$1; R=$2
for L in frontier cells deep500 deep3000 mix mixh search neqforest c3 c4 f4 explicit setforest; do
  [ -f $T/${L}_summary.txt ] && cp $T/${L}_summary.txt profiles/${R}_${L}_rocprofv3_summary.txt
done
[ -f $T/forest8k/neqforest_summary.txt ] && cp $T/forest8k/neqforest_summary.txt profiles/${R}_neqforest8k_rocprofv3_summary.txt
[ -f $T/benchcmd/summary.txt ] && cp $T/benchcmd/summary.txt profiles/${R}_benchcmd_rocprofv3_summary.txt
[ -f $T/phases/phases_summary.txt ] && cp $T/phases/phases_summary.txt profiles/${R}_phases_summary.txt
[ -f $T/neq_probe.txt ] && {
  echo "# the box's ceilings and the round's stand-alone probes, same gpurun call as the profiles (tools/final_run.sh)"
  echo "## tools/box_probe.py (tools/micro/box_probe.hip)"; cat $T/box_probe.json 2>/dev/null
  echo; echo "## tools/micro/stream_probe (D = 0, 10000, 20000 cycles of synthetic compute per tile) and tools/micro/stage_probe (D = 20000)"; cat $T/stream_probe.txt 2>/dev/null
  echo; echo "## tools/neq_probe.py frontier: the launch under neq_stagger (cycles), the old staging loop (neq_debug 32768) and phases switched off (neq_debug 3 staging only (+32768: old loop), 1 no rounds, 2 no status scan)"
  grep -v amdgpu $T/neq_probe.txt 2>/dev/null
} > profiles/${R}_probes.txt
python tools/resource_usage.py > profiles/${R}_resource_usage.txt 2>/dev/null
python tools/valu_json.py $R > profiles/${R}_valu_counts.json
# headline traffic: the 7 grid=512 dispatches of the frontier replay = 6 launches + the 8192-node last level of the frontier generation (6.5 launch equivalents)
python tools/traffic_json.py profiles/${R}_frontier_rocprofv3_summary.txt neqfix_kernelILb1ELb1ELb0ELi16E 512 1000 16384 implicit profiles/${R}_headline_traffic.json 6.5
ls -la profiles/${R}_*
# the average duration of the six full launches (the seventh grid=512 dispatch, the shortest, is the half-size level of the frontier generation),
# and the average over the bench command's own dispatches
python - "$R" <<'PY'
import json, re, sys
R = sys.argv[1]
def row(path, kern, grid):
    for line in open(path):
        m = re.match(r"\s*(\d+)\s+(\S+)\s+(\S+)\s+(\S+)\s+(\S+)\s+\S+\s+\d+\s+\d+\s+\d+\s+(\S+) grid=(\d+)", line)
        if m and kern in m.group(6) and m.group(7) == grid:
            return int(m.group(1)), float(m.group(2)) * 1e3, float(m.group(4))
    return None
p = f"profiles/{R}_headline_traffic.json"
d = json.load(open(p))
fr = row(f"profiles/{R}_frontier_rocprofv3_summary.txt", d["kernel"], "512")
bc = row(f"profiles/{R}_benchcmd_rocprofv3_summary.txt", d["kernel"], "512")
if fr:
    d["rocprof_avg_kernel_us"] = round((fr[1] - fr[2]) / (fr[0] - 1), 2)
if bc:
    d["rocprof_avg_kernel_us_bench_command"] = round((bc[1] - bc[2]) / (bc[0] - 1), 2)
d["note"] += (f"  The {fr[0] if fr else '?'} dispatches with grid=512 of tools/replay_leg.py run frontier are 6 launches of the 16384-node batch and the 8192-node last level of the "
              "frontier generation (persistent workgroups: same grid): counter sums / 6.5 launch equivalents; rocprof_avg_kernel_us = (total - the shortest, half-size dispatch) / the others; "
              f"rocprof_avg_kernel_us_bench_command = the same over the grid=512 dispatches of `python bench.py --legs none --cpu-budget 0` ({R}_benchcmd_rocprofv3_summary.txt).")
json.dump(d, open(p, "w"), indent=1)
print(d["rocprof_avg_kernel_us"], d.get("rocprof_avg_kernel_us_bench_command"))
PY
