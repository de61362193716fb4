#!/usr/bin/env python
"""Turn the FETCH_SIZE / WRITE_SIZE lines of a tools/profile_round.sh summary into profiles/<tag>_traffic.json, which bench.py
reads for `roofline.traffic`.  Correction exactly as /opt/skills/guides/MI355X_MICROARCH.md §HBM prescribes: on gfx950 FETCH_SIZE
reports half the bytes of a wide coalesced read ("double it before comparing with a byte count"); WRITE_SIZE is taken as reported
(uncalibrated in the guide).  Both are in KiB per dispatch, from separate --pmc passes.
usage: traffic_json.py summary.txt kernel-substring grid n nodes_per_launch active out.json [launch-equivalents]
launch-equivalents: when the dispatches of that kernel and grid are not all bench launches (round 4: persistent workgroups give the
8192-node last level of the frontier generation the same grid as the 16384-node launches), the counters' SUM is divided by this number
of bench-launch equivalents (6 launches + half a launch = 6.5) instead of taking the per-dispatch average."""
import json
import re
import sys

summary, kern, grid, n, nodes, active, out = sys.argv[1], sys.argv[2], sys.argv[3], int(sys.argv[4]), int(sys.argv[5]), sys.argv[6], sys.argv[7]
equiv = float(sys.argv[8]) if len(sys.argv) > 8 else 0.0
cur, vals, dur = None, {}, None
for line in open(summary):
    m = re.match(r"kernel (\S+) grid=(\d+)", line)
    if m:
        cur = (m.group(1), m.group(2))
    m = re.match(r"\s+(FETCH_SIZE|WRITE_SIZE)\s+sum=\s*(\S+)\s+per_dispatch=\s*(\S+)", line)
    if m and cur and kern in cur[0] and cur[1] == grid:
        vals[m.group(1)] = float(m.group(2)) / equiv if equiv else float(m.group(3))
    m = re.match(r"\s*(\d+)\s+\S+\s+(\S+)\s+\S+\s+\S+\s+\S+\s+\d+\s+\d+\s+\d+\s+(\S+) grid=(\d+)", line)
    if m and kern in m.group(3) and m.group(4) == grid:
        dur = float(m.group(2))
records = 3 * n * (n - 1) // 2                      # N-queens x[i] != x[j] + k decomposition
words = (records + 63) // 64
compulsory = nodes * (8 * n + (8 * words if active == "explicit" else 0))  # reads the contract forces (bench.py node_bytes)
fetch, write = vals["FETCH_SIZE"] * 1024, vals["WRITE_SIZE"] * 1024
d = {
    "n": n, "nodes_per_launch": nodes, "active": active, "kernel": kern, "grid": int(grid),
    "fetch_kib_per_launch": vals["FETCH_SIZE"], "write_kib_per_launch": vals["WRITE_SIZE"],
    "traffic_bytes": 2 * fetch + write, "compulsory_bytes": compulsory,
    "rocprof_avg_kernel_us": dur,
    "source": summary,
    "launch_equivalents": equiv or None,
    "note": "traffic = 2 x FETCH_SIZE + WRITE_SIZE (MI355X_MICROARCH.md: gfx950 FETCH_SIZE counts 64 B per 128 B request), separate --pmc passes, "
            "per-dispatch averages over the bench launches (same kernel instantiation and grid)",
}
json.dump(d, open(out, "w"), indent=1)
print(d)
