#!/usr/bin/env python
"""Turn the FETCH_SIZE / WRITE_SIZE lines of a tools/profile_bench.sh summary into profiles/<tag>_traffic.json.
HBM bytes per launch of the dominant kernel = 2 x FETCH_SIZE + WRITE_SIZE (KiB -> bytes): on gfx950 FETCH_SIZE reads
half the bytes of a wide coalesced stream (MI355X_MICROARCH.md §HBM); applying the x2 to the whole fetch count is an
upper bound because the 8-byte `active`-word reads are not calibrated.
usage: traffic_json.py summary.txt kernel-substring n nodes_per_launch out.json"""
import json
import re
import sys

summary, kern, n, nodes, out = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), sys.argv[5]
cur, vals = None, {}
for line in open(summary):
    m = re.match(r"kernel (\S+)", line)
    if m:
        cur = m.group(1)
    m = re.match(r"\s+(FETCH_SIZE|WRITE_SIZE)\s+sum=\s*\S+\s+per_dispatch=\s*(\S+)", line)
    if m and cur and kern in cur:
        vals[m.group(1)] = float(m.group(2))
d = {
    "n": n, "nodes_per_launch": nodes, "kernel": kern,
    "fetch_kib_per_launch": vals["FETCH_SIZE"], "write_kib_per_launch": vals["WRITE_SIZE"],
    "traffic_bytes": (2 * vals["FETCH_SIZE"] + vals["WRITE_SIZE"]) * 1024,
    "source": summary, "note": "2*FETCH_SIZE + WRITE_SIZE, separate --pmc passes, gfx950 FETCH_SIZE x2 correction applied to all fetches (upper bound)",
}
json.dump(d, open(out, "w"), indent=1)
print(d)
