#!/usr/bin/env python
"""VALU wave-instructions per launch of each compute-bound bench leg, from the committed rocprofv3 PMC summaries (tools/profile_leg.sh):
   python tools/valu_json.py r05 > profiles/r05_valu_counts.json
For every profiles/<tag>_<leg>_rocprofv3_summary.txt: the dispatch group (kernel, grid) with the most SQ_INSTS_VALU per dispatch — the
leg's timed launch (its input-building launches are smaller) — and its SALU / LDS counts, VALU-active cycles and average duration.
bench.py reads the file and reports `<leg>_vfrac` = VALU wave-instructions per launch / live launch time / (the box's measured VALU issue
rate x CUs): the fraction of the integer-VALU issue ceiling the leg reaches (roofline bound "valu")."""
import glob, json, os, re, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r05"
out = {}
for f in sorted(glob.glob(os.path.join(ROOT, "profiles", f"{tag}_*_rocprofv3_summary.txt"))):
    leg = os.path.basename(f)[len(tag) + 1:-len("_rocprofv3_summary.txt")]
    groups, cur, durs = {}, None, {}
    for line in open(f):
        m = re.match(r"kernel (\S+) grid=(\d+)\s+dispatches=(\d+)", line)
        if m:
            cur = (m.group(1), int(m.group(2)))
            groups.setdefault(cur, {"dispatches": int(m.group(3))})
            continue
        m = re.match(r"\s+(SQ_\w+)\s+sum=\s*(\S+)\s+per_dispatch=\s*(\S+)", line)
        if m and cur:
            groups[cur][m.group(1)] = float(m.group(3))
            continue
        m = re.match(r"\s*(\d+)\s+(\S+)\s+(\S+)\s+\S+\s+\S+\s+\S+\s+\d+\s+\d+\s+\d+\s+(\S+) grid=(\d+)", line)
        if m:
            durs[(m.group(4), int(m.group(5)))] = float(m.group(3))
    best = max((g for g in groups.items() if "SQ_INSTS_VALU" in g[1]), key=lambda g: g[1]["SQ_INSTS_VALU"], default=None)
    if not best:
        continue
    (kern, grid), c = best
    out[leg] = {"kernel": kern, "grid": grid, "valu_per_launch": c.get("SQ_INSTS_VALU"), "salu_per_launch": c.get("SQ_INSTS_SALU"), "lds_per_launch": c.get("SQ_INSTS_LDS"),
                "valu_active_cycles_per_launch": c.get("SQ_ACTIVE_INST_VALU"), "wave_cycles_per_launch": c.get("SQ_WAVE_CYCLES"), "wait_any_cycles_per_launch": c.get("SQ_WAIT_ANY"),
                "rocprof_avg_kernel_us": durs.get((kern, grid)), "source": os.path.relpath(f, ROOT)}
json.dump(out, sys.stdout, indent=1)
