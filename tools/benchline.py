#!/usr/bin/env python
"""Pretty-print the interesting fields of bench.py's JSON line(s) read from stdin."""
import json
import sys

for line in sys.stdin:
    line = line.strip()
    if not line.startswith("{"):
        continue
    d = json.loads(line)
    r = d.get("roofline", {})
    print(f"{' '.join(sys.argv[1:])} value={d['value']/1e9:.1f} Gsteps/s ms_per_step={d['ms_per_step']:.3f} kernel_ms={r.get('kernel_ms', 0):.3f} "
          f"alg_GB/s={r.get('achieved', 0):.0f} frac={r.get('frac', 0):.3f} n_gpus={d['n_gpus']}")
