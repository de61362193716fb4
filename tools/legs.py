#!/usr/bin/env python
"""One line per leg of a bench.py JSON line (its brief per-leg figures), or of the full legs record bench.py writes to stderr and
gpurun_out/bench_legs.json.  usage: python tools/legs.py < bench.json   |   python tools/legs.py gpurun_out/bench_legs.json"""
import json, sys
src = open(sys.argv[1]).read() if len(sys.argv) > 1 else sys.stdin.read()
d = json.loads([l for l in src.strip().splitlines() if l.startswith("{")][-1])
if "config" in d:
    c = d["config"]
    print(f"headline: value={d['value']:.3e} executed steps/s (reference-equivalent {c.get('steps_reference_equivalent_per_s', float('nan')):.3e})  nodes/s={c['nodes_per_s']:.3e}  ms/step={d['ms_per_step']:.3f}  kernel_ms={c['kernel_ms_per_launch']}  evaluated/s={c['steps_evaluated_per_s']:.3e}  full/s={c['full_filter_evals_per_s']:.3e}  "
          f"hbm_frac={d['roofline']['frac']:.3f}  eval/step={c['evaluated_per_step_per_gpu']:.3e} full/step={c['full_evals_per_step_per_gpu']:.3e} parity_nodes={c.get('parity_checked_nodes')}")
    legs = c.get("legs", [])
else:
    legs = d["legs"]
for l in legs:
    if isinstance(l.get("kernel_ms"), dict):
        print(f"{l['name']:45s} nodes={l['nodes']:6d} ms={l['kernel_ms']['median']:9.3f} steps/s={l['steps_per_s']:.3e} eval/s={l['evaluated_per_s']:.3e} eval={l['evaluated_per_launch']:.3e} full={l['full_evals_per_launch']:.3e} "
              f"narrow={l['narrowings_per_launch']:.0f} waves/node={l['waves_per_node']:.2f} hbm_frac={l['hbm_frac']:.4f} status={l['status_false_true_unknown']} B={l['plan']['nodes_per_block']} wl={l['plan'].get('word_level')}")
    elif "seconds" in l:
        print(f"{l['name']:45s} nodes={l['nodes']:6d} us/node={l['us_per_node']:.2f} nodes/s={l['nodes'] / l['seconds']:.3e} steps/s={l['steps_per_s']:.3e} kernel_us={l.get('last_kernel_us', float('nan')):.1f} team={l.get('plan', {}).get('team', '-')}")
    else:
        print(f"{l['name']:45s} " + " ".join(f"{k}={v}" for k, v in l.items() if k != "name"))
