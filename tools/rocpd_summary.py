#!/usr/bin/env python
"""Summarise a rocprofv3 rocpd database (the default output of `rocprofv3 --kernel-trace --stats` / `--pmc` on
ROCm 7.2) as plain text: per-kernel call count / total / average duration, and per-kernel PMC counter sums and
per-dispatch averages.  Usage: tools/rocpd_summary.py results.db [kernel-name-filter]"""
import sqlite3
import sys
from collections import defaultdict


def main():
    db = sys.argv[1]
    filt = sys.argv[2] if len(sys.argv) > 2 else ""
    c = sqlite3.connect(db)
    tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]

    def T(p):
        return [t for t in tabs if t.startswith(p)][0]

    names = dict(c.execute(f"select id, kernel_name from {T('rocpd_info_kernel_symbol')}"))
    rows = list(c.execute(f"select id, kernel_id, start, end, grid_size_x, workgroup_size_x, group_segment_size, event_id from {T('rocpd_kernel_dispatch')} order by start"))
    agg = defaultdict(list)
    ev2k = {}
    for _id, kid, s, e, gx, wx, lds, ev in rows:
        # one line per (kernel, grid): the same instantiation runs the bench batch and, e.g., a 4096-node side leg
        key = f"{names.get(kid, str(kid))[:86]} grid={gx // max(wx, 1)}"
        agg[key].append((e - s, gx, wx, lds))
        ev2k[ev] = key
    tot = sum(sum(d[0] for d in v) for v in agg.values()) or 1
    print(f"# kernel stats from {db}")
    print(f"{'calls':>6} {'total_ms':>12} {'avg_us':>12} {'min_us':>10} {'max_us':>10} {'pct':>6}  grid  wg  lds  name")
    for k, v in sorted(agg.items(), key=lambda kv: -sum(d[0] for d in kv[1])):
        if filt and filt not in k:
            continue
        d = [x[0] for x in v]
        print(f"{len(d):6d} {sum(d)/1e6:12.3f} {sum(d)/len(d)/1e3:12.2f} {min(d)/1e3:10.2f} {max(d)/1e3:10.2f} {100*sum(d)/tot:6.2f}  {v[-1][1]}  {v[-1][2]}  {v[-1][3]}  {k[:110]}")
    n_pmc = c.execute(f"select count(*) from {T('rocpd_pmc_event')}").fetchone()[0]
    if n_pmc:
        pmc = dict(c.execute(f"select id, name from {T('rocpd_info_pmc')}"))
        sums = defaultdict(lambda: defaultdict(float))
        for ev, pid, val in c.execute(f"select event_id, pmc_id, value from {T('rocpd_pmc_event')}"):
            sums[ev2k.get(ev, '?')][pmc.get(pid, str(pid))] += val
        print("\n# PMC counters: sum over all dispatches of the kernel, and per-dispatch average")
        for k, d in sums.items():
            if filt and filt not in k:
                continue
            n = len(agg.get(k, [])) or 1
            print(f"kernel {k[:110]}  dispatches={n}")
            for name, val in sorted(d.items()):
                print(f"   {name:28s} sum={val:18.0f}  per_dispatch={val/n:18.1f}")


if __name__ == "__main__":
    main()
