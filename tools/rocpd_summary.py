#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd SQLite result (`*_results.db`) into a per-kernel table:
calls, total/avg/min/max duration and, when the run collected --pmc counters, the per-dispatch
average of every counter over its hardware instances (`=value xN`: N instances per dispatch, total = value x N).  Usage: rocpd_summary.py results.db [out.md]
`--timeline N` (after the db): instead of the table, the LAST N dispatches in start order (start / end in ms from the first of them, kernel, grid)."""
import sqlite3
import sys
from collections import defaultdict


def main():
    db = sqlite3.connect(sys.argv[1])
    tabs = [r[0] for r in db.execute("select name from sqlite_master where type='table'")]
    t = lambda key: [x for x in tabs if x.startswith(key)][0]
    kd, ks = t("rocpd_kernel_dispatch"), t("rocpd_info_kernel_symbol")
    names = {r[0]: r[1] for r in db.execute(f"select id, kernel_name from {ks}")}
    rows = db.execute(f"select id, kernel_id, start, end, grid_size_x, grid_size_y, workgroup_size_x, event_id from {kd}").fetchall()
    if len(sys.argv) > 3 and sys.argv[2] == "--timeline":
        last = sorted(rows, key=lambda r: r[2])[-int(sys.argv[3]):]
        t0 = last[0][2]
        for did, kid, st, en, gx, gy, wx, ev in last:
            print(f"{(st - t0) / 1e6:9.3f} {(en - t0) / 1e6:9.3f}  {(en - st) / 1e6:7.3f} ms  {names[kid].split('(')[0][:60]:60s} {gx}x{gy}")
        return
    stats = defaultdict(list)
    disp_kernel = {}
    for did, kid, st, en, gx, gy, wx, ev in rows:
        stats[names[kid]].append((en - st, gx, gy, wx))
        disp_kernel[ev] = names[kid]
    pmc = defaultdict(lambda: defaultdict(list))
    try:
        pe, pi = t("rocpd_pmc_event"), t("rocpd_info_pmc")
        pnames = {r[0]: r[1] for r in db.execute(f"select id, name from {pi}")}
        for ev, pid, val in db.execute(f"select event_id, pmc_id, value from {pe}"):
            if ev in disp_kernel:
                pmc[disp_kernel[ev]][pnames[pid]].append(val)
    except Exception as e:  # no counters in this run
        pass
    total = sum(sum(d for d, *_ in v) for v in stats.values())
    lines = ["| kernel | calls | total ms | avg ms | min ms | max ms | % | grid | block |", "|---|---|---|---|---|---|---|---|---|"]
    for k, v in sorted(stats.items(), key=lambda kv: -sum(d for d, *_ in kv[1])):
        ds = [d for d, *_ in v]
        short = k.split("(")[0]
        lines.append(f"| {short} | {len(ds)} | {sum(ds)/1e6:.3f} | {sum(ds)/len(ds)/1e6:.4f} | {min(ds)/1e6:.4f} | {max(ds)/1e6:.4f} | "
                     f"{100*sum(ds)/total:.1f} | {v[0][1]}x{v[0][2]} | {v[0][3]} |")
    if pmc:
        lines.append("")
        lines.append("PMC counters (average per dispatch):")
        for k, cs in pmc.items():
            short = k.split("(")[0]
            ncalls = len(stats[k])   # (a counter has one row per hardware instance and dispatch: `xN` = instances, total per dispatch = average x N)
            lines.append(f"- {short}: " + ", ".join(f"{c}={sum(v)/len(v):.4g}x{len(v)//max(1, ncalls)}" for c, v in sorted(cs.items())))
    out = "\n".join(lines)
    print(out)
    if len(sys.argv) > 2:
        open(sys.argv[2], "w").write(out + "\n")


if __name__ == "__main__":
    main()
