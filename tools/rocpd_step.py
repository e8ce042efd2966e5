#!/usr/bin/env python3
"""One steady-state step of a pipelined run in a rocprofv3 rocpd result, per kernel: first start, last end, launches, busy time —
relative to the start of the step's sequential kernel (the step = from one K5 start to the next).  Usage: rocpd_step.py results.db [step-from-the-end=2]"""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
back = int(sys.argv[2]) if len(sys.argv) > 2 else 2
tabs = [r[0] for r in db.execute("select name from sqlite_master where type='table'")]
t = lambda key: [x for x in tabs if x.startswith(key)][0]
kd, ks = t("rocpd_kernel_dispatch"), t("rocpd_info_kernel_symbol")
names = {r[0]: r[1] for r in db.execute(f"select id, kernel_name from {ks}")}
rows = sorted(db.execute(f"select start, end, kernel_id, queue_id from {kd}").fetchall())
short = lambda n: ("K5" if "demod_wave" in n else "K2p" if "limit_track_persist" in n else "K2" if "limit_track" in n else "K1" if "fir_rrc" in n else "K3" if "dcd_" in n
                   else "K4d" if "decode_deferred" in n else "other")
k5 = [r for r in rows if short(names[r[2]]) == "K5"]
# with segmented K5 there are many launches per step: a step starts at the K5 launch that follows a deferred decode
starts = []
prev = None
for r in rows:
    s = short(names[r[2]])
    if s == "K5" and prev in (None, "K4d"):
        starts.append(r[0])
    if s in ("K5", "K4d"):
        prev = s
a, b = starts[-back - 1], starts[-back]
print(f"step of {(b - a) / 1e6:.2f} ms")
agg = {}
for st, en, kid, q in rows:
    if en <= a or st >= b + (b - a):
        continue
    s = short(names[kid])
    g = agg.setdefault((s, q), [st, en, 0, 0.0])
    g[0] = min(g[0], st); g[1] = max(g[1], en); g[2] += 1; g[3] += (en - st) / 1e6
for (s, q), g in sorted(agg.items(), key=lambda kv: kv[1][0]):
    print(f"{s:5s} q{q}: {(g[0] - a) / 1e6:8.2f} -> {(g[1] - a) / 1e6:8.2f} ms, {g[2]:3d} launches, busy {g[3]:7.2f} ms")
