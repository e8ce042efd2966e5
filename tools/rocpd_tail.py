#!/usr/bin/env python3
"""Kernel timeline of the last `ms` milliseconds (default 60) of dispatches in a rocprofv3 rocpd SQLite result — for regimes without a
reset between steps (tools/stream_bench.py).  Usage: rocpd_tail.py results.db [ms] [skip_ms_from_end]"""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
span = float(sys.argv[2]) if len(sys.argv) > 2 else 60.0
skip = float(sys.argv[3]) if len(sys.argv) > 3 else 0.0
tabs = [r[0] for r in db.execute("select name from sqlite_master where type='table'")]
t = lambda key: [x for x in tabs if x.startswith(key)][0]
kd, ks = t("rocpd_kernel_dispatch"), t("rocpd_info_kernel_symbol")
names = {r[0]: r[1] for r in db.execute(f"select id, kernel_name from {ks}")}
rows = sorted(db.execute(f"select start, end, kernel_id, queue_id from {kd}").fetchall())
end = max(r[1] for r in rows) - skip * 1e6
t0 = end - span * 1e6
for st, en, kid, q in rows:
    if en < t0 or st > end:
        continue
    n = names[kid].split('(')[0].replace('m17::', '').replace('(anonymous namespace)::', '')
    print(f"{(st - t0) / 1e6:8.3f} -> {(en - t0) / 1e6:8.3f} ms  ({(en - st) / 1e6:7.3f})  q{q}  {n[:60]}")
