#!/usr/bin/env python3
"""Kernel timeline of the LAST step in a rocprofv3 rocpd SQLite result: start/end (ms, relative) of every dispatch after the
last seq_reset_kernel.  Usage: rocpd_timeline.py results.db"""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
tabs = [r[0] for r in db.execute("select name from sqlite_master where type='table'")]
t = lambda key: [x for x in tabs if x.startswith(key)][0]
kd, ks = t("rocpd_kernel_dispatch"), t("rocpd_info_kernel_symbol")
names = {r[0]: r[1] for r in db.execute(f"select id, kernel_name from {ks}")}
rows = sorted(db.execute(f"select start, end, kernel_id, queue_id from {kd}").fetchall())
last = max(i for i, r in enumerate(rows) if 'seq_reset' in names[r[2]])
t0 = rows[last][0]
for st, en, kid, q in rows[last:]:
    n = names[kid].split('(')[0].replace('m17::', '').replace('(anonymous namespace)::', '')
    print(f"{(st - t0) / 1e6:8.3f} -> {(en - t0) / 1e6:8.3f} ms  ({(en - st) / 1e6:7.3f})  q{q}  {n[:60]}")
