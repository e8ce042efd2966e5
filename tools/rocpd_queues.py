#!/usr/bin/env python3
"""The last N dispatches of a rocprofv3 rocpd result in start order, with the hardware queue and the stream each ran on:
start / end (ms from the first of them), duration, queue id, stream id, kernel, grid.  Usage: rocpd_queues.py results.db [N]"""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
n = int(sys.argv[2]) if len(sys.argv) > 2 else 200
tabs = [r[0] for r in db.execute("select name from sqlite_master where type='table'")]
t = lambda key: [x for x in tabs if x.startswith(key)][0]
kd, ks = t("rocpd_kernel_dispatch"), t("rocpd_info_kernel_symbol")
cols = [r[1] for r in db.execute(f"pragma table_info({kd})")]
names = {r[0]: r[1] for r in db.execute(f"select id, kernel_name from {ks}")}
sid = "stream_id" if "stream_id" in cols else "0"
rows = sorted(db.execute(f"select start, end, kernel_id, queue_id, {sid}, grid_size_x, grid_size_y from {kd}").fetchall())[-n:]
t0 = rows[0][0]
for st, en, kid, q, s, gx, gy in rows:
    nm = names[kid].split('(')[0].replace('m17::', '').replace('(anonymous namespace)::', '')
    print(f"{(st - t0) / 1e6:9.3f} {(en - t0) / 1e6:9.3f} {(en - st) / 1e6:7.3f}  q{q:<3} s{s:<3} {nm[:44]:44s} {gx}x{gy}")
