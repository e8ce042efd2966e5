#!/usr/bin/env python3
"""Which hardware queue and stream every role of a context ran on, per context, from a rocprofv3 --kernel-trace result (rocpd):
    rocprofv3 --kernel-trace -d out -o t -- python3 tools/stream_history.py 2 1;  python3 tools/queue_map.py out/t_results.db
One line per (stream, queue) with the kernels seen there and the time span: the layout of the library's stream sets across create / use / destroy cycles."""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
tabs = [r[0] for r in db.execute("select name from sqlite_master where type='table'")]
t = lambda key: [x for x in tabs if x.startswith(key)][0]
kd, ks = t("rocpd_kernel_dispatch"), t("rocpd_info_kernel_symbol")
cols = [r[1] for r in db.execute(f"pragma table_info({kd})")]
names = {r[0]: r[1] for r in db.execute(f"select id, kernel_name from {ks}")}
sid = "stream_id" if "stream_id" in cols else "0"
rows = db.execute(f"select start, end, kernel_id, queue_id, {sid} from {kd}").fetchall()
t0 = min(r[0] for r in rows)
role = lambda n: ("K1 matched filter" if "fir_rrc150" in n else "K3 carrier detect" if "dcd_" in n else "K2 limit replay" if "limit_track" in n else
                  "K5 sequential (main)" if "demod_wave" in n else "payload (decode / compact)" if ("decode_deferred" in n or "compact_kernel" in n) else None)
seen = {}
for st, en, kid, q, s in rows:
    r = role(names[kid])
    if not r: continue
    e = seen.setdefault((s, q), {"roles": {}, "first": st, "last": en})
    e["roles"][r] = e["roles"].get(r, 0) + 1
    e["first"] = min(e["first"], st); e["last"] = max(e["last"], en)
print("| stream | hardware queue | roles (launches) | active from - to (s) |")
print("|---|---|---|---|")
for (s, q), e in sorted(seen.items(), key=lambda kv: kv[1]["first"]):
    print(f"| {s} | {q} | " + ", ".join(f"{k} ({v})" for k, v in sorted(e["roles"].items())) + f" | {(e['first'] - t0) / 1e9:.2f} - {(e['last'] - t0) / 1e9:.2f} |")
