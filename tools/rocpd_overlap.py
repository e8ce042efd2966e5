#!/usr/bin/env python3
"""Where a steady-state step goes, from a rocprofv3 rocpd SQLite result of `bench.py` with batches in flight: over a window of
the timed region (between two fir_rrc150 launches `--from` and `--to` batches apart) the time during which nothing ran, during
which exactly one kernel class ran (per class), and the union time of every class.
Usage: rocpd_overlap.py results.db [--skip-ms X] [--span-ms Y] [--timeline]"""
import argparse
import sqlite3
from collections import defaultdict

CLASSES = [("fir_rrc150", "K1"), ("dcd_", "K3"), ("limit_track", "K2"), ("demod_wave", "K5"), ("mod_", "synth")]


def cls(name):
    for key, c in CLASSES:
        if key in name:
            return c
    return "other"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("db")
    ap.add_argument("--skip-ms", type=float, default=None, help="window start, ms after the last synth kernel (default: 40%% into the rest)")
    ap.add_argument("--span-ms", type=float, default=None, help="window length (default: to 90%% of the rest)")
    ap.add_argument("--timeline", action="store_true")
    a = ap.parse_args()
    db = sqlite3.connect(a.db)
    tabs = [r[0] for r in db.execute("select name from sqlite_master where type='table'")]
    t = lambda key: [x for x in tabs if x.startswith(key)][0]
    kd, ks = t("rocpd_kernel_dispatch"), t("rocpd_info_kernel_symbol")
    names = {r[0]: r[1] for r in db.execute(f"select id, kernel_name from {ks}")}
    rows = sorted(db.execute(f"select start, end, kernel_id, queue_id from {kd}").fetchall())
    synth_end = max((en for st, en, kid, q in rows if cls(names[kid]) == "synth"), default=rows[0][0])
    rows = [r for r in rows if r[0] >= synth_end]
    t_first, t_last = rows[0][0], max(r[1] for r in rows)
    span = t_last - t_first
    w0 = t_first + (int(a.skip_ms * 1e6) if a.skip_ms is not None else int(0.4 * span))
    w1 = w0 + int(a.span_ms * 1e6) if a.span_ms is not None else t_first + int(0.9 * span)
    ev = []
    for st, en, kid, q in rows:
        c = cls(names[kid])
        s, e = max(st, w0), min(en, w1)
        if s < e:
            ev.append((s, 1, c))
            ev.append((e, -1, c))
            if a.timeline:
                print(f"{(st - w0) / 1e6:9.3f} -> {(en - w0) / 1e6:9.3f} ms ({(en - st) / 1e6:7.3f})  q{q:<3} {c}")
    ev.sort()
    live = defaultdict(int)
    excl = defaultdict(int)
    union = defaultdict(int)
    combos = defaultdict(int)
    prev = w0
    for ts, d, c in ev:
        dt = ts - prev
        if dt > 0:
            on = tuple(sorted(k for k, v in live.items() if v > 0))
            combos[on] += dt
            for k in on:
                union[k] += dt
            if len(on) == 1:
                excl[on[0]] += dt
        live[c] += d
        prev = ts
    combos[()] += w1 - prev
    W = (w1 - w0) / 1e6
    print(f"window {W:.3f} ms (starts {(w0 - t_first) / 1e6:.1f} ms after the synthetic input was made)")
    print(f"idle (no kernel running): {combos[()] / 1e6:.3f} ms = {combos[()] / (w1 - w0):.1%}")
    for c in sorted(union):
        print(f"  {c:6s} running {union[c] / 1e6:8.3f} ms ({union[c] / (w1 - w0):.1%}), alone {excl[c] / 1e6:8.3f} ms ({excl[c] / (w1 - w0):.1%})")
    print("combinations:")
    for on, dt in sorted(combos.items(), key=lambda kv: -kv[1])[:12]:
        print(f"  {'+'.join(on) or 'idle':24s} {dt / 1e6:8.3f} ms ({dt / (w1 - w0):.1%})")


if __name__ == "__main__":
    main()
