#!/usr/bin/env python3
"""How long does each K5 wave (= channel) work on each segment of a run, and which ones make a launch last (a launch ends with its
slowest wave)?  m17hip_tune key 19.  One run of the bench workload alone on the chip, in the stream regime (state carried from the
runs before) or after a reset.   python tools/wave_times.py [--reset 1] [--channels 4096] [--tune k=v,...]"""
import argparse, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "m17-cxx-demod_amd"), os.path.join(ROOT, "tests"), os.path.join(ROOT, "tools")):
    sys.path.insert(0, p)
import _toolslib  # noqa: F401  (the measurement build of the library)
import numpy as np
import torch
import m17hip, oracle_lib as ol

ap = argparse.ArgumentParser()
ap.add_argument("--reset", type=int, default=0)
ap.add_argument("--channels", type=int, default=4096)
ap.add_argument("--samples", type=int, default=480000)
ap.add_argument("--tune", default="")
ap.add_argument("--top", type=int, default=6)
a = ap.parse_args()
C, T = a.channels, a.samples
p = ol.gen_params(seed=20260101, kind=-1, n_frames=max(1, T // 1920 - 6), lead_in=3072, noise_sigma=600.0, tail_sigma=600.0, lead_sigma=40000.0, total=T)
ctx = m17hip.Context(C, T)
for kv in filter(None, a.tune.split(",")):
    k, v = kv.split("=")
    ctx.tune(int(k), int(v))
ctx.synth(p, C, T)
ctx.tune(16, 1); ctx.synth(p, C, T); ctx.tune(16, 0)
ctx.reset(); ctx.run()
for _ in range(3):
    ctx.input_alternate(C, T); ctx.run()
torch.cuda.synchronize()
ctx.tune(19, 1)
if a.reset:
    ctx.reset()
ctx.input_alternate(C, T); ctx.run()
ctx.frames_count()
d = ctx.debug_counters(C)[:C]
nseg = (T + 47999) // 48000
print("segment:  median    p90     p99     max (ms)   dropped   slowest channels (ms, D = dropped the speculation)")
for k in range(nseg):
    t = (d[:, k] & ((1 << 62) - 1)).astype(np.float64) / 1e5
    dr = (d[:, k] >> np.uint64(62)) & np.uint64(1)
    order = np.argsort(-t)[: a.top]
    worst = " ".join(f"{int(c)}:{t[c]:.2f}{'D' if dr[c] else ''}" for c in order)
    print(f"{k:4d}    {np.median(t):7.3f} {np.percentile(t, 90):7.3f} {np.percentile(t, 99):7.3f} {t.max():7.3f}   {int(dr.sum()):6d}    {worst}")
tot = ((d[:, :nseg] & ((1 << 62) - 1)).astype(np.float64) / 1e5).sum(axis=1)
print(f"per channel over the run: median {np.median(tot):.2f} ms, p99 {np.percentile(tot, 99):.2f}, max {tot.max():.2f} (channel {int(tot.argmax())}); sum of the per-segment maxima "
      f"{sum(((d[:, k] & ((1 << 62) - 1)).astype(np.float64) / 1e5).max() for k in range(nseg)):.2f} ms")
dd = ((d[:, :nseg] >> np.uint64(62)) & np.uint64(1)).astype(int)
print("channels that dropped in n segments:", np.bincount(dd.sum(axis=1)))
print("odd (voice) share of the drops per segment:", [int(dd[1::2, k].sum()) for k in range(nseg)], "of", [int(dd[:, k].sum()) for k in range(nseg)])
