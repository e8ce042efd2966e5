#!/usr/bin/env python3
"""PROF build of K5 over the FIRST segments of a run in the stream regime (state carried from two full runs before): section breakdown of the
slowest channels.  python tools/slow_channels_stream.py [samples=48000]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "m17-cxx-demod_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np
import torch
import m17hip, oracle_lib as ol
T = int(sys.argv[1]) if len(sys.argv) > 1 else 48000
C, TT = 4096, 480000
p = ol.gen_params(seed=20260101, kind=-1, n_frames=TT // 1920 - 6, lead_in=3072, noise_sigma=600.0, tail_sigma=600.0, lead_sigma=40000.0, total=TT)
ctx = m17hip.Context(C, TT)
ctx.synth(p, C, TT)
ctx.reset(); ctx.run(); ctx.run(); ctx.frames_count()
ctx.tune(1, 1)
ctx.run(samples=T); ctx.frames_count()
d = ctx.debug_counters(C).astype(np.float64)
tot = d[:, 0] / 1e5
order = np.argsort(-tot)
def row(c):
    t = d[c]
    return (f"ch {c:5d}: total {t[0]/1e5:6.2f} ms | bulk {t[1]/1e5:5.2f} (ens {t[12]/1e5:4.2f} sym {t[13]/1e5:4.2f} iir/serve {t[14]/1e5:4.2f}) search {t[15]/1e5:5.2f} single {t[2]/1e5:5.2f} "
            f"decode {t[3]/1e5:5.2f} patch {t[8]/1e5:4.2f} off {t[16]/1e5:4.2f} | #bulk {int(t[4]):5d} #single {int(t[5]):5d} bulk_samples {int(t[6]):6d} #dec {int(t[7]) >> 32:3d} drops {int(t[17])}"
            f" | singles by state L/S/P/B/W/F {[int(x) for x in t[19:24]]} | (experiment build) single-step parts: pre {t[18]/1e5:4.2f} lim {t[9]/1e5:4.2f} upd {t[10]/1e5:4.2f} clk {t[11]/1e5:4.2f}")
print(f"first {T} samples of a run in the stream regime, one launch, PROF build; median {np.median(tot):.2f} ms, p99 {np.percentile(tot, 99):.2f}, max {tot.max():.2f}")
for c in order[:6]:
    print(row(int(c)))
print("around the median:")
for c in order[C // 2: C // 2 + 2]:
    print(row(int(c)))
