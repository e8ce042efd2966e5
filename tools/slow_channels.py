#!/usr/bin/env python3
"""PROF build of K5 (m17hip_tune key 1: the run is ONE launch): where do the slowest channels of the bench workload spend their time?
python tools/slow_channels.py [samples=96000] [channels=4096]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "m17-cxx-demod_amd"), os.path.join(ROOT, "tests"), os.path.join(ROOT, "tools")):
    sys.path.insert(0, p)
import _toolslib  # noqa: F401  (the measurement build of the library)
import numpy as np
import torch
import m17hip, oracle_lib as ol
T = int(sys.argv[1]) if len(sys.argv) > 1 else 96000
C = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
TT = 480000
p = ol.gen_params(seed=20260101, kind=-1, n_frames=TT // 1920 - 6, lead_in=3072, noise_sigma=600.0, tail_sigma=600.0, lead_sigma=40000.0, total=TT)
ctx = m17hip.Context(C, TT)
ctx.synth(p, C, TT)
ctx.tune(1, 1)
ctx.reset(); ctx.run(samples=T); ctx.frames_count()
ctx.reset(); ctx.run(samples=T); ctx.frames_count()
d = ctx.debug_counters(C).astype(np.float64)
names = ["total", "bulk", "single", "decode", "#bulk", "#single", "bulk_samples", "flips|dec", "patch", "st0", "st1", "st2", "ens", "sym", "iir", "search", "off", "n_despec"]
tot = d[:, 0] / 1e5
order = np.argsort(-tot)
def row(c):
    t = d[c]
    return (f"ch {c:5d}: total {t[0]/1e5:6.2f} ms | bulk {t[1]/1e5:5.2f} (ens {t[12]/1e5:4.2f} sym {t[13]/1e5:4.2f} iir {t[14]/1e5:4.2f}) search {t[15]/1e5:5.2f} single {t[2]/1e5:5.2f} "
            f"decode {t[3]/1e5:5.2f} patch {t[8]/1e5:4.2f} off {t[16]/1e5:4.2f} | #bulk {int(t[4]):5d} #single {int(t[5]):5d} bulk_samples {int(t[6]):6d} #dec {int(t[7]) >> 32:3d} drops {int(t[17])}")
print(f"{T} samples after a reset, one launch, PROF build; median total {np.median(tot):.2f} ms, p99 {np.percentile(tot, 99):.2f}, max {tot.max():.2f}")
print("slowest:")
for c in order[:8]:
    print(row(int(c)))
print("around the median:")
for c in order[C // 2: C // 2 + 3]:
    print(row(int(c)))
    t = d[int(c)]
    print("      iterations by kind: none %d init %d quiet %d feed %d frame %d search %d syncwin %d | frame chunks cut by a clock move %d, ended by a pending clock flag %d, by the DCD point %d, by the 480-sample chunk size %d"
          % tuple(int(t[24 + k]) for k in (0, 1, 2, 3, 4, 5, 6, 8, 9, 10, 11)))
    print("      chunk selection %.2f ms, update points at the tail %.2f ms; single-sample steps by state: UNLOCKED %d LSF_SYNC %d STREAM_SYNC %d PACKET_SYNC %d BERT_SYNC %d SYNC_WAIT/FRAME %d"
          % ((t[36] / 1e5, t[37] / 1e5) + tuple(int(t[18 + q]) for q in range(6))))
dr = d[:, 17] > 0
print(f"dropped {int(dr.sum())} channels: mean total {tot[dr].mean():.2f} ms vs {tot[~dr].mean():.2f} ms; mean iir {d[dr, 14].mean()/1e5:.2f}, search {d[dr, 15].mean()/1e5:.2f}, single {d[dr, 2].mean()/1e5:.2f}, "
      f"bulk {d[dr, 1].mean()/1e5:.2f} (not dropped: iir {d[~dr, 14].mean()/1e5:.2f} search {d[~dr, 15].mean()/1e5:.2f} single {d[~dr, 2].mean()/1e5:.2f} bulk {d[~dr, 1].mean()/1e5:.2f})")
