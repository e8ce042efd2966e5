"""Which channels make the tail of the sequential kernel: per-channel tick counters (profiling build) by channel parity and index range."""
import sys, os, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'm17-cxx-demod_amd')); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import m17hip, oracle_lib as ol
C, T = 4096, 480000
p = ol.gen_params(seed=20260101, kind=-1, n_frames=T // 1920 - 6, lead_in=3072, noise_sigma=600., tail_sigma=600., lead_sigma=40000.0, total=T)
ctx = m17hip.Context(C, T); ctx.synth(p, C, T)
ctx.tune(1, 1); ctx.reset(); ctx.run(); d = ctx.diag()
dc = ctx.debug_counters(C).astype(np.float64) / 1e5   # ms
names = {0: 'total', 1: 'chunks', 2: 'scalar', 3: 'decode', 12: 'window', 13: 'symbols', 14: 'iir', 15: 'search'}
for par in (0, 1):
    sel = dc[par::2]
    print('parity', par, ' '.join('%s %.2f/%.2f' % (n, np.median(sel[:, k]), sel[:, k].max()) for k, n in names.items()), ' n_chunks %.0f flips %.0f' % (np.median(sel[:, 4]), np.median(sel[:, 7].astype(np.int64) & 0xFFFFFFFF)))
for lo in range(0, C, 512):
    sel = dc[lo:lo + 512]
    print('channels %4d..%4d: total median %.1f p90 %.1f max %.1f | chunks median %.1f max %.1f' % (lo, lo + 511, np.median(sel[:, 0]), np.percentile(sel[:, 0], 90), sel[:, 0].max(), np.median(sel[:, 1]), sel[:, 1].max()))
slow = np.argsort(-dc[:, 0])[:40]
print('slowest 40 channel ids:', sorted(int(i) for i in slow))
print('their workgroups (c // 4):', sorted(set(int(i) // 4 for i in slow)))
