"""When do channels of the bench workload drop the limit-filter speculation?  The profiling build processes a run as ONE segment
and counts the first drop of every channel, so a run cut after 48000 k samples gives the channels that dropped somewhere in the
first k segments (cumulative); the differences between consecutive lines are the drops per segment.  drop_by_segment.py [C]"""
import sys, os, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'm17-cxx-demod_amd')); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch
import m17hip, oracle_lib as ol
C = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
T = 480000
p = ol.gen_params(seed=20260101, kind=-1, n_frames=T // 1920 - 6, lead_in=3072, noise_sigma=600., tail_sigma=600., lead_sigma=40000.0, total=T)
ctx = m17hip.Context(C, T)
ctx.synth(p, C, T)
ctx.tune(1, 1)
for k in range(1, 11):
    ctx.reset(); ctx.run(samples=48000 * k)
    d = ctx.debug_counters(C)
    n = d[:, 17]
    wg = (n.reshape(-1, 16) > 0).any(axis=1)
    print('by the end of segment %d: channels that dropped %4d (%.1f %%), K2 workgroups (16 channels) with one %3d of %d' % (k - 1, int((n > 0).sum()), 100.0 * (n > 0).mean(), int(wg.sum()), len(wg)), flush=True)
