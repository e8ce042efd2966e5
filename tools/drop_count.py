"""How many channels of the bench workload drop the limit-filter speculation at least once in a run (profiling build, one segment)."""
import sys, os, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'm17-cxx-demod_amd')); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import m17hip, oracle_lib as ol
C, T = int(sys.argv[1]), int(sys.argv[2])
p = ol.gen_params(seed=20260101, kind=-1, n_frames=T // 1920 - 6, lead_in=3072, noise_sigma=600., tail_sigma=600., lead_sigma=40000.0, total=T)
ctx = m17hip.Context(C, T)
ctx.synth(p, C, T)
ctx.tune(1, 1)
ctx.reset(); ctx.run()
d = ctx.debug_counters(C)
n = d[:, 17]
print('channels', C, 'dropped at least once', int((n > 0).sum()), 'events', int(n.sum()), 'max per channel', int(n.max()))
