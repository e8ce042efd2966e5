"""K1 alone: one launch over the whole run vs ten launches of a tenth each (per-launch tail / ramp effects)."""
import sys, os, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'm17-cxx-demod_amd')); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import m17hip, oracle_lib as ol
C, T = 4096, 480000
p = ol.gen_params(seed=20260101, kind=-1, n_frames=T // 1920 - 6, lead_in=3072, noise_sigma=600., tail_sigma=600., lead_sigma=40000.0, total=T)
ctx = m17hip.Context(C, T)
ctx.synth(p, C, T)
def timed(f, n=5):
    ts = []
    for _ in range(n):
        torch.cuda.synchronize(); t = time.perf_counter(); f(); torch.cuda.synchronize(); ts.append((time.perf_counter() - t) * 1e3)
    return min(ts)
ctx.T = T
print('fir whole run      %.2f ms' % timed(lambda: ctx.fir(fetch=False)))
for seg in (48000, 46080, 96000):
    ctx.T = seg
    n = T // seg
    print('fir %d x %d      %.2f ms' % (n, seg, timed(lambda: [ctx.fir(fetch=False) for _ in range(n)])))
ctx.T = T
print('dcd whole run      %.2f ms' % timed(lambda: ctx.dcd(fetch=False)))
ctx.T = 48000
print('dcd 10 x 48000     %.2f ms' % timed(lambda: [ctx.dcd(fetch=False) for _ in range(10)]))
