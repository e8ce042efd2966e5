"""Front-end kernels timed alone (GPU box): K1 FIR and K3 DCD through the per-operator entry points, then the chain."""
import sys, os, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'm17-cxx-demod_amd')); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import m17hip, oracle_lib as ol
C, T = int(sys.argv[1]), int(sys.argv[2])
p = ol.gen_params(seed=20260101, kind=-1, n_frames=T//1920-6, lead_in=3072, noise_sigma=600., tail_sigma=600., lead_sigma=40000.0, total=T)
x = ol.generate_batch(p, C, T, threads=64)
ctx = m17hip.Context(C, T); ctx.upload(x)
ctx.timing(True)
for rep in range(3):
    ctx.timing_reset(); ctx.reset()
    ctx.dcd(fetch=False); ctx.fir(fetch=False)
    print('alone: fir %.2f dcd %.2f ms' % (ctx.timing_get('fir_rrc150')[0], ctx.timing_get('dcd')[0]), flush=True)
    ctx.timing_reset(); ctx.reset(); ctx.run()
    print('chain: fir %.2f dcd %.2f seq %.2f' % tuple(ctx.timing_get(k)[0] for k in ('fir_rrc150', 'dcd', 'demod_seq')), flush=True)
print('limit_track %.2f ms' % ctx.timing_get('limit_track')[0])
