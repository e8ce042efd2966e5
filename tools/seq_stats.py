"""Diagnostic: per-wave iteration statistics and kernel times of the sequential kernel (GPU box)."""
import sys, os, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'm17-cxx-demod_amd')); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import m17hip, oracle_lib as ol
C = int(sys.argv[1]) if len(sys.argv) > 1 else 256
for T in [int(v) for v in sys.argv[2:]] or [480000]:
    p = ol.gen_params(seed=20260101, kind=-1, n_frames=T//1920-6, lead_in=3072, noise_sigma=600., tail_sigma=600., lead_sigma=40000.0, total=T)
    x = ol.generate_batch(p, C, T, threads=64)
    ctx = m17hip.Context(C, T); ctx.upload(x); ctx.reset(); ctx.timing(True); ctx.run(); d = ctx.diag()
    ms = {k: ctx.timing_get(k)[0] for k in ('fir_rrc150', 'dcd', 'demod_seq')}
    it = d['pad'][:, 0] & 0xFFFFF; nb = d['pad'][:, 0] >> 20; slow = d['pad'][:, 1] & 0xFFFFF; flip = d['pad'][:, 1] >> 20
    print(f"C={C} T={T} ms={ms} ns/sample(seq)={ms['demod_seq']*1e6/T:.1f} iters/wave~{int(np.median(it))} us/iter={ms['demod_seq']*1e3/np.max(it):.2f} "
          f"slow med={int(np.median(slow))} batches~{int(np.median(nb))} frames={int(d['n_frames'].sum())}")
    ctx.close()
