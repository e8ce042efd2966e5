"""K3 pipeline ablation (GPU box): time of the four-wave kernel with single roles switched off (results are wrong then; timing only)."""
import sys, os, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'm17-cxx-demod_amd')); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import m17hip, oracle_lib as ol
C, T = int(sys.argv[1]), int(sys.argv[2])
p = ol.gen_params(seed=20260101, kind=-1, n_frames=T // 1920 - 6, lead_in=3072, noise_sigma=600., tail_sigma=600., lead_sigma=40000.0, total=T)
ctx = m17hip.Context(C, T); ctx.synth(p, C, T); ctx.timing(True); ctx.tune(10, 1); ctx.tune(1, 1)   # (the role switches are accepted only with the diagnostics knob on)
for name, fl in (('all roles', 0), ('no P', 16), ('no R', 32), ('no A0', 64), ('no A1', 128), ('only R', 16 | 64 | 128), ('only P', 32 | 64 | 128), ('only A0', 16 | 32 | 128), ('none (barriers only)', 16 | 32 | 64 | 128)):
    ts = []
    for rep in range(3):
        ctx.timing_reset(); ctx.dcd(flags=fl, fetch=False); ts.append(ctx.timing_get('dcd')[0])
    print(f'{name:22s}: {min(ts):.3f} ms ({min(ts) * 1e6 / T:.2f} ns/sample)', flush=True)
