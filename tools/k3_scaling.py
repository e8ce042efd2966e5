"""K3 (DCD) launch time as a function of the channel count (GPU box)."""
import sys, os, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'm17-cxx-demod_amd'))
import m17hip
T = int(sys.argv[1])
rng = np.random.default_rng(1)
for C in [int(v) for v in sys.argv[2].split(',')]:
    x = rng.integers(-20000, 20000, size=(C, T), dtype=np.int16)
    ctx = m17hip.Context(C, T); ctx.upload(x); ctx.timing(True)
    ts = []
    for rep in range(3):
        ctx.timing_reset(); ctx.dcd(fetch=False); ts.append(ctx.timing_get('dcd')[0])
    ctx.timing_reset(); ctx.fir(fetch=False)
    print(f'C={C} T={T}: dcd {min(ts):.3f} ms ({min(ts)*1e6/T:.1f} ns/sample)  fir {ctx.timing_get("fir_rrc150")[0]:.3f} ms', flush=True)
    del ctx
