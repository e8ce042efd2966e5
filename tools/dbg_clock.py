"""Debugging aid (GPU box): the config5 sigma = 0 scenario of tests/test_gpu_configs.py on a few channels; finds the channels whose diag differs
from the oracle and bisects the first prefix length at which the live state differs (tools/dbg_bisect.py on generated input)."""
import os, sys, subprocess
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'm17-cxx-demod_amd')); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import oracle_lib as ol
import m17hip
Cn, T = 512, 96000
sigma = float(sys.argv[1]) if len(sys.argv) > 1 else 0.0
p = ol.gen_params(seed=777, kind=0, n_frames=T // 1920 + 2, lead_in=3072, lead_sigma=40000.0, noise_sigma=sigma, tail_sigma=max(sigma, 100.0), dc_offset=0.0, gain=1.0, total=T)
x = ol.generate_batch(p, Cn, T, threads=8)
ctx = m17hip.Context(Cn, T); ctx.upload(x); ctx.reset(); ctx.run(); d = ctx.diag(); got = ctx.frames()
recs, counts, diags = ol.demod_batch(x, cap=2 * (T // 1920 + 2) + 4, threads=8)
exp = np.concatenate([recs[c, :counts[c]] for c in range(Cn)])
print('records equal:', got.tobytes() == exp.tobytes())
bad = [c for c in range(Cn) if any(not np.array_equal(d[f][c:c + 1], diags[f][c:c + 1], equal_nan=True) for f in d.dtype.names if f in diags.dtype.names)]
print('channels with a differing diag:', bad)
if bad:
    c = bad[0]
    print({f: (d[f][c], diags[f][c]) for f in d.dtype.names if f in diags.dtype.names and not np.array_equal(d[f][c:c + 1], diags[f][c:c + 1], equal_nan=True)})
    np.save('/tmp/x_dbg.npy', x)
    ctx.close()
    subprocess.run([sys.executable, os.path.join(ROOT, 'tools', 'dbg_bisect.py'), '/tmp/x_dbg.npy', str(c), '2000', str(T)], env=dict(os.environ, SEG='0'))
