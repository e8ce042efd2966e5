"""Randomised shapes for BASELINE configs[1] (m17hip_fir_correlator: the relayed limit recurrence, four-sample correlations, pieces in time):
channel counts that do and do not fill a workgroup of sixteen, lengths that are multiples of 128 / of 4 / of neither, bursts, exact zeros
(subnormal decay of the limit filter), INVERT — the one call against the two operators and the oracle, bit for bit.
    python tools/config2_sweep.py [first seed] [seeds]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'm17-cxx-demod_amd')); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch  # noqa: F401
import m17hip, oracle_lib as ol

first, n = (int(sys.argv[1]) if len(sys.argv) > 1 else 1), (int(sys.argv[2]) if len(sys.argv) > 2 else 40)
bad = 0
for seed in range(first, first + n):
    rng = np.random.default_rng(seed)
    Cn = int(rng.choice([1, 2, 15, 16, 17, 31, 33, 48, 70]))
    kind = int(rng.integers(0, 4))
    T = int(rng.integers(1, 160)) * 128 if kind < 2 else (int(rng.integers(64, 5000)) * 4 if kind == 2 else int(rng.integers(300, 20000)))
    p = ol.gen_params(seed=seed, kind=-1, n_frames=max(1, T // 1920 - 2), lead_in=int(rng.integers(100, 900)), noise_sigma=float(rng.choice([0.0, 300.0, 3000.0])),
                      tail_sigma=800.0, lead_sigma=30000.0, total=T)
    x = ol.generate_batch(p, Cn, T, threads=8)
    if rng.random() < 0.5: x[0, min(int(rng.integers(50, 400)), T // 2):] = 0          # exact zeros behind a burst
    if rng.random() < 0.3: x[Cn - 1, :] = rng.integers(-32768, 32767, T, dtype=np.int64).astype(np.int16)   # full-scale noise
    ctx = m17hip.Context(Cn, T)
    ctx.upload(x)
    flags = m17hip.FLAG_INVERT if rng.random() < 0.3 else 0
    y, limit, corr = ctx.fir_correlator(flags=flags)
    y2 = ctx.fir(flags=flags)
    limit2, corr2 = ctx.correlator()
    ok = np.array_equal(y, y2) and np.array_equal(limit.view(np.uint32), limit2.view(np.uint32)) and np.array_equal(corr.view(np.uint32), corr2.view(np.uint32))
    for c in sorted({0, Cn // 2, Cn - 1}):
        ye = ol.fir_i16(x[c], invert=1 if flags else 0)
        le, ce = ol.correlator(ye)
        ok = ok and np.array_equal(y[c].view(np.uint32), ye.view(np.uint32)) and np.array_equal(limit[c].view(np.uint32), le.view(np.uint32)) and np.array_equal(corr[:, c, :].view(np.uint32), ce.view(np.uint32))
    ctx.close()
    bad += 0 if ok else 1
    print(f'seed {seed}: C {Cn} T {T} invert {int(bool(flags))}: {"ok" if ok else "DIFFERS"}', flush=True)
print('TOTAL shapes that differ:', bad)
