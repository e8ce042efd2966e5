"""One configuration of tools/parity_sweep.py with the differences spelled out.  Usage: sweep_one.py <seed> <spec> <seg> <piped> [tune33] [lib]"""
import sys, os, numpy as np
os.environ.setdefault('GPU_MAX_HW_QUEUES', '16')
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'm17-cxx-demod_amd')); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import oracle_lib as ol
import torch, m17hip
seed, spec, seg, piped = (int(v) for v in sys.argv[1:5])
ramp = int(sys.argv[5]) if len(sys.argv) > 5 else 0
C, T = 64, 96000
ctx = m17hip.Context(C, T)
rng = np.random.default_rng(seed)
x = np.zeros((C, T), dtype=np.int16)
for c in range(C):
    pos = 0
    while pos < T - 8000:
        n = min(int(rng.integers(6000, 40000)), T - pos)
        p = ol.gen_params(seed=int(rng.integers(1, 1 << 30)), kind=int(rng.choice([0, 1, 2, 4])), n_frames=int(rng.integers(1, 16)),
                          lead_in=int(rng.integers(0, 5000)), lead_sigma=float(rng.choice([0.0, 100.0, 1000.0, 10000.0, 40000.0])),
                          noise_sigma=float(rng.choice([0.0, 100.0, 500.0, 1200.0, 2500.0])), tail_sigma=float(rng.choice([0.0, 100.0, 1000.0, 5000.0])),
                          dc_offset=float(rng.choice([0.0, 0.0, 300.0, -2000.0, 6000.0])), gain=float(rng.choice([1.0, 0.3, 0.7, 1.6])),
                          phase=int(rng.integers(-1, 10)), invert=0, total=n)
        x[c, pos:pos + n] = ol.generate(p)[:n]; pos += n
inv = seed & 1
recs, counts, diags = ol.demod_batch(x, invert=inv, cap=2 * (T // 1920 + 2) + 4, threads=os.cpu_count())
cuts = sorted(int(v) for v in rng.integers(1, T, size=2))
rseg = int(rng.integers(3000, 30000))
pieces = [0] + cuts + [T]
keys = {15: spec, 3: seg, 33: ramp, 20: (seed + spec + piped) & 1, 17: (seed + piped + (seg & 1)) & 1, 26: (seed + spec + (seg >> 2)) & 1}
for a in sys.argv[6:]:
    k, v = a.split('='); keys[int(k)] = int(v)
print('keys', keys, 'pieces', pieces)
for k, v in keys.items():
    try: ctx.tune(k, v)
    except Exception as e: print('tune', k, 'refused:', e)
ctx.reset()
spans = [(a, b) for a, b in zip(pieces[:-1], pieces[1:]) if b > a]
pins = [torch.from_numpy(np.ascontiguousarray(x[:, a:b])).pin_memory() for a, b in spans]
parts = []
if piped == 0:
    for a, b in spans:
        ctx.upload(x[:, a:b]); ctx.run(flags=inv); parts.append(ctx.frames().copy())
else:
    ctx.upload_async(pins[0].data_ptr(), C, spans[0][1] - spans[0][0]); ctx.run(flags=inv, channels=C, samples=spans[0][1] - spans[0][0])
    for i in range(len(spans)):
        if i + 1 < len(spans):
            n1 = spans[i + 1][1] - spans[i + 1][0]
            ctx.upload_async(pins[i + 1].data_ptr(), C, n1); ctx.front(flags=inv, channels=C, samples=n1)
            if piped == 2:
                ctx.run(flags=inv, channels=C, samples=n1); ctx.frames_select(1)
        parts.append(ctx.frames().copy())
        if hasattr(ctx.lib, 'm17hip_frames_select'): ctx.frames_select(0)
        if i + 1 < len(spans) and piped != 2:
            ctx.run(flags=inv, channels=C, samples=n1)
    ctx.upload_wait()
got = np.concatenate(parts); got = got[np.lexsort((got['seq'], got['channel']))]
d = ctx.diag()
nbad = 0
for c in range(C):
    g = got[got['channel'] == c]; e = recs[c, :counts[c]]
    rec_bad = g.tobytes() != e.tobytes()
    dbad = [f for f in d.dtype.names if f in diags.dtype.names and not np.array_equal(d[f][c:c + 1], diags[f][c:c + 1], equal_nan=True)]
    if rec_bad or dbad:
        nbad += 1
        msg = f'channel {c}: records {g.size} vs {e.size}'
        if rec_bad and g.size == e.size:
            idx = [i for i in range(g.size) if g[i].tobytes() != e[i].tobytes()]
            msg += f' differ at {idx[:6]}: ' + '; '.join(f"#{i} pos {int(g[i]['sample_pos'])} type {int(g[i]['frame_type'])} cost {int(g[i]['cost'])}/{int(e[i]['cost'])} payload_eq {g[i]['payload'].tobytes() == e[i]['payload'].tobytes()}" for i in idx[:4])
        msg += f' diag fields {[(f, d[f][c].item(), diags[f][c].item()) for f in dbad]}'
        print(msg)
print('bad channels', nbad)
