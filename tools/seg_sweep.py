"""Wall time of one step (reset + run + compact) for several segment lengths / knobs (GPU box)."""
import sys, os, time, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'm17-cxx-demod_amd')); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import m17hip, oracle_lib as ol, torch
C, T = int(sys.argv[1]), int(sys.argv[2])
p = ol.gen_params(seed=20260101, kind=-1, n_frames=T//1920-6, lead_in=3072, noise_sigma=600., tail_sigma=600., lead_sigma=40000.0, total=T)
x = ol.generate_batch(p, C, T, threads=64)
ctx = m17hip.Context(C, T); ctx.upload(x)
buf = torch.zeros(C * (2 * (T // 1920 + 2) + 4) * 64, dtype=torch.uint8, device='cuda')
def step():
    ctx.reset(); ctx.run(); return ctx.frames_compact_device(buf.data_ptr(), buf.numel() // 64)
seg0s = [int(v) for v in sys.argv[4].split(',')] if len(sys.argv) > 4 else [11520]
if len(sys.argv) > 5: ctx.tune(5, int(sys.argv[5]))
if len(sys.argv) > 6: ctx.tune(0, int(sys.argv[6]))
for spec, seg, seg0 in [(1, int(v), z) for v in sys.argv[3].split(',') for z in seg0s] + [(0, 0, 0)]:
    ctx.tune(2, spec); ctx.tune(3, seg); ctx.tune(4, seg0)
    step(); torch.cuda.synchronize()
    ctx.timing(True); ctx.timing_reset()
    t0 = time.perf_counter()
    for _ in range(3): n = step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 3 * 1e3
    k = {name: ctx.timing_get(name)[0] / 3 for name in ('fir_rrc150', 'dcd', 'limit_track', 'demod_seq')}
    ctx.timing(False)
    print(f'limit_ahead={spec} seg={seg} seg0={seg0}: {dt:.2f} ms/step  frames={n}  kernel ms/step: ' + ' '.join(f'{a}={b:.1f}' for a, b in k.items()), flush=True)
