"""K3 A/B on the GPU box: the four-wave pipeline (tune 10 = 1) against the single-wave kernel (tune 10 = 0, default): time alone per
launch over the whole slab and per 48 000-sample segment, the tables compared bit for bit; then the chain with each.
Usage: k3_ab.py <channels> <samples>"""
import sys, os, time, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'm17-cxx-demod_amd')); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import m17hip, oracle_lib as ol, torch
C, T = int(sys.argv[1]), int(sys.argv[2])
p = ol.gen_params(seed=20260101, kind=-1, n_frames=T // 1920 - 6, lead_in=3072, noise_sigma=600., tail_sigma=600., lead_sigma=40000.0, total=T)
ctx = m17hip.Context(C, T); ctx.synth(p, C, T); ctx.timing(True)
tabs = {}
for single in (1, 0):
    ctx.tune(10, 0 if single else 1)
    ts = []
    for rep in range(3):
        ctx.timing_reset(); ctx.dcd(fetch=False); ts.append(ctx.timing_get('dcd')[0])
    tabs[single] = ctx.dcd()[: min(C, 512)].copy()
    print(f'K3 {"single wave" if single else "pipeline   "}: {min(ts):.3f} ms alone per {T} samples ({min(ts) * 1e6 / T:.2f} ns/sample)', flush=True)
print('tables identical:', tabs[0].tobytes() == tabs[1].tobytes(), flush=True)
buf = torch.zeros(C * (2 * (T // 1920 + 2) + 4) * 64, dtype=torch.uint8, device='cuda')
for single in (1, 0, 1, 0):
    ctx.tune(10, 0 if single else 1)
    for rep in range(2):
        ctx.reset(); ctx.run(); ctx.frames_compact_device(buf.data_ptr(), buf.numel() // 64)
    ctx.timing_reset(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for rep in range(5):
        ctx.reset(); ctx.run(); n = ctx.frames_compact_device(buf.data_ptr(), buf.numel() // 64)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
    print(f'chain with K3 {"single wave" if single else "pipeline   "}: {dt * 1e3:.2f} ms/step, frames {n}; per step: ' +
          ', '.join(f'{k} {ctx.timing_get(k)[0] / 5:.1f}' for k in ('fir_rrc150', 'dcd', 'limit_track', 'demod_seq')), flush=True)
