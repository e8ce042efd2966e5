"""Stream priorities of the front end (tune 11) x K3 form (tune 10): step time of the chain (GPU box)."""
import sys, os, time, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'm17-cxx-demod_amd')); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch, m17hip, oracle_lib as ol
C, T = 4096, 480000
p = ol.gen_params(seed=20260101, kind=-1, n_frames=T // 1920 - 6, lead_in=3072, noise_sigma=600., tail_sigma=600., lead_sigma=40000.0, total=T)
ctx = m17hip.Context(C, T); ctx.synth(p, C, T); ctx.timing(True)
buf = torch.zeros(C * (2 * (T // 1920 + 2) + 4) * 64, dtype=torch.uint8, device='cuda')
for k3, prio in [(1, 0), (1, 4), (0, 0), (0, 4), (1, 0), (1, 4), (0, 0), (0, 4), (1, 5), (0, 5), (1, 4), (0, 4)]:
    if True:
        ctx.tune(10, 0 if k3 else 1); ctx.tune(11, prio)
        for rep in range(2):
            ctx.reset(); ctx.run(); ctx.frames_compact_device(buf.data_ptr(), buf.numel() // 64)
        ctx.timing_reset(); torch.cuda.synchronize(); t0 = time.perf_counter()
        for rep in range(5):
            ctx.reset(); ctx.run(); n = ctx.frames_compact_device(buf.data_ptr(), buf.numel() // 64)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
        print(f'k3_single={k3} prio={prio}: {dt * 1e3:.2f} ms/step, frames {n}; ' + ', '.join(f'{k} {ctx.timing_get(k)[0] / 5:.1f}' for k in ('fir_rrc150', 'dcd', 'limit_track', 'demod_seq')), flush=True)
