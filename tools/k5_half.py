"""Does K5 need all four waves per SIMD?  Step time and K5 launch times for 4096 / 2048 / 1024 channels (rocprof-free: HIP-event sums)."""
import sys, os, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'm17-cxx-demod_amd')); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import m17hip, oracle_lib as ol
T = 480000
for C in (4096, 2048, 1024):
    p = ol.gen_params(seed=20260101, kind=-1, n_frames=T // 1920 - 6, lead_in=3072, noise_sigma=600., tail_sigma=600., lead_sigma=40000.0, total=T)
    ctx = m17hip.Context(C, T); ctx.synth(p, C, T)
    for rep in range(3):
        ctx.reset(); torch.cuda.synchronize(); t = time.perf_counter(); ctx.run(); torch.cuda.synchronize(); dt = (time.perf_counter() - t) * 1e3
    ctx.timing(True); ctx.timing_reset(); ctx.reset(); ctx.run(); torch.cuda.synchronize()
    print('C=%d step %.2f ms | event sums: %s' % (C, dt, ' '.join('%s %.2f/%d' % (k, *ctx.timing_get(k)) for k in ('fir_rrc150', 'dcd', 'limit_track', 'demod_seq'))), flush=True)
    ctx.close()
