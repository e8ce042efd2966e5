"""Two batches in flight: the pipeline (step k + 1 queued before step k is waited for) started in step or half a step apart, and pairs
of steps launched and waited for together.  regime_bench.py (GPU box)"""
import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'm17-cxx-demod_amd')); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import m17hip, oracle_lib as ol
C, T = 4096, 480000
p = ol.gen_params(seed=20260101, kind=-1, n_frames=T // 1920 - 6, lead_in=3072, noise_sigma=600., tail_sigma=600., lead_sigma=40000.0, total=T)
ctxs, streams = [], []
for f in range(2):
    c = m17hip.Context(C, T); streams.append(torch.cuda.Stream()); c.set_stream(streams[-1].cuda_stream); c.synth(p, C, T); ctxs.append(c)
def pipeline(n, delay_ms):
    torch.cuda.synchronize()
    ts = []
    t0 = time.perf_counter()
    ctxs[0].reset(); ctxs[0].run()
    if delay_ms: time.sleep(delay_ms / 1e3)
    tp = time.perf_counter()
    for k in range(1, n):
        c = ctxs[k % 2]; c.reset(); c.run(); ctxs[(k - 1) % 2].frames_count()
        now = time.perf_counter(); ts.append((now - tp) * 1e3); tp = now
    ctxs[(n - 1) % 2].frames_count(); torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) * 1e3
    return dt / n, ts
for rep in range(2):
    for d in (0, 14):
        pipeline(6, d)
        ms, ts = pipeline(40, d)
        print('start delay %2d ms: %.2f ms/step   last waits: %s' % (d, ms, ' '.join('%.0f' % v for v in ts[-10:])), flush=True)
def groups(n):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for k0 in range(0, n, 2):
        for k in (k0, k0 + 1): ctxs[k % 2].reset(); ctxs[k % 2].run()
        for k in (k0, k0 + 1): ctxs[k % 2].frames_count()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) * 1e3 / n
groups(4)
print('pairs launched and waited together: %.2f ms/step' % groups(40))
print('pairs launched and waited together: %.2f ms/step' % groups(40))
