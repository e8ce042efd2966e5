"""Upper bound of pipelining consecutive steps: N full-size contexts (4096 channels each, own streams), one step each in flight at
the same time, against one context doing the steps one after the other.  GPU box; run with GPU_MAX_HW_QUEUES=16.
Usage: overlap_bench.py <contexts in flight,...>"""
import sys, os, time, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'm17-cxx-demod_amd')); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import m17hip, oracle_lib as ol, torch
C, T = 4096, 480000
p = ol.gen_params(seed=20260101, kind=-1, n_frames=T // 1920 - 6, lead_in=3072, noise_sigma=600., tail_sigma=600., lead_sigma=40000.0, total=T)
for G in [int(v) for v in sys.argv[1].split(',')]:
    ctxs, streams = [], [torch.cuda.Stream() for _ in range(G)]
    for g in range(G):
        c = m17hip.Context(C, T); c.synth(p, C, T)
        if G > 1: c.set_stream(streams[g].cuda_stream)
        ctxs.append(c)
    bufs = [torch.zeros(C * (2 * (T // 1920 + 2) + 4) * 64, dtype=torch.uint8, device='cuda') for _ in range(G)]
    def round_():
        for c in ctxs: c.reset(); c.run()
        return [c.frames_compact_device(b.data_ptr(), b.numel() // 64) for c, b in zip(ctxs, bufs)]
    for _ in range(2): n = round_()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    R = 4
    for _ in range(R): n = round_()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / (R * G)
    print(f'{G} step(s) in flight: {dt * 1e3:.2f} ms per step = {C * T / dt / 1e6:.0f} Msamples/s, frames {n}', flush=True)
    for c in ctxs: c.close()
    del ctxs, bufs
