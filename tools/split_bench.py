"""Does the chain gain from running the channels as G independent groups side by side (each with its own streams and segment
chain), so that one group's latency gaps (K2 -> K5 hand-overs, K3 chain) are filled by the others?  GPU box.
Usage: split_bench.py <channels> <samples> <groups,...> [limit_ahead=1] [k3_single=0] [seg=48000]"""
import sys, os, time, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'm17-cxx-demod_amd')); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import m17hip, oracle_lib as ol, torch
C, T = int(sys.argv[1]), int(sys.argv[2])
spec = int(sys.argv[4]) if len(sys.argv) > 4 else 1
k3s = int(sys.argv[5]) if len(sys.argv) > 5 else 0
seg = int(sys.argv[6]) if len(sys.argv) > 6 else 48000
p = ol.gen_params(seed=20260101, kind=-1, n_frames=T // 1920 - 6, lead_in=3072, noise_sigma=600., tail_sigma=600., lead_sigma=40000.0, total=T)
for G in [int(v) for v in sys.argv[3].split(',')]:
    per = C // G
    ctxs = []
    streams = [torch.cuda.Stream() for _ in range(G)]   # own main stream per group (the default is the shared NULL stream)
    for g in range(G):
        c = m17hip.Context(per, T); c.set_channel_base(g * per); c.synth(p, per, T, chan0=g * per)
        if G > 1: c.set_stream(streams[g].cuda_stream)
        c.tune(2, spec); c.tune(10, 0 if k3s else 1); c.tune(3, seg)
        ctxs.append(c)
    bufs = [torch.zeros(per * (2 * (T // 1920 + 2) + 4) * 64, dtype=torch.uint8, device='cuda') for _ in range(G)]
    def step():
        for c in ctxs: c.reset()
        for c in ctxs: c.run()
        return sum(c.frames_compact_device(b.data_ptr(), b.numel() // 64) for c, b in zip(ctxs, bufs))
    for _ in range(2): n = step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5): n = step()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
    print(f'groups={G} x {per} channels (limit_ahead={spec}, k3_single={k3s}, seg={seg}): {dt * 1e3:.2f} ms/step = {C * T / dt / 1e6:.0f} Msamples/s, frames {n}', flush=True)
    for c in ctxs: c.close()
    del ctxs, bufs
