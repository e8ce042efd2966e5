"""Which stream placements let two batches overlap?  Two contexts, steps overlapped as bench.py does; after each measurement one
of context B's streams is replaced (a new stream lands on the next hardware queue).  queue_probe.py [C] [T] [trials]"""
import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", sys.argv[4] if len(sys.argv) > 4 else "16")
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'm17-cxx-demod_amd')); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import m17hip, oracle_lib as ol
if os.environ.get('M17HIP_LIB'): m17hip.LIB_PATH = os.environ['M17HIP_LIB']   # experiment builds
C = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
T = int(sys.argv[2]) if len(sys.argv) > 2 else 480000
trials = int(sys.argv[3]) if len(sys.argv) > 3 else 10
p = ol.gen_params(seed=20260101, kind=-1, n_frames=T // 1920 - 6, lead_in=3072, noise_sigma=600., tail_sigma=600., lead_sigma=40000.0, total=T)
ctxs, streams = [], []
for f in range(2):
    c = m17hip.Context(C, T)
    streams.append(torch.cuda.Stream())
    c.set_stream(streams[-1].cuda_stream)
    c.synth(p, C, T)
    for kv in os.environ.get('M17_TUNE', '').split(','):
        if kv: c.tune(int(kv.split('=')[0]), int(kv.split('=')[1]))
    ctxs.append(c)

def steps(n):   # bench.py's overlapped loop: step k + 1 is queued before step k is waited for
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ctxs[0].reset(); ctxs[0].run()
    for k in range(1, n):
        c = ctxs[k % 2]
        c.reset(); c.run()
        ctxs[(k - 1) % 2].frames_count()
    ctxs[(n - 1) % 2].frames_count()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3

steps(2)
print("initial placement:        %.2f ms/step" % steps(10), flush=True)
for i in range(trials):
    if i % 2 == 0:
        ctxs[1].tune(11, 0)            # new side / side2 streams (K3, K1) for context B
        what = "B: new front-end streams"
    else:
        streams[1] = torch.cuda.Stream()
        ctxs[1].set_stream(streams[1].cuda_stream)
        what = "B: new main stream      "
    steps(2)
    print("%s  %.2f ms/step" % (what, steps(6)), flush=True)
