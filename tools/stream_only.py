"""The single-stream regime alone (two groups of 2048 channels, state carried, m17hip_demod_front), with N real placeholder HIP streams created
in front of every context (argv[1], default 0) — for rocprofv3 traces of the good and the bad stream layouts (NOTES 5.3):
    rocprofv3 --kernel-trace -d out -o t -- python3 tools/stream_only.py 3;  python3 tools/rocpd_step.py out/t_results.db"""
import ctypes, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
import torch
sys.path.insert(0, os.path.join(ROOT, 'm17-cxx-demod_amd')); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import m17hip, oracle_lib as ol
m17hip.Context._warned = True
nph = int(sys.argv[1]) if len(sys.argv) > 1 else 0
ORDER = os.environ.get('ORDER', 'new')
pre = int(sys.argv[2]) if len(sys.argv) > 2 else 0      # streams created before anything else (shifts every later index)
torch.zeros(1, device='cuda')
hip = ctypes.CDLL([l.split()[-1] for l in open('/proc/self/maps') if 'libamdhip64' in l][0])
keep = []
def hip_stream():
    h = ctypes.c_void_p(); assert hip.hipStreamCreateWithFlags(ctypes.byref(h), 1) == 0; keep.append(h)
for _ in range(pre): hip_stream()
C, T, G = 4096, 480000, 2
Cg = C // G
p = ol.gen_params(seed=20260101, kind=-1, n_frames=T // 1920 - 6, lead_in=3072, noise_sigma=600., tail_sigma=600., lead_sigma=40000.0, total=T)
gs, ss = [], []
for g in range(G):
    for _ in range(nph): hip_stream()
    c = m17hip.Context(Cg, T)
    if os.environ.get('MAIN') == 'torch': ss.append(torch.cuda.Stream()); c.set_stream(ss[-1].cuda_stream)   # (rounds 2-5: a host stream per context; default now: the library's own)
    c.synth(p, Cg, T, chan0=g * Cg); c.tune(16, 1); c.synth(p, Cg, T, chan0=g * Cg); c.tune(16, 0)
    c.reset(); c.run(); gs.append(c)
def stream(n):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for k in range(n):
        for c in gs: c.input_alternate(Cg, T); c.front()
        if ORDER == 'new':     # run k + 1's chain queued before run k's records are collected (m17hip_frames_select)
            for c in gs: c.run()
            for c in gs: c.frames_select(1); c.frames_count(); c.frames_select(0)
        else:
            for c in gs: c.frames_count(); c.run()
    for c in gs: c.frames_count()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) * 1e3 / n
stream(8)
if os.environ.get('TIMING'):
    for c in gs: c.timing(True); c.timing_reset()
if os.environ.get('COMPACT'):   # records compacted into a device buffer, as bench.py's leg does
    bufs = [torch.zeros(Cg * (2 * (T // 1920 + 2) + 4) * 64, dtype=torch.uint8, device='cuda') for _ in gs]
    for c, b in zip(gs, bufs): c.frames_count = (lambda c=c, b=b: c.frames_compact_device(b.data_ptr(), Cg * (2 * (T // 1920 + 2) + 4)))
print('placeholders %d pre %d order %s: single-stream ms/step %.2f %.2f' % (nph, pre, ORDER, stream(12), stream(12)), flush=True)
