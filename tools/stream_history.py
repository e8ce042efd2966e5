"""What a process's HISTORY does to the stream layout (VERDICT r5 #7b): N cycles of create / use / destroy in one process — each cycle the two-batch
regime (two contexts of 4096 channels) and then the continued-stream regime (two groups of 2048) — with F foreign HIP streams created in front of every
cycle.  MAIN=own (default): the library's own main stream; MAIN=torch: the host hands a torch stream to every context (m17hip_set_stream) — rounds 2-5, and
still the slow layouts (23.2 / 32.8 / 22.6 ms over three cycles WITH the role streams parked).  M17HIP_STREAM_SETS=0: per-context role streams as well.
    python3 tools/stream_history.py 3 2;  MAIN=torch M17HIP_STREAM_SETS=0 python3 tools/stream_history.py 3 2;  TIMING=1: what the per-kernel events cost"""
import ctypes, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
import torch
sys.path.insert(0, os.path.join(ROOT, 'm17-cxx-demod_amd')); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import m17hip, oracle_lib as ol
m17hip.Context._warned = True
cycles = int(sys.argv[1]) if len(sys.argv) > 1 else 3
foreign = int(sys.argv[2]) if len(sys.argv) > 2 else 0
MAIN = os.environ.get('MAIN', 'own')
torch.zeros(1, device='cuda')
hip = ctypes.CDLL([l.split()[-1] for l in open('/proc/self/maps') if 'libamdhip64' in l][0])
keep = []
def hip_stream():
    h = ctypes.c_void_p(); assert hip.hipStreamCreateWithFlags(ctypes.byref(h), 1) == 0; keep.append(h)
C, T = 4096, 480000
p = ol.gen_params(seed=20260101, kind=-1, n_frames=T // 1920 - 6, lead_in=3072, noise_sigma=600., tail_sigma=600., lead_sigma=40000.0, total=T)
def context(Cn):
    c = m17hip.Context(Cn, T)
    if MAIN == 'torch':
        st = torch.cuda.Stream(); keep.append(st); c.set_stream(st.cuda_stream)
    return c
for cyc in range(cycles):
    for _ in range(foreign): hip_stream()
    ctxs = [context(C) for _ in range(2)]
    for c in ctxs: c.synth(p, C, T)
    def groups(n):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for k0 in range(0, n, 2):
            for c in ctxs: c.reset(); c.run()
            for c in ctxs: c.frames_count()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) * 1e3 / n
    groups(16); two = groups(24)
    if os.environ.get('TIMING'):   # what the library's per-kernel events cost this regime (bench.py's headline leg keeps them inside its timed region)
        for c in ctxs: c.timing(True); c.timing_reset()
        two_t = groups(24)
        for c in ctxs: c.timing(False)
        print('  two-batch with kernel timers %.2f, without again %.2f' % (two_t, groups(24)), flush=True)
    for c in ctxs: c.close()
    Cg = C // 2
    gs = []
    for g in range(2):
        c = context(Cg)
        c.synth(p, Cg, T, chan0=g * Cg); c.tune(16, 1); c.synth(p, Cg, T, chan0=g * Cg); c.tune(16, 0)
        c.reset(); c.run(); gs.append(c)
    def stream(n):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for k in range(n):
            for c in gs: c.input_alternate(Cg, T); c.front()
            for c in gs: c.run()
            for c in gs: c.frames_select(1); c.frames_count(); c.frames_select(0)
        for c in gs: c.frames_count()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) * 1e3 / n
    stream(8); one = (stream(12), stream(12))
    for c in gs: c.close()
    print('sets %s main %s foreign %d cycle %d: two-batch %.2f ms/step   single-stream %.2f %.2f' % (os.environ.get('M17HIP_STREAM_SETS', '1'), MAIN, foreign, cyc, two, *one), flush=True)
