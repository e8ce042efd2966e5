"""Same-box A/B of library builds in the two-batch regime (bench.py's default: two independent batches launched and waited for together).
Every build runs in its own process (M17HIP_LIB), the list is gone through `--rounds` times so that box drift shows.
    python tools/bisect_bench.py [--rounds 2] [--single] _exp/bis/libm17hip_d8d49ef.so m17-cxx-demod_amd/libm17hip.so ...
M17_BISECT_TUNE=key=value[,key=value] applies m17hip_tune settings to every context of every build.
--single adds the single-stream regime (two groups of 2048 channels, state carried, m17hip_demod_front) where the build has it — created AFTER the two-batch
contexts were used and destroyed (a process with history).  M17_BISECT_PLACEHOLDERS=n: n foreign streams in front of every context; M17_BISECT_HOST_STREAMS=1: a host
stream per context as up to round 5 (builds without m17hip_get_stream always)."""
import os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

def worker(single):
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "24")
    import torch
    sys.path.insert(0, os.path.join(ROOT, 'm17-cxx-demod_amd')); sys.path.insert(0, os.path.join(ROOT, 'tests'))
    import m17hip, oracle_lib as ol
    m17hip.Context._warned = True
    tunes = [tuple(int(x) for x in kv.split('=')) for kv in os.environ.get('M17_BISECT_TUNE', '').split(',') if kv]
    # M17_BISECT_HOST_STREAMS=1: a torch stream handed to every context (m17hip_set_stream), as up to round 5; default: the library's own stream sets (round 6)
    HOST_STREAMS = os.environ.get('M17_BISECT_HOST_STREAMS', '0') == '1' or not hasattr(m17hip.load_library(), 'm17hip_get_stream')
    class Context(m17hip.Context):
        def __init__(self, *a, **k):
            super().__init__(*a, **k)
            for key, val in tunes: self.tune(key, val)
    m17hip.Context = Context
    C, T = 4096, 480000
    p = ol.gen_params(seed=20260101, kind=-1, n_frames=T // 1920 - 6, lead_in=3072, noise_sigma=600., tail_sigma=600., lead_sigma=40000.0, total=T)
    ctxs, streams = [], []
    nph = int(os.environ.get('M17_BISECT_PLACEHOLDERS', '0'))   # streams created (and never used) in front of every context: shifts which streams share a dispatch pipe (NOTES 4.14)
    placeholders = []
    def hip_stream():   # a REAL new HIP stream (torch.cuda.Stream() hands out streams of a pool it creates all at once)
        import ctypes
        path = [l.split()[-1] for l in open('/proc/self/maps') if 'libamdhip64' in l][0]
        h = ctypes.c_void_p()
        rc = ctypes.CDLL(path).hipStreamCreateWithFlags(ctypes.byref(h), 1)
        assert rc == 0, rc
        return h
    for f in range(2):
        placeholders += [hip_stream() for _ in range(nph)]
        c = m17hip.Context(C, T)
        if HOST_STREAMS: streams.append(torch.cuda.Stream(priority=int(os.environ.get("M17_BISECT_PRIO", "0")))); c.set_stream(streams[-1].cuda_stream)
        c.synth(p, C, T); ctxs.append(c)
    def groups(n):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for k0 in range(0, n, 2):
            for k in (k0, k0 + 1): ctxs[k % 2].reset(); ctxs[k % 2].run()
            for k in (k0, k0 + 1): ctxs[k % 2].frames_count()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) * 1e3 / n
    groups(48)
    out = ['%.2f' % groups(40) for _ in range(3)]
    n_rec = int(ctxs[0].frames_count())
    res = 'two-batch ms/step: ' + ' '.join(out) + '  records %d' % n_rec
    if single and hasattr(m17hip.load_library(), 'm17hip_demod_front'):
        for c in ctxs: c.close()
        G = 2; Cg = C // G
        gs, ss = [], []
        try:
            for g in range(G):
                placeholders += [hip_stream() for _ in range(nph)]
                c = m17hip.Context(Cg, T)
                if HOST_STREAMS: ss.append(torch.cuda.Stream(priority=int(os.environ.get("M17_BISECT_PRIO", "0")))); c.set_stream(ss[-1].cuda_stream)
                c.synth(p, Cg, T, chan0=g * Cg); c.tune(16, 1); c.synth(p, Cg, T, chan0=g * Cg); c.tune(16, 0)
                c.reset(); c.run(); gs.append(c)
            def stream(n):
                torch.cuda.synchronize(); t0 = time.perf_counter()
                for k in range(n):
                    for c in gs: c.input_alternate(Cg, T); c.front()
                    for c in gs: c.frames_count(); c.run()
                for c in gs: c.frames_count()
                torch.cuda.synchronize()
                return (time.perf_counter() - t0) * 1e3 / n
            stream(8)
            res += '   single-stream ms/step: ' + ' '.join('%.2f' % stream(20) for _ in range(2))
        except Exception as e:
            res += '   single-stream: %r' % (e,)
    print(res, flush=True)

if __name__ == '__main__':
    if sys.argv[1] == '--worker':
        worker(sys.argv[2] == '1'); sys.exit(0)
    rounds, single, libs = 2, False, []
    a = sys.argv[1:]
    while a:
        x = a.pop(0)
        if x == '--rounds': rounds = int(a.pop(0))
        elif x == '--single': single = True
        else: libs.append(x)
    for r in range(rounds):
        for lib in libs:
            env = dict(os.environ, M17HIP_LIB=os.path.abspath(lib))
            pr = subprocess.run([sys.executable, os.path.abspath(__file__), '--worker', '1' if single else '0'], env=env, capture_output=True, text=True)
            tail = (pr.stdout.strip().splitlines() or ['<no output> ' + pr.stderr.strip()[-300:]])[-1]
            print('round %d  %-40s %s' % (r, os.path.basename(lib), tail), flush=True)
