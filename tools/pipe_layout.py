"""Which streams may share a hardware dispatch pipe?  (NOTES 4.14 / 5.x)  A pool of real HIP streams is created in ONE go, so that two of them
share a pipe exactly when their pool indices are equal mod 4; every role stream of two contexts (main, K3, K1, replay, copy) is then taken
from the pool by class, and the two regimes are timed per layout.  Tools build of the library (m17hip_tune keys 40-43).
    python tools/pipe_layout.py [--regime both|two|single] [layout ...]      layout = 5 digits (classes of main, K3, K1, replay, copy) + optional
    '+r' = the second context's classes rotated by r"""
import ctypes, os, subprocess, sys, time
import _toolslib  # noqa: F401
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

def worker(layout, regime):
    os.environ["GPU_MAX_HW_QUEUES"] = "24"
    import torch
    sys.path.insert(0, os.path.join(ROOT, 'm17-cxx-demod_amd')); sys.path.insert(0, os.path.join(ROOT, 'tests'))
    import m17hip, oracle_lib as ol
    m17hip.Context._warned = True
    torch.cuda.init(); torch.zeros(1, device='cuda')
    hip = ctypes.CDLL([l.split()[-1] for l in open('/proc/self/maps') if 'libamdhip64' in l][0])
    lo, hi = ctypes.c_int(), ctypes.c_int()
    hip.hipDeviceGetStreamPriorityRange(ctypes.byref(lo), ctypes.byref(hi))
    rot = 0
    if '+' in layout: layout, rot = layout.split('+')[0], int(layout.split('+')[1])
    cls = [int(ch) for ch in layout]          # classes of main, K3, K1, replay, copy
    # pool: 24 streams created consecutively; stream i is in class i % 4.  The replay stream of a context has the highest priority: its pool
    # entries are created that way (positions known in advance: the classes of the two contexts' replay streams)
    taken, pool = set(), []
    def need(c, hi_prio):   # first free pool index of class c (whose priority matches)
        for i in range(len(plan)):
            if i % 4 == c and i not in taken and plan[i] == hi_prio: taken.add(i); return i
        raise SystemExit('pool too small')
    plan = [False] * 24
    for j in range(2):   # the two replay streams: the first free entries of their classes are created with the highest priority
        c = (cls[3] + (rot if j else 0)) % 4
        i = next(i for i in range(24) if i % 4 == c and not plan[i]); plan[i] = True
    for i in range(24):
        h = ctypes.c_void_p()
        rc = hip.hipStreamCreateWithPriority(ctypes.byref(h), 1, hi.value if plan[i] else lo.value)
        assert rc == 0
        pool.append(h)
    C, T = 4096, 480000
    p = ol.gen_params(seed=20260101, kind=-1, n_frames=T // 1920 - 6, lead_in=3072, noise_sigma=600., tail_sigma=600., lead_sigma=40000.0, total=T)
    def make(j, Cn, chan0, staged):
        c = m17hip.Context(Cn, T)
        r = rot if j else 0
        idx = [need((cls[k] + r) % 4, k == 3) for k in range(5)]
        c.set_stream(pool[idx[0]].value)
        for k in (1, 2, 3, 4): c.tune(39 + k, pool[idx[k]].value)
        c.synth(p, Cn, T, chan0=chan0)
        if staged:
            c.tune(16, 1); c.synth(p, Cn, T, chan0=chan0); c.tune(16, 0); c.reset(); c.run()
        return c, idx
    out = 'layout %s+%d' % (layout, rot)
    if regime in ('both', 'two'):
        taken.clear()
        ctxs = [make(j, C, 0, False) for j in range(2)]
        out += '  pool idx %s' % [i for _, i in ctxs]
        ctxs = [c for c, _ in ctxs]
        def groups(n):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for k0 in range(0, n, 2):
                for k in (k0, k0 + 1): ctxs[k % 2].reset(); ctxs[k % 2].run()
                for k in (k0, k0 + 1): ctxs[k % 2].frames_count()
            torch.cuda.synchronize()
            return (time.perf_counter() - t0) * 1e3 / n
        groups(40)
        out += '  two-batch %.2f %.2f' % (groups(30), groups(30))
        for c in ctxs: c.close()
    if regime in ('both', 'single'):
        taken.clear()
        G = 2; Cg = C // G
        gs = [make(j, Cg, j * Cg, True)[0] for j in range(G)]
        def stream(n):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for k in range(n):
                for c in gs: c.input_alternate(Cg, T); c.front()
                for c in gs: c.frames_count(); c.run()
            for c in gs: c.frames_count()
            torch.cuda.synchronize()
            return (time.perf_counter() - t0) * 1e3 / n
        stream(8)
        out += '  single-stream %.2f %.2f' % (stream(16), stream(16))
    print(out, flush=True)

if __name__ == '__main__':
    if len(sys.argv) > 1 and sys.argv[1] == '--worker':
        worker(sys.argv[2], sys.argv[3]); sys.exit(0)
    a = sys.argv[1:]
    regime = 'both'
    if a and a[0] == '--regime': regime = a[1]; a = a[2:]
    layouts = a or ['01230', '01230+1', '01230+2', '01230+3',            # all roles apart (copy with main); second context rotated
                    '00123', '01023', '01203', '01123', '01213', '01223', '01233',  # one pair of roles together
                    '00112', '01012', '01102', '00012', '00102', '01002', '01112']
    for l in layouts:
        pr = subprocess.run([sys.executable, os.path.abspath(__file__), '--worker', l, regime], capture_output=True, text=True)
        print((pr.stdout.strip().splitlines() or ['<no output> ' + pr.stderr.strip()[-400:]])[-1], flush=True)
