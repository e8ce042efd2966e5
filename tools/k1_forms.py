"""K1 alone, form against form on one box: the rolled R = 15 kernel (m17hip_tune key 11 = 0) and the skewed-pair kernel on bounded grids of
several sizes (key 13), whole run and as ten segments; outputs compared bit for bit between the forms.
    python tools/k1_forms.py [lib.so ...]        (every library in its own process when several are given)"""
import os, subprocess, sys, time
import _toolslib  # noqa: F401  (key 11 = round 4's kernel exists in the measurement build only)
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

def worker():
    import numpy as np, torch
    sys.path.insert(0, os.path.join(ROOT, 'm17-cxx-demod_amd')); sys.path.insert(0, os.path.join(ROOT, 'tests'))
    import m17hip, oracle_lib as ol
    C, T = 4096, 480000
    p = ol.gen_params(seed=20260101, kind=-1, n_frames=T // 1920 - 6, lead_in=3072, noise_sigma=600., tail_sigma=600., lead_sigma=40000.0, total=T)
    ctx = m17hip.Context(C, T)
    ctx.synth(p, C, T)
    def timed(f, n=5):
        ts = []
        for _ in range(n):
            torch.cuda.synchronize(); t = time.perf_counter(); f(); torch.cuda.synchronize(); ts.append((time.perf_counter() - t) * 1e3)
        return min(ts)
    def ybytes(rows=64):   # the first rows of the matched-filter output of a 48 000-sample call, for the comparison
        c2 = m17hip.Context(rows, 48000)
        for k, v in cur.items(): c2.tune(k, v)
        c2.synth(p, rows, 48000)
        y = c2.fir()
        c2.close()
        return y.tobytes()
    ref = None
    for name, tunes in (("rolled R=15, one workgroup per tile", {11: 0}), ("skewed pairs, grid 256", {11: 1, 13: 256}), ("skewed pairs, grid 512", {11: 1, 13: 512}),
                        ("skewed pairs, grid 1024", {11: 1, 13: 1024}), ("skewed pairs, grid 1280 (default)", {11: 1, 13: 0}), ("skewed pairs, grid 2048", {11: 1, 13: 2048}),
                        ("skewed pairs, grid 65536", {11: 1, 13: 65536})):
        cur = tunes
        try:
            for k, v in tunes.items(): ctx.tune(k, v)
        except Exception as e:
            print('%-40s not in this build (%s)' % (name, e)); continue
        ctx.T = T
        whole = timed(lambda: ctx.fir(fetch=False))
        ctx.T = 48000
        segs = timed(lambda: [ctx.fir(fetch=False) for _ in range(10)])
        yb = ybytes()
        if ref is None: ref = yb
        print('%-40s whole run %.2f ms | 10 x 48000 %.2f ms | output == first form: %s' % (name, whole, segs, yb == ref), flush=True)

if __name__ == '__main__':
    if len(sys.argv) > 1 and sys.argv[1] == '--worker':
        worker(); sys.exit(0)
    libs = sys.argv[1:] or [os.path.join(ROOT, 'm17-cxx-demod_amd', 'libm17hip_tools.so')]
    for lib in libs:
        print('==', lib, flush=True)
        env = dict(os.environ, M17HIP_LIB=os.path.abspath(lib))
        subprocess.run([sys.executable, os.path.abspath(__file__), '--worker'], env=env)
