"""Debugging aid: first prefix length at which the HIP chain and the oracle disagree on one channel's live state (demod state, clock
count, sync counters, frame count, last diagnostics) — usage: dbg_bisect.py <x.npy> <channel> [lo hi]"""
import sys, os, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'm17-cxx-demod_amd')); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import m17hip, oracle_lib as ol
x = np.load(sys.argv[1])[int(sys.argv[2]):int(sys.argv[2]) + 1].copy()
x = np.concatenate([x, x])  # two identical rows (a context wants >= 1 channel; keeps shapes simple)
ctx = m17hip.Context(2, x.shape[1]); ctx.tune(2, 0); ctx.tune(3, 0)
FIELDS = ('demod_state', 'n_frames', 'n_diag', 'dcd', 'locked', 'sample_index', 'sync_index', 'clock_index', 'viterbi_cost', 'clock', 'evm', 'offset', 'deviation', 'dcd_level')
def live(n):
    recs, counts, diags = ol.demod_batch(x[:, :n], cap=256, threads=2)
    ctx.upload(x[:, :n]); ctx.reset(); ctx.run(); d = ctx.diag()
    f = lambda q: tuple(np.asarray(q[k][0]).tobytes() for k in FIELDS) + (q['pad'][0].tobytes(),)
    return f(d), f(diags), d, diags
lo = int(sys.argv[3]) if len(sys.argv) > 3 else 500
hi = int(sys.argv[4]) if len(sys.argv) > 4 else x.shape[1]
g, o, _, _ = live(lo); assert g == o, 'already different at lo'
g, o, _, _ = live(hi); assert g != o, 'no difference at hi'
while hi - lo > 1:
    mid = (lo + hi) // 2
    g, o, _, _ = live(mid)
    if g != o: hi = mid
    else: lo = mid
print('first prefix length with a difference:', hi, '(sample', hi - 1, ')')
for n in (hi - 2, hi - 1, hi, hi + 1):
    _, _, d, q = live(n)
    fmt = lambda z: {k: z[k][0] for k in FIELDS} | {'pad0': hex(int(z['pad'][0][0])), 'sync_count': int(z['pad'][0][1]) & 0xFFFF, 'missing': int(z['pad'][0][1]) >> 16}
    a, b = fmt(d), fmt(q)
    print(n, 'oracle', b)
    print(' ' * len(str(n)), 'diff  ', {k: (a[k], b[k]) for k in a if str(a[k]) != str(b[k])})
