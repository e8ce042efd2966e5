"""Debugging aid: first prefix length at which the HIP chain and the oracle disagree on one channel's live state (demod state, clock
count, sync counters, frame count, last diagnostics) — usage: [DEFER=0|1 SEG=n INV=0|1] dbg_bisect.py <x.npy> <channel> [lo hi]"""
import sys, os, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'm17-cxx-demod_amd')); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import m17hip, oracle_lib as ol
CH = int(sys.argv[2])
FULL = int(os.environ.get('FULL', 0))   # 1: run the whole batch (neighbouring channels share waves), compare channel CH
x = np.load(sys.argv[1])
if not FULL:
    x = np.concatenate([x[CH:CH + 1], x[CH:CH + 1]]); CH = 0
DEFER, SEG, INV = int(os.environ.get('DEFER', 1)), int(os.environ.get('SEG', 0)), int(os.environ.get('INV', 0))
ctx = m17hip.Context(x.shape[0], x.shape[1]); ctx.tune(15, DEFER); ctx.tune(3, SEG)
FIELDS = ('demod_state', 'n_frames', 'n_diag', 'dcd', 'locked', 'sample_index', 'sync_index', 'clock_index', 'viterbi_cost', 'clock', 'evm', 'offset', 'deviation', 'dcd_level')
def live(n):
    recs, counts, diags = ol.demod_batch(x[CH:CH + 1, :n], invert=INV, cap=256, threads=1)
    ctx.upload(x[:, :n]); ctx.reset(); ctx.run(flags=INV); d = ctx.diag()
    f = lambda q, i: tuple(np.asarray(q[k][i]).tobytes() for k in FIELDS) + (q['pad'][i].tobytes(),)
    return f(d, CH), f(diags, 0), d[CH:CH + 1], diags
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
