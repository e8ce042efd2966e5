import sys, os, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'm17-cxx-demod_amd')); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import m17hip, oracle_lib as ol
C = 32
a = ol.generate_batch(ol.gen_params(seed=61, kind=-1, n_frames=6, lead_in=3072, noise_sigma=500.0, tail_sigma=2500.0, lead_sigma=40000.0, total=36000), C, 36000, threads=8)
b = ol.generate_batch(ol.gen_params(seed=62, kind=-1, n_frames=8, lead_in=2000, noise_sigma=500.0, tail_sigma=500.0, lead_sigma=2500.0, total=36000), C, 36000, threads=8)
x = np.concatenate([a, b], axis=1)
recs, counts, diags = ol.demod_batch(x, cap=2 * (x.shape[1] // 1920 + 2) + 4, threads=8)
exp = np.concatenate([recs[c, :counts[c]] for c in range(C)])
ctx = m17hip.Context(C, x.shape[1])
for spec, seg, seg0 in [(0, 0, 0), (1, 9600, 0)]:
    try:
        ctx.tune(2, spec); ctx.tune(3, seg); ctx.tune(4, seg0)
    except Exception as e:
        print('tune failed', e)
    ctx.upload(x); ctx.reset(); ctx.run()
    got = ctx.frames()
    ok = got.tobytes() == exp.tobytes()
    msg = ''
    if not ok:
        n = min(got.size, exp.size)
        bad = [i for i in range(n) if got[i].tobytes() != exp[i].tobytes()]
        i = bad[0] if bad else n
        msg = f' sizes {got.size}/{exp.size} first bad rec {i}: got ch={got[i]["channel"]} seq={got[i]["seq"]} pos={got[i]["sample_pos"]} cost={got[i]["cost"]} | exp ch={exp[i]["channel"]} seq={exp[i]["seq"]} pos={exp[i]["sample_pos"]} cost={exp[i]["cost"]}; bad recs {len(bad)} channels {sorted(set(int(exp[j]["channel"]) for j in bad))[:10]}'
    print(f'spec={spec} seg={seg} seg0={seg0}: {"OK" if ok else "MISMATCH"}{msg}', flush=True)
    if not ok and spec == 0:
        for j in bad:
            g, e = got[j], exp[j]
            print('  rec', j, 'type', g['frame_type'], e['frame_type'], 'sync', g['sync_type'] if 'sync_type' in g.dtype.names else '', 'cost', g['cost'], e['cost'], 'len', g['len'], 'payload diff bytes', int((g['payload'] != e['payload']).sum()), 'pos', g['sample_pos'])
        prev = exp[bad[0] - 1]
        print('  previous record of that channel: type', prev['frame_type'], 'cost', prev['cost'], 'pos', prev['sample_pos'])
        d = ctx.diag()
        for f in d.dtype.names:
            if f in diags.dtype.names and not np.array_equal(d[f][14:15], diags[f][14:15], equal_nan=True): print('  diag differs', f, d[f][14], diags[f][14])
