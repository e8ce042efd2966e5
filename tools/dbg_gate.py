import os, sys, numpy as np
os.environ.setdefault('GPU_MAX_HW_QUEUES', '16')
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'm17-cxx-demod_amd')); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch, m17hip, oracle_lib as ol
Cn, T = 256, 480000
p = ol.gen_params(seed=20260101, kind=-1, n_frames=T // 1920 - 6, lead_in=3072, noise_sigma=600.0, tail_sigma=600.0, lead_sigma=40000.0, total=T)
for mode in (0, 1, -1):
    a, b = m17hip.Context(Cn, T), m17hip.Context(Cn, T)
    for c_ in (a, b): c_.synth(p, Cn, T); c_.tune(26, mode)
    x = a.download()
    b.reset(); seq = []
    for k in range(3):
        b.run(); seq.append(b.frames().copy())
    pin = torch.from_numpy(x).pin_memory()
    a.reset(); a.run(); parts = []
    for k in range(3):
        if k + 1 < 3:
            if k == 0: a.upload_async(pin.data_ptr(), Cn, T)
            else: a.input_alternate(Cn, T)
            a.front(channels=Cn, samples=T)
        parts.append(a.frames().copy())
        if k + 1 < 3: a.run(channels=Cn, samples=T)
    a.upload_wait()
    for k in range(3):
        same = parts[k].tobytes() == seq[k].tobytes()
        print('mode', mode, 'run', k, 'pipelined == sequential', same, parts[k].size, seq[k].size)
        if not same:
            n = min(parts[k].size, seq[k].size)
            bad = sorted(set(int(parts[k][i]['channel']) for i in range(n) if parts[k][i].tobytes() != seq[k][i].tobytes()))
            print('   channels', bad[:30], len(bad))
    if mode == 0:
        recs, counts, _ = ol.demod_batch(np.tile(x[:8], (1, 3)), cap=2 * (3 * T // 1920 + 2) + 4, threads=8)
        exp = np.concatenate([recs[i, :counts[i]] for i in range(8)])
        for name, pp in (('pipelined', parts), ('sequential', seq)):
            g = np.concatenate(pp); g = g[np.lexsort((g['seq'], g['channel']))]; g = g[g['channel'] < 8]
            print('   ', name, '== oracle (8 ch):', g.tobytes() == exp.tobytes())
    a.close(); b.close()
