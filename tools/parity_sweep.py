"""Randomised parity sweep on the GPU box: the scenario generator of tests/test_gpu_parity.py::test_full_chain_random_scenarios over
many seeds, both decode placements, several segment lengths (with and without a ramp), run boundaries at random samples, pipelined runs in both call orders; prints the channels whose records or diagnostics differ from the oracle.
Usage: parity_sweep.py <first seed> <n seeds>"""
import sys, os, numpy as np
os.environ.setdefault('GPU_MAX_HW_QUEUES', '16')
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'm17-cxx-demod_amd')); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import oracle_lib as ol
C, T = 64, 96000
DUMP = len(sys.argv) > 3 and sys.argv[3] == 'dump'   # write the input of the first seed to tools/x_dbg.npy (no GPU needed) and stop
if not DUMP:
    import torch   # before libm17hip.so initialises the system HIP runtime (torch carries its own; tests/conftest.py)
    import m17hip
    ctx = m17hip.Context(C, T)
total_bad = 0
for seed in range(int(sys.argv[1]), int(sys.argv[1]) + int(sys.argv[2])):
    rng = np.random.default_rng(seed)
    x = np.zeros((C, T), dtype=np.int16)
    for c in range(C):
        pos = 0
        while pos < T - 8000:
            n = min(int(rng.integers(6000, 40000)), T - pos)
            p = ol.gen_params(seed=int(rng.integers(1, 1 << 30)), kind=int(rng.choice([0, 1, 2, 4])), n_frames=int(rng.integers(1, 16)),
                              lead_in=int(rng.integers(0, 5000)), lead_sigma=float(rng.choice([0.0, 100.0, 1000.0, 10000.0, 40000.0])),
                              noise_sigma=float(rng.choice([0.0, 100.0, 500.0, 1200.0, 2500.0])), tail_sigma=float(rng.choice([0.0, 100.0, 1000.0, 5000.0])),
                              dc_offset=float(rng.choice([0.0, 0.0, 300.0, -2000.0, 6000.0])), gain=float(rng.choice([1.0, 0.3, 0.7, 1.6])),
                              phase=int(rng.integers(-1, 10)), invert=0, total=n)
            x[c, pos:pos + n] = ol.generate(p)[:n]; pos += n
    inv = seed & 1
    if DUMP:
        np.save(os.path.join(ROOT, 'tools', 'x_dbg.npy'), x); print('wrote tools/x_dbg.npy, invert =', inv); sys.exit(0)
    recs, counts, diags = ol.demod_batch(x, invert=inv, cap=2 * (T // 1920 + 2) + 4, threads=os.cpu_count())
    cuts = sorted(int(v) for v in rng.integers(1, T, size=2))
    rseg = int(rng.integers(3000, 30000))
    # (payload frames decoded after the run [m17hip_tune 15], segment length, run boundaries, staged + m17hip_demod_front; the redo policy [20] and the EVM fold's place [17] alternate)
    # piped: 1 = the front end of each run queued before the previous run's records are fetched, then the run (the order of rounds 3-5);
    #        2 = front, RUN, then the previous run's records (m17hip_frames_select(1): round 6)
    for spec, seg, pieces, piped in ((1, 19200, None, 0), (1, rseg, None, 0), (0, rseg, None, 0), (1, 0, None, 0), (1, 19200, [0] + cuts + [T], 0),
                                     (1, 19200, [0] + cuts + [T], 1), (0, rseg, [0] + cuts + [T], 1), (1, 4800, [0] + cuts + [T], 1),
                                     (1, 19200, [0] + cuts + [T], 2), (1, rseg, [0] + cuts + [T], 2), (0, 4800, [0] + cuts + [T], 2)):
        ctx.tune(15, spec); ctx.tune(3, seg); ctx.tune(33, (2400, 0, -1)[(seed + piped) % 3]); ctx.tune(20, (seed + spec + piped) & 1); ctx.tune(17, (seed + piped + (seg & 1)) & 1); ctx.tune(26, (seed + spec + (seg >> 2)) & 1); ctx.reset()
        if pieces is None:
            ctx.upload(x); ctx.run(flags=inv); got = ctx.frames()
        elif not piped:   # the same stream as three runs (state, filter history and DCD sums carried between them)
            parts = []
            for a, b in zip(pieces[:-1], pieces[1:]):
                if b > a:
                    ctx.upload(x[:, a:b]); ctx.run(flags=inv); parts.append(ctx.frames().copy())
            got = np.concatenate(parts); got = got[np.lexsort((got['seq'], got['channel']))]
        else:             # ... staged in the second slab pair, the front end of each run queued before the previous run's records are fetched
            import torch
            spans = [(a, b) for a, b in zip(pieces[:-1], pieces[1:]) if b > a]
            pins = [torch.from_numpy(np.ascontiguousarray(x[:, a:b])).pin_memory() for a, b in spans]
            parts = []
            ctx.upload_async(pins[0].data_ptr(), C, spans[0][1] - spans[0][0]); ctx.run(flags=inv, channels=C, samples=spans[0][1] - spans[0][0])
            for i in range(len(spans)):
                if i + 1 < len(spans):
                    n1 = spans[i + 1][1] - spans[i + 1][0]
                    ctx.upload_async(pins[i + 1].data_ptr(), C, n1); ctx.front(flags=inv, channels=C, samples=n1)
                    if piped == 2:
                        ctx.run(flags=inv, channels=C, samples=n1); ctx.frames_select(1)
                parts.append(ctx.frames().copy())
                ctx.frames_select(0)
                if i + 1 < len(spans) and piped != 2:
                    ctx.run(flags=inv, channels=C, samples=n1)
            ctx.upload_wait()
            got = np.concatenate(parts); got = got[np.lexsort((got['seq'], got['channel']))]
        d = ctx.diag()
        bad = [c for c in range(C) if got[got['channel'] == c].tobytes() != recs[c, :counts[c]].tobytes()
               or any(not np.array_equal(d[f][c:c + 1], diags[f][c:c + 1], equal_nan=True) for f in d.dtype.names if f in diags.dtype.names)]
        total_bad += len(bad)
        print(f'seed {seed} invert={inv} deferred_decode={spec} seg={seg} piped={piped} runs={"1" if pieces is None else pieces}: frames {int(counts.sum())}, bad channels {bad}', flush=True)
print('TOTAL bad channel-runs:', total_bad, ' replay drops since the last reset:', ctx.replay_drops())
