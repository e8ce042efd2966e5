"""How many frame records does the unpinned evaluation order of the Kalman updates move?  (VERDICT r1 'Next round' #1)

KalmanFilter.h:49-64 keeps S and K as lazy blaze expressions; blaze is absent from the reference tree, so both the oracle
(oracle/m17_oracle_dsp.hpp kalman_order) and the HIP path (m17hip_set_kalman_order) carry the order as a switch:
  bit 0  x += K*y   : blaze-restructured (float product, then * 1/S)  vs eager double gain
  bit 1  P -= K*H*P : blaze-restructured                               vs eager double gain
  bit 2  F*(P*F^T)  instead of (F*P)*F^T
This tool runs the randomised scenarios of tools/parity_sweep.py (bursts of random kind / length / noise / lead-in / DC /
gain / phase, noise between them) under every order and counts, against the default order 3, the channels and the frame
records that differ at all, and the records whose payload / cost / sample_pos differ.

  kalman_sensitivity.py <first seed> <n seeds> [cpu|gpu] [channels] [samples]
cpu: the oracle on all host threads (no GPU needed).  gpu: the HIP path for the sweep (fast) and, for every order, the
oracle on the first seed to assert HIP == oracle bit-exact under that order."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "m17-cxx-demod_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import ctypes as C  # noqa: E402

import oracle_lib as ol  # noqa: E402

first, nseeds = int(sys.argv[1]), int(sys.argv[2])
mode = sys.argv[3] if len(sys.argv) > 3 else "cpu"
Cn = int(sys.argv[4]) if len(sys.argv) > 4 else 64
T = int(sys.argv[5]) if len(sys.argv) > 5 else 96000
ORDERS = list(range(8))
BASE = 3
CAP = 2 * (T // 1920 + 2) + 4
threads = len(os.sched_getaffinity(0))


def scenario(seed):
    rng = np.random.default_rng(seed)
    x = np.zeros((Cn, T), dtype=np.int16)
    for c in range(Cn):
        pos = 0
        while pos < T - 8000:
            n = min(int(rng.integers(6000, 40000)), T - pos)
            p = ol.gen_params(seed=int(rng.integers(1, 1 << 30)), kind=int(rng.choice([0, 1, 2, 4])), n_frames=int(rng.integers(1, 16)),
                              lead_in=int(rng.integers(0, 5000)), lead_sigma=float(rng.choice([0.0, 100.0, 1000.0, 10000.0, 40000.0])),
                              noise_sigma=float(rng.choice([0.0, 100.0, 500.0, 1200.0, 2500.0])), tail_sigma=float(rng.choice([0.0, 100.0, 1000.0, 5000.0])),
                              dc_offset=float(rng.choice([0.0, 0.0, 300.0, -2000.0, 6000.0])), gain=float(rng.choice([1.0, 0.3, 0.7, 1.6])),
                              phase=int(rng.integers(-1, 10)), invert=0, total=n)
            x[c, pos:pos + n] = ol.generate(p)[:n]
            pos += n
    return x, seed & 1


def oracle_run(x, inv, order):
    ol.oracle().m17o_set_kalman_order(C.c_int(order))
    recs, counts, diags = ol.demod_batch(x, invert=inv, cap=CAP, threads=threads)
    ol.oracle().m17o_set_kalman_order(C.c_int(BASE))
    return [recs[c, :counts[c]].copy() for c in range(x.shape[0])], diags


ctx = None
if mode == "gpu":
    import m17hip
    ctx = m17hip.Context(Cn, T)


def hip_run(x, inv, order):
    ctx.set_kalman_order(order)
    ctx.upload(x); ctx.reset(); ctx.run(flags=inv)
    got = ctx.frames(); d = ctx.diag()
    ctx.set_kalman_order(BASE)
    return [got[got["channel"] == c] for c in range(x.shape[0])], d


stat = {o: dict(channels=0, records=0, payload=0, cost=0, pos=0, count=0, diag=0) for o in ORDERS if o != BASE}
tot_records = tot_channels = 0
checked_hip = 0
for seed in range(first, first + nseeds):
    x, inv = scenario(seed)
    runs = {}
    for o in ORDERS:
        runs[o] = hip_run(x, inv, o) if ctx else oracle_run(x, inv, o)
        if ctx and seed == first:   # HIP == oracle under this very order
            exp, ed = oracle_run(x, inv, o)
            for c in range(Cn):
                assert runs[o][0][c].tobytes() == exp[c].tobytes(), (seed, o, c)
            for f in ("evm", "deviation", "offset", "clock", "sample_index", "clock_index", "viterbi_cost", "n_frames"):
                assert np.array_equal(runs[o][1][f], ed[f], equal_nan=True), (seed, o, f)
            checked_hip += 1
    base, bd = runs[BASE]
    tot_channels += Cn
    tot_records += sum(r.size for r in base)
    for o in stat:
        got, gd = runs[o]
        for c in range(Cn):
            a, b = base[c], got[c]
            if a.tobytes() == b.tobytes():
                continue
            stat[o]["channels"] += 1
            if a.size != b.size:
                stat[o]["count"] += 1
            n = min(a.size, b.size)
            diff = np.array([a[i].tobytes() != b[i].tobytes() for i in range(n)])
            stat[o]["records"] += int(diff.sum()) + abs(a.size - b.size)
            stat[o]["payload"] += int(((a["payload"][:n] != b["payload"][:n]).any(axis=1) | (a["frame_type"][:n] != b["frame_type"][:n])).sum())
            stat[o]["cost"] += int((a["cost"][:n] != b["cost"][:n]).sum())
            stat[o]["pos"] += int((a["sample_pos"][:n] != b["sample_pos"][:n]).sum())
        stat[o]["diag"] += int(sum((bd[f].view(np.uint32) != gd[f].view(np.uint32)).sum() for f in ("deviation", "offset", "clock")))
    print(f"seed {seed}: {sum(r.size for r in base)} records", {o: (stat[o]['channels'], stat[o]['records']) for o in stat}, flush=True)

print(f"\n{tot_channels} channel-runs x {T} samples, {tot_records} frame records under order {BASE} ({mode}; "
      f"HIP == oracle asserted for {checked_hip} orders)" if ctx else f"\n{tot_channels} channel-runs x {T} samples, {tot_records} frame records under order {BASE} ({mode})")
print("| order | x += K*y | P -= K*H*P | F*P*F^T | channels that differ | records that differ | payload / type | cost | sample_pos | record count | diag floats (dev, offset, clock) that differ |")
print("|---|---|---|---|---|---|---|---|---|---|---|")
for o in sorted(stat):
    v = stat[o]
    print(f"| {o} | {'blaze' if o & 1 else 'eager'} | {'blaze' if o & 2 else 'eager'} | {'F(PF^T)' if o & 4 else '(FP)F^T'} | {v['channels']} | {v['records']} | "
          f"{v['payload']} | {v['cost']} | {v['pos']} | {v['count']} | {v['diag']} |")
