"""BASELINE config 5 on one GPU: impairment sweep with per-channel BER (PRBS9 receiver on the device, m17hip_bert_stats) and EVM
(SymbolEvm through m17hip_diag_fetch).  Input is synthesized on the device (m17hip_synth_i16): C BERT channels x T samples per point."""
import sys, os, time, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'm17-cxx-demod_amd')); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import m17hip, oracle_lib as ol
C, T = int(sys.argv[1]), int(sys.argv[2])
ctx = m17hip.Context(C, T); ctx.tune(6, 1)
print('| AWGN sigma (LSB) | DC offset (LSB) | gain | channels decoding | frames | mean BER | worst-channel BER | median EVM | Msamples/s (run only) |')
print('|---|---|---|---|---|---|---|---|---|')
for sigma in (0.0, 400.0, 800.0, 1500.0, 2500.0, 4000.0):
    for dc, gain in ((0.0, 1.0), (1000.0, 1.0), (-2500.0, 0.7)):
        p = ol.gen_params(seed=777, kind=0, n_frames=T // 1920 + 2, lead_in=3072, lead_sigma=40000.0, noise_sigma=sigma, tail_sigma=max(sigma, 100.0),   # the burst runs to the end of the slab
                          dc_offset=dc, gain=gain, total=T)
        ctx.synth(p, C, T)
        ctx.reset()
        t0 = time.perf_counter(); ctx.run(); n = ctx.frames_count() if hasattr(ctx, 'frames_count') else None; dt = time.perf_counter() - t0
        st = ctx.bert_stats(C); d = ctx.diag()
        ok = st['bits'] > 0
        ber = st['errors'][ok] / np.maximum(st['bits'][ok], 1)
        print(f"| {sigma:.0f} | {dc:.0f} | {gain} | {int(ok.sum())} / {C} | {int(st['frames'].sum())} | {ber.mean() if ok.any() else float('nan'):.2e} | "
              f"{ber.max() if ok.any() else float('nan'):.2e} | {np.median(d['evm'][ok]) if ok.any() else float('nan'):.3f} | {C * T / dt / 1e6:.0f} |", flush=True)
