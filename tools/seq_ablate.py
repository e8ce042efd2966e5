"""Timing experiments on the sequential kernel (GPU box): waves-per-block sweep + per-channel tick counters."""
import sys, os, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'm17-cxx-demod_amd')); sys.path.insert(0, os.path.join(ROOT, 'tests')); sys.path.insert(0, os.path.join(ROOT, 'tools'))
import _toolslib  # noqa: F401  (the measurement build of the library)
import m17hip, oracle_lib as ol
C, T = int(sys.argv[1]), int(sys.argv[2])
p = ol.gen_params(seed=20260101, kind=-1, n_frames=T//1920-6, lead_in=3072, noise_sigma=600., tail_sigma=600., lead_sigma=40000.0, total=T)
x = ol.generate_batch(p, C, T, threads=64)
ctx = m17hip.Context(C, T); ctx.upload(x)
for wpb in (4,):
    ctx.tune(1, 1); ctx.reset(); ctx.timing(True); ctx.timing_reset(); ctx.run(); d = ctx.diag()
    dc = ctx.debug_counters(C).astype(np.float64); m = np.median(dc, axis=0); mx = dc.max(axis=0)
    print(f"wpb={wpb} ms: fir={ctx.timing_get('fir_rrc150')[0]:.2f} dcd={ctx.timing_get('dcd')[0]:.2f} seq={ctx.timing_get('demod_seq')[0]:.2f} frames={int(d['n_frames'].sum())}")
    print('   per-channel ticks(10ns) median: total %.3g bulk %.3g scalar %.3g decode %.3g | n_bulk %d n_scalar %d bulk_samples %d flips %d decodes %d | max total %.3g'
          % (m[0], m[1], m[2], m[3], m[4], m[5], m[6], int(m[7]) & 0xFFFFFFFF, int(m[7]) >> 32, mx[0]), flush=True)
    print('   decode sections per frame (us, median channel): depuncture %.1f trellis %.1f chainback %.1f | whole decode %.1f' % tuple(m[k] / max(int(m[7]) >> 32, 1) / 100 for k in (9, 10, 11, 3)), flush=True)
    print('   ms (median channel): chunks %.2f [window %.2f symbols %.2f iir %.2f rest %.2f] search %.2f scalar %.2f decode %.2f patch %.2f carrier-off %.2f | unaccounted %.2f' % (m[1]/1e5, m[12]/1e5, m[13]/1e5, m[14]/1e5, (m[1]-m[12]-m[13]-m[14])/1e5, m[15]/1e5, m[2]/1e5, m[3]/1e5, m[8]/1e5, m[16]/1e5, (m[0]-m[1]-m[2]-m[3]-m[15]-m[16]-m[8])/1e5), flush=True)
    dr = dc[:, 17] > 0
    print('   channels that dropped the limit speculation: %d of %d; their total ms median %.1f max %.1f; others median %.1f max %.1f' % (dr.sum(), len(dr), np.median(dc[dr, 0]) / 1e5 if dr.any() else 0, dc[dr, 0].max() / 1e5 if dr.any() else 0, np.median(dc[~dr, 0]) / 1e5, dc[~dr, 0].max() / 1e5), flush=True)
    print('   single-sample steps by state (median channel): UNLOCKED %d LSF_SYNC %d STREAM_SYNC %d PACKET_SYNC %d BERT_SYNC %d SYNC_WAIT/FRAME %d' % tuple(int(m[18 + q]) for q in range(6)), flush=True)
    tot = dc.sum(axis=0)
    order = np.argsort(-dc[:, 0])
    print('   slowest channels: ' + ' | '.join(f"ch{int(i)} tot={dc[i,0]/1e5:.1f}ms bulk={dc[i,1]/1e5:.1f} scal={dc[i,2]/1e5:.1f} dec={dc[i,3]/1e5:.1f} nb={int(dc[i,4])} ns={int(dc[i,5])} fl={int(dc[i,7])&0xFFFFFFFF} nd={int(dc[i,7])>>32} frames={int(d['n_frames'][i])} st={int(d['demod_state'][i])}" for i in order[:6]))
    pct = np.percentile(dc[:, 0], [50, 90, 99, 100]) / 1e5
    print('   total ms percentiles 50/90/99/100:', np.round(pct, 1), ' n_scalar pct:', np.percentile(dc[:, 5], [50, 90, 99, 100]).astype(int))
