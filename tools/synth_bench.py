"""Device-side input synthesis (m17hip_synth_i16) for the bench workload: wall time, and a spot check against the test generator."""
import sys, os, time, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'm17-cxx-demod_amd')); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import m17hip, oracle_lib as ol
C, T = int(sys.argv[1]), int(sys.argv[2])
p = ol.gen_params(seed=20260101, kind=-1, n_frames=T // 1920 - 6, lead_in=3072, noise_sigma=600., tail_sigma=600., lead_sigma=40000.0, total=T)
ctx = m17hip.Context(C, T)
ctx.synth(p, C, T)
t0 = time.perf_counter(); ctx.synth(p, C, T); dt = time.perf_counter() - t0
t1 = time.perf_counter(); x = ol.generate_batch(p, 64, T, threads=64); dc = time.perf_counter() - t1
got = ctx.download()[:64]
print(f'{C} x {T}: device synthesis {dt*1e3:.1f} ms = {C*T/dt/1e6:.0f} Msamples/s; first 64 channels equal to the test generator: {np.array_equal(got, x)} (CPU generator: {64*T/dc/1e6:.1f} Msamples/s on 64 threads)')
ctx.reset(); ctx.run(); print('frames decoded from the synthesized slab:', ctx.frames_count() if hasattr(ctx, 'frames_count') else len(ctx.frames()))
