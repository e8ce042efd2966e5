"""Step time of the bench workload under tuning-knob settings: knob_bench.py C T key=value[,key=value] ...  (one configuration per argument)"""
import sys, os, time, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'm17-cxx-demod_amd')); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import m17hip, oracle_lib as ol
if os.environ.get('M17HIP_LIB'): m17hip.LIB_PATH = os.environ['M17HIP_LIB']   # experiment builds
C, T = int(sys.argv[1]), int(sys.argv[2])
p = ol.gen_params(seed=20260101, kind=-1, n_frames=T // 1920 - 6, lead_in=3072, noise_sigma=600., tail_sigma=600., lead_sigma=40000.0, total=T)
ctx = m17hip.Context(C, T)
ctx.synth(p, C, T)
base = None
for rnd in range(2):
    for cfg in sys.argv[3:]:
        kv = [tuple(int(v) for v in item.split('=')) for item in cfg.split(',') if item]
        for k, v in kv: ctx.tune(k, v)
        ts = []
        for rep in range(6):
            ctx.reset(); torch.cuda.synchronize()
            t = time.perf_counter(); ctx.run(); torch.cuda.synchronize(); ts.append((time.perf_counter() - t) * 1e3)
        n = ctx.frames_count()
        print('%-24s frames %d  step ms: %s  (min %.2f)' % (cfg, n, ' '.join('%.2f' % v for v in ts), min(ts[1:])), flush=True)
