"""K1 alone, ten launches over 4096 x 480 000 samples per form (round 4's rolled kernel: m17hip_tune key 11 = 0; the skewed-pair kernel), for the
PMC pass that measures the clock the chip holds under it (GRBM_GUI_ACTIVE / duration): tools/profile_round.sh."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import _toolslib  # noqa: F401  (key 11 = round 4's kernel exists in the measurement build only)
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'm17-cxx-demod_amd')); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch  # noqa: F401
import m17hip, oracle_lib as ol
C, T = 4096, 480000
p = ol.gen_params(seed=20260101, kind=-1, n_frames=T // 1920 - 6, lead_in=3072, noise_sigma=600., tail_sigma=600., lead_sigma=40000.0, total=T)
ctx = m17hip.Context(C, T)
ctx.synth(p, C, T)
for form in (0, 1):
    ctx.tune(11, form)
    for _ in range(10):
        ctx.fir(fetch=False)
print("done")
