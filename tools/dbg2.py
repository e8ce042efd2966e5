import sys, os
R = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, os.path.join(R, "m17-cxx-demod_amd")); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np, torch
import m17hip, oracle_lib as ol
T = 96000
p = ol.gen_params(seed=61, kind=-1, n_frames=max(1, T // 1920 - 4), lead_in=3072, noise_sigma=500.0, tail_sigma=500.0, lead_sigma=40000.0, total=T)
x = ol.generate_batch(p, 32, T, threads=8)
recs, counts, _ = ol.demod_batch(x, cap=2*(T//1920+2)+4, threads=8)
exp = np.concatenate([recs[c,:counts[c]] for c in range(32)])
c = m17hip.Context(32, T)
c.tune(3, int(os.environ.get("SEG", "48000"))); c.upload(x); c.reset(); c.run(); got = c.frames()
print(os.environ.get("M17HIP_LIB", "default"), got.tobytes() == exp.tobytes(), got.size, exp.size)
if "CNT" in os.environ.get("M17HIP_LIB", ""):
    dd = c.debug_counters(32)
    d = dd[:, 23]
    for ch in range(8):
        w = int(dd[ch, 22]); print(ch, "served", w & 0xFFFFFF, "tt_div", (w >> 24) & 0xFFFFFF, "count_at_div", w >> 48)
        v = int(d[ch]); print(ch, "n_serve", v & 0xFFFFFFFF, "diverged", (v >> 32) & 1, "h_until", (v >> 33) & 0x7FFFFF, "n_despec", v >> 56)
