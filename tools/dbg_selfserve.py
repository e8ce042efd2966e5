import sys, os
R = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, os.path.join(R, "m17-cxx-demod_amd")); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np, torch
import m17hip, oracle_lib as ol
T = 96000
p = ol.gen_params(seed=61, kind=-1, n_frames=max(1, T // 1920 - 4), lead_in=3072, noise_sigma=500.0, tail_sigma=500.0, lead_sigma=40000.0, total=T)
x = ol.generate_batch(p, 32, T, threads=8)[:4, :24000]
c = m17hip.Context(4, 24000)
c.tune(9, 200); c.tune(3, 0)
c.upload(x); c.reset(); c.run()
got = c.frames()
logs = c.diag_log(4, 200)
for ch in (1,):
    e = ol.demod_diag_log(x[ch])
    g = logs[ch]
    print("entries", len(g), len(e))
    for i in range(min(len(g), len(e))):
        if g[i].tobytes() != e[i].tobytes():
            for j in range(max(0, i - 2), min(len(g), len(e), i + 3)):
                print(j, "GPU", g[j]); print(j, "ORA", e[j])
            break
    recs, _ = ol.demod(x[ch])
    print("oracle recs", [(int(r["sample_pos"]), int(r["frame_type"]), int(r["cost"])) for r in recs[:6]])
    gr = got[got["channel"] == ch]
    print("gpu    recs", [(int(r["sample_pos"]), int(r["frame_type"]), int(r["cost"])) for r in gr[:6]])
