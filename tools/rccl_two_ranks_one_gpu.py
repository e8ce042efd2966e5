"""Can two ranks share one GPU under RCCL?  (If yes, m17hip_gather_frames can be exercised with nranks = 2 on a 1-GPU box.)
Usage: rccl_two_ranks_one_gpu.py  (spawns two processes)"""
import os, sys, subprocess, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WORKER = r'''
import os, sys, numpy as np
sys.path.insert(0, os.path.join(%r, "m17-cxx-demod_amd")); sys.path.insert(0, os.path.join(%r, "tests"))
import m17hip, oracle_lib as ol
rank, idfile = int(sys.argv[1]), sys.argv[2]
CT, T = 20, 48000
p = ol.gen_params(seed=77, kind=-1, n_frames=20, lead_in=3072, noise_sigma=500.0, tail_sigma=500.0, lead_sigma=40000.0, total=T)
lo, hi = (0, 10) if rank == 0 else (10, 20)
ctx = m17hip.Context(hi - lo, T); ctx.set_channel_base(lo); ctx.synth(p, hi - lo, T, chan0=lo)
if rank == 0:
    open(idfile + ".tmp", "wb").write(m17hip.comm_get_id()); os.rename(idfile + ".tmp", idfile)
import time
while not os.path.exists(idfile): time.sleep(0.05)
cid = open(idfile, "rb").read()
comm = m17hip.Comm(ctx, cid, rank, 2)
ctx.reset(); ctx.run()
recs, counts = ctx.gather_frames(comm, root=0)
print("rank", rank, "counts", counts.tolist(), "recs", None if recs is None else recs.size, flush=True)
if rank == 0:
    whole = m17hip.Context(CT, T); whole.synth(p, CT, T); whole.reset(); whole.run(); one = whole.frames()
    print("gathered == one big run:", recs.tobytes() == one.tobytes(), flush=True)
'''
d = tempfile.mkdtemp()
open(os.path.join(d, "w.py"), "w").write(WORKER % (ROOT, ROOT))
ps = [subprocess.Popen([sys.executable, os.path.join(d, "w.py"), str(r), os.path.join(d, "id")]) for r in range(2)]
print("exit codes", [p.wait(timeout=300) for p in ps])
