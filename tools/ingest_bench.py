"""PCIe-inclusive rate of the full chain (GPU box): every step gets fresh input from host memory.
   sync   : m17hip_upload_i16 (pinned source), then reset + run + compact
   overlap: m17hip_upload_i16_async of step k+1 queued right after run k (second slab, copy stream)"""
import sys, os, time, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'm17-cxx-demod_amd')); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import m17hip, oracle_lib as ol, torch
C, T, steps = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
p = ol.gen_params(seed=20260101, kind=-1, n_frames=T // 1920 - 6, lead_in=3072, noise_sigma=600., tail_sigma=600., lead_sigma=40000.0, total=T)
x = ol.generate_batch(p, C, T, threads=64)
a = torch.from_numpy(x).pin_memory(); b = torch.from_numpy(x.copy()).pin_memory()
ctx = m17hip.Context(C, T)
buf = torch.zeros(C * (2 * (T // 1920 + 2) + 4) * 64, dtype=torch.uint8, device='cuda')
def finish(): return ctx.frames_compact_device(buf.data_ptr(), buf.numel() // 64)
# sync
ctx.upload(x); ctx.reset(); ctx.run(); finish(); torch.cuda.synchronize()
t0 = time.perf_counter()
for k in range(steps):
    ctx.upload((a if k & 1 else b).numpy()); ctx.reset(); ctx.run(); n = finish()
torch.cuda.synchronize(); ts = (time.perf_counter() - t0) / steps
# overlapped
ctx.upload_async(a.data_ptr(), C, T)
t0 = time.perf_counter()
for k in range(steps):
    ctx.reset(); ctx.run()
    ctx.upload_async((b if k & 1 == 0 else a).data_ptr(), C, T)
    n2 = finish()
torch.cuda.synchronize(); to = (time.perf_counter() - t0) / steps
print(f'C={C} T={T}: upload then run {ts*1e3:.1f} ms/step = {C*T/ts/1e6:.0f} Msamples/s | upload of the next step overlapped {to*1e3:.1f} ms/step = {C*T/to/1e6:.0f} Msamples/s  (frames {n} / {n2}; input {C*T*2/1e9:.2f} GB per step)')
