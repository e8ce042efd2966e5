#!/usr/bin/env python3
"""The persistent K2 / K5 (m17hip_tune key 22) on its own: runs of C channels x T samples, one after the other, with the hand-over statistics.

    python tools/persist_probe.py [--channels 4096] [--samples 480000] [--runs 4] [--k2-wait-us 20000] [--k5-wait-us 300000]
"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "m17-cxx-demod_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")

import torch  # noqa: E402

import m17hip  # noqa: E402
import oracle_lib as ol  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--channels", type=int, default=4096)
    ap.add_argument("--samples", type=int, default=480000)
    ap.add_argument("--runs", type=int, default=4)
    ap.add_argument("--k2-wait-us", type=int, default=20000)
    ap.add_argument("--k5-wait-us", type=int, default=300000)
    ap.add_argument("--lds", type=int, default=0, help="m17hip_tune key 14: LDS bytes of a K5 workgroup (0 = default)")
    ap.add_argument("--stream", type=int, default=1, help="run on a torch stream of its own (0: the default stream)")
    args = ap.parse_args()
    C, T = args.channels, args.samples
    p = ol.gen_params(seed=20260101, kind=-1, n_frames=max(1, T // 1920 - 6), lead_in=3072, noise_sigma=600.0, tail_sigma=600.0, lead_sigma=40000.0, total=T)
    for persist in (0, 1):
        ctx = m17hip.Context(C, T)
        if args.stream:
            st = torch.cuda.Stream()
            ctx.set_stream(st.cuda_stream)
        ctx.tune(22, persist)
        if args.lds:
            ctx.tune(14, args.lds)
        ctx.tune(23, args.k2_wait_us)
        ctx.tune(24, args.k5_wait_us)
        ctx.synth(p, C, T)
        ctx.reset()
        for i in range(args.runs):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            ctx.run()
            try:
                n = ctx.frames_count()
            except m17hip.M17HipError as e:
                n = str(e)[:40]
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) * 1e3
            print(f"persist {persist} run {i}: {dt:8.2f} ms  frames {n}  stats (k5 gave up, k2 went on) {ctx.persist_stats()}", flush=True)
        ctx.close()


if __name__ == "__main__":
    main()
