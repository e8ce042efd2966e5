#!/usr/bin/env python3
"""Single-stream regime (one context, state carried, m17hip_demod_front pipelining) under a list of tuning-knob settings.

    python tools/stream_bench.py [--steps 12] [--channels 4096] "17=3" "17=3,18=3" ...

Each setting is `key=value[,key=value...]` for m17hip_tune ("" = defaults); prints ms/step of the pipelined loop and of the same
runs made strictly one after the other (no front call)."""
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
    ap.add_argument("settings", nargs="*", default=[""])
    ap.add_argument("--steps", type=int, default=12)
    ap.add_argument("--warm", type=int, default=30)
    ap.add_argument("--channels", type=int, default=4096)
    ap.add_argument("--samples", type=int, default=480000)
    ap.add_argument("--serial", type=int, default=1, help="also time the un-pipelined loop")
    ap.add_argument("--groups", type=int, default=1, help="split the channels into this many contexts (independent chains), each pipelined")
    args = ap.parse_args()
    C, T = args.channels, args.samples
    p = ol.gen_params(seed=20260101, kind=-1, n_frames=max(1, T // 1920 - 6), lead_in=3072, noise_sigma=600.0, tail_sigma=600.0, lead_sigma=40000.0, total=T)
    buf = torch.zeros(C * (2 * (T // 1920 + 2) + 4) * 64, dtype=torch.uint8, device="cuda")
    cap = C * (2 * (T // 1920 + 2) + 4)
    if args.groups > 1:
        return groups(args, p)
    for setting in args.settings:
        ctx = m17hip.Context(C, T)
        for kv in filter(None, setting.split(",")):
            k, v = kv.split("=")
            ctx.tune(int(k), int(v))
        ctx.synth(p, C, T)
        ctx.tune(16, 1); ctx.synth(p, C, T); ctx.tune(16, 0)
        ctx.reset(); ctx.run()

        def piped(n):
            for _ in range(n):
                ctx.input_alternate(C, T); ctx.front()
                ctx.frames_compact_device(buf.data_ptr(), cap)
                ctx.run()

        def serial(n):
            for _ in range(n):
                ctx.input_alternate(C, T)
                ctx.frames_compact_device(buf.data_ptr(), cap)
                ctx.run()

        res = {}
        for name, fn in (("pipelined", piped),) + ((("serial", serial),) if args.serial else ()):
            fn(args.warm)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            fn(args.steps)
            ctx.frames_compact_device(buf.data_ptr(), cap)
            torch.cuda.synchronize()
            res[name] = (time.perf_counter() - t0) / args.steps * 1e3
        print(f"{setting or 'defaults':24s} " + "  ".join(f"{k} {v:7.3f} ms/step" for k, v in res.items()), flush=True)
        ctx.close()


def groups(args, p):
    G, C, T = args.groups, args.channels, args.samples
    Cg = C // G
    cap = Cg * (2 * (T // 1920 + 2) + 4)
    for setting in args.settings:
        ctxs, bufs, streams = [], [], []
        for g in range(G):
            c = m17hip.Context(Cg, T)
            st = torch.cuda.Stream(); streams.append(st); c.set_stream(st.cuda_stream)
            for kv in filter(None, setting.split(",")):
                k, v = kv.split("="); c.tune(int(k), int(v))
            c.synth(p, Cg, T, chan0=g * Cg)
            c.tune(16, 1); c.synth(p, Cg, T, chan0=g * Cg); c.tune(16, 0)
            c.reset(); c.run()
            ctxs.append(c); bufs.append(torch.zeros(cap * 64, dtype=torch.uint8, device="cuda"))

        def piped(n):
            for _ in range(n):
                for c in ctxs:
                    c.input_alternate(Cg, T); c.front()
                for c, b in zip(ctxs, bufs):
                    c.frames_compact_device(b.data_ptr(), cap)
                    c.run()
        piped(args.warm)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        piped(args.steps)
        for c, b in zip(ctxs, bufs):
            c.frames_compact_device(b.data_ptr(), cap)
        torch.cuda.synchronize()
        print(f"{setting or 'defaults':24s} {G} groups of {Cg} channels: pipelined {(time.perf_counter() - t0) / args.steps * 1e3:7.3f} ms/step", flush=True)
        for c in ctxs:
            c.close()


if __name__ == "__main__":
    main()
