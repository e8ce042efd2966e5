#!/usr/bin/env python3
"""Headline benchmark: Msamples/s demodulated (48 kSPS int16 4-FSK in -> decoded M17 frames) on MI355X.

One "step" = one pass of the whole demodulation chain (K1 RRC FIR, K3 sliding-DFT carrier detect, K2 limit filter run
ahead of K5, K5 sequential demodulator with K4 Viterbi/frame decode, record compaction) over C channels x T samples of synthetic
baseband that is already resident in HBM.  N > 1: one process per GPU (torch.distributed / RCCL), channels sharded
contiguously (records carry global channel ids), no data-path collective; every step ends with the RCCL gather of the decoded
frame records to rank 0 (m17hip_gather_frames_device; the torch all_gather of m17hip/dist.py if the C-ABI communicator cannot
be had).

    python bench.py --gpus 1 --steps 5 --warmup 2
    python bench.py --gpus 8                       # starts its 8 ranks itself (one process per GPU, 127.0.0.1 rendezvous); --dry-launch shows them
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...
    python bench.py --config 2        # BASELINE configs[1]: 1024 channels, FIR + correlator outputs materialised (26 B/sample)

Prints ONE JSON line on rank 0 (contract in the task statement) incl. `roofline` and `cpu_baseline`.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (os.path.join(ROOT, "m17-cxx-demod_amd"), os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np  # noqa: E402

HBM_PEAK_GBS = 8000.0           # MI355X HBM3E spec (MI355X_MICROARCH.md)
# ALGORITHMIC bytes per input sample — what `roofline.achieved` / `frac` are computed on
CHAIN_BYTES = 2.0 + 64.0 / 1920.0          # full chain: 2 B read per sample + one 64-byte record per 1920-sample frame
FRONT_BYTES = 2.0 + 4.0 + 4.0 + 4 * 4.0    # config 2: int16 in, FIR out, limit out, four correlations out
# what each kernel of the four-pass structure moves per input sample BY DESIGN (intermediates ybuf / hbuf / DCD
# table included) — reported separately as `kernel_design_*`, never as the roofline fraction
FRONT_OWN_BYTES = {"fir_rrc150": 6.0, "limit_track": 8.0, "correlator": 20.0}   # config 2: what each of its three kernels moves itself
DESIGN_BYTES = {"fir_rrc150": 6.0, "dcd": 2.0 + 48.0 / 192.0, "limit_track": 8.0 + 48.0 / 192.0, "demod_seq": 4.0 + 48.0 / 192.0 + 64.0 / 1920.0,
                "compact": 2 * 64.0 / 1920.0, "correlator": 4.0 + 20.0}


def kernel_source_sha16():
    """First 16 hex digits of the sha256 over the device sources of the library (tools/make_valu.py stamps profiles/valu.json with it)."""
    import glob
    import hashlib
    h = hashlib.sha256()
    csrc = os.path.join(ROOT, "m17-cxx-demod_amd", "csrc")
    for f in sorted(glob.glob(os.path.join(csrc, "*.hpp")) + glob.glob(os.path.join(csrc, "*.hip")) +
                    [os.path.join(ROOT, "m17-cxx-demod_amd", "include", "m17cxx", "detail", "core.h")]):
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


def usable_cpus():
    """Threads this process may really use: the affinity mask, cut by the cgroup CPU quota when there is one."""
    n = len(os.sched_getaffinity(0))
    quota = None
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            quota = float(q) / float(per)
    except Exception:
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                quota = q / per
        except Exception:
            pass
    use = n if quota is None else max(1, min(n, int(quota + 0.5)))
    return use, n, quota


def visible_gpus():
    """GPUs a rank of this job would see, counted in a throw-away child process so that THIS process never touches the HIP runtime
    (it goes on to start the ranks).  None when the count cannot be had (no torch, time-out): the ranks then fail by themselves."""
    import subprocess
    try:
        r = subprocess.run([sys.executable, "-c", "import torch; print(torch.cuda.device_count())"], capture_output=True, text=True, timeout=600)
        return int(r.stdout.strip().splitlines()[-1]) if r.returncode == 0 else None
    except Exception:   # noqa: BLE001
        return None


def rank_commands(n, argv, port):
    """The N (argv, environment additions) pairs `bench.py --gpus N` starts when it was not started by a launcher itself:
    one process per GPU, rendezvous on 127.0.0.1 — what `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr
    127.0.0.1 --master-port P bench.py ...` sets for its workers."""
    out = []
    for r in range(n):
        env = {"RANK": str(r), "LOCAL_RANK": str(r), "WORLD_SIZE": str(n), "LOCAL_WORLD_SIZE": str(n), "GROUP_RANK": "0",
               "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port)}
        out.append(([sys.executable, os.path.abspath(__file__)] + [a for a in argv if a != "--dry-launch"], env))
    return out


def launch_ranks(args, argv):
    """`python3 bench.py --gpus N` with N > 1 and no launcher around it (WORLD_SIZE unset): start the N ranks as child processes BEFORE
    anything in this process imports torch or touches the GPU, relay rank 0's JSON line as the LAST (and only) line of stdout, everything
    else any rank prints goes to stderr.  Exit code = the worst child's; a failing rank takes the others down (they would wait in a
    collective for ever) — never a retry, never a smaller world.  Fewer visible GPUs than N: one line on stderr, exit code 2."""
    import socket
    import subprocess
    import threading
    n = args.gpus
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmds = rank_commands(n, argv, port)
    if args.dry_launch:
        print(json.dumps({"dry_launch": [{"rank": r, "argv": c, "env": e} for r, (c, e) in enumerate(cmds)]}))
        return 0
    have = visible_gpus()
    if have is not None and have < n:
        print(f"bench.py: --gpus {n} but {have} GPU(s) visible to this process: not measuring a smaller world", file=sys.stderr)
        return 2
    procs = []
    for r, (c, e) in enumerate(cmds):
        procs.append(subprocess.Popen(c, env={**os.environ, **e}, stdout=subprocess.PIPE, stderr=None, text=True, bufsize=1))
    line0 = []

    def pump(r, pr):   # rank 0: keep the result line back until every rank is done; anything else -> stderr, tagged
        for ln in pr.stdout:
            s_ = ln.rstrip("\n")
            if r == 0 and s_.startswith("{") and '"metric"' in s_:
                line0.append(s_)
            else:
                print(f"[rank {r}] {s_}", file=sys.stderr, flush=True)

    threads = [threading.Thread(target=pump, args=(r, pr), daemon=True) for r, pr in enumerate(procs)]
    for t in threads:
        t.start()
    worst, alive, stopped = 0, set(range(n)), set()
    t_start, t_stop = time.time(), None
    limit = float(getattr(args, "launch_timeout", 3600.0))

    def stop_others(why):      # exactly the processes started above: SIGTERM now, SIGKILL to whoever is still there ten seconds later
        nonlocal t_stop
        print(f"bench.py: {why}; stopping the other ranks", file=sys.stderr)
        for q in alive:
            procs[q].terminate()
            stopped.add(q)
        t_stop = t_stop or time.time()

    while alive:
        for r in sorted(alive):
            code = procs[r].poll()
            if code is None:
                continue
            alive.discard(r)
            if code != 0 and r not in stopped:   # (a rank stopped from here reports the signal: not its own failure)
                worst = max(worst, code if code > 0 else 128 - code)
                stop_others(f"rank {r} exited with code {code}")
        if alive and t_stop is None and time.time() - t_start > limit:   # a rank stuck in a collective or a HIP call: nobody waits for ever
            worst = max(worst, 124)
            stop_others(f"still running after --launch-timeout {limit:.0f} s")
        if alive and t_stop is not None and time.time() - t_stop > 10.0:
            for q in alive:
                procs[q].kill()
        time.sleep(0.2)
    for t in threads:
        t.join(timeout=5)
    if worst == 0 and len(line0) != 1:
        print(f"bench.py: rank 0 printed {len(line0)} result lines", file=sys.stderr)
        worst = 1
    if worst == 0:
        print(line0[0], flush=True)
    return worst


KNAMES = ("fir_rrc150", "dcd", "limit_track", "demod_seq", "compact")


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=6)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--config", type=int, default=3, choices=(2, 3), help="3 = BASELINE configs[2] full chain (headline); 2 = configs[1] FIR + correlator only")
    ap.add_argument("--channels", type=int, default=0, help="channels PER GPU (weak scaling); default 4096 (config 3) / 1024 (config 2)")
    ap.add_argument("--samples", type=int, default=480000, help="samples per channel per step (10 s at 48 kSPS)")
    ap.add_argument("--sigma", type=float, default=600.0, help="AWGN sigma in LSB")
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="target wall time of the cpu_baseline leg (0 = skip)")
    ap.add_argument("--parity-channels", type=int, default=64, help="channels of the last timed step compared with the oracle record for record before anything else (+ 16 in the "
                    "single-stream and bursty checks); the cpu_baseline leg then compares EVERY channel it demodulates")
    ap.add_argument("--config2-steps", type=int, default=3, help="N = 1: steps of the BASELINE configs[1] leg (1024 channels, FIR + correlator, outputs materialised) reported as `config2`; 0 = skip")
    ap.add_argument("--h2d-steps", type=int, default=3, help="steps of the PCIe-inclusive leg (fresh pinned host input every step; 0 = skip)")
    ap.add_argument("--gather", choices=("auto", "cabi", "torch"), default="auto", help="N > 1: m17hip_gather_frames_device (C ABI) or m17hip/dist.py")
    ap.add_argument("--in-flight", type=int, default=2, help="independent batches (contexts) whose steps overlap: the tail of one step (K2/K5 "
                    "alternation, chip half idle) runs beside the front end of the next; 1 = one step after the other")
    ap.add_argument("--prewarm", type=int, default=48, help="untimed steps before the --warmup steps (clock ramp of an idle GPU: about 1.3 s)")
    ap.add_argument("--stagger", type=int, default=1, help="with --in-flight > 1: 1 (default) = queue step k + 1 before waiting for step k (a pipeline, what a host loop that "
                    "feeds several independent batches does); 0 = launch the batches of a group together and wait for them together (the default up to round 5: 1.1-1.8 %% slower now, "
                    "6 %% faster in round 2)")
    ap.add_argument("--tune", action="append", default=[], metavar="KEY=VALUE", help="m17hip_tune knob for experiments (e.g. 10=1: K3 as the four-wave pipeline); reported in config")
    ap.add_argument("--single-stream", type=int, default=1, help="1 = also time the single-stream regime (ONE context, state carried from run to run, no "
                    "reset: a live feed; the front end of run k + 1 is queued through m17hip_demod_front while run k's K2/K5 chain drains) -> value_single_stream")
    ap.add_argument("--one-at-a-time", type=int, default=1, help="1 = also run 2 steps strictly one after the other (no overlap of any kind): the per-launch kernel "
                    "durations the roofline object is computed from (the regime in which HIP events and rocprofv3 agree)")
    ap.add_argument("--one-at-a-time-steps", type=int, default=4)
    ap.add_argument("--stream-order", choices=("run_then_fetch", "fetch_then_run"), default="run_then_fetch", help="single-stream regime: queue run k + 1's state-machine half "
                    "before run k's records are collected (m17hip_frames_select(1); default) or after (the order of rounds 3-5)")
    ap.add_argument("--stream-groups", type=int, default=2, help="contexts the channels of the single-stream regime are split into (independent chains)")
    ap.add_argument("--dry-launch", action="store_true", help="with --gpus N > 1 and no launcher: print the N command lines / environments bench.py would start, and exit")
    ap.add_argument("--duty", type=float, default=1.0, help="experiments: every channel's transmission covers this fraction of a run, loud noise for the rest (1 = BASELINE's "
                    "always-on workload; the `bursty` leg of the default line is 0.2)")
    ap.add_argument("--bursty-steps", type=int, default=6, help="N = 1: steps of the bursty leg (the same 4096 x 480 000, every channel ONE transmission of a fifth of the run, "
                    "loud noise for the rest: the carrier detect is off 80 % of the time) reported as `bursty`; 0 = skip")
    ap.add_argument("--launch-timeout", type=float, default=3600.0, help="with --gpus N > 1 and no launcher: seconds after which the ranks are stopped (exit code 124)")
    ap.add_argument("--force-gather", action="store_true", help="run the N > 1 code path (process group, communicators, gather per step) with WORLD_SIZE = 1")
    return ap.parse_args()


class Bench:
    """State shared by the legs of one bench.py process (one rank): the process group, the contexts of the headline regime, the synthetic
    input, the record buffers.  Every leg is a method; `main` runs them in order and rank 0 prints the line."""

    # ---- set-up ---------------------------------------------------------------------------------------------------------------------
    def __init__(self, args):
        import torch
        import torch.distributed as dist

        self.args, self.torch, self.dist = args, torch, dist
        self.rank = int(os.environ.get("RANK", "0"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        if self.world != args.gpus:   # a launcher that started another number of ranks than the command line names: refuse, do not measure that
            sys.exit(f"bench.py: WORLD_SIZE={self.world} but --gpus {args.gpus}")
        if self.local_rank >= torch.cuda.device_count():
            sys.exit(f"bench.py: rank {self.rank} wants GPU {self.local_rank} but {torch.cuda.device_count()} GPU(s) are visible")
        torch.cuda.set_device(self.local_rank)
        self.dev = torch.device("cuda", self.local_rank)
        self.multi = self.world > 1 or args.force_gather     # the N > 1 code path (with --force-gather also for a world of one rank)
        if self.multi:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29531")
            dist.init_process_group("nccl", device_id=self.dev, rank=self.rank, world_size=self.world)

        import m17hip
        import oracle_lib as ol  # synthetic input parameters + the cpu_baseline / parity checker only
        from m17hip import dist as mdist
        self.m17hip, self.ol, self.mdist = m17hip, ol, mdist

        self.C = args.channels or (4096 if args.config == 3 else 1024)
        self.T = args.samples
        self.ncpu, self.ncpu_affinity, self.cpu_quota = usable_cpus()
        self.make_batches()

    def make_batches(self):
        """Synthetic input (seeded; even channels BERT, odd channels voice-like streams; loud lead-in, AWGN) and the F independent batches of
        C channels, each with its own device slabs and streams.  Generated ON the device, straight into the input slab (m17hip_synth_i16:
        m17-mod framing, RRC shaping, impairments; bit-identical to the test generator ol.generate_batch,
        tests/test_gpu_parity.py::test_device_synthesis_bit_exact): inputs are resident in HBM."""
        args, C, T = self.args, self.C, self.T
        t_gen = time.time()
        self.p = self.ol.gen_params(seed=20260101, kind=-1, n_frames=max(1, int(args.duty * T / 1920) - (6 if args.duty >= 1.0 else 2)), lead_in=3072,
                                    noise_sigma=args.sigma, tail_sigma=args.sigma if args.duty >= 1.0 else 20000.0, lead_sigma=40000.0, total=T)
        self.F = max(1, args.in_flight) if args.config == 3 else 1
        self.ctxs, self.streams, self.tuned = [], [], {}
        for f in range(self.F):
            c_ = self.m17hip.Context(C, T, device=self.local_rank)
            c_.set_channel_base(self.rank * C)     # records carry GLOBAL channel ids: the gathered set is the record set of one big run
            self.apply_tune(c_)
            # (every context works on the library's own main stream, created with its other streams in a fixed role order: include/m17hip.h,
            #  m17hip_get_stream — a host stream handed in per context made the layout depend on the process's history, NOTES 6.6)
            self.ctxs.append(c_)
        self.ctx = self.ctxs[0]
        for c_ in self.ctxs:
            c_.synth(self.p, C, T, chan0=self.rank * C)
        self.x = self.ctx.download() if self.rank == 0 else None   # host copy for the parity checks, the cpu_baseline and the PCIe-inclusive leg (rank 0)
        self.t_gen = time.time() - t_gen
        self.rec_cap_local = C * (2 * (T // 1920 + 2) + 4)

    def apply_tune(self, c_):
        for kv in self.args.tune:
            k_, v_ = kv.split("=")
            c_.tune(int(k_), int(v_))
            self.tuned[k_] = int(v_)

    def sync(self):
        self.torch.cuda.synchronize()
        if self.multi:
            self.dist.barrier()
            self.torch.cuda.synchronize()

    def max_over_ranks(self, v):
        if not self.multi:
            return v
        t = self.torch.tensor([v], dtype=self.torch.float64, device=self.dev)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())

    @staticmethod
    def kernel_times(cs, names, steps):
        out = {}
        for name in names:
            ms = n = 0   # a run is processed in segments: several launches of each kernel per step; summed over the contexts
            for c_ in cs:
                ms_, n_ = c_.timing_get(name)
                ms, n = ms + ms_, n + n_
            out[name] = {"ms_avg": (ms / n) if n else None, "launches": n, "ms_per_step": ms / steps}
        return out

    def setup_gather(self):
        """The one exchange of the path (N > 1): frame records of every shard to rank 0 over RCCL.  One communicator per batch in flight (gathers
        of different batches are issued from different streams; a communicator takes one call at a time).  Every step of the set-up is agreed
        on collectively: a rank that cannot bind RCCL through the library, or whose ncclCommInitRank fails, takes every rank to the
        torch.distributed gather of m17hip/dist.py instead of leaving them in a collective."""
        args, torch, dist, m17hip, F = self.args, self.torch, self.dist, self.m17hip, self.F
        self.gather_kind, self.comms = "none (1 GPU)", []
        self.rec_bufs = [torch.zeros(self.rec_cap_local * (self.world if self.rank == 0 else 1) * 64, dtype=torch.uint8, device=self.dev) for _ in range(F)]
        self.last = {}
        if not self.multi:
            return

        def all_ok(flag):
            t = torch.tensor([1 if flag else 0], dtype=torch.int32, device=self.dev)
            dist.all_reduce(t, op=dist.ReduceOp.MIN)
            return bool(t.item())

        cabi = args.gather in ("auto", "cabi")
        my_ids = None
        if cabi:   # 1. every rank must be able to bind RCCL through the library (probe, not collective) ...
            try:
                my_ids = [m17hip.comm_get_id() for _ in range(F)]
            except Exception as e:   # noqa: BLE001
                print(f"rank {self.rank}: C-ABI RCCL binding unavailable ({e})", file=sys.stderr)
                my_ids = None
            cabi = all_ok(my_ids is not None)
        if cabi:   # 2. ... then rank 0's ids go round and every rank joins (collective); the outcome is agreed on again
            idt = torch.zeros(F * m17hip.COMM_ID_BYTES, dtype=torch.uint8, device=self.dev)
            if self.rank == 0:
                idt.copy_(torch.frombuffer(bytearray(b"".join(my_ids)), dtype=torch.uint8))
            dist.broadcast(idt, 0)
            ids = bytes(idt.cpu().numpy().tobytes())
            try:
                for f in range(F):
                    self.comms.append(m17hip.Comm(self.ctxs[f], ids[f * m17hip.COMM_ID_BYTES:(f + 1) * m17hip.COMM_ID_BYTES], self.rank, self.world))
            except Exception as e:   # noqa: BLE001
                print(f"rank {self.rank}: m17hip_comm_create failed ({e})", file=sys.stderr)
            cabi = all_ok(len(self.comms) == F)
            if not cabi:
                for m_ in self.comms:
                    m_.close()
                self.comms = []
        if not cabi and args.gather == "cabi":
            raise SystemExit("--gather cabi: the RCCL communicators could not be created through libm17hip.so on every rank")
        self.gather_kind = "m17hip_gather_frames_device: counts all-gathered, records ncclSend/ncclRecv to rank 0" if self.comms else \
            "torch fallback (m17hip/dist.py: padded torch.distributed all_gather_into_tensor over RCCL)"

    # ---- the headline regime: F independent batches in flight, fresh demodulators every step ------------------------------------------------
    def launch(self, k):            # one step = C fresh demodulators over C x T samples: queued, not waited for
        c_ = self.ctxs[k % self.F]
        c_.reset()
        c_.run()

    def finish(self, k):            # ... its records compacted on the device (N > 1: gathered to rank 0): waits for that step only
        F, rank, world = self.F, self.rank, self.world
        c_, buf = self.ctxs[k % F], self.rec_bufs[k % F]
        self.last["buf"] = buf
        if not self.multi:
            return c_.frames_compact_device(buf.data_ptr(), self.rec_cap_local)
        if self.comms:
            total, counts = c_.gather_frames_device(self.comms[k % F], buf.data_ptr() if rank == 0 else 0, self.rec_cap_local * world if rank == 0 else 0, root=0)
            self.last["counts"] = counts
            return int(total)
        n = c_.frames_compact_device(buf.data_ptr(), self.rec_cap_local)
        allrecs, counts = self.mdist.gather_records(buf[: self.rec_cap_local * 64], n)
        self.last["counts"], self.last["allrecs"] = counts, allrecs
        return int(allrecs.shape[0])

    def run_steps(self, n_steps):   # with F > 1 step k + 1 is queued before step k is waited for: its front end fills the gaps of k's tail
        F, total = self.F, 0
        if F == 1:
            for k in range(n_steps):
                self.launch(k)
                total = self.finish(k)
            return total
        if self.args.stagger:   # step k + 1 queued before step k is waited for: a pipeline whose batches drift half a step apart
            self.launch(0)
            for k in range(1, n_steps):
                self.launch(k)
                total = self.finish(k - 1)
            return self.finish(n_steps - 1)
        for k0 in range(0, n_steps, F):   # groups of F steps queued together and waited for together: the batches go through the same
            ks = range(k0, min(k0 + F, n_steps))   # phases side by side (round 2: 6 % faster than half a step apart; round 5: 1.1-1.8 % slower, NOTES 5.12)
            for k in ks:
                self.launch(k)
            for k in ks:
                total = self.finish(k)
        return total

    def leg_headline(self):
        """W warm-up steps, then EXACTLY K timed steps between barrier + synchronize brackets; max over ranks.  Then, outside the timed region:
        the records of the last timed step (rank 0), the parity spot check against the oracle, and (N > 1) the check that the gathered set is
        one (channel, seq)-ordered, duplicate-free set."""
        args, np_, m17hip, C, T, F = self.args, np, self.m17hip, self.C, self.T, self.F
        # Untimed pre-warm before the W warm-up steps: a GPU that has been idle needs about a second of load before its clocks and the
        # two batches' interleaving settle (measured: the first process on a fresh box ran 29.4 ms/step with 5 warm-up steps, 26.4 with 60)
        # (a step count, not a time: every rank has to make the same number of gather calls)
        if args.prewarm > 0:
            self.run_steps(args.prewarm)
            self.sync()
        if args.warmup:
            self.run_steps(args.warmup)
        for c_ in self.ctxs:
            c_.timing(True)
            c_.timing_reset()
        self.sync()
        t0 = time.perf_counter()
        wall0 = time.time()
        self.total_frames = self.run_steps(args.steps)
        self.sync()
        dt = time.perf_counter() - t0
        if self.rank == 0:   # (wall-clock bracket of the timed region on stderr: tools/clock_probe.hip's series is aligned with it)
            print(f"timed region: epoch {wall0:.3f} .. {wall0 + dt:.3f}", file=sys.stderr)
        self.dt = self.max_over_ranks(dt)
        for c_ in self.ctxs:
            c_.timing(False)
        self.kern = self.kernel_times(self.ctxs, KNAMES, args.steps)

        self.parity = self.good = self.gathered_ok = self.recs = None
        if self.rank != 0:
            return
        if not self.multi:
            recs = self.ctxs[(args.steps - 1) % F].frames()
        elif self.comms:
            recs = np_.frombuffer(self.last["buf"][: self.total_frames * 64].cpu().numpy().tobytes(), dtype=m17hip.FRAME_REC)
        else:
            recs = np_.frombuffer(self.last["allrecs"].cpu().numpy().tobytes(), dtype=m17hip.FRAME_REC)
        if self.multi:
            key = (recs["channel"].astype(np_.int64) << 32) | recs["seq"].astype(np_.int64)
            counts = np_.asarray(self.last["counts"], dtype=np_.int64)
            self.gathered_ok = bool(recs.size == int(counts.sum()) and (np_.diff(key) > 0).all() and int(recs["channel"].max()) < C * self.world
                                    and np_.array_equal(np_.bincount(recs["channel"] // C, minlength=self.world), counts))
            assert self.gathered_ok, "gathered frame records are not one (channel, seq)-ordered, duplicate-free set"
        if args.parity_channels > 0:
            k = min(args.parity_channels, C)
            got = recs[recs["channel"] < k]
            exp_recs, exp_counts, _ = self.ol.demod_batch(self.x[:k], cap=2 * (T // 1920 + 2) + 4, threads=min(k, self.ncpu))
            exp = np_.concatenate([exp_recs[c, : exp_counts[c]] for c in range(k)])
            self.parity = bool(got.tobytes() == exp.tobytes())
        self.good = int(((recs["cost"] >= 0) & (recs["cost"] < 10) & (recs["frame_type"] != 1)).sum())
        self.recs = recs

    # ---- the single-stream regime ------------------------------------------------------------------------------------------------------------
    def leg_single_stream(self):
        """ONE set of channels, the SAME channels run after run, state carried (no reset) — a live feed, and what the literal configs[3] split (one
        4096-channel batch per GPU) gives.  Two resident input slabs alternate (m17hip_input_alternate: no copy); the front end of run k + 1 is
        queued (m17hip_demod_front) beside the K2/K5 chain of run k; then K2/K5 of run k + 1 follow.  Same barrier / synchronize bracket, same K
        steps.  The C channels as G groups (contexts of C / G channels, default 2): independent channels make independent chains, and two
        half-size K2/K5 chains side by side keep the chip fuller than one (measured: 38.3 -> 33.4 ms per step)."""
        args, np_, m17hip, C, T, rank, world = self.args, np, self.m17hip, self.C, self.T, self.rank, self.world
        if not args.single_stream:
            return None
        G = max(1, args.stream_groups)
        Cg = C // G
        assert Cg * G == C, "--stream-groups must divide the channel count"
        sctx, sbuf, sstreams = [], [], []
        cap_g = Cg * (2 * (T // 1920 + 2) + 4)
        for g in range(G):
            c_ = m17hip.Context(Cg, T, device=self.local_rank)
            c_.set_channel_base(rank * C + g * Cg)
            self.apply_tune(c_)
            c_.synth(self.p, Cg, T, chan0=rank * C + g * Cg)
            c_.tune(16, 1)
            c_.synth(self.p, Cg, T, chan0=rank * C + g * Cg)   # the same synthetic slab into the context's second input slab (staged)
            c_.tune(16, 0)
            c_.reset()
            c_.run()                                  # run 0 consumes it; from here on the two slabs alternate without copies
            sctx.append(c_)
            sbuf.append(self.torch.zeros(cap_g * (world if rank == 0 else 1) * 64, dtype=self.torch.uint8, device=self.dev))
        new_order = args.stream_order == "run_then_fetch"

        def sfinish(g):   # records of the selected run of group g (compaction; N > 1: gather to rank 0) — waits for that run only
            c_, buf = sctx[g], sbuf[g]
            if not self.multi:
                return c_.frames_compact_device(buf.data_ptr(), cap_g)
            if self.comms:
                total, _ = c_.gather_frames_device(self.comms[g % len(self.comms)], buf.data_ptr() if rank == 0 else 0, cap_g * world if rank == 0 else 0, root=0)
                return int(total)
            n = c_.frames_compact_device(buf.data_ptr(), cap_g)
            allrecs, _ = self.mdist.gather_records(buf[: cap_g * 64], n)
            return int(allrecs.shape[0])

        def stream_steps(n_steps):   # the call sequence of a live feed (include/m17hip.h, m17hip_demod_front)
            tot = 0
            for k in range(n_steps):
                for c_ in sctx:
                    c_.input_alternate(Cg, T)
                    c_.front()                       # K1 / K3 of the next run: queued now, beside the tail of the run in flight
                tot = 0
                if new_order:
                    for c_ in sctx:
                        c_.run()                     # K2 / K5 chain of the next run: queued behind the chain in flight — nothing of it waits for the
                    for g, c_ in enumerate(sctx):    # payload work of the run before (deferred decode, compaction) or for the host
                        c_.frames_select(1)          # the records of the run BEFORE the one just queued
                        tot += sfinish(g)
                        c_.frames_select(0)
                else:                                # (up to round 5: the records collected first, then the next chain queued)
                    for g, c_ in enumerate(sctx):
                        tot += sfinish(g)
                        c_.run()
            return tot

        def timed_pass():
            self.sync()
            ts = time.perf_counter()
            stream_steps(args.steps)
            for g in range(G):
                sfinish(g)
            self.sync()
            return self.max_over_ranks(time.perf_counter() - ts)

        # Two timed passes of K steps: the first as a deployment runs it, the second with the library's per-kernel HIP events on for `kernel_ms` — the
        # events are not free on a chain of dependent launches (0.45-0.65 ms of a 21.7 ms step, tools/stream_only.py TIMING=1), and the `value` of this
        # leg is what the feed costs, not what measuring it costs.  (The headline leg keeps its events inside its timed region, as the contract asks.)
        # (the legs before this one end with seconds of host work — the oracle's parity run: the GPU's clocks have dropped; as in front of the headline leg)
        stream_steps(max(2, args.warmup) + args.prewarm // 3)
        dts = timed_pass()
        for c_ in sctx:
            c_.timing(True); c_.timing_reset()
        dts_timers = timed_pass()
        for c_ in sctx:
            c_.timing(False)
        skern = self.kernel_times(sctx, KNAMES, args.steps)
        sparity = None
        if rank == 0 and args.parity_channels > 0 and not self.multi:   # three pipelined runs from a fresh start == the oracle over slab x 3
            k = min(16 // G if G <= 16 else 1, args.parity_channels, Cg)
            parts = []
            for g, c_ in enumerate(sctx):
                c_.reset()
                for r_ in range(3):
                    c_.input_alternate(Cg, T)
                    if r_:
                        c_.front()
                        if new_order:   # (the order the timed loop uses)
                            c_.run()
                            c_.frames_select(1)
                        q = c_.frames()
                        parts.append(q[q["channel"] < g * Cg + k])
                        if new_order:
                            c_.frames_select(0)
                            continue
                    c_.run()
                q = c_.frames()
                parts.append(q[q["channel"] < g * Cg + k])
            got = np_.concatenate(parts)
            got = got[np_.lexsort((got["seq"], got["channel"]))]
            rows = np_.concatenate([np_.arange(g * Cg, g * Cg + k) for g in range(G)])
            exp_recs, exp_counts, _ = self.ol.demod_batch(np_.tile(self.x[rows], (1, 3)), cap=2 * (3 * T // 1920 + 2) + 4, threads=min(len(rows), self.ncpu))
            exp = np_.concatenate([exp_recs[i, : exp_counts[i]] for i in range(len(rows))])
            exp["channel"] = np_.concatenate([np_.full(int(exp_counts[i]), rows[i], dtype=np_.uint32) for i in range(len(rows))])
            sparity = bool(got.tobytes() == exp.tobytes())
        single = {"value": round(C * T * world * args.steps / dts / 1e6, 2), "ms_per_step": round(dts / args.steps * 1e3, 3), "steps": args.steps,
                  "ms_per_step_with_kernel_timers": round(dts_timers / args.steps * 1e3, 3), "channel_groups": G, "order": args.stream_order,
                  "what": "the same %d channels per GPU continued run after run (state carried, no reset) as %d contexts of %d channels, two resident "
                          "input slabs alternating, front end of run k + 1 queued by m17hip_demod_front beside run k's K2/K5 chain, records of every "
                          "run compacted (those of run k after run k + 1's chain was queued: m17hip_frames_select)" % (C, G, Cg) + (" and gathered" if self.multi else ""),
                  "kernel_ms": {k_: round(v["ms_per_step"], 4) for k_, v in skern.items()},
                  "parity_vs_oracle_3_runs_first_channels": sparity}
        for c_ in sctx:
            c_.close()
        return single

    # ---- one step strictly after the other -----------------------------------------------------------------------------------------------------
    def leg_one_at_a_time(self):
        """The same kernels with ONE step strictly after the other (outside the timed regions): per-launch durations that are not stretched by
        another batch's or another run's kernels.  This is the regime in which the HIP-event brackets and rocprofv3's kernel durations agree
        (profiles/r*_one_at_a_time_kernel_trace_stats.md is this very loop), and the one the `roofline` object is computed from."""
        args, ctx = self.args, self.ctx
        if not args.one_at_a_time:
            return None, None, None
        rec_buf = self.rec_bufs[0]
        for _ in range(1 + args.prewarm // 12):   # (untimed: the library picks the carrier-detect kernel's form, the segment ramp and the redo policy from whether runs
            ctx.reset(); ctx.run(); ctx.frames_count()   #  overlapped lately — the legs before this one did; and a GPU that has idled needs load before its clocks settle)
        def timed_pass():
            ctx.reset()
            self.torch.cuda.synchronize()
            ts = time.perf_counter()
            for _ in range(args.one_at_a_time_steps):
                ctx.reset(); ctx.run(); ctx.frames_compact_device(rec_buf.data_ptr(), self.rec_cap_local)
            self.torch.cuda.synchronize()
            return (time.perf_counter() - ts) / args.one_at_a_time_steps * 1e3

        seq_ms_plain = timed_pass()               # what a step costs ...
        ctx.timing(True); ctx.timing_reset()
        seq_ms = timed_pass()                     # ... and with every kernel bracketed by HIP events: the durations `roofline` is computed from
        seq_kern = self.kernel_times([ctx], KNAMES[:4], args.one_at_a_time_steps)
        ctx.timing(False)
        return seq_kern, seq_ms, seq_ms_plain

    # ---- bursty input --------------------------------------------------------------------------------------------------------------------------
    def leg_bursty(self):
        """N = 1, outside the timed regions: what the chain does when the reference's carrier detect is OFF most of the time.  The reference runs
        neither the matched filter nor the correlator while it is (M17Demodulator.h:675-689) and gets ~10 x cheaper there; here the library turns
        its gate-aware front end on by itself on such input (m17hip_tune key 26 = -1: K1 skips what the carrier cannot be on for, forecast from
        K5's true gate state), K3 still sees every sample.  Same regime as `value` (the batches in flight, fresh demodulators every step);
        bit-exactness on it checked."""
        args, np_, C, T, F = self.args, np, self.C, self.T, self.F
        if self.multi or args.bursty_steps <= 0:
            return None
        nb = max(1, int(0.2 * T / 1920) - 2)
        pb = self.ol.gen_params(seed=20260102, kind=-1, n_frames=nb, lead_in=3072, noise_sigma=args.sigma, tail_sigma=20000.0, lead_sigma=40000.0, total=T)
        for c_ in self.ctxs:
            c_.synth(pb, C, T, chan0=self.rank * C)
        self.run_steps(2 * F)
        for c_ in self.ctxs:
            c_.timing(True); c_.timing_reset()
        self.sync()
        tb = time.perf_counter()
        nfr = self.run_steps(args.bursty_steps)
        self.sync()
        dtb = (time.perf_counter() - tb) / args.bursty_steps
        for c_ in self.ctxs:
            c_.timing(False)
        bkern = self.kernel_times(self.ctxs, KNAMES, args.bursty_steps)
        bpar = None
        if args.parity_channels > 0:
            k = min(16, C)
            last_ctx = self.ctxs[(args.bursty_steps - 1) % F]
            xb = last_ctx.download()[:k]
            got = last_ctx.frames()
            got = got[got["channel"] < k]
            er, ec, _ = self.ol.demod_batch(xb, cap=2 * (T // 1920 + 2) + 4, threads=min(k, self.ncpu))
            bpar = bool(got.tobytes() == np_.concatenate([er[c, : ec[c]] for c in range(k)]).tobytes())
        bursty = {"value": round(C * T / dtb / 1e6, 2), "unit": "Msamples/s", "ms_per_step": round(dtb * 1e3, 3), "steps": args.bursty_steps,
                  "ratio_to_always_on": round((C * T / dtb / 1e6) / (C * T * args.steps / self.dt / 1e6), 3), "frames_decoded_per_step": int(nfr),
                  "kernel_ms": {k_: round(v["ms_per_step"], 3) for k_, v in bkern.items()},
                  "what": "every channel one transmission of %d frames (a fifth of the run) behind a loud lead-in, loud noise (sigma 20000) for the rest; "
                          "%d batches in flight as for `value`" % (nb, F), "parity_vs_oracle_first_channels": bpar}
        for c_ in self.ctxs:   # (the legs below run on the always-on input again)
            c_.synth(self.p, C, T, chan0=self.rank * C)
        return bursty

    # ---- PCIe-inclusive rate -------------------------------------------------------------------------------------------------------------------
    def leg_h2d(self):
        """N = 1: every step gets fresh input from pinned host memory, upload of step k + 1 overlapped."""
        args, torch, ctx, C, T = self.args, self.torch, self.ctx, self.C, self.T
        if self.multi or args.h2d_steps <= 0:
            return None
        rec_buf = self.rec_bufs[0]
        a = torch.from_numpy(self.x).pin_memory()
        b = torch.from_numpy(self.x.copy()).pin_memory()
        ctx.upload_async(a.data_ptr(), C, T)
        step_h = None
        for k in range(args.h2d_steps + 1):
            if k == 1:
                torch.cuda.synchronize(); step_h = time.perf_counter()
            ctx.reset(); ctx.run()
            ctx.upload_async((b if k % 2 == 0 else a).data_ptr(), C, T)
            ctx.frames_compact_device(rec_buf.data_ptr(), self.rec_cap_local)
        torch.cuda.synchronize()
        th = (time.perf_counter() - step_h) / args.h2d_steps
        ctx.upload_wait()
        return {"value_with_h2d": round(C * T / th / 1e6, 2), "ms_per_step": round(th * 1e3, 3), "steps": args.h2d_steps,
                "what": "fresh pinned host slab every step through m17hip_upload_i16_async (second device slab, copy stream), overlapped with the run before it",
                "input_GB_per_step": round(C * T * 2 / 1e9, 3)}

    # ---- BASELINE configs[1] beside the headline ------------------------------------------------------------------------------------------------
    def leg_config2(self):
        """N = 1: 1024 channels, FIR + correlator only, every output materialised in HBM, ONE call (m17hip_fir_correlator: the matched filter, the
        limit filter and the correlations of a run pipelined in time).  With its own roofline object, a bit-exact check of its outputs and a CPU
        figure — the reference's own BaseFirFilter + Correlator where oracle/_ref travelled."""
        import ctypes as Ct
        args, np_, m17hip, ol, T = self.args, np, self.m17hip, self.ol, self.T
        if self.multi or args.config2_steps <= 0:
            return None
        for c_ in self.ctxs[1:]:
            c_.close()
        C2 = 1024
        c2 = m17hip.Context(C2, T, device=self.local_rank)
        c2.synth(self.p, C2, T, chan0=0)
        c2.fir_correlator(fetch=False)
        c2.timing(True); c2.timing_reset()
        self.torch.cuda.synchronize()
        t2 = time.perf_counter()
        for _ in range(args.config2_steps):
            c2.fir_correlator(fetch=False)
        self.torch.cuda.synchronize()
        dt2 = (time.perf_counter() - t2) / args.config2_steps
        c2.timing(False)
        k2 = {}
        for name in ("fir_rrc150", "limit_track", "correlator"):   # (matched filter, the limit filter's chain, the four correlations: each on its own stream)
            ms, n = c2.timing_get(name)
            k2[name] = {"ms_avg": (ms / n) if n else None, "ms_per_step": ms / args.config2_steps}
        dom2 = max(k2, key=lambda k: k2[k]["ms_per_step"])
        ach2 = FRONT_BYTES * C2 * T / (k2[dom2]["ms_per_step"] / 1e3) / 1e9   # (the call runs its kernels in pieces in time: bytes of a step / that kernel's launches of a step = bytes per launch / average launch)
        config2 = {"workload": "configs[1]: 1024 independent 48 kSPS channels x %d samples, FIR + Correlator only, outputs (FIR out, limit, 4 correlations) left in HBM" % T,
                   "value": round(C2 * T / dt2 / 1e6, 2), "unit": "Msamples/s", "ms_per_step": round(dt2 * 1e3, 3), "steps": args.config2_steps,
                   "roofline": {"bound": "hbm", "kernel": dom2, "achieved": round(ach2, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(ach2 / HBM_PEAK_GBS, 5),
                                "traffic": None, "alg_bytes_per_sample": FRONT_BYTES, "kernel_ms": {k: round(v["ms_per_step"], 4) for k, v in k2.items()},
                                "chain_achieved_GBs": round(FRONT_BYTES * C2 * T / dt2 / 1e9, 2), "chain_frac": round(FRONT_BYTES * C2 * T / dt2 / 1e9 / HBM_PEAK_GBS, 6),
                                # each kernel on the bytes IT moves (K1: int16 in + f32 out; the limit chain: f32 in + f32 out; the correlations: f32 in + 4 x f32 out)
                                "kernel_own_bytes_per_sample": dict(FRONT_OWN_BYTES),
                                "kernel_own_frac": {k: round(ob * C2 * T / (k2[k]["ms_per_step"] / 1e3) / 1e9 / HBM_PEAK_GBS, 5) for k, ob in FRONT_OWN_BYTES.items() if k2[k]["ms_per_step"]}}}
        c2.close()
        if args.parity_channels > 0:   # soft outputs: the north star asks 1e-5 relative; they are bit-exact (4 channels x 48 000 samples, one call)
            k, n = 4, min(T, 48000)
            c3 = m17hip.Context(k, n, device=self.local_rank)
            c3.upload(self.x[:k, :n])
            y, lim, corr = c3.fir_correlator()
            c3.close()
            ok2, against = True, "oracle"
            for c in range(k):
                ye = ol.fir_i16(self.x[c, :n]); le, ce = ol.correlator(ye)
                ok2 = ok2 and np_.array_equal(y[c], ye) and np_.array_equal(lim[c], le) and np_.array_equal(corr[:, c, :], ce)
                if ol.ref() is not None:   # ... and against the reference's own classes where their build travelled
                    against = "oracle and the reference's BaseFirFilter / Correlator (oracle/_ref)"
                    yr = np_.zeros(n, np_.float32)
                    ol.ref().ref_fir_f32(ol._p(ol.taps()), ol._p(ol.scale(self.x[c, :n])), Ct.c_size_t(n), ol._p(yr))
                    lr, cr = ol.correlator(yr, lib=ol.ref(), prefix="ref_")
                    ok2 = ok2 and np_.array_equal(y[c], yr) and np_.array_equal(lim[c], lr) and np_.array_equal(corr[:, c, :], cr)
            config2["outputs_bit_exact_first_channels"] = bool(ok2)
            config2["outputs_checked_against"] = against
        if args.cpu_seconds > 0:
            a2 = argparse.Namespace(cpu_seconds=min(args.cpu_seconds, 1.0))
            config2["cpu_baseline"] = cpu_baseline(a2, ol, self.x[:C2], C2, T, self.ncpu, self.ncpu_affinity, self.cpu_quota, chain=False)
        return config2

    # ---- the roofline objects --------------------------------------------------------------------------------------------------------------------
    def roofline(self, value, seq_kern, seq_ms, single, seq_ms_plain=None):
        args, C, T, world = self.args, self.C, self.T, self.world
        dt, kern = self.dt, self.kern
        src = seq_kern if seq_kern else kern
        src_steps = args.one_at_a_time_steps if seq_kern else args.steps
        # (the limit filter's launches are of three kinds — a run's first segment, replays ahead, redos of a few channels: 4 us ... 1.7 ms — so their average is
        #  no launch's duration, and all but the first run beside K5: K2 is reported in kernel_ms but is not a candidate for the kernel `roofline` is about)
        dom = max((k for k in src if src[k]["ms_avg"] and k != "limit_track"), key=lambda k: src[k]["ms_per_step"])
        dom_s = src[dom]["ms_avg"] / 1e3                                    # average duration of ONE launch of the dominant kernel
        launches_per_step = src[dom]["launches"] / src_steps
        units = C * T / launches_per_step                                    # samples one launch processes
        achieved = CHAIN_BYTES * units / dom_s / 1e9                         # SURVEY §8(d) algorithmic bytes of one launch / its duration
        traffic = None  # HBM bytes per launch of the dominant kernel from the PMC passes of tools/profile_round.sh (profiles/)
        tpath = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tpath):
            tj = json.load(open(tpath))
            tk = tj.get("kernels", {}).get(dom)
            if tj.get("channels") == C and tj.get("samples") == T and tk and tk.get("launches_per_step", tj.get("launches_per_step")) == launches_per_step:
                traffic = tk["hbm_bytes_per_launch"]   # per launch (= per segment), like `achieved`
        roofline = {"bound": "hbm", "kernel": dom, "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic,
                    "regime": ("one step strictly after the other, %d steps outside the timed regions (launch durations as rocprofv3 reports them)" % src_steps)
                    if seq_kern else "the timed region",
                    "alg_bytes_per_sample": round(CHAIN_BYTES, 4), "launches_per_step": launches_per_step,
                    "kernel_ms_per_launch": {k: (round(v["ms_avg"], 4) if v["ms_avg"] else None) for k, v in src.items()},
                    "kernel_ms": {k: round(v["ms_per_step"], 4) for k, v in src.items()},
                    # (one step after the other: a pass WITHOUT the per-kernel events — what a step costs — and the pass the launch durations were taken in)
                    "ms_per_step": round(seq_ms_plain, 3) if seq_kern else round(dt / args.steps * 1e3, 3),
                    "ms_per_step_with_kernel_timers": round(seq_ms, 3) if seq_kern else None,
                    "kernel_design_bytes_per_sample": round(DESIGN_BYTES[dom], 4),
                    "kernel_design_GBs": round(DESIGN_BYTES[dom] * units / dom_s / 1e9, 2),
                    "chain_achieved_GBs": round(CHAIN_BYTES * C * T * args.steps / dt / 1e9, 2),
                    "chain_frac": round(CHAIN_BYTES * C * T * args.steps / dt / 1e9 / HBM_PEAK_GBS, 6),
                    "timed_region_kernel_ms": {k: round(v["ms_per_step"], 4) for k, v in kern.items()},
                    "timed_region_note": "HIP-event brackets of launches that share the chip with another batch's kernels include the time a launch waits for free CUs"}
        # ---- the BINDING roofline: fp32 VALU.  SURVEY §8(d): the exact-order matched filter is 149 multiplies + 149 additions per sample, no FMA allowed:
        #      298 flop per sample against the non-FMA fp32 vector peak (157.3 / 2 = 78.6 Tflop/s).  Instruction counts and clocks come from the committed PMC
        #      pass (profiles/valu.json <- tools/profile_round.sh + tools/make_valu.py); `issue_util` = VALU instructions of a step x 4 cycles (a packed
        #      operation's issue time; an unpacked one takes 2: NOTES 5.1) / (SIMDs x cycles of the step at the clock measured in the mix).
        valu = {"flop_per_sample": 298, "achieved_tflops": round(value * 1e6 / world * 298 / 1e12, 3), "peak_nonfma_tflops": 78.6,
                "frac": round(value * 1e6 / world * 298 / 1e12 / 78.6, 4), "valu_insts_per_step": None, "issue_util": None}
        vpath = os.path.join(ROOT, "profiles", "valu.json")
        if os.path.exists(vpath):
            vj = json.load(open(vpath))
            stale = vj.get("kernel_source_sha16") not in (None, kernel_source_sha16())   # counted on other kernels than the ones that just ran: not this build's figure
            if stale:
                valu["note"] = "profiles/valu.json was made from other kernel sources than this build's: instruction counts and issue_util left out"
            if vj.get("channels") == C and vj.get("samples") == T and not stale:
                mix = vj.get("clock_in_mix") if isinstance(vj.get("clock_in_mix"), dict) else {}
                clk_mhz = mix.get("busy_mean_mhz") or 2100.0
                n_simd = 4 * int(self.torch.cuda.get_device_properties(self.dev).multi_processor_count)
                cycles = dt / args.steps * clk_mhz * 1e6
                valu.update({"valu_insts_per_step": int(vj["valu_insts_per_step"]), "issue_util": round(vj["valu_insts_per_step"] * 4 / (n_simd * cycles), 4),
                             "clock_mhz_in_mix": round(clk_mhz, 1), "simds": n_simd,
                             "valu_insts_per_step_by_kernel": {k: int(v["valu_insts_per_step"]) for k, v in vj["kernels"].items()},
                             "clock_ghz_alone_by_kernel": {k: round(v["clock_ghz_alone"], 3) for k, v in vj["kernels"].items()},
                             "source": "profiles/valu.json (rocprofv3 PMC passes of tools/profile_round.sh: SQ_INSTS_VALU, GRBM_GUI_ACTIVE; tools/clock_probe.hip beside the two-batch regime)"})
        roofline["valu"] = valu
        if single:
            roofline["single_stream_chain_achieved_GBs"] = round(CHAIN_BYTES * single["value"] * 1e6 / world / 1e9, 2)
            roofline["single_stream_chain_frac"] = round(CHAIN_BYTES * single["value"] * 1e6 / world / 1e9 / HBM_PEAK_GBS, 6)
        return roofline

    def shutdown(self):
        for m_ in self.comms:
            m_.close()
        if self.multi:
            self.dist.destroy_process_group()


def main():
    args = parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:   # no launcher around us: be the launcher (before torch / the GPU are touched)
        sys.exit(launch_ranks(args, sys.argv[1:]))
    if args.dry_launch:
        sys.exit("--dry-launch: only with --gpus N > 1 and WORLD_SIZE unset")
    # the streams of a context (main, K1, K3, K2-ahead, copy) and of different contexts must not share a hardware queue: a kernel queued
    # behind another stream's event wait in the same queue waits with it (the runtime's default is 4 queues)
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")

    B = Bench(args)
    if args.config == 2:
        return bench_front(args, B.ctx, B.ol, B.x, B.C, B.T, B.rank, B.world, B.dev, B.sync, B.ncpu, B.ncpu_affinity, B.cpu_quota, B.t_gen)
    B.setup_gather()
    B.leg_headline()                          # `value`
    single = B.leg_single_stream()            # `value_single_stream`
    seq_kern, seq_ms, seq_ms_plain = B.leg_one_at_a_time()  # `roofline`'s launch durations
    bursty = B.leg_bursty()
    if B.rank != 0:
        B.shutdown()
        return

    C, T, F, world = B.C, B.T, B.F, B.world
    value = C * T * world * args.steps / B.dt / 1e6
    roofline = B.roofline(value, seq_kern, seq_ms, single, seq_ms_plain)
    h2d = B.leg_h2d()
    config2 = B.leg_config2()

    # the oracle over (as many as fit the time of) the channels of the step: its throughput is the cpu_baseline, its RECORDS are compared, all of
    # them, with the last timed step's — a mismatch is printed in the line and is the exit code
    kept, parity_all, parity_all_channels = {}, None, 0
    cpu = cpu_baseline(args, B.ol, B.x, C, T, B.ncpu, B.ncpu_affinity, B.cpu_quota, chain=True, keep=kept) if (args.cpu_seconds > 0 and not B.multi) else None
    if kept:
        parity_all_channels = int(kept["channels"])
        got = B.recs[B.recs["channel"] < parity_all_channels]
        parity_all = bool(got.tobytes() == kept["recs"].tobytes())
        if not parity_all:
            nbad = len(set(np.unique(got["channel"]).tolist()) ^ set(np.unique(kept["recs"]["channel"]).tolist()))
            print(f"bench.py: PARITY FAILURE: the GPU's records of the last timed step differ from the oracle's on the first {parity_all_channels} channels "
                  f"({got.size} against {kept['recs'].size} records, {nbad} channels present on one side only)", file=sys.stderr)

    out = {
        "metric": "Msamples/s demodulated (48 kSPS 4-FSK in -> decoded frames)",
        "value": round(value, 2), "unit": "Msamples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(B.dt / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic (generated on the device)",
        "value_single_stream": single["value"] if single else None, "ms_per_step_single_stream": single["ms_per_step"] if single else None,
        "config": {"duty": args.duty, "workload": "configs[2]: full demod chain incl. Viterbi/Trellis, %d channels x %d samples per step and GPU, bit-exact frame check; "
                               "`value`: %d INDEPENDENT batches of that size resident and in flight per GPU (fresh demodulators every step); "
                               "`value_single_stream`: one batch, the same channels continued run after run" % (C, T, F),
                   "channels_per_gpu": C, "channels_resident_per_gpu": C * F, "samples_per_channel": T, "awgn_sigma_lsb": args.sigma, "frames_decoded_per_step": B.total_frames,
                   "frames_cost_lt_10": B.good, "parity_vs_oracle_first_channels": B.parity, "parity_first_channels": args.parity_channels,
                   "parity_channels": parity_all_channels if kept else args.parity_channels,
                   "parity_vs_oracle_all_channels": (parity_all and parity_all_channels == C) if kept else None,
                   "parity_vs_oracle_compared_channels": parity_all,
                   "realtime_factor_per_channel": round(value * 1e6 / (C * world) / 48000.0, 1), "input_gen_s": round(B.t_gen, 1),
                   "parallelism": f"channels sharded contiguously over {world} GPU(s), global channel ids", "gather": B.gather_kind,
                   "steps_in_flight": F, "hw_queues": os.environ.get("GPU_MAX_HW_QUEUES"), "hw_queue_advice": int(B.ctx.lib.m17hip_advice(B.ctx.h)), "prewarm_steps": args.prewarm, "batches": "pipelined" if args.stagger else "launched and waited for in groups",
                   "gathered_set_ordered_and_unique": B.gathered_ok, "tune": B.tuned or None,
                   "redo_policy": "library default (m17hip_tune key 20 = 0: beside K5)"},
        "single_stream": single,
        "value_with_h2d": h2d["value_with_h2d"] if h2d else None, "h2d": h2d,
        "roofline": roofline, "cpu_baseline": cpu, "config2": config2, "bursty": bursty,
    }
    B.shutdown()
    try:   # RCCL writes a version banner through C stdio: flush it so that the JSON line is the LAST line of stdout
        import ctypes
        ctypes.CDLL(None).fflush(None)
    except Exception:   # noqa: BLE001
        pass
    print(json.dumps(out), flush=True)
    if B.parity is False or parity_all is False or (single and single.get("parity_vs_oracle_3_runs_first_channels") is False) or \
            (bursty and bursty.get("parity_vs_oracle_first_channels") is False) or (config2 and config2.get("outputs_bit_exact_first_channels") is False):
        sys.exit(3)   # a fast result that differs from the reference's is not a result


def cpu_baseline(args, ol, x, C, T, ncpu, ncpu_affinity, cpu_quota, chain, keep=None):
    """The CPU side of the line, on the host cores this process may use, one channel per thread, on a bounded sample of the same workload;
    plus the same binary on ONE thread.
    chain=True  (configs[2]): the oracle (scalar C++ restatement, g++ -O3) — kind "port": the reference's chain cannot be built here (blaze
                absent).  The records it computes are KEPT (keep["recs"], keep["channels"]): the caller compares every one of them with the GPU's.
    chain=False (configs[1]): FIR + correlator, outputs materialised.  The reference's OWN BaseFirFilter<float,150> and Correlator<float>
                (oracle/_ref/libm17ref.so = its headers compiled where they lay, FirFilter.h:28-43, Correlator.h:43-64) when that build
                travelled with the repository — kind "reference" — with the oracle's figure beside it; the oracle alone otherwise."""
    import concurrent.futures as cf
    import ctypes as Ct

    cap = 2 * (T // 1920 + 2) + 4
    if chain:
        def run(xs, threads):
            t = time.perf_counter()
            out = ol.demod_batch(xs, cap=cap, threads=threads)
            return time.perf_counter() - t, out

        t1, _ = run(x[:2], 1)                                     # one thread, two channels (one BERT, one voice-like)
        one_core = 2 * T / t1 / 1e6
        probe, _ = run(x[: min(C, ncpu)], ncpu)
        rate = min(C, ncpu) * T / max(probe, 1e-6)
        nch = int(min(C, max(ncpu, rate * args.cpu_seconds / T)))
        nch = min(C, max(ncpu, nch // ncpu * ncpu))
        tc, (er, ec, _) = run(x[:nch], ncpu)
        if keep is not None:
            keep["recs"] = np.concatenate([er[c, : ec[c]] for c in range(nch)]) if int(ec[:nch].sum()) else er[0, :0]
            keep["channels"] = nch
        return {"value": round(nch * T / tc / 1e6, 3), "unit": "Msamples/s", "cores": ncpu, "kind": "port", "one_core": round(one_core, 3),
                "cores_affinity": ncpu_affinity, "cgroup_cpu_quota": cpu_quota,
                "sample": f"{nch} of the {C} channels x {T} samples, one channel per thread, oracle/libm17oracle.so (g++ -O3 -ffp-contract=off)"}

    n = x.shape[1]
    taps = ol.taps()

    def port_channel(c):        # scaling + BaseFirFilter + Correlator::sample / limit / 4 x correlate, every output materialised
        lim = np.zeros(n, np.float32); corr = np.zeros((4, n), np.float32)
        y = ol.fir_i16(x[c])
        ol.oracle().m17o_correlator(ol._p(y), Ct.c_size_t(n), ol._p(lim), ol._p(corr))
        return y, lim, corr

    def ref_channel(c):         # the same with the reference's classes (the int16 -> float scaling is the application's one divide: the oracle's)
        lim = np.zeros(n, np.float32); corr = np.zeros((4, n), np.float32); y = np.zeros(n, np.float32)
        xs = ol.scale(x[c])
        ol.ref().ref_fir_f32(ol._p(taps), ol._p(xs), Ct.c_size_t(n), ol._p(y))
        ol.ref().ref_correlator(ol._p(y), Ct.c_size_t(n), ol._p(lim), ol._p(corr))
        return y, lim, corr

    def timed(fn):
        t = time.perf_counter(); fn(0); t1 = time.perf_counter() - t
        nch = int(min(C, max(ncpu, ncpu * args.cpu_seconds / max(t1, 1e-6))))
        nch = min(C, max(ncpu, nch // ncpu * ncpu))
        t = time.perf_counter()
        with cf.ThreadPoolExecutor(ncpu) as ex:   # the ctypes calls release the GIL
            list(ex.map(lambda c: fn(c)[0][0], range(nch)))
        return n / t1 / 1e6, nch * n / (time.perf_counter() - t) / 1e6, nch

    one_p, val_p, nch_p = timed(port_channel)
    port = {"value": round(val_p, 3), "unit": "Msamples/s", "cores": ncpu, "kind": "port", "one_core": round(one_p, 3),
            "sample": f"{nch_p} of the {C} channels x {n} samples (scaling + FIR + limit + 4 correlations materialised), one channel per thread, oracle/libm17oracle.so"}
    if ol.ref() is None:
        port.update({"cores_affinity": ncpu_affinity, "cgroup_cpu_quota": cpu_quota})
        return port
    one_r, val_r, nch_r = timed(ref_channel)
    same = all(np.array_equal(a_, b_) for a_, b_ in zip(port_channel(1 % C), ref_channel(1 % C)))
    return {"value": round(val_r, 3), "unit": "Msamples/s", "cores": ncpu, "kind": "reference", "one_core": round(one_r, 3),
            "cores_affinity": ncpu_affinity, "cgroup_cpu_quota": cpu_quota,
            "sample": f"{nch_r} of the {C} channels x {n} samples, one channel per thread: the reference's BaseFirFilter<float,150> and Correlator<float> "
                      "(sample / limit / correlate x 4 sync words), every output materialised — oracle/_ref/libm17ref.so, its headers compiled where they lay (g++ -O3)",
            "port": port, "port_outputs_equal_reference": bool(same)}


def bench_front(args, ctx, ol, x, C, T, rank, world, dev, sync, ncpu, ncpu_affinity, cpu_quota, t_gen):
    """BASELINE configs[1]: FIR + correlator only, outputs materialised in HBM (FIR out, limit, four correlations): 26 B/sample."""
    import torch
    import torch.distributed as dist

    def step():   # K1 (scaling + BaseFirFilter<float,150>), Correlator::sample (limit) and correlate x 4 words, pipelined in time; results stay on the device
        ctx.fir_correlator(fetch=False)

    for _ in range(args.warmup):
        step()
    ctx.timing(True); ctx.timing_reset()
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    sync()
    dt = time.perf_counter() - t0
    if world > 1:
        tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
    ctx.timing(False)
    kern = {}
    for name in ("fir_rrc150", "limit_track", "correlator"):
        ms, n = ctx.timing_get(name)
        kern[name] = {"ms_avg": (ms / n) if n else None, "launches": n, "ms_per_step": ms / args.steps}
    parity = None
    if rank == 0 and args.parity_channels > 0:   # soft outputs: north star asks 1e-5 relative; they are bit-exact
        k = min(4, C)
        n = min(T, 48000)
        ctx2 = type(ctx)(k, n)
        ctx2.upload(x[:k, :n]); y = ctx2.fir(); lim, corr = ctx2.correlator()
        parity = True
        for c in range(k):
            ye = ol.fir_i16(x[c, :n]); le, ce = ol.correlator(ye)
            parity = parity and np.array_equal(y[c], ye) and np.array_equal(lim[c], le) and np.array_equal(corr[:, c, :], ce)
        ctx2.close()
    if rank != 0:
        if world > 1:
            dist.destroy_process_group()
        return
    value = C * T * world * args.steps / dt / 1e6
    dom = max(kern, key=lambda k: kern[k]["ms_per_step"])
    dom_s = kern[dom]["ms_per_step"] / 1e3      # (the call runs its kernels in pieces in time: all of a step's launches of the dominant kernel)
    achieved = FRONT_BYTES * C * T / dom_s / 1e9 if dom_s else None
    roofline = {"bound": "hbm", "kernel": dom, "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 5),
                "traffic": None, "alg_bytes_per_sample": FRONT_BYTES, "launches_per_step": kern[dom]["launches"] / args.steps,
                "kernel_ms": {k: round(v["ms_per_step"], 4) for k, v in kern.items()},
                "chain_achieved_GBs": round(FRONT_BYTES * C * T * args.steps / dt / 1e9 / world, 2),
                "chain_frac": round(FRONT_BYTES * C * T * args.steps / dt / 1e9 / world / HBM_PEAK_GBS, 6)}
    cpu = cpu_baseline(args, ol, x, C, T, ncpu, ncpu_affinity, cpu_quota, chain=False) if (args.cpu_seconds > 0 and world == 1) else None
    print(json.dumps({
        "metric": "Msamples/s through FIR + correlator (48 kSPS 4-FSK in -> matched-filter output, limit, 4 sync correlations)",
        "value": round(value, 2), "unit": "Msamples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
        "data": "synthetic (generated on the device)",
        "config": {"workload": "configs[1]: 1024 independent 48 kSPS channels, FIR + Correlator only (NOT the headline; the headline is --config 3)",
                   "channels_per_gpu": C, "samples_per_channel": T, "awgn_sigma_lsb": args.sigma, "outputs_bit_exact_vs_oracle_first_channels": parity,
                   "input_gen_s": round(t_gen, 1)},
        "roofline": roofline, "cpu_baseline": cpu}))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
