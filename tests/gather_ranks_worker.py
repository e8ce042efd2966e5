"""One RANK of the N > 1 tests of m17hip_gather_frames[_device] (tests/test_gpu_gather_ranks.py starts N of these on the ONE GPU of a test
box, with tests/fake_rccl first in LD_LIBRARY_PATH so that the product's dlopen("librccl.so.1") binds the test double).

    gather_ranks_worker.py <rank> <world> <dir> <mode>

mode "protocol": the scenarios below, one after the other; "hang": the bounded wait (fake library in hang mode).
Every rank writes <dir>/result<rank>.json ({scenario: ...}) and the root of a scenario its gathered records as <dir>/<scenario>.npy.
No torch in this process: torch carries an RCCL of its own and the product would bind that one.
"""
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "m17-cxx-demod_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import m17hip  # noqa: E402
import oracle_lib as ol  # noqa: E402  (generator parameters only: the layout of m17_synth_params)

assert "torch" not in sys.modules

CT = 24
SHARDS = {2: [0, 10, 24], 4: [0, 5, 12, 18, 24]}
T1, T2 = 48000, 240000
EHIP, ENOMEM, ECOMM = -2, -3, -7


def params(total):
    return ol.gen_params(seed=77, kind=-1, n_frames=total // 1920 - 3, lead_in=3072, noise_sigma=500.0, tail_sigma=500.0, lead_sigma=40000.0, total=total)


class Rank:
    def __init__(self, rank, world, d):
        self.rank, self.world, self.dir = rank, world, d
        self.lo, self.hi = SHARDS[world][rank], SHARDS[world][rank + 1]
        self.n = self.hi - self.lo
        self.ctx = m17hip.Context(self.n, T2)
        self.ctx.set_channel_base(self.lo)
        self.lib = self.ctx.lib
        self.ids = 0
        self.res = {}
        self.hip = C.CDLL("libamdhip64.so")
        self.logpos = 0

    # ---- plumbing ---------------------------------------------------------------------------------------------------------
    def new_comm(self):
        """Collective: rank 0 draws an id and leaves it in a file."""
        self.ids += 1
        path = os.path.join(self.dir, f"id{self.ids}")
        if self.rank == 0:
            with open(path + ".tmp", "wb") as f:
                f.write(m17hip.comm_get_id())
            os.rename(path + ".tmp", path)
        t0 = time.time()
        while not os.path.exists(path):
            assert time.time() - t0 < 300, "no communicator id from rank 0"
            time.sleep(0.02)
        return m17hip.Comm(self.ctx, open(path, "rb").read(), self.rank, self.world)

    def run_signal(self, T, silent=False):
        if silent:
            self.ctx.upload(np.zeros((self.n, T), dtype=np.int16))
        else:
            self.ctx.synth(params(T), self.n, T, chan0=self.lo)
        self.ctx.reset()
        self.ctx.run()

    def gather_raw(self, comm, root=0, capacity=1 << 16):
        """The C entry point itself: (code, records or None, counts, total)."""
        counts = np.zeros(self.world, dtype=np.uint64)
        total = C.c_uint64(0)
        recs = np.zeros(capacity, dtype=m17hip.FRAME_REC) if self.rank == root else None
        code = self.lib.m17hip_gather_frames(self.ctx.h, comm.h, C.c_int(root), None if recs is None else recs.ctypes.data_as(C.c_void_p),
                                             C.c_uint64(capacity if recs is not None else 0), counts.ctypes.data_as(C.c_void_p), C.byref(total))
        return code, (recs[: min(total.value, capacity)] if recs is not None else None), counts.tolist(), int(total.value)

    def log_lines(self):
        """Lines the test double has logged for this rank since the last call of this function."""
        pre = os.environ.get("M17_FAKE_RCCL_LOG")
        path = f"{pre}.rank{self.rank}"
        lines = open(path).read().splitlines() if os.path.exists(path) else []
        new, self.logpos = lines[self.logpos:], len(lines)
        return new

    def save(self, name, recs):
        np.save(os.path.join(self.dir, name + ".npy"), recs)

    def barrier_file(self, tag):
        """Host-side rendezvous outside the library under test (files): nobody goes on before everybody is here."""
        open(os.path.join(self.dir, f"b_{tag}_{self.rank}"), "w").close()
        t0 = time.time()
        while not all(os.path.exists(os.path.join(self.dir, f"b_{tag}_{k}")) for k in range(self.world)):
            assert time.time() - t0 < 300, f"barrier {tag}"
            time.sleep(0.01)

    # ---- scenarios --------------------------------------------------------------------------------------------------------------
    def protocol(self):
        R, W = self.rank, self.world
        comm = self.new_comm()
        maps = open("/proc/self/maps").read()
        self.res["bound_fake"] = "tests/fake_rccl/librccl.so.1" in maps and "/opt/rocm" not in [ln for ln in maps.splitlines() if "librccl" in ln][0]
        self.log_lines()

        # basic: host destination on root 0; every rank learns every count
        self.run_signal(T1)
        code, recs, counts, total = self.gather_raw(comm)
        self.res["basic"] = {"code": code, "counts": counts, "total": total}
        if R == 0:
            self.save("basic", recs)
        clean = None if recs is None else recs.tobytes()
        lines = self.log_lines()
        self.res["basic"]["ops"] = [ln.split()[0] for ln in lines]

        # device destination on the LAST rank as root
        root = W - 1
        cap = 1 << 14
        dptr = C.c_void_p()
        assert self.hip.hipMalloc(C.byref(dptr), C.c_size_t(cap * 64)) == 0
        tot, cnts = self.ctx.gather_frames_device(comm, dptr.value, cap, root=root)
        self.res["device_root_last"] = {"counts": cnts.tolist(), "total": tot}
        if R == root:
            out = np.zeros(tot, dtype=m17hip.FRAME_REC)
            assert self.hip.hipMemcpy(out.ctypes.data_as(C.c_void_p), dptr, C.c_size_t(tot * 64), C.c_int(2)) == 0
            self.save("device_root_last", out)
        self.hip.hipFree(dptr)
        self.log_lines()

        # a truncated host destination on the root: ETRUNC there, the count still right everywhere
        code, recs, counts, total = self.gather_raw(comm, capacity=7)
        self.res["trunc"] = {"code": code, "total": total, "head_ok": (recs.tobytes() == clean[: 7 * 64]) if R == 0 else None}

        # a shard without records (the last rank hears silence): skipped on both sides of exchange 3
        self.run_signal(T1, silent=(R == W - 1))
        self.log_lines()
        code, recs, counts, total = self.gather_raw(comm)
        ops = [ln.split()[0] + ":" + ln.split()[1] for ln in self.log_lines() if ln.startswith(("send", "recv"))]
        self.res["zero_shard"] = {"code": code, "counts": counts, "total": total, "p2p": ops}
        if R == 0:
            self.save("zero_shard", recs)
        # ... and nobody has any
        self.run_signal(T1, silent=True)
        code, recs, counts, total = self.gather_raw(comm)
        self.res["all_silent"] = {"code": code, "counts": counts, "total": total, "p2p": [ln for ln in self.log_lines() if ln.startswith(("send", "recv"))]}

        # the root's staging outgrown (first sized for the 48 000-sample runs above): grown inside exchange 2
        self.run_signal(T2)
        code, recs, counts, total = self.gather_raw(comm)
        self.res["growth"] = {"code": code, "counts": counts, "total": total}
        if R == 0:
            self.save("growth", recs)
        self.log_lines()

        # fault injection, on a non-root rank and on the root: every rank returns from the SAME call; the communicator stays usable
        self.run_signal(T1)
        faults = {}
        for fault in (1, 3, 4):
            for f in (1, 0):
                if R == f:
                    self.ctx.tune(30, fault)
                t0 = time.time()
                code, _, _, _ = self.gather_raw(comm)
                self.ctx.tune(30, 0)
                code2, recs, counts, total = self.gather_raw(comm)
                faults[f"{fault}@{f}"] = {"code": code, "s": time.time() - t0, "after": code2, "after_ok": (recs.tobytes() == clean) if R == 0 else None}
        comm.close()
        for f in (1, 0):   # fault 2 bites on a root without staging: a fresh communicator
            comm = self.new_comm()
            if R == f:
                self.ctx.tune(30, 2)
            code, _, _, _ = self.gather_raw(comm)
            self.ctx.tune(30, 0)
            code2, recs, counts, total = self.gather_raw(comm)
            faults[f"2@{f}"] = {"code": code, "after": code2, "after_ok": (recs.tobytes() == clean) if R == 0 else None}
            comm.close()
        self.res["faults"] = faults
        timeouts_so_far = sum(ln.startswith("timeout") for ln in self.log_lines())
        self.res["timeouts_before_fault5"] = timeouts_so_far

        # 300 consecutive calls: the 16-bit call serial passes 255 (an 8-bit one would meet its own stale words again)
        comm = self.new_comm()
        bad = 0
        t0 = time.time()
        for _ in range(300):
            code, recs, counts, total = self.gather_raw(comm)
            if code != 0 or (R == 0 and recs.tobytes() != clean):
                bad += 1
        self.res["serial_300"] = {"bad": bad, "s": time.time() - t0}

        # fault 5 (a rank that read exchange 1 but cannot read exchange 2): the one case words cannot settle.  With the test double in
        # error mode the waiting side's call fails after the double's own deadline; the communicator is given up and a new one works
        os.environ["M17_FAKE_RCCL_TIMEOUT_MS"] = "1500"
        five = {}
        for f in (1, 0):
            self.barrier_file(f"five{f}")
            if R == f:
                self.ctx.tune(30, 5)
            t0 = time.time()
            code, _, _, _ = self.gather_raw(comm)
            self.ctx.tune(30, 0)
            again, _, _, _ = self.gather_raw(comm) if code == ECOMM else (None, None, None, None)   # a given-up communicator answers ECOMM at once
            five[f"5@{f}"] = {"code": code, "s": time.time() - t0, "again": again}
            comm.close()
            self.barrier_file(f"five{f}done")
            os.environ["M17_FAKE_RCCL_TIMEOUT_MS"] = "60000"
            comm = self.new_comm()
            os.environ["M17_FAKE_RCCL_TIMEOUT_MS"] = "1500"
            code2, recs, counts, total = self.gather_raw(comm)
            five[f"5@{f}"].update({"after": code2, "after_ok": (recs.tobytes() == clean) if R == 0 else None})
        self.res["fault5"] = five
        comm.close()

    def hang(self):
        """The library's own bounded wait: the test double, in hang mode, answers a missing peer the way the real library does — the call
        succeeds and the stream never gets there."""
        R = self.rank
        comm = self.new_comm()
        self.run_signal(T1)
        code, recs, counts, total = self.gather_raw(comm)
        clean = None if recs is None else recs.tobytes()
        self.res["clean"] = code
        os.environ["M17_FAKE_RCCL_TIMEOUT_MS"] = "1000"
        os.environ["M17_FAKE_RCCL_ON_TIMEOUT"] = "hang"
        self.ctx.tune(31, 3000)
        if R == 1:
            self.ctx.tune(30, 5)
        t0 = time.time()
        code, _, _, _ = self.gather_raw(comm)
        dt = time.time() - t0
        self.ctx.tune(30, 0)
        again, _, _, _ = self.gather_raw(comm)
        self.res["hang"] = {"code": code, "s": dt, "again": again, "last_error": self.lib.m17hip_comm_last_error(comm.h)}
        comm.close()
        self.barrier_file("hang")
        os.environ["M17_FAKE_RCCL_TIMEOUT_MS"] = "60000"
        os.environ["M17_FAKE_RCCL_ON_TIMEOUT"] = "error"
        comm = self.new_comm()
        code2, recs, counts, total = self.gather_raw(comm)
        self.res["after"] = {"code": code2, "ok": (recs.tobytes() == clean) if R == 0 else None}
        comm.close()


def main():
    rank, world, d, mode = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], sys.argv[4]
    r = Rank(rank, world, d)
    try:
        getattr(r, mode)()
        r.res["done"] = True
    finally:
        with open(os.path.join(d, f"result{rank}.json"), "w") as f:
            json.dump(r.res, f)
        r.ctx.close()


if __name__ == "__main__":
    main()
