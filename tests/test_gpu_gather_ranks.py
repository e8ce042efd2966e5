"""m17hip_gather_frames / m17hip_gather_frames_device with MORE THAN ONE RANK, on the one GPU of a test box.

The real librccl refuses two ranks on one device, and no multi-GPU node was ever available to the builder; so the N > 1 branch of the
product's gather (the count / confirmation all-gathers, the root's staging growth, the grouped ncclRecv against one ncclSend per peer,
the peers without records that both sides skip, the failure agreement, the bounded waits) runs here against a TEST DOUBLE of librccl
(tests/fake_rccl/fake_rccl.hip -> tests/fake_rccl/librccl.so.1), which the product's dlopen("librccl.so.1") finds first in the CHILD
processes only (LD_LIBRARY_PATH).  The double moves the bytes between the processes through shared memory and fails loudly when the two
sides of an exchange disagree about a size.  SURVEY §8(e); the reference itself is one process (apps/m17-demod.cpp:484-490).
"""
import glob
import json
import os
import subprocess
import sys
import time

import numpy as np
import pytest

import m17hip
import oracle_lib as ol

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FAKE_DIR = os.path.join(ROOT, "tests", "fake_rccl")
WORKER = os.path.join(ROOT, "tests", "gather_ranks_worker.py")
CT, T1, T2 = 24, 48000, 240000
SHARDS = {2: [0, 10, 24], 4: [0, 5, 12, 18, 24]}
EHIP, ENOMEM, ETRUNC, ECOMM = -2, -3, -6, -7


def _params(total):   # (the same as the worker's)
    return ol.gen_params(seed=77, kind=-1, n_frames=total // 1920 - 3, lead_in=3072, noise_sigma=500.0, tail_sigma=500.0, lead_sigma=40000.0, total=total)


_whole_cache = {}


def _whole(T, silent_from=None):
    """One big run over all 24 channels in this process (channels from `silent_from` on hear silence)."""
    key = (T, silent_from)
    if key not in _whole_cache:
        c = m17hip.Context(CT, T)
        c.synth(_params(T), CT, T)
        if silent_from is not None:
            x = c.download()
            x[silent_from:] = 0
            c.upload(x)
        c.reset(); c.run()
        _whole_cache[key] = c.frames().copy()
        c.close()
    return _whole_cache[key]


def _run_ranks(tmp_path, world, mode, wall_s=900):
    if not os.path.exists(os.path.join(FAKE_DIR, "librccl.so.1")):
        pytest.fail("tests/fake_rccl/librccl.so.1 is not built (__graft_entry__.build() / make -C tests/fake_rccl)")
    env = dict(os.environ)
    env["LD_LIBRARY_PATH"] = FAKE_DIR + os.pathsep + env.get("LD_LIBRARY_PATH", "")
    env["M17_FAKE_RCCL_LOG"] = str(tmp_path / "fake")
    env["M17_FAKE_RCCL_TIMEOUT_MS"] = "60000"
    env.pop("M17_FAKE_RCCL_ON_TIMEOUT", None)
    procs = [subprocess.Popen([sys.executable, WORKER, str(r), str(world), str(tmp_path), mode], env=env, stderr=open(tmp_path / f"err{r}.txt", "w"))
             for r in range(world)]
    t0 = time.time()
    try:
        while any(p.poll() is None for p in procs):
            if time.time() - t0 > wall_s:   # the wall-clock guard: a protocol that leaves a rank waiting fails here, it does not hang the suite
                pytest.fail(f"{mode}: ranks still running after {wall_s} s: " + str([p.poll() for p in procs]))
            time.sleep(0.1)
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
        for f in glob.glob("/dev/shm/m17fakerccl_*"):   # (segments a killed rank left behind)
            try:
                os.unlink(f)
            except OSError:
                pass
    errs = [open(tmp_path / f"err{r}.txt").read() for r in range(world)]
    for r, p in enumerate(procs):
        assert p.returncode == 0, f"rank {r}: exit {p.returncode}\n{errs[r][-3000:]}"
    res = [json.load(open(tmp_path / f"result{r}.json")) for r in range(world)]
    assert all(x.get("done") for x in res)
    return res, errs


@pytest.mark.parametrize("world", [2, 4])
def test_gather_protocol_between_ranks_on_one_gpu(tmp_path, world):
    res, errs = _run_ranks(tmp_path, world, "protocol")
    sh = SHARDS[world]
    one = _whole(T1)
    per_rank = [int(((one["channel"] >= sh[r]) & (one["channel"] < sh[r + 1])).sum()) for r in range(world)]
    assert one.size > CT and all(per_rank)
    for r in range(world):
        assert res[r]["bound_fake"] is True, "the product did not bind the test double"
        # basic: the gathered set == one big run, record for record; every rank knows every count
        b = res[r]["basic"]
        assert b["code"] == 0 and b["counts"] == per_rank and b["total"] == one.size
        assert b["ops"] == ["allgather", "allgather"] + (["recv"] * (world - 1) if r == 0 else ["send"])
        d = res[r]["device_root_last"]
        assert d["counts"] == per_rank and d["total"] == one.size
        t = res[r]["trunc"]
        assert t["total"] == one.size and t["code"] == (ETRUNC if r == 0 else 0) and t["head_ok"] in (True, None)
    assert np.load(tmp_path / "basic.npy").tobytes() == one.tobytes()
    assert np.load(tmp_path / "device_root_last.npy").tobytes() == one.tobytes()
    # a shard without records: its count is 0, neither side of exchange 3 names it
    zs = _whole(T1, silent_from=sh[world - 1])
    assert np.load(tmp_path / "zero_shard.npy").tobytes() == zs.tobytes()
    for r in range(world):
        z = res[r]["zero_shard"]
        assert z["code"] == 0 and z["counts"] == per_rank[:-1] + [0] and z["total"] == zs.size
        if r == 0:
            assert z["p2p"] == [f"recv:peer={k}" for k in range(1, world - 1)]
        elif r == world - 1:
            assert z["p2p"] == []
        else:
            assert z["p2p"] == ["send:peer=0"]
        a = res[r]["all_silent"]
        assert a["code"] == 0 and a["total"] == 0 and a["counts"] == [0] * world and a["p2p"] == []
    # the root's staging outgrown by a longer run
    big = _whole(T2)
    assert big.size > 1024 and big.size > 2 * one.size
    assert np.load(tmp_path / "growth.npy").tobytes() == big.tobytes()
    assert all(res[r]["growth"]["code"] == 0 and res[r]["growth"]["total"] == big.size for r in range(world))
    # failures: the failing rank returns its own code, every other rank ECOMM, all from the same call and at once; then the same
    # communicator delivers again
    for fault, own in ((1, EHIP), (3, EHIP), (4, EHIP), (2, ENOMEM)):
        for f in (1, 0):
            key = f"{fault}@{f}"
            for r in range(world):
                got = res[r]["faults"][key]
                if fault == 2 and f != 0:
                    want = 0            # only a root allocates staging
                else:
                    want = own if r == f else ECOMM
                assert got["code"] == want, (key, r, got)
                assert got["after"] == 0 and got["after_ok"] in (True, None), (key, r, got)
                assert got.get("s", 0) < 20, (key, r, got)
    assert all(res[r]["timeouts_before_fault5"] == 0 for r in range(world)), "a rank was left waiting in one of the agreed failures"
    assert all(res[r]["serial_300"]["bad"] == 0 for r in range(world))
    # fault 5: the failing rank leaves with its own code; whoever waits for it is released by a deadline (here: the double's), its
    # communicator is given up and answers ECOMM from then on; a new communicator works
    for f in (1, 0):
        for r in range(world):
            got = res[r]["fault5"][f"5@{f}"]
            assert got["code"] == (EHIP if r == f else ECOMM), (f, r, got)
            assert got["again"] in (None, ECOMM) and got["s"] < 30
            assert got["after"] == 0 and got["after_ok"] in (True, None)
    for e in errs:
        assert "FAKE_RCCL_SIZE_MISMATCH" not in e


def test_gather_wait_is_bounded_when_a_peer_never_comes(tmp_path):
    """The double in HANG mode behaves like the real library with a missing peer: the call succeeds, the stream never gets there.
    m17hip_tune key 31 bounds the wait, the communicator is given up (ncclCommAbort), later calls answer ECOMM, a new one works."""
    res, errs = _run_ranks(tmp_path, 2, "hang")
    for r in range(2):
        assert res[r]["clean"] == 0
        h = res[r]["hang"]
        assert h["code"] == (EHIP if r == 1 else ECOMM), h
        assert h["again"] == ECOMM and h["s"] < 30, h
        assert res[r]["after"]["code"] == 0 and res[r]["after"]["ok"] in (True, None)
    assert res[0]["hang"]["s"] >= 2.5 and res[0]["hang"]["last_error"] != 0   # (1 s of the double + 3 s of key 31 on the root)
