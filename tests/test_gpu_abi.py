"""GPU: C-ABI behaviour beyond the happy path — error codes, truncation, overflow, buffer lifetimes, the device guard, global
channel ids and the RCCL gather entry points (include/m17hip.h)."""
import ctypes as C
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

import m17hip
import oracle_lib as ol

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _signals(Cn, T, seed=5, sigma=500.0, kind=-1):
    p = ol.gen_params(seed=seed, kind=kind, n_frames=max(1, T // 1920 - 4), lead_in=3072, noise_sigma=sigma, tail_sigma=sigma,
                      lead_sigma=40000.0, total=T)
    return ol.generate_batch(p, Cn, T, threads=8)


def _oracle_flat(x, base=0):
    recs, counts, _ = ol.demod_batch(x, cap=2 * (x.shape[1] // 1920 + 2) + 4, threads=8)
    flat = np.concatenate([recs[c, :counts[c]] for c in range(x.shape[0])]).copy()
    flat["channel"] += base
    return flat


def test_error_codes():
    lib = m17hip.load_library()
    h = C.c_void_p()
    assert lib.m17hip_ctx_create(0, 0, 100, C.byref(h)) == -1                       # EINVAL: no channels
    assert lib.m17hip_ctx_create(0, 4, 0, C.byref(h)) == -1
    assert lib.m17hip_ctx_create(0, 4, 100, None) == -1
    assert lib.m17hip_ctx_create(9999, 4, 4800, C.byref(h)) == -2 and lib.m17hip_strerror(-2) == b"HIP runtime error"
    c = m17hip.Context(4, 4800)
    n = C.c_uint64(0)
    assert lib.m17hip_demod_run(c.h, 4, 4800, 0) == -4                              # ESTATE: nothing uploaded
    assert lib.m17hip_frames_count(c.h, C.byref(n)) == -4                           # ESTATE: no run yet
    assert lib.m17hip_fir_rrc150(c.h, 4, 4800, 0, None) == -4
    x = _signals(4, 4800)
    c.upload(x)
    assert lib.m17hip_demod_run(c.h, 5, 4800, 0) == -1                              # EINVAL: beyond the context
    assert lib.m17hip_demod_run(c.h, 4, 4801, 0) == -1
    assert lib.m17hip_demod_run(c.h, 4, 4800, 2) == -1 and lib.m17hip_fir_rrc150(c.h, 4, 4800, 16, None) == -1   # unknown flag bits
    assert lib.m17hip_dcd(c.h, 4, 4800, 16, None, None) == -1                       # (role switches of K3: diagnostics knob only)
    assert lib.m17hip_upload_i16(c.h, x.ctypes.data_as(C.c_void_p), 4, 4800, C.c_size_t(100)) == -1   # pitch < samples
    assert lib.m17hip_set_kalman_order(c.h, 8) == -1 and lib.m17hip_set_kalman_order(c.h, 3) == 0
    assert lib.m17hip_tune(c.h, 99, 0) == -1 and lib.m17hip_tune(c.h, 0, 3) == -1
    bs = np.zeros(4, dtype=m17hip.BERT_STAT)
    assert lib.m17hip_bert_stats(c.h, bs.ctypes.data_as(C.c_void_p), 4) == -4       # consumer not enabled
    c.run()
    assert lib.m17hip_demod_run(c.h, 2, 4800, 0) == -1                              # a continued stream keeps its channel count
    assert lib.m17hip_strerror(-6) == b"output truncated to the caller's capacity"
    c.close()


def test_truncated_fetch_reports_total():
    x = _signals(16, 48000, seed=8)
    c = m17hip.Context(16, 48000)
    c.upload(x); c.reset(); c.run()
    full = c.frames()
    assert full.size > 32
    part = np.zeros(10, dtype=m17hip.FRAME_REC)
    n = C.c_uint64(0)
    code = c.lib.m17hip_frames_fetch(c.h, part.ctypes.data_as(C.c_void_p), C.c_uint64(10), C.byref(n))
    assert code == m17hip.ETRUNC and n.value == full.size and part.tobytes() == full[:10].tobytes()
    import torch
    with torch.cuda.stream(c.torch_stream()):      # the zero fill ON the context's stream: it is non-blocking, the default stream orders nothing with it
        dev = torch.zeros(7 * 64, dtype=torch.uint8, device="cuda")
    code = c.lib.m17hip_frames_compact_device(c.h, C.c_void_p(dev.data_ptr()), C.c_uint64(7), C.byref(n))
    assert code == m17hip.ETRUNC and n.value == full.size
    assert dev.cpu().numpy().tobytes() == full[:7].tobytes()
    c.close()


def test_record_buffer_overflow_is_reported():
    """M17HIP_EOVERFLOW: with the default sizing a run cannot outgrow its record slots (2 callbacks per 1920 samples + 8; 25 short
    runs on a context sized for one frame prove the slots start afresh every run); with the slots cut to 3 per channel (knob 8)
    the fetch reports the overflow, still returns the records that fit, and a reset clears the condition."""
    x = _signals(8, 48000, seed=3)
    c = m17hip.Context(8, 1920)      # 10 record slots per channel and run
    c.reset()
    total = 0
    for k in range(25):
        c.upload(x[:, 1920 * k:1920 * (k + 1)]); c.run()
        total += c.frames().size
    exp = _oracle_flat(x[:, :1920 * 25])
    assert total == exp.size and total > 8
    c.close()
    c = m17hip.Context(8, 48000)
    c.tune(8, 3)
    c.upload(x); c.reset(); c.run()
    n = C.c_uint64(0)
    assert c.lib.m17hip_frames_count(c.h, C.byref(n)) == -5 and n.value == 8 * 3
    got = np.zeros(64, dtype=m17hip.FRAME_REC)
    assert c.lib.m17hip_frames_fetch(c.h, got.ctypes.data_as(C.c_void_p), C.c_uint64(64), C.byref(n)) == -5
    full = _oracle_flat(x)
    first3 = np.concatenate([full[full["channel"] == ch][:3] for ch in range(8)])
    assert n.value == 24 and got[:24].tobytes() == first3.tobytes()
    c.tune(8, 0)
    c.upload(x); c.reset(); c.run()
    assert c.frames().tobytes() == full.tobytes()
    # the packet consumer has its own room: more completed packets than tune(7, room) -> EOVERFLOW from the fetch
    p = ol.gen_params(seed=6, kind=4, n_frames=3, lead_in=3072, noise_sigma=300.0, tail_sigma=300.0, lead_sigma=40000.0, total=48000)
    c.tune(7, 2)
    c.synth(p, 8, 48000); c.reset(); c.run()
    out = np.zeros(8, dtype=m17hip.PACKET_REC); cnt = C.c_uint32(0)
    assert c.lib.m17hip_packets_fetch(c.h, out.ctypes.data_as(C.c_void_p), 8, C.byref(cnt)) == -5 and cnt.value > 2
    c.close()


def test_async_upload_buffer_may_be_reused_after_upload_wait():
    """m17hip_demod_run does not wait for the staged copy on the host.  The documented rule: the pinned buffer is the
    caller's again after m17hip_upload_wait.  One pinned buffer, refilled for every run right after the wait."""
    import torch
    Cn, T, n = 32, 9600, 6
    x = _signals(Cn, n * T, seed=21)
    exp = _oracle_flat(x)
    c = m17hip.Context(Cn, T)
    pinned = torch.zeros((Cn, T), dtype=torch.int16).pin_memory()
    c.reset()
    parts = []
    for k in range(n):
        pinned.numpy()[:] = x[:, k * T:(k + 1) * T]
        c.upload_async(pinned.data_ptr(), Cn, T)
        c.run()
        c.upload_wait()
        pinned.numpy()[:] = -12345          # scribble: the copy has left the buffer
        parts.append(c.frames().copy())
    got = np.concatenate(parts)
    got = got[np.lexsort((got["seq"], got["channel"]))]
    assert got.tobytes() == exp.tobytes()
    c.close()


def test_device_guard_restores_callers_device_and_ignores_it():
    """Entry points run on the context's device whatever the caller's current device is, and leave the caller's in place."""
    import torch
    c = m17hip.Context(4, 4800)
    x = _signals(4, 4800)
    torch.cuda.set_device(0)
    c.upload(x); c.reset(); c.run()
    assert torch.cuda.current_device() == 0
    assert c.frames().tobytes() == _oracle_flat(x).tobytes()
    c.close()


def test_channel_base_gives_global_channel_ids():
    """SURVEY §8e's correctness check on one GPU: two contexts = two shards of a 24-channel job; with channel bases set the
    concatenation of their record sets IS the record set of the single 24-channel run (and of the oracle)."""
    x = _signals(24, 48000, seed=31)
    whole = m17hip.Context(24, 48000)
    whole.upload(x); whole.reset(); whole.run()
    one = whole.frames().copy()
    whole.close()
    parts = []
    for lo, hi in ((0, 10), (10, 24)):
        c = m17hip.Context(hi - lo, 48000)
        c.set_channel_base(lo)
        c.upload(x[lo:hi]); c.reset(); c.run()
        parts.append(c.frames().copy())
        c.close()
    got = np.concatenate(parts)
    assert got.tobytes() == one.tobytes() == _oracle_flat(x).tobytes()
    key = got["channel"].astype(np.int64) << 32 | got["seq"]
    assert (np.diff(key) > 0).all()                  # globally (channel, seq)-ordered, no duplicates


def test_two_batches_in_flight_equal_one_after_the_other():
    """bench.py's default: two contexts (independent batches) on their own streams, the second step queued before the first is
    waited for, so that its front end runs beside the first one's tail.  Same records as the batches run one after the other."""
    import torch
    xs = [_signals(32, 96000, seed=61), _signals(32, 96000, seed=62, sigma=1200.0)]
    exp = [_oracle_flat(x) for x in xs]
    ctxs, streams = [], [torch.cuda.Stream(), torch.cuda.Stream()]
    for x, st in zip(xs, streams):
        c = m17hip.Context(32, 96000)
        c.set_stream(st.cuda_stream)
        c.upload(x)
        ctxs.append(c)
    for rep in range(3):
        for c in ctxs:
            c.reset(); c.run()          # both queued ...
        for c, e in zip(ctxs, exp):
            assert c.frames().tobytes() == e.tobytes(), rep   # ... then waited for
    for c in ctxs:
        c.close()


def test_the_library_owns_a_contexts_streams_and_hands_them_on_as_a_set():
    """m17hip_get_stream (include/m17hip.h): a context's main stream is the library's own — non-blocking, distinct per live context — and a
    destroyed context's streams are parked as a set: the next context gets the same main stream.  m17hip_set_stream stays as the opt-in
    (NULL = the default stream).  A host orders its tensor work with the context's by wrapping the stream (torch.cuda.ExternalStream):
    input produced, demodulated, records compacted into a torch buffer and read — no device-wide synchronisation anywhere."""
    import torch
    a, b = m17hip.Context(16, 48000), m17hip.Context(16, 48000)
    sa, sb = a.stream, b.stream
    assert sa and sb and sa != sb
    a.close()
    c = m17hip.Context(32, 96000)
    assert c.stream == sa                                        # the parked set, whatever the new context's size
    b.set_stream(0); assert b.stream == 0
    b.set_stream(sb); assert b.stream == sb
    b.close()
    x = _signals(32, 96000, seed=63, sigma=700.0)
    exp = _oracle_flat(x)
    host = torch.from_numpy(x).pin_memory()
    st = c.torch_stream()
    for rep in range(3):
        with torch.cuda.stream(st):
            dev = host.to("cuda", non_blocking=True) + 0         # a producer ON the context's stream: the copy and an elementwise kernel
            c.upload_device(dev.data_ptr(), 32, 96000)
            c.reset(); c.run()
            out = torch.full((exp.size + 8, 64), 0xEE, dtype=torch.uint8, device="cuda")
            n = c.frames_compact_device(out.data_ptr(), exp.size + 8)
            got = out[:n].cpu().numpy()                           # (.cpu() waits for THIS stream only)
        assert n == exp.size and got.tobytes() == exp.tobytes(), rep
    c.close()


def test_performance_knobs_do_not_change_results():
    """m17hip_tune keys that only move work around (segment length, where payload frames are decoded): the same records under every
    setting; keys the production library does not have (the measurement build's) are M17HIP_EINVAL."""
    x = _signals(48, 96000, seed=71, sigma=900.0)
    exp = _oracle_flat(x)
    c = m17hip.Context(48, 96000)
    c.upload(x)
    for settings in ({}, {3: 19200}, {3: 0}, {15: 0}, {15: 0, 3: 7001}, {3: 4800}, {3: 1000}, {10: 1}, {10: 1, 3: 7008}, {10: 0}, {20: 1}, {20: 1, 3: 4800}, {20: 0}, {17: 0}, {17: 0, 3: 7001, 15: 0}, {17: 1, 3: 4800},
                     {13: 256}, {13: 7, 3: 7001}, {13: 100000, 3: 4800},   # K1's bounded grid at other sizes
                     {26: 1}, {26: 1, 3: 7001}, {26: 1, 3: 4800, 15: 0}, {26: 1, 3: 19200, 10: 1}, {26: 0},   # the gate-aware front end forced on / off
                     {33: 0}, {33: 2400}, {33: 9600, 3: 19200}, {33: 4801, 3: 7001, 15: 0}, {33: 12000, 20: 1}):   # the ramp of short first segments off / explicit (default -1: per run)
        for k, v in settings.items():
            c.tune(k, v)
        c.reset(); c.run()
        assert c.frames().tobytes() == exp.tobytes(), settings
        for k in settings:
            c.tune(k, {3: 48000, 15: 1, 10: -1, 20: 0, 17: 1, 13: 0, 26: -1, 33: -1}[k])   # back to the defaults
    import ctypes as C_
    for key in (0, 1, 2, 4, 5, 11, 12, 14, 19, 21, 22, 25):
        assert c.lib.m17hip_tune(c.h, key, C_.c_int64(1)) == -1, key
    c.close()


_TOOLS_WORKER = r"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join({root!r}, "m17-cxx-demod_amd")); sys.path.insert(0, os.path.join({root!r}, "tests"))
import torch  # noqa: F401
import m17hip, oracle_lib as ol
T = 96000
p = ol.gen_params(seed=5, kind=-1, n_frames=T // 1920 - 4, lead_in=3072, noise_sigma=900.0, tail_sigma=900.0, lead_sigma=40000.0, total=T)
x = ol.generate_batch(p, 24, T, threads=8)
recs, counts, _ = ol.demod_batch(x, cap=2 * (T // 1920 + 2) + 4, threads=8)
exp = np.concatenate([recs[c, :counts[c]] for c in range(24)]).tobytes()
c = m17hip.Context(24, T)
defaults = {{4: 0, 5: 0, 11: 1, 12: 0, 14: 0, 21: 0, 25: 1}}
for settings in ({{11: 0}}, {{5: 2}}, {{12: 3}}, {{14: 32640}}, {{4: 9600}}, {{5: 1, 12: 2, 4: 4800}}):
    for k, v in settings.items(): c.tune(k, v)
    c.upload(x); c.reset(); c.run()
    assert c.frames().tobytes() == exp, settings
    for k in settings: c.tune(k, defaults[k])
# the knobs of the staged path: K1 of a staged run held back (21), the first replay not queued by m17hip_demod_front (25)
for settings in ({{21: 2}}, {{25: 0}}, {{21: 1, 25: 0}}):
    for k, v in settings.items(): c.tune(k, v)
    c.reset()
    got = []
    cuts = (0, 31000, 66000, T)     # three runs, the second and third staged and queued through m17hip_demod_front
    for i in range(3):
        lo, hi = cuts[i], cuts[i + 1]
        if i == 0:
            c.upload(np.ascontiguousarray(x[:, lo:hi]))
        else:
            c.tune(16, 1); c.upload(np.ascontiguousarray(x[:, lo:hi])); c.tune(16, 0)
            c.front()
            got.append(c.frames())
        c.run()
    got.append(c.frames())
    g = np.concatenate(got)
    g = g[np.lexsort((g["seq"], g["channel"]))]
    assert g.tobytes() == exp, settings
    for k in settings: c.tune(k, defaults[k])
# round 4's limit pipeline (key 27 = 0) against the relayed recurrence (default): configs[1] in one call and as the per-operator call
c.upload(x)
ref = c.fir_correlator()
lim_ref, corr_ref = c.correlator()
c.tune(27, 0)
got = c.fir_correlator()
lim_old, corr_old = c.correlator()
c.tune(27, 1)
assert all(np.array_equal(a, b) for a, b in zip(ref, got)) and np.array_equal(lim_ref, lim_old) and np.array_equal(corr_ref, corr_old) and np.array_equal(ref[1], lim_ref)
print("tools-build knobs ok")
"""


def test_measurement_build_knobs_do_not_change_results(tmp_path):
    """the schedule-experiment members of the context (front_ahead, front_first, seq_lds_bytes, front_k1_after, gate0_early, the
    first-segment length, round 4's K1, round 4's limit pipeline) can only be set in the measurement build (libm17hip_tools.so, -DM17_TOOLS): drive their non-default
    branches there and compare with the oracle, in a process of its own (this one has the production library loaded)."""
    lib = os.path.join(ROOT, "m17-cxx-demod_amd", "libm17hip_tools.so")
    if not os.path.exists(lib):
        pytest.skip("measurement build not made (make -C m17-cxx-demod_amd/csrc tools)")
    script = tmp_path / "w.py"
    script.write_text(_TOOLS_WORKER.format(root=ROOT))
    r = subprocess.run([sys.executable, str(script)], env=dict(os.environ, M17HIP_LIB=lib), capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "tools-build knobs ok" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]


_LEGACY_STREAMS_WORKER = r"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join({root!r}, "m17-cxx-demod_amd")); sys.path.insert(0, os.path.join({root!r}, "tests"))
import m17hip, oracle_lib as ol
T = 96000
p = ol.gen_params(seed=9, kind=-1, n_frames=T // 1920 - 4, lead_in=3072, noise_sigma=700.0, tail_sigma=700.0, lead_sigma=40000.0, total=T)
x = ol.generate_batch(p, 24, T, threads=8)
recs, counts, _ = ol.demod_batch(x, cap=2 * (T // 1920 + 2) + 4, threads=8)
exp = np.concatenate([recs[c, :counts[c]] for c in range(24)]).tobytes()
a = m17hip.Context(24, T)
assert a.stream == 0, a.stream                    # the default stream, as up to round 5
a.upload(x); a.reset(); a.run()
assert a.frames().tobytes() == exp
a.close()
b = m17hip.Context(24, T)                        # streams per context: created and destroyed with it
b.upload(x[:, :48000]); b.reset(); b.run(); first = b.frames().copy()
b.upload(x[:, 48000:]); b.run()
got = np.concatenate([first, b.frames()]); got = got[np.lexsort((got["seq"], got["channel"]))]
assert got.tobytes() == exp
b.close()
print("legacy streams ok")
"""


def test_per_context_streams_on_the_default_stream_still_work(tmp_path):
    """M17HIP_STREAM_SETS=0 (include/m17hip.h, m17hip_set_stream): the diagnostic mode that restores rounds 1-5 — main = the default stream, role
    streams created and destroyed with the context — delivers the same records (a process of its own: the mode is read at context creation)."""
    script = tmp_path / "w.py"
    script.write_text(_LEGACY_STREAMS_WORKER.format(root=ROOT))
    r = subprocess.run([sys.executable, str(script)], env=dict(os.environ, M17HIP_STREAM_SETS="0"), capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "legacy streams ok" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]


def test_deferred_evm_row_overflow_is_reported_not_silent():
    """the rows of deferred EVM operations (m17hip_tune key 17) are sized for what a run can produce; should a channel ever
    outrun its row, the operations beyond are dropped — m17hip_diag_fetch must SAY so (M17HIP_EOVERFLOW) instead of handing out a wrong
    EVM.  Key 18 (tests) shrinks the rows to make that reachable; the frame records are unaffected, and with the default rows the same
    run reports nothing."""
    x = _signals(8, 48000, seed=33)
    exp = _oracle_flat(x)
    c = m17hip.Context(8, 48000)
    d = np.zeros(8, dtype=m17hip.DIAG)
    c.upload(x); c.reset(); c.run()
    assert c.lib.m17hip_diag_fetch(c.h, d.ctypes.data_as(C.c_void_p), C.c_uint32(8)) == 0
    good = d.copy()
    c.tune(18, 64)                       # 64 operations per channel: a 48 000-sample run writes thousands
    c.reset(); c.run()
    assert c.frames().tobytes() == exp.tobytes()
    assert c.lib.m17hip_diag_fetch(c.h, d.ctypes.data_as(C.c_void_p), C.c_uint32(8)) == -5
    for f in d.dtype.names:
        if f != "evm":
            assert np.array_equal(d[f], good[f]), f
    c.tune(18, 0)
    c.reset(); c.run()                   # (the reset clears the flag)
    assert c.lib.m17hip_diag_fetch(c.h, d.ctypes.data_as(C.c_void_p), C.c_uint32(8)) == 0 and d.tobytes() == good.tobytes()
    c.close()


def test_rccl_gather_single_rank():
    """m17hip_comm_* / m17hip_gather_frames with a 1-rank communicator: RCCL is bound, the counts all-gather and the
    compaction run, the root receives its own records."""
    x = _signals(12, 48000, seed=41)
    c = m17hip.Context(12, 48000)
    c.set_channel_base(100)
    comm = m17hip.Comm(c, m17hip.comm_get_id(), 0, 1)
    c.upload(x); c.reset(); c.run()
    recs, counts = c.gather_frames(comm, root=0)
    assert counts.tolist() == [recs.size] and recs.tobytes() == _oracle_flat(x, base=100).tobytes()
    small = np.zeros(5, dtype=m17hip.FRAME_REC)
    tot = C.c_uint64(0)
    code = c.lib.m17hip_gather_frames(c.h, comm.h, 0, small.ctypes.data_as(C.c_void_p), C.c_uint64(5), None, C.byref(tot))
    assert code == m17hip.ETRUNC and tot.value == recs.size and small.tobytes() == recs[:5].tobytes()
    # a run that outran its record slots (tune 8) followed by a gather — the dense buffer is sized for what the run did
    # produce whatever the first compaction pass returned; the call reports EOVERFLOW and delivers the records that exist
    c2 = m17hip.Context(12, 48000)
    c2.tune(8, 3)
    c2.upload(x); c2.reset(); c2.run()
    big = np.zeros(4096, dtype=m17hip.FRAME_REC)
    cnt = np.zeros(1, dtype=np.uint64)
    code = c2.lib.m17hip_gather_frames(c2.h, comm.h, 0, big.ctypes.data_as(C.c_void_p), C.c_uint64(big.size), cnt.ctypes.data_as(C.c_void_p), C.byref(tot))
    assert code == -5 and tot.value == 12 * 3 and int(cnt[0]) == 36
    full = _oracle_flat(x)
    first3 = np.concatenate([full[full["channel"] == ch][:3] for ch in range(12)])
    assert big[:36].tobytes() == first3.tobytes()
    # ... and before any run the collective still completes on every rank, with the call-sequence error as its result
    c3 = m17hip.Context(4, 4800)
    assert c3.lib.m17hip_gather_frames(c3.h, comm.h, 0, big.ctypes.data_as(C.c_void_p), C.c_uint64(big.size), None, C.byref(tot)) == -4
    c3.close(); c2.close()
    # fault injection (m17hip_tune key 30): a rank whose compaction fails, a root whose staging allocation fails — the call makes all
    # its collective calls and returns the failure; the communicator stays usable
    comm2 = m17hip.Comm(c, m17hip.comm_get_id(), 0, 1)
    c.tune(30, 2)
    assert c.lib.m17hip_gather_frames(c.h, comm2.h, 0, big.ctypes.data_as(C.c_void_p), C.c_uint64(big.size), None, C.byref(tot)) == -3   # ENOMEM on the root
    c.tune(30, 1)
    assert c.lib.m17hip_gather_frames(c.h, comm2.h, 0, big.ctypes.data_as(C.c_void_p), C.c_uint64(big.size), None, C.byref(tot)) == -2   # this rank's records: EHIP
    c.tune(30, 3)   # exchange 2 takes place and this rank's word of it cannot be written: its slot keeps the word of exchange 1 — no phase tag — and every rank leaves
    assert c.lib.m17hip_gather_frames(c.h, comm2.h, 0, big.ctypes.data_as(C.c_void_p), C.c_uint64(big.size), None, C.byref(tot)) == -2
    c.tune(30, 0)
    for _ in range(3):   # (serials go on; the staging buffer grown under fault 3 is in use)
        again, counts = c.gather_frames(comm2, root=0)
        assert again.tobytes() == recs.tobytes()
    comm2.close()
    comm.close(); c.close()


def test_bench_multi_gpu_code_path_with_one_rank(tmp_path):
    """bench.py's N > 1 branch (process group, one C-ABI communicator per batch in flight agreed on collectively, gather per step to
    rank 0, the ordered-and-unique check, both regimes) driven with a world of ONE rank: --force-gather --gather cabi."""
    import json, os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29617", RANK="0", LOCAL_RANK="0", WORLD_SIZE="1")
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--force-gather", "--gather", "cabi", "--channels", "64", "--samples", "48000",
                        "--steps", "2", "--warmup", "1", "--prewarm", "2", "--cpu-seconds", "0", "--h2d-steps", "0", "--parity-channels", "8"],
                       capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])   # (RCCL prints its version banner on stdout too)
    assert line["n_gpus"] == 1 and line["config"]["gather"].startswith("m17hip_gather_frames_device")
    assert line["config"]["gathered_set_ordered_and_unique"] is True and line["config"]["parity_vs_oracle_first_channels"] is True
    assert line["value"] > 0 and line["value_single_stream"] > 0


_WORKER = r"""
import os, sys, numpy as np
sys.path.insert(0, os.path.join({root!r}, "m17-cxx-demod_amd")); sys.path.insert(0, os.path.join({root!r}, "tests"))
import torch, torch.distributed as dist
import m17hip, oracle_lib as ol
from m17hip import dist as mdist
rank, world = int(sys.argv[1]), int(sys.argv[2])
os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = sys.argv[3]
dist.init_process_group("gloo", rank=rank, world_size=world)
CT, T = 20, 48000
p = ol.gen_params(seed=77, kind=-1, n_frames=20, lead_in=3072, noise_sigma=500.0, tail_sigma=500.0, lead_sigma=40000.0, total=T)
lo, hi = mdist.shard_range(CT, rank, world)
ctx = m17hip.Context(hi - lo, T)
ctx.set_channel_base(lo)
ctx.synth(p, hi - lo, T, chan0=lo)            # this shard's channels of the global job, generated on the device
ctx.reset(); ctx.run()
with torch.cuda.stream(ctx.torch_stream()):   # (the fill and the read-back on the context's own, non-blocking stream)
    buf = torch.zeros((hi - lo) * 64 * 64, dtype=torch.uint8, device="cuda")
    n = ctx.frames_compact_device(buf.data_ptr(), buf.numel() // 64)
    host = buf.cpu()
allrecs, counts = mdist.gather_records(host, n)       # gloo: host tensors (two ranks share the one GPU of this box)
np.save(os.path.join(sys.argv[4], f"rank{{rank}}.npy"), allrecs.numpy())
dist.barrier(); dist.destroy_process_group()
"""


def test_two_ranks_one_gpu_gather_equals_one_big_run(tmp_path):
    """Two processes (ranks) on cuda:0, each running the real HIP path on its shard with its channel base, records compacted
    on the device and all-gathered (gloo here: RCCL refuses two ranks on one device) == one 20-channel run."""
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    script = tmp_path / "worker.py"
    script.write_text(_WORKER.format(root=ROOT))
    procs = [subprocess.Popen([sys.executable, str(script), str(r), "2", str(port), str(tmp_path)]) for r in range(2)]
    for p in procs:
        assert p.wait(timeout=600) == 0
    CT, T = 20, 48000
    p = ol.gen_params(seed=77, kind=-1, n_frames=20, lead_in=3072, noise_sigma=500.0, tail_sigma=500.0, lead_sigma=40000.0, total=T)
    whole = m17hip.Context(CT, T)
    whole.synth(p, CT, T); whole.reset(); whole.run()
    one = whole.frames().copy()
    whole.close()
    assert one.size > CT
    for r in range(2):
        got = np.load(tmp_path / f"rank{r}.npy")
        assert got.tobytes() == one.tobytes(), r
