"""GPU tests of the streaming pipeline (include/m17hip.h m17hip_demod_front): consecutive runs of the SAME channels,
state carried, with the front end (matched filter, carrier-detect sums) of run k + 1 queued while the state-machine half of run k is
still at work.  The reference's use is one endless stream per channel (apps/m17-demod.cpp:484-490): whatever the chunking and
however the chunks are staged, the frame records must be those of the oracle over the whole stream, bit for bit."""
import ctypes as C

import numpy as np
import pytest

import m17hip
import oracle_lib as ol

pytestmark = pytest.mark.gpu

ESTATE, EINVAL = -4, -1


def _signals(Cn, T, seed=77, sigma=500.0, kind=-1, lead_in=3072, n_frames=None, tail_sigma=None):
    p = ol.gen_params(seed=seed, kind=kind, n_frames=max(1, T // 1920 - 4) if n_frames is None else n_frames, lead_in=lead_in, noise_sigma=sigma,
                      tail_sigma=sigma if tail_sigma is None else tail_sigma, lead_sigma=40000.0, total=T)
    return ol.generate_batch(p, Cn, T, threads=8)


def _oracle(x):
    recs, counts, diags = ol.demod_batch(x, cap=2 * (x.shape[1] // 1920 + 2) + 4, threads=8)
    flat = np.concatenate([recs[c, : counts[c]] for c in range(x.shape[0])]) if counts.sum() else recs[0, :0]
    return flat, diags


def _sorted(parts):
    got = np.concatenate(parts)
    return got[np.lexsort((got["seq"], got["channel"]))]


@pytest.fixture(scope="module", params=[{}, {15: 0, 17: 0}, {3: 7001}, {3: 9600, 15: 0}, {10: 0, 20: 1, 17: 0}, {26: 1, 3: 4800}, {26: 1, 3: 7001, 15: 0}],
                ids=["default", "decode_and_evm_in_k5", "seg7001", "seg9600_decode_in_k5", "k3_throughput_form_redo_in_front_evm_in_k5", "gate_aware_seg4800", "gate_aware_seg7001_decode_in_k5"])
def ctx(request):
    c = m17hip.Context(64, 48000)
    for k, v in request.param.items():
        c.tune(k, v)
    yield c
    c.close()


@pytest.fixture(params=["fetch_then_run", "run_then_fetch"])
def order(request):
    """The two call sequences of a live feed (include/m17hip.h): run k's records collected BEFORE run k + 1's state-machine half is queued
    (up to round 5 the only one), or AFTER it (m17hip_frames_select(1): nothing of run k + 1 waits for run k's payload work)."""
    return request.param


def _pipelined(ctx, Cn, lengths, stage, order="fetch_then_run"):
    """The call sequence of a live feed: stage(k + 1), front(k + 1), then fetch run k and run(k + 1) in either order."""
    ctx.reset()
    stage(0)
    ctx.run(channels=Cn, samples=lengths[0])
    parts = []
    for k in range(len(lengths)):
        if k + 1 < len(lengths):
            stage(k + 1)
            ctx.front(channels=Cn, samples=lengths[k + 1])
            if order == "run_then_fetch":
                ctx.run(channels=Cn, samples=lengths[k + 1])
                ctx.frames_select(1)               # run k's records, with run k + 1 queued behind it
        parts.append(ctx.frames().copy())          # still run k's records: neither the front end nor the next run touches them
        ctx.frames_select(0)                       # (the selection stays until the next run or until it is changed)
        if k + 1 < len(lengths) and order != "run_then_fetch":
            ctx.run(channels=Cn, samples=lengths[k + 1])
    return _sorted(parts)


def _check_diag(ctx, Cn, diags):
    d = ctx.diag(Cn)
    for f in ("dcd", "locked", "sample_index", "viterbi_cost", "n_diag", "demod_state", "n_frames", "evm", "deviation", "offset", "clock"):
        assert np.array_equal(d[f], diags[f], equal_nan=True), f


def test_pipelined_runs_from_pinned_host_memory(ctx, order):
    import torch
    Cn, T, n = 64, 24000, 5
    x = _signals(Cn, n * T, seed=101, sigma=600.0)
    exp, diags = _oracle(x)
    pinned = [torch.from_numpy(np.ascontiguousarray(x[:, k * T:(k + 1) * T])).pin_memory() for k in range(n)]
    got = _pipelined(ctx, Cn, [T] * n, lambda k: ctx.upload_async(pinned[k].data_ptr(), Cn, T), order)
    assert got.tobytes() == exp.tobytes() and got.size > 3 * Cn
    _check_diag(ctx, Cn, diags)


def test_pipelined_runs_from_device_memory(ctx, order):
    """m17hip_upload_i16_device_async: the chunks are handed over by a producer on the GPU (pitch > samples: a view into its buffer)."""
    import torch
    Cn, T, n = 48, 19200, 6
    x = _signals(Cn, n * T, seed=102, sigma=800.0)
    exp, diags = _oracle(x)
    dev = torch.from_numpy(x).cuda()               # [Cn][n * T]: chunk k = columns [k T, (k + 1) T), row pitch n * T
    torch.cuda.synchronize()                       # the producer's buffer is COMPLETE before it is handed over: a copy from pageable memory may still be in
                                                   # flight when .cuda() returns, and the context's copy stream does not wait for torch's (seen once in ~15 full runs)
    got = _pipelined(ctx, Cn, [T] * n, lambda k: ctx.upload_device_async(dev.data_ptr() + 2 * k * T, Cn, T, pitch=n * T), order)
    ctx.upload_wait()
    assert got.tobytes() == exp.tobytes() and got.size > 3 * Cn
    _check_diag(ctx, Cn, diags)


def test_two_resident_slabs_alternate_without_copies(ctx, order):
    """m17hip_input_alternate: slab A, slab B, A, B, ... — the regime of bench.py's single-stream leg.  The stream the channels see
    is A B A B A B; the oracle demodulates exactly that."""
    import torch
    Cn, T = 32, 48000
    a = _signals(Cn, T, seed=103, sigma=600.0)
    b = _signals(Cn, T, seed=104, sigma=900.0, kind=1)
    exp, diags = _oracle(np.concatenate([a, b, a, b, a, b], axis=1))
    pb = torch.from_numpy(b).pin_memory()

    def stage(k):
        if k == 0:
            ctx.upload(a)                          # in place: slab pair 0 ...
            return
        if k == 1:
            ctx.upload_async(pb.data_ptr(), Cn, T)  # ... slab pair 1 ...
            return
        ctx.input_alternate(Cn, T)                 # ... and from then on no copy at all

    ctx.reset()
    ctx.upload(a)
    ctx.run()
    parts = []
    for k in range(6):
        if k + 1 < 6:
            stage(k + 1)
            ctx.front(channels=Cn, samples=T)
            if order == "run_then_fetch":
                ctx.run(channels=Cn, samples=T)
                ctx.frames_select(1)
        parts.append(ctx.frames().copy())
        ctx.frames_select(0)
        if k + 1 < 6 and order != "run_then_fetch":
            ctx.run(channels=Cn, samples=T)
    got = _sorted(parts)
    assert got.tobytes() == exp.tobytes() and got.size > 6 * Cn
    _check_diag(ctx, Cn, diags)


def test_staged_run_after_an_in_place_overwrite_of_the_previous_slab(ctx):
    """a run in place, then the same slab overwritten in place (an input that is never run), then the NEXT chunk staged and run:
    the staged run's 152-sample prefix must be the tail the last RUN carried, not what the data region holds now."""
    import torch
    Cn, T = 16, 24000
    x = _signals(Cn, 2 * T, seed=131, sigma=600.0)
    exp, diags = _oracle(x)
    junk = np.full((Cn, T), 12345, dtype=np.int16)
    pinned = torch.from_numpy(np.ascontiguousarray(x[:, T:])).pin_memory()
    ctx.reset()
    ctx.upload(x[:, :T]); ctx.run(channels=Cn, samples=T)
    parts = [ctx.frames().copy()]
    ctx.upload(junk)                                   # overwrites the slab the run used (never run)
    ctx.upload_async(pinned.data_ptr(), Cn, T)         # the real continuation, staged
    ctx.run(channels=Cn, samples=T)
    parts.append(ctx.frames().copy())
    ctx.upload_wait()
    assert _sorted(parts).tobytes() == exp.tobytes()
    _check_diag(ctx, Cn, diags)


def test_pipelined_ragged_chunks_and_mixed_staging(ctx):
    """Chunks shorter than the carried prefixes (152 input samples, 96 filter outputs), not multiples of 8 / 192 / 1920, staged runs
    with and without m17hip_demod_front and in-place runs in between."""
    import torch
    Cn, T = 32, 40000
    x = _signals(Cn, T, seed=105, sigma=700.0)
    exp, diags = _oracle(x)
    ctx.reset()
    parts, pos, keep = [], 0, []
    for i, n in enumerate((1, 7, 95, 149, 153, 1919, 3841, 9601, 5000, 333, 12345, T)):
        n = min(n, T - pos)
        if n <= 0:
            break
        chunk = np.ascontiguousarray(x[:, pos: pos + n])
        mode = i % 3
        if mode == 0:                              # in place
            ctx.upload(chunk)
            ctx.run()
        else:
            pin = torch.from_numpy(chunk).pin_memory()
            keep.append(pin)
            ctx.upload_async(pin.data_ptr(), Cn, n)
            if mode == 1:
                ctx.front(channels=Cn, samples=n)
            ctx.run(channels=Cn, samples=n)
        parts.append(ctx.frames().copy())
        pos += n
    assert pos == T
    ctx.upload_wait()
    got = _sorted(parts)
    assert got.tobytes() == exp.tobytes()
    _check_diag(ctx, Cn, diags)


def test_pipelined_lost_sync_across_run_boundaries(ctx, order):
    """Bursts followed by loud noise: sync is lost, dcd.unlock() is forced (K2's speculation is dropped), the gated FIR restarts —
    with run boundaries falling anywhere in that."""
    import torch
    Cn, T, n = 32, 9600, 10
    x = _signals(Cn, n * T, seed=106, sigma=500.0, n_frames=14, tail_sigma=3000.0)
    exp, diags = _oracle(x)
    pinned = [torch.from_numpy(np.ascontiguousarray(x[:, k * T:(k + 1) * T])).pin_memory() for k in range(n)]
    got = _pipelined(ctx, Cn, [T] * n, lambda k: ctx.upload_async(pinned[k].data_ptr(), Cn, T), order)
    assert got.tobytes() == exp.tobytes() and got.size > Cn
    _check_diag(ctx, Cn, diags)


def test_front_call_sequence_errors():
    import torch
    Cn, T = 8, 9600
    x = _signals(Cn, 3 * T, seed=107)
    exp, _ = _oracle(x)
    c = m17hip.Context(Cn, T)
    lib, h = c.lib, c.h
    front = lambda C_, T_, fl=0: lib.m17hip_demod_front(h, C.c_uint32(C_), C.c_uint32(T_), C.c_uint32(fl))  # noqa: E731
    run = lambda C_, T_, fl=0: lib.m17hip_demod_run(h, C.c_uint32(C_), C.c_uint32(T_), C.c_uint32(fl))      # noqa: E731
    assert front(Cn, T) == ESTATE                                   # nothing staged
    assert lib.m17hip_input_alternate(h, C.c_uint32(Cn), C.c_uint32(T)) == ESTATE   # no second slab yet
    pins = [torch.from_numpy(np.ascontiguousarray(x[:, k * T:(k + 1) * T])).pin_memory() for k in range(3)]
    c.reset()
    c.upload_async(pins[0].data_ptr(), Cn, T)
    assert front(Cn, T // 2) == EINVAL                              # not the staged shape
    assert front(Cn, T, 2) == EINVAL                                # unknown flag
    assert front(Cn, T) == 0
    assert front(Cn, T) == ESTATE                                   # already queued
    assert lib.m17hip_upload_i16_async(h, C.c_void_p(pins[1].data_ptr()), C.c_uint32(Cn), C.c_uint32(T), C.c_size_t(T)) == ESTATE
    assert lib.m17hip_upload_i16(h, C.c_void_p(pins[1].data_ptr()), C.c_uint32(Cn), C.c_uint32(T), C.c_size_t(T)) == ESTATE
    assert lib.m17hip_fir_rrc150(h, C.c_uint32(Cn), C.c_uint32(T), C.c_uint32(0), None) == ESTATE
    assert lib.m17hip_tune(h, C.c_int(3), C.c_int64(4800)) == ESTATE
    assert run(Cn, T, 1) == ESTATE and run(Cn, T // 2) == ESTATE     # the run must be the one the front end was queued for
    assert run(Cn, T) == 0
    parts = [c.frames().copy()]
    # a reset abandons a queued front end; the stream then starts over and must decode as a fresh one
    c.upload_async(pins[1].data_ptr(), Cn, T)
    assert front(Cn, T) == 0
    c.reset()
    parts = []
    for k in range(3):
        c.upload_async(pins[k].data_ptr(), Cn, T)
        c.run(channels=Cn, samples=T)
        parts.append(c.frames().copy())
    c.upload_wait()
    assert _sorted(parts).tobytes() == exp.tobytes()
    c.close()


def test_pipelined_full_size_run_properties():
    """At bench size per channel (480 000 samples, ten segments per run) on 256 channels: three pipelined runs over alternating resident
    slabs == the same three runs made one after the other in a second context (records and diagnostics), and the first 8 channels
    == the oracle."""
    Cn, T = 256, 480000
    p = ol.gen_params(seed=20260101, kind=-1, n_frames=T // 1920 - 6, lead_in=3072, noise_sigma=600.0, tail_sigma=600.0, lead_sigma=40000.0, total=T)
    a, b = m17hip.Context(Cn, T), m17hip.Context(Cn, T)
    import torch
    for c_ in (a, b):
        c_.synth(p, Cn, T)
    x = a.download()
    # reference order: in place, one run after the other
    b.reset()
    seq_parts = []
    for k in range(3):
        b.run()
        seq_parts.append(b.frames().copy())
    # pipelined: the same slab in both slab pairs
    pin = torch.from_numpy(x).pin_memory()
    a.reset()
    a.run()
    parts = []
    for k in range(3):
        if k + 1 < 3:
            if k == 0:
                a.upload_async(pin.data_ptr(), Cn, T)
            else:
                a.input_alternate(Cn, T)
            a.front(channels=Cn, samples=T)
        parts.append(a.frames().copy())
        if k + 1 < 3:
            a.run(channels=Cn, samples=T)
    a.upload_wait()
    for k in range(3):
        assert parts[k].tobytes() == seq_parts[k].tobytes(), k
    assert a.diag(Cn).tobytes() == b.diag(Cn).tobytes()
    exp, _ = _oracle(np.tile(x[:8], (1, 3)))
    got = _sorted(parts)
    assert got[got["channel"] < 8].tobytes() == exp.tobytes()
    a.close(); b.close()


def test_the_evm_fold_can_move_between_runs_of_a_stream():
    """m17hip_tune key 17 between the runs of one stream: RunningStandardDeviation's state moves with the mode (deferred fold <-> inside
    the sequential kernel), the channels' m17_diag (evm included) ends up as the oracle's over the whole stream."""
    Cn, T, n = 64, 24000, 6
    x = _signals(Cn, n * T, seed=131, sigma=700.0)
    exp, diags = _oracle(x)
    c = m17hip.Context(Cn, T)
    parts = []
    for k, mode in enumerate((1, 0, 0, 1, 1, 0)):
        c.tune(17, mode)
        c.upload(x[:, k * T:(k + 1) * T]); c.run(); parts.append(c.frames().copy())
    assert _sorted(parts).tobytes() == exp.tobytes()
    _check_diag(c, Cn, diags)
    c.close()


def test_a_runs_last_evm_fold_pass_beside_the_next_runs_first_segment_leaves_its_marks_alone():
    """Found by tools/parity_sweep.py (seed 6082, round 6; the signals and cuts below are that configuration's): with the fold deferred (key 17)
    the last pass of run k rides the replay launch of run k + 1 and works BESIDE K5 of that run's first segment.  When it settled m17_diag.evm
    there, it could put run k's value over the mark that segment had just written — and a run whose last segment is too short to fire another
    callback (706 samples behind a 4800-sample segment) kept it: 4-5 of the 64 channels ended with a stale evm, records equal."""
    import torch
    Cn, T = 64, 96000
    rng = np.random.default_rng(6082)
    x = np.zeros((Cn, T), dtype=np.int16)
    for c in range(Cn):                                          # bursts of every kind and level with gaps, as the sweep makes them
        pos = 0
        while pos < T - 8000:
            n = min(int(rng.integers(6000, 40000)), T - pos)
            p = ol.gen_params(seed=int(rng.integers(1, 1 << 30)), kind=int(rng.choice([0, 1, 2, 4])), n_frames=int(rng.integers(1, 16)),
                              lead_in=int(rng.integers(0, 5000)), lead_sigma=float(rng.choice([0.0, 100.0, 1000.0, 10000.0, 40000.0])),
                              noise_sigma=float(rng.choice([0.0, 100.0, 500.0, 1200.0, 2500.0])), tail_sigma=float(rng.choice([0.0, 100.0, 1000.0, 5000.0])),
                              dc_offset=float(rng.choice([0.0, 0.0, 300.0, -2000.0, 6000.0])), gain=float(rng.choice([1.0, 0.3, 0.7, 1.6])),
                              phase=int(rng.integers(-1, 10)), invert=0, total=n)
            x[c, pos:pos + n] = ol.generate(p)[:n]; pos += n
    cuts = [0] + sorted(int(v) for v in rng.integers(1, T, size=2)) + [T]
    assert cuts == [0, 43854, 90494, 96000]
    lengths = [b - a for a, b in zip(cuts[:-1], cuts[1:])]
    exp, diags = _oracle(x)
    pinned = [torch.from_numpy(np.ascontiguousarray(x[:, a:b])).pin_memory() for a, b in zip(cuts[:-1], cuts[1:])]
    for order in ("fetch_then_run", "run_then_fetch"):
        c = m17hip.Context(Cn, T)
        for k, v in {15: 1, 3: 4800, 33: 0, 20: 0, 17: 1, 26: 1}.items():
            c.tune(k, v)
        for _ in range(3):                                       # (a race: more than one throw)
            got = _pipelined(c, Cn, lengths, lambda k: c.upload_async(pinned[k].data_ptr(), Cn, lengths[k]), order)
            c.upload_wait()
            assert got.tobytes() == exp.tobytes() and got.size > 4 * Cn
            _check_diag(c, Cn, diags)
        c.close()


def test_records_and_consumers_of_run_k_are_collected_after_run_k_plus_1_was_queued():
    """the deferred decode, the payload consumers, the compaction and the host's wait for them are off the chain of a continued
    stream — a run ends on the main stream with its state settled (settle_tail_kernel), everything else works on the payload stream on the
    record set of its own run, and the fetch family names the run before the latest after m17hip_frames_select(ctx, 1).  Six runs of
    BERT / voice / packet channels with the BERT and packet consumers on: records, PRBS9 statistics, reassembled packets and m17_diag
    equal the oracle's over the whole stream, with every run's records fetched only after the NEXT run was queued."""
    import torch
    Cn, T, n = 48, 28800, 6
    xs = [_signals(16, n * T, seed=141, sigma=600.0, kind=0), _signals(16, n * T, seed=142, sigma=600.0, kind=1),
          _signals(16, n * T, seed=143, sigma=500.0, kind=4, n_frames=20)]
    x = np.concatenate(xs)
    exp, diags = _oracle(x)
    c = m17hip.Context(Cn, T)
    c.tune(6, 1); c.tune(7, 256)
    assert c.lib.m17hip_frames_select(c.h, C.c_uint32(1)) == ESTATE      # no run yet
    assert c.lib.m17hip_frames_select(c.h, C.c_uint32(2)) == EINVAL
    pins = [torch.from_numpy(np.ascontiguousarray(x[:, k * T:(k + 1) * T])).pin_memory() for k in range(n)]
    c.reset()
    c.upload_async(pins[0].data_ptr(), Cn, T)
    c.run(channels=Cn, samples=T)
    assert c.lib.m17hip_frames_select(c.h, C.c_uint32(1)) == ESTATE      # one run: there is no run before it
    parts, counts, pkts = [], [], []
    for k in range(n):
        if k + 1 < n:
            c.upload_async(pins[k + 1].data_ptr(), Cn, T)
            c.front(channels=Cn, samples=T)
            c.run(channels=Cn, samples=T)
            assert c.frames_count() >= 0                                  # (run k + 1's own count: selected by default)
            latest = c.frames_count()
            c.frames_select(1)
        counts.append(c.frames_count())
        parts.append(c.frames().copy())
        pkts.append(c.packets().copy())                                   # the packets run k completed (the selected run's store)
        assert parts[-1].size == counts[-1]
        if k + 1 < n:
            c.frames_select(0)
            assert c.frames_count() == latest
    c.upload_wait()
    got = _sorted(parts)
    assert got.tobytes() == exp.tobytes() and got.size > 4 * Cn
    _check_diag(c, Cn, diags)
    # the packets, run by run, are those of the same runs made strictly one after the other in a second context
    d = m17hip.Context(Cn, T)
    d.tune(7, 256)
    d.reset()
    npk = 0
    for k in range(n):
        d.upload(x[:, k * T:(k + 1) * T]); d.run()
        one = d.packets()
        assert one.tobytes() == pkts[k].tobytes(), k
        npk += one.size
    assert npk >= 16      # every packet channel completed its packet somewhere in the stream
    d.close()
    # the consumers worked run by run on their own run's records: PRBS9 statistics == the oracle's receiver over the oracle's BERT frames
    st = c.bert_stats(Cn)
    for ch in range(16):
        r = exp[exp["channel"] == ch]
        bert = r[r["frame_type"] == 5]
        bits, errs, sync = ol.bert_count(bert["payload"][:, :25]) if bert.size else (0, 0, False)
        assert (int(st["bits"][ch]), int(st["errors"][ch]), bool(st["synced"][ch]), int(st["frames"][ch])) == (bits, errs, sync, bert.size), ch
    # after a reset nothing of the old stream can be selected
    c.reset()
    assert c.lib.m17hip_frames_select(c.h, C.c_uint32(0)) == ESTATE and c.lib.m17hip_frames_select(c.h, C.c_uint32(1)) == ESTATE
    got0 = C.c_uint64(0)
    assert c.lib.m17hip_frames_count(c.h, C.byref(got0)) == ESTATE
    c.close()


def test_gather_names_the_selected_run():
    """m17hip_gather_frames after m17hip_frames_select(ctx, 1): the records of the run BEFORE the latest travel (one-rank communicator)."""
    Cn, T = 12, 24000
    x = _signals(Cn, 2 * T, seed=151)
    c = m17hip.Context(Cn, T)
    comm = m17hip.Comm(c, m17hip.comm_get_id(), 0, 1)
    c.reset()
    c.upload(x[:, :T]); c.run()
    first = c.frames().copy()
    c.upload(x[:, T:]); c.run()
    second = c.frames().copy()
    assert first.size and second.size and first.tobytes() != second.tobytes()
    recs, counts = c.gather_frames(comm, root=0)
    assert recs.tobytes() == second.tobytes()
    c.frames_select(1)
    recs, counts = c.gather_frames(comm, root=0)
    assert recs.tobytes() == first.tobytes() and counts.tolist() == [first.size]
    comm.close(); c.close()
