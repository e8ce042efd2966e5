"""Python (ctypes) binding of the C ABI in include/m17hip.h — the MI355X-native M17 demodulation hot path.

This is plumbing over `libm17hip.so` (hand-written HIP kernels for gfx950): no CPU fallback exists and
none is attempted — if the library is missing or a HIP call fails, a `M17HipError` is raised.
Nothing here imports the oracle.
"""
import ctypes as C
import os

import numpy as np

# The HIP runtime reads GPU_MAX_HW_QUEUES when it initialises (a context's five streams must not share hardware queues: include/m17hip.h,
# m17hip_advice): ask for 16 unless the host has decided otherwise.  Effective when this import comes before the process's first contact with
# the GPU; m17hip_ctx_create refuses (M17HIP_ECONFIG) when fewer than 8 were asked for.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("M17HIP_LIB") or os.path.join(os.path.dirname(_HERE), "libm17hip.so")   # (M17HIP_LIB: an experiment build of the same library)

FRAME_REC = np.dtype(
    [("channel", "<u4"), ("seq", "<u4"), ("sample_pos", "<u8"), ("cost", "<i4"), ("frame_type", "u1"), ("sync_type", "u1"),
     ("len", "u1"), ("flags", "u1"), ("payload", "u1", (32,)), ("pad", "u1", (8,))]
)
DIAG = np.dtype(
    [("dcd", "<i4"), ("evm", "<f4"), ("deviation", "<f4"), ("offset", "<f4"), ("locked", "<i4"), ("clock", "<f4"),
     ("sample_index", "<i4"), ("sync_index", "<i4"), ("clock_index", "<i4"), ("viterbi_cost", "<i4"), ("dcd_level", "<f4"),
     ("n_diag", "<u4"), ("demod_state", "<u4"), ("n_frames", "<u4"), ("pad", "<u4", (2,))]
)
assert FRAME_REC.itemsize == 64 and DIAG.itemsize == 64

FLAG_INVERT = 1
KERNELS = {"fir_rrc150": 0, "dcd": 1, "demod_seq": 2, "decode": 3, "correlator": 4, "compact": 5, "limit_track": 6}
LSF_INFO = np.dtype([("dst", "S10"), ("src", "S10"), ("type", "<u2"), ("crc_ok", "u1"), ("reserved", "u1", (9,))])
BERT_STAT = np.dtype([("bits", "<u4"), ("errors", "<u4"), ("synced", "<u4"), ("frames", "<u4")])
PACKET_REC = np.dtype([("channel", "<u4"), ("seq", "<u4"), ("sample_pos", "<u8"), ("size", "<u2"), ("checksum", "<u2"), ("crc_ok", "u1"),
                       ("frames", "u1"), ("seq_errors", "u1"), ("reserved", "u1"), ("data", "u1", (840,))])
assert PACKET_REC.itemsize == 864
VITERBI_SHAPES = {0: (488, 240), 1: (296, 144), 2: (420, 206), 3: (402, 197)}

EXPORTS = [
    "m17hip_strerror", "m17hip_last_hip_error", "m17hip_version", "m17hip_ctx_create", "m17hip_ctx_destroy", "m17hip_set_stream", "m17hip_get_stream",
    "m17hip_upload_i16", "m17hip_upload_i16_device", "m17hip_upload_i16_async", "m17hip_synth_i16", "m17hip_download_i16", "m17hip_fir_rrc150", "m17hip_correlator", "m17hip_fir_correlator", "m17hip_dcd", "m17hip_viterbi",
    "m17hip_slice_llr", "m17hip_decode_frames", "m17hip_demod_reset", "m17hip_demod_run", "m17hip_frames_count", "m17hip_frames_fetch",
    "m17hip_frames_compact_device", "m17hip_diag_fetch", "m17hip_bert_stats", "m17hip_packets_fetch", "m17hip_packets_feed", "m17hip_lsf_info", "m17hip_tune", "m17hip_debug_counters", "m17hip_timing_enable", "m17hip_timing_get", "m17hip_timing_reset",
    "m17hip_set_kalman_order", "m17hip_kalman_trace", "m17hip_set_channel_base", "m17hip_upload_wait", "m17hip_comm_get_id", "m17hip_comm_create",
    "m17hip_comm_destroy", "m17hip_comm_last_error", "m17hip_gather_frames", "m17hip_gather_frames_device", "m17hip_diag_log_fetch",
    "m17hip_upload_i16_device_async", "m17hip_input_alternate", "m17hip_demod_front", "m17hip_advice", "m17hip_replay_drops", "m17hip_frames_select",
]
ETRUNC = -6
COMM_ID_BYTES = 128


class M17HipError(RuntimeError):
    pass


_lib = None


def load_library():
    """Load libm17hip.so (built by `__graft_entry__.build()` / csrc/Makefile).  Raises if it is absent."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise M17HipError(f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'`; "
                              "there is no CPU fallback for the demodulation hot path")
        lib = C.CDLL(LIB_PATH)
        lib.m17hip_strerror.restype = C.c_char_p
        lib.m17hip_comm_destroy.restype = None
        lib.m17hip_ctx_destroy.restype = None
        _lib = lib
    return _lib


def _ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def comm_get_id():
    """128-byte RCCL id (rank 0 calls this and hands the bytes to the other ranks)."""
    buf = C.create_string_buffer(COMM_ID_BYTES)
    code = load_library().m17hip_comm_get_id(buf)
    if code != 0:
        raise M17HipError(f"m17hip_comm_get_id: {load_library().m17hip_strerror(C.c_int(code)).decode()}")
    return buf.raw


class Comm:
    """RCCL communicator of one rank (one context = one GPU = one rank); creation is collective."""

    def __init__(self, ctx, comm_id, rank, nranks):
        self.lib, self.ctx, self.rank, self.nranks = ctx.lib, ctx, int(rank), int(nranks)
        self.h = C.c_void_p()
        ctx._chk(self.lib.m17hip_comm_create(ctx.h, C.c_char_p(bytes(comm_id)), C.c_int(rank), C.c_int(nranks), C.byref(self.h)))

    def close(self):
        if self.h:
            self.lib.m17hip_comm_destroy(self.h)
            self.h = C.c_void_p()


class Context:
    """One demodulation context = `channels` independent 48 kSPS channels on one GPU (include/m17hip.h)."""

    _warned = False

    def __init__(self, max_channels, max_samples, device=0, stream=None):
        self.lib = load_library()
        self.h = C.c_void_p()
        self.max_channels, self.max_samples = int(max_channels), int(max_samples)
        self._chk(self.lib.m17hip_ctx_create(C.c_int(device), C.c_uint32(max_channels), C.c_uint32(max_samples), C.byref(self.h)))
        if stream is not None:
            self.set_stream(stream)
        if not Context._warned and self.lib.m17hip_advice(self.h) & 3:
            Context._warned = True
            import warnings
            warnings.warn("m17hip: fewer than 16 hardware queues were asked for (GPU_MAX_HW_QUEUES) — with several contexts the streams will share "
                          "hardware queues and serialise; export GPU_MAX_HW_QUEUES=16 before the process touches the GPU (include/m17hip.h)")

    def _chk(self, code):
        if code != 0:
            hip = self.lib.m17hip_last_hip_error(self.h) if self.h else 0
            raise M17HipError(f"m17hip error {code}: {self.lib.m17hip_strerror(C.c_int(code)).decode()} (hip error {hip})")

    def close(self):
        if self.h:
            self.lib.m17hip_ctx_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_stream(self, stream_handle):
        self._chk(self.lib.m17hip_set_stream(self.h, C.c_void_p(int(stream_handle))))

    @property
    def stream(self):
        """The context's main stream (a hipStream_t as an int): the library's own unless set_stream named another (m17hip_get_stream)."""
        h = C.c_void_p()
        self._chk(self.lib.m17hip_get_stream(self.h, C.byref(h)))
        return h.value or 0

    def torch_stream(self):
        """The main stream as a torch.cuda.ExternalStream: `with torch.cuda.stream(ctx.torch_stream()): ...` puts the host's tensor work in order
        with the context's."""
        import torch
        return torch.cuda.ExternalStream(self.stream)

    # ---- input -------------------------------------------------------------------------------------------------
    def upload(self, samples):
        s = np.ascontiguousarray(samples, dtype=np.int16)
        if s.ndim == 1:
            s = s[None, :]
        self.C, self.T = s.shape
        self._chk(self.lib.m17hip_upload_i16(self.h, _ptr(s), C.c_uint32(self.C), C.c_uint32(self.T), C.c_size_t(self.T)))

    def upload_async(self, host_ptr, channels, samples, pitch=None):
        """Stage the input of the NEXT run (pinned host memory at `host_ptr`, kept alive by the caller) while the current run computes."""
        self.C, self.T = int(channels), int(samples)
        self._chk(self.lib.m17hip_upload_i16_async(self.h, C.c_void_p(int(host_ptr)), C.c_uint32(self.C), C.c_uint32(self.T),
                                                   C.c_size_t(self.T if pitch is None else pitch)))

    def upload_device_async(self, dev_ptr, channels, samples, pitch=None):
        """Stage the input of the NEXT run from device memory at `dev_ptr` (kept alive and unmodified by the caller until upload_wait)."""
        self.C, self.T = int(channels), int(samples)
        self._chk(self.lib.m17hip_upload_i16_device_async(self.h, C.c_void_p(int(dev_ptr)), C.c_uint32(self.C), C.c_uint32(self.T),
                                                          C.c_size_t(self.T if pitch is None else pitch)))

    def input_alternate(self, channels=None, samples=None):
        """Stage, without a copy, the input the context's second slab still holds (two resident slabs that alternate)."""
        self._chk(self.lib.m17hip_input_alternate(self.h, C.c_uint32(channels or self.C), C.c_uint32(samples or self.T)))

    def front(self, flags=0, channels=None, samples=None):
        """Queue the front end (matched filter, carrier-detect sums) of the STAGED run now, beside the latest run's state-machine half;
        the run() that follows (same arguments) queues the rest."""
        self._chk(self.lib.m17hip_demod_front(self.h, C.c_uint32(channels or self.C), C.c_uint32(samples or self.T), C.c_uint32(flags)))

    def upload_wait(self):
        """Block until the copy queued by upload_async has left the host buffer."""
        self._chk(self.lib.m17hip_upload_wait(self.h))

    def synth(self, params, channels, samples, chan0=0):
        """Generate the input slab on the device (m17-mod framing + impairments); `params` = a ctypes block laid out as m17_synth_params."""
        self.C, self.T = int(channels), int(samples)
        self._chk(self.lib.m17hip_synth_i16(self.h, C.byref(params), C.c_uint32(self.C), C.c_uint32(self.T), C.c_uint32(chan0)))

    def download(self):
        out = np.empty((self.C, self.T), dtype=np.int16)
        self._chk(self.lib.m17hip_download_i16(self.h, _ptr(out), C.c_uint32(self.C), C.c_uint32(self.T), C.c_size_t(self.T)))
        return out

    def upload_device(self, dev_ptr, channels, samples, pitch=None):
        self.C, self.T = int(channels), int(samples)
        self._chk(self.lib.m17hip_upload_i16_device(self.h, C.c_void_p(int(dev_ptr)), C.c_uint32(self.C), C.c_uint32(self.T),
                                                    C.c_size_t(self.T if pitch is None else pitch)))

    # ---- per-operator entry points --------------------------------------------------------------------------------
    def fir(self, flags=0, fetch=True):
        out = np.empty((self.C, self.T), dtype=np.float32) if fetch else None
        self._chk(self.lib.m17hip_fir_rrc150(self.h, C.c_uint32(self.C), C.c_uint32(self.T), C.c_uint32(flags), _ptr(out)))
        return out

    def correlator(self):
        limit = np.empty((self.C, self.T), dtype=np.float32)
        corr = np.empty((4, self.C, self.T), dtype=np.float32)
        self._chk(self.lib.m17hip_correlator(self.h, C.c_uint32(self.C), C.c_uint32(self.T), _ptr(limit), _ptr(corr)))
        return limit, corr

    def fir_correlator(self, flags=0, fetch=True):
        """configs[1] as one call: matched filter + limit filter + the four correlations, pipelined in time (m17hip_fir_correlator)."""
        if not fetch:
            self._chk(self.lib.m17hip_fir_correlator(self.h, C.c_uint32(self.C), C.c_uint32(self.T), C.c_uint32(flags), None, None, None))
            return None
        y = np.empty((self.C, self.T), dtype=np.float32)
        limit = np.empty((self.C, self.T), dtype=np.float32)
        corr = np.empty((4, self.C, self.T), dtype=np.float32)
        self._chk(self.lib.m17hip_fir_correlator(self.h, C.c_uint32(self.C), C.c_uint32(self.T), C.c_uint32(flags), _ptr(y), _ptr(limit), _ptr(corr)))
        return y, limit, corr

    def correlator_device(self):
        """Same computation, results left in device memory (benchmarks)."""
        self._chk(self.lib.m17hip_correlator(self.h, C.c_uint32(self.C), C.c_uint32(self.T), None, None))

    def dcd(self, flags=0, fetch=True, samples=None):
        T = int(samples) if samples is not None else self.T
        ticks = T // 192
        sums = np.empty((self.C, ticks, 2, 6), dtype=np.float32) if fetch else None
        n = C.c_uint32(0)
        self._chk(self.lib.m17hip_dcd(self.h, C.c_uint32(self.C), C.c_uint32(T), C.c_uint32(flags), _ptr(sums), C.byref(n)))
        assert n.value == ticks
        return sums

    def viterbi(self, soft, kind):
        IN, OUT = VITERBI_SHAPES[kind]
        s = np.ascontiguousarray(soft, dtype=np.int8).reshape(-1, IN)
        n = s.shape[0]
        bits = np.empty((n, OUT), dtype=np.uint8)
        cost = np.empty(n, dtype=np.int32)
        self._chk(self.lib.m17hip_viterbi(self.h, _ptr(s), C.c_uint32(n), C.c_int(kind), _ptr(bits), _ptr(cost)))
        return bits, cost

    def slice_llr(self, sym):
        s = np.ascontiguousarray(sym, dtype=np.float32)
        if s.ndim == 1:
            s = s[None, :]
        rows, n = s.shape
        llr = np.empty((rows, n, 2), dtype=np.int8)
        evm = np.empty((rows, n), dtype=np.float32)
        self._chk(self.lib.m17hip_slice_llr(self.h, _ptr(s), C.c_uint32(rows), C.c_uint32(n), _ptr(llr), _ptr(evm)))
        return llr, evm

    def decode_frames(self, llr368, sync_type, state=None, lich=None, lsf=None, dep401=None, cost=None):
        l = np.ascontiguousarray(llr368, dtype=np.int8).reshape(-1, 368)
        n = l.shape[0]
        st = np.ascontiguousarray(sync_type, dtype=np.uint8)
        state = np.zeros(n, np.uint8) if state is None else np.ascontiguousarray(state, dtype=np.uint8).copy()
        lich = np.zeros(n, np.uint8) if lich is None else np.ascontiguousarray(lich, dtype=np.uint8).copy()
        lsf = np.zeros((n, 30), np.uint8) if lsf is None else np.ascontiguousarray(lsf, dtype=np.uint8).copy()
        dep401 = np.zeros(n, np.int8) if dep401 is None else np.ascontiguousarray(dep401, dtype=np.int8).copy()
        cost = np.zeros(n, np.int64) if cost is None else np.ascontiguousarray(cost, dtype=np.int64).copy()
        recs = np.zeros((n, 2), dtype=FRAME_REC)
        nrec = np.zeros(n, dtype=np.uint8)
        self._chk(self.lib.m17hip_decode_frames(self.h, _ptr(l), C.c_uint32(n), _ptr(st), _ptr(state), _ptr(lich), _ptr(lsf), _ptr(dep401),
                                                _ptr(cost), _ptr(recs), _ptr(nrec)))
        return recs, nrec, state, lich, lsf, dep401, cost

    def kalman_trace(self, z, dt, wrap, z0=0.0, order=3):
        """rows x n updates of the 2-state Kalman filter (kal_update, as the full-chain kernel runs it): state after each update."""
        zz = np.ascontiguousarray(z, dtype=np.float32)
        if zz.ndim == 1:
            zz = zz[None, :]
        dd = np.ascontiguousarray(np.broadcast_to(dt, zz.shape), dtype=np.uint32)
        out = np.empty(zz.shape + (6,), dtype=np.float32)
        self._chk(self.lib.m17hip_kalman_trace(self.h, _ptr(zz), _ptr(dd), C.c_uint32(zz.shape[0]), C.c_uint32(zz.shape[1]), C.c_int(wrap),
                                               C.c_float(z0), C.c_int(order), _ptr(out)))
        return out

    # ---- the full chain ----------------------------------------------------------------------------------------------
    def set_kalman_order(self, order):
        """Evaluation order of the Kalman updates (bit set, include/m17hip.h); default 3."""
        self._chk(self.lib.m17hip_set_kalman_order(self.h, C.c_int(order)))

    def set_channel_base(self, base):
        """Global id of this context's channel 0: frame records carry channel = base + local index."""
        self._chk(self.lib.m17hip_set_channel_base(self.h, C.c_uint32(base)))

    def gather_frames(self, comm, root=0, capacity=None):
        """Collective: the records of the last run of every rank, gathered to `root` over RCCL in rank (= channel) order.
        Returns (records or None off the root, counts per rank)."""
        counts = np.zeros(comm.nranks, dtype=np.uint64)
        total = C.c_uint64(0)
        is_root = comm.rank == root
        if capacity is None:
            capacity = comm.nranks * self.max_channels * (2 * (self.max_samples // 1920 + 2) + 4) if is_root else 0
        recs = np.empty(capacity, dtype=FRAME_REC) if is_root else None
        self._chk(self.lib.m17hip_gather_frames(self.h, comm.h, C.c_int(root), _ptr(recs), C.c_uint64(capacity), _ptr(counts), C.byref(total)))
        return (recs[: total.value] if is_root else None), counts

    def gather_frames_device(self, comm, dev_ptr, capacity, root=0):
        """Collective: like gather_frames, the root's copy written to device memory at dev_ptr.  Returns (total, counts per rank)."""
        counts = np.zeros(comm.nranks, dtype=np.uint64)
        total = C.c_uint64(0)
        self._chk(self.lib.m17hip_gather_frames_device(self.h, comm.h, C.c_int(root), C.c_void_p(int(dev_ptr)), C.c_uint64(capacity), _ptr(counts),
                                                       C.byref(total)))
        return total.value, counts

    def reset(self):
        self._chk(self.lib.m17hip_demod_reset(self.h))

    def run(self, flags=0, channels=None, samples=None):
        self._chk(self.lib.m17hip_demod_run(self.h, C.c_uint32(channels or self.C), C.c_uint32(samples or self.T), C.c_uint32(flags)))

    def frames_select(self, back):
        """Which run's records frames_count / frames / frames_compact_device / gather_frames name: 0 = the latest run (every run() selects it
        again), 1 = the run before it (a live feed collects run k after it has queued run k + 1)."""
        self._chk(self.lib.m17hip_frames_select(self.h, C.c_uint32(back)))

    def frames_count(self):
        n = C.c_uint64(0)
        self._chk(self.lib.m17hip_frames_count(self.h, C.byref(n)))
        return n.value

    def frames(self):
        """Records of the last run, ordered by (channel, seq).  One compaction and one synchronisation when the guessed capacity (the
        previous fetch's count plus a margin) suffices; M17HIP_ETRUNC reports the real count and the fetch is repeated once."""
        cap = max(1024, getattr(self, "_last_frames", 0) * 5 // 4 + 64)
        while True:
            recs = np.zeros(cap, dtype=FRAME_REC)
            got = C.c_uint64(0)
            code = self.lib.m17hip_frames_fetch(self.h, _ptr(recs), C.c_uint64(recs.size), C.byref(got))
            if code == ETRUNC and got.value > cap:
                cap = int(got.value)
                continue
            self._chk(code)
            self._last_frames = int(got.value)
            return recs[: got.value]

    def frames_compact_device(self, dev_ptr, capacity):
        n = C.c_uint64(0)
        self._chk(self.lib.m17hip_frames_compact_device(self.h, C.c_void_p(int(dev_ptr)), C.c_uint64(capacity), C.byref(n)))
        return n.value

    def diag(self, channels=None):
        n = channels or self.C
        d = np.zeros(n, dtype=DIAG)
        self._chk(self.lib.m17hip_diag_fetch(self.h, _ptr(d), C.c_uint32(n)))
        return d

    def diag_log(self, channels=None, capacity=None):
        """Every diagnostic callback of the last run per channel (enable with tune(9, room) before the run): list of DIAG arrays."""
        n = channels or self.C
        cap = capacity or (self.T // 384 + 2)
        log = np.zeros((n, cap), dtype=DIAG)
        counts = np.zeros(n, dtype=np.uint32)
        self._chk(self.lib.m17hip_diag_log_fetch(self.h, _ptr(log), _ptr(counts), C.c_uint32(n), C.c_uint32(cap)))
        return [log[c, : counts[c]] for c in range(n)]

    def lsf_info(self, lsf30):
        """Callsigns, type field and CRC status of a batch of 30-byte link setup frames."""
        a = np.ascontiguousarray(lsf30, dtype=np.uint8).reshape(-1, 30)
        out = np.zeros(a.shape[0], dtype=LSF_INFO)
        self._chk(self.lib.m17hip_lsf_info(self.h, _ptr(a), C.c_uint32(a.shape[0]), _ptr(out)))
        return out

    def bert_stats(self, channels=None):
        """PRBS9 bit / error counts per channel over the BERT frames since reset (enable with tune(6, 1) before the runs)."""
        n = channels or self.C
        st = np.zeros(n, dtype=BERT_STAT)
        self._chk(self.lib.m17hip_bert_stats(self.h, _ptr(st), C.c_uint32(n)))
        return st

    def packets(self, capacity=4096):
        """Packets the last run completed, ordered by (channel, seq) (enable with tune(7, room) before the runs)."""
        out = np.zeros(capacity, dtype=PACKET_REC)
        n = C.c_uint32(0)
        self._chk(self.lib.m17hip_packets_fetch(self.h, _ptr(out), C.c_uint32(capacity), C.byref(n)))
        return out[: min(n.value, capacity)]

    def packets_feed(self, recs2d, counts):
        """Run the packet consumer over caller-supplied frame records [channels][pitch] (counts[c] used per row)."""
        r = np.ascontiguousarray(recs2d, dtype=FRAME_REC)
        n = np.ascontiguousarray(counts, dtype=np.uint32)
        self._chk(self.lib.m17hip_packets_feed(self.h, _ptr(r), _ptr(n), C.c_uint32(r.shape[0]), C.c_uint32(r.shape[1])))

    def replay_drops(self):
        """Times a channel left the limit-filter replay since the last reset (m17hip_replay_drops)."""
        n = C.c_uint64(0)
        self._chk(self.lib.m17hip_replay_drops(self.h, C.byref(n)))
        return int(n.value)

    def tune(self, key, value):
        self._chk(self.lib.m17hip_tune(self.h, C.c_int(key), C.c_int64(value)))

    def debug_counters(self, max_waves=4096):
        buf = np.zeros((max_waves, 40), dtype=np.uint64)   # (csrc/m17_state.hpp DBG_SLOTS)
        n = C.c_uint32(0)
        self._chk(self.lib.m17hip_debug_counters(self.h, _ptr(buf), C.c_uint32(max_waves), C.byref(n)))
        return buf[: n.value]

    # ---- measurement ---------------------------------------------------------------------------------------------------
    def timing(self, on=True):
        self._chk(self.lib.m17hip_timing_enable(self.h, C.c_int(1 if on else 0)))

    def timing_reset(self):
        self._chk(self.lib.m17hip_timing_reset(self.h))

    def timing_get(self, kernel):
        ms, n = C.c_double(0), C.c_uint64(0)
        self._chk(self.lib.m17hip_timing_get(self.h, C.c_int(KERNELS[kernel]), C.byref(ms), C.byref(n)))
        return ms.value, n.value
