"""ctypes access to the CPU oracle (oracle/libm17oracle.so) and, when it was built in the
build container, to the reference-header shim (oracle/_ref/libm17ref.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg.  The product (m17-cxx-demod_amd/) never imports this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")

FRAME_REC = np.dtype(
    [
        ("channel", "<u4"),
        ("seq", "<u4"),
        ("sample_pos", "<u8"),
        ("cost", "<i4"),
        ("frame_type", "u1"),
        ("sync_type", "u1"),
        ("len", "u1"),
        ("flags", "u1"),
        ("payload", "u1", (32,)),
        ("pad", "u1", (8,)),
    ]
)
assert FRAME_REC.itemsize == 64

DIAG = np.dtype(
    [
        ("dcd", "<i4"),
        ("evm", "<f4"),
        ("deviation", "<f4"),
        ("offset", "<f4"),
        ("locked", "<i4"),
        ("clock", "<f4"),
        ("sample_index", "<i4"),
        ("sync_index", "<i4"),
        ("clock_index", "<i4"),
        ("viterbi_cost", "<i4"),
        ("dcd_level", "<f4"),
        ("n_diag", "<u4"),
        ("demod_state", "<u4"),
        ("n_frames", "<u4"),
        ("pad", "<u4", (2,)),
    ]
)
assert DIAG.itemsize == 64


class GenParams(C.Structure):
    _fields_ = [
        ("seed", C.c_uint64),
        ("kind", C.c_int32),
        ("n_frames", C.c_int32),
        ("lead_in", C.c_int32),
        ("phase", C.c_int32),
        ("tail", C.c_int32),
        ("total", C.c_int32),
        ("invert", C.c_int32),
        ("n_preamble", C.c_int32),
        ("lead_sigma", C.c_double),
        ("noise_sigma", C.c_double),
        ("dc_offset", C.c_double),
        ("gain", C.c_double),
        ("tail_sigma", C.c_double),
    ]


def gen_params(seed=1, kind=0, n_frames=8, lead_in=0, phase=-1, tail=0, total=0, invert=0, lead_sigma=20000.0,
               noise_sigma=0.0, dc_offset=0.0, gain=1.0, tail_sigma=0.0, n_preamble=0):
    return GenParams(seed, kind, n_frames, lead_in, phase, tail, total, invert, n_preamble, lead_sigma, noise_sigma, dc_offset,
                     gain, tail_sigma)


def build_oracle():
    """Compile the oracle's C++ restatement (and oracle/_ref when the reference is present)."""
    subprocess.run(["make", "-s", "-C", ORACLE_DIR], check=True)


_oracle = None
_ref = None


def _p(a, t=None):
    return a.ctypes.data_as(C.c_void_p)


def oracle():
    global _oracle
    if _oracle is None:
        path = os.environ.get("M17_ORACLE_LIB") or os.path.join(ORACLE_DIR, "libm17oracle.so")   # (M17_ORACLE_LIB: the sanitizer build, tools/sanitize_cpu.sh)
        if not os.path.exists(path):
            build_oracle()
        lib = C.CDLL(path)
        for name in ("m17o_demod", "m17o_demod_symbols", "m17o_generate", "m17o_dcd_trace", "m17o_viterbi",
                     "m17o_puncture", "m17o_depuncture", "m17o_conv_encode", "m17o_qpp"):
            getattr(lib, name).restype = C.c_size_t
        lib.m17o_crc16.restype = C.c_uint16
        lib.m17o_golay_encode24.restype = C.c_uint32
        lib.m17o_golay_decode.restype = C.c_int
        lib.m17o_decode_frame.restype = C.c_int
        _oracle = lib
    return _oracle


def ref():
    """The reference-header shim, or None when oracle/_ref was not built (e.g. on the GPU box without it)."""
    global _ref
    if _ref is None:
        path = os.path.join(ORACLE_DIR, "_ref", "libm17ref.so")
        if not os.path.exists(path):
            return None
        lib = C.CDLL(path)
        for name in ("ref_dcd_trace", "ref_viterbi", "ref_depuncture"):
            getattr(lib, name).restype = C.c_size_t
        lib.ref_crc16.restype = C.c_uint16
        lib.ref_golay_encode24.restype = C.c_uint32
        lib.ref_golay_decode.restype = C.c_int
        lib.ref_decode_frame.restype = C.c_int
        _ref = lib
    return _ref


# ---------------------------------------------------------------- convenience wrappers (oracle) --
def generate(params, with_truth=False):
    lib = oracle()
    n = lib.m17o_generate(C.byref(params), None, C.c_size_t(0), None, None, None)
    out = np.zeros(n, dtype=np.int16)
    nf = max(1, params.n_frames)
    payloads = np.zeros((nf, 32), dtype=np.uint8)
    lsf = np.zeros(30, dtype=np.uint8)
    bs = C.c_int32(0)
    lib.m17o_generate(C.byref(params), _p(out), C.c_size_t(n), _p(payloads), _p(lsf), C.byref(bs))
    if with_truth:
        return out, dict(payloads=payloads, lsf=lsf, burst_start=bs.value)
    return out


def generate_batch(params, channels, total, threads=8, chan0=0):
    lib = oracle()
    out = np.zeros((channels, total), dtype=np.int16)
    lib.m17o_generate_batch(C.byref(params), C.c_size_t(channels), C.c_size_t(total), C.c_size_t(total), C.c_int(threads),
                            _p(out), C.c_uint32(chan0))
    return out


def make_frames(kind, seed, n_frames):
    """On-air 368-bit frames of a synthetic burst and the sync type of each (see m17o_make_frames)."""
    lib = oracle()
    lib.m17o_make_frames.restype = C.c_size_t
    bits = np.zeros((n_frames + 1, 368), dtype=np.int8)
    st = np.zeros(n_frames + 1, dtype=np.uint8)
    n = lib.m17o_make_frames(C.c_int(kind), C.c_uint64(seed), C.c_int(n_frames), _p(bits), _p(st))
    return bits[:n].copy(), st[:n].copy()


def demod(samples, invert=0, cap=4096):
    lib = oracle()
    s = np.ascontiguousarray(samples, dtype=np.int16)
    recs = np.zeros(cap, dtype=FRAME_REC)
    diag = np.zeros(1, dtype=DIAG)
    n = lib.m17o_demod(_p(s), C.c_size_t(s.size), C.c_int(invert), _p(recs), C.c_size_t(cap), _p(diag))
    assert n <= cap
    return recs[:n].copy(), diag[0].copy()


def hybrid_demod(samples, invert=0, cap=4096, taps150=None):
    """The oracle's ORCHESTRATOR over the REFERENCE's own operator objects (oracle/ref_shim.cpp, ref_hybrid_demod): records and the last
    diagnostic callback of one channel, laid out like demod()'s.  Needs oracle/_ref (the build container)."""
    lib = ref()
    lib.ref_hybrid_demod.restype = C.c_size_t
    s = np.ascontiguousarray(samples, dtype=np.int16)
    t = np.ascontiguousarray(taps() if taps150 is None else taps150, dtype=np.float32)
    recs = np.zeros(cap, dtype=FRAME_REC)
    diag = np.zeros(1, dtype=DIAG)
    n = lib.ref_hybrid_demod(_p(t), _p(s), C.c_size_t(s.size), C.c_int(invert), _p(recs), C.c_size_t(cap), _p(diag))
    assert n <= cap
    return recs[:n].copy(), diag[0].copy()


def hybrid_diag_log(samples, invert=0, cap=4096, taps150=None):
    lib = ref()
    lib.ref_hybrid_diag_log.restype = C.c_size_t
    s = np.ascontiguousarray(samples, dtype=np.int16)
    t = np.ascontiguousarray(taps() if taps150 is None else taps150, dtype=np.float32)
    log = np.zeros(cap, dtype=DIAG)
    n = lib.ref_hybrid_diag_log(_p(t), _p(s), C.c_size_t(s.size), C.c_int(invert), _p(log), C.c_size_t(cap))
    assert n <= cap
    return log[:n].copy()


def random_scenario(seed, total=96000, kinds=(0, 1, 2, 3, 4)):
    """One channel of tools/parity_sweep.py's scenario generator: bursts of random kind (BERT, voice-like stream, RAW packet, noise only, packet
    closed by an FCS), length, lead-in, noise, DC offset, gain and symbol phase, back to back — lost sync, forced carrier-detect unlocks,
    restarts of the gated matched filter, and (a silent stretch) the carrier detect's 0 / 0."""
    rng = np.random.default_rng(seed)
    x = np.zeros(total, dtype=np.int16)
    pos = 0
    while pos < total - 8000:
        n = min(int(rng.integers(6000, 40000)), total - pos)
        p = gen_params(seed=int(rng.integers(1, 1 << 30)), kind=int(rng.choice(kinds)), n_frames=int(rng.integers(1, 16)),
                       lead_in=int(rng.integers(0, 5000)), lead_sigma=float(rng.choice([0.0, 100.0, 1000.0, 10000.0, 40000.0])),
                       noise_sigma=float(rng.choice([0.0, 100.0, 500.0, 1200.0, 2500.0])), tail_sigma=float(rng.choice([0.0, 100.0, 1000.0, 5000.0])),
                       dc_offset=float(rng.choice([0.0, 0.0, 300.0, -2000.0, 6000.0])), gain=float(rng.choice([1.0, 0.3, 0.7, 1.6])),
                       phase=int(rng.integers(-1, 10)), invert=0, total=n)
        x[pos:pos + n] = generate(p)[:n]
        pos += n
    return x


def demod_batch(samples2d, invert=0, cap=600, threads=8):
    lib = oracle()
    s = np.ascontiguousarray(samples2d, dtype=np.int16)
    Cn, T = s.shape
    recs = np.zeros((Cn, cap), dtype=FRAME_REC)
    counts = np.zeros(Cn, dtype=np.uint32)
    diags = np.zeros(Cn, dtype=DIAG)
    lib.m17o_demod_batch(_p(s), C.c_size_t(Cn), C.c_size_t(T), C.c_size_t(T), C.c_int(invert), C.c_int(threads), _p(recs),
                         C.c_size_t(cap), _p(counts), _p(diags))
    return recs, counts, diags


def demod_diag_log(samples, invert=0, cap=4096):
    """Every diagnostic callback of one channel in order (entries laid out like m17hip_diag_log_fetch's)."""
    lib = oracle()
    lib.m17o_demod_diag_log.restype = C.c_size_t
    s = np.ascontiguousarray(samples, dtype=np.int16)
    log = np.zeros(cap, dtype=DIAG)
    n = lib.m17o_demod_diag_log(_p(s), C.c_size_t(s.size), C.c_int(invert), _p(log), C.c_size_t(cap))
    assert n <= cap
    return log[:n].copy()


def demod_symbols(samples, invert=0, cap=1 << 20):
    lib = oracle()
    s = np.ascontiguousarray(samples, dtype=np.int16)
    out = np.zeros(cap, dtype=np.float32)
    n = lib.m17o_demod_symbols(_p(s), C.c_size_t(s.size), C.c_int(invert), _p(out), C.c_size_t(cap))
    return out[: min(n, cap)].copy()


def scale(samples, invert=0, lib=None):
    s = np.ascontiguousarray(samples, dtype=np.int16)
    out = np.zeros(s.size, dtype=np.float32)
    oracle().m17o_scale(_p(s), C.c_size_t(s.size), C.c_int(invert), _p(out))
    return out


def taps():
    t = np.zeros(150, dtype=np.float32)
    oracle().m17o_taps(_p(t))
    return t


def fir_f32(x):
    x = np.ascontiguousarray(x, dtype=np.float32)
    y = np.zeros_like(x)
    oracle().m17o_fir_f32(_p(x), C.c_size_t(x.size), _p(y))
    return y


def fir_i16(s, invert=0):
    s = np.ascontiguousarray(s, dtype=np.int16)
    y = np.zeros(s.size, dtype=np.float32)
    oracle().m17o_fir_i16(_p(s), C.c_size_t(s.size), C.c_int(invert), _p(y))
    return y


def correlator(y, lib=None, prefix="m17o_"):
    lib = lib or oracle()
    y = np.ascontiguousarray(y, dtype=np.float32)
    limit = np.zeros(y.size, dtype=np.float32)
    corr = np.zeros((4, y.size), dtype=np.float32)
    getattr(lib, prefix + "correlator")(_p(y), C.c_size_t(y.size), _p(limit), _p(corr))
    return limit, corr


def syncword(y, which, lib=None, prefix="m17o_"):
    lib = lib or oracle()
    y = np.ascontiguousarray(y, dtype=np.float32)
    timing = np.zeros(y.size, dtype=np.uint8)
    upd = np.zeros(y.size, dtype=np.int8)
    trig = np.zeros(y.size, dtype=np.float32)
    getattr(lib, prefix + "syncword")(_p(y), C.c_size_t(y.size), C.c_int(which), _p(timing), _p(upd), _p(trig))
    return timing, upd, trig


def outer_levels(y, si, lib=None, prefix="m17o_"):
    lib = lib or oracle()
    y = np.ascontiguousarray(y, dtype=np.float32)
    mn, mx = C.c_float(0), C.c_float(0)
    getattr(lib, prefix + "outer_levels")(_p(y), C.c_size_t(y.size), C.c_size_t(si), C.byref(mn), C.byref(mx))
    return np.float32(mn.value), np.float32(mx.value)


def dcd_trace(x, period, lib=None, prefix="m17o_"):
    lib = lib or oracle()
    x = np.ascontiguousarray(x, dtype=np.float32)
    k = x.size // period
    level = np.zeros(k, dtype=np.float32)
    trig = np.zeros(k, dtype=np.uint8)
    n = getattr(lib, prefix + "dcd_trace")(_p(x), C.c_size_t(x.size), C.c_size_t(period), _p(level), _p(trig))
    assert n == k
    return level, trig


def dcd_sums(x, start, length, lib=None, prefix="m17o_"):
    lib = lib or oracle()
    x = np.ascontiguousarray(x, dtype=np.float32)
    a, b = C.c_float(0), C.c_float(0)
    getattr(lib, prefix + "dcd_sums")(_p(x), C.c_size_t(start), C.c_size_t(length), C.byref(a), C.byref(b))
    return np.float32(a.value), np.float32(b.value)


def evm_trace(sym, do_reset=1, lib=None, prefix="m17o_"):
    lib = lib or oracle()
    sym = np.ascontiguousarray(sym, dtype=np.float32)
    out = np.zeros_like(sym)
    getattr(lib, prefix + "evm_trace")(_p(sym), C.c_size_t(sym.size), C.c_int(do_reset), _p(out))
    return out


def llr(sym, lib=None, prefix="m17o_"):
    lib = lib or oracle()
    sym = np.ascontiguousarray(sym, dtype=np.float32)
    out = np.zeros(2 * sym.size, dtype=np.int8)
    getattr(lib, prefix + "llr")(_p(sym), C.c_size_t(sym.size), _p(out))
    return out


def crc16(data, lib=None, prefix="m17o_"):
    lib = lib or oracle()
    d = np.ascontiguousarray(np.frombuffer(bytes(data), dtype=np.uint8))
    return int(getattr(lib, prefix + "crc16")(_p(d) if d.size else None, C.c_size_t(d.size)))


def decode_callsign(enc6, lib=None, prefix="m17o_"):
    lib = lib or oracle()
    e = np.ascontiguousarray(np.frombuffer(bytes(enc6), dtype=np.uint8))
    out = np.zeros(10, dtype=np.uint8)
    getattr(lib, prefix + "decode_callsign")(_p(e), _p(out))
    return bytes(out)


def encode_callsign(call, lib=None, prefix="m17o_"):
    lib = lib or oracle()
    out = np.zeros(6, dtype=np.uint8)
    getattr(lib, prefix + "encode_callsign")(C.c_char_p(call.encode()), _p(out))
    return bytes(out)


def golay_encode24(v, lib=None, prefix="m17o_"):
    lib = lib or oracle()
    return int(getattr(lib, prefix + "golay_encode24")(C.c_uint16(v)))


def golay_decode(v, lib=None, prefix="m17o_"):
    lib = lib or oracle()
    out = C.c_uint32(0)
    ok = getattr(lib, prefix + "golay_decode")(C.c_uint32(v), C.byref(out))
    return bool(ok), int(out.value)


def frame_op(name, f, lib=None, prefix="m17o_"):
    lib = lib or oracle()
    a = np.ascontiguousarray(f, dtype=np.int8).copy()
    assert a.size == 368
    getattr(lib, prefix + name)(_p(a))
    return a


def viterbi(soft, out_bits, llr_bits=4, lib=None, prefix="m17o_"):
    lib = lib or oracle()
    s = np.ascontiguousarray(soft, dtype=np.int8)
    out = np.zeros(out_bits, dtype=np.uint8)
    if prefix == "m17o_":
        cost = lib.m17o_viterbi(_p(s), C.c_size_t(s.size), _p(out), C.c_size_t(out_bits), C.c_int(llr_bits))
    else:
        cost = lib.ref_viterbi(_p(s), C.c_size_t(s.size), _p(out), C.c_size_t(out_bits))
    return int(cost), out


def depuncture(soft, out_len, which, prefill=None, lib=None, prefix="m17o_"):
    lib = lib or oracle()
    s = np.ascontiguousarray(soft, dtype=np.int8)
    out = np.zeros(out_len, dtype=np.int8) if prefill is None else np.ascontiguousarray(prefill, dtype=np.int8).copy()
    getattr(lib, prefix + "depuncture")(_p(s), C.c_size_t(s.size), _p(out), C.c_size_t(out_len), C.c_int(which))
    return out


def puncture(bits, out_len, which):
    b = np.ascontiguousarray(bits, dtype=np.uint8)
    out = np.zeros(out_len, dtype=np.int8)
    n = oracle().m17o_puncture(_p(b), C.c_size_t(b.size), _p(out), C.c_size_t(out_len), C.c_int(which))
    return out[:n]


def conv_encode(bits):
    b = np.ascontiguousarray(bits, dtype=np.uint8)
    out = np.zeros(2 * (b.size + 4), dtype=np.uint8)
    n = oracle().m17o_conv_encode(_p(b), C.c_size_t(b.size), _p(out))
    return out[:n]


def prbs9(n, state=1, lib=None, prefix="m17o_"):
    lib = lib or oracle()
    st = C.c_uint16(state)
    bits = np.zeros(n, dtype=np.uint8)
    getattr(lib, prefix + "prbs9")(C.byref(st), _p(bits), C.c_size_t(n))
    return bits, st.value


def bert_count(payloads25, lib=None, prefix="m17o_"):
    lib = lib or oracle()
    p = np.ascontiguousarray(payloads25, dtype=np.uint8).reshape(-1, 25)
    bits, errs, sync = C.c_uint32(0), C.c_uint32(0), C.c_int(0)
    getattr(lib, prefix + "bert_count")(_p(p), C.c_size_t(p.shape[0]), C.byref(bits), C.byref(errs), C.byref(sync))
    return bits.value, errs.value, bool(sync.value)


def crc16_x25(data, lib=None, prefix="m17o_"):
    lib = lib or oracle()
    d = np.ascontiguousarray(data, dtype=np.uint8)
    f = getattr(lib, prefix + "crc16_x25")
    f.restype = C.c_uint16
    return f(_p(d), C.c_size_t(d.size))


class PacketAssembler:
    """The packet consumer of apps/m17-demod.cpp (decode_packet + dump_lsf's reset) for one channel, state kept across calls."""

    def __init__(self):
        self.cur = np.zeros(832, dtype=np.uint8)
        self.st = np.zeros(4, dtype=np.uint32)

    def feed(self, types, payloads32, cap=64):
        lib = oracle()
        t = np.ascontiguousarray(types, dtype=np.uint8)
        pl = np.ascontiguousarray(payloads32, dtype=np.uint8).reshape(-1, 32)
        size, csum = np.zeros(cap, np.uint16), np.zeros(cap, np.uint16)
        frames, errs, idx = np.zeros(cap, np.uint8), np.zeros(cap, np.uint8), np.zeros(cap, np.uint32)
        data = np.zeros((cap, 840), np.uint8)
        f = lib.m17o_packet_reassemble
        f.restype = C.c_size_t
        n = f(_p(t), _p(pl), C.c_size_t(t.size), _p(self.cur), _p(self.st), C.c_size_t(cap), _p(size), _p(csum), _p(frames), _p(errs),
              _p(idx), _p(data))
        assert n <= cap
        return [dict(size=int(size[k]), checksum=int(csum[k]), frames=int(frames[k]), seq_errors=int(errs[k]), rec_index=int(idx[k]),
                     data=data[k].copy()) for k in range(n)]


def decode_frame(sync_type, llr368, state=0, lich=0, lsf=None, dep401=0, cost=0, lib=None, prefix="m17o_"):
    """One frame through the frame decoder; returns (records, state, lich_segments, lsf, dep401, cost)."""
    lib = lib or oracle()
    l = np.ascontiguousarray(llr368, dtype=np.int8)
    st, li, d4, co = C.c_uint8(state), C.c_uint8(lich), C.c_int8(dep401), C.c_int64(cost)
    lsfb = np.zeros(30, dtype=np.uint8) if lsf is None else np.ascontiguousarray(lsf, dtype=np.uint8).copy()
    recs = np.zeros(2, dtype=FRAME_REC)
    n = getattr(lib, prefix + "decode_frame")(C.c_int(sync_type), _p(l), C.byref(st), C.byref(li), _p(lsfb), C.byref(d4),
                                              C.byref(co), _p(recs))
    return recs[:n].copy(), st.value, li.value, lsfb, d4.value, co.value
