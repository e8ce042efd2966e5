#!/usr/bin/env python3
"""Regenerates the golden fixtures in this directory.  RUNS ONLY IN THE BUILD CONTAINER
(needs /root/reference and oracle/_ref/libm17ref.so, the reference's own headers compiled
where they lie).  The fixtures are DATA: seeded inputs and the outputs the reference's own
code produced for them, plus the known-answer vectors held by the reference's unit tests.

  ref_vectors.npz   inputs -> outputs of the reference operators (via oracle/_ref)
  kat_vectors.json  literal vectors from reference tests/ViterbiTest.cpp, tests/UtilTest.cpp
  hybrid_vectors.npz  what the reference's OWN operator objects deliver under the oracle's orchestrator (oracle/ref_shim.cpp,
                    ref_hybrid_demod) for 48 scenarios of oracle_lib.random_scenario: frame records, last diagnostic callback, and a
                    checksum of each input (`make_golden.py hybrid` writes this file alone)

The input signals come from the repository's own synthetic generator (oracle/m17_oracle_gen.hpp).
"""
import json
import os
import re
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import oracle_lib as ol  # noqa: E402

REF = "/root/reference"


def ref_taps():
    src = open(os.path.join(REF, "include/m17cxx/M17Demodulator.h")).read()
    m = re.search(r"struct Taps<float>.*?std::array<float, 150>\{(.*?)\};", src, re.S)
    vals = [float(v) for v in m.group(1).replace("\n", " ").split(",") if v.strip()]
    assert len(vals) == 150
    return np.array(vals, dtype=np.float64).astype(np.float32)


def parse_array(text, name):
    m = re.search(name + r"\s*=?\s*\{([^}]*)\}", text, re.S)
    return [int(v) for v in m.group(1).replace("\n", " ").split(",") if v.strip()]


def main():
    ol.build_oracle()
    R = ol.ref()
    assert R is not None, "oracle/_ref not built"
    import ctypes as C
    out = {}
    taps = ref_taps()
    out["taps"] = taps

    # --- front end on a noisy voice-like burst -------------------------------------------
    p = ol.gen_params(seed=11, kind=1, n_frames=2, lead_in=700, noise_sigma=500, tail=300, tail_sigma=500, lead_sigma=40000.0)
    s = ol.generate(p)[:6000]
    x = ol.scale(s)
    out["sig_i16"] = s
    out["sig_scaled"] = x
    y = np.zeros_like(x)
    R.ref_fir_f32(ol._p(taps), ol._p(x), C.c_size_t(x.size), ol._p(y))
    out["fir_out"] = y
    lim, corr = ol.correlator(y, lib=R, prefix="ref_")
    out["corr_limit"] = lim
    out["corr_values"] = corr
    for w in range(4):
        t, u, tr = ol.syncword(y, w, lib=R, prefix="ref_")
        out[f"sync{w}_timing"], out[f"sync{w}_updated"], out[f"sync{w}_trig"] = t, u, tr
    lv = []
    for n in (80, 83, 1500, 2999, 4000):
        for si in (0, 3, 9):
            lv.append((n, si) + ol.outer_levels(y[:n], si, lib=R, prefix="ref_"))
    out["outer_levels"] = np.array(lv, dtype=np.float64)
    for period in (384, 960):
        l, t = ol.dcd_trace(x, period, lib=R, prefix="ref_")
        out[f"dcd{period}_level"], out[f"dcd{period}_trig"] = l, t
    sums = []
    for (st, ln) in ((0, 2304), (2304, 384), (2688, 960), (1920, 384), (3648, 960)):
        sums.append((st, ln) + ol.dcd_sums(x, st, ln, lib=R, prefix="ref_"))
    out["dcd_sums"] = np.array(sums, dtype=np.float64)
    sd = np.zeros(4 * 600, dtype=np.float32)
    R.ref_sdft(ol._p(x), C.c_size_t(600), ol._p(sd))
    out["sdft_first600"] = sd

    # --- slicer / EVM -------------------------------------------------------------------------
    rng = np.random.default_rng(5)
    sym = np.concatenate([np.linspace(-4, 4, 4001, dtype=np.float32), rng.normal(0, 2, 2000).astype(np.float32),
                          np.array([0.0001, -0.0001, 1.0001, 0.9999, 2.0001, 1.9999, -1.0001, -0.9999, -2.0001, -1.9999], dtype=np.float32)])
    out["llr_in"] = sym
    out["llr_out"] = ol.llr(sym, lib=R, prefix="ref_")
    out["evm_out"] = ol.evm_trace(sym[4001:6001], 1, lib=R, prefix="ref_")

    # --- FEC: Viterbi on noisy coded frames of all four shapes --------------------------------
    shapes = [(488, 240, 1, 368), (296, 144, 2, 272), (420, 206, 3, 368), (402, 197, 2, 368)]
    vit_in, vit_out, vit_cost = [], [], []
    for (IN, OUT, pm, npun) in shapes:
        for trial in range(6):
            bits = rng.integers(0, 2, OUT).astype(np.uint8)
            enc = ol.conv_encode(bits)
            assert enc.size == IN
            pun = ol.puncture(enc, npun, pm).astype(np.int16)
            soft = (pun * 2 - 1) * rng.integers(1, 8, pun.size)
            flips = rng.random(pun.size) < (0.03 * trial)
            soft = np.where(flips, -soft, soft).astype(np.int8)
            dep = ol.depuncture(soft, IN, pm, prefill=np.full(IN, 3, np.int8), lib=R, prefix="ref_")
            cost, dec = ol.viterbi(dep, OUT, lib=R, prefix="ref_")
            vit_in.append(np.pad(dep, (0, 488 - IN))); vit_out.append(np.pad(dec, (0, 240 - OUT))); vit_cost.append((IN, OUT, cost))
    out["vit_in"] = np.array(vit_in, dtype=np.int8)
    out["vit_out"] = np.array(vit_out, dtype=np.uint8)
    out["vit_meta"] = np.array(vit_cost, dtype=np.int64)

    # --- frame decoder sequences (state carried across frames, incl. the stale dep[401] byte) ------
    seqs = []
    for seed, kind in ((21, 0), (22, 1), (23, 2), (24, 1), (25, 0)):
        fbits, stypes = ol.make_frames(kind, seed, 8)
        state, lich, lsf, d401, cost = 0, 0, np.zeros(30, np.uint8), 0, 0
        for f in range(fbits.shape[0]):
            st = int(stypes[f])
            if seed == 24 and f == 0:
                continue                      # late entry: the LSF is missed, LICH must rebuild it
            mag = rng.integers(1, 8, 368)
            fr = ((fbits[f].astype(np.int16) * 2 - 1) * mag)
            if seed in (22, 25):
                flips = rng.random(368) < (0.05 if f != 3 else 0.25)
                fr = np.where(flips, -fr, fr)
            fr = fr.astype(np.int8)
            if seed == 25 and f == 4:
                st = 0                        # a BERT frame mis-tagged as LSF leaves a different stale dep[401] (Q4)
            recs, state, lich, lsf, d401, cost = ol.decode_frame(st, fr, state, lich, lsf, d401, cost, lib=R, prefix="ref_")
            seqs.append(dict(seed=seed, f=f, st=st, llr=fr.tolist(), state=int(state), lich=int(lich), lsf=lsf.tolist(), d401=int(d401),
                             cost=int(cost), recs=[(int(r["frame_type"]), int(r["cost"]), int(r["len"]), bytes(r["payload"]).hex()) for r in recs]))
    # --- more front-end sets (round 2): inverted input, DC offset + low gain, a window of exact zeros (DCD NaN) ----
    extra = {
        "inv_": (ol.gen_params(seed=31, kind=0, n_frames=2, lead_in=900, noise_sigma=700, tail=200, tail_sigma=700, lead_sigma=40000.0, invert=1), 1),
        "dc_": (ol.gen_params(seed=32, kind=2, n_frames=2, lead_in=500, noise_sigma=300, tail=400, tail_sigma=300, lead_sigma=20000.0,
                              dc_offset=-1500.0, gain=0.6), 0),
        "zero_": (ol.gen_params(seed=33, kind=1, n_frames=2, lead_in=0, noise_sigma=0, tail=0, tail_sigma=0), 0),
    }
    for tag, (gp, inv) in extra.items():
        s2 = ol.generate(gp)[:6000].copy()
        if tag == "zero_":
            s2[:768] = 0               # the stream opens with two whole 384-sample update windows of digital silence
        x2 = ol.scale(s2, invert=inv)
        y2 = np.zeros_like(x2)
        R.ref_fir_f32(ol._p(taps), ol._p(x2), C.c_size_t(x2.size), ol._p(y2))
        out[tag + "sig_i16"], out[tag + "fir_out"] = s2, y2
        out[tag + "corr_limit"], out[tag + "corr_values"] = ol.correlator(y2, lib=R, prefix="ref_")
        for period in (384, 960):
            out[f"{tag}dcd{period}_level"], out[f"{tag}dcd{period}_trig"] = ol.dcd_trace(x2, period, lib=R, prefix="ref_")
        out[tag + "dcd_sums"] = np.array([(st, ln) + ol.dcd_sums(x2, st, ln, lib=R, prefix="ref_")
                                          for (st, ln) in ((0, 384), (384, 384), (1920, 384), (2304, 960), (4800, 960))], dtype=np.float64)
    assert np.isnan(out["zero_dcd384_level"]).any(), "the zero-window set must poison the DCD level"
    np.savez_compressed(os.path.join(HERE, "ref_vectors.npz"), **out)

    # --- literal KAT vectors from the reference's own unit tests ---------------------------------
    vt = open(os.path.join(REF, "tests/ViterbiTest.cpp")).read()
    body = vt[vt.index("TEST_F(ViterbiTest, decode_ber_lsf)"):]
    kat = dict(
        lsf_expected240=parse_array(body, r"std::array<uint8_t, 240> expected"),
        lsf_encoded488=parse_array(body, r"std::array<int8_t, 488> encoded"),
        frame_decoder_sequences=seqs,
    )
    ut = open(os.path.join(REF, "tests/UtilTest.cpp")).read()
    ub = ut[ut.index("TEST_F(UtilTest, BERT_first_frame)"):]
    m = re.search(r"bool baseline\[\] = \{(.*?)\};", ub, re.S)
    kat["bert_first_frame_baseline"] = [int(v) for v in m.group(1).replace("\n", " ").split(",") if v.strip()]
    json.dump(kat, open(os.path.join(HERE, "kat_vectors.json"), "w"))
    print("wrote", os.path.join(HERE, "ref_vectors.npz"), os.path.getsize(os.path.join(HERE, "ref_vectors.npz")), "bytes;",
          os.path.getsize(os.path.join(HERE, "kat_vectors.json")), "bytes json")


def hybrid():
    """The composition pin, frozen: tests/test_oracle_kat.py compares the pure oracle with these wherever oracle/_ref is absent."""
    import zlib
    assert ol.ref() is not None, "oracle/_ref not built"
    out = {"seeds": np.arange(48, dtype=np.int64), "total": np.int64(48000)}
    recs, diags, sums, counts = [], [], [], []
    for seed in out["seeds"]:
        x = ol.random_scenario(int(seed), total=48000)
        r, d = ol.hybrid_demod(x, invert=int(seed) & 1, taps150=ref_taps())
        recs.append(r); diags.append(d); counts.append(r.size); sums.append(zlib.crc32(x.tobytes()))
    out["records"] = np.concatenate(recs).view(np.uint8).reshape(-1, 64)
    out["counts"] = np.array(counts, dtype=np.int64)
    out["diags"] = np.array(diags).view(np.uint8).reshape(-1, 64)
    out["input_crc32"] = np.array(sums, dtype=np.uint32)
    np.savez_compressed(os.path.join(HERE, "hybrid_vectors.npz"), **out)
    print("wrote hybrid_vectors.npz:", int(out["counts"].sum()), "records of", len(counts), "scenarios,", os.path.getsize(os.path.join(HERE, "hybrid_vectors.npz")), "bytes")


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "hybrid":
        hybrid()
    else:
        main()
        hybrid()
