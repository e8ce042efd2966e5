"""One-hop GPU parity: the HIP operators, through the C ABI, against outputs of the REFERENCE'S OWN HEADERS frozen in
tests/golden/ref_vectors.npz (tests/golden/make_golden.py ran them in the build container through oracle/_ref).  No oracle in
between: golden input -> libm17hip.so -> golden output, bit for bit.  Covers SURVEY §8(a) rows a1-a4, a7, a8, a11, a12, a17
(rows a14-a16, a18 are in test_gpu_parity.py::test_decode_frames_golden_sequences)."""
import numpy as np
import pytest

import m17hip

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    c = m17hip.Context(8, 24000)
    yield c
    c.close()


def _sets(golden):
    """(tag, int16 input, invert flag) of every front-end fixture set: 'sig' and, when present, the extra seeded sets."""
    out = [("", golden["sig_i16"], 0)]
    for tag in ("inv_", "dc_", "zero_"):
        if tag + "sig_i16" in golden:
            out.append((tag, golden[tag + "sig_i16"], 1 if tag == "inv_" else 0))
    return out


def test_fir_and_correlator_equal_reference_outputs(ctx, golden):
    """a1 + a2 (scaling, BaseFirFilter<float,150>) and a3 + a4 (limit IIR, the four sync-word correlations)."""
    for tag, s, inv in _sets(golden):
        ctx.upload(np.stack([s, s]))
        y = ctx.fir(flags=inv)
        assert np.array_equal(y[0], golden[tag + "fir_out"]) and np.array_equal(y[1], golden[tag + "fir_out"]), tag
        limit, corr = ctx.correlator()
        assert np.array_equal(limit[1], golden[tag + "corr_limit"]), tag
        assert np.array_equal(corr[:, 1, :], golden[tag + "corr_values"]), tag


def test_dcd_sums_and_levels_equal_reference(ctx, golden):
    """a7 + a8: the sliding-DFT sums table against the reference's own sums over the same windows, and
    DataCarrierDetect::level_ recomputed from the table for the two update cadences (384 / 960 samples)."""
    for tag, s, inv in _sets(golden):
        ctx.upload(s[None, :])
        sums = ctx.dcd(flags=inv)[0]                      # [ticks][2 bins][6 sums]
        n32 = s.size // 32 * 32                            # the four-wave latency form of K3 (m17hip_tune key 10) needs whole 32-sample blocks
        ctx.tune(10, 1)
        try:
            sums_p = ctx.dcd(flags=inv, samples=n32)[0]
        finally:
            ctx.tune(10, -1)
        assert np.array_equal(sums_p.view(np.uint32), sums[: n32 // 192].view(np.uint32)), tag
        if tag + "dcd_sums" in golden:
            for st, ln, l1, l2 in golden[tag + "dcd_sums"]:
                a0, k = int(st) // 192, (int(st) + int(ln)) // 192 - 1
                j = 5 if a0 == 0 and k >= 5 else a0 % 5
                assert (float(sums[k, 0, j]), float(sums[k, 1, j])) == (float(np.float32(l1)), float(np.float32(l2))), (tag, st, ln)
        for period, ticks in ((384, 2), (960, 5)):
            level, trig = np.float32(0.0), False
            exp_l, exp_t = golden[f"{tag}dcd{period}_level"], golden[f"{tag}dcd{period}_trig"]
            for u in range(exp_l.size):
                a0, k = u * ticks, u * ticks + ticks - 1
                l1, l2 = sums[k, 0, a0 % 5], sums[k, 1, a0 % 5]
                with np.errstate(invalid="ignore", divide="ignore"):
                    level = np.float32(np.float64(level) * 0.8 + 0.2 * np.float64(np.float32(l1) / np.float32(l2)))   # DataCarrierDetect.h:65
                trig = bool(level > np.float32(0.1)) if trig else bool(level > np.float32(4.0))                   # :66-69 with M17Demodulator.h:149
                assert np.array_equal(level, exp_l[u], equal_nan=True), (tag, period, u)
                assert trig == bool(exp_t[u]), (tag, period, u)


def test_slicer_and_evm_equal_reference(ctx, golden):
    """a11 + a12: llr<float,4> and SymbolEvm."""
    llr, _ = ctx.slice_llr(golden["llr_in"])
    assert np.array_equal(llr.reshape(-1), golden["llr_out"])
    _, evm = ctx.slice_llr(golden["llr_in"][4001:6001])
    assert np.array_equal(evm[0], golden["evm_out"])


def test_viterbi_equals_reference(ctx, golden):
    """a17: Viterbi<Trellis<4,2>,4>::decode, all four frame shapes, noisy punctured input."""
    kinds = {(488, 240): 0, (296, 144): 1, (420, 206): 2, (402, 197): 3}
    seen = set()
    for row, (IN, OUT, cost) in enumerate(golden["vit_meta"]):
        kind = kinds[(int(IN), int(OUT))]
        bits, c = ctx.viterbi(golden["vit_in"][row, :IN][None, :], kind)
        assert int(c[0]) == int(cost) and np.array_equal(bits[0], golden["vit_out"][row, :OUT]), row
        seen.add(kind)
    assert seen == {0, 1, 2, 3}
