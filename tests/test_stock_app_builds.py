"""Boundary check (build container only): the reference's UNMODIFIED apps/m17-demod.cpp compiles against this repository's
operator surface (m17-cxx-demod_amd/include/m17cxx) and links with libm17hip.so — the "drops into apps/m17-demod unchanged"
half of the north star.  The file is read from /root/reference where it lies and never copied; codec2 / boost, which it also
includes and which are absent here, are declared by tests/shims (not an oracle, see tests/shims/README.md).  A second
translation unit instantiates every class of the surface with the reference's signatures."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
INC = os.path.join(ROOT, "m17-cxx-demod_amd", "include", "m17cxx")
APP = "/root/reference/apps/m17-demod.cpp"
FLAGS = ["g++", "-std=c++20", "-O2", "-ffp-contract=off", "-Wall", "-Wno-unused-variable", "-Wno-unused-but-set-variable", "-Wno-sign-compare",
         "-Wno-unused-function"]


@pytest.mark.skipif(not os.path.exists(APP), reason="reference not present (build container only)")
def test_unmodified_stock_app_builds_against_the_mirror(tmp_path):
    out = tmp_path / "m17-demod-stock"
    cmd = FLAGS + ["-I", INC, "-I", os.path.join(ROOT, "tests", "shims"), APP, "-L", os.path.join(ROOT, "m17-cxx-demod_amd"), "-lm17hip",
                   "-Wl,-rpath," + os.path.join(ROOT, "m17-cxx-demod_amd"), "-L/opt/rocm/lib", "-Wl,-rpath,/opt/rocm/lib", "-o", str(out)]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-4000:]
    assert out.exists()
    # the include path really resolved to the mirror, not to the reference's own headers
    deps = subprocess.run(FLAGS + ["-I", INC, "-I", os.path.join(ROOT, "tests", "shims"), "-MM", APP], capture_output=True, text=True).stdout
    assert "m17-cxx-demod_amd/include/m17cxx/M17Demodulator.h" in deps and "/root/reference/include" not in deps
    # --version is handled before anything touches the GPU
    v = subprocess.run([str(out), "--version"], capture_output=True, text=True)
    assert v.returncode == 0 and "2.2" in v.stdout


@pytest.mark.skipif(not os.path.exists(APP), reason="reference not present (build container only)")
def test_unmodified_stock_app_demodulates_a_stream_on_the_host_form(tmp_path):
    """BASELINE configs[0]: one 48 kSPS stream through the stock m17-demod with no GPU.  The unmodified application, built as a release
    build (NDEBUG, as the reference's CMake does) against the mirror, is told to stay on the host (M17_DEMOD_DEVICE=cpu ->
    detail/scalar_demod.h) and fed a BERT transmission on stdin: it locks, counts the PRBS9 bits of the frames the oracle
    decodes too, and reports the bit error rate the application prints (apps/m17-demod.cpp:353-368)."""
    import re
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as ol
    out = tmp_path / "m17-demod-stock"
    cmd = FLAGS + ["-DNDEBUG", "-I", INC, "-I", os.path.join(ROOT, "tests", "shims"), APP, "-L", os.path.join(ROOT, "m17-cxx-demod_amd"), "-lm17hip",
                   "-Wl,-rpath," + os.path.join(ROOT, "m17-cxx-demod_amd"), "-L/opt/rocm/lib", "-Wl,-rpath,/opt/rocm/lib", "-o", str(out)]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-4000:]
    p = ol.gen_params(seed=5, kind=0, n_frames=12, lead_in=3072, noise_sigma=300.0, tail_sigma=300.0, lead_sigma=40000.0, total=48000)
    x = ol.generate(p)
    run = subprocess.run([str(out), "-d"], input=x.tobytes(), capture_output=True, env=dict(os.environ, M17_DEMOD_DEVICE="cpu"))
    assert run.returncode == 0
    err = run.stderr.replace(b"\r", b"\n").decode(errors="replace")
    recs, _ = ol.demod(x)
    bert = [r_ for r_ in recs if int(r_["frame_type"]) == 5]
    assert len(bert) >= 12
    bits = [int(m) for m in re.findall(r"BER: [0-9.]+ \((\d+) bits\)", err)]
    assert bits and 197 * 11 <= max(bits) <= 197 * len(bert)     # the 12 transmitted frames (the receiver needs 18 good bits to lock) ... the garbage after them
    assert re.search(r"BER: 0\.0", err)
    assert "locked:  true" in err and "dcd: 1" in err


def test_surface_signatures_compile(tmp_path):
    """Every class the north star names, instantiated and called with the reference's signatures."""
    src = tmp_path / "surface.cpp"
    src.write_text(r'''
#include "M17Demodulator.h"
#include "CRC16.h"
#include "ax25_frame.h"
#include "FirFilter.h"
#include "SlidingDFT.h"
#include "IirFilter.h"
#include "KalmanFilter.h"
bool display_lsf = false;
using namespace mobilinkd;
static bool on_frame(M17FrameDecoder::output_buffer_t const&, int) { return true; }
int main(int argc, char**)
{
    static const std::array<float, 150> taps = detail::Taps<float>::rrc_taps;
    BaseFirFilter<float, 150> fir(taps);
    auto fir2 = makeFirFilter(taps);
    float y = fir(1.0f) + fir2(1.0f); fir.reset();
    Correlator<float> corr; corr.sample(y);
    Correlator<float>::sync_t w = {+3, -3, +3, -3, +3, -3, +3, -3};
    float c = corr.correlate(w) + corr.limit() + float(corr.index());
    auto [mn, mx] = corr.outer_symbol_levels(3);
    corr.apply([](float) {}, 2);
    SyncWord<Correlator<float>> sw{{+3, +3, +3, +3, -3, -3, +3, -3}, 31.f, -31.f};
    float t = sw.triggered(corr); size_t ti = sw(corr); int8_t up = sw.updated(); bool trg = sw.is_triggered();
    SlidingDFT<float, 48000, 2400> sdft; auto s1 = sdft(0.5f);
    NSlidingDFT<float, 48000, 120, 2> ndft({2400, 3600}); NSlidingDFT<float, 48000, 120, 2>::result_type s2 = ndft(0.5f);
    DataCarrierDetect<float, 48000, 400> dcd{2400, 3600, 0.1, 4.0}; dcd(0.25f); dcd.update(); dcd.unlock(); float lv = dcd.level(); bool on = dcd.dcd();
    ClockRecovery<float, 10> clk; clk.reset(3.f); clk(0.f); bool u1 = clk.update(uint8_t(4)); bool u2 = clk.update();
    float ce = clk.clock_estimate(); uint8_t si = clk.sample_index();
    FreqDevEstimator<float> fde; fde.update(-3, 3); fde.reset(); float dv = fde.deviation() + fde.idev() + fde.offset() + fde.error();
    m17::KalmanFilter<float, 10> kf; auto kx = kf.update(4.f, 1920); float k0 = kx[0] + kf.x[1] + kf.P(0, 0);
    auto trellis = makeTrellis<4, 2>({031, 027});
    Viterbi<decltype(trellis), 4> vit(trellis);
    std::array<int8_t, 488> soft{}; std::array<uint8_t, 240> bits{};
    size_t cost = vit.decode(soft, bits);
    std::array<int8_t, 368> frame{}; std::array<int8_t, 488> dep{}; size_t erased = depuncture(frame, dep, P1);
    auto [l0, l1] = llr<float, 4>(0.7f);
    M17FrameDecoder dec(on_frame); size_t vc = 0;
    M17FrameDecoder::DecodeResult res = dec(M17FrameDecoder::SyncWordType::LSF, frame, vc); dec.reset(); auto st = dec.state();
    M17Framer<368> framer; int8_t* fp = nullptr; size_t fl = framer(std::make_tuple(int8_t(1), int8_t(-1)), &fp); framer.reset();
    SymbolEvm<float> evm; evm.update(0.9f); evm.reset(); float e = evm.evm();
    PRBS9 prbs; bool pb = prbs.generate(); prbs.validate(pb); prbs.reset();
    CRC16<0x5935, 0xFFFF> crc; crc.reset(); crc(uint8_t(1)); uint16_t cv = crc.get();
    LinkSetupFrame::encoded_call_t enc = LinkSetupFrame::encode_callsign({'N', '0', 'C', 'A', 'L', 'L', 0, 0, 0, 0}); auto call = LinkSetupFrame::decode_callsign(enc);
    if (argc > 100) {   // GPU-backed pieces: compiled and linked, not run here
        M17Demodulator<float> demod(on_frame);
        demod.diagnostics([](bool, float, float, float, bool, float, int, int, int, int) {});
        demod(0.1f); bool lk = demod.locked(); demod.passall(false); (void)lk;
        batched::Device dev(2, 4800);
        std::array<int8_t, 488> fr[2]{}; std::array<uint8_t, 240> ob[2]; int32_t cs[2];
        int r1 = Viterbi<decltype(trellis), 4>::decode(dev, fr, 2, ob, cs);
        std::vector<int16_t> in(2 * 4800); std::vector<float> out(2 * 4800);
        int r2 = fir(dev, in.data(), 2, 4800, out.data());
        (void)r1; (void)r2;
    }
    return int(c + mn + mx + t + ti + up + trg + s1.real() + s2[0].real() + lv + on + u1 + u2 + ce + si + dv + k0 + cost + erased + l0 + l1 + int(res) + int(st) + fl + e + cv + call[0]) & 0;
}
''')
    out = tmp_path / "surface"
    cmd = FLAGS + ["-I", INC, str(src), "-L", os.path.join(ROOT, "m17-cxx-demod_amd"), "-lm17hip", "-Wl,-rpath," + os.path.join(ROOT, "m17-cxx-demod_amd"),
                   "-L/opt/rocm/lib", "-Wl,-rpath,/opt/rocm/lib", "-o", str(out)]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-4000:]
    assert subprocess.run([str(out)]).returncode == 0      # the scalar classes run on the CPU
