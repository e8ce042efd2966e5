// ORACLE — TEST INFRASTRUCTURE ONLY.
// Thin extern "C" shim over the REFERENCE's own headers, compiled where they lie
// (-I$(REF)/include/m17cxx) into oracle/_ref/libm17ref.so by oracle/Makefile.  No
// reference source is copied into this repository.  Only the operators whose
// headers compile stand-alone are covered; KalmanFilter.h / ClockRecovery.h /
// FreqDevEstimator.h / M17Demodulator.h need the absent third-party `blaze`
// library and are therefore NOT buildable here (no stand-ins are written).
//
// Harness rules from SURVEY §8c/§9: <stdlib.h> before the reference headers so
// that the unqualified abs() in SyncWord::find_peak resolves to the float
// overload as it does in the real application (Q5); objects are constructed by
// placement-new into zero-filled storage (Q4).
#include <stdlib.h>
#include <math.h>

#include "FirFilter.h"
#include "IirFilter.h"
#include "Correlator.h"
#include "SlidingDFT.h"
#include "DataCarrierDetect.h"
#include "SymbolEvm.h"
#include "Util.h"
#include "Trellis.h"
#include "Viterbi.h"
#include "M17Framer.h"
#include "M17Randomizer.h"
#include "PolynomialInterleaver.h"
#include "CRC16.h"
#include "Golay24.h"
#include "M17FrameDecoder.h"
#include "LinkSetupFrame.h"

#include <cstring>
#include <memory>
#include <new>
#include <type_traits>
#include <vector>

// the oracle's orchestrator (namespace m17o), for ref_hybrid_demod below: its state machine over THIS file's reference operators
#include "m17_oracle_demod.hpp"

bool display_lsf = false;  // M17FrameDecoder.h:19 declares it extern

static_assert(std::is_same<decltype(abs(1.0f)), float>::value, "abs(float) must be the float overload (Q5)");

using namespace mobilinkd;

template <typename T>
struct Zeroed {  // placement-new into zero-filled storage
    alignas(64) unsigned char raw[sizeof(T)];
    T* p = nullptr;
    template <typename... A>
    T& make(A&&... a) { std::memset(raw, 0, sizeof(raw)); p = new (raw) T(std::forward<A>(a)...); return *p; }
    ~Zeroed() { if (p) p->~T(); }
};

template <size_t IN, size_t OUT>
static size_t vit(const int8_t* in, uint8_t* out, size_t* cost)
{
    auto trellis = makeTrellis<4, 2>({031, 027});
    Zeroed<Viterbi<decltype(trellis), 4>> z;
    auto& v = z.make(trellis);
    std::array<int8_t, IN> a;
    std::array<uint8_t, OUT> o;
    std::memcpy(a.data(), in, IN);
    *cost = v.template decode<IN, OUT>(a, o);
    std::memcpy(out, o.data(), OUT);
    return OUT;
}

extern "C" {

struct ref_frame_rec {
    uint32_t channel, seq;
    uint64_t sample_pos;
    int32_t cost;
    uint8_t frame_type, sync_type, len, flags;
    uint8_t payload[32];
    uint8_t pad[8];
};

void ref_fir_f32(const float* taps150, const float* x, size_t n, float* y)
{
    std::array<float, 150> taps;
    std::memcpy(taps.data(), taps150, sizeof(float) * 150);
    BaseFirFilter<float, 150> f(taps);
    for (size_t i = 0; i < n; ++i) y[i] = f(x[i]);
}

static const Correlator<float>::sync_t WORDS[4] = {{+3, -3, +3, -3, +3, -3, +3, -3}, {+3, +3, +3, +3, -3, -3, +3, -3},
                                                   {3, -3, 3, 3, -3, -3, -3, -3}, {+3, +3, +3, +3, +3, +3, -3, +3}};

void ref_correlator(const float* y, size_t n, float* limit, float* corr)
{
    Zeroed<Correlator<float>> z;
    auto& c = z.make();
    for (size_t i = 0; i < n; ++i) {
        c.sample(y[i]);
        limit[i] = c.limit();
        for (int k = 0; k < 4; ++k) corr[k * n + i] = c.correlate(WORDS[k]);
    }
}

void ref_syncword(const float* y, size_t n, int which, uint8_t* timing, int8_t* updated, float* trig)
{
    using SW = SyncWord<Correlator<float>>;
    Zeroed<Correlator<float>> zc;
    auto& c = zc.make();
    Zeroed<SW> zs;
    SW::buffer_t w;
    for (int i = 0; i < 8; ++i) w[i] = WORDS[which][i];
    SW* s;
    switch (which) {
    case 0: s = &zs.make(std::move(w), 29.f); break;
    case 1: s = &zs.make(std::move(w), 31.f, -31.f); break;
    case 2: s = &zs.make(std::move(w), 31.f, -31.f); break;
    default: s = &zs.make(std::move(w), 31.f); break;
    }
    for (size_t i = 0; i < n; ++i) {
        c.sample(y[i]);
        trig[i] = s->triggered(c);
        timing[i] = (uint8_t)(*s)(c);
        updated[i] = s->updated();
    }
}

void ref_outer_levels(const float* y, size_t n, size_t si, float* mn, float* mx)
{
    Zeroed<Correlator<float>> z;
    auto& c = z.make();
    for (size_t i = 0; i < n; ++i) c.sample(y[i]);
    auto [a, b] = c.outer_symbol_levels(si);
    *mn = a; *mx = b;
}

using RefDcd = DataCarrierDetect<float, 48000, 400>;

size_t ref_dcd_trace(const float* x, size_t n, size_t period, float* level, uint8_t* trig)
{
    Zeroed<RefDcd> z;
    auto& d = z.make(2400, 3600, 0.1, 4.0);
    size_t k = 0;
    for (size_t i = 0; i < n; ++i) {
        d(x[i]);
        if ((i + 1) % period == 0) { d.update(); level[k] = d.level(); trig[k] = d.dcd(); ++k; }
    }
    return k;
}
void ref_dcd_sums(const float* x, size_t start, size_t len, float* l1, float* l2)
{
    Zeroed<RefDcd> z;
    auto& d = z.make(2400, 3600, 0.1, 4.0);
    for (size_t i = 0; i < start + len; ++i) {
        if (i == start) { d.level_1 = 0.f; d.level_2 = 0.f; }
        d(x[i]);
    }
    *l1 = d.level_1; *l2 = d.level_2;
}
void ref_sdft(const float* x, size_t n, float* out4n)  // raw NSlidingDFT outputs (re0, im0, re1, im1)
{
    NSlidingDFT<float, 48000, 120, 2> dft({2400, 3600});
    for (size_t i = 0; i < n; ++i) {
        auto r = dft(x[i]);
        out4n[4 * i] = r[0].real(); out4n[4 * i + 1] = r[0].imag(); out4n[4 * i + 2] = r[1].real(); out4n[4 * i + 3] = r[1].imag();
    }
}
void ref_evm_trace(const float* sym, size_t n, int do_reset, float* out)
{
    SymbolEvm<float> e;
    if (do_reset) e.reset();
    for (size_t i = 0; i < n; ++i) { e.update(sym[i]); out[i] = e.evm(); }
}
void ref_llr(const float* sym, size_t n, int8_t* out2n)
{
    for (size_t i = 0; i < n; ++i) {
        auto [a, b] = llr<float, 4>(sym[i]);
        out2n[2 * i] = a; out2n[2 * i + 1] = b;
    }
}
uint16_t ref_crc16(const uint8_t* d, size_t n)
{
    CRC16<0x5935, 0xFFFF> c;
    c.reset();
    for (size_t i = 0; i < n; ++i) c(d[i]);
    return c.get();
}
uint32_t ref_golay_encode24(uint16_t v) { return Golay24::encode24(v); }
void ref_decode_callsign(const uint8_t* enc6, char* out10)
{
    LinkSetupFrame::encoded_call_t e;
    for (int i = 0; i < 6; ++i) e[i] = enc6[i];
    auto r = LinkSetupFrame::decode_callsign(e);
    for (int i = 0; i < 10; ++i) out10[i] = r[i];
}
void ref_encode_callsign(const char* call, uint8_t* out6)
{
    LinkSetupFrame::call_t c;
    c.fill(0);
    for (int i = 0; i < 9 && call[i]; ++i) c[i] = call[i];
    auto r = LinkSetupFrame::encode_callsign(c);
    for (int i = 0; i < 6; ++i) out6[i] = r[i];
}
int ref_golay_decode(uint32_t in, uint32_t* out) { return Golay24::decode(in, *out) ? 1 : 0; }
void ref_interleave(int8_t* f)
{
    PolynomialInterleaver<45, 92, 368> il;
    std::array<int8_t, 368> a;
    std::memcpy(a.data(), f, 368); il.interleave(a); std::memcpy(f, a.data(), 368);
}
void ref_deinterleave(int8_t* f)
{
    PolynomialInterleaver<45, 92, 368> il;
    std::array<int8_t, 368> a;
    std::memcpy(a.data(), f, 368); il.deinterleave(a); std::memcpy(f, a.data(), 368);
}
void ref_derandomize(int8_t* f)
{
    M17Randomizer<368> r;
    std::array<int8_t, 368> a;
    std::memcpy(a.data(), f, 368); r(a); std::memcpy(f, a.data(), 368);
}
void ref_randomize_bits(int8_t* f)
{
    M17Randomizer<368> r;
    std::array<int8_t, 368> a;
    std::memcpy(a.data(), f, 368); r.randomize(a); std::memcpy(f, a.data(), 368);
}
size_t ref_viterbi(const int8_t* in, size_t IN, uint8_t* out, size_t OUT)
{
    size_t cost = 0;
    if (IN == 488 && OUT == 240) vit<488, 240>(in, out, &cost);
    else if (IN == 296 && OUT == 144) vit<296, 144>(in, out, &cost);
    else if (IN == 420 && OUT == 206) vit<420, 206>(in, out, &cost);
    else if (IN == 402 && OUT == 197) vit<402, 197>(in, out, &cost);
    else return (size_t)-2;
    return cost;
}
size_t ref_depuncture(const int8_t* in, size_t IN, int8_t* out, size_t OUT, int which)
{
    // only the shapes the frame decoder uses
    if (which == 1 && IN == 368 && OUT == 488) { std::array<int8_t, 368> a; std::array<int8_t, 488> o; std::memcpy(a.data(), in, IN); std::memcpy(o.data(), out, OUT); auto r = depuncture(a, o, P1); std::memcpy(out, o.data(), OUT); return r; }
    if (which == 2 && IN == 272 && OUT == 296) { std::array<int8_t, 272> a; std::array<int8_t, 296> o; std::memcpy(a.data(), in, IN); std::memcpy(o.data(), out, OUT); auto r = depuncture(a, o, P2); std::memcpy(out, o.data(), OUT); return r; }
    if (which == 2 && IN == 368 && OUT == 402) { std::array<int8_t, 368> a; std::array<int8_t, 402> o; std::memcpy(a.data(), in, IN); std::memcpy(o.data(), out, OUT); auto r = depuncture(a, o, P2); std::memcpy(out, o.data(), OUT); return r; }
    if (which == 3 && IN == 368 && OUT == 420) { std::array<int8_t, 368> a; std::array<int8_t, 420> o; std::memcpy(a.data(), in, IN); std::memcpy(o.data(), out, OUT); auto r = depuncture(a, o, P3); std::memcpy(out, o.data(), OUT); return r; }
    return (size_t)-2;
}
void ref_prbs9(uint16_t* state, uint8_t* bits, size_t n)
{
    PRBS9 p; p.state = *state;
    for (size_t i = 0; i < n; ++i) bits[i] = p.generate();
    *state = p.state;
}
void ref_bert_count(const uint8_t* payloads25, size_t n_frames, uint32_t* bits, uint32_t* errs, int* synced)
{
    PRBS9 p;
    p.history.fill(0);
    for (size_t f = 0; f < n_frames; ++f) {
        const uint8_t* b = payloads25 + 25 * f;
        for (int j = 0; j < 24; ++j) { uint8_t v = b[j]; for (int i = 0; i < 8; ++i) { p.validate(v & 0x80); v <<= 1; } }
        uint8_t v = b[24];
        for (int i = 0; i < 5; ++i) { p.validate(v & 0x80); v <<= 1; }
    }
    *bits = p.bit_count; *errs = p.err_count; *synced = p.synced;
}

int ref_decode_frame(int sync_type, const int8_t* llr368, uint8_t* state_io, uint8_t* lich_io, uint8_t* lsf_io,
                     int8_t* dep401_io, int64_t* cost_io, ref_frame_rec* recs)
{
    std::vector<ref_frame_rec> outs;
    Zeroed<M17FrameDecoder> z;
    auto& d = z.make([&](const M17FrameDecoder::output_buffer_t& ob, int cost) {
        ref_frame_rec r;
        std::memset(&r, 0, sizeof(r));
        r.cost = cost; r.frame_type = (uint8_t)ob.type; r.sync_type = (uint8_t)sync_type;
        switch (ob.type) {
        case M17FrameDecoder::FrameType::LSF: r.len = 30; std::memcpy(r.payload, ob.lsf.data(), 30); break;
        case M17FrameDecoder::FrameType::LICH: r.len = 6; std::memcpy(r.payload, ob.lich.data(), 6); break;
        case M17FrameDecoder::FrameType::STREAM: r.len = 18; std::memcpy(r.payload, ob.stream.data(), 18); break;
        case M17FrameDecoder::FrameType::BERT: r.len = 25; std::memcpy(r.payload, ob.bert.data(), 25); break;
        default: r.len = 26; std::memcpy(r.payload, ob.packet.data(), 26); break;
        }
        outs.push_back(r);
        return true;
    });
    d.state_ = (M17FrameDecoder::State)*state_io;
    d.lich_segments = *lich_io;
    std::memcpy(d.output_buffer.lsf.data(), lsf_io, 30);
    d.depuncture_buffer.bert[401] = *dep401_io;
    M17FrameDecoder::input_buffer_t buf;
    std::memcpy(buf.data(), llr368, 368);
    size_t cost = (size_t)*cost_io;
    d((M17FrameDecoder::SyncWordType)sync_type, buf, cost);
    *state_io = (uint8_t)d.state_; *lich_io = d.lich_segments;
    std::memcpy(lsf_io, d.output_buffer.lsf.data(), 30);
    *dep401_io = d.depuncture_buffer.bert[401]; *cost_io = (int64_t)cost;
    int n = 0;
    for (auto& r : outs) recs[n++] = r;
    return n;
}

}  // extern "C"

// ---- the composition pin ------------------------------------------------------------------------------------------------
// M17Demodulator.h cannot be compiled here (it includes KalmanFilter.h -> blaze, absent), so what vouches for the orchestrator is
// reading.  What CAN be pinned is everything the orchestrator is made of and how the pieces are wired: the oracle's state machine
// (m17o::DemodulatorT) instantiated over the REFERENCE's own operator objects — BaseFirFilter<float,150>, Correlator<float>,
// SyncWord<Correlator<float>>, DataCarrierDetect<float,48000,400>{2400, 3600, 0.1, 4.0}, SymbolEvm<float>, llr<float,4>,
// M17Framer<368>, M17FrameDecoder (M17Demodulator.h:148-164), each placement-new'ed into zero-filled storage (Q4) — with only
// ClockRecovery and FreqDevEstimator (the two blaze users) from the oracle.  tests/test_oracle_vs_ref.py: hybrid == pure oracle,
// record for record and diagnostic for diagnostic.  What then remains unpinned is the state machine's own 330 lines and the
// 2 x 2 Kalman arithmetic.
namespace {

std::array<float, 150> g_hybrid_taps;   // (set by the entry points before an orchestrator is constructed: detail::Taps lives in M17Demodulator.h)

struct RefOps {
    struct Fir {
        Zeroed<BaseFirFilter<float, 150>> z;
        BaseFirFilter<float, 150>* f;
        Fir() : f(&z.make(g_hybrid_taps)) {}
        float step(float x) { return (*f)(x); }
    };
    struct Carrier {
        using T = DataCarrierDetect<float, 48000, 400>;
        Zeroed<T> z;
        T* p;
        Carrier() : p(&z.make(2400, 3600, 0.1, 4.0)) {}
        void step(float s) { (*p)(s); }
        bool dcd() const { return p->dcd(); }
        float level() const { return p->level(); }
        void update() { p->update(); }
        void unlock() { p->unlock(); }
    };
    struct Evm {
        Zeroed<SymbolEvm<float>> z;
        SymbolEvm<float>* p;
        Evm() : p(&z.make()) {}
        void reset() { p->reset(); }
        void update(float s) { p->update(s); }
        float evm() const { return p->evm(); }
    };
    struct Corr {
        Zeroed<Correlator<float>> z;
        Correlator<float>* p;
        Corr() : p(&z.make()) {}
        void sample(float v) { p->sample(v); }
        size_t index() const { return p->index(); }
        float limit() const { return p->limit(); }
        void outer_symbol_levels(size_t si, float& mn, float& mx) const
        {
            auto r = p->outer_symbol_levels(si);
            mn = std::get<0>(r); mx = std::get<1>(r);
        }
    };
    struct Sync {
        using T = SyncWord<Correlator<float>>;
        Zeroed<T> z;
        T* p;
        Sync(std::initializer_list<int> w, float m1, float m2 = std::numeric_limits<float>::lowest())
        {
            T::buffer_t b;
            int i = 0;
            for (int v : w) b[i++] = (int8_t)v;
            p = &z.make(std::move(b), m1, m2);
        }
        float triggered(const Corr& c) { return p->triggered(*c.p); }
        size_t step(const Corr& c) { return (*p)(*c.p); }
        int8_t updated() { return p->updated(); }
    };
    struct Framer {
        Zeroed<M17Framer<368>> z;
        M17Framer<368>* p;
        Framer() : p(&z.make()) {}
        void reset() { p->reset(); }
        const int8_t* push(float sample)
        {
            auto n = llr<float, 4>(sample);
            int8_t* tmp = nullptr;
            const size_t len = (*p)(n, &tmp);
            return len ? tmp : nullptr;
        }
    };
    template <typename Sink>
    struct Decoder {
        Zeroed<M17FrameDecoder> z;
        M17FrameDecoder* p;
        Sink sink;
        explicit Decoder(Sink s) : sink(s)
        {
            p = &z.make([this](const M17FrameDecoder::output_buffer_t& ob, int cost) {
                m17o::FrameOut f;
                std::memset(&f, 0, sizeof(f));
                f.cost = cost; f.type = (m17o::FrameType)ob.type;
                switch (ob.type) {
                case M17FrameDecoder::FrameType::LSF: f.len = 30; std::memcpy(f.data, ob.lsf.data(), 30); break;
                case M17FrameDecoder::FrameType::LICH: f.len = 6; std::memcpy(f.data, ob.lich.data(), 6); break;
                case M17FrameDecoder::FrameType::STREAM: f.len = 18; std::memcpy(f.data, ob.stream.data(), 18); break;
                case M17FrameDecoder::FrameType::BERT: f.len = 25; std::memcpy(f.data, ob.bert.data(), 25); break;
                default: f.len = 26; std::memcpy(f.data, ob.packet.data(), 26); break;
                }
                sink(f);
                return true;
            });
        }
        void reset() { p->reset(); }
        m17o::DecState state() const { return (m17o::DecState)p->state(); }
        void run(m17o::SyncType t, int8_t* buffer, size_t& cost)
        {
            M17FrameDecoder::input_buffer_t b;
            std::memcpy(b.data(), buffer, 368);
            (*p)((M17FrameDecoder::SyncWordType)t, b, cost);
        }
    };
};
static_assert((int)M17FrameDecoder::State::BERT == (int)m17o::DecState::BERT && (int)M17FrameDecoder::SyncWordType::BERT == (int)m17o::SyncType::BERT &&
              (int)M17FrameDecoder::FrameType::BERT == (int)m17o::FrameType::BERT, "enumerations in the reference's order");

using HybridDemodulator = m17o::DemodulatorT<RefOps>;

struct ref_diag {   // (= m17_diag / m17o_diag)
    int32_t dcd; float evm, deviation, offset; int32_t locked; float clock; int32_t sample_index, sync_index, clock_index, viterbi_cost;
    float dcd_level; uint32_t n_diag, demod_state, n_frames, pad[2];
};
static_assert(sizeof(ref_diag) == 64 && sizeof(ref_frame_rec) == 64, "record layouts");

}  // namespace

extern "C" {

size_t ref_hybrid_demod(const float* taps150, const int16_t* s, size_t n, int invert, ref_frame_rec* recs, size_t cap, ref_diag* diag)
{
    std::memcpy(g_hybrid_taps.data(), taps150, sizeof(float) * 150);
    return m17o::run_channel_t<HybridDemodulator>(s, n, invert, 0u, recs, cap, diag, (float*)nullptr, (size_t)0, (size_t*)nullptr);
}

size_t ref_hybrid_diag_log(const float* taps150, const int16_t* s, size_t n, int invert, ref_diag* log, size_t cap)
{
    std::memcpy(g_hybrid_taps.data(), taps150, sizeof(float) * 150);
    return m17o::diag_log_t<HybridDemodulator>(s, n, invert, log, cap);
}

}  // extern "C"
