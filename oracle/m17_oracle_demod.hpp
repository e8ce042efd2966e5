// ORACLE — TEST INFRASTRUCTURE ONLY (see m17_oracle_dsp.hpp header).
// a13 + a19: M17Framer<368> and the M17Demodulator<float> orchestrator / state
// machine (reference M17Framer.h:42-53, M17Demodulator.h:123-753), restated as a
// per-sample scalar machine.  The orchestrator itself cannot be compiled from the
// reference here (it includes KalmanFilter.h -> blaze, absent), so its parity is
// anchored on: the operators pinned individually (oracle/_ref), the reference's
// KATs, and the end-to-end property of SURVEY Appendix A (a clean BERT burst
// decodes to the PRBS9 payloads with viterbi_cost == 0).
//
// Per-channel semantics for the reference's process-wide statics (Q3):
// `initializing` and `eot_flag` are members, i.e. "one fresh process per channel".
#pragma once

#include "m17_oracle_dsp.hpp"
#include "m17_oracle_fec.hpp"

#include <functional>
#include <memory>
#include <vector>

namespace m17o {

struct FrameRecord {  // one per reference callback invocation
    uint64_t sample_pos;  // 0-based index of the input sample being processed
    int32_t cost;
    uint8_t frame_type;   // FrameType
    uint8_t sync_type;    // SyncType that selected the decode
    uint8_t len;
    uint8_t data[30];
};

struct Diag {  // arguments of the last diagnostic callback (M17Demodulator.h:681-685,746-750)
    int32_t dcd;
    float evm, deviation, offset;
    int32_t locked;
    float clock;
    int32_t sample_index, sync_index, clock_index, viterbi_cost;
    float dcd_level;   // extra: DataCarrierDetect::level()
    uint32_t n_diag;   // number of diagnostic callbacks so far
};

// The operators the orchestrator is made of, as a policy: OracleOps = this directory's restatements (the oracle proper).
// oracle/ref_shim.cpp instantiates the SAME orchestrator over the REFERENCE's own operator classes (BaseFirFilter, Correlator,
// SyncWord, DataCarrierDetect, SymbolEvm, llr, M17Framer, M17FrameDecoder — compiled from the reference's headers where they lie)
// to pin their composition: only ClockRecovery / FreqDevEstimator (KalmanFilter.h -> blaze, absent) stay the oracle's there.
struct OracleOps {
    using Fir = Fir150;
    using Carrier = Dcd;
    using Evm = SymbolEvm;
    using Corr = Correlator;
    using Sync = SyncWord;
    template <typename Sink> using Decoder = FrameDecoder<Sink>;
    struct Framer {  // llr<float,4> (Util.h:128-145) + M17Framer<368> (M17Framer.h:42-53): the completed frame, or nullptr
        int8_t buf[368];
        size_t idx = 0;
        Framer() { reset(); }
        void reset() { std::memset(buf, 0, 368); idx = 0; }
        const int8_t* push(float sample)
        {
            int8_t a, b;
            llr_table().lookup(sample, a, b);
            buf[idx++] = a;
            buf[idx++] = b;
            if (idx == 368) { idx = 0; return buf; }
            return nullptr;
        }
    };
};

template <typename Ops>
struct DemodulatorT {
    enum class St : uint8_t { UNLOCKED, LSF_SYNC, STREAM_SYNC, PACKET_SYNC, BERT_SYNC, SYNC_WAIT, FRAME };
    static constexpr size_t STREAM_COST_LIMIT = 80, PACKET_COST_LIMIT = 60;
    static constexpr int MAX_MISSING_SYNC = 10, MIN_SYNC_COUNT = 78, MAX_SYNC_COUNT = 86;

    typename Ops::Fir fir;
    typename Ops::Carrier dcd;
    ClockRecovery clock;
    typename Ops::Evm evm;
    typename Ops::Corr corr;
    typename Ops::Sync preamble_sync{{+3, -3, +3, -3, +3, -3, +3, -3}, 29.f};
    typename Ops::Sync lsf_sync{{+3, +3, +3, +3, -3, -3, +3, -3}, 31.f, -31.f};
    typename Ops::Sync packet_sync{{3, -3, 3, 3, -3, -3, -3, -3}, 31.f, -31.f};
    typename Ops::Sync eot_sync{{+3, +3, +3, +3, +3, +3, -3, +3}, 31.f};
    FreqDevEstimator dev;
    size_t count_ = 0;
    int8_t polarity = 1;
    typename Ops::Framer framer;
    St st = St::UNLOCKED;
    SyncType sync_word_type = SyncType::LSF;
    uint8_t sample_index = 0;
    bool dcd_ = false, need_clock_reset_ = false, need_clock_update_ = false;
    size_t viterbi_cost = 0;
    int sync_count = 0, missing_sync_count = 0;
    uint8_t sync_sample_index = 0;
    int16_t initializing = 1920;   // per channel (Q3)
    bool eot_flag = false;         // per channel (Q3)

    uint64_t pos = 0;  // index of the sample currently being processed
    std::vector<FrameRecord>* out = nullptr;
    Diag diag{};

    struct Sink {
        DemodulatorT* d;
        void operator()(const FrameOut& f) const
        {
            if (!d->out) return;
            FrameRecord r;
            r.sample_pos = d->pos; r.cost = f.cost; r.frame_type = (uint8_t)f.type;
            r.sync_type = (uint8_t)d->sync_word_type; r.len = f.len;
            std::memcpy(r.data, f.data, 30);
            d->out->push_back(r);
        }
    };
    typename Ops::template Decoder<Sink> decoder{Sink{this}};

    // optional taps for tests: called with (pos, filtered sample) / per symbol
    std::function<void(uint64_t, float, float)> on_symbol;  // pos, normalised symbol, raw filtered
    std::function<void(uint64_t, const Diag&)> on_diag;     // every diagnostic callback: sample position, arguments

    DemodulatorT() {}
    DemodulatorT(const DemodulatorT&) = delete;

    void framer_reset() { framer.reset(); }

    void update_values(uint8_t index)  // :233-241
    {
        float mn, mx;
        corr.outer_symbol_levels(sample_index, mn, mx);
        dev.update(mn, mx);
        sync_sample_index = index;
    }
    void dcd_on()  // :244-257
    {
        dcd_ = true;
        if (st == St::UNLOCKED) {
            sync_count = 0; missing_sync_count = 0;
            framer_reset(); decoder.reset(); evm.reset();
        }
    }
    void dcd_off() { st = St::UNLOCKED; dcd_ = false; }
    void update_dcd()  // :275-286
    {
        if (!dcd_ && dcd.dcd()) { dcd_on(); need_clock_reset_ = true; }
        else if (dcd_ && !dcd.dcd()) dcd_off();
    }
    void fire_diag(float evm_arg)
    {
        diag.dcd = (int)dcd_; diag.evm = evm_arg; diag.deviation = dev.deviation(); diag.offset = dev.offset();
        diag.locked = (st != St::UNLOCKED); diag.clock = clock.clock_estimate();
        diag.sample_index = sample_index; diag.sync_index = sync_sample_index;
        diag.clock_index = clock.sample_index(); diag.viterbi_cost = (int)viterbi_cost;
        diag.dcd_level = dcd.level(); diag.n_diag++;
        if (on_diag) on_diag(pos, diag);
    }

    void do_unlocked()  // :289-342
    {
        if (missing_sync_count < 1920) {
            missing_sync_count += 1;
            size_t si = preamble_sync.step(corr);
            int8_t up = preamble_sync.updated();
            if (up) {
                sync_count = 0; missing_sync_count = 0; need_clock_reset_ = true;
                dev.reset(); sample_index = (uint8_t)si; update_values((uint8_t)si);
                st = St::LSF_SYNC;
            }
            return;
        }
        size_t si = lsf_sync.step(corr);
        int8_t up = lsf_sync.updated();
        if (up) {
            sync_count = MAX_SYNC_COUNT; missing_sync_count = 0; need_clock_reset_ = true;
            dev.reset(); sample_index = (uint8_t)si; update_values((uint8_t)si);
            st = St::FRAME;
            sync_word_type = up < 0 ? SyncType::STREAM : SyncType::LSF;
        }
        si = packet_sync.step(corr);
        up = packet_sync.updated();
        if (up < 0) {
            sync_count = MAX_SYNC_COUNT; missing_sync_count = 0; need_clock_reset_ = true;
            dev.reset(); sample_index = (uint8_t)si; update_values((uint8_t)si);
            st = St::FRAME;
            sync_word_type = SyncType::BERT;
        }
    }
    void do_lsf_sync()  // :350-411
    {
        if (corr.index() != sample_index) return;
        float sync_triggered = preamble_sync.triggered(corr);
        if (sync_triggered > 0.1) { need_clock_update_ = true; sync_count += 1; return; }
        sync_triggered = lsf_sync.triggered(corr);
        float bert_triggered = packet_sync.triggered(corr);
        if (bert_triggered < 0) {
            missing_sync_count = 0; sync_count = MAX_SYNC_COUNT; need_clock_update_ = true;
            update_values(sample_index); st = St::FRAME; sync_word_type = SyncType::BERT;
        } else if (std::fabs(sync_triggered) > 0.1) {
            missing_sync_count = 0; sync_count = MAX_SYNC_COUNT; need_clock_update_ = true;
            update_values(sample_index); st = St::FRAME;
            sync_word_type = sync_triggered > 0 ? SyncType::LSF : SyncType::STREAM;
        } else if (++missing_sync_count > 192) {
            if (sync_count >= 10) { missing_sync_count = 0; need_clock_update_ = true; }
            else { sync_count = 0; st = St::UNLOCKED; missing_sync_count = 0; dcd.unlock(); }
        } else {
            update_values(sample_index);
        }
    }
    void do_stream_sync()  // :420-482
    {
        sync_count += 1;
        if (sync_count < MIN_SYNC_COUNT) return;
        if (eot_sync.triggered(corr) > 0.1f) {  // EOT_TRIGGER_LEVEL is FloatType(0.1)
            sync_word_type = SyncType::STREAM; st = St::FRAME; eot_flag = true; missing_sync_count = 0;
            return;
        }
        uint8_t si = (uint8_t)lsf_sync.step(corr);
        int8_t up = lsf_sync.updated();
        if (up < 0) {
            missing_sync_count = 0; update_values(si);
            sync_word_type = SyncType::STREAM; st = St::SYNC_WAIT; eot_flag = false;
        } else if (sync_count > MAX_SYNC_COUNT) {
            if (viterbi_cost < STREAM_COST_LIMIT) {
                if (!missing_sync_count) missing_sync_count = 1;
                sync_word_type = SyncType::STREAM; st = St::FRAME;
            } else if (eot_flag) {
                st = St::UNLOCKED; dcd.unlock();
            } else if (missing_sync_count < MAX_MISSING_SYNC) {
                missing_sync_count += 1; sync_word_type = SyncType::STREAM; st = St::FRAME;
            } else {
                st = St::UNLOCKED; dcd.unlock();
            }
            eot_flag = false;
        }
    }
    void do_packet_sync()  // :489-530
    {
        sync_count += 1;
        if (sync_count < MIN_SYNC_COUNT) return;
        uint8_t si = (uint8_t)packet_sync.step(corr);
        int8_t up = packet_sync.updated();
        if (up) {
            missing_sync_count = 0; update_values(si);
            sync_word_type = SyncType::PACKET; st = St::SYNC_WAIT;
        } else if (sync_count > MAX_SYNC_COUNT) {
            if (viterbi_cost < PACKET_COST_LIMIT) {
                if (!missing_sync_count) missing_sync_count = 1;
                sync_word_type = SyncType::PACKET; st = St::FRAME;
            } else if (missing_sync_count < MAX_MISSING_SYNC) {
                missing_sync_count += 1; sync_word_type = SyncType::PACKET; st = St::FRAME;
            } else { st = St::UNLOCKED; dcd.unlock(); }
        }
    }
    void do_bert_sync()  // :536-574
    {
        sync_count += 1;
        if (sync_count < MIN_SYNC_COUNT) return;
        uint8_t si = (uint8_t)packet_sync.step(corr);
        int8_t up = packet_sync.updated();
        if (up < 0) {
            missing_sync_count = 0; update_values(si);
            sync_word_type = SyncType::BERT; st = St::SYNC_WAIT;
        } else if (sync_count > MAX_SYNC_COUNT) {
            if (viterbi_cost < STREAM_COST_LIMIT) {
                if (!missing_sync_count) missing_sync_count = 1;
                sync_word_type = SyncType::BERT; st = St::FRAME;
            } else if (missing_sync_count < MAX_MISSING_SYNC) {
                missing_sync_count += 1; sync_word_type = SyncType::BERT; st = St::FRAME;
            } else { st = St::UNLOCKED; dcd.unlock(); }
        }
    }
    void do_sync_wait()  // :583-593
    {
        if (sync_count < MAX_SYNC_COUNT) { sync_count += 1; return; }
        need_clock_update_ = true;
        st = St::FRAME;
    }
    void do_frame(float filtered)  // :596-654
    {
        int d = (int)sample_index - (int)corr.index();
        if (std::abs(d) == 5) {
            clock.update();
            sample_index = clock.sample_index();
            return;
        }
        if (corr.index() != sample_index) return;
        float sample = filtered - dev.offset();
        sample = sample * dev.idev();
        sample = sample * (float)polarity;
        evm.update(sample);
        if (on_symbol) on_symbol(pos, sample, filtered);
        if (const int8_t* frame = framer.push(sample)) {   // llr<FloatType,4>(sample) + framer(n, &tmp), :617-619
            sync_count = 0;
            int8_t buffer[368];
            std::memcpy(buffer, frame, 368);
            decoder.run(sync_word_type, buffer, viterbi_cost);
            switch (decoder.state()) {
            case DecState::STREAM: st = St::STREAM_SYNC; break;
            case DecState::LSF: st = St::STREAM_SYNC; break;
            case DecState::BERT: st = St::BERT_SYNC; break;
            default: st = St::PACKET_SYNC; break;
            }
        }
    }

    void step(float input)  // operator(), :657-753
    {
        count_++;
        dcd.step(input);
        if (initializing) {
            --initializing;
            float f = fir.step(input);
            corr.sample(f);
            count_ = 0;
            return;
        }
        if (!dcd_) {
            if (count_ % 384 == 0) {
                update_dcd();
                dcd.update();
                fire_diag(dev.error());
                count_ = 0;
            }
            return;
        }
        float filtered = fir.step(input);
        corr.sample(filtered);
        if (corr.index() == 0) {
            if (need_clock_reset_) {
                clock.reset((float)sync_sample_index);
                need_clock_reset_ = false;
                sample_index = sync_sample_index;
            } else if (need_clock_update_) {
                clock.update(sync_sample_index);
                need_clock_update_ = false;
            }
        }
        clock.tick();
        switch (st) {
        case St::UNLOCKED: do_unlocked(); break;
        case St::LSF_SYNC: do_lsf_sync(); break;
        case St::STREAM_SYNC: do_stream_sync(); break;
        case St::PACKET_SYNC: do_packet_sync(); break;
        case St::BERT_SYNC: do_bert_sync(); break;
        case St::SYNC_WAIT: do_sync_wait(); break;
        case St::FRAME: do_frame(filtered); break;
        }
        if (count_ % 960 == 0) {
            update_dcd();
            count_ = 0;
            fire_diag(evm.evm());
            dcd.update();
        }
    }

    void run(const int16_t* s, size_t n, bool invert)
    {
        for (size_t i = 0; i < n; ++i) { step(scale_sample(s[i], invert)); pos++; }
    }
};
using Demodulator = DemodulatorT<OracleOps>;

// One channel through a demodulator of type D: frame records (Rec: the 64-byte record of the C APIs), the last diagnostic callback
// (DiagOut: their 64-byte diagnostic record), optionally the normalised symbols.  Shared by oracle/m17_oracle_capi.cpp (D = Demodulator)
// and oracle/ref_shim.cpp (D = the orchestrator over the reference's operators).
template <typename D, typename Rec, typename DiagOut>
inline size_t run_channel_t(const int16_t* s, size_t n, int invert, uint32_t channel, Rec* recs, size_t cap, DiagOut* diag, float* sym_out, size_t sym_cap,
                            size_t* n_sym)
{
    std::vector<FrameRecord> out;
    auto d = std::make_unique<D>();
    d->out = &out;
    size_t ns = 0;
    if (sym_out) d->on_symbol = [&](uint64_t, float sym, float) { if (ns < sym_cap) sym_out[ns] = sym; ns++; };
    d->run(s, n, invert != 0);
    size_t cnt = 0;
    for (auto& f : out) {
        if (cnt < cap) {
            Rec& r = recs[cnt];
            std::memset(&r, 0, sizeof(r));
            r.channel = channel; r.seq = (uint32_t)cnt; r.sample_pos = f.sample_pos; r.cost = f.cost;
            r.frame_type = f.frame_type; r.sync_type = f.sync_type; r.len = f.len;
            std::memcpy(r.payload, f.data, 30);
        }
        cnt++;
    }
    if (diag) {
        std::memset(diag, 0, sizeof(*diag));
        const Diag& g = d->diag;
        diag->dcd = g.dcd; diag->evm = g.evm; diag->deviation = g.deviation; diag->offset = g.offset;
        diag->locked = g.locked; diag->clock = g.clock; diag->sample_index = g.sample_index;
        diag->sync_index = g.sync_index; diag->clock_index = g.clock_index; diag->viterbi_cost = g.viterbi_cost;
        diag->dcd_level = g.dcd_level; diag->n_diag = g.n_diag; diag->demod_state = (uint32_t)d->st;
        diag->n_frames = (uint32_t)cnt;
        // live counters at the end of the run (debugging aid; the HIP path fills the same words)
        diag->pad[0] = (uint32_t)d->clock.count;
        diag->pad[1] = ((uint32_t)d->sync_count & 0xFFFFu) | ((uint32_t)d->missing_sync_count << 16);
    }
    if (n_sym) *n_sym = ns;
    return cnt;
}

// Every diagnostic callback of one channel, in order (same layout as the log entries of m17hip_diag_log_fetch: demod_state and
// n_frames at that moment, pad[0] | pad[1] << 32 = the sample that fired it).  Returns the number of callbacks.
template <typename D, typename DiagOut>
inline size_t diag_log_t(const int16_t* s, size_t n, int invert, DiagOut* log, size_t cap)
{
    std::vector<FrameRecord> out;
    auto d = std::make_unique<D>();
    d->out = &out;
    size_t cnt = 0;
    D* dp = d.get();
    d->on_diag = [&](uint64_t pos, const Diag& g) {
        if (cnt < cap) {
            DiagOut& o = log[cnt];
            std::memset(&o, 0, sizeof(o));
            o.dcd = g.dcd; o.evm = g.evm; o.deviation = g.deviation; o.offset = g.offset; o.locked = g.locked; o.clock = g.clock;
            o.sample_index = g.sample_index; o.sync_index = g.sync_index; o.clock_index = g.clock_index; o.viterbi_cost = g.viterbi_cost;
            o.dcd_level = g.dcd_level; o.n_diag = g.n_diag; o.demod_state = (uint32_t)dp->st; o.n_frames = (uint32_t)out.size();
            o.pad[0] = (uint32_t)pos; o.pad[1] = (uint32_t)(pos >> 32);
        }
        ++cnt;
    };
    d->run(s, n, invert != 0);
    return cnt;
}

}  // namespace m17o
