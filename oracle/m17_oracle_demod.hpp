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

struct Demodulator {
    enum class St : uint8_t { UNLOCKED, LSF_SYNC, STREAM_SYNC, PACKET_SYNC, BERT_SYNC, SYNC_WAIT, FRAME };
    static constexpr size_t STREAM_COST_LIMIT = 80, PACKET_COST_LIMIT = 60;
    static constexpr int MAX_MISSING_SYNC = 10, MIN_SYNC_COUNT = 78, MAX_SYNC_COUNT = 86;

    Fir150 fir;
    Dcd dcd;
    ClockRecovery clock;
    SymbolEvm evm;
    Correlator corr;
    SyncWord preamble_sync{{+3, -3, +3, -3, +3, -3, +3, -3}, 29.f};
    SyncWord lsf_sync{{+3, +3, +3, +3, -3, -3, +3, -3}, 31.f, -31.f};
    SyncWord packet_sync{{3, -3, 3, 3, -3, -3, -3, -3}, 31.f, -31.f};
    SyncWord eot_sync{{+3, +3, +3, +3, +3, +3, -3, +3}, 31.f};
    FreqDevEstimator dev;
    size_t count_ = 0;
    int8_t polarity = 1;
    int8_t framer[368];
    size_t framer_idx = 0;
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
        Demodulator* d;
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
    FrameDecoder<Sink> decoder{Sink{this}};

    // optional taps for tests: called with (pos, filtered sample) / per symbol
    std::function<void(uint64_t, float, float)> on_symbol;  // pos, normalised symbol, raw filtered
    std::function<void(uint64_t, const Diag&)> on_diag;     // every diagnostic callback: sample position, arguments

    Demodulator() { std::memset(framer, 0, 368); }
    Demodulator(const Demodulator&) = delete;

    void framer_reset() { std::memset(framer, 0, 368); framer_idx = 0; }

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
        int8_t a, b;
        llr_table().lookup(sample, a, b);
        framer[framer_idx++] = a;
        framer[framer_idx++] = b;
        if (framer_idx == 368) {
            framer_idx = 0;
            sync_count = 0;
            int8_t buffer[368];
            std::memcpy(buffer, framer, 368);
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

}  // namespace m17o
