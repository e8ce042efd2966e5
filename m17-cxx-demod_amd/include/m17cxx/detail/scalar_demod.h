// The demodulator's orchestrator in scalar form, one sample per call, on the HOST: the operator classes of this directory
// (BaseFirFilter, Correlator / SyncWord, DataCarrierDetect, ClockRecovery, FreqDevEstimator, SymbolEvm, M17Framer,
// M17FrameDecoder) wired together the way the reference's M17Demodulator<FloatType>::operator() wires them
// (include/m17cxx/M17Demodulator.h:233-753).  It is what mobilinkd::M17Demodulator runs when it is told to stay off the GPU
// (BASELINE configs[0]: one stream, no GPU), and it puts the arithmetic cores of detail/core.h — the ones the HIP kernels are
// built from — under a host-testable orchestrator: tests/test_cxx_mirror.py compares its callback sequence with the oracle's.
//
// Behaviour notes: the reference's two function-local statics (`initializing`, `eot_flag`, Q3) are members here —
// one demodulator = one fresh reference process; objects are value-initialised (Q4); the carrier-detect gate freezes the matched
// filter and the correlator while the carrier is off (Q2).
#pragma once

#include "../ClockRecovery.h"
#include "../Correlator.h"
#include "../DataCarrierDetect.h"
#include "../FirFilter.h"
#include "../FreqDevEstimator.h"
#include "../M17FrameDecoder.h"
#include "../M17Framer.h"
#include "../SymbolEvm.h"
#include "../Util.h"

#include <array>
#include <cmath>
#include <cstdint>
#include <functional>

namespace mobilinkd
{
namespace detail
{

template <typename FloatType>
class ScalarDemodulator
{
public:
    using correlator_t = Correlator<FloatType>;
    using sync_word_t = SyncWord<correlator_t>;
    using frame_callback_t = M17FrameDecoder::callback_t;
    using diag_callback_t = std::function<void(bool, FloatType, FloatType, FloatType, bool, FloatType, int, int, int, int)>;
    enum class State : uint8_t { UNLOCKED, LSF_SYNC, STREAM_SYNC, PACKET_SYNC, BERT_SYNC, SYNC_WAIT, FRAME };

    ScalarDemodulator(const std::array<FloatType, 150>& rrc_taps, frame_callback_t on_frame)
    : matched_filter_(rrc_taps), decoder_(std::move(on_frame))
    {}

    diag_callback_t on_diagnostics;

    State state() const { return state_; }
    bool carrier() const { return carrier_; }

    // M17Demodulator::operator() :657-753
    void step(FloatType input)
    {
        ++since_update_;
        carrier_detect_(input);                       // the sliding DFT sees every sample

        if (warm_up_ != 0) {                          // :667-673 the first 1920 samples only fill the filter and the correlator
            --warm_up_;
            correlator_.sample(matched_filter_(input));
            since_update_ = 0;
            return;
        }
        if (!carrier_) {                              // :675-689 gate closed: nothing moves but the carrier detector
            if (since_update_ % OFF_PERIOD == 0) {
                follow_carrier_detect();
                carrier_detect_.update();
                report(deviation_.error());
                since_update_ = 0;
            }
            return;
        }

        const FloatType y = matched_filter_(input);
        correlator_.sample(y);
        if (correlator_.index() == 0) {               // :695-709 pending clock work is done where a symbol period begins
            if (clock_reset_due_) {
                clock_.reset(FloatType(sync_timing_));
                clock_reset_due_ = false;
                symbol_phase_ = sync_timing_;
            } else if (clock_update_due_) {
                clock_.update(sync_timing_);
                clock_update_due_ = false;
            }
        }
        clock_(y);

        switch (state_) {
        case State::UNLOCKED: search(); break;
        case State::LSF_SYNC: after_preamble(); break;
        case State::STREAM_SYNC: next_sync(Seek{&stream_word_, -1, STREAM_COST_LIMIT, M17FrameDecoder::SyncWordType::STREAM, true}); break;
        case State::PACKET_SYNC: next_sync(Seek{&packet_word_, 0, PACKET_COST_LIMIT, M17FrameDecoder::SyncWordType::PACKET, false}); break;
        case State::BERT_SYNC: next_sync(Seek{&packet_word_, -1, STREAM_COST_LIMIT, M17FrameDecoder::SyncWordType::BERT, false}); break;
        case State::SYNC_WAIT: wait_for_payload(); break;
        case State::FRAME: payload(y); break;
        }

        if (since_update_ % ON_PERIOD == 0) {          // :742-752
            follow_carrier_detect();
            since_update_ = 0;
            report(evm_.evm());
            carrier_detect_.update();
        }
    }

private:
    static constexpr size_t OFF_PERIOD = 384, ON_PERIOD = 960;          // samples between carrier-detect decisions
    static constexpr size_t STREAM_COST_LIMIT = 80, PACKET_COST_LIMIT = 60;
    static constexpr int MAX_MISSING_SYNC = 10, FIRST_SYNC_SAMPLE = 78, LAST_SYNC_SAMPLE = 86;

    // what a *_SYNC state looks for after a frame: the word, the sign its peak must have (0 = either), the Viterbi cost below
    // which a missing word is forgiven, the frame type that follows, and whether an end-of-transmission word ends the search
    struct Seek {
        sync_word_t* word;
        int sign;
        size_t cost_limit;
        M17FrameDecoder::SyncWordType type;
        bool watch_eot;
    };

    void measure_levels(uint8_t timing)               // update_values :233-241
    {
        const auto [lo, hi] = correlator_.outer_symbol_levels(symbol_phase_);
        deviation_.update(lo, hi);
        sync_timing_ = timing;
    }
    void follow_carrier_detect()                      // update_dcd :275-286 with dcd_on :244-257 / dcd_off :260-265
    {
        const bool detected = carrier_detect_.dcd();
        if (!carrier_ && detected) {
            carrier_ = true;
            if (state_ == State::UNLOCKED) {
                sync_count_ = 0;
                missing_syncs_ = 0;
                framer_.reset();
                decoder_.reset();
                evm_.reset();
            }
            clock_reset_due_ = true;
        } else if (carrier_ && !detected) {
            state_ = State::UNLOCKED;
            carrier_ = false;
        }
    }
    void report(FloatType evm_value)
    {
        if (!on_diagnostics) return;
        on_diagnostics(carrier_, evm_value, deviation_.deviation(), deviation_.offset(), state_ != State::UNLOCKED, clock_.clock_estimate(),
                       int(symbol_phase_), int(sync_timing_), int(clock_.sample_index()), int(viterbi_cost_));
    }
    void lock_on(uint8_t timing, M17FrameDecoder::SyncWordType type)   // a sync word found with no preamble before it
    {
        sync_count_ = LAST_SYNC_SAMPLE;
        missing_syncs_ = 0;
        clock_reset_due_ = true;
        deviation_.reset();
        symbol_phase_ = timing;
        measure_levels(timing);
        state_ = State::FRAME;
        word_type_ = type;
    }
    void give_up()                                    // back to the search, and the carrier detector has to prove itself again
    {
        state_ = State::UNLOCKED;
        carrier_detect_.unlock();
    }

    void search()                                     // do_unlocked :289-342
    {
        if (missing_syncs_ < 1920) {                  // first a preamble, for one frame's worth of samples
            ++missing_syncs_;
            const auto timing = uint8_t(preamble_word_(correlator_));
            if (preamble_word_.updated()) {
                sync_count_ = 0;
                missing_syncs_ = 0;
                clock_reset_due_ = true;
                deviation_.reset();
                symbol_phase_ = timing;
                measure_levels(timing);
                state_ = State::LSF_SYNC;
            }
            return;
        }
        auto timing = uint8_t(stream_word_(correlator_));       // the LSF word; its negative is the stream word
        if (const int8_t peak = stream_word_.updated())
            lock_on(timing, peak < 0 ? M17FrameDecoder::SyncWordType::STREAM : M17FrameDecoder::SyncWordType::LSF);
        timing = uint8_t(packet_word_(correlator_));            // the packet word; only its negative (BERT) counts here
        if (packet_word_.updated() < 0) lock_on(timing, M17FrameDecoder::SyncWordType::BERT);
    }

    void after_preamble()                             // do_lsf_sync :350-411, once per symbol
    {
        if (correlator_.index() != symbol_phase_) return;
        if (double(preamble_word_.triggered(correlator_)) > 0.1) {    // still preamble
            clock_update_due_ = true;
            ++sync_count_;
            return;
        }
        const FloatType lsf = stream_word_.triggered(correlator_);
        const FloatType bert = packet_word_.triggered(correlator_);
        auto start_frame = [&](M17FrameDecoder::SyncWordType type) {
            missing_syncs_ = 0;
            sync_count_ = LAST_SYNC_SAMPLE;
            clock_update_due_ = true;
            measure_levels(symbol_phase_);
            state_ = State::FRAME;
            word_type_ = type;
        };
        if (bert < 0) start_frame(M17FrameDecoder::SyncWordType::BERT);
        else if (double(std::fabs(lsf)) > 0.1) start_frame(lsf > 0 ? M17FrameDecoder::SyncWordType::LSF : M17FrameDecoder::SyncWordType::STREAM);
        else if (++missing_syncs_ > 192) {            // a frame's worth of symbols without any word
            if (sync_count_ >= 10) {
                missing_syncs_ = 0;
                clock_update_due_ = true;
            } else {
                sync_count_ = 0;
                missing_syncs_ = 0;
                give_up();
            }
        } else {
            measure_levels(symbol_phase_);
        }
    }

    void next_sync(const Seek& seek)                  // do_stream_sync :420-482, do_packet_sync :489-530, do_bert_sync :536-574
    {
        ++sync_count_;
        if (sync_count_ < FIRST_SYNC_SAMPLE) return;
        if (seek.watch_eot && eot_word_.triggered(correlator_) > EOT_LEVEL) {
            word_type_ = seek.type;
            state_ = State::FRAME;
            eot_seen_ = true;
            missing_syncs_ = 0;
            return;
        }
        const auto timing = uint8_t((*seek.word)(correlator_));
        const int8_t peak = seek.word->updated();
        if (seek.sign == 0 ? peak != 0 : peak < 0) {
            missing_syncs_ = 0;
            measure_levels(timing);
            word_type_ = seek.type;
            state_ = State::SYNC_WAIT;
            if (seek.watch_eot) eot_seen_ = false;
        } else if (sync_count_ > LAST_SYNC_SAMPLE) {  // no word where one was due
            if (viterbi_cost_ < seek.cost_limit) {    // the last frame was good: assume the word was there
                if (missing_syncs_ == 0) missing_syncs_ = 1;
                word_type_ = seek.type;
                state_ = State::FRAME;
            } else if (seek.watch_eot && eot_seen_) {
                give_up();
            } else if (missing_syncs_ < MAX_MISSING_SYNC) {
                ++missing_syncs_;
                word_type_ = seek.type;
                state_ = State::FRAME;
            } else {
                give_up();
            }
            if (seek.watch_eot) eot_seen_ = false;
        }
    }

    void wait_for_payload()                           // do_sync_wait :583-593
    {
        if (sync_count_ < LAST_SYNC_SAMPLE) {
            ++sync_count_;
            return;
        }
        clock_update_due_ = true;
        state_ = State::FRAME;
    }

    void payload(FloatType y)                         // do_frame :596-654
    {
        const int lag = int(symbol_phase_) - int(correlator_.index());
        if (lag == 5 || lag == -5) {                  // half a symbol away from the sampling point: let the clock move it
            clock_.update();
            symbol_phase_ = clock_.sample_index();
            return;
        }
        if (correlator_.index() != symbol_phase_) return;
        FloatType symbol = y - deviation_.offset();
        symbol = symbol * deviation_.idev();
        symbol = symbol * FloatType(polarity_);
        evm_.update(symbol);
        int8_t* full = nullptr;
        if (framer_(llr<FloatType, 4>(symbol), &full) == 0) return;
        M17FrameDecoder::input_buffer_t frame;
        std::copy(full, full + frame.size(), frame.begin());
        sync_count_ = 0;
        decoder_(word_type_, frame, viterbi_cost_);
        switch (decoder_.state()) {
        case M17FrameDecoder::State::LSF:
        case M17FrameDecoder::State::STREAM: state_ = State::STREAM_SYNC; break;
        case M17FrameDecoder::State::BERT: state_ = State::BERT_SYNC; break;
        default: state_ = State::PACKET_SYNC; break;
        }
    }

    static constexpr FloatType EOT_LEVEL = FloatType(0.1);

    BaseFirFilter<FloatType, 150> matched_filter_;
    DataCarrierDetect<FloatType, 48000, 400> carrier_detect_{2400, 3600, FloatType(0.1), FloatType(4.0)};
    ClockRecovery<FloatType, 10> clock_;
    SymbolEvm<FloatType> evm_;
    correlator_t correlator_;
    sync_word_t preamble_word_{{+3, -3, +3, -3, +3, -3, +3, -3}, FloatType(29)};
    sync_word_t stream_word_{{+3, +3, +3, +3, -3, -3, +3, -3}, FloatType(31), FloatType(-31)};     // positive: LSF, negative: stream
    sync_word_t packet_word_{{+3, -3, +3, +3, -3, -3, -3, -3}, FloatType(31), FloatType(-31)};     // positive: packet, negative: BERT
    sync_word_t eot_word_{{+3, +3, +3, +3, +3, +3, -3, +3}, FloatType(31)};
    FreqDevEstimator<FloatType> deviation_;
    M17Framer<368> framer_;
    M17FrameDecoder decoder_;

    State state_ = State::UNLOCKED;
    M17FrameDecoder::SyncWordType word_type_ = M17FrameDecoder::SyncWordType::LSF;
    size_t since_update_ = 0;
    size_t viterbi_cost_ = 0;
    int sync_count_ = 0, missing_syncs_ = 0;
    int16_t warm_up_ = 1920;
    uint8_t symbol_phase_ = 0, sync_timing_ = 0;
    int8_t polarity_ = 1;
    bool carrier_ = false, clock_reset_due_ = false, clock_update_due_ = false, eot_seen_ = false;
};

} // detail
} // mobilinkd
