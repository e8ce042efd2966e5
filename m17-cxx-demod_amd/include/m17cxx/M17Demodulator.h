// mobilinkd::M17Demodulator<FloatType> — the reference's demodulator object (include/m17cxx/M17Demodulator.h:123-217) as
// apps/m17-demod.cpp drives it: construct with the frame callback (:455), install a diagnostic callback (:478), push one scaled
// sample per call (:484-490).  This one runs on the MI355X: samples are collected into blocks (default 1920 = one M17 frame,
// 40 ms), every full block goes through the GPU chain (include/m17hip.h; matched filter, carrier detect, correlator / clock
// recovery state machine, Viterbi) with the demodulator state carried from block to block, and the callbacks are then
// delivered in exactly the reference's order: every frame callback and every diagnostic callback (one per 960 samples while
// the carrier is on, one per 384 while it is off), ordered by the sample that fired them, frame callbacks of a sample first.
// What a caller can observe beyond that: callbacks arrive up to one block late, and the end of the input needs flush() —
// the destructor calls it, so a stock read loop loses nothing.  There is no silent CPU fallback: without libm17hip.so / a GPU the
// constructor throws.  For thousands of channels at once use BatchedDemodulator.h (this class is the 1-channel case of it).
//
// The same object can be told to stay on the HOST (BASELINE configs[0]: one stream, no GPU): construct it with the tag
// `mobilinkd::scalar_cpu`, or — for an application that is not to be touched, like the stock apps/m17-demod.cpp — export
// M17_DEMOD_DEVICE=cpu (M17_DEMOD_DEVICE=<n> picks GPU n; unset = GPU 0).  It then runs detail/scalar_demod.h: the operator classes
// of this directory, i.e. the arithmetic cores the kernels are built from, one sample per call, callbacks delivered at once.
//
// Like the reference's header this one pulls in the whole operator surface (Correlator, FirFilter, DataCarrierDetect,
// ClockRecovery, FreqDevEstimator, M17FrameDecoder with LinkSetupFrame / CRC16 / Viterbi / Trellis, M17Framer, SymbolEvm,
// Util with PRBS9 and llr), so an application written against the reference compiles unchanged.
#pragma once

#include "BatchedDemodulator.h"
#include "ClockRecovery.h"
#include "Correlator.h"
#include "DataCarrierDetect.h"
#include "FirFilter.h"
#include "FreqDevEstimator.h"
#include "M17FrameDecoder.h"
#include "M17Framer.h"
#include "SymbolEvm.h"
#include "Util.h"
#include "detail/scalar_demod.h"

#include <algorithm>
#include <cstdlib>
#include <memory>
#include <array>
#include <cmath>
#include <cstring>
#include <functional>
#include <optional>
#include <tuple>
#include <vector>

namespace mobilinkd
{

namespace detail
{

// The demodulator's matched filter: root-raised-cosine, alpha = 0.5, 10 samples per symbol, 150 taps (the last one 0.0).
// Reference M17Demodulator.h:29-118; the values live in detail/rrc_half_taps.inc (75 distinct ones, the filter is symmetric).
template <typename FloatType>
struct Taps
{
    static constexpr std::array<FloatType, 150> make()
    {
        std::array<FloatType, 150> t{};
        for (size_t i = 0; i != 149; ++i) t[i] = FloatType(i <= 74 ? core::RRC_HALF[i] : core::RRC_HALF[148 - i]);
        return t;
    }
    static constexpr auto rrc_taps = make();
};

} // detail

struct scalar_cpu_t { explicit scalar_cpu_t() = default; };
inline constexpr scalar_cpu_t scalar_cpu{};   // M17Demodulator<float> demod(callback, scalar_cpu): the host form, no GPU involved

template <typename FloatType>
struct M17Demodulator
{
    static constexpr uint16_t SAMPLE_RATE = 48000;
    static constexpr uint16_t SYMBOL_RATE = 4800;
    static constexpr uint16_t SAMPLES_PER_SYMBOL = SAMPLE_RATE / SYMBOL_RATE;
    static constexpr uint16_t BLOCK_SIZE = 192;

    static constexpr FloatType sample_rate = SAMPLE_RATE;
    static constexpr FloatType symbol_rate = SYMBOL_RATE;

    static constexpr size_t STREAM_COST_LIMIT = 80;
    static constexpr size_t PACKET_COST_LIMIT = 60;
    static constexpr uint8_t MAX_MISSING_SYNC = 10;
    static constexpr uint8_t MIN_SYNC_COUNT = 78;
    static constexpr uint8_t MAX_SYNC_COUNT = 86;
    static constexpr FloatType EOT_TRIGGER_LEVEL = 0.1;

    using collelator_t = Correlator<FloatType>;
    using sync_word_t = SyncWord<collelator_t>;
    using callback_t = M17FrameDecoder::callback_t;
    using diagnostic_callback_t = std::function<void(bool, FloatType, FloatType, FloatType, bool, FloatType, int, int, int, int)>;

    enum class DemodState { UNLOCKED, LSF_SYNC, STREAM_SYNC, PACKET_SYNC, BERT_SYNC, SYNC_WAIT, FRAME };

    DemodState demodState = DemodState::UNLOCKED;   // state at the end of the last block

    // the reference's constructor (M17Demodulator.h:180-182): where it runs is the environment's choice (see the header comment)
    explicit M17Demodulator(callback_t callback) : callback_(std::move(callback)), block_(1920)
    {
        const char* where = std::getenv("M17_DEMOD_DEVICE");
        if (where && std::string(where) == "cpu") start_cpu();
        else start_gpu(where && *where ? std::atoi(where) : 0);
    }
    M17Demodulator(callback_t callback, uint32_t block_samples, int device = 0) : callback_(std::move(callback)), block_(block_samples)
    {
        start_gpu(device);
    }
    M17Demodulator(callback_t callback, scalar_cpu_t) : callback_(std::move(callback)), block_(1920) { start_cpu(); }

    virtual ~M17Demodulator()
    {
        try { flush(); } catch (...) {}
    }

    // The reference takes sample / 41067.0 (apps/m17-demod.cpp:489); the int16 the GPU path scales itself is recovered exactly.
    void operator()(const FloatType input)
    {
        if (cpu_) {   // the host form: this very sample, callbacks before the call returns
            cpu_->step(input);
            demodState = (DemodState)cpu_->state();
            return;
        }
        buffer_.push_back((int16_t)std::lrint((double)input * 41067.0));
        if (buffer_.size() == block_) run_block();
    }

    // demodulate what is buffered (end of input); safe to call at any time
    void flush() { if (gpu_ && !buffer_.empty()) run_block(); }
    bool on_gpu() const { return gpu_ != nullptr; }

    bool locked() const { return dcd_; }
    void passall(bool enabled) { passall_ = enabled; }
    void diagnostics(diagnostic_callback_t callback) { diagnostic_callback = std::move(callback); }

    diagnostic_callback_t diagnostic_callback;

private:
    void deliver(const m17_frame_rec& r)
    {
        M17FrameDecoder::output_buffer_t ob;
        std::memset(&ob, 0, sizeof(ob));
        ob.type = (M17FrameDecoder::FrameType)r.frame_type;
        switch (ob.type) {
        case M17FrameDecoder::FrameType::LSF: std::memcpy(ob.lsf.data(), r.payload, 30); break;
        case M17FrameDecoder::FrameType::LICH: std::memcpy(ob.lich.data(), r.payload, 6); break;
        case M17FrameDecoder::FrameType::STREAM: std::memcpy(ob.stream.data(), r.payload, 18); break;
        case M17FrameDecoder::FrameType::BERT: std::memcpy(ob.bert.data(), r.payload, 25); break;
        default: std::memcpy(ob.packet.data(), r.payload, 26); break;
        }
        if (callback_) callback_(ob, r.cost);
    }
    void deliver(const m17_diag& d)
    {
        dcd_ = d.dcd != 0;
        if (diagnostic_callback)
            diagnostic_callback(d.dcd != 0, (FloatType)d.evm, (FloatType)d.deviation, (FloatType)d.offset, d.locked != 0, (FloatType)d.clock,
                                d.sample_index, d.sync_index, d.clock_index, d.viterbi_cost);
    }
    void start_gpu(int device)
    {
        gpu_ = std::make_unique<BatchedDemodulator>(1, block_, device);
        buffer_.reserve(block_);
        gpu_->enable_diag_log(block_ / 384 + 2);
        gpu_->reset();
    }
    void start_cpu()
    {
        cpu_ = std::make_unique<detail::ScalarDemodulator<FloatType>>(detail::Taps<FloatType>::rrc_taps,
            [this](const M17FrameDecoder::output_buffer_t& f, int cost) { return callback_ ? callback_(f, cost) : true; });
        cpu_->on_diagnostics = [this](bool dcd, FloatType evm, FloatType dev, FloatType off, bool locked, FloatType clock, int si, int sy, int ci, int vc) {
            dcd_ = dcd;
            if (diagnostic_callback) diagnostic_callback(dcd, evm, dev, off, locked, clock, si, sy, ci, vc);
        };
    }
    void run_block()
    {
        auto& gpu_ = *this->gpu_;
        gpu_.upload(buffer_.data(), 1, (uint32_t)buffer_.size(), buffer_.size());
        gpu_.run();
        const auto frames = gpu_.frames();
        const auto diags = gpu_.diag_log();
        // merge by the sample that fired the callback; within one sample the reference calls the frame callback(s) first
        size_t f = 0, g = 0;
        while (f < frames.size() || g < diags.size()) {
            const uint64_t fp = f < frames.size() ? frames[f].sample_pos : ~0ull;
            const uint64_t gp = g < diags.size() ? (uint64_t)diags[g].pad[0] | ((uint64_t)diags[g].pad[1] << 32) : ~0ull;
            if (f < frames.size() && fp <= gp) deliver(frames[f++]);
            else deliver(diags[g++]);
        }
        demodState = (DemodState)gpu_.diagnostics()[0].demod_state;
        buffer_.clear();
    }

    std::unique_ptr<BatchedDemodulator> gpu_;                        // one of the two
    std::unique_ptr<detail::ScalarDemodulator<FloatType>> cpu_;
    callback_t callback_;
    std::vector<int16_t> buffer_;
    uint32_t block_;
    bool dcd_ = false;
    bool passall_ = false;
};

} // mobilinkd
