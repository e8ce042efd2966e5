// Drop-in for the reference's `mobilinkd::M17Demodulator<FloatType>` (reference include/m17cxx/M17Demodulator.h:123-217)
// as apps/m17-demod.cpp uses it: construct with the frame callback (:455), install a diagnostic callback (:478), push
// one scaled sample per call (:484-490).  Samples are collected into blocks; every full block goes through the GPU chain
// (include/m17hip.h) with the demodulator state carried from block to block, and the callbacks fire in stream order
// when the block returns.  Differences a caller can observe: callbacks are delivered with up to one block of latency,
// and the diagnostic callback fires once per block with the arguments of the LAST of the reference's per-384/960-sample
// calls.  `flush()` pushes a partial block (end of input).
#pragma once

#include "BatchedDemodulator.h"
#include "M17FrameDecoder.h"

#include <cmath>
#include <cstring>
#include <functional>

namespace mobilinkd
{

template <typename FloatType>
struct M17Demodulator
{
    static constexpr uint16_t SAMPLE_RATE = 48000;
    static constexpr uint16_t SYMBOL_RATE = 4800;
    static constexpr uint16_t SAMPLES_PER_SYMBOL = SAMPLE_RATE / SYMBOL_RATE;

    using callback_t = M17FrameDecoder::callback_t;
    using diagnostic_callback_t = std::function<void(bool, FloatType, FloatType, FloatType, bool, FloatType, int, int, int, int)>;

    explicit M17Demodulator(callback_t callback, uint32_t block_samples = 9600, int device = 0)
    : gpu_(1, block_samples, device), callback_(std::move(callback)), block_(block_samples)
    {
        buffer_.reserve(block_);
        gpu_.reset();
    }

    // The reference takes sample / 41067.0 (apps/m17-demod.cpp:489); the int16 is recovered exactly.
    void operator()(const FloatType input)
    {
        buffer_.push_back((int16_t)std::lrint((double)input * 41067.0));
        if (buffer_.size() == block_) run_block();
    }
    void flush() { if (!buffer_.empty()) run_block(); }

    bool locked() const { return dcd_; }
    void passall(bool) {}
    void diagnostics(diagnostic_callback_t callback) { diagnostic_callback_ = std::move(callback); }

private:
    void run_block()
    {
        gpu_.upload(buffer_.data(), 1, (uint32_t)buffer_.size(), buffer_.size());
        gpu_.run();
        for (const auto& r : gpu_.frames()) {
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
        const auto d = gpu_.diagnostics()[0];
        dcd_ = d.dcd != 0;
        if (diagnostic_callback_ && d.n_diag != last_n_diag_) {
            last_n_diag_ = d.n_diag;
            diagnostic_callback_(d.dcd != 0, (FloatType)d.evm, (FloatType)d.deviation, (FloatType)d.offset, d.locked != 0,
                                 (FloatType)d.clock, d.sample_index, d.sync_index, d.clock_index, d.viterbi_cost);
        }
        buffer_.clear();
    }

    BatchedDemodulator gpu_;
    callback_t callback_;
    diagnostic_callback_t diagnostic_callback_;
    std::vector<int16_t> buffer_;
    uint32_t block_;
    uint32_t last_n_diag_ = 0;
    bool dcd_ = false;
};

} // mobilinkd
