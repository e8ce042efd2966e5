// mobilinkd::Correlator / mobilinkd::SyncWord — the reference's sync-word correlator (include/m17cxx/Correlator.h:19-208).
// Scalar forms: one sample per call, arithmetic from detail/core.h (the same functions kernels K2 / K5 are built from).
// Batched form: Correlator::operator()(batched::Device&, ...) = m17hip_correlator over the matched-filter output that the
// batched BaseFirFilter call left on the device.
// Semantics kept on purpose: `limit_` is the IIR-smoothed |sample| (:43-45); correlate() walks the ring oldest symbol first
// (:51-64); outer_symbol_levels() has the reference's `avg = max + min / 2.` (:97); SyncWord::find_peak compares float
// magnitudes (the float overload of abs — SURVEY Q5) and an exact-zero correlation is "not triggered" (:179-183).
#pragma once

#include "IirFilter.h"
#include "detail/batched.h"
#include "detail/core.h"

#include <algorithm>
#include <array>
#include <cmath>
#include <cstddef>
#include <cstdint>
#include <limits>
#include <tuple>
#include <type_traits>

namespace mobilinkd {

template <typename FloatType>
struct Correlator
{
    static constexpr size_t SYMBOLS = 8;
    static constexpr size_t SAMPLES_PER_SYMBOL = 10;

    using value_type = FloatType;
    using buffer_t = std::array<FloatType, SYMBOLS * SAMPLES_PER_SYMBOL>;
    using sync_t = std::array<int8_t, SYMBOLS>;
    using sample_filter_t = BaseIirFilter<FloatType, 3>;

    buffer_t buffer_{};   // value-initialised: the reference reads it before it is full (zero-filled object)

    FloatType limit_ = 0.;
    size_t symbol_pos_ = 0;
    size_t buffer_pos_ = 0;
    size_t prev_buffer_pos_ = 0;
    int code = -1;

    // IIR with Nyquist of 1/240 (Correlator.h:38-39)
    static constexpr std::array<FloatType, 3> b = {FloatType(core::LimitIir::b0), FloatType(core::LimitIir::b1), FloatType(core::LimitIir::b2)};
    static constexpr std::array<FloatType, 3> a = {FloatType(1.0), FloatType(core::LimitIir::a1), FloatType(core::LimitIir::a2)};
    sample_filter_t sample_filter{b, a};

    void sample(FloatType value)
    {
        limit_ = sample_filter(std::abs(value));
        buffer_[buffer_pos_] = value;
        prev_buffer_pos_ = buffer_pos_;
        buffer_pos_ = (buffer_pos_ + 1 == buffer_.size()) ? 0 : buffer_pos_ + 1;
    }

    FloatType correlate(sync_t sync)
    {
        // the eight ring samples one symbol apart that end at the newest sample, oldest first
        FloatType acc = 0.;
        size_t pos = prev_buffer_pos_;
        for (size_t i = 0; i < SYMBOLS; ++i) {
            pos += SAMPLES_PER_SYMBOL;
            if (pos >= buffer_.size()) pos -= buffer_.size();
            acc += sync[i] * buffer_[pos];
        }
        return acc;
    }

    FloatType limit() const { return limit_; }
    size_t index() const { return prev_buffer_pos_ % SAMPLES_PER_SYMBOL; }

    // Mean of the samples above / below `avg` at one sampling phase (Correlator.h:81-114): core::outer_symbol_levels.
    std::tuple<FloatType, FloatType> outer_symbol_levels(size_t sample_index)
    {
        float mn, mx;
        core::outer_symbol_levels([this](uint32_t i) { return (float)buffer_[i]; }, (uint32_t)sample_index, mn, mx);
        return std::make_tuple(FloatType(mn), FloatType(mx));
    }

    template <typename F>
    void apply(F func, uint8_t index)
    {
        for (size_t i = index; i < buffer_.size(); i += SAMPLES_PER_SYMBOL) func(buffer_[i]);
    }

    // Batched form (GPU): limit[channels][samples] = limit() after every sample, corr[4][channels][samples] = correlate() against
    // the preamble, LSF, packet and EOT words after every sample, for the matched-filter output the preceding batched
    // BaseFirFilter call left on the device.
    static int run(batched::Device& dev, uint32_t channels, uint32_t samples, float* limit, float* corr)
    {
        return m17hip_correlator(dev.ctx(), channels, samples, limit, corr);
    }
};

template <typename Correlator>
struct SyncWord
{
    static constexpr size_t SYMBOLS = Correlator::SYMBOLS;
    static constexpr size_t SAMPLES_PER_SYMBOL = Correlator::SAMPLES_PER_SYMBOL;
    using value_type = typename Correlator::value_type;

    using buffer_t = std::array<int8_t, SYMBOLS>;
    using sample_buffer_t = std::array<value_type, SAMPLES_PER_SYMBOL>;

    buffer_t sync_word_;
    sample_buffer_t samples_{};
    size_t pos_ = 0;
    size_t timing_index_ = 0;
    bool triggered_ = false;
    int8_t updated_ = 0;
    value_type magnitude_1_ = 1.;
    value_type magnitude_2_ = -1.;

    SyncWord(buffer_t&& sync_word, value_type magnitude_1, value_type magnitude_2 = std::numeric_limits<value_type>::lowest())
    : sync_word_(std::move(sync_word)), magnitude_1_(magnitude_1), magnitude_2_(magnitude_2)
    {}

    // the correlation if it lies beyond either threshold (a multiple of the correlator's limit), else 0
    value_type triggered(Correlator& correlator)
    {
        const value_type upper = correlator.limit() * magnitude_1_;
        const value_type lower = correlator.limit() * magnitude_2_;
        const value_type v = correlator.correlate(sync_word_);
        return (v > upper || v < lower) ? v : value_type(0.0);
    }

    bool is_triggered() const { return triggered_; }

    // the trigger fell: the phase with the largest |correlation| among the stored ones becomes the timing index
    void find_peak(value_type value)
    {
        triggered_ = false;
        timing_index_ = 0;
        value_type best = value;
        for (size_t k = 0; k < samples_.size(); ++k) {
            if (std::fabs(samples_[k]) > std::fabs(best)) { best = samples_[k]; timing_index_ = k; }
        }
        updated_ = best > 0 ? 1 : -1;
    }

    size_t operator()(Correlator& correlator)
    {
        const value_type v = triggered(correlator);
        if (v != 0) {
            if (!triggered_) { samples_.fill(0); triggered_ = true; }
            samples_[correlator.index()] = v;
        } else if (triggered_) {
            find_peak(v);
        }
        return timing_index_;
    }

    int8_t updated()
    {
        const int8_t r = updated_;
        updated_ = 0;
        return r;
    }
};

} // mobilinkd
