// mobilinkd::DataCarrierDetect — the reference's carrier detect (include/m17cxx/DataCarrierDetect.h:28-74): a two-bin sliding
// DFT (length SampleRate / Accuracy) of the RAW baseband; per sample the bin powers are accumulated, update() turns their
// ratio into an exponentially smoothed level (in double, :65) and applies the hysteresis.  An update window of exact zeros
// gives 0/0 = NaN and poisons the level for good — reproduced, not repaired.
// Batched form: sums(batched::Device&, ...) = m17hip_dcd, kernel K3's table of window sums for every 192-sample tick.
#pragma once

#include "SlidingDFT.h"
#include "detail/batched.h"
#include "detail/core.h"

#include <array>
#include <complex>
#include <cstddef>
#include <cstdint>

namespace mobilinkd
{

template <typename FloatType, size_t SampleRate, size_t Accuracy = 1000>
struct DataCarrierDetect
{
    using ComplexType = std::complex<FloatType>;
    using NDFT = NSlidingDFT<FloatType, SampleRate, SampleRate / Accuracy, 2>;

    NDFT dft_;
    FloatType ltrigger_;
    FloatType htrigger_;
    FloatType level_1 = 0.0;
    FloatType level_2 = 0.0;
    FloatType level_ = 0.0;
    bool triggered_ = false;

    DataCarrierDetect(size_t freq1, size_t freq2, FloatType ltrigger = 2.0, FloatType htrigger = 5.0)
    : dft_({freq1, freq2}), ltrigger_(ltrigger), htrigger_(htrigger)
    {}

    void operator()(FloatType sample)
    {
        const auto bins = dft_(sample);
        level_1 += std::norm(bins[0]);
        level_2 += std::norm(bins[1]);
    }

    void update()
    {
        const FloatType ratio = level_1 / level_2;                       // 0/0 -> NaN (Q1)
        level_ = FloatType(double(level_) * 0.8 + 0.2 * double(ratio));   // the EMA runs in double (core::dcd_level)
        level_1 = level_2 = 0.0;
        const FloatType threshold = triggered_ ? ltrigger_ : htrigger_;   // hysteresis
        triggered_ = level_ > threshold;
    }

    void unlock() { triggered_ = false; }
    FloatType level() const { return level_; }
    bool dcd() const { return triggered_; }

    // Batched form (GPU), for the demodulator's detector (48 kSPS, 120-sample DFT, 2400 / 3600 Hz): for every 192-sample tick k
    // of each channel, sums[channel][k][bin][j] = sum of norm(bin) over the samples since the start of tick a, where a % 5 == j
    // is one of the last five tick starts (j < 5), or since the stream start (j == 5) — every window a 384- or 960-sample
    // update cadence can ask for.  in: [channels][samples] int16 (row pitch = samples).
    static int sums(batched::Device& dev, const int16_t* in, uint32_t channels, uint32_t samples, float* sums_out, bool invert = false)
    {
        static_assert(SampleRate == 48000 && SampleRate / Accuracy == 120, "the batched detector is the demodulator's (48 kSPS, N = 120)");
        int r = m17hip_upload_i16(dev.ctx(), in, channels, samples, samples);
        if (r != M17HIP_OK) return r;
        uint32_t ticks = 0;
        return m17hip_dcd(dev.ctx(), channels, samples, invert ? M17HIP_FLAG_INVERT : 0u, sums_out, &ticks);
    }
};

} // mobilinkd
