// mobilinkd::StandardDeviation / RunningStandardDeviation (reference include/m17cxx/StandardDeviation.h:9-84): Welford's
// running variance, and the exponentially weighted mean square the EVM meter uses (S starts at 1, reset() makes it 0;
// S -= S*alpha; S += x*x*alpha — core::evm_capture for float, N = 184).
#pragma once

#include <cmath>
#include <cstddef>
#include <cstdint>

namespace mobilinkd {

template <typename FloatType>
struct StandardDeviation
{
    FloatType mean{0.0};
    FloatType S{0.0};
    size_t samples{0};

    void reset() { mean = 0.0; S = 0.0; samples = 0; }

    void capture(float sample)
    {
        const FloatType before = mean;
        ++samples;
        mean = before + (sample - before) / samples;
        S = S + (sample - mean) * (sample - before);
    }

    FloatType variance() const { return samples ? S / samples : FloatType(-1.0); }
    FloatType stdev() const { return samples ? std::sqrt(variance()) : FloatType(-1.0); }
    FloatType SNR() const { return 10.0 * std::log10(mean / stdev()); }
};

template <typename FloatType, size_t N>
struct RunningStandardDeviation
{
    FloatType S{1.0};
    FloatType alpha{1.0 / N};

    void reset() { S = 0.0; }

    void capture(float sample)
    {
        S -= S * alpha;
        S += (sample * sample) * alpha;
    }

    FloatType variance() const { return S; }
    FloatType stdev() const { return std::sqrt(S); }
};

} // mobilinkd
