// mobilinkd::SlidingDFT / NSlidingDFT — the reference's sliding DFT (include/m17cxx/SlidingDFT.h:20-133).
// X_k <- (X_k + (x[n] - x[n-N])) * c_k with the complex multiply spelled out as libstdc++ evaluates it for floats without
// -ffast-math: (ac - bd, ad + bc), four products, one subtraction, one addition, nothing fused (core::sdft_step — the
// recurrence kernel K3 runs for 32 channels per wave).  Coefficients c_k = exp(-j * 2*pi * f_k / SampleRate) from std::exp on
// the host, exactly as the reference builds them (the C ABI builds its own the same way: csrc/m17hip.hip build_coef).
#pragma once

#include "detail/core.h"

#include <array>
#include <cmath>
#include <complex>
#include <cstddef>

namespace mobilinkd
{

namespace detail
{
template <typename FloatType>
inline std::complex<FloatType> sdft_coefficient(size_t frequency, size_t sample_rate)
{
    const std::complex<FloatType> j{0, 1};
    const FloatType pi2 = M_PI * 2.0;
    const FloatType kth = FloatType(frequency) / FloatType(sample_rate);
    return std::exp(-j * pi2 * kth);
}
template <typename FloatType>
inline std::complex<FloatType> sdft_advance(std::complex<FloatType> x, FloatType delta, std::complex<FloatType> c)
{
    if constexpr (std::is_same_v<FloatType, float>) {
        float re = x.real(), im = x.imag();
        core::sdft_step(re, im, delta, c.real(), c.imag());
        return {re, im};
    } else {
        const FloatType a = x.real() + delta, b = x.imag();
        return {a * c.real() - b * c.imag(), a * c.imag() + b * c.real()};
    }
}
} // detail

/**
 * Single-bin sliding DFT with a leaky integrator (reference SlidingDFT.h:20-62; unused by the demodulator, kept for API parity).
 */
template <typename FloatType, size_t SampleRate, size_t Frequency, size_t Accuracy = 1000>
class SlidingDFT
{
    using ComplexType = std::complex<FloatType>;
    static constexpr size_t N = SampleRate / Accuracy;

    const ComplexType coeff_ = detail::sdft_coefficient<FloatType>(Frequency, SampleRate);
    std::array<FloatType, N> samples_{};
    ComplexType result_{0, 0};
    size_t index_ = 0;

public:
    SlidingDFT() = default;

    ComplexType operator()(FloatType sample)
    {
        const FloatType delta = sample - samples_[index_];
        samples_[index_] = sample;
        index_ = (index_ + 1 == N) ? 0 : index_ + 1;
        const ComplexType r = detail::sdft_advance(result_, delta, coeff_);
        result_ = r * FloatType(0.999999999999999);
        return r;
    }
};

/**
 * K-bin sliding DFT of length N (reference SlidingDFT.h:64-133).  The result is meaningful once N samples are in.
 */
template <typename FloatType, size_t SampleRate, size_t N, size_t K>
class NSlidingDFT
{
    using ComplexType = std::complex<FloatType>;

    std::array<ComplexType, K> coeff_;
    std::array<FloatType, N> samples_{};
    std::array<ComplexType, K> result_{};
    size_t index_ = 0;

public:
    using result_type = std::array<ComplexType, K>;

    NSlidingDFT(const std::array<size_t, K>& frequencies)
    {
        for (size_t k = 0; k < K; ++k) coeff_[k] = detail::sdft_coefficient<FloatType>(frequencies[k], SampleRate);
    }

    result_type operator()(FloatType sample)
    {
        const FloatType delta = sample - samples_[index_];
        samples_[index_] = sample;
        index_ = (index_ + 1 == N) ? 0 : index_ + 1;
        for (size_t k = 0; k < K; ++k) result_[k] = detail::sdft_advance(result_[k], delta, coeff_[k]);
        return result_;
    }
};

} // mobilinkd
