// mobilinkd::BaseIirFilter / makeIirFilter — the reference's direct-form-II IIR (include/m17cxx/IirFilter.h:13-50).
// Per sample: w = in - a[1] w1 - a[2] w2 - ... (one rounded subtraction per term, in that order), then
// out = ((0 + b[0] w) + b[1] w1) + ...; for N = 3 with the correlator's coefficients these are core::iir_advance /
// core::iir_output, the functions kernels K2 / K5 evaluate.  The numerator is referenced, not copied (reference semantics: the
// caller's array has to outlive the filter); the denominator is copied.
#pragma once

#include "Filter.h"

#include <array>
#include <cstddef>

namespace mobilinkd
{

template <typename FloatType, size_t N>
struct BaseIirFilter : FilterBase<FloatType>
{
    using coeff_t = std::array<FloatType, N>;

    BaseIirFilter(const coeff_t& b, const coeff_t& a) : b_(b), a_(a) { w_.fill(FloatType(0)); }

    FloatType operator()(FloatType input) override
    {
        // w_[j] is the internal state j + 1 samples ago until the shift below
        FloatType now = input;
        for (size_t j = 0; j + 1 < N; ++j) now -= a_[j + 1] * w_[j];
        FloatType out = FloatType(0);
        out += b_[0] * now;
        for (size_t j = 0; j + 1 < N; ++j) out += b_[j + 1] * w_[j];
        for (size_t j = N - 1; j-- > 1;) w_[j] = w_[j - 1];
        if (N > 1) w_[0] = now;
        return out;
    }

private:
    const coeff_t& b_;
    const coeff_t a_;
    std::array<FloatType, (N > 1 ? N - 1 : 1)> w_;
};

template <typename FloatType, size_t N>
BaseIirFilter<FloatType, N> makeIirFilter(const std::array<FloatType, N>& b, const std::array<FloatType, N>& a)
{
    return {b, a};
}

} // mobilinkd
