// mobilinkd::BaseIirFilter / makeIirFilter — the reference's direct-form-II IIR (include/m17cxx/IirFilter.h:13-50).
// history_[0] = in - a[1]*history_[1] - ... (one subtraction per term), out = 0 + b[0]*history_[0] + b[1]*history_[1] + ...
// For N = 3 with the correlator's coefficients this is core::iir_advance / core::iir_output, the functions kernels K2/K5 use.
#pragma once

#include "Filter.h"

#include <array>
#include <cstddef>

namespace mobilinkd
{

template <typename FloatType, size_t N>
struct BaseIirFilter : FilterBase<FloatType>
{
    const std::array<FloatType, N>& numerator_;    // caller-owned (reference semantics)
    const std::array<FloatType, N> denominator_;
    std::array<FloatType, N> history_{};

    BaseIirFilter(const std::array<FloatType, N>& b, const std::array<FloatType, N>& a) : numerator_(b), denominator_(a) {}

    FloatType operator()(FloatType input) override
    {
        FloatType w = input;
        for (size_t i = 1; i < N; ++i) w -= denominator_[i] * history_[i - 1];   // history_[i - 1] is "i samples ago" before the shift
        for (size_t i = N - 1; i > 0; --i) history_[i] = history_[i - 1];
        history_[0] = w;
        FloatType out = FloatType(0);
        for (size_t i = 0; i < N; ++i) out += numerator_[i] * history_[i];
        return out;
    }
};

template <typename FloatType, size_t N>
BaseIirFilter<FloatType, N> makeIirFilter(const std::array<FloatType, N>& b, const std::array<FloatType, N>& a)
{
    return BaseIirFilter<FloatType, N>(b, a);
}

} // mobilinkd
