// mobilinkd::BaseFirFilter / makeFirFilter — the reference's FIR operator (include/m17cxx/FirFilter.h:13-56) on this
// framework's terms: the scalar, one-sample form (what apps/m17-demod.cpp-style hosts call per sample) and the batched form
// that runs the RRC matched filter of the demodulation path on the MI355X through the C ABI (m17hip_fir_rrc150, kernel K1).
// Arithmetic contract: y = sum taps[i] * x[n - i], i = 0 .. N-1 in that order, every multiply and add rounded
// separately — compile with -ffp-contract=off.
#pragma once

#include "Filter.h"
#include "detail/batched.h"
#include "detail/core.h"

#include <array>
#include <cstddef>
#include <cstdint>

namespace mobilinkd
{

template <typename FloatType, size_t N>
struct BaseFirFilter : FilterBase<FloatType>
{
    using array_t = std::array<FloatType, N>;

    const array_t& taps_;   // caller-owned, must outlive the filter (as in the reference)
    array_t history_;       // circular: history_[pos_] receives the next sample
    size_t pos_ = 0;

    BaseFirFilter(const array_t& taps) : taps_(taps) { history_.fill(FloatType(0)); }

    FloatType operator()(FloatType input) override
    {
        history_[pos_] = input;
        const size_t newest = pos_;
        pos_ = (pos_ + 1 == N) ? 0 : pos_ + 1;
        // newest sample meets taps_[0]; walk back to the start of the ring, then from its end
        FloatType acc = FloatType(0);
        size_t t = 0;
        for (size_t h = newest + 1; h-- > 0; ++t) acc += history_[h] * taps_[t];
        for (size_t h = N; t < N; ++t) acc += history_[--h] * taps_[t];
        return acc;
    }

    void reset()
    {
        history_.fill(FloatType(0));
        pos_ = 0;
    }

    // Batched form on the GPU: `channels` independent 48 kSPS int16 streams of `samples` samples each (row pitch = samples),
    // scaled as apps/m17-demod.cpp:486-489 does and filtered from zero history: out[channels][samples].  Only for the
    // demodulator's own filter (150 float taps equal to detail::Taps<float>::rrc_taps): that is what kernel K1 computes.
    int operator()(batched::Device& dev, const int16_t* in, uint32_t channels, uint32_t samples, FloatType* out, bool invert = false) const
    {
        static_assert(N == 150 && sizeof(FloatType) == sizeof(float), "the batched filter is the 150-tap float RRC matched filter");
        for (size_t i = 0; i < N; ++i)
            if (taps_[i] != core::rrc_tap((int)i)) return M17HIP_EINVAL;
        int r = m17hip_upload_i16(dev.ctx(), in, channels, samples, samples);
        if (r != M17HIP_OK) return r;
        return m17hip_fir_rrc150(dev.ctx(), channels, samples, invert ? M17HIP_FLAG_INVERT : 0u, out);
    }
};

template <typename FloatType, size_t N>
BaseFirFilter<FloatType, N> makeFirFilter(const std::array<FloatType, N>& taps)
{
    return BaseFirFilter<FloatType, N>(taps);
}

} // mobilinkd
