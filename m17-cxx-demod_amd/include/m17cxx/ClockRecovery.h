// mobilinkd::ClockRecovery — the reference's symbol clock tracker (include/m17cxx/ClockRecovery.h:16-111): a Kalman filter
// on (timing index, clock rate) fed by the sync-word timing index, and between sync words a prediction from the last
// estimate.  Arithmetic: core::kalman2_update / core::clock_index_of / core::clock_predict, shared with kernel K5.
#pragma once

#include "KalmanFilter.h"
#include "detail/core.h"

#include <cstddef>
#include <cstdint>

namespace mobilinkd
{

template <typename FloatType, size_t SamplesPerSymbol>
struct ClockRecovery
{
    static_assert(SamplesPerSymbol == 10, "M17: 10 samples per symbol (core::wrap10)");
    m17::KalmanFilter<FloatType, SamplesPerSymbol> kf_;
    size_t count_ = 0;
    int8_t sample_index_ = 0;
    FloatType clock_estimate_ = 0.;
    FloatType sample_estimate_ = 0.;

    // start over from the timing index of the first sync word
    void reset(FloatType index)
    {
        kf_.reset(index);
        count_ = 0;
        sample_index_ = (int8_t)index;
        clock_estimate_ = 0.;
    }

    // one sample has gone by
    void operator()(FloatType) { ++count_; }

    // a sync word gave a fresh timing index: filter it
    bool update(uint8_t index)
    {
        const auto est = kf_.update(FloatType(index), count_);
        sample_estimate_ = est[0];
        clock_estimate_ = est[1];
        sample_index_ = (int8_t)core::clock_index_of(sample_estimate_);
        count_ = 0;
        return true;
    }

    // no sync word: predict the timing index from the last estimate and the samples gone by (estimates stay as they are)
    bool update()
    {
        sample_index_ = (int8_t)core::clock_predict(sample_estimate_, clock_estimate_, (uint32_t)count_);
        return true;
    }

    FloatType clock_estimate() const { return clock_estimate_; }
    uint8_t sample_index() const { return sample_index_; }
};

} // mobilinkd
