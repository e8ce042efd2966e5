// mobilinkd::FreqDevEstimator — the reference's deviation / offset estimator (include/m17cxx/FreqDevEstimator.h:14-54): the
// outer symbol levels of each sync word, smoothed by two Kalman filters with a fixed 192-sample step; offset = their mean,
// idev = 6 / their distance (both in double).  A NaN estimate, or reset(), restarts both filters from the raw levels.
#pragma once

#include "KalmanFilter.h"
#include "detail/core.h"

#include <cmath>
#include <cstddef>

namespace mobilinkd {

template <typename FloatType>
class FreqDevEstimator
{
    static constexpr FloatType DEVIATION = 2400.;

    m17::SymbolKalmanFilter<FloatType> minFilter_;
    m17::SymbolKalmanFilter<FloatType> maxFilter_;
    FloatType idev_ = 0.;
    FloatType offset_ = 0.;
    bool reset_ = true;

public:
    void reset() { reset_ = true; }

    // evaluation order of the Kalman updates (detail/core.h); default core::KALMAN_ORDER_DEFAULT
    void kalman_order(uint32_t order) { minFilter_.order = maxFilter_.order = order; }

    void update(FloatType minValue, FloatType maxValue)
    {
        const auto lo = minFilter_.update(minValue, 192);
        const auto hi = maxFilter_.update(maxValue, 192);
        offset_ = core::freqdev_offset(hi[0], lo[0]);
        idev_ = core::freqdev_idev(hi[0], lo[0]);
        if (isnan(lo) || isnan(hi)) reset_ = true;
        if (reset_) {
            reset_ = false;
            minFilter_.reset(minValue);
            maxFilter_.reset(maxValue);
            offset_ = (minValue + maxValue) / 2;
            idev_ = core::freqdev_idev(maxValue, minValue);
        }
    }

    FloatType idev() const { return idev_; }
    FloatType offset() const { return offset_; }
    FloatType deviation() const { return DEVIATION / idev_; }
    FloatType error() const { return 0.; }
};

} // mobilinkd
