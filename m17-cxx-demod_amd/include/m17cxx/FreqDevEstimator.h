// mobilinkd::FreqDevEstimator — the reference's deviation / offset estimator (include/m17cxx/FreqDevEstimator.h:14-54): the
// outer symbol levels of each sync word go through one Kalman filter each (fixed 192-sample step); offset = mean of the two
// filtered levels, idev = 6 / their distance (core::freqdev_*, evaluated in double).  A NaN estimate, or reset(), restarts both
// filters from the raw levels of the same call.  Kernel K5 runs the same sequence per channel (csrc/m17_state.hpp).
#pragma once

#include "KalmanFilter.h"
#include "detail/core.h"

#include <cmath>
#include <cstddef>
#include <cstdint>

namespace mobilinkd {

template <typename FloatType>
class FreqDevEstimator
{
public:
    void reset() { restart_ = true; }

    // evaluation order of the Kalman updates (detail/core.h); default core::KALMAN_ORDER_DEFAULT
    void kalman_order(uint32_t order) { level_[0].order = level_[1].order = order; }

    void update(FloatType minValue, FloatType maxValue)
    {
        const FloatType raw[2] = {minValue, maxValue};
        bool poisoned = false;
        FloatType est[2];
        for (int k = 0; k != 2; ++k) {
            const auto state = level_[k].update(raw[k], SYNC_SPACING);
            est[k] = state[0];
            poisoned = poisoned || isnan(state);
        }
        if (poisoned || restart_) {
            restart_ = false;
            for (int k = 0; k != 2; ++k) { level_[k].reset(raw[k]); est[k] = raw[k]; }
            offset_ = (raw[0] + raw[1]) / 2;        // FloatType arithmetic on the restart path, as in the reference (:44)
        } else {
            offset_ = core::freqdev_offset(est[1], est[0]);
        }
        idev_ = core::freqdev_idev(est[1], est[0]);
    }

    FloatType idev() const { return idev_; }
    FloatType offset() const { return offset_; }
    FloatType deviation() const { return NOMINAL_HZ / idev_; }
    FloatType error() const { return FloatType(0); }

private:
    static constexpr size_t SYNC_SPACING = 192;          // the update's dt, in samples
    static constexpr FloatType NOMINAL_HZ = 2400.;

    m17::SymbolKalmanFilter<FloatType> level_[2];      // [0] lowest, [1] highest symbol level
    FloatType idev_ = 0.;
    FloatType offset_ = 0.;
    bool restart_ = true;
};

} // mobilinkd
