// mobilinkd::m17::KalmanFilter / SymbolKalmanFilter — the reference's 2-state Kalman filters (include/m17cxx/KalmanFilter.h:
// 18-108) WITHOUT blaze.  The reference builds them on the third-party blaze library (absent from its tree: empty submodule),
// with the gain and the innovation covariance kept as lazy blaze expressions, so the order in which the products are
// associated and rounded is blaze's, not the source text's.  Here the update is core::kalman2_update, the function the HIP
// kernels run, with that order as an explicit switch (`order`, see detail/core.h; default = blaze's documented restructuring).
// State is exposed the way the reference's members read: x[0], x[1], P(i, j).
#pragma once

#include "detail/core.h"

#include <cmath>
#include <cstddef>
#include <cstdint>

namespace mobilinkd { namespace m17 {

namespace detail
{
// views over core::Kalman2 with the element access of blaze::StaticVector / StaticMatrix
struct StateVector {
    core::Kalman2* k;
    float& operator[](size_t i) { return i == 0 ? k->x0 : k->x1; }
    float operator[](size_t i) const { return i == 0 ? k->x0 : k->x1; }
};
struct Covariance {
    core::Kalman2* k;
    float& operator()(size_t i, size_t j) { return i == 0 ? (j == 0 ? k->p00 : k->p01) : (j == 0 ? k->p10 : k->p11); }
};
// what update() returns: a copy of the state vector
struct StateCopy {
    float v[2];
    float operator[](size_t i) const { return v[i]; }
};
inline bool isnan(const StateCopy& s) { return std::isnan(s.v[0]) || std::isnan(s.v[1]); }
} // detail

template <typename FloatType, size_t SamplesPerSymbol>
struct KalmanFilter
{
    static_assert(sizeof(FloatType) == sizeof(float), "the demodulation path runs these filters in float");
    core::Kalman2 state_;
    uint32_t order = core::KALMAN_ORDER_DEFAULT;
    detail::StateVector x{&state_};
    detail::Covariance P{&state_};

    KalmanFilter() { reset(0.); }
    KalmanFilter(const KalmanFilter& o) : state_(o.state_), order(o.order) {}
    KalmanFilter& operator=(const KalmanFilter& o) { state_ = o.state_; order = o.order; return *this; }

    void reset(FloatType z) { core::kalman2_reset(state_, z); }

    // z: the new timing index measurement, dt: samples since the previous update; estimate wrapped into [0, SamplesPerSymbol)
    detail::StateCopy update(FloatType z, size_t dt)
    {
        core::kalman2_update(state_, z, (uint32_t)dt, (int)SamplesPerSymbol, order);
        return detail::StateCopy{{state_.x0, state_.x1}};
    }
};

template <typename FloatType>
struct SymbolKalmanFilter
{
    static_assert(sizeof(FloatType) == sizeof(float), "the demodulation path runs these filters in float");
    core::Kalman2 state_;
    uint32_t order = core::KALMAN_ORDER_DEFAULT;
    detail::StateVector x{&state_};
    detail::Covariance P{&state_};

    SymbolKalmanFilter() { reset(0.); }
    SymbolKalmanFilter(const SymbolKalmanFilter& o) : state_(o.state_), order(o.order) {}
    SymbolKalmanFilter& operator=(const SymbolKalmanFilter& o) { state_ = o.state_; order = o.order; return *this; }

    void reset(FloatType z) { core::kalman2_reset(state_, z); }

    detail::StateCopy update(FloatType z, size_t dt)
    {
        core::kalman2_update(state_, z, (uint32_t)dt, 0, order);
        return detail::StateCopy{{state_.x0, state_.x1}};
    }
};

}} // mobilinkd::m17
