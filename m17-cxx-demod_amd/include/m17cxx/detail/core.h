// Arithmetic cores of the M17 demodulation path, written ONCE and compiled for both sides of the boundary: the HIP kernels
// (csrc/, under hipcc these are __host__ __device__) and the scalar operator classes of this directory (any C++17/20 host
// compiler).  Every rounding step of the path that decides a decoded bit is defined here and nowhere else, so the scalar
// classes and the kernels cannot drift apart.  Reference file:line is cited per function
// (/root/reference/include/m17cxx unless stated).  No dependency on the oracle, on HIP or on the C ABI.
//
// Build contract: -ffp-contract=off (every fp32 multiply and add rounds separately, as in the reference's default x86-64
// build; the few fused operations below are written as explicit fma calls and are exact by construction).
#pragma once

#include <stdint.h>

#if defined(__HIPCC__)
#define M17_HD __host__ __device__ __forceinline__
#else
#define M17_HD inline
#endif

namespace mobilinkd
{
namespace core
{

// ---- a1: sample scaling (apps/m17-demod.cpp:486-489) ---------------------------------------------------------------
// x = float(double(s16) / 41067.0), optionally s16 *= -1 first (in int16: -(-32768) wraps to -32768).
// (float)s / 41067.0f is bit-identical for all 65536 inputs (41067 is odd and < 2^16: s / 41067 is never within 2^-40 of a
// float midpoint, so the double rounding is harmless), and so is one Newton step on q = s * RN(1/41067):
// r = fma(-q, 41067, s) is the exact remainder, fma(r, 1/41067, q) the correctly rounded quotient.
M17_HD float scale_i16(int s, bool invert)
{
    if (invert) s = (int)(int16_t)(-s);
    const float rcp = 1.0f / 41067.0f;
    const float fs = (float)s;
    const float q = fs * rcp;
    const float r = __builtin_fmaf(-q, 41067.0f, fs);
    return __builtin_fmaf(r, rcp, q);
}

// ---- a2: RRC matched filter taps (M17Demodulator.h:79-118): alpha = 0.5, 10 samples per symbol, 149 symmetric taps and
// a trailing 0.0; the double literals narrowed to float.  FIR order: FirFilter.h:36-40 (newest sample first, i = 0..149).
constexpr int RRC_TAPS = 150;
constexpr double RRC_HALF[75] = {
#include "rrc_half_taps.inc"
};
M17_HD constexpr float rrc_tap(int i) { return i >= 149 ? 0.0f : (float)(i <= 74 ? RRC_HALF[i] : RRC_HALF[148 - i]); }
// y = sum_{i=0}^{n-1} taps[i] * x[newest - i], accumulated in that order, multiply and add rounded separately.
template <typename GetSample>
M17_HD float fir_dot(const float* taps, int n, GetSample newest_minus)
{
    float acc = 0.0f;
    for (int i = 0; i < n; ++i) {
        const float p = newest_minus(i) * taps[i];
        acc = acc + p;
    }
    return acc;
}

// ---- a3: the correlator's limit filter: BaseIirFilter<float,3> (IirFilter.h:26-42) with Correlator.h:38-39 ----------
// h0 = in - a1*h1 - a2*h2 (two separate subtractions), out = 0 + b0*h0 + b1*h1 + b2*h2 (accumulated from 0 in that order).
struct LimitIir {
    static constexpr float b0 = 4.24433681e-05f, b1 = 8.48867363e-05f, b2 = 4.24433681e-05f;
    static constexpr float a1 = -1.98148851f, a2 = 0.98165828f;
};
M17_HD float iir_advance(float in_abs, float h1, float h2)
{
    float h0 = in_abs;
    h0 = h0 - LimitIir::a1 * h1;
    h0 = h0 - LimitIir::a2 * h2;
    return h0;
}
M17_HD float iir_output(float h0, float h1, float h2)
{
    float r = 0.0f;
    r = r + LimitIir::b0 * h0;
    r = r + LimitIir::b1 * h1;
    r = r + LimitIir::b2 * h2;
    return r;
}

// ---- a4: Correlator::correlate (Correlator.h:51-64) against the M17 sync words (M17Demodulator.h:154-157) -------------
// The words as sign masks (bit i set = symbol i is -3): (float)(-3) * x == -(3.0f * x) exactly and r + (-p) is what r - p
// computes, so a correlation is eight multiplies by 3.0f and eight adds, oldest symbol first.
constexpr uint32_t SYNC_NEG[4] = {0xAAu, 0xB0u, 0xF2u, 0x40u};   // preamble, LSF(/stream), packet(/BERT), EOT
constexpr int8_t SYNC_SYMBOLS[4][8] = {{+3, -3, +3, -3, +3, -3, +3, -3}, {+3, +3, +3, +3, -3, -3, +3, -3},
                                       {+3, -3, +3, +3, -3, -3, -3, -3}, {+3, +3, +3, +3, +3, +3, -3, +3}};
M17_HD float correlate_mask(uint32_t neg, const float (&r)[8])
{
    float v = 0.f;
#if defined(__HIPCC__)
#pragma unroll
#endif
    for (int i = 0; i < 8; ++i) {
        const float p = 3.0f * r[i];
        const float q = __builtin_bit_cast(float, __builtin_bit_cast(uint32_t, p) ^ (((neg >> i) & 1u) << 31));
        v = v + q;
    }
    return v;
}
// general form: int8 coefficient * float sample, accumulated from 0 (what the reference's loop does for any sync_t)
M17_HD float correlate_symbols(const int8_t* sync, const float (&r)[8])
{
    float v = 0.f;
    for (int i = 0; i < 8; ++i) v = v + (float)sync[i] * r[i];
    return v;
}

// ---- a6: Correlator::outer_symbol_levels (Correlator.h:81-114) -------------------------------------------------------
// get(i) = buffer_[i], i = sample_index, sample_index + 10, ... < 80.  `avg = max + min / 2.` (sic, in double).
template <typename Get>
M17_HD void outer_symbol_levels(Get get, uint32_t sample_index, float& mn, float& mx)
{
    float min_sum = 0.f, max_sum = 0.f;
    uint32_t min_count = 0, max_count = 0;
    float lo = get(sample_index), hi = lo;
    for (uint32_t i = sample_index; i < 80u; i += 10u) {
        const float v = get(i);
        lo = (v < lo) ? v : lo;   // std::min(lo, v)
        hi = (hi < v) ? v : hi;   // std::max(hi, v)
    }
    const float avg = (float)((double)hi + (double)lo / 2.);
    for (uint32_t i = sample_index; i < 80u; i += 10u) {
        const float v = get(i);
        const bool high = v > avg, low = v < avg;
        max_sum = max_sum + v * (high ? 1.f : 0.f);
        min_sum = min_sum + v * (low ? 1.f : 0.f);
        max_count += high; min_count += low;
    }
    mn = min_count > 0 ? min_sum / (float)min_count : lo;
    mx = max_count > 0 ? max_sum / (float)max_count : hi;
}

// ---- a7: one bin of NSlidingDFT::operator() (SlidingDFT.h:118-132): X = (X + delta) * c, libstdc++ complex multiply ------
// (a + bi)(c + di) = (ac - bd) + (ad + bc)i in fp32, no FMA; delta is real.
M17_HD void sdft_step(float& re, float& im, float delta, float cr, float ci)
{
    const float a = re + delta, b = im;
    const float ac = a * cr, bd = b * ci, ad = a * ci, bc = b * cr;
    re = ac - bd;
    im = ad + bc;
}
M17_HD float complex_norm(float re, float im) { return re * re + im * im; }   // std::norm
// ---- a8: DataCarrierDetect::update (DataCarrierDetect.h:63-69), level EMA in double ---------------------------------------
M17_HD float dcd_level(float level, float l1, float l2) { return (float)((double)level * 0.8 + 0.2 * (double)(l1 / l2)); }

// ---- a9/a10: the 2-state Kalman filter of KalmanFilter.h:18-108 (F = [[1,dt],[0,1]], H = [1 0], R = 0.5, Q :26) ---------
// blaze (the reference's linear-algebra dependency) is absent from the reference tree; `S` and `K` are lazy blaze
// expressions there (`auto`), so the association / rounding of `x += K*y` and `P = P - K*H*P` follows blaze's restructuring
// operators.  The order is a switch, the same in the oracle and in the C ABI (m17hip_set_kalman_order):
//   bit 0: x += double(fl32(P(:,0)*y)) * invS      [(A*s)*v -> (A*v)*s]   else  x += (double(P(:,0))*invS) * double(y)
//   bit 1: P -= double(fl32(P(i,0)*P(0,j))) * invS [(A*s)*B -> (A*B)*s]   else  P -= ((double(P(i,0))*invS) * double(P(0,j)))
//   bit 2: F*(P*F^T) instead of (F*P)*F^T
constexpr uint32_t KALMAN_ORDER_DEFAULT = 3;
struct Kalman2 {
    float x0, x1, p00, p01, p10, p11;
};
M17_HD void kalman2_reset(Kalman2& k, float z)
{
    k.x0 = z; k.x1 = 0.f;
    k.p00 = 4.f; k.p01 = 0.f; k.p10 = 0.f; k.p11 = (float)0.00000025;
}
// wrap != 0: KalmanFilter<float,SPS> (index filter, modulo SPS, :41-65); 0: SymbolKalmanFilter (:91-107).
// The order is a template parameter so that each variant is straight-line code (the kernels dispatch once per update).
//
// The update is written as its two independent halves.  The COVARIANCE half (P = F P F^T + Q; S = H P H^T + R; P = P - K H P) never
// reads x or z: it leaves the covariance after the predict step and 1 / S — all the state half needs — in a Kalman2Gain.  The STATE
// half (x = F x; y = z - H x; x += K y) reads that gain and nothing of P.  kalman2_update_as = the one after the other, operation for
// operation what the single function of rounds 1-3 did.
struct Kalman2Gain {
    float p00, p01, p10, p11;   // P after the predict step
    double invS;
};
template <uint32_t order>
M17_HD void kalman2_cov_step(Kalman2& k, uint32_t dt_u, Kalman2Gain& g)
{
    const float F00 = 1.f, F01 = (float)dt_u, F10 = 0.f, F11 = 1.f;
    const float Q00 = (float)6.25e-13, Q01 = (float)1.25e-12, Q10 = (float)1.25e-12, Q11 = (float)2.50e-12;
    float B00, B01, B10, B11;
    if (!(order & 4u)) {
        const float A00 = F00 * k.p00 + F01 * k.p10, A01 = F00 * k.p01 + F01 * k.p11;
        const float A10 = F10 * k.p00 + F11 * k.p10, A11 = F10 * k.p01 + F11 * k.p11;
        B00 = A00 * F00 + A01 * F01; B01 = A00 * F10 + A01 * F11;
        B10 = A10 * F00 + A11 * F01; B11 = A10 * F10 + A11 * F11;
    } else {
        const float A00 = k.p00 * F00 + k.p01 * F01, A01 = k.p00 * F10 + k.p01 * F11;
        const float A10 = k.p10 * F00 + k.p11 * F01, A11 = k.p10 * F10 + k.p11 * F11;
        B00 = F00 * A00 + F01 * A10; B01 = F00 * A01 + F01 * A11;
        B10 = F10 * A00 + F11 * A10; B11 = F10 * A01 + F11 * A11;
    }
    k.p00 = B00 + Q00; k.p01 = B01 + Q01; k.p10 = B10 + Q10; k.p11 = B11 + Q11;
    const float hp0 = 1.f * k.p00 + 0.f * k.p10;
    const float hp1 = 1.f * k.p01 + 0.f * k.p11;
    const float S = (hp0 * 1.f + hp1 * 0.f) + 0.5f;
    const float ph0 = k.p00 * 1.f + k.p01 * 0.f;
    const float ph1 = k.p10 * 1.f + k.p11 * 0.f;
    const double invS = 1.0 / (double)S;
    g.p00 = k.p00; g.p01 = k.p01; g.p10 = k.p10; g.p11 = k.p11; g.invS = invS;
    float n00, n01, n10, n11;
    if (order & 2u) {
        const float G00 = ph0 * 1.f, G01 = ph0 * 0.f, G10 = ph1 * 1.f, G11 = ph1 * 0.f;
        const float T00 = G00 * k.p00 + G01 * k.p10, T01 = G00 * k.p01 + G01 * k.p11;
        const float T10 = G10 * k.p00 + G11 * k.p10, T11 = G10 * k.p01 + G11 * k.p11;
        n00 = (float)((double)k.p00 - (double)T00 * invS);
        n01 = (float)((double)k.p01 - (double)T01 * invS);
        n10 = (float)((double)k.p10 - (double)T10 * invS);
        n11 = (float)((double)k.p11 - (double)T11 * invS);
    } else {
        const double K0 = (double)ph0 * invS, K1 = (double)ph1 * invS;
        const double KH00 = K0 * 1.0, KH01 = K0 * 0.0, KH10 = K1 * 1.0, KH11 = K1 * 0.0;
        n00 = (float)((double)k.p00 - (KH00 * (double)k.p00 + KH01 * (double)k.p10));
        n01 = (float)((double)k.p01 - (KH00 * (double)k.p01 + KH01 * (double)k.p11));
        n10 = (float)((double)k.p10 - (KH10 * (double)k.p00 + KH11 * (double)k.p10));
        n11 = (float)((double)k.p11 - (KH10 * (double)k.p01 + KH11 * (double)k.p11));
    }
    k.p00 = n00; k.p01 = n01; k.p10 = n10; k.p11 = n11;
}
template <uint32_t order>
M17_HD void kalman2_state_step(float& x0, float& x1, float z, uint32_t dt_u, int wrap, const Kalman2Gain& g)
{
    const float F00 = 1.f, F01 = (float)dt_u, F10 = 0.f, F11 = 1.f;
    const float nx0 = F00 * x0 + F01 * x1;
    const float nx1 = F10 * x0 + F11 * x1;
    x0 = nx0; x1 = nx1;
    const float fw = (float)wrap;
    if (wrap) {
        if ((double)(z - x0) < ((double)wrap / -2.0)) z += fw;
        else if ((double)(z - x0) > ((double)wrap / 2.0)) z -= fw;
    }
    const float y = z - (1.f * x0 + 0.f * x1);
    if (order & 1u) {
        const float hy0 = 1.f * y, hy1 = 0.f * y;
        const float t0 = g.p00 * hy0 + g.p01 * hy1;
        const float t1 = g.p10 * hy0 + g.p11 * hy1;
        x0 = (float)((double)x0 + (double)t0 * g.invS);
        x1 = (float)((double)x1 + (double)t1 * g.invS);
    } else {
        const float ph0 = g.p00 * 1.f + g.p01 * 0.f;
        const float ph1 = g.p10 * 1.f + g.p11 * 0.f;
        const double K0 = (double)ph0 * g.invS, K1 = (double)ph1 * g.invS;
        x0 = (float)((double)x0 + K0 * (double)y);
        x1 = (float)((double)x1 + K1 * (double)y);
    }
    if (wrap) {
        while (x0 >= fw) x0 -= fw;
        while (x0 < 0.f) x0 += fw;
    }
}
template <uint32_t order>
M17_HD void kalman2_update_as(Kalman2& k, float z, uint32_t dt_u, int wrap)
{
    Kalman2Gain g;
    kalman2_cov_step<order>(k, dt_u, g);
    kalman2_state_step<order>(k.x0, k.x1, z, dt_u, wrap, g);
}
M17_HD void kalman2_update(Kalman2& k, float z, uint32_t dt_u, int wrap, uint32_t order)
{
    switch (order & 7u) {
    case 0: kalman2_update_as<0>(k, z, dt_u, wrap); break;
    case 1: kalman2_update_as<1>(k, z, dt_u, wrap); break;
    case 2: kalman2_update_as<2>(k, z, dt_u, wrap); break;
    case 3: kalman2_update_as<3>(k, z, dt_u, wrap); break;
    case 4: kalman2_update_as<4>(k, z, dt_u, wrap); break;
    case 5: kalman2_update_as<5>(k, z, dt_u, wrap); break;
    case 6: kalman2_update_as<6>(k, z, dt_u, wrap); break;
    default: kalman2_update_as<7>(k, z, dt_u, wrap); break;
    }
}

// ---- a10: the gain SCHEDULE of FreqDevEstimator's two level filters (FreqDevEstimator.h:31-48: both are updated with dt = 192 and
// reset together, KalmanFilter.h:91-107).  Their covariance is therefore a function of the number of updates since the reset alone,
// the same for both, and in fp32 it reaches a FIXED POINT after 558 / 559 updates under every evaluation order (checked where the
// table is built and in tests/cxx/mirror_check.cpp): entry n = the gain of update n + 1 after a reset, entry LEVEL_SCHED_LAST for
// every later one.  A filter is then (x0, x1) plus the shared count: an update is kalman2_state_step with a table entry — no
// covariance arithmetic, no f64 division.
constexpr uint32_t LEVEL_DT = 192;
constexpr int LEVEL_SCHED_N = 576, LEVEL_SCHED_LAST = LEVEL_SCHED_N - 1;
template <uint32_t order>
inline bool level_schedule_as(Kalman2Gain* tab)   // [LEVEL_SCHED_N]; false if the covariance has not settled (cannot happen: see above)
{
    Kalman2 k;
    kalman2_reset(k, 0.f);
    for (int n = 0; n < LEVEL_SCHED_N; ++n) kalman2_cov_step<order>(k, LEVEL_DT, tab[n]);
    Kalman2 k2 = k;
    Kalman2Gain g;
    kalman2_cov_step<order>(k2, LEVEL_DT, g);
    const Kalman2Gain& l = tab[LEVEL_SCHED_LAST];
    return k2.p00 == k.p00 && k2.p01 == k.p01 && k2.p10 == k.p10 && k2.p11 == k.p11 && g.p00 == l.p00 && g.p01 == l.p01 && g.p10 == l.p10 &&
           g.p11 == l.p11 && g.invS == l.invS;
}
inline bool level_schedule(Kalman2Gain* tab, uint32_t order)
{
    switch (order & 7u) {
    case 0: return level_schedule_as<0>(tab);
    case 1: return level_schedule_as<1>(tab);
    case 2: return level_schedule_as<2>(tab);
    case 3: return level_schedule_as<3>(tab);
    case 4: return level_schedule_as<4>(tab);
    case 5: return level_schedule_as<5>(tab);
    case 6: return level_schedule_as<6>(tab);
    default: return level_schedule_as<7>(tab);
    }
}
// one level filter, scheduled: (x0, x1) after the update that uses entry g; `order` only selects how x += K y is associated (bit 0)
M17_HD void level_update(float& x0, float& x1, float z, const Kalman2Gain& g, uint32_t order)
{
    if (order & 1u) kalman2_state_step<1>(x0, x1, z, LEVEL_DT, 0, g);
    else kalman2_state_step<0>(x0, x1, z, LEVEL_DT, 0, g);
}

// ClockRecovery (ClockRecovery.h:54-88): int8 wrap of the rounded estimate into 0..9
M17_HD int32_t wrap10(int32_t v)
{
    v = (int32_t)(int8_t)v;
    v = v < 0 ? v + 10 : v;
    v = v >= 10 ? v - 10 : v;
    return (int32_t)(int8_t)v;
}
// ClockRecovery::update() (:76-88) as a pure function of (sample_estimate_, clock_estimate_, count_):
// fmod(double(float v), 10) is exact; for |v| < 1e12 it is v - 10*trunc(v/10) with one fma (the quotient estimate
// trunc(v * 0.1) is within one of the true one and the +-10 correction below makes up for it), else the library fmod.
M17_HD int32_t clock_predict(float sample_est, float clock_est, uint32_t count)
{
    const float v = sample_est + clock_est * (float)count;
    const double dv = (double)v;
    double csw;
    if (__builtin_fabs(dv) < 1.0e12) {
        const double q = __builtin_trunc(dv * 0.1);
        csw = __builtin_fma(-q, 10.0, dv);
        if (dv >= 0.0) { if (csw < 0.0) csw += 10.0; else if (csw >= 10.0) csw -= 10.0; }
        else { if (csw > 0.0) csw -= 10.0; else if (csw <= -10.0) csw += 10.0; }
    } else {
        csw = __builtin_fmod(dv, 10.0);
    }
    if (csw < 0.) csw += 10;
    else if (csw >= 10) csw -= 10;
    return wrap10((int32_t)__builtin_round(csw));
}
// The same question as a predicate, for the callers that only need "does the update leave sample_index at S?" (K5 checks every
// anti-phase sample of a chunk at once): with v = sample_estimate_ + clock_estimate_ * count_ as the float the reference forms
// (:78), and v in [-10, 19.5), fmod / the +-10 wrap / round-half-away reduce to exact interval tests on v itself:
// clock_predict == S  <=>  v, v + 10 or v - 10 lies in [S - 0.5, S + 0.5)   (S = 0 picks up [9.5, 10) through the third one).
// Outside that range (or NaN) clock_predict_near() is false and the caller has to evaluate clock_predict().
M17_HD float clock_predict_arg(float sample_est, float clock_est, uint32_t count) { return sample_est + clock_est * (float)count; }
M17_HD bool clock_predict_near(float v) { return v >= -10.f && v < 19.5f; }
M17_HD bool clock_predict_equals(float v, int32_t S)
{
    const float lo = (float)S - 0.5f, hi = (float)S + 0.5f;   // exact: S is 0..9
    return (v >= lo && v < hi) || (v >= lo - 10.f && v < hi - 10.f) || (v >= lo + 10.f && v < hi + 10.f);
}
// ClockRecovery::update(uint8_t) (:54-67): the sample index of a fresh filter estimate
M17_HD int32_t clock_index_of(float sample_est) { return wrap10((int32_t)__builtin_round((double)sample_est)); }

// FreqDevEstimator (FreqDevEstimator.h:31-48): offset and inverse deviation from the two smoothed levels
M17_HD float freqdev_offset(float mx0, float mn0) { return (float)((double)(mx0 + mn0) / 2.); }
M17_HD float freqdev_idev(float mx0, float mn0) { return (float)(6.0 / (double)(mx0 - mn0)); }

// ---- a11: SymbolEvm::update (SymbolEvm.h:31-51): distance to the nearest of {-3,-1,1,3} -----------------------------------
M17_HD float evm_error(float sample)
{
    if (sample > 2.f) return sample - 3.f;
    if (sample > 0.f) return sample - 1.f;
    if (sample > -2.f) return sample + 1.f;
    return sample + 3.f;
}
constexpr float EVM_ALPHA = (float)(1.0 / 184);   // RunningStandardDeviation<float,184>::alpha (StandardDeviation.h:60-72)
M17_HD float evm_capture(float S, float err)
{
    S = S - S * EVM_ALPHA;
    S = S + (err * err) * EVM_ALPHA;
    return S;
}

// ---- a12: llr<float,4> (Util.h:63-104,128-145) ------------------------------------------------------------------------------
// 43 table rows whose edges are ACCUMULATED in fp32 (k = -3 + 1/7; k += 1/7), lookup = first edge >= the clamped sample.
constexpr int LLR_ROWS = 43;
M17_HD void llr_edges(float* e43)
{
    const float inc = (float)(1.0 / (double)7.0f);
    float k = (float)(-3.0 + (double)inc);
    for (int n = 0; n < LLR_ROWS; ++n) { e43[n] = k; k = k + inc; }
}
// the (int8, int8) pair of row n: i falls 7..1,-1..-7 over rows 14..27, j falls over rows 0..13 and rises over 28..41
M17_HD uint32_t llr_pair_of_row(int n)
{
    int li, lj;
    if (n <= 14) { li = 7; lj = (n <= 6) ? 7 - n : ((n <= 13) ? 6 - n : -7); }
    else if (n <= 28) { lj = -7; li = (n <= 20) ? 21 - n : ((n <= 27) ? 20 - n : -7); }
    else { li = -7; lj = (n <= 34) ? n - 35 : ((n <= 41) ? n - 34 : 7); }
    return ((uint32_t)(uint8_t)(int8_t)li) | (((uint32_t)(uint8_t)(int8_t)lj) << 8);
}
// row index = std::lower_bound over the edges: guessed arithmetically, corrected against the exact edges (the guess is
// within one row of the answer because the edges are strictly increasing and within 1e-6 of -3 + (n+1)/7)
M17_HD int llr_row(float sample, const float* edges)
{
    const float cl = __builtin_fminf(3.0f, __builtin_fmaxf(-3.0f, sample));
    int n = (int)__builtin_ceilf((cl + 3.0f) * 7.0f) - 1;
    n = n < 1 ? 1 : (n > 41 ? 41 : n);
    const float e0 = edges[n - 1], e1 = edges[n], e2 = edges[n + 1];
    if (e0 >= cl) n = n - 1;
    else if (e1 >= cl) { /* the guess is the row */ }
    else if (e2 >= cl) n = n + 1;
    else n = n + 2;
    return n;
}
M17_HD uint32_t llr_slice(float sample, const float* edges) { return llr_pair_of_row(llr_row(sample, edges)); }

// ---- a17: Viterbi<Trellis<4,2>,4> (Viterbi.h:94-240) scalar pieces -----------------------------------------------------------
template <int LIMIT = 7>
M17_HD uint32_t viterbi_cost_of(int32_t min_metric) { return (uint32_t)(int64_t)__builtin_roundf((float)min_metric / (float)LIMIT); }   // :223

}  // namespace core
}  // namespace mobilinkd
