// ORACLE — TEST INFRASTRUCTURE ONLY.  Scalar CPU restatement of the M17 4-FSK
// demodulation chain of mobilinkd/m17-cxx-demod, written from its behaviour
// (SURVEY.md §8a, §9).  Nothing under oracle/ is part of the product: only
// tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use it,
// and only as the checker / reported CPU baseline.
//
// Parity status of this file: the operators here (FIR, IIR, correlator, sync
// word, sliding DFT, DCD, EVM, LLR slicer) are PINNED against the reference's
// own headers compiled in oracle/_ref (tests/test_oracle_vs_ref.py) and against
// the reference's known-answer tests.  The 2x2 Kalman filters (clock recovery,
// deviation estimator) depend on the absent third-party "blaze" library
// (reference .gitmodules:1-3, pinned version unknown): their float values are
// PARITY UNPINNED, defined here as plain scalar C++ with the usual arithmetic
// conversions, eager evaluation (see kalman section).
//
// Build: g++ -O2 -std=c++17 -ffp-contract=off (no FMA: the reference is built
// for baseline x86-64 where mul and add round separately).
#pragma once

#include <algorithm>
#include <cmath>
#include <complex>
#include <cstddef>
#include <cstdint>
#include <cstring>
#include <limits>

namespace m17o {

// ---------------------------------------------------------------------------
// RRC taps (reference M17Demodulator.h:31-118: 149 symmetric taps + a trailing 0).
// ---------------------------------------------------------------------------
static const double RRC_HALF[75] = {
#include "rrc_half_taps.inc"
};
inline double rrc_tap_d(int i) { return i == 149 ? 0.0 : (i <= 74 ? RRC_HALF[i] : RRC_HALF[148 - i]); }
inline float rrc_tap_f(int i) { return (float)rrc_tap_d(i); }  // double literal narrowed to float

// a1: apps/m17-demod.cpp:486-489 — int16 -> float(double(s) / 41067.0), optional invert first.
inline float scale_sample(int16_t s, bool invert)
{
    if (invert) s = (int16_t)(s * -1);
    return (float)((double)s / 41067.0);
}

// ---------------------------------------------------------------------------
// a2: BaseFirFilter<float,150>  (reference FirFilter.h:13-50)
// circular history, newest sample times taps[0] first, sequential fp32 mul then add.
// ---------------------------------------------------------------------------
struct Fir150 {
    float taps[150];
    float hist[150];
    size_t pos;
    Fir150() { for (int i = 0; i < 150; ++i) taps[i] = rrc_tap_f(i); reset(); }
    void reset() { for (auto& h : hist) h = 0.0f; pos = 0; }
    float step(float in)  // FirFilter.h:28-43
    {
        hist[pos++] = in;
        if (pos == 150) pos = 0;
        float acc = 0.0f;
        size_t idx = pos;
        for (size_t i = 0; i != 150; ++i) {
            idx = (idx != 0 ? idx - 1 : 149);
            float prod = hist[idx] * taps[i];
            acc = acc + prod;
        }
        return acc;
    }
};

// ---------------------------------------------------------------------------
// a3: BaseIirFilter<float,3> (reference IirFilter.h:26-42), coefficients Correlator.h:38-39.
// ---------------------------------------------------------------------------
struct Iir3 {
    float b[3] = {4.24433681e-05f, 8.48867363e-05f, 4.24433681e-05f};
    float a[3] = {1.0f, -1.98148851f, 0.98165828f};
    float h[3] = {0.f, 0.f, 0.f};
    float step(float in)
    {
        h[2] = h[1];
        h[1] = h[0];
        h[0] = in;
        for (int i = 1; i != 3; ++i) { float p = a[i] * h[i]; h[0] = h[0] - p; }
        float r = 0;
        for (int i = 0; i != 3; ++i) { float p = b[i] * h[i]; r = r + p; }
        return r;
    }
};

// ---------------------------------------------------------------------------
// a3/a4/a6: Correlator<float> (reference Correlator.h:19-125).  Storage is
// zero-initialised (SURVEY §9-Q4 semantics).
// ---------------------------------------------------------------------------
struct Correlator {
    static constexpr size_t SYMBOLS = 8, SPS = 10, LEN = 80;
    float ring[LEN];
    float limit_ = 0.f;
    size_t pos = 0, prev_pos = 0;
    Iir3 lpf;
    Correlator() { for (auto& r : ring) r = 0.f; }

    void sample(float v)  // Correlator.h:43-49
    {
        limit_ = lpf.step(std::fabs(v));
        ring[pos] = v;
        prev_pos = pos;
        if (++pos == LEN) pos = 0;
    }
    float correlate(const int8_t* sync) const  // Correlator.h:51-64 (oldest symbol first)
    {
        float r = 0.f;
        size_t p = prev_pos + SPS;
        for (size_t i = 0; i != SYMBOLS; ++i) {
            if (p >= LEN) p -= LEN;
            float prod = (float)sync[i] * ring[p];
            r = r + prod;
            p += SPS;
        }
        return r;
    }
    float limit() const { return limit_; }
    size_t index() const { return prev_pos % SPS; }

    // Correlator.h:81-114 incl. the `max + min/2.` precedence quirk (Q7), slot order.
    void outer_symbol_levels(size_t si, float& mn, float& mx) const
    {
        float min_sum = 0, max_sum = 0;
        size_t min_count = 0, max_count = 0;
        float lo = ring[si], hi = ring[si];
        for (size_t i = si; i < LEN; i += SPS) {
            lo = std::min(lo, ring[i]);
            hi = std::max(hi, ring[i]);
        }
        float avg = (float)((double)hi + (double)lo / 2.);
        for (size_t i = si; i < LEN; i += SPS) {
            bool high = ring[i] > avg;
            bool low = ring[i] < avg;
            max_sum = max_sum + ring[i] * (float)high;
            min_sum = min_sum + ring[i] * (float)low;
            max_count += high;
            min_count += low;
        }
        mn = min_count > 0 ? min_sum / (float)min_count : lo;
        mx = max_count > 0 ? max_sum / (float)max_count : hi;
    }
};

// ---------------------------------------------------------------------------
// a5: SyncWord<Correlator> (reference Correlator.h:127-208).  find_peak uses the
// float overload of abs (Q5).
// ---------------------------------------------------------------------------
struct SyncWord {
    int8_t word[8];
    float samples[10];
    size_t timing_index = 0;
    bool trig = false;
    int8_t updated_ = 0;
    float mag1, mag2;
    SyncWord(std::initializer_list<int> w, float m1, float m2 = std::numeric_limits<float>::lowest())
        : mag1(m1), mag2(m2)
    {
        int i = 0;
        for (int v : w) word[i++] = (int8_t)v;
        for (auto& s : samples) s = 0.f;
    }
    float triggered(const Correlator& c) const  // Correlator.h:150-157
    {
        float l1 = c.limit() * mag1;
        float l2 = c.limit() * mag2;
        float v = c.correlate(word);
        return (v > l1 || v < l2) ? v : 0.0f;
    }
    void find_peak(float value)  // Correlator.h:161-177
    {
        trig = false;
        timing_index = 0;
        float peak = value;
        uint8_t idx = 0;
        for (float f : samples) {
            if (std::fabs(f) > std::fabs(peak)) { peak = f; timing_index = idx; }
            idx += 1;
        }
        updated_ = peak > 0 ? 1 : -1;
    }
    size_t step(const Correlator& c)  // operator(), Correlator.h:179-200
    {
        float v = triggered(c);
        if (v != 0) {
            if (!trig) { for (auto& s : samples) s = 0.f; trig = true; }
            samples[c.index()] = v;
        } else if (trig) {
            find_peak(v);
        }
        return timing_index;
    }
    int8_t updated() { int8_t r = updated_; updated_ = 0; return r; }
};

// ---------------------------------------------------------------------------
// a7: NSlidingDFT<float,48000,120,2> (reference SlidingDFT.h:70-133)
// a8: DataCarrierDetect<float,48000,400>{2400,3600,0.1,4.0} (DataCarrierDetect.h:28-74)
// ---------------------------------------------------------------------------
inline std::complex<float> dft_coeff(size_t freq)  // SlidingDFT.h:85-95
{
    const std::complex<float> j{0, 1};
    const float pi2 = (float)(M_PI * 2.0);
    float k = float(freq) / float(48000);
    return std::exp(-j * pi2 * k);
}

struct Dcd {
    size_t N = 120;       // SampleRate / Accuracy (hot path: 48000 / 400)
    float cr[2], ci[2];
    float ring[1024];
    float xr[2] = {0, 0}, xi[2] = {0, 0};
    size_t idx = 0;
    float ltrig = 0.1f, htrig = 4.0f;
    float level_1 = 0.f, level_2 = 0.f, level_ = 0.f;
    bool trig = false;
    explicit Dcd(size_t n = 120, size_t f1 = 2400, size_t f2 = 3600, float lt = 0.1f, float ht = 4.0f)
        : N(n), ltrig(lt), htrig(ht)
    {
        auto c0 = dft_coeff(f1), c1 = dft_coeff(f2);
        cr[0] = c0.real(); ci[0] = c0.imag();
        cr[1] = c1.real(); ci[1] = c1.imag();
        for (auto& r : ring) r = 0.f;
    }
    void step(float s)  // DataCarrierDetect.h:53-58 + SlidingDFT.h:118-132
    {
        size_t i = idx;
        idx += 1;
        if (idx == N) idx = 0;
        float delta = s - ring[i];
        for (int k = 0; k < 2; ++k) {
            float a = xr[k] + delta, b = xi[k];           // complex + real
            float ac = a * cr[k], bd = b * ci[k];        // libstdc++ complex multiply,
            float ad = a * ci[k], bc = b * cr[k];        // (ac-bd, ad+bc), no FMA
            xr[k] = ac - bd;
            xi[k] = ad + bc;
        }
        ring[i] = s;
        level_1 = level_1 + (xr[0] * xr[0] + xi[0] * xi[0]);  // std::norm
        level_2 = level_2 + (xr[1] * xr[1] + xi[1] * xi[1]);
    }
    void update()  // DataCarrierDetect.h:63-69 (EMA evaluated in double, Q7)
    {
        level_ = (float)((double)level_ * 0.8 + 0.2 * (double)(level_1 / level_2));
        level_1 = 0.f;
        level_2 = 0.f;
        trig = trig ? level_ > ltrig : level_ > htrig;
    }
    void unlock() { trig = false; }
    float level() const { return level_; }
    bool dcd() const { return trig; }
};

// ---------------------------------------------------------------------------
// a11: SymbolEvm<float> + RunningStandardDeviation<float,184>
// (reference SymbolEvm.h:21-52, StandardDeviation.h:57-83)
// ---------------------------------------------------------------------------
struct SymbolEvm {
    float S = 1.0f;
    float alpha = (float)(1.0 / 184);
    void reset() { S = 0.0f; }
    void capture(float e) { S = S - S * alpha; S = S + (e * e) * alpha; }
    void update(float s)
    {
        if (s > 2) capture(s - 3);
        else if (s > 0) capture(s - 1);
        else if (s > -2) capture(s + 1);
        else capture(s + 3);
    }
    float evm() const { return std::sqrt(S); }
};

// ---------------------------------------------------------------------------
// a12: llr<float,4> (reference Util.h:63-104,128-145): 43-entry table with
// float-accumulated edges (Q8); lookup = first edge >= clamped sample.
// ---------------------------------------------------------------------------
struct LlrTable {
    float edge[43];
    int8_t l0[43], l1[43];
    LlrTable()
    {
        const int8_t limit = 7;
        const float inc = (float)(1.0 / (double)float(limit));
        int8_t i = limit, j = limit;
        float k = (float)(-3.0 + (double)inc);
        for (size_t n = 0; n != 43; ++n) {
            edge[n] = k; l0[n] = i; l1[n] = j;
            if ((double)k + 1.0 < 0) { j--; if (j == 0) j = -1; if (j < -limit) j = -limit; }
            else if ((double)k - 1.0 < 0) { i--; if (i == 0) i = -1; if (i < -limit) i = -limit; }
            else { j++; if (j == 0) j = 1; if (j > limit) j = limit; }
            k = k + inc;
        }
    }
    void lookup(float sample, int8_t& a, int8_t& b) const
    {
        float s = std::min(3.0f, std::max(-3.0f, sample));
        size_t n = 0;
        while (n != 43 && edge[n] < s) ++n;  // lower_bound
        if (n == 43) n = 42;
        a = l0[n]; b = l1[n];
    }
};
inline const LlrTable& llr_table() { static const LlrTable t; return t; }

// ---------------------------------------------------------------------------
// a9/a10: Kalman filters — PARITY UNPINNED (blaze, the reference's linear-algebra dependency, is absent from
// /root/reference: `.gitmodules:1-3`, empty submodule, pinned version unknown).  Reference call sites:
// KalmanFilter.h:18-108, ClockRecovery.h:16-111, FreqDevEstimator.h:14-54.
//
// KalmanFilter.h:49-64 keeps `S` and `K` as `auto`, i.e. as lazy blaze expression templates, so how the products are
// associated and where they are rounded is decided by blaze's restructuring operators, not by the source text.  The
// evaluation order is therefore a switch (kalman_order(), bit set), the same one the HIP path has
// (m17hip_set_kalman_order):
//   bit 0  x += K*y   : 1 = blaze's `(A*s)*v -> (A*v)*s` and `(A*B)*v -> A*(B*v)`: t = fl32(P(:,0)*y), x += double(t)*invS
//                        0 = eager K = double(P(:,0))*invS, x += K*double(y)
//   bit 1  P -= K*H*P : 1 = blaze's `(A*s)*B -> (A*B)*s` twice: T = fl32(P(i,0)*P(0,j)), P -= double(T)*invS
//                        0 = eager (K*H)*P in double
//   bit 2  F*P*F^T    : 0 = (F*P)*F^T as C++ parses it (blaze evaluates the left product into a float temporary)
//                        1 = F*(P*F^T)
// All variants share: element type of a product = common type of its operands (`* (1.0 / S(0,0))` promotes to double),
// results narrowed when assigned to the float members, S(0,0) = P(0,0) + R in float.
// Default 3 = what blaze's documented restructuring rules give.  tools/kalman_sensitivity.py measures how many frame
// records the choice can move.
// ---------------------------------------------------------------------------
#ifndef M17O_KALMAN_ORDER
#define M17O_KALMAN_ORDER 3
#endif
inline int& kalman_order() { static int v = M17O_KALMAN_ORDER; return v; }

struct Kalman2 {
    float x[2], P[2][2], F[2][2];
    float Q[2][2] = {{(float)6.25e-13, (float)1.25e-12}, {(float)1.25e-12, (float)2.50e-12}};
    float R = 0.5f;
    Kalman2() { reset(0.f); }
    void reset(float z)
    {
        x[0] = z; x[1] = 0.f;
        P[0][0] = 4.f; P[0][1] = 0.f; P[1][0] = 0.f; P[1][1] = (float)0.00000025;
        F[0][0] = 1.f; F[0][1] = 1.f; F[1][0] = 0.f; F[1][1] = 1.f;
    }
    // wrap != 0: KalmanFilter<float,SPS> (index filter, modulo SPS); 0: SymbolKalmanFilter.
    void update(float z, size_t dt, int wrap)
    {
        const int order = kalman_order();
        F[0][1] = (float)dt;
        // x = F * x
        float nx0 = F[0][0] * x[0] + F[0][1] * x[1];
        float nx1 = F[1][0] * x[0] + F[1][1] * x[1];
        x[0] = nx0; x[1] = nx1;
        // P = F * P * trans(F) + Q
        float A[2][2], B[2][2];
        if (!(order & 4)) {
            for (int i = 0; i < 2; ++i)
                for (int j = 0; j < 2; ++j) A[i][j] = F[i][0] * P[0][j] + F[i][1] * P[1][j];
            for (int i = 0; i < 2; ++i)
                for (int j = 0; j < 2; ++j) B[i][j] = A[i][0] * F[j][0] + A[i][1] * F[j][1];
        } else {
            for (int i = 0; i < 2; ++i)
                for (int j = 0; j < 2; ++j) A[i][j] = P[i][0] * F[j][0] + P[i][1] * F[j][1];
            for (int i = 0; i < 2; ++i)
                for (int j = 0; j < 2; ++j) B[i][j] = F[i][0] * A[0][j] + F[i][1] * A[1][j];
        }
        for (int i = 0; i < 2; ++i)
            for (int j = 0; j < 2; ++j) P[i][j] = B[i][j] + Q[i][j];
        // S = H * P * trans(H) + R, H = [1 0]
        float hp0 = 1.f * P[0][0] + 0.f * P[1][0];
        float hp1 = 1.f * P[0][1] + 0.f * P[1][1];
        float S = (hp0 * 1.f + hp1 * 0.f) + R;
        // K = P * trans(H) * (1.0 / S)   -> double elements
        float ph0 = P[0][0] * 1.f + P[0][1] * 0.f;
        float ph1 = P[1][0] * 1.f + P[1][1] * 0.f;
        double invS = 1.0 / (double)S;
        double K0 = (double)ph0 * invS, K1 = (double)ph1 * invS;
        if (wrap) {
            if ((double)(z - x[0]) < (wrap / -2.0)) z += wrap;
            else if ((double)(z - x[0]) > (wrap / 2.0)) z -= wrap;
        }
        float y = z - (1.f * x[0] + 0.f * x[1]);
        if (order & 1) {
            const float hy0 = 1.f * y, hy1 = 0.f * y;              // trans(H) * y
            const float t0 = P[0][0] * hy0 + P[0][1] * hy1;         // P * (trans(H) * y), float
            const float t1 = P[1][0] * hy0 + P[1][1] * hy1;
            x[0] = (float)((double)x[0] + (double)t0 * invS);
            x[1] = (float)((double)x[1] + (double)t1 * invS);
        } else {
            x[0] = (float)((double)x[0] + K0 * (double)y);
            x[1] = (float)((double)x[1] + K1 * (double)y);
        }
        if (wrap) {
            while (x[0] >= wrap) x[0] -= wrap;
            while (x[0] < 0) x[0] += wrap;
        }
        // P = P - K * H * P
        float NP[2][2];
        if (order & 2) {
            const float G[2][2] = {{ph0 * 1.f, ph0 * 0.f}, {ph1 * 1.f, ph1 * 0.f}};   // (P * trans(H)) * H, float
            for (int i = 0; i < 2; ++i)
                for (int j = 0; j < 2; ++j) {
                    const float t = G[i][0] * P[0][j] + G[i][1] * P[1][j];            // (...) * P, float
                    NP[i][j] = (float)((double)P[i][j] - (double)t * invS);
                }
        } else {
            double KH[2][2] = {{K0 * 1.0, K0 * 0.0}, {K1 * 1.0, K1 * 0.0}};
            for (int i = 0; i < 2; ++i)
                for (int j = 0; j < 2; ++j)
                    NP[i][j] = (float)((double)P[i][j] - (KH[i][0] * (double)P[0][j] + KH[i][1] * (double)P[1][j]));
        }
        std::memcpy(P, NP, sizeof(P));
    }
};

struct ClockRecovery {  // reference ClockRecovery.h:16-111
    Kalman2 kf;
    size_t count = 0;
    int8_t sample_index_ = 0;
    float clock_estimate_ = 0.f, sample_estimate_ = 0.f;
    static int8_t wrap10(int8_t v)
    {
        v = v < 0 ? v + 10 : v;
        v = v >= 10 ? v - 10 : v;
        return v;
    }
    void reset(float index)
    {
        kf.reset(index);
        count = 0;
        sample_index_ = (int8_t)index;
        clock_estimate_ = 0.f;
    }
    void tick() { ++count; }
    void update(uint8_t index)
    {
        kf.update((float)index, count, 10);
        sample_estimate_ = kf.x[0];
        sample_index_ = wrap10((int8_t)std::round((double)sample_estimate_));
        clock_estimate_ = kf.x[1];
        count = 0;
    }
    void update()
    {
        float v = sample_estimate_ + clock_estimate_ * (float)count;
        double csw = std::fmod((double)v, 10.0);
        if (csw < 0.) csw += 10;
        else if (csw >= 10) csw -= 10;
        sample_index_ = wrap10((int8_t)std::round(csw));
    }
    float clock_estimate() const { return clock_estimate_; }
    uint8_t sample_index() const { return (uint8_t)sample_index_; }
};

struct FreqDevEstimator {  // reference FreqDevEstimator.h:14-54
    Kalman2 minF, maxF;
    float idev_ = 0.f, offset_ = 0.f;
    bool reset_ = true;
    void reset() { reset_ = true; }
    void update(float mn, float mx)
    {
        minF.update(mn, 192, 0);
        maxF.update(mx, 192, 0);
        offset_ = (float)((double)(maxF.x[0] + minF.x[0]) / 2.);
        idev_ = (float)(6.0 / (double)(maxF.x[0] - minF.x[0]));
        if (std::isnan(minF.x[0]) || std::isnan(minF.x[1]) || std::isnan(maxF.x[0]) || std::isnan(maxF.x[1]))
            reset_ = true;
        if (reset_) {
            reset_ = false;
            minF.reset(mn);
            maxF.reset(mx);
            offset_ = (mn + mx) / 2;
            idev_ = (float)(6.0 / (double)(mx - mn));
        }
    }
    float idev() const { return idev_; }
    float offset() const { return offset_; }
    float deviation() const { return 2400.f / idev_; }
    float error() const { return 0.f; }
};

}  // namespace m17o
