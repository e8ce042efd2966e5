// Per-channel demodulator state (hot/cold split), kernel parameters and the out-of-line helpers shared by the
// sequential demodulator kernel (m17_wave_kernel.hpp).  Reference: M17Demodulator.h:123-217 (members),
// KalmanFilter.h, ClockRecovery.h, FreqDevEstimator.h, Correlator.h:81-114, DataCarrierDetect.h:63-73.
#pragma once

#include "m17_common.hpp"
#include "m17_decode_device.hpp"
#include "m17_frontend_kernels.hpp"

namespace m17 {

enum : uint32_t { ST_UNLOCKED = 0, ST_LSF_SYNC, ST_STREAM_SYNC, ST_PACKET_SYNC, ST_BERT_SYNC, ST_SYNC_WAIT, ST_FRAME };

struct Kal2 {  // 2-state Kalman filter (KalmanFilter.h:18-108); F, H, R, Q are constants
    float x0, x1, p00, p01, p10, p11;
};

struct Hot {  // per-channel scalars kept in registers while the kernel runs
    uint32_t dcd_trig, dcd_on, count;
    int32_t run_pos;            // samples already fed in the current gated-on run, saturating at 148
    float h0, h1, h2;           // limit IIR history (Correlator.h:38-45)
    uint32_t ring_pos, prev_pos;
    uint32_t sw_trig[4], sw_timing[4];  // SyncWord state: preamble, lsf, packet, eot
    int32_t sw_updated[4];
    uint32_t ck_count;          // ClockRecovery::count_
    int32_t ck_sample_index;
    float ck_clock_est, ck_sample_est;
    float idev, offset, evm_S;
    uint32_t framer_idx, framer_half;
    uint32_t st, sync_word_type, sample_index, sync_sample_index;
    uint32_t need_clock_reset, need_clock_update, eot_flag, viterbi_cost;
    int32_t sync_count, missing_sync_count, initializing;
    uint32_t spec_ok;           // this run still trusts K2's speculative limit-filter history (m17_gate_kernel.hpp)
};
// The hot scalars while K5 runs: the same names as Hot, each one a wave-uniform value that every write forces into a SCALAR
// register (v_readfirstlane), so the state machine's tests and counters are SALU work on registers instead of LDS round
// trips (an LDS word is 64+ cycles away and the wave is latency-bound) and cost no VGPR.  The three SyncWord arrays are
// indexed at run time and stay in the LDS copy of Hot.
template <typename T> struct SReg {
    T v;
    __device__ __forceinline__ static T uni(T x)
    {
        return __builtin_bit_cast(T, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, x)));
    }
    __device__ __forceinline__ operator T() const { return v; }
    __device__ __forceinline__ SReg& operator=(T x) { v = uni(x); return *this; }
    __device__ __forceinline__ SReg& operator=(const SReg& o) { v = o.v; return *this; }
    __device__ __forceinline__ SReg& operator+=(T x) { v = uni((T)(v + x)); return *this; }
    __device__ __forceinline__ SReg& operator-=(T x) { v = uni((T)(v - x)); return *this; }
    __device__ __forceinline__ SReg& operator++() { v = uni((T)(v + 1)); return *this; }
    __device__ __forceinline__ T operator++(int) { const T o = v; v = uni((T)(v + 1)); return o; }
    __device__ __forceinline__ SReg& operator--() { v = uni((T)(v - 1)); return *this; }
};
#define M17_HOT_SCALARS(X)                                                                                                   \
    X(uint32_t, dcd_trig) X(uint32_t, dcd_on) X(uint32_t, count) X(int32_t, run_pos) X(float, h0) X(float, h1) X(float, h2)      \
    X(uint32_t, ring_pos) X(uint32_t, prev_pos) X(uint32_t, ck_count) X(int32_t, ck_sample_index) X(float, ck_clock_est)          \
    X(float, ck_sample_est) X(float, idev) X(float, offset) X(float, evm_S) X(uint32_t, framer_idx) X(uint32_t, framer_half)       \
    X(uint32_t, st) X(uint32_t, sync_word_type) X(uint32_t, sample_index) X(uint32_t, sync_sample_index)                           \
    X(uint32_t, need_clock_reset) X(uint32_t, need_clock_update) X(uint32_t, eot_flag) X(uint32_t, viterbi_cost)                   \
    X(int32_t, sync_count) X(int32_t, missing_sync_count) X(int32_t, initializing) X(uint32_t, spec_ok)
struct HotRegs {
#define X(T, n) SReg<T> n;
    M17_HOT_SCALARS(X)
#undef X
    M17_LDS uint32_t* sw_trig;      // [4] in the LDS copy of Hot
    M17_LDS uint32_t* sw_timing;    // [4]
    M17_LDS int32_t* sw_updated;    // [4]
    __device__ __forceinline__ void load(M17_LDS Hot* h)
    {
#define X(T, n) n = h->n;
        M17_HOT_SCALARS(X)
#undef X
        sw_trig = h->sw_trig; sw_timing = h->sw_timing; sw_updated = h->sw_updated;
    }
    __device__ __forceinline__ void store(M17_LDS Hot* h) const
    {
#define X(T, n) h->n = n.v;
        M17_HOT_SCALARS(X)
#undef X
    }
};
struct Cold {  // per-channel state touched a few times per frame (out-of-line helpers); lives in LDS while K5 runs, like Hot
    Kal2 ck, kmin, kmax;
    uint32_t dev_reset;
    float dcd_level;
    uint32_t seg_start_tick;    // absolute tick index where the current DCD accumulation segment began
    uint32_t dec_state, lich_segments;
    int32_t stale401;
    uint32_t seq;               // frame callbacks since reset
    uint32_t n_run;             // frame callbacks in the current run
    Diag diag;
};
struct SeqState {
    Hot hot;
    Cold cold;
    float ring[80];
    float sw_samples[4][10];
    uint32_t llr[92];
    uint32_t lsf[8];
    int16_t hist[150];          // last 149 gated FIR inputs (raw int16) at the end of the previous run
};

struct SeqParams {
    const int16_t* x;
    size_t xpitch;
    const float* y;
    size_t ypitch;
    const float* dcd_table;   // [C][ticks_cap][12]
    uint32_t ticks_cap;
    SeqState* state;
    FrameRec* recs;           // [C][rec_cap]
    uint32_t rec_cap;
    uint32_t* rec_count;      // [C] records written this run
    uint32_t* overflow;
    const DecodeTables* tables;
    const float* taps;        // 149 floats
    const float* llr_edges;   // 43 floats (Util.h:63-104, float-accumulated; built on the host)
    uint32_t C, T;
    uint64_t pos0;            // absolute index of sample 0 of this (segment of a) run
    uint64_t tick_row0;       // absolute tick stored in row 0 of the DCD table
    uint32_t flags;           // bit 0 invert, bit 1 continuation segment of a run
    unsigned long long* dbg;  // optional [channels][24] counters (diagnostics)
    const float* h;           // K2's limit-filter history, pitch ypitch (nullptr: no speculation, K5 runs the filter itself)
    const float* final_h;     // [C][4] K2's filter history after the last fed sample of the run
    uint32_t* dropped;        // [C] out: this segment dropped the speculation (K2 must redo the channel's next segment from K5's state)
    uint32_t kalman_order;    // evaluation order of the Kalman update (kal_update)
    uint32_t channel_base;    // global id of channel 0 of this context (written into the frame records)
};

// LDS words for a wave of `ls` channels: ring, sync samples, llr, hist, outb, lsf columns; edges, src maps, lich map

__device__ __constant__ const float SW_MAG1[4] = {29.f, 31.f, 31.f, 31.f};
__device__ __constant__ const float SW_MAG2[4] = {-3.402823466e+38f, -31.f, -31.f, -3.402823466e+38f};

// ---- Kalman pieces --------------------------------------------------------------------------------------------
// KalmanFilter.h:41-65,91-107.  `S` and `K` are lazy blaze expressions there (`auto`), so the association / rounding of
// `x += K*y` and `P = P - K*H*P` follows blaze's restructuring operators; blaze is absent from the reference tree, hence the
// order is a switch shared with the oracle (m17hip_set_kalman_order):
//   bit 0: x += double(fl32(P(:,0)*y)) * invS      [(A*s)*v -> (A*v)*s]   else  x += (double(P(:,0))*invS) * double(y)
//   bit 1: P -= double(fl32(P(i,0)*P(0,j))) * invS [(A*s)*B -> (A*B)*s]   else  P -= ((double(P(i,0))*invS) * double(P(0,j)))
//   bit 2: F*(P*F^T) instead of (F*P)*F^T
__device__ __forceinline__ void kal_reset(Kal2& k, float z)
{
    k.x0 = z; k.x1 = 0.f;
    k.p00 = 4.f; k.p01 = 0.f; k.p10 = 0.f; k.p11 = (float)0.00000025;
}
// wrap != 0: KalmanFilter<float,SPS> (index filter, modulo SPS); 0: SymbolKalmanFilter.  State in LDS.
__device__ __noinline__ void kal_update(M17_LDS Kal2* kp, float z, uint32_t dt_u, int wrap, uint32_t order)
{
    Kal2 k = lds_get(kp);
    const float F00 = 1.f, F01 = (float)dt_u, F10 = 0.f, F11 = 1.f;
    const float Q00 = (float)6.25e-13, Q01 = (float)1.25e-12, Q10 = (float)1.25e-12, Q11 = (float)2.50e-12;
    const float nx0 = F00 * k.x0 + F01 * k.x1;
    const float nx1 = F10 * k.x0 + F11 * k.x1;
    k.x0 = nx0; k.x1 = nx1;
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
    const double K0 = (double)ph0 * invS, K1 = (double)ph1 * invS;
    const float fw = (float)wrap;
    if (wrap) {
        if ((double)(z - k.x0) < ((double)wrap / -2.0)) z += fw;
        else if ((double)(z - k.x0) > ((double)wrap / 2.0)) z -= fw;
    }
    const float y = z - (1.f * k.x0 + 0.f * k.x1);
    if (order & 1u) {
        const float hy0 = 1.f * y, hy1 = 0.f * y;
        const float t0 = k.p00 * hy0 + k.p01 * hy1;
        const float t1 = k.p10 * hy0 + k.p11 * hy1;
        k.x0 = (float)((double)k.x0 + (double)t0 * invS);
        k.x1 = (float)((double)k.x1 + (double)t1 * invS);
    } else {
        k.x0 = (float)((double)k.x0 + K0 * (double)y);
        k.x1 = (float)((double)k.x1 + K1 * (double)y);
    }
    if (wrap) {
        while (k.x0 >= fw) k.x0 -= fw;
        while (k.x0 < 0.f) k.x0 += fw;
    }
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
        const double KH00 = K0 * 1.0, KH01 = K0 * 0.0, KH10 = K1 * 1.0, KH11 = K1 * 0.0;
        n00 = (float)((double)k.p00 - (KH00 * (double)k.p00 + KH01 * (double)k.p10));
        n01 = (float)((double)k.p01 - (KH00 * (double)k.p01 + KH01 * (double)k.p11));
        n10 = (float)((double)k.p10 - (KH10 * (double)k.p00 + KH11 * (double)k.p10));
        n11 = (float)((double)k.p11 - (KH10 * (double)k.p01 + KH11 * (double)k.p11));
    }
    k.p00 = n00; k.p01 = n01; k.p10 = n10; k.p11 = n11;
    lds_put(kp, k);
}

__device__ __forceinline__ int32_t wrap10(int32_t v)
{
    v = (int32_t)(int8_t)v;
    v = v < 0 ? v + 10 : v;
    v = v >= 10 ? v - 10 : v;
    return (int32_t)(int8_t)v;
}

// llr<float,4> (Util.h:63-104,128-145): index of the first float-accumulated table edge >= the clamped sample,
// then the (int8,int8) pair of that row.  The index is guessed arithmetically and corrected against the exact
// edges (LDS or global), so the result equals std::lower_bound over the reference's table for every float.
__device__ __forceinline__ uint32_t slice_llr(float sample, const float* edges)
{
    const float cl = fminf(3.0f, fmaxf(-3.0f, sample));
    int n = (int)ceilf((cl + 3.0f) * 7.0f) - 1;
    n = n < 1 ? 1 : (n > 41 ? 41 : n);
    const float e0 = edges[n - 1], e1 = edges[n], e2 = edges[n + 1];
    // edges are strictly increasing; the guess is within one row of the answer
    if (e0 >= cl) n = n - 1;
    else if (e1 >= cl) { /* the guess is the row */ }
    else if (e2 >= cl) n = n + 1;
    else n = n + 2;
    int li, lj;  // Util.h:63-104: i falls 7..1,-1..-7 over rows 14..27, j falls over rows 0..13 and rises over 28..41
    if (n <= 14) { li = 7; lj = (n <= 6) ? 7 - n : ((n <= 13) ? 6 - n : -7); }
    else if (n <= 28) { lj = -7; li = (n <= 20) ? 21 - n : ((n <= 27) ? 20 - n : -7); }
    else { li = -7; lj = (n <= 34) ? n - 35 : ((n <= 41) ? n - 34 : 7); }
    return ((uint32_t)(uint8_t)(int8_t)li) | (((uint32_t)(uint8_t)(int8_t)lj) << 8);
}

// ClockRecovery::update() (ClockRecovery.h:76-88) as a pure function of (sample_estimate_, clock_estimate_, count_).
// std::fmod(double(v), 10) is exact; for |v| < 1e12 it is computed as v - 10*trunc(v/10) with one fma (exact, see
// DESIGN.md §4.5), otherwise by the library fmod.
__device__ __forceinline__ int32_t clock_predict(float sample_est, float clock_est, uint32_t count)
{
    const float v = sample_est + clock_est * (float)count;
    const double dv = (double)v;
    double csw;
    if (fabs(dv) < 1.0e12) {
        const double q = trunc(dv * 0.1);  // within one of trunc(dv/10); the exact remainder below is corrected by +-10
        csw = fma(-q, 10.0, dv);
        if (dv >= 0.0) { if (csw < 0.0) csw += 10.0; else if (csw >= 10.0) csw -= 10.0; }
        else { if (csw > 0.0) csw -= 10.0; else if (csw <= -10.0) csw += 10.0; }
    } else {
        csw = fmod(dv, 10.0);
    }
    if (csw < 0.) csw += 10;
    else if (csw >= 10) csw -= 10;
    return wrap10((int32_t)round(csw));
}

// ---- out-of-line helpers on COLD state ------------------------------------------------------------------------------
// M17Demodulator::update_values (:233-241) = Correlator::outer_symbol_levels (Correlator.h:81-114) +
// FreqDevEstimator::update (FreqDevEstimator.h:31-48).  Returns (idev, offset).
__device__ __forceinline__ float2 nf_update_values(M17_LDS Cold* cd, const float* ring, int stride, int lane, uint32_t si, uint32_t order)
{
    float min_sum = 0.f, max_sum = 0.f;
    uint32_t min_count = 0, max_count = 0;
    float lo = ring[si * stride + lane], hi = lo;
    for (uint32_t i = si; i < 80u; i += 10u) {
        const float v = ring[i * stride + lane];
        lo = (v < lo) ? v : lo;  // std::min(lo, v)
        hi = (hi < v) ? v : hi;  // std::max(hi, v)
    }
    const float avg = (float)((double)hi + (double)lo / 2.);  // sic: Correlator.h:97
    for (uint32_t i = si; i < 80u; i += 10u) {
        const float v = ring[i * stride + lane];
        const bool high = v > avg, low = v < avg;
        max_sum = max_sum + v * (high ? 1.f : 0.f);
        min_sum = min_sum + v * (low ? 1.f : 0.f);
        max_count += high; min_count += low;
    }
    const float mn = min_count > 0 ? min_sum / (float)min_count : lo;
    const float mx = max_count > 0 ? max_sum / (float)max_count : hi;
    kal_update(&cd->kmin, mn, 192u, 0, order);
    kal_update(&cd->kmax, mx, 192u, 0, order);
    const Kal2 a = lds_get(&cd->kmin), b = lds_get(&cd->kmax);
    float offset = (float)((double)(b.x0 + a.x0) / 2.);
    float idev = (float)(6.0 / (double)(b.x0 - a.x0));
    uint32_t rst = cd->dev_reset;
    if (isnan(a.x0) || isnan(a.x1) || isnan(b.x0) || isnan(b.x1)) rst = 1;
    if (rst) {
        Kal2 k;
        kal_reset(k, mn); lds_put(&cd->kmin, k);
        kal_reset(k, mx); lds_put(&cd->kmax, k);
        offset = (mn + mx) / 2.f;
        idev = (float)(6.0 / (double)(mx - mn));
    }
    cd->dev_reset = 0;
    return make_float2(idev, offset);
}
struct ClockOut { float sample_est, clock_est; int32_t sample_index; };
// ClockRecovery::update(uint8_t) (ClockRecovery.h:54-67)
__device__ __forceinline__ ClockOut nf_clock_update_idx(M17_LDS Cold* cd, uint32_t index, uint32_t ck_count, uint32_t order)
{
    kal_update(&cd->ck, (float)index, ck_count, 10, order);
    ClockOut o;
    o.sample_est = cd->ck.x0;
    o.sample_index = wrap10((int32_t)round((double)o.sample_est));
    o.clock_est = cd->ck.x1;
    return o;
}
// DataCarrierDetect::update (:63-69) with the sums K3 produced for the segment [seg_start_tick, k]; returns the trigger.
// (have: the two sums were fetched ahead, pl1 / pl2)
__device__ __forceinline__ uint32_t nf_dcd_update(M17_LDS Cold* cd, const float* tab, uint64_t tick0, uint64_t k, uint32_t trig,
                                                  bool have = false, float pl1 = 0.f, float pl2 = 0.f)
{
    const float* row = tab + (size_t)(k - tick0) * 12;
    const uint32_t span = (uint32_t)(k + 1 - cd->seg_start_tick);
    const int j = span > 5 ? 5 : (int)(cd->seg_start_tick % 5u);
    float l1, l2;  // table row: [2 bins][6 sums]
    if (have) { l1 = pl1; l2 = pl2; } else { l1 = row[j]; l2 = row[6 + j]; }
    const float level = (float)((double)cd->dcd_level * 0.8 + 0.2 * (double)(l1 / l2));
    cd->dcd_level = level;
    cd->seg_start_tick = (uint32_t)(k + 1);
    return trig ? (level > 0.1f) : (level > 4.0f);
}
// arguments of the diagnostic callback (M17Demodulator.h:681-685, 746-750)
__device__ __forceinline__ void nf_fire_diag(M17_LDS Cold* cd, uint32_t dcd_on, float evm_arg, float idev, float offset, uint32_t locked, float clock,
                                          uint32_t sample_index, uint32_t sync_index, int32_t clock_index, uint32_t vcost)
{
    Diag d = lds_get(&cd->diag);
    d.dcd = (int32_t)dcd_on; d.evm = evm_arg; d.deviation = 2400.f / idev; d.offset = offset;
    d.locked = (int32_t)locked; d.clock = clock; d.sample_index = (int32_t)sample_index;
    d.sync_index = (int32_t)sync_index; d.clock_index = (int32_t)(uint8_t)clock_index;
    d.viterbi_cost = (int32_t)vcost; d.dcd_level = cd->dcd_level; d.n_diag++;
    lds_put(&cd->diag, d);
}
__device__ __noinline__ void nf_snapshot_hist(int16_t* hist, const int16_t* xr, uint32_t te)
{
    for (int k = 0; k < 149; ++k) hist[k] = xr[(int64_t)te - 148 + k];
}
// =====================================================================================================
}  // namespace m17
