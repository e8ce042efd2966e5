// K5: the per-channel sequential glue — reference a3, a5, a6, a8 (decision half), a9-a13, a19:
// Correlator::sample / SyncWord / outer_symbol_levels (Correlator.h), DataCarrierDetect::update,
// ClockRecovery + KalmanFilter, FreqDevEstimator, SymbolEvm, llr<float,4>, M17Framer and the
// M17Demodulator state machine (M17Demodulator.h:233-753) — plus K4 (frame decode) run in batches.
//
// Everything here is a recurrence or a state machine per channel, so the mapping is ONE LANE PER CHANNEL,
// one wave per workgroup.  The massively parallel work (K1 FIR) and the state-machine-independent
// recurrence (K3 sliding DFT) have already run as their own passes; this kernel consumes
//   ybuf[c][t]    the matched-filter output, valid wherever the last 149 FIR inputs were consecutive samples
//   dcd table     the sequential carrier-detect sums for every possible segment (see K3)
// The reference gates the FIR and the correlator with the carrier detect: their input is the
// concatenation of gated-on runs.  Runs start and end on tick boundaries and last >= 960 samples, so only the
// first 148 outputs of a run see samples of the previous run; for those the lane recomputes the FIR itself from
// a 149-sample snapshot taken when the previous run ended ("slow path", rare).
//
// Lanes do NOT advance in lock step in time.  A lane that completes a frame parks until the wave runs a
// decode batch (one lane per frame, see m17_decode_device.hpp); since every locked channel completes a frame
// every 1920 samples, parked lanes re-align themselves after the first batch and from then on the wave decodes
// up to 64 frames at once, once per frame period, with no Viterbi divergence.
#pragma once

#include "m17_common.hpp"
#include "m17_decode_device.hpp"
#include "m17_frontend_kernels.hpp"

namespace m17 {

enum : uint32_t { ST_UNLOCKED = 0, ST_LSF_SYNC, ST_STREAM_SYNC, ST_PACKET_SYNC, ST_BERT_SYNC, ST_SYNC_WAIT, ST_FRAME };

struct Kal2 {  // 2-state Kalman filter (KalmanFilter.h:18-108); F, H, R, Q are constants
    float x0, x1, p00, p01, p10, p11;
};

// Everything one channel carries between runs.  AoS in global memory; the scalars are loaded into
// registers once per launch, the arrays live in LDS (or stay in global memory: hist) while the kernel runs.
struct SeqScalars {
    // carrier-detect decision (DataCarrierDetect.h:63-73) and gating
    float dcd_level;
    uint32_t dcd_trig, dcd_on, count;
    uint32_t seg_start_tick;   // absolute tick index where the current DCD accumulation segment began
    uint32_t first_update_done;
    int32_t run_pos;           // samples already fed in the current gated-on run, saturating at 148
    // correlator: IIR history, ring position
    float h0, h1, h2;
    uint32_t ring_pos, prev_pos;
    // sync words: preamble, lsf, packet, eot
    uint32_t sw_trig[4], sw_timing[4];
    int32_t sw_updated[4];
    // clock recovery
    Kal2 ck;
    uint32_t ck_count;
    int32_t ck_sample_index;
    float ck_clock_est, ck_sample_est;
    // deviation / offset estimator
    Kal2 kmin, kmax;
    float idev, offset;
    uint32_t dev_reset;
    float evm_S;
    uint32_t framer_idx;
    uint32_t framer_half;      // LLR pair of the previous (even) symbol, waiting to be packed
    // frame decoder
    uint32_t dec_state, lich_segments;
    int32_t stale401;
    // demodulator
    uint32_t st, sync_word_type, sample_index, sync_sample_index;
    uint32_t need_clock_reset, need_clock_update, eot_flag;
    uint32_t viterbi_cost;
    int32_t sync_count, missing_sync_count, initializing;
    uint32_t seq;              // frame callbacks since reset
    Diag diag;
};
struct SeqState {
    SeqScalars sc;
    float ring[80];
    float sw_samples[4][10];
    uint32_t llr[92];
    uint32_t lsf[8];
    int16_t hist[150];         // last 149 gated FIR inputs (raw int16) at the end of the previous run
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
    uint64_t pos0;
    uint32_t flags;
};

constexpr int SEQ_LDS_WORDS = (80 + 40 + 92 + 122 + 8 + 8) * 64;  // ring, sync samples, llr, hist, outb, lsf

__device__ __constant__ const float SW_MAG1[4] = {29.f, 31.f, 31.f, 31.f};
__device__ __constant__ const float SW_MAG2[4] = {-3.402823466e+38f, -31.f, -31.f, -3.402823466e+38f};

// ---- Kalman pieces (semantics: eager evaluation, usual arithmetic conversions) --------------
__device__ __forceinline__ void kal_reset(Kal2& k, float z)
{
    k.x0 = z; k.x1 = 0.f;
    k.p00 = 4.f; k.p01 = 0.f; k.p10 = 0.f; k.p11 = (float)0.00000025;
}
template <int WRAP>
__device__ __forceinline__ void kal_update(Kal2& k, float z, uint32_t dt_u)
{
    const float F00 = 1.f, F01 = (float)dt_u, F10 = 0.f, F11 = 1.f;
    const float Q00 = (float)6.25e-13, Q01 = (float)1.25e-12, Q10 = (float)1.25e-12, Q11 = (float)2.50e-12;
    const float nx0 = F00 * k.x0 + F01 * k.x1;
    const float nx1 = F10 * k.x0 + F11 * k.x1;
    k.x0 = nx0; k.x1 = nx1;
    const float A00 = F00 * k.p00 + F01 * k.p10, A01 = F00 * k.p01 + F01 * k.p11;
    const float A10 = F10 * k.p00 + F11 * k.p10, A11 = F10 * k.p01 + F11 * k.p11;
    const float B00 = A00 * F00 + A01 * F01, B01 = A00 * F10 + A01 * F11;
    const float B10 = A10 * F00 + A11 * F01, B11 = A10 * F10 + A11 * F11;
    k.p00 = B00 + Q00; k.p01 = B01 + Q01; k.p10 = B10 + Q10; k.p11 = B11 + Q11;
    const float hp0 = 1.f * k.p00 + 0.f * k.p10;
    const float hp1 = 1.f * k.p01 + 0.f * k.p11;
    const float S = (hp0 * 1.f + hp1 * 0.f) + 0.5f;
    const float ph0 = k.p00 * 1.f + k.p01 * 0.f;
    const float ph1 = k.p10 * 1.f + k.p11 * 0.f;
    const double invS = 1.0 / (double)S;
    const double K0 = (double)ph0 * invS, K1 = (double)ph1 * invS;
    if (WRAP) {
        if ((double)(z - k.x0) < (WRAP / -2.0)) z += (float)WRAP;
        else if ((double)(z - k.x0) > (WRAP / 2.0)) z -= (float)WRAP;
    }
    const float y = z - (1.f * k.x0 + 0.f * k.x1);
    k.x0 = (float)((double)k.x0 + K0 * (double)y);
    k.x1 = (float)((double)k.x1 + K1 * (double)y);
    if (WRAP) {
        while (k.x0 >= (float)WRAP) k.x0 -= (float)WRAP;
        while (k.x0 < 0.f) k.x0 += (float)WRAP;
    }
    const double KH00 = K0 * 1.0, KH01 = K0 * 0.0, KH10 = K1 * 1.0, KH11 = K1 * 0.0;
    const float n00 = (float)((double)k.p00 - (KH00 * (double)k.p00 + KH01 * (double)k.p10));
    const float n01 = (float)((double)k.p01 - (KH00 * (double)k.p01 + KH01 * (double)k.p11));
    const float n10 = (float)((double)k.p10 - (KH10 * (double)k.p00 + KH11 * (double)k.p10));
    const float n11 = (float)((double)k.p11 - (KH10 * (double)k.p01 + KH11 * (double)k.p11));
    k.p00 = n00; k.p01 = n01; k.p10 = n10; k.p11 = n11;
}

__device__ __forceinline__ int32_t wrap10(int32_t v)
{
    v = (int32_t)(int8_t)v;
    v = v < 0 ? v + 10 : v;
    v = v >= 10 ? v - 10 : v;
    return (int32_t)(int8_t)v;
}

// =====================================================================================================
__global__ __launch_bounds__(64) void demod_seq_kernel(SeqParams P)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
    float* ring = reinterpret_cast<float*>(lds);            // [80][64]
    float* swsm = ring + 80 * 64;                           // [4*10][64]
    DecodeLds DL;
    DL.llr = lds + (80 + 40) * 64;                          // [92][64]
    DL.hist = DL.llr + 92 * 64;                             // [122][64]
    DL.outb = DL.hist + 122 * 64;                           // [8][64]
    DL.lsf = DL.outb + 8 * 64;                              // [8][64]

    const int lane = threadIdx.x;
    const uint32_t c = blockIdx.x * 64 + lane;
    const bool valid = c < P.C;
    const uint32_t cc = valid ? c : P.C - 1;
    const bool invert = P.flags & 1u;
    SeqState* gs = P.state + cc;
    SeqScalars s = gs->sc;  // scalar members -> registers; arrays are copied to LDS below
    for (int k = 0; k < 80; ++k) ring[k * 64 + lane] = gs->ring[k];
    for (int k = 0; k < 40; ++k) swsm[k * 64 + lane] = gs->sw_samples[k / 10][k % 10];
    for (int k = 0; k < 92; ++k) DL.llr[k * 64 + lane] = gs->llr[k];
    for (int k = 0; k < 8; ++k) DL.lsf[k * 64 + lane] = gs->lsf[k];

    const int16_t* xr = P.x + (size_t)cc * P.xpitch + XPRE;
    const float* yr = P.y + (size_t)cc * P.ypitch + YPRE;
    const float* tab = P.dcd_table + (size_t)cc * P.ticks_cap * 12;
    const uint64_t tick0 = P.pos0 / TICK;
    FrameRec* rec_base = P.recs + (size_t)cc * P.rec_cap;
    uint32_t n_run = 0;
    uint32_t t = valid ? 0u : P.T;   // next sample (relative to this run)
    bool pending = false;            // frame complete, waiting for the decode batch
    uint32_t pend_sync_type = 0;
    uint64_t pend_pos = 0;

    // ---------------- helpers (all per-lane) ---------------------------------------------------------------
    auto corr_index = [&]() -> uint32_t { return s.prev_pos % 10u; };
    auto limit_now = [&]() -> float { return iir_output(s.h0, s.h1, s.h2); };
    auto correlate = [&](int w) -> float {  // Correlator.h:51-64
        float r = 0.f;
        uint32_t p = s.prev_pos + 10u;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (p >= 80u) p -= 80u;
            const float prod = (float)SYNC_WORDS[w][i] * ring[p * 64 + lane];
            r = r + prod;
            p += 10u;
        }
        return r;
    };
    auto sw_triggered = [&](int w) -> float {  // Correlator.h:150-157
        const float lim = limit_now();
        const float l1 = lim * SW_MAG1[w];
        const float l2 = lim * SW_MAG2[w];
        const float v = correlate(w);
        return (v > l1 || v < l2) ? v : 0.0f;
    };
    auto sw_step = [&](int w) -> uint32_t {  // SyncWord::operator() :179-200 (+ find_peak :161-177)
        const float v = sw_triggered(w);
        if (v != 0.f) {
            if (!s.sw_trig[w]) {
                for (int k = 0; k < 10; ++k) swsm[(w * 10 + k) * 64 + lane] = 0.f;
                s.sw_trig[w] = 1;
            }
            swsm[(w * 10 + (int)corr_index()) * 64 + lane] = v;
        } else if (s.sw_trig[w]) {
            s.sw_trig[w] = 0;
            s.sw_timing[w] = 0;
            float peak = v;
            for (int k = 0; k < 10; ++k) {
                const float f = swsm[(w * 10 + k) * 64 + lane];
                if (fabsf(f) > fabsf(peak)) { peak = f; s.sw_timing[w] = (uint32_t)k; }
            }
            s.sw_updated[w] = peak > 0.f ? 1 : -1;
        }
        return s.sw_timing[w];
    };
    auto sw_updated = [&](int w) -> int32_t { const int32_t r = s.sw_updated[w]; s.sw_updated[w] = 0; return r; };

    auto outer_levels = [&](uint32_t si, float& mn, float& mx) {  // Correlator.h:81-114
        float min_sum = 0.f, max_sum = 0.f;
        uint32_t min_count = 0, max_count = 0;
        float lo = ring[si * 64 + lane], hi = lo;
        for (uint32_t i = si; i < 80u; i += 10u) {
            const float v = ring[i * 64 + lane];
            lo = (v < lo) ? v : lo;  // std::min(lo, v)
            hi = (hi < v) ? v : hi;  // std::max(hi, v)
        }
        const float avg = (float)((double)hi + (double)lo / 2.);
        for (uint32_t i = si; i < 80u; i += 10u) {
            const float v = ring[i * 64 + lane];
            const bool high = v > avg, low = v < avg;
            max_sum = max_sum + v * (high ? 1.f : 0.f);
            min_sum = min_sum + v * (low ? 1.f : 0.f);
            max_count += high; min_count += low;
        }
        mn = min_count > 0 ? min_sum / (float)min_count : lo;
        mx = max_count > 0 ? max_sum / (float)max_count : hi;
    };
    auto dev_update = [&](float mn, float mx) {  // FreqDevEstimator.h:31-48
        kal_update<0>(s.kmin, mn, 192u);
        kal_update<0>(s.kmax, mx, 192u);
        s.offset = (float)((double)(s.kmax.x0 + s.kmin.x0) / 2.);
        s.idev = (float)(6.0 / (double)(s.kmax.x0 - s.kmin.x0));
        if (isnan(s.kmin.x0) || isnan(s.kmin.x1) || isnan(s.kmax.x0) || isnan(s.kmax.x1)) s.dev_reset = 1;
        if (s.dev_reset) {
            s.dev_reset = 0;
            kal_reset(s.kmin, mn);
            kal_reset(s.kmax, mx);
            s.offset = (mn + mx) / 2.f;
            s.idev = (float)(6.0 / (double)(mx - mn));
        }
    };
    auto update_values = [&](uint32_t index) {  // M17Demodulator.h:233-241
        float mn, mx;
        outer_levels(s.sample_index, mn, mx);
        dev_update(mn, mx);
        s.sync_sample_index = index;
    };
    auto clock_reset = [&](float index) {  // ClockRecovery.h:33-39
        kal_reset(s.ck, index);
        s.ck_count = 0;
        s.ck_sample_index = (int32_t)(int8_t)index;
        s.ck_clock_est = 0.f;
    };
    auto clock_update_idx = [&](uint32_t index) {  // ClockRecovery.h:54-67
        kal_update<10>(s.ck, (float)index, s.ck_count);
        s.ck_sample_est = s.ck.x0;
        s.ck_sample_index = wrap10((int32_t)round((double)s.ck_sample_est));
        s.ck_clock_est = s.ck.x1;
        s.ck_count = 0;
    };
    auto clock_update = [&]() {  // ClockRecovery.h:76-88
        const float v = s.ck_sample_est + s.ck_clock_est * (float)s.ck_count;
        double csw = fmod((double)v, 10.0);
        if (csw < 0.) csw += 10;
        else if (csw >= 10) csw -= 10;
        s.ck_sample_index = wrap10((int32_t)round(csw));
    };
    auto dcd_unlock = [&]() { s.dcd_trig = 0; };
    auto fire_diag = [&](float evm_arg) {
        s.diag.dcd = (int32_t)s.dcd_on; s.diag.evm = evm_arg; s.diag.deviation = 2400.f / s.idev; s.diag.offset = s.offset;
        s.diag.locked = (s.st != ST_UNLOCKED); s.diag.clock = s.ck_clock_est; s.diag.sample_index = (int32_t)s.sample_index;
        s.diag.sync_index = (int32_t)s.sync_sample_index; s.diag.clock_index = (int32_t)(uint8_t)s.ck_sample_index;
        s.diag.viterbi_cost = (int32_t)s.viterbi_cost; s.diag.dcd_level = s.dcd_level; s.diag.n_diag++;
    };
    // snapshot of the last 149 gated FIR inputs when a run ends at relative sample te (inclusive)
    auto snapshot_hist = [&](uint32_t te) {
        for (int k = 0; k < 149; ++k) gs->hist[k] = xr[(int64_t)te - 148 + k];
    };
    auto update_dcd = [&](uint32_t te) {  // M17Demodulator.h:275-286 (+ dcd_on :244-257, dcd_off :260-265)
        if (!s.dcd_on && s.dcd_trig) {
            s.dcd_on = 1;
            if (s.st == ST_UNLOCKED) {
                s.sync_count = 0; s.missing_sync_count = 0;
                for (int k = 0; k < 92; ++k) DL.llr[k * 64 + lane] = 0;  // framer.reset()
                s.framer_idx = 0; s.framer_half = 0;
                s.dec_state = 0;                                          // decoder.reset()
                s.evm_S = 0.f;                                            // evm.reset()
            }
            s.need_clock_reset = 1;
            s.run_pos = 0;  // a new gated run starts with the next sample
        } else if (s.dcd_on && !s.dcd_trig) {
            s.st = ST_UNLOCKED;
            s.dcd_on = 0;
            snapshot_hist(te);
        }
    };
    auto dcd_update = [&](uint32_t te) {  // DataCarrierDetect::update :63-69 with the sums K3 produced
        const uint64_t k = (P.pos0 + te + 1) / TICK - 1;  // absolute index of the tick that just ended
        const float* row = tab + (size_t)(k - tick0) * 12;
        const uint32_t span = (uint32_t)(k + 1 - s.seg_start_tick);
        const int j = span > 5 ? 5 : (int)(s.seg_start_tick % 5u);
        const float l1 = row[2 * j], l2 = row[2 * j + 1];
        s.dcd_level = (float)((double)s.dcd_level * 0.8 + 0.2 * (double)(l1 / l2));
        s.dcd_trig = s.dcd_trig ? (s.dcd_level > 0.1f) : (s.dcd_level > 4.0f);
        s.seg_start_tick = (uint32_t)(k + 1);
    };
    // FIR output for relative sample tt (reference a2 with the gating of Q2)
    auto fir_out = [&](uint32_t tt) -> float {
        if (s.run_pos >= 148) return yr[tt];
        const int j = s.run_pos;  // outputs 0..147 of a run still see the previous run's tail
        float acc = 0.f;
        for (int i = 0; i < NTAPS; ++i) {
            const int k = j - i;
            const int sv = k >= 0 ? (int)xr[(int64_t)tt - i] : (int)gs->hist[149 + k];
            const float p = scale_sample(sv, invert) * P.taps[i];
            acc = acc + p;
        }
        return acc;
    };
    auto corr_sample = [&](float v) {  // Correlator::sample :43-49
        const float h0n = iir_advance(fabsf(v), s.h0, s.h1);  // history shifts: h2 <- h1, h1 <- h0
        s.h2 = s.h1; s.h1 = s.h0; s.h0 = h0n;
        ring[s.ring_pos * 64 + lane] = v;
        s.prev_pos = s.ring_pos;
        if (++s.ring_pos == 80u) s.ring_pos = 0;
        if (s.run_pos < 148) s.run_pos++;
    };
    // tail of M17Demodulator::operator() (:742-752)
    auto step_tail = [&](uint32_t te) {
        if (s.count == 960u) {
            update_dcd(te);
            s.count = 0;
            fire_diag(sqrtf(s.evm_S));
            dcd_update(te);
        }
    };

    auto do_unlocked = [&]() {  // :289-342
        if (s.missing_sync_count < 1920) {
            s.missing_sync_count += 1;
            const uint32_t si = sw_step(0);
            const int32_t up = sw_updated(0);
            if (up) {
                s.sync_count = 0; s.missing_sync_count = 0; s.need_clock_reset = 1;
                s.dev_reset = 1; s.sample_index = si; update_values(si);
                s.st = ST_LSF_SYNC;
            }
            return;
        }
        uint32_t si = sw_step(1);
        int32_t up = sw_updated(1);
        if (up) {
            s.sync_count = 86; s.missing_sync_count = 0; s.need_clock_reset = 1;
            s.dev_reset = 1; s.sample_index = si; update_values(si);
            s.st = ST_FRAME;
            s.sync_word_type = up < 0 ? 1u : 0u;
        }
        si = sw_step(2);
        up = sw_updated(2);
        if (up < 0) {
            s.sync_count = 86; s.missing_sync_count = 0; s.need_clock_reset = 1;
            s.dev_reset = 1; s.sample_index = si; update_values(si);
            s.st = ST_FRAME;
            s.sync_word_type = 3u;
        }
    };
    auto do_lsf_sync = [&]() {  // :350-411
        if (corr_index() != s.sample_index) return;
        float sync_triggered = sw_triggered(0);
        if ((double)sync_triggered > 0.1) { s.need_clock_update = 1; s.sync_count += 1; return; }
        sync_triggered = sw_triggered(1);
        const float bert_triggered = sw_triggered(2);
        if (bert_triggered < 0.f) {
            s.missing_sync_count = 0; s.sync_count = 86; s.need_clock_update = 1;
            update_values(s.sample_index); s.st = ST_FRAME; s.sync_word_type = 3u;
        } else if ((double)fabsf(sync_triggered) > 0.1) {
            s.missing_sync_count = 0; s.sync_count = 86; s.need_clock_update = 1;
            update_values(s.sample_index); s.st = ST_FRAME;
            s.sync_word_type = sync_triggered > 0.f ? 0u : 1u;
        } else if (++s.missing_sync_count > 192) {
            if (s.sync_count >= 10) { s.missing_sync_count = 0; s.need_clock_update = 1; }
            else { s.sync_count = 0; s.st = ST_UNLOCKED; s.missing_sync_count = 0; dcd_unlock(); }
        } else {
            update_values(s.sample_index);
        }
    };
    // shared shape of do_stream_sync / do_packet_sync / do_bert_sync (:420-574)
    auto do_stream_sync = [&]() {
        s.sync_count += 1;
        if (s.sync_count < 78) return;
        if (sw_triggered(3) > 0.1f) {
            s.sync_word_type = 1u; s.st = ST_FRAME; s.eot_flag = 1; s.missing_sync_count = 0;
            return;
        }
        const uint32_t si = sw_step(1) & 0xFFu;
        const int32_t up = sw_updated(1);
        if (up < 0) {
            s.missing_sync_count = 0; update_values(si);
            s.sync_word_type = 1u; s.st = ST_SYNC_WAIT; s.eot_flag = 0;
        } else if (s.sync_count > 86) {
            if (s.viterbi_cost < 80u) {
                if (!s.missing_sync_count) s.missing_sync_count = 1;
                s.sync_word_type = 1u; s.st = ST_FRAME;
            } else if (s.eot_flag) {
                s.st = ST_UNLOCKED; dcd_unlock();
            } else if (s.missing_sync_count < 10) {
                s.missing_sync_count += 1; s.sync_word_type = 1u; s.st = ST_FRAME;
            } else {
                s.st = ST_UNLOCKED; dcd_unlock();
            }
            s.eot_flag = 0;
        }
    };
    auto do_packet_sync = [&]() {
        s.sync_count += 1;
        if (s.sync_count < 78) return;
        const uint32_t si = sw_step(2) & 0xFFu;
        const int32_t up = sw_updated(2);
        if (up) {
            s.missing_sync_count = 0; update_values(si);
            s.sync_word_type = 2u; s.st = ST_SYNC_WAIT;
        } else if (s.sync_count > 86) {
            if (s.viterbi_cost < 60u) {
                if (!s.missing_sync_count) s.missing_sync_count = 1;
                s.sync_word_type = 2u; s.st = ST_FRAME;
            } else if (s.missing_sync_count < 10) {
                s.missing_sync_count += 1; s.sync_word_type = 2u; s.st = ST_FRAME;
            } else { s.st = ST_UNLOCKED; dcd_unlock(); }
        }
    };
    auto do_bert_sync = [&]() {
        s.sync_count += 1;
        if (s.sync_count < 78) return;
        const uint32_t si = sw_step(2) & 0xFFu;
        const int32_t up = sw_updated(2);
        if (up < 0) {
            s.missing_sync_count = 0; update_values(si);
            s.sync_word_type = 3u; s.st = ST_SYNC_WAIT;
        } else if (s.sync_count > 86) {
            if (s.viterbi_cost < 80u) {
                if (!s.missing_sync_count) s.missing_sync_count = 1;
                s.sync_word_type = 3u; s.st = ST_FRAME;
            } else if (s.missing_sync_count < 10) {
                s.missing_sync_count += 1; s.sync_word_type = 3u; s.st = ST_FRAME;
            } else { s.st = ST_UNLOCKED; dcd_unlock(); }
        }
    };
    auto do_sync_wait = [&]() {  // :583-593
        if (s.sync_count < 86) { s.sync_count += 1; return; }
        s.need_clock_update = 1;
        s.st = ST_FRAME;
    };
    // do_frame :596-654 up to the point where the frame buffer is full; returns true when a decode is due
    auto do_frame = [&](float filtered) -> bool {
        const int d = (int)s.sample_index - (int)corr_index();
        if (abs(d) == 5) {
            clock_update();
            s.sample_index = (uint32_t)(uint8_t)s.ck_sample_index;
            return false;
        }
        if (corr_index() != s.sample_index) return false;
        float sample = filtered - s.offset;
        sample = sample * s.idev;
        sample = sample * 1.0f;  // polarity
        {  // SymbolEvm::update :31-51, RunningStandardDeviation::capture (alpha = 1/184)
            float e;
            if (sample > 2.f) e = sample - 3.f;
            else if (sample > 0.f) e = sample - 1.f;
            else if (sample > -2.f) e = sample + 1.f;
            else e = sample + 3.f;
            const float alpha = (float)(1.0 / 184);
            s.evm_S = s.evm_S - s.evm_S * alpha;
            s.evm_S = s.evm_S + (e * e) * alpha;
        }
        // llr<float,4> (Util.h:128-145): first table edge >= clamped sample
        const float cl = fminf(3.0f, fmaxf(-3.0f, sample));
        int n = 0;
        {  // lower_bound over 43 float-accumulated edges
            int lo = 0, hi = 43;
            while (lo < hi) {
                const int mid = (lo + hi) >> 1;
                if (P.llr_edges[mid] < cl) lo = mid + 1; else hi = mid;
            }
            n = lo == 43 ? 42 : lo;
        }
        // LLR pair for table row n (Util.h:63-104): i falls 7..1,-1..-7 over rows 14..27; j as below
        int li, lj;
        if (n <= 14) { li = 7; lj = (n <= 6) ? 7 - n : ((n <= 13) ? 6 - n : -7); }
        else if (n <= 28) { lj = -7; li = (n <= 20) ? 21 - n : ((n <= 27) ? 20 - n : -7); }
        else { li = -7; lj = (n <= 34) ? n - 35 : ((n <= 41) ? n - 34 : 7); }
        const uint32_t pair = ((uint32_t)(uint8_t)(int8_t)li) | (((uint32_t)(uint8_t)(int8_t)lj) << 8);
        if ((s.framer_idx & 2u) == 0) {
            s.framer_half = pair;
        } else {
            DL.llr[(s.framer_idx >> 2) * 64 + lane] = s.framer_half | (pair << 16);
        }
        s.framer_idx += 2;
        if (s.framer_idx == 368u) {
            s.framer_idx = 0;
            s.sync_count = 0;
            return true;
        }
        return false;
    };
    // the part of do_frame after decoder(...) returns (:627-642)
    auto after_decode = [&]() {
        switch (s.dec_state) {
        case 1: s.st = ST_STREAM_SYNC; break;
        case 0: s.st = ST_STREAM_SYNC; break;
        case 4: s.st = ST_BERT_SYNC; break;
        default: s.st = ST_PACKET_SYNC; break;
        }
    };

    // one input sample: M17Demodulator::operator() :657-753.  Returns true if the lane must park for a decode.
    auto step = [&](uint32_t tt) -> bool {
        s.count++;
        if (s.initializing) {
            --s.initializing;
            const float f = fir_out(tt);
            corr_sample(f);
            s.count = 0;
            if (s.initializing == 0) snapshot_hist(tt);  // the init run ends here; the carrier is off
            return false;
        }
        if (!s.dcd_on) {
            if (s.count == 384u) {
                update_dcd(tt);
                dcd_update(tt);
                fire_diag(0.f);
                s.count = 0;
            }
            return false;
        }
        const float filtered = fir_out(tt);
        corr_sample(filtered);
        if (corr_index() == 0) {
            if (s.need_clock_reset) {
                clock_reset((float)s.sync_sample_index);
                s.need_clock_reset = 0;
                s.sample_index = s.sync_sample_index;
            } else if (s.need_clock_update) {
                clock_update_idx(s.sync_sample_index);
                s.need_clock_update = 0;
            }
        }
        s.ck_count++;
        bool decode_due = false;
        switch (s.st) {
        case ST_UNLOCKED: do_unlocked(); break;
        case ST_LSF_SYNC: do_lsf_sync(); break;
        case ST_STREAM_SYNC: do_stream_sync(); break;
        case ST_PACKET_SYNC: do_packet_sync(); break;
        case ST_BERT_SYNC: do_bert_sync(); break;
        case ST_SYNC_WAIT: do_sync_wait(); break;
        default: decode_due = do_frame(filtered); break;
        }
        if (decode_due) return true;  // tail runs after the batch decode
        step_tail(tt);
        return false;
    };

    // ---------------- main loop -----------------------------------------------------------------------------
    uint32_t wait_iters = 0;
    for (;;) {
        const bool active = (t < P.T) && !pending;
        const unsigned long long act_mask = __ballot(active);
        const unsigned long long pend_mask = __ballot(pending);
        bool run_batch = false;
        if (act_mask == 0ull) {
            if (pend_mask == 0ull) break;
            run_batch = true;
        } else if (pend_mask != 0ull) {
            // decode when no running lane is about to deliver a frame as well (locked lanes deliver one per 1920
            // samples), or when the parked lanes have waited a full frame period anyway
            const bool expected = active && s.dcd_on && (s.st != ST_UNLOCKED);
            ++wait_iters;
            if (__ballot(expected) == 0ull || wait_iters > 2200u) run_batch = true;
        }
        if (run_batch) {
            if (pending) {
                DecoderRegs D{s.dec_state, s.lich_segments, s.stale401};
                RecSink S{rec_base, P.rec_cap, nullptr, nullptr, c, pend_pos, pend_sync_type, P.overflow};
                s.viterbi_cost = decode_frame(P.tables, DL, lane, pend_sync_type, D, s.viterbi_cost, S, n_run, s.seq);
                s.dec_state = D.state; s.lich_segments = D.lich_segments; s.stale401 = D.stale401;
                after_decode();
                step_tail(t);
                ++t;
                pending = false;
            }
            wait_iters = 0;
            continue;
        }
        if (active) {
            if (step(t)) {
                pending = true;
                pend_sync_type = s.sync_word_type;
                pend_pos = P.pos0 + t;
            } else {
                ++t;
            }
        }
    }

    // ---------------- save state ------------------------------------------------------------------------------
    if (valid) {
        s.diag.demod_state = s.st;
        s.diag.n_frames = s.seq;
        gs->sc = s;
        for (int k = 0; k < 80; ++k) gs->ring[k] = ring[k * 64 + lane];
        for (int k = 0; k < 40; ++k) gs->sw_samples[k / 10][k % 10] = swsm[k * 64 + lane];
        for (int k = 0; k < 92; ++k) gs->llr[k] = DL.llr[k * 64 + lane];
        for (int k = 0; k < 8; ++k) gs->lsf[k] = DL.lsf[k * 64 + lane];
        P.rec_count[c] = n_run;
    }
}

// Standalone K4 entry points (parity API): one lane per frame.
__global__ __launch_bounds__(64) void viterbi_kernel(const int8_t* __restrict__ soft, uint32_t n_frames, int kind,
                                                     uint8_t* __restrict__ bits, int32_t* __restrict__ cost,
                                                     const DecodeTables* ident)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
    DecodeLds L;
    L.llr = lds;                 // [122][64] here: up to 488 soft bits per lane
    L.hist = lds + 122 * 64;     // [122][64]
    L.outb = L.hist + 122 * 64;  // [8][64]
    L.lsf = L.outb + 8 * 64;
    const int lane = threadIdx.x;
    const uint32_t f = blockIdx.x * 64 + lane;
    const int IN = DEC_IN[kind], OUT = DEC_OUT[kind];
    if (f < n_frames) {
        const int8_t* src = soft + (size_t)f * IN;
        for (int k = 0; k < (IN + 3) / 4; ++k) {
            uint32_t w = 0;
            for (int q = 0; q < 4; ++q)
                if (4 * k + q < IN) w |= (uint32_t)(uint8_t)src[4 * k + q] << (8 * q);
            L.llr[k * 64 + lane] = w;
        }
        int stale = llr_at(L.llr, lane, kind == 3 ? 401 : 0);  // BERT callers pass position 401 explicitly
        const uint32_t cst = viterbi_decode(ident, L, lane, kind + 4, stale);  // tables 4..7: identity source map
        cost[f] = (int32_t)cst;
        for (int n = 0; n < OUT; ++n) bits[(size_t)f * OUT + n] = (uint8_t)((byte_at(L.outb, lane, n >> 3) >> (7 - (n & 7))) & 1u);
    }
}

struct DecodeFramesParams {
    const int8_t* llr;        // [n][368]
    const uint8_t* sync_type; // [n]
    uint8_t* state_io;        // [n]
    uint8_t* lich_io;         // [n]
    uint8_t* lsf_io;          // [n][30]
    int8_t* dep401_io;        // [n]
    int64_t* cost_io;         // [n]
    FrameRec* recs;           // [n][2]
    uint8_t* nrec;            // [n]
    uint32_t n;
    const DecodeTables* tables;
    uint32_t* overflow;
};

__global__ __launch_bounds__(64) void decode_frames_kernel(DecodeFramesParams P)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
    DecodeLds L;
    L.llr = lds;                // [92][64]
    L.hist = lds + 92 * 64;     // [122][64]
    L.outb = L.hist + 122 * 64; // [8][64]
    L.lsf = L.outb + 8 * 64;    // [8][64]
    const int lane = threadIdx.x;
    const uint32_t f = blockIdx.x * 64 + lane;
    if (f >= P.n) return;
    const int8_t* src = P.llr + (size_t)f * 368;
    for (int k = 0; k < 92; ++k) {
        uint32_t w = 0;
        for (int q = 0; q < 4; ++q) w |= (uint32_t)(uint8_t)src[4 * k + q] << (8 * q);
        L.llr[k * 64 + lane] = w;
    }
    for (int k = 0; k < 8; ++k) {
        uint32_t w = 0;
        for (int q = 0; q < 4; ++q)
            if (4 * k + q < 30) w |= (uint32_t)P.lsf_io[(size_t)f * 30 + 4 * k + q] << (8 * q);
        L.lsf[k * 64 + lane] = w;
    }
    DecoderRegs D{P.state_io[f], P.lich_io[f], (int)P.dep401_io[f]};
    const int64_t cin = P.cost_io[f];
    uint32_t cost = cin < 0 ? 0xFFFFFFFFu : (uint32_t)cin;
    uint32_t n_run = 0, seq = 0;
    RecSink S{P.recs + (size_t)f * 2, 2, nullptr, nullptr, f, 0, P.sync_type[f], P.overflow};
    cost = decode_frame(P.tables, L, lane, P.sync_type[f], D, cost, S, n_run, seq);
    P.state_io[f] = (uint8_t)D.state;
    P.lich_io[f] = (uint8_t)D.lich_segments;
    P.dep401_io[f] = (int8_t)D.stale401;
    P.cost_io[f] = cost == 0xFFFFFFFFu ? (int64_t)-1 : (int64_t)cost;
    P.nrec[f] = (uint8_t)n_run;
    for (int k = 0; k < 30; ++k) P.lsf_io[(size_t)f * 30 + k] = (uint8_t)byte_at(L.lsf, lane, k);
}

}  // namespace m17
