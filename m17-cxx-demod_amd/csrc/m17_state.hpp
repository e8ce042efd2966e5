// Per-channel demodulator state (hot/cold split), kernel parameters and the out-of-line helpers shared by the
// sequential demodulator kernel (m17_wave_kernel.hpp).  Reference: M17Demodulator.h:123-217 (members),
// KalmanFilter.h, ClockRecovery.h, FreqDevEstimator.h, Correlator.h:81-114, DataCarrierDetect.h:63-73.
#pragma once

#include "m17_common.hpp"
#include "m17_decode_device.hpp"
#include "m17_frontend_kernels.hpp"

namespace m17 {

constexpr int DBG_SLOTS = 40;   // 64-bit counters per wave of the measurement build (m17hip_debug_counters)
enum : uint32_t { ST_UNLOCKED = 0, ST_LSF_SYNC, ST_STREAM_SYNC, ST_PACKET_SYNC, ST_BERT_SYNC, ST_SYNC_WAIT, ST_FRAME };

using Kal2 = core::Kalman2;  // 2-state Kalman filter (KalmanFilter.h:18-108); F, H, R, Q are constants

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
    X(int32_t, sync_count) X(int32_t, missing_sync_count) X(int32_t, initializing)
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
    Kal2 ck;                    // ClockRecovery's index filter (dt varies: the full update)
    float min_x0, min_x1, max_x0, max_x1;   // FreqDevEstimator's two level filters: state only, the covariance is the gain schedule's
    uint32_t lvl_n;             // their updates since the last reset, saturating at core::LEVEL_SCHED_LAST (core.h: level_schedule)
    uint32_t dev_reset;
    float dcd_level;
    uint32_t seg_start_tick;    // absolute tick index where the current DCD accumulation segment began
    uint32_t dec_state, lich_segments;
    int32_t stale401;
    uint32_t seq;               // frame callbacks since reset
    uint32_t n_run;             // frame callbacks in the current run
    uint32_t n_diag_run;        // diagnostic callbacks in the current run (only counted while the diagnostic log is on)
    uint32_t ev_cursor;         // deferred EVM: operations written to the channel's row of SeqParams::ev_ops since the run began
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

// What a replay needs of a channel's state at a segment boundary, written by K5 where the channel left the replay in the segment that ends
// there (a copy: the replay that takes the channel up again runs beside K5 of the next segment, which goes on changing the state itself)
struct Boundary {
    int32_t init;
    uint32_t on, trig, count;
    int32_t run_pos;
    float h0, h1, h2, level;
    uint32_t seg;
    int16_t hist[150];
};
// The TRUE gate state of a channel at the end of a segment (forced unlocks included): what the gate-aware front end forecasts from
// (gate_forecast_kernel, m17_gate_kernel.hpp).  Same fields as the replay's start state.
struct GateTruth {
    int32_t init;
    uint32_t on, trig, count;
    float level;
    uint32_t seg;
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
    const float* h;           // K2's limit-filter history (hbuf), pitch ypitch
    const float* final_h;     // [C][4] K2's filter history after the last fed sample of the run
    uint32_t* dropped;        // [C] out: this segment left K2's replay (K2 redoes the replay's state from K5's; K5 serves itself meanwhile)
    const uint32_t* dropped_in;   // [C] the same flags of the PREVIOUS segment (nullptr: first segment of a run): set = hbuf holds nothing for this channel
    Diag* diag_log;           // optional [C][diag_cap]: one entry per diagnostic callback of the run (m17hip_tune key 9), else nullptr
    uint32_t diag_cap;
    uint32_t* diag_count;     // [C] entries written this run
    uint32_t kalman_order;    // evaluation order of the Kalman update (kal_update)
    uint32_t channel_base;    // global id of channel 0 of this context (written into the frame records)
    const core::Kalman2Gain* level_gain;   // [core::LEVEL_SCHED_N] gain schedule of the level filters under `kalman_order`
    Boundary* bnd_out;        // optional [C]: boundary records for the end of this segment (m17_gate_kernel.hpp reads them)
    uint32_t* defer;          // optional [C][rec_cap][46]: LLR frames (nibbles) whose decoding is deferred to decode_deferred_kernel (nullptr: none is)
    float* ev_ops;            // optional [C][ev_pitch]: the running EVM is deferred to evm_fold_pass (below); nullptr: K5 folds it itself
    uint32_t ev_pitch;
    uint32_t* ev_cursor_out;  // [C] the channel's operation cursor at the end of this segment
    GateTruth* truth_out;     // optional [C]: the gate state at the end of this segment (gate-aware front end); overflow[3] counts the channels whose carrier is off there
};

// ---- the running EVM, deferred ------------------------------------------------------------------------------------------------------
// RunningStandardDeviation<float,184>::capture (StandardDeviation.h:60-72; SymbolEvm.h:31-51) is S <- (S - S alpha) + err^2 alpha once per
// payload symbol: three dependent instructions that nothing in the demodulator reads — the value goes into the diagnostic callback
// (M17Demodulator.h:746-750) and nowhere else.  Folded inside K5 it is 550 of a frame's ~2600 instructions on a wave that carries ONE channel;
// deferred, K5 writes the term of every symbol (and a mark where the callback wants the value / where evm.reset() falls) to the channel's
// row of operations, and a kernel with one LANE per channel folds them in the reference's order — 64 channels per instruction — and
// puts sqrt(S) where the marks say (the diagnostic log entry, the channel's m17_diag).  Operations: v >= 0 a term; EV_RESET evm.reset()
// (:255); EV_EMIT the callback's value (no log entry); v <= -2: the callback's value, into log entry (-v) - 2.
constexpr float EV_RESET = -1.0f, EV_EMIT = -1.5f, EV_NOP = -1.125f;   // (EV_NOP: inside the fold only)
constexpr uint32_t EVM_PENDING = 0x7FC0E7A1u;   // m17_diag::evm until evm_deferred_kernel has been through the run (a quiet NaN)
struct EvState { float S, last; uint32_t pos, pad; };   // per channel: RunningStandardDeviation::S, the last value a callback got (both carried from run to run), operations of the current run already folded
__device__ __host__ inline uint32_t ev_row_floats(uint32_t T) { return (T / 10u + T / 960u + T / 384u + 64u + 3u) & ~3u; }   // terms + callbacks + resets of a run of T samples

struct EvParams {
    const float* ops;     // [C][pitch] (nullptr: nothing to fold)
    uint32_t pitch;
    EvState* es;          // [C]
    Diag* diag_log;       // optional [C][diag_cap]
    uint32_t diag_cap;
    SeqState* state;      // cold.ev_cursor = operations of the run; cold.diag.evm = EVM_PENDING where the last callback waits for its value
    uint32_t C;
    const uint32_t* upto; // [C] fold the operations below this cursor (K5's at the end of a segment)
    uint32_t last;        // != 0: the last pass of a run (upto = K5's cursor at the end of its last segment): the next run's operations start at 0.  1: m17_diag is
                          // settled as well (the pass runs behind the run, nothing else at work on the state); 2: it is not — the pass rides a launch of the NEXT
                          // run, whose K5 has the state in its hands (it would put this run's value over a mark of the next run's): the value stays in
                          // EvState::last, and the pass that does settle (m17hip_diag_fetch's, or the last of a stream) finds it there
};
constexpr int EV_CPB = 16;                        // channels per workgroup (one wave) of a fold pass
constexpr int EV_TILE_FLOATS = EV_CPB * 68;       // its LDS: 16 channels x 64 operations, rows of 68 words
__host__ __device__ inline uint32_t ev_fold_blocks(uint32_t C) { return (C + EV_CPB - 1) / EV_CPB; }
// One pass for the sixteen channels [16 blk, 16 blk + 16): the operations from where the previous pass stopped up to the cursor given.
// Lane l < 16 folds channel 16 blk + l (sixteen lanes in one quarter of the wave: the dependent chain at its shortest, NOTES 4.12); all 64
// lanes load — four lanes per row, 64 operations per row and tile — and the tile goes to the folding lanes through LDS, the next tile's
// loads in flight meanwhile.  The passes of a run: segment k's operations as extra blocks of the limit-filter replay that runs beside K5
// of segment k + 2 (m17_gate_kernel.hpp), the rest behind the last K5 beside the deferred decode (m17_parity_kernels.hpp) — nothing of it
// on a stream of its own (NOTES 4.14) or in the chain of K5 launches.
__device__ __forceinline__ void evm_fold_pass(const EvParams& E, uint32_t blk, float* tile)
{
    const uint32_t l = threadIdx.x & 63u, c0 = blk * EV_CPB;
    const uint32_t c = c0 + l;                       // (l < 16) the channel this lane folds
    const bool valid = l < (uint32_t)EV_CPB && c < E.C;
    const uint32_t cc = min(c, E.C - 1u);
    EvState e = E.es[cc];
    const uint32_t from = valid ? min(e.pos, E.pitch) : 0u;
    const uint32_t upto = valid ? min(E.upto[cc], E.pitch) : 0u;   // (never the state's own cursor: by now that may be the NEXT run's)
    const uint32_t n = upto > from ? upto - from : 0u;   // operations of this lane's row in this pass
    uint32_t nmax = n;
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) nmax = max(nmax, (uint32_t)__shfl_xor((int)nmax, o));
    nmax = (uint32_t)__builtin_amdgcn_readfirstlane((int)nmax);
    // loading: lane l serves row l / 4, operations 4 i + l % 4 of the tile
    const uint32_t lr = l >> 2, lq = l & 3u;
    const float* lrow = E.ops + (size_t)min(c0 + lr, E.C - 1u) * E.pitch;
    const uint32_t lfrom = (uint32_t)__shfl((int)from, (int)lr);
    float v[16];
    auto load_tile = [&](uint32_t k0) {
#pragma unroll
        for (uint32_t i = 0; i < 16u; ++i) v[i] = lrow[min(lfrom + k0 + 4u * i + lq, E.pitch - 1u)];
    };
    auto lds_sync = [] {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront", "local");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront", "local");
    };
    if (nmax) load_tile(0);
    for (uint32_t k0 = 0; k0 < nmax; k0 += 64u) {
#pragma unroll
        for (uint32_t i = 0; i < 16u; ++i) tile[lr * 68u + 4u * i + lq] = v[i];
        lds_sync();
        if (k0 + 64u < nmax) load_tile(k0 + 64u);
        if (l < (uint32_t)EV_CPB) {
            const uint32_t m = n > k0 ? min(64u, n - k0) : 0u;       // this row's operations in the tile; beyond them: no-ops
            const uint32_t mmax = min(64u, nmax - k0);
            const float4* t4 = reinterpret_cast<const float4*>(tile + l * 68u);
            float4 cur = t4[0], nxt = t4[1];
            for (uint32_t j0 = 0; j0 < mmax; j0 += 4u) {
                const float4 g = cur;
                cur = nxt;
                nxt = t4[min(j0 / 4u + 2u, 15u)];
                const float xs[4] = {g.x, g.y, g.z, g.w};
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float x = j0 + i < m ? xs[i] : EV_NOP;
                    const float w = (e.S - e.S * core::EVM_ALPHA) + x;
                    if (__ballot(x < -1.25f) != 0ull && (x == EV_EMIT || x <= -2.f)) {   // (rare: a callback's value)
                        e.last = sqrtf(e.S);
                        if (x <= -2.f) {
                            const uint32_t idx = (uint32_t)(-x) - 2u;
                            if (E.diag_log && idx < E.diag_cap) E.diag_log[(size_t)cc * E.diag_cap + idx].evm = e.last;
                        }
                    }
                    e.S = !(x < 0.f) ? w : (x == EV_RESET ? 0.f : e.S);   // a term (>= 0, or NaN: S is NaN from there on, as in the reference); evm.reset(); a mark / no-op
                }
            }
        }
        lds_sync();
    }
    if (valid) {
        e.pos = E.last ? 0u : upto;   // (the last pass of a run: the next run's operations start at 0)
        E.es[c] = e;
        if (E.last == 1u) {
            uint32_t* w = reinterpret_cast<uint32_t*>(&E.state[c].cold.diag.evm);
            if (*w == EVM_PENDING) *w = __float_as_uint(e.last);
        }
    }
}
// (on its own where the payload decode is not deferred; otherwise as the last blocks of decode_deferred_kernel's launch, m17_parity_kernels.hpp)
__global__ __launch_bounds__(64) void evm_deferred_kernel(EvParams E)
{
    __shared__ float tile[EV_TILE_FLOATS];
    evm_fold_pass(E, blockIdx.x, tile);
}

// LDS words for a wave of `ls` channels: ring, sync samples, llr, hist, outb, lsf columns; edges, src maps, lich map

__device__ __constant__ const float SW_MAG1[4] = {29.f, 31.f, 31.f, 31.f};
__device__ __constant__ const float SW_MAG2[4] = {-3.402823466e+38f, -31.f, -31.f, -3.402823466e+38f};

// ---- Kalman pieces: core::kalman2_update (evaluation-order switch: DESIGN.md §4.4, m17hip_set_kalman_order) -------------
__device__ __forceinline__ void kal_reset(Kal2& k, float z) { core::kalman2_reset(k, z); }
// wrap != 0: KalmanFilter<float,SPS> (index filter, modulo SPS); 0: SymbolKalmanFilter.  State in LDS.
template <uint32_t ORDER>
__device__ __noinline__ void kal_update_as(M17_LDS Kal2* kp, float z, uint32_t dt_u, int wrap)
{
    Kal2 k = lds_get(kp);
    core::kalman2_update_as<ORDER>(k, z, dt_u, wrap);
    lds_put(kp, k);
}
// (one out-of-line function per order: the default one is as small as the single-order build was; a run-time branch inside
// one function cost the sequential kernel 8 VGPRs and a 32-byte spill)
__device__ __forceinline__ void kal_update(M17_LDS Kal2* kp, float z, uint32_t dt_u, int wrap, uint32_t order)
{
    if (order == 3u) { kal_update_as<3>(kp, z, dt_u, wrap); return; }
    switch (order & 7u) {
    case 0: kal_update_as<0>(kp, z, dt_u, wrap); break;
    case 1: kal_update_as<1>(kp, z, dt_u, wrap); break;
    case 2: kal_update_as<2>(kp, z, dt_u, wrap); break;
    case 4: kal_update_as<4>(kp, z, dt_u, wrap); break;
    case 5: kal_update_as<5>(kp, z, dt_u, wrap); break;
    case 6: kal_update_as<6>(kp, z, dt_u, wrap); break;
    default: kal_update_as<7>(kp, z, dt_u, wrap); break;
    }
}
using core::wrap10;
// llr<float,4> (Util.h:63-104,128-145): core::llr_slice — the (int8,int8) pair of the first float-accumulated table edge >= the
// clamped sample (edges in LDS or global)
__device__ __forceinline__ uint32_t slice_llr(float sample, const float* edges) { return core::llr_slice(sample, edges); }
// ClockRecovery::update() (ClockRecovery.h:76-88): core::clock_predict
using core::clock_predict;

// ---- out-of-line helpers on COLD state ------------------------------------------------------------------------------
// M17Demodulator::update_values (:233-241) = Correlator::outer_symbol_levels (Correlator.h:81-114) +
// FreqDevEstimator::update (FreqDevEstimator.h:31-48).  Returns (idev, offset).
// WAVE-UNIFORM ARITHMETIC ON SIXTEEN LANES.  The helpers below compute one value per channel = per wave; every lane would compute the same.
// A VALU instruction costs the same issue time from 16 enabled lanes up (2.7 x more below 16, NOTES 3.5 of round 2) but a quarter of the
// lane energy, and the matched filter this kernel shares the chip with runs at its power limit: they run under `uniform16()` and hand
// their results back through scalar registers (SReg / readfirstlane: lane 0 is among the sixteen) or through LDS.
__device__ __forceinline__ bool uniform16() { return (threadIdx.x & 63u) < 16u; }

__device__ __forceinline__ float2 nf_update_values(M17_LDS Cold* cd, const float* ring, int stride, int lane, uint32_t si, uint32_t order, const core::Kalman2Gain* gain)
{
    float2 out = make_float2(0.f, 0.f);
    if (uniform16()) {
    // the gain of this update: a function of the update count alone (core.h, level_schedule) — in flight while the levels are formed
    const uint32_t n = cd->lvl_n;
    const core::Kalman2Gain g = gain[n];
    float mn, mx;
    core::outer_symbol_levels([&](uint32_t i) { return ring[i * stride + lane]; }, si, mn, mx);
    float a0 = cd->min_x0, a1 = cd->min_x1, b0 = cd->max_x0, b1 = cd->max_x1;
    core::level_update(a0, a1, mn, g, order);
    core::level_update(b0, b1, mx, g, order);
    float offset = core::freqdev_offset(b0, a0);
    float idev = core::freqdev_idev(b0, a0);
    uint32_t rst = cd->dev_reset;
    if (isnan(a0) || isnan(a1) || isnan(b0) || isnan(b1)) rst = 1;
    uint32_t nn = min(n + 1u, (uint32_t)core::LEVEL_SCHED_LAST);
    if (rst) {
        a0 = mn; a1 = 0.f; b0 = mx; b1 = 0.f; nn = 0;
        offset = (mn + mx) / 2.f;
        idev = core::freqdev_idev(mx, mn);
    }
    cd->min_x0 = a0; cd->min_x1 = a1; cd->max_x0 = b0; cd->max_x1 = b1; cd->lvl_n = nn;
    cd->dev_reset = 0;
    out = make_float2(idev, offset);
    }
    return out;
}
struct ClockOut { float sample_est, clock_est; int32_t sample_index; };
// ClockRecovery::update(uint8_t) (ClockRecovery.h:54-67)
// KORDER >= 0: the evaluation order is a compile-time constant and the update is inlined (the production kernel of the default order: no
// call, hence no stack, in the whole kernel); KORDER < 0: the run-time order through the out-of-line variants.
template <int KORDER = -1>
__device__ __forceinline__ ClockOut nf_clock_update_idx(M17_LDS Cold* cd, uint32_t index, uint32_t ck_count, uint32_t order)
{
    ClockOut o{0.f, 0.f, 0};
    if constexpr (KORDER >= 0) {
        if (uniform16()) {
            Kal2 k = lds_get(&cd->ck);
            core::kalman2_update_as<(uint32_t)KORDER>(k, (float)index, ck_count, 10);
            lds_put(&cd->ck, k);
            o.sample_est = k.x0;
            o.sample_index = core::clock_index_of(o.sample_est);
            o.clock_est = k.x1;
        }
    } else {
        kal_update(&cd->ck, (float)index, ck_count, 10, order);   // (a call: every lane)
        o.sample_est = cd->ck.x0;
        o.sample_index = core::clock_index_of(o.sample_est);
        o.clock_est = cd->ck.x1;
    }
    return o;
}
// DataCarrierDetect::update (:63-69) with the sums K3 produced for the segment [seg_start_tick, k]; returns the trigger.
// (have: the two sums were fetched ahead, pl1 / pl2)
// (k: the tick that ends at the point, as the 32-bit tick count seg_start_tick is kept in; row: its row in the table of this run)
__device__ __forceinline__ uint32_t nf_dcd_update(M17_LDS Cold* cd, const float* tab, uint32_t row_index, uint32_t k, uint32_t trig,
                                                  bool have = false, float pl1 = 0.f, float pl2 = 0.f)
{
    uint32_t out = 0;
    if (uniform16()) {
        const float* row = tab + (size_t)row_index * 12;
        const uint32_t span = k + 1u - cd->seg_start_tick;
        const int j = span > 5 ? 5 : (int)(cd->seg_start_tick % 5u);
        float l1, l2;  // table row: [2 bins][6 sums]
        if (have) { l1 = pl1; l2 = pl2; } else { l1 = row[j]; l2 = row[6 + j]; }
        const float level = core::dcd_level(cd->dcd_level, l1, l2);
        cd->dcd_level = level;
        cd->seg_start_tick = k + 1u;
        out = trig ? (level > 0.1f) : (level > 4.0f);
    }
    return out;
}
// arguments of the diagnostic callback (M17Demodulator.h:681-685, 746-750)
__device__ __forceinline__ void nf_fire_diag(M17_LDS Cold* cd, uint32_t dcd_on, float evm_arg, float idev, float offset, uint32_t locked, float clock,
                                          uint32_t sample_index, uint32_t sync_index, int32_t clock_index, uint32_t vcost)
{
    if (uniform16()) {
        Diag d = lds_get(&cd->diag);
        d.dcd = (int32_t)dcd_on; d.evm = evm_arg; d.deviation = 2400.f / idev; d.offset = offset;
        d.locked = (int32_t)locked; d.clock = clock; d.sample_index = (int32_t)sample_index;
        d.sync_index = (int32_t)sync_index; d.clock_index = (int32_t)(uint8_t)clock_index;
        d.viterbi_cost = (int32_t)vcost; d.dcd_level = cd->dcd_level; d.n_diag++;
        lds_put(&cd->diag, d);
    }
}
// the 149 raw samples that end with sample te: the FIR history a later gated run splices in front of its own samples (lane-parallel)
__device__ __forceinline__ void nf_snapshot_hist(int16_t* hist, const int16_t* xr, uint32_t te, int lane)
{
#pragma clang loop vectorize(disable) interleave(disable) unroll(disable)
    for (int k = lane; k < 149; k += 64) hist[k] = xr[(int64_t)te - 148 + k];
}
// =====================================================================================================
}  // namespace m17
