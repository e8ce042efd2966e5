// C-ABI implementation (include/m17hip.h) over the HIP kernels.  Host side: context, device slabs,
// constant tables, launches, record compaction, per-kernel HIP-event timing.  gfx950 only.
// Product code: no fallback path of any kind — if a HIP call fails the entry point returns M17HIP_EHIP.
#include "../../include/m17hip.h"

#include "m17_common.hpp"
#include "m17_decode_device.hpp"
#include "m17_frontend_kernels.hpp"
#include "m17_state.hpp"
#include "m17_wave_kernel.hpp"
#include "m17_gate_kernel.hpp"
#include "m17_mod_kernels.hpp"
#include "m17_parity_kernels.hpp"
#include "m17_gather.hpp"

#include <hip/hip_ext.h>

#include <algorithm>
#include <cmath>
#include <complex>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <mutex>
#include <new>
#include <vector>

using namespace m17;

namespace {

enum { KT_FIR = 0, KT_DCD, KT_SEQ, KT_DEC, KT_CORR, KT_COMPACT, KT_GATE, KT_N };

struct TimedLaunch { hipEvent_t a, b; int which; };

}  // namespace

struct m17hip_ctx {
    int device = 0;
    int last_hip = 0;
    hipStream_t stream = nullptr;
    hipStream_t own_main = nullptr;   // the main stream the library created with the others (StreamSet); `stream` is this one until m17hip_set_stream names another
    int set_mode = 0;
    hipStream_t side = nullptr;        // K3 runs here, concurrently with K1 (side2) and with K2/K5 of earlier segments (stream)
    hipStream_t side2 = nullptr;       // K1 of the segments of a run
    hipStream_t side3 = nullptr;       // K2 of segment k+1 while K5 works on segment k
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    // per segment: K1 / K3 / K2 (ahead) done, K2 redo done, K5 done — one set per slab pair (consecutive staged runs alternate)
    std::vector<hipEvent_t> ev_fir_[2], ev_dcd_[2], ev_gate_[2], ev_redo_[2], ev_seq_[2];
    uint32_t front_ahead = 0;         // tuning knob 5: segments the front end (K1, K3) may run ahead of K5 (0 = unlimited, measured best)
    uint32_t maxC = 0, maxT = 0;
    size_t xpitch = 0, ypitch = 0;
    uint32_t ticks_cap = 0, rec_cap = 0, rec_cap_alloc = 0;   // rec_cap: record slots per channel and run in use (<= allocated)
    int16_t* xbuf = nullptr;
    // Streaming: a second set of the per-run slabs.  The input of the NEXT run is staged in `xstage` while the current
    // run computes; that run's front end (K1 -> yalt, K3 -> dcd_alt) may start before the current run's K2/K5 chain has ended
    // (m17hip_demod_front), and K2 / K5 of the next run then work on yalt / halt.  The pointers swap when a staged run begins.
    int16_t* xstage = nullptr;
    float* yalt = nullptr;
    float* halt = nullptr;
    float* dcd_alt = nullptr;
    bool foreign_streams[4] = {false, false, false, false};   // tools build, keys 40-43: side / side2 / side3 / copy belong to the experiment, not to the context
    hipStream_t copy = nullptr;       // host -> device copies of staged input only
    hipEvent_t ev_copy = nullptr;     // the staged copy has left its source buffer
    hipEvent_t ev_in_ready = nullptr; // the staged slab and its carried 152-sample prefix are complete
    hipEvent_t ev_end[2] = {nullptr, nullptr};   // the last run on slab pair 0 / 1 is done with its slabs
    hipEvent_t ev_mark = nullptr;     // last main-stream operation a front end must not overtake (reset)
    bool slot_used[2] = {false, false};
    bool inplace_after_run = false;   // the current input slab was overwritten in place after its last run: its data region no longer holds that run's tail
    int slot = 0;                     // slab pair the pointers xbuf / ybuf / hbuf / dcd_table name
    bool staged = false, staged_h2d = false;
    bool stage_inputs = false;        // tuning knob 16: the in-place producers write the staging slab
    void* synth_scratch = nullptr;    // symbol staging of m17hip_synth_i16
    size_t synth_bytes = 0;
    uint32_t runT = 0;                // samples of the latest run
    hipEvent_t ev_tail = nullptr;     // the latest run has carried its tails into its prefixes (K5 is done with its last segment)
    bool gate0_queued = false;        // m17hip_demod_front has queued the replay of the staged run's first segment (and the prefix copies in front of it)
    int gate0_early = 1;              // tuning knob 25: 1 = it does so
    Boundary* bnd = nullptr;          // [2][maxC] boundary records (by segment parity): K5 -> the redo of K2 (m17_state.hpp)
    uint32_t front_k1_after = 0;      // tuning knob 21: the matched filter of a staged run starts after K5 of this segment (1-based) of the run before it; 0 = at once
    uint32_t last_nseg = 0;           // segments of the latest run
    bool wave_times = false;          // tuning knob 19: K5 writes each wave's working time per segment (m17hip_debug_counters)
    uint32_t stagedC = 0, stagedT = 0;
    uint32_t slabC[2] = {0, 0}, slabT[2] = {0, 0};   // what the input slab of each pair holds (m17hip_input_alternate)
    bool front_pending = false;       // m17hip_demod_front has queued the front end of the run that must follow
    uint32_t frontC = 0, frontT = 0, front_flags = 0, front_segs = 0;
    bool front_was_staged = false;    // the run whose front end is queued works on freshly swapped slabs (prefixes still to be carried)
    uint32_t carryT = 0;              // length of the run whose tail those prefixes come from (0 = none since the reset)
    float* ybuf = nullptr;
    float* hbuf = nullptr;            // K2's limit-filter history, same pitch as ybuf
    float* final_h = nullptr;         // [2][maxC][4], by segment parity
    GateExport* gate_exp = nullptr;   // [maxC] K2's own state at the end of a segment
    uint32_t* dropped = nullptr;      // [maxC] K5: the segment dropped the speculation
    void* bert_state = nullptr;       // [maxC] BertState (tuning knob 6)
    bool bert = false;
    void* pkt_state = nullptr;        // [maxC] PacketState (tuning knob 7)
    void* pkt_recs2[2] = {nullptr, nullptr};        // [pkt_cap] PacketRec: packets completed by a run, one store per record set
    uint32_t* pkt_count2 = nullptr;   // [2]
    int pkt_fed_set = 0;              // the store m17hip_packets_feed wrote last
    uint32_t pkt_cap = 0;
    bool pkt_fed = false;
    Diag* diag_log = nullptr;         // [maxC][diag_cap] one entry per diagnostic callback of the last run (tuning knob 9)
    uint32_t* diag_count = nullptr;   // [maxC]
    uint32_t diag_cap = 0;
    uint32_t kalman_order = 3;        // evaluation order of the Kalman updates (m17hip_set_kalman_order)
    int gather_fault = 0;             // tuning knob 30 (tests): 1 = this rank's compaction fails inside the gather, 2 = the root's staging allocation fails, 3 = its word of exchange 2 is not written,
                                      // 4 = it cannot read exchange 1, 5 = it cannot read exchange 2
    uint32_t gather_timeout_ms = 120000;   // tuning knob 31: bound of every wait inside m17hip_gather_frames (0 = none)
    uint32_t channel_base = 0;        // global id of channel 0 (m17hip_set_channel_base): records carry channel_base + c
    uint32_t front_first = 0;         // tuning knob 12: segments of K1 that must be complete before the first K5 starts (0 = its own only)
    int redo_form = 0;                // tuning knob 20: the replay's redo beside K5, state only (0, default), or in front of K5 with the history stored (1)
    int dcd_form = -1;                // tuning knob 10: K3 as one wave per 32 channels (0), as the four-wave latency pipeline (1), or chosen per run (-1: the pipeline for runs queued by m17hip_demod_front)
    bool dcd_latency = false;         // what the launches of the run being queued use
    uint64_t seen_overlap = 0;        // (run registry below) the overlap count this context's previous run saw
    hipEvent_t last_end = nullptr;    // ev_end of the run queued last
    uint32_t seg_len = 48000;         // tuning knob 3: samples per K2+K5 segment of a run (0 = the whole run)
    uint32_t seg0_len = 0;            // tuning knob 4: samples of the FIRST segment (a short one starts K5 early; 0 = like the others; measured neutral)
    int64_t seg_ramp = -1;            // tuning knob 33: first segment of a ramp r, 2 r, 4 r, ... up to seg_len (0 = none; -1 = per run: AUTO_RAMP when the run is the only thing in flight)
    uint32_t ramp_now = 0;            // what the run being queued uses (a staged run: decided when its front end is queued)
    float* dcd_table = nullptr;
    DcdState* dcd_state = nullptr;
    SeqState* seq_state = nullptr;
    // Two RECORD SETS that alternate run by run (what a run writes for its consumers: record slots, counts, the deferred-frame store, the
    // run's overflow words).  The payload work of run k — deferred decode, consumers, compaction, the gather — runs on the PAYLOAD stream
    // beside the chain of run k + 1, which writes the other set; m17hip_frames_select says which run's records the fetch family names.
    struct RecSet {
        FrameRec* recs = nullptr;         // [maxC][rec_cap_alloc]
        uint32_t* rec_count = nullptr;    // [maxC]
        uint32_t* defer_llr = nullptr;    // [maxC][rec_cap_alloc][46]: LLR frames (nibbles) K5 leaves for decode_deferred_kernel (tune 15), lazily
        uint32_t* ovf = nullptr;          // -> overflow + 4 * index: [0] record overflow of the run, [1] channels that left the replay (since reset, this set's runs),
                                          //    [2] deferred EVM operations dropped (since reset), [3] channel-segments of the run that ended with the carrier off
        hipEvent_t chain = nullptr;       // the run that wrote this set has left the main stream (state settled): what its payload work waits for
        hipEvent_t done = nullptr;        // the payload stream is through with the run that wrote this set
        uint32_t C = 0, rec_cap = 0, nseg = 0;
        bool valid = false;               // holds a finished run's records in the layout (rec_cap) they were written with
        bool pending = false;             // its payload work (deferred decode, consumers) is not queued yet (flush_payload)
        bool bert = false, pkt = false;   // the consumers that were on when the run was made
    } sets[2];
    int cur = 0;                      // the set of the latest run
    uint32_t sel_back = 0;            // m17hip_frames_select: 0 = the latest run's records, 1 = the run's before it
    // The PAYLOAD stream (deferred decode, consumers, compaction, gather): the copy stream of a context that streams (it exists from the first
    // staged input on), the main stream otherwise.  Not a stream of its own: one more stream per context moved the continued-stream regime from
    // 24 to 27-37 ms whatever its place in the creation order (which streams share a hardware pipe: NOTES 4.14, 6.3).  The copy stream also
    // carries the next run's staged input and prefix copies, which must never wait for a FUTURE event: so a run's payload work is queued
    // when somebody asks for its results (or needs its record set back), not when the run is queued — flush_payload.
    bool streams() const { return copy && xstage; }   // input has been staged (stage_prepare): the copy stream is at work (a parked set may bring one along — that alone changes nothing)
    hipStream_t pay() const { return streams() ? copy : stream; }
    hipEvent_t ev_dst = nullptr;      // the caller's main-stream work on a device destination is done (a fetch of the LATEST run orders itself behind it)
    uint32_t* overflow = nullptr;     // [8]: four words per record set
    uint64_t* rec_offsets = nullptr;  // exclusive prefix of rec_count (+ total at [C]): scratch of a compaction (payload stream)
    FrameRec* compact = nullptr;      // lazily sized
    uint64_t compact_cap = 0;
    DecodeTables* tables = nullptr;
    float* taps = nullptr;
    float* taps_skew = nullptr;      // tap table of fir_rrc150_skew_kernel (fs_build_tap_table)
    int limit_form = 1;              // (tools build only, key 27) configs[1]'s limit filter: 1 = the chain relayed between two waves, 0 = round 4's one recurrence wave
    int fir_form = 1;                // (tools build, key 11) K1: 1 = skewed-pair form on a bounded grid, 0 = round 4's rolled R = 15 form, one workgroup per tile
    int gate_aware = -1;             // tuning knob 26: K1 skips what the carrier cannot be on for (1), never (0), or chosen per run from how much of the previous run's channel-segments ended with the carrier off (-1, default)
    bool gate_run = false;           // the run being queued is gate-aware
    GateTruth* truth = nullptr;      // [2][maxC] by segment parity: K5's gate state at the end of a segment
    uint32_t* first_needed = nullptr;   // [maxC] gate_forecast_kernel -> K1
    uint32_t off_segs_prev = 0, chan_segs_prev = 0;   // channel-segments of the last FETCHED run that ended with the carrier off / in all
    uint32_t fir_grid = 0;           // tuning knob 13: workgroups of K1's grid (0 = default: from the items per workgroup below)
    bool fir_latency = false;        // the run being queued is one whose chain of K5 launches decides (= it gets K3's latency form): few items per K1 workgroup
    uint32_t n_cu = 256;
    uint32_t* defer_hist = nullptr;  // [maxC][101][64]: decode_deferred_kernel's decision words (one launch at a time: payload stream)
    bool defer_decode = true;
    // the running EVM folded outside K5, one lane per channel (m17_state.hpp, evm_fold_pass; tune 17)
    bool defer_evm = true;
    float* ev_ops2[2] = {nullptr, nullptr};   // [maxC][ev_pitch] operations of a run (lazily allocated); the second one where a run begins while the last fold pass of the run before is still to come
    int ev_par = 0;                  // the buffer of the current / latest run
    const float* fold_ops = nullptr; // what the pending last fold pass works on: the run's buffer, its channels, its end-of-run cursors
    uint32_t fold_C = 0;
    const uint32_t* fold_end = nullptr;
    uint32_t ev_pitch = 0;
    uint32_t ev_pitch_override = 0;  // tuning knob 18 (tests): floats per operation row instead of ev_row_floats(maxT)
    uint32_t* ev_cur = nullptr;      // [3][maxC] K5's operation cursor at the end of a segment, by segment parity; [2]: at the end of the run's LAST segment
    bool fold_pending = false;       // the last EVM fold pass of the latest run (the operations of its last two segments) is still to be made: it rides the next
                                     // run's first limit-filter replay (which K5 of that run waits for anyway), or is made when somebody asks for m17_diag (flush_fold)
    bool fold_with_decode = false;   // (inside m17hip_demod_run) ... or the latest run's deferred decode, where that is queued behind the run at once
    EvState* ev_state = nullptr;     // [maxC]
    uint32_t seq_lds_bytes = 0; // tune 14: LDS bytes a workgroup of the sequential kernel asks for (0 = SEQ_LDS_BYTES_4 for four waves)
    float* llr_edges = nullptr;
    core::Kalman2Gain* level_gain = nullptr;   // [8 orders][LEVEL_SCHED_N] gain schedules of the level filters (core.h)
    void* scratch = nullptr;          // per-operator staging (correlator outputs, viterbi io)
    size_t scratch_bytes = 0;
    DcdCoef coef{};
    uint64_t pos = 0;          // samples consumed since reset
    uint32_t lastC = 0, lastT = 0;
    bool have_run = false;     // a run has been made since the last reset (the stream continues)
    bool uploaded = false;
    bool timing = false;
    bool profile = false;      // K5 writes per-channel tick counters (tuning knob 1)
    unsigned long long* dbg = nullptr;  // [maxC][8] diagnostic cycle counters of K5
    uint32_t dbg_waves = 0;
    std::vector<TimedLaunch> pending;
    std::vector<hipEvent_t> pool;
    double acc_ms[KT_N] = {0};
    uint64_t acc_n[KT_N] = {0};
};

struct m17hip_comm {
    ncclComm_t comm = nullptr;
    int device = 0;                   // any context on this device may gather through the communicator (one call at a time)
    int rank = 0, nranks = 1;
    int last_rccl = 0;
    uint64_t* counts_dev = nullptr;   // [2 * nranks] words of the status / count exchanges
    uint64_t* words_host = nullptr;   // pinned, [2 + 2 * nranks]: this rank's word pair on its way out, every rank's on the way in — no copy of an exchange ever
                                      // targets memory that a call which ran out of time has already given back
    uint32_t serial = 0;              // gather calls made through this communicator (every rank counts the same)
    FrameRec* gathered = nullptr;     // root: every rank's records, rank after rank
    uint64_t gathered_cap = 0;
    bool dead = false;                // an exchange did not end in time or a collective call failed: given up, every later call returns M17HIP_ECOMM
};

namespace {

#define HIPCHK(ctx, expr)                                   \
    do {                                                    \
        hipError_t e_ = (expr);                             \
        if (e_ != hipSuccess) {                             \
            (ctx)->last_hip = (int)e_;                      \
            return M17HIP_EHIP;                             \
        }                                                   \
    } while (0)

size_t round_up(size_t v, size_t m) { return (v + m - 1) / m * m; }

// Give device memory back and forget the pointer.  hipFree fails only for a pointer the runtime does not know (or a dead device): the
// pointer is dropped either way — never reused, never freed twice — and the code is kept where a context is at hand.
template <typename T>
void free_dev(T*& p, int* last_hip = nullptr)
{
    if (!p) return;
    const hipError_t e = hipFree((void*)p);
    if (e != hipSuccess && last_hip) *last_hip = (int)e;
    p = nullptr;
}

// ---- constant tables ------------------------------------------------------------------------------------------
const uint8_t DC_SEQ[46] = {0xd6, 0xb5, 0xe2, 0x30, 0x82, 0xFF, 0x84, 0x62, 0xba, 0x4e, 0x96, 0x90, 0xd8, 0x98, 0xdd, 0x5d,
                            0x0c, 0xc8, 0x52, 0x43, 0x91, 0x1d, 0xf8, 0x6e, 0x68, 0x2F, 0x35, 0xda, 0x14, 0xea, 0xcd, 0x76,
                            0x19, 0x8d, 0xd5, 0x80, 0xd1, 0x33, 0x87, 0x13, 0x57, 0x18, 0x2d, 0x29, 0x78, 0xc3};  // M17 spec / M17Randomizer.h:16-22

uint16_t frame_source(size_t deinterleaved_index)
{
    const size_t i = deinterleaved_index;
    const size_t src = (45 * i + 92 * i * i) % 368;                       // PolynomialInterleaver.h:21-24
    const bool neg = (DC_SEQ[src >> 3] >> (7 - (src & 7))) & 1;           // M17Randomizer.h:30-49
    return (uint16_t)(src | (neg ? 0x200u : 0u));
}

void build_tables(DecodeTables& t)
{
    std::memset(&t, 0, sizeof(t));
    // puncture matrices Trellis.h:17-40
    std::vector<int> p1(61), p2(12, 1), p3(8, 1);
    for (size_t i = 0, j = 2; i != 61; ++i) { if (i == j) { p1[i] = 0; j += 4; } else p1[i] = 1; }
    p2[11] = 0;
    p3[7] = 0;
    struct Lay { int out, in, first; const std::vector<int>* p; };
    const Lay lay[4] = {{488, 368, 0, &p1}, {296, 272, 96, &p2}, {420, 368, 0, &p3}, {402, 368, 0, &p2}};
    for (int k = 0; k < 4; ++k) {
        size_t index = 0, pindex = 0;
        for (int i = 0; i < 488; ++i) t.src[k][i] = 0x8000;
        int i = 0;
        for (; i != lay[k].out && index < (size_t)lay[k].in; ++i) {     // depuncture, Util.h:169-190
            if (!(*lay[k].p)[pindex++]) t.src[k][i] = 0x8000;
            else t.src[k][i] = frame_source(lay[k].first + index++);
            if (pindex == lay[k].p->size()) pindex = 0;
        }
        for (; i < lay[k].out; ++i) t.src[k][i] = 0x4000;               // never written (BERT position 401)
        for (int q = 0; q < 488; ++q) t.src[4 + k][q] = (uint16_t)q;    // identity maps for already depunctured input
    }
    for (int i = 0; i < 96; ++i) t.lich_src[i] = frame_source(i);
    // Golay syndrome -> error pattern (weight <= 3 over 23 bits): Golay24.h:131-177
    auto syndrome = [](uint32_t cw) { cw &= 0xFFFFFFu; for (int i = 0; i != 12; ++i) { if (cw & 1u) cw ^= 0xC75u; cw >>= 1; } return cw; };
    std::vector<int> seen(2048, 0);
    auto put = [&](uint32_t v) { const uint32_t s = syndrome(v) & 0x7FF; t.golay_fix[s] = v; seen[s]++; };
    put(0);
    for (int i = 0; i < 23; ++i) put(1u << i);
    for (int i = 0; i < 22; ++i) for (int j = i + 1; j < 23; ++j) put((1u << i) | (1u << j));
    for (int i = 0; i < 21; ++i) for (int j = i + 1; j < 22; ++j) for (int k = j + 1; k < 23; ++k) put((1u << i) | (1u << j) | (1u << k));
}

void build_llr_edges(float* e)  // Util.h:63-104: edges accumulated in fp32
{
    const float inc = (float)(1.0 / (double)7.0f);
    float k = (float)(-3.0 + (double)inc);
    for (int n = 0; n < 43; ++n) { e[n] = k; k = k + inc; }
}

DcdCoef build_coef()  // SlidingDFT.h:85-95
{
    const std::complex<float> j{0, 1};
    const float pi2 = (float)(M_PI * 2.0);
    auto coeff = [&](size_t f) { const float k = float(f) / float(48000); return std::exp(-j * pi2 * k); };
    const auto c0 = coeff(2400), c1 = coeff(3600);
    return DcdCoef{c0.real(), c0.imag(), c1.real(), c1.imag()};
}

// ---- timing ---------------------------------------------------------------------------------------------------
hipEvent_t get_event(m17hip_ctx* c)   // nullptr (and last_hip set) when the runtime cannot create one
{
    if (!c->pool.empty()) { hipEvent_t e = c->pool.back(); c->pool.pop_back(); return e; }
    hipEvent_t e = nullptr;
    const hipError_t r = hipEventCreate(&e);
    if (r != hipSuccess) { c->last_hip = (int)r; return nullptr; }
    return e;
}
struct Timed {  // HIP events on the stream the kernel is launched on; a launch whose events could not be had is not timed
    m17hip_ctx* c; int which; hipStream_t st; hipEvent_t a = nullptr, b = nullptr;
    Timed(m17hip_ctx* ctx, int w, hipStream_t stream) : c(ctx), which(w), st(stream)
    {
        if (!c->timing) return;
        a = get_event(c); b = get_event(c);
        if (!a || !b || hipEventRecord(a, st) != hipSuccess) {
            if (a) c->pool.push_back(a);
            if (b) c->pool.push_back(b);
            a = b = nullptr;
        }
    }
    Timed(m17hip_ctx* ctx, int w) : Timed(ctx, w, ctx->stream) {}
    ~Timed()
    {
        if (!a) return;
        if (hipEventRecord(b, st) == hipSuccess) c->pending.push_back({a, b, which});
        else { c->pool.push_back(a); c->pool.push_back(b); }
    }
};
// One kernel, timed by events BOUND to its launch (hipExtLaunchKernelGGL's start / stop events: the dispatch's own time stamps, what rocprofv3 reports) — nothing
// is put into the stream around it.  Two recorded events per launch (Timed above) cost a continued stream 0.65 ms of its 21.7 ms step (tools/stream_only.py,
// TIMING=1): for the five kernels of a run's chain that is the wrong price for being measured.
struct TimedK {
    m17hip_ctx* c; int which; hipEvent_t a = nullptr, b = nullptr;
    TimedK(m17hip_ctx* ctx, int w) : c(ctx), which(w)
    {
        if (!c->timing) return;
        a = get_event(c); b = get_event(c);
        if (!a || !b) {
            if (a) c->pool.push_back(a);
            if (b) c->pool.push_back(b);
            a = b = nullptr;
        }
    }
    ~TimedK() { if (a) c->pending.push_back({a, b, which}); }
    template <typename... P, typename... A>
    void launch(void (*kernel)(P...), dim3 grid, dim3 block, uint32_t lds, hipStream_t st, A... args)
    {   // (the arguments converted to the kernel's own parameter types: hipExtLaunchKernelGGL copies them as they come)
        if (a) hipExtLaunchKernelGGL(kernel, grid, block, lds, st, a, b, 0, static_cast<P>(args)...);
        else hipLaunchKernelGGL(kernel, grid, block, lds, st, static_cast<P>(args)...);
    }
};
// Every entry point works on the context's own device whatever the calling thread's current device is (a host that also
// drives other GPUs, torch.cuda.set_device, ...): set it for the duration of the call and put the caller's back.
struct DeviceGuard {
    int prev = -1; bool ok = true;
    explicit DeviceGuard(m17hip_ctx* c)
    {
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        if (prev != c->device) { const hipError_t e = hipSetDevice(c->device); if (e != hipSuccess) { c->last_hip = (int)e; ok = false; prev = -1; } }
        else prev = -1;
        (void)hipGetLastError();   // a stale error of an earlier call (ours or the host's) must not be blamed on this one's launches
    }
    ~DeviceGuard() { if (prev >= 0) hipSetDevice(prev); }
};
#define GUARD(ctx) DeviceGuard guard_(ctx); if (!guard_.ok) return M17HIP_EHIP

// ---- is this context the only one with a run in flight? ------------------------------------------------------------------------------
// The form of the carrier-detect kernel (m17hip_tune key 10 = -1) depends on it: with nothing else in flight on the device a step waits for
// that kernel's ten-launch chain and the four-wave latency form is 12 % faster; with other batches in flight the one-wave form is (the chip is
// full, the extra wave slots cost more than the chain saves).  A context cannot see the caller's intentions, only what happened: a run that is
// queued while another context's latest run has not finished counts as an overlap, and a context whose previous run was followed by one
// (its own launch or somebody's since) takes the process for one that overlaps batches.
// (Process-wide state: heap-allocated once and never destroyed — a context destroyed during static destruction at process exit still finds it;
//  every field of it, and every context's `last_end`, is read and written under `mu` only.)
struct RunRegistry { std::mutex mu; std::vector<m17hip_ctx*> ctxs; uint64_t overlaps = 0; };
RunRegistry& g_runs_ref() { static RunRegistry* r = new RunRegistry; return *r; }
#define g_runs (g_runs_ref())
bool runs_overlap(m17hip_ctx* c)   // at the launch of a run of c
{
    std::lock_guard<std::mutex> lk(g_runs.mu);
    bool other = false;
    for (m17hip_ctx* o : g_runs.ctxs)
        if (o != c && o->device == c->device && o->last_end && hipEventQuery(o->last_end) == hipErrorNotReady) other = true;
    (void)hipGetLastError();
    if (other) ++g_runs.overlaps;
    const bool overlapped = other || g_runs.overlaps != c->seen_overlap;
    c->seen_overlap = g_runs.overlaps;
    return overlapped;
}
void drain_timing(m17hip_ctx* c)
{
    for (auto& t : c->pending) {
        hipEventSynchronize(t.b);
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, t.a, t.b) == hipSuccess) { c->acc_ms[t.which] += ms; c->acc_n[t.which]++; }
        c->pool.push_back(t.a);
        c->pool.push_back(t.b);
    }
    c->pending.clear();
}

int ensure_scratch(m17hip_ctx* c, size_t bytes)
{
    if (bytes <= c->scratch_bytes) return M17HIP_OK;
    free_dev(c->scratch, &c->last_hip); c->scratch_bytes = 0;
    HIPCHK(c, hipMalloc(&c->scratch, bytes));
    c->scratch_bytes = bytes;
    return M17HIP_OK;
}

// ---- small device kernels owned by the host layer ----------------------------------------------------------------
// m17hip_tune key 17 changed: RunningStandardDeviation::S moves to where the new mode keeps it
__global__ void ev_move_kernel(SeqState* st, EvState* es, uint32_t C, int to_deferred)
{
    const uint32_t c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    if (to_deferred) { es[c].S = st[c].hot.evm_S; es[c].pos = 0; } else st[c].hot.evm_S = es[c].S;
}

__global__ void seq_reset_kernel(SeqState* st, DcdState* ds, EvState* es, uint32_t C)
{
    const uint32_t c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    es[c] = EvState{1.0f, 0.f, 0u, 0u};   // RunningStandardDeviation::S{1.0}
    // zero-initialised object + the constructors' values
    uint32_t* w = reinterpret_cast<uint32_t*>(st + c);
    for (size_t k = 0; k < sizeof(SeqState) / 4; ++k) w[k] = 0;
    uint32_t* d = reinterpret_cast<uint32_t*>(ds + c);
    for (size_t k = 0; k < sizeof(DcdState) / 4; ++k) d[k] = 0;
    Hot& s = st[c].hot;
    Cold& k = st[c].cold;
    s.run_pos = 148;              // the stream start is exact in ybuf (zero history)
    kal_reset(k.ck, 0.f);         // KalmanFilter() : reset(0.)
    // (the level filters: SymbolKalmanFilter() : reset(0.) = zero state, update count 0 — the zeroes above)
    k.dev_reset = 1;              // FreqDevEstimator::reset_ = true
    s.evm_S = 1.0f;               // RunningStandardDeviation::S{1.0}
    s.initializing = 1920;        // M17Demodulator.h:659 (per channel)
    s.st = ST_UNLOCKED;
}

__global__ void zero_prefix_kernel(int16_t* x, size_t xpitch, float* y, size_t ypitch, uint32_t C)
{
    const uint32_t c = blockIdx.x;
    for (int k = threadIdx.x; k < XPRE; k += blockDim.x) x[(size_t)c * xpitch + k] = 0;
    for (int k = threadIdx.x; k < YPRE; k += blockDim.x) y[(size_t)c * ypitch + k] = 0.f;
}

// carry the last XPRE / YPRE samples of a run into the prefix of the next one
__global__ void carry_tail_kernel(int16_t* x, size_t xpitch, float* y, size_t ypitch, uint32_t C, uint32_t T)
{
    const uint32_t c = blockIdx.x;
    __shared__ int16_t xs[XPRE];
    __shared__ float ys[YPRE];
    int16_t* xr = x + (size_t)c * xpitch;
    float* yr = y + (size_t)c * ypitch;
    for (int k = threadIdx.x; k < XPRE; k += blockDim.x) xs[k] = xr[(size_t)T + k];   // = row[XPRE + T - XPRE + k]
    for (int k = threadIdx.x; k < YPRE; k += blockDim.x) ys[k] = yr[(size_t)T + k];
    __syncthreads();
    for (int k = threadIdx.x; k < XPRE; k += blockDim.x) xr[k] = xs[k];
    for (int k = threadIdx.x; k < YPRE; k += blockDim.x) yr[k] = ys[k];
}

// PRBS9 receiver state per channel (Util.h:320-441), carried between runs
struct BertState {
    uint32_t lfsr, synced, sync_count, bit_count, err_count, hist_count, hist_pos, frames;
    uint32_t hist[4];   // error flags of the last 128 validated bits
};
// decode_bert (apps/m17-demod.cpp:286-304) over the BERT records of the run just finished: one lane per channel
__global__ void bert_stats_kernel(const FrameRec* recs, uint32_t rec_cap, const uint32_t* rec_count, BertState* state, uint32_t C)
{
    const uint32_t c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    BertState b = state[c];
    const uint32_t n = min(rec_count[c], rec_cap);
    for (uint32_t r = 0; r < n; ++r) {
        const uint32_t* w = reinterpret_cast<const uint32_t*>(recs + (size_t)c * rec_cap + r);
        if ((w[5] & 0xFFu) != 5u) continue;   // frame_type BERT
        ++b.frames;
        for (int i = 0; i < 197; ++i) {       // 24 bytes MSB first + the top 5 bits of byte 24
            const uint32_t byte = (w[6 + (i >> 5)] >> (8 * ((i >> 3) & 3))) & 0xFFu;
            const uint32_t bit = (byte >> (7 - (i & 7))) & 1u;
            if (!b.synced) {                  // PRBS9::syncronize
                const uint32_t res = (bit ^ (b.lfsr >> 8) ^ (b.lfsr >> 4)) & 1u;
                b.lfsr = ((b.lfsr << 1) | bit) & 0x1FFu;
                if (res) b.sync_count = 0;
                else if (++b.sync_count == 18u) {
                    b.synced = 1; b.bit_count += 18u;
                    b.hist[0] = b.hist[1] = b.hist[2] = b.hist[3] = 0; b.hist_count = 0; b.hist_pos = 0; b.sync_count = 0;
                }
            } else {                          // PRBS9::generate + count_errors
                const uint32_t g = ((b.lfsr >> 8) ^ (b.lfsr >> 4)) & 1u;
                b.lfsr = ((b.lfsr << 1) | g) & 0x1FFu;
                const uint32_t err = bit ^ g;
                b.bit_count += 1;
                const uint32_t wi = b.hist_pos >> 5, m = 1u << (b.hist_pos & 31u);
                uint32_t h = wi == 0 ? b.hist[0] : (wi == 1 ? b.hist[1] : (wi == 2 ? b.hist[2] : b.hist[3]));
                b.hist_count -= (h & m) ? 1u : 0u;
                if (err) { b.err_count += 1; b.hist_count += 1; h |= m; if (b.hist_count >= 25u) b.synced = 0; }
                else h &= ~m;
                if (wi == 0) b.hist[0] = h; else if (wi == 1) b.hist[1] = h; else if (wi == 2) b.hist[2] = h; else b.hist[3] = h;
                if (++b.hist_pos == 128u) b.hist_pos = 0;
            }
        }
    }
    state[c] = b;
}

// LinkSetupFrame::decode_callsign + type field + CRC of a batch of LSFs: one lane per frame
struct LsfInfo { char dst[10], src[10]; uint16_t type; uint8_t crc_ok; uint8_t reserved[9]; };
__global__ void lsf_info_kernel(const uint8_t* lsf, uint32_t n, LsfInfo* out)
{
    const uint32_t f = blockIdx.x * blockDim.x + threadIdx.x;
    if (f >= n) return;
    const uint8_t* b = lsf + (size_t)f * 30;
    LsfInfo o{};
    for (int which = 0; which < 2; ++which) {
        const uint8_t* e = b + 6 * which;
        char* dst = which ? o.src : o.dst;
        bool bc = true;
        uint64_t v = 0;
        for (int i = 0; i < 6; ++i) { bc = bc && e[i] == 0xFF; v = (v << 8) | e[i]; }
        if (bc) { const char t[10] = {'B', 'R', 'O', 'A', 'D', 'C', 'A', 'S', 'T', 0}; for (int i = 0; i < 10; ++i) dst[i] = t[i]; continue; }
        const char map[41] = "xABCDEFGHIJKLMNOPQRSTUVWXYZ0123456789-/.";
        int idx = 0;
        while (v) { dst[idx++] = map[v % 40u]; v /= 40u; }   // at most 9 digits below 2^48
    }
    o.type = (uint16_t)((b[12] << 8) | b[13]);
    uint8_t tmp[30];
    for (int i = 0; i < 30; ++i) tmp[i] = b[i];
    o.crc_ok = mod_crc16(tmp, 30) == 0u ? 1 : 0;
    out[f] = o;
}

// Packet reassembly per channel (apps/m17-demod.cpp:32-33,154-155,207-253), carried between runs
struct PacketState {
    uint32_t size, counter, seq_errors, frames, completed, pad[3];
    uint8_t data[832];   // 32 numbered frames x 25 bytes + a last frame of up to 25
};
struct PacketRec {       // = m17_packet_rec
    uint32_t channel, seq;
    uint64_t sample_pos;
    uint16_t size, checksum;
    uint8_t crc_ok, frames, seq_errors, reserved;
    uint8_t data[840];
};
// decode_packet over the packet records of the run just finished: one lane per channel.  An LSF callback starts a new packet
// (dump_lsf clears current_packet and the frame counter); numbered frames must arrive in order (a frame out of sequence is
// dropped, the counter stays); the frame with the EOF bit appends its last `n` bytes and closes the packet with the
// CRC-16/X.25 check (0x0f47 over contents + FCS).  Completed packets go to `out` in arrival order of the atomics;
// (channel, seq) orders them.
__global__ void packet_asm_kernel(const FrameRec* recs, uint32_t rec_cap, const uint32_t* rec_count, PacketState* state, uint32_t C,
                                  PacketRec* out, uint32_t out_cap, uint32_t* out_count, uint32_t channel_base)
{
    const uint32_t c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    PacketState* st = state + c;
    uint32_t size = st->size, counter = st->counter, seq_errors = st->seq_errors, frames = st->frames, completed = st->completed;
    const uint32_t n = min(rec_count[c], rec_cap);
    for (uint32_t r = 0; r < n; ++r) {
        const FrameRec* rec = recs + (size_t)c * rec_cap + r;
        const uint32_t type = rec->frame_type;
        if (type == 0u) { size = 0; counter = 0; seq_errors = 0; frames = 0; continue; }   // FrameType::LSF -> dump_lsf
        if (type != 3u && type != 4u) continue;
        const uint8_t* p = rec->payload;
        const uint32_t tag = p[25];
        const uint32_t num = (tag & 0x7Fu) >> 2;
        if (!(tag & 0x80u)) {
            if (num != counter) { ++seq_errors; continue; }
            ++counter; ++frames;
            for (uint32_t i = 0; i < 25u; ++i) if (size + i < 832u) st->data[size + i] = p[i];
            size = min(size + 25u, 832u);
            continue;
        }
        const uint32_t take = min(num, 25u);
        ++frames;
        for (uint32_t i = 0; i < take; ++i) if (size + i < 832u) st->data[size + i] = p[i];
        size = min(size + take, 832u);
        uint32_t crc = 0xFFFFu;
        for (uint32_t i = 0; i < size; ++i) crc = mod_crc16_x25_update(crc, st->data[i]);
        crc = ~crc & 0xFFFFu;
        const uint32_t slot = atomicAdd(out_count, 1u);
        if (slot < out_cap) {
            PacketRec* o = out + slot;
            o->channel = channel_base + c; o->seq = completed; o->sample_pos = rec->sample_pos;
            o->size = (uint16_t)size; o->checksum = (uint16_t)crc;
            o->crc_ok = crc == 0x0F47u ? 1 : 0; o->frames = (uint8_t)frames; o->seq_errors = (uint8_t)min(seq_errors, 255u); o->reserved = 0;
            for (uint32_t i = 0; i < 840u; ++i) o->data[i] = i < size ? st->data[i] : (uint8_t)0;
        }
        ++completed;
    }
    st->size = size; st->counter = counter; st->seq_errors = seq_errors; st->frames = frames; st->completed = completed;
}
__global__ void packet_reset_kernel(PacketState* state, uint32_t C)
{
    const uint32_t c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    state[c].size = 0; state[c].counter = 0; state[c].seq_errors = 0; state[c].frames = 0; state[c].completed = 0;
}

__global__ void bert_reset_kernel(BertState* state, uint32_t C)
{
    const uint32_t c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    BertState b{};
    b.lfsr = 1;
    state[c] = b;
}

__global__ void copy_prefix_i16_kernel(const int16_t* src, int16_t* dst, size_t xpitch)
{
    for (int k = threadIdx.x; k < XPRE; k += blockDim.x) dst[(size_t)blockIdx.x * xpitch + k] = src[(size_t)blockIdx.x * xpitch + k];
}

// staged runs: the other slab's prefix = the last XPRE samples of the previous run's input (read where they lie: row[T .. T + XPRE))
__global__ void copy_tail_i16_kernel(const int16_t* src, int16_t* dst, size_t xpitch, uint32_t T)
{
    for (int k = threadIdx.x; k < XPRE; k += blockDim.x) dst[(size_t)blockIdx.x * xpitch + k] = src[(size_t)blockIdx.x * xpitch + T + k];
}
__global__ void copy_prefix_f32_kernel(const float* src, float* dst, size_t ypitch)
{
    for (int k = threadIdx.x; k < YPRE; k += blockDim.x) dst[(size_t)blockIdx.x * ypitch + k] = src[(size_t)blockIdx.x * ypitch + k];
}

__global__ void carry_tail_f32_kernel(float* y, size_t ypitch, uint32_t T)
{
    __shared__ float ys[YPRE];
    float* yr = y + (size_t)blockIdx.x * ypitch;
    for (int k = threadIdx.x; k < YPRE; k += blockDim.x) ys[k] = yr[(size_t)T + k];
    __syncthreads();
    for (int k = threadIdx.x; k < YPRE; k += blockDim.x) yr[k] = ys[k];
}

__global__ void copy_rows_i16_kernel(const int16_t* src, size_t spitch, int16_t* dst, size_t dpitch, uint32_t T)
{
    const uint32_t c = blockIdx.y;
    const uint32_t t = (blockIdx.x * blockDim.x + threadIdx.x) * 8;
    if (t >= T) return;
    const int16_t* s = src + (size_t)c * spitch + t;
    int16_t* d = dst + (size_t)c * dpitch + XPRE + t;
    if (t + 8 <= T && (((uintptr_t)s) & 15) == 0) {
        *reinterpret_cast<int4*>(d) = *reinterpret_cast<const int4*>(s);
    } else {
        for (uint32_t q = 0; q < 8 && t + q < T; ++q) d[q] = s[q];
    }
}

// exclusive prefix sum of the per-channel record counts (single block; C <= a few 100k)
__global__ void rec_offsets_kernel(const uint32_t* counts, uint32_t rec_cap, uint64_t* offsets, uint32_t C)
{
    __shared__ uint64_t partial[256];
    const uint32_t tid = threadIdx.x;
    const uint32_t per = (C + 255) / 256;
    const uint32_t lo = tid * per, hi = min(C, lo + per);
    uint64_t sum = 0;
    for (uint32_t c = lo; c < hi; ++c) sum += min(counts[c], rec_cap);
    partial[tid] = sum;
    __syncthreads();
    if (tid == 0) {
        uint64_t run = 0;
        for (int k = 0; k < 256; ++k) { const uint64_t v = partial[k]; partial[k] = run; run += v; }
        offsets[C] = run;
    }
    __syncthreads();
    uint64_t run = partial[tid];
    for (uint32_t c = lo; c < hi; ++c) { offsets[c] = run; run += min(counts[c], rec_cap); }
}

__global__ void compact_kernel(const FrameRec* recs, uint32_t rec_cap, const uint32_t* counts, const uint64_t* offsets,
                               FrameRec* out, uint64_t out_cap, uint32_t C)
{
    const uint32_t c = blockIdx.x;
    const uint32_t n = min(counts[c], rec_cap);
    const uint4* src = reinterpret_cast<const uint4*>(recs + (size_t)c * rec_cap);
    uint4* dst = reinterpret_cast<uint4*>(out + offsets[c]);
    for (uint32_t k = threadIdx.x; k < n * 4; k += blockDim.x)
        if (offsets[c] + k / 4 < out_cap) dst[k] = src[k];
}

// t0: first sample of the slab to process (segment of a run); the kernels see the slab from there on
constexpr size_t SEQ_LDS_BYTES_4 = 34816;   // see the K5 launch
// K1's grid.  A workgroup loops over (channel, tile) items; how many it takes decides how long it holds its place on a CU, i.e. how long a K5
// launch that becomes ready waits for room: 38 items per workgroup (five workgroups per CU) cost a continued stream 1.5 ms per step and one batch
// at a time 2 ms against 3 items per workgroup, while two batches in flight are 2 % faster with 6-12 than with 3 (NOTES 5.7).  So: about three
// items per workgroup for the runs that get the latency form of K3 (a continued stream, or nothing else in flight), eight for the others, and
// never fewer workgroups than a CU can hold of them five times over.
constexpr uint32_t FIR_GRID_PER_CU = 5, FIR_ITEMS_LATENCY = 3, FIR_ITEMS_THROUGHPUT = 8;
int launch_fir(m17hip_ctx* c, uint32_t C, uint32_t T, uint32_t flags, hipStream_t st, uint32_t t0 = 0, const uint32_t* first_needed = nullptr)
{
    TimedK tm(c, KT_FIR);
#ifdef M17_TOOLS
    if (c->fir_form == 0) {   // round 4's kernel: the measurement build keeps it for same-box comparisons (tools/k1_forms.py, the clk_k1 pass of tools/profile_round.sh)
        dim3 grid((T + FIR_TILE - 1) / FIR_TILE, C);
        tm.launch((fir_rrc150_rolled_kernel<FIR_R, 4>), grid, dim3(FIR_THREADS), 0, st, c->xbuf + t0, c->xpitch, c->ybuf + t0, c->ypitch, T, flags, c->taps);
    } else
#endif
    {
        const uint32_t tiles = (T + FS_TILE - 1) / FS_TILE;
        const uint64_t items64 = (uint64_t)tiles * C;
        if (items64 > 0xFFFFFFFFull) return M17HIP_EINVAL;
        const uint32_t items = (uint32_t)items64;
        const uint32_t per = c->fir_latency ? FIR_ITEMS_LATENCY : FIR_ITEMS_THROUGHPUT;
        const uint32_t cap = c->fir_grid ? c->fir_grid : std::max(FIR_GRID_PER_CU * c->n_cu, (items + per - 1) / per);
        const dim3 grid(std::min(items, cap));
        if (flags & 1u)
            tm.launch(fir_rrc150_skew_kernel<true>, grid, dim3(FS_THREADS), 0, st, c->xbuf + t0, c->xpitch, c->ybuf + t0, c->ypitch, T, c->taps_skew, tiles, items, first_needed);
        else
            tm.launch(fir_rrc150_skew_kernel<false>, grid, dim3(FS_THREADS), 0, st, c->xbuf + t0, c->xpitch, c->ybuf + t0, c->ypitch, T, c->taps_skew, tiles, items, first_needed);
    }
    HIPCHK(c, hipGetLastError());
    return M17HIP_OK;
}
int launch_dcd(m17hip_ctx* c, uint32_t C, uint32_t T, uint32_t flags, hipStream_t st, uint32_t t0 = 0)
{
    TimedK tm(c, KT_DCD);
    // table rows are numbered from the first tick of the RUN: a later segment continues where the previous one stopped
    const uint64_t row0 = (c->pos + t0) / TICK - c->pos / TICK;
    // (the pipeline needs whole 32-sample blocks that start on a block boundary of the stream: ragged pieces take the one-wave form)
    if (!c->dcd_latency || T % DP_BLK != 0 || (c->pos + t0) % DP_BLK != 0 || t0 % 8 != 0 || T < 4 * DP_BLK)
        tm.launch(dcd_kernel, dim3((C + DCD_CPW * DCD_WPB - 1) / (DCD_CPW * DCD_WPB)), dim3(64 * DCD_WPB), 0, st, c->xbuf + t0, c->xpitch, c->dcd_state,
                           c->dcd_table + row0 * 12, c->ticks_cap, C, T, c->pos + t0, c->coef, flags);
    else if (flags & 1u)
        tm.launch(dcd_pipe_kernel<true>, dim3((C + DP_CPB - 1) / DP_CPB), dim3(256), 0, st, c->xbuf + t0, c->xpitch, c->dcd_state,
                           c->dcd_table + row0 * 12, c->ticks_cap, C, T, c->pos + t0, c->coef, flags);
    else
        tm.launch(dcd_pipe_kernel<false>, dim3((C + DP_CPB - 1) / DP_CPB), dim3(256), 0, st, c->xbuf + t0, c->xpitch, c->dcd_state,
                           c->dcd_table + row0 * 12, c->ticks_cap, C, T, c->pos + t0, c->coef, flags);
    HIPCHK(c, hipGetLastError());
    return M17HIP_OK;
}

}  // namespace

extern "C" {

const char* m17hip_strerror(int code)
{
    switch (code) {
    case M17HIP_OK: return "ok";
    case M17HIP_EINVAL: return "invalid argument";
    case M17HIP_EHIP: return "HIP runtime error";
    case M17HIP_ENOMEM: return "out of memory";
    case M17HIP_ESTATE: return "call sequence error";
    case M17HIP_EOVERFLOW: return "frame record buffer overflow";
    case M17HIP_ETRUNC: return "output truncated to the caller's capacity";
    case M17HIP_ECOMM: return "RCCL communication error";
    case M17HIP_ECONFIG: return "GPU_MAX_HW_QUEUES is unset or below 8: a context's streams would share hardware queues and serialise (export GPU_MAX_HW_QUEUES=16 before the process's first HIP call, or M17HIP_FEW_HW_QUEUES_OK=1 to go on regardless)";
    default: return "unknown error";
    }
}
int m17hip_last_hip_error(const m17hip_ctx* ctx) { return ctx ? ctx->last_hip : 0; }
// The streams of a context — main, carrier-detect (K3), matched filter (K1), replay (K2 ahead / redo), and the copy / payload stream once something is staged —
// are created by the library in ONE go in a fixed role order, and when a context goes they are parked as a SET: the next context of that device gets the same
// streams in the same roles.  Which role streams share a hardware pipe decides 10-40 % of a continued stream's step time (NOTES 4.14, 5.3, 5.10), the runtime maps
// streams to pipes in creation order, and a process that creates and destroys contexts beside a host's own streams walks through the bad layouts
// (tools/stream_history.py: 24.3 -> 26.2 -> 31.8 ms per step over three cycles with per-context streams and a host main stream; 22.1-22.5 in every cycle this way).
// M17HIP_STREAM_SETS=0 in the environment restores per-context streams on the default stream (tools/stream_history.py's A/B).
struct StreamSet { hipStream_t main, side, side2, side3, copy; int device; };
// (heap-allocated once and never destroyed, like the run registry: a context destroyed during static destruction at process exit still finds it)
struct StreamSets { std::mutex mu; std::vector<StreamSet> parked; };
static StreamSets& g_sets_ref() { static StreamSets* r = new StreamSets; return *r; }
#define g_sets (g_sets_ref())
static int stream_set_mode()
{
    const char* e = getenv("M17HIP_STREAM_SETS");
    return (e && atoi(e) == 0) ? 0 : 1;
}

static int hw_queues_env()   // what the process asked the HIP runtime for (the runtime's default is 4)
{
    const char* q = std::getenv("GPU_MAX_HW_QUEUES");
    return q ? std::atoi(q) : 4;
}
int m17hip_advice(const m17hip_ctx* ctx)
{
    if (!ctx) return 0;
    const int n = hw_queues_env();
    return (n < 8 ? M17HIP_ADVICE_HW_QUEUES : 0) | (n < 16 ? M17HIP_ADVICE_HW_QUEUES_16 : 0);
}
int m17hip_version(void) { return 601; }

int m17hip_ctx_create(int device, uint32_t max_channels, uint32_t max_samples, m17hip_ctx** out)
{
    if (!out || max_channels == 0 || max_samples == 0 || max_samples > M17HIP_MAX_SAMPLES_PER_RUN) return M17HIP_EINVAL;   // (argument checks: before any HIP call)
    // The slow state is not entered silently: with the runtime's default of four hardware queues the five streams of a context share queues,
    // a kernel queued behind another stream's event wait waits with it, and a step takes 1.4-1.6 x as long (NOTES 4.14, 5.3).
    if (hw_queues_env() < 8) {
        const char* ok = std::getenv("M17HIP_FEW_HW_QUEUES_OK");
        if (!ok || std::atoi(ok) == 0) return M17HIP_ECONFIG;
    }
    m17hip_ctx* c = new (std::nothrow) m17hip_ctx();
    if (!c) return M17HIP_ENOMEM;
    c->device = device;
    c->maxC = max_channels;
    c->maxT = max_samples;
    auto fail = [&](int code) { m17hip_ctx_destroy(c); return code; };
    DeviceGuard guard_(c);
    if (!guard_.ok) return fail(M17HIP_EHIP);
    c->xpitch = round_up((size_t)XPRE + max_samples + 8, 8);
    c->ypitch = round_up((size_t)YPRE + max_samples + 4, 4);
    // K2 stores the limit-filter history of a workgroup's GT_CPW channel rows through ONE buffer descriptor with 32-bit byte offsets
    // (m17_gate_kernel.hpp, limit_track_pass): the rows of a workgroup must fit below the offset that stands for "no store"
    static_assert(M17HIP_MAX_SAMPLES_PER_RUN == (0x7FFF0000u / (4u * GT_CPW)) - 256u, "include/m17hip.h: M17HIP_MAX_SAMPLES_PER_RUN");
    if ((size_t)GT_CPW * c->ypitch * 4u > (size_t)0x7FFF0000u) return fail(M17HIP_EINVAL);
    c->ticks_cap = max_samples / TICK + 2;
    c->rec_cap = c->rec_cap_alloc = 2 * (max_samples / 1920 + 2) + 4;  // <= 2 callbacks per 1920-sample frame
    const size_t C = max_channels;
#define ALLOC(ptr, bytes)                                                            \
    do {                                                                             \
        hipError_t e_ = hipMalloc((void**)&(ptr), (bytes));                          \
        if (e_ != hipSuccess) { c->last_hip = (int)e_; return fail(e_ == hipErrorOutOfMemory ? M17HIP_ENOMEM : M17HIP_EHIP); } \
    } while (0)
    ALLOC(c->xbuf, C * c->xpitch * sizeof(int16_t));
    ALLOC(c->ybuf, C * c->ypitch * sizeof(float));
    // (a gate-aware K1 leaves tiles unwritten: what K2's replay may read there must be finite.  A memset is queued, not done, when it returns: wait —
    //  the first run's K1 is on a stream of its own that would not)
    if (hipMemset(c->ybuf, 0, C * c->ypitch * sizeof(float)) != hipSuccess || hipDeviceSynchronize() != hipSuccess) return fail(M17HIP_EHIP);
    ALLOC(c->hbuf, C * c->ypitch * sizeof(float));
    ALLOC(c->final_h, 2 * C * 4 * sizeof(float));
    ALLOC(c->gate_exp, C * sizeof(GateExport));
    ALLOC(c->dropped, 2 * C * sizeof(uint32_t));
    ALLOC(c->bert_state, C * sizeof(BertState));
    hipLaunchKernelGGL(bert_reset_kernel, dim3((unsigned)((C + 63) / 64)), dim3(64), 0, 0, (BertState*)c->bert_state, (uint32_t)C);
    if (hipGetLastError() != hipSuccess) return fail(M17HIP_EHIP);
    ALLOC(c->dcd_table, C * c->ticks_cap * 12 * sizeof(float));
    ALLOC(c->dcd_state, C * sizeof(DcdState));
    ALLOC(c->seq_state, C * sizeof(SeqState));
    ALLOC(c->ev_state, C * sizeof(EvState));
    ALLOC(c->ev_cur, 4 * C * sizeof(uint32_t));   // [0], [1]: by segment parity; [2], [3]: at the end of a run, by the run's buffer
    ALLOC(c->sets[0].recs, C * c->rec_cap * sizeof(FrameRec));      // (the second set: with the second run, ensure_set)
    ALLOC(c->sets[0].rec_count, C * sizeof(uint32_t));
    ALLOC(c->rec_offsets, (C + 1) * sizeof(uint64_t));
    ALLOC(c->overflow, 8 * sizeof(uint32_t));
    c->sets[0].ovf = c->overflow; c->sets[1].ovf = c->overflow + 4;
    ALLOC(c->tables, sizeof(DecodeTables));
    ALLOC(c->taps, 160 * sizeof(float));
    ALLOC(c->taps_skew, FS_NBODY * FS_TAB * sizeof(float));
    ALLOC(c->llr_edges, 64 * sizeof(float));
    ALLOC(c->level_gain, 8 * (size_t)core::LEVEL_SCHED_N * sizeof(core::Kalman2Gain));
    ALLOC(c->dbg, (C + 1) * DBG_SLOTS * sizeof(unsigned long long));   // (tools build: per-wave counters of K5)
#undef ALLOC
    {
        DecodeTables* t = new DecodeTables;
        build_tables(*t);
        hipError_t e = hipMemcpy(c->tables, t, sizeof(DecodeTables), hipMemcpyHostToDevice);
        delete t;
        if (e != hipSuccess) { c->last_hip = (int)e; return fail(M17HIP_EHIP); }
        float taps[160] = {0};
        for (int i = 0; i < NTAPS; ++i) taps[i] = rrc_tap(i);
        float edges[64] = {0};
        build_llr_edges(edges);
        if (hipMemcpy(c->taps, taps, sizeof(taps), hipMemcpyHostToDevice) != hipSuccess) return fail(M17HIP_EHIP);
        float skew[FS_NBODY * FS_TAB];
        fs_build_tap_table(skew);
        if (hipMemcpy(c->taps_skew, skew, sizeof(skew), hipMemcpyHostToDevice) != hipSuccess) return fail(M17HIP_EHIP);
        int ncu = 0;
        if (hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, device) != hipSuccess || ncu <= 0) return fail(M17HIP_EHIP);
        c->n_cu = (uint32_t)ncu;
        if (hipMemcpy(c->llr_edges, edges, sizeof(edges), hipMemcpyHostToDevice) != hipSuccess) return fail(M17HIP_EHIP);
        std::vector<core::Kalman2Gain> sched(8 * (size_t)core::LEVEL_SCHED_N);
        for (uint32_t o = 0; o < 8; ++o)   // (false = the covariance has not reached its fixed point inside the table: the scheduled update would be wrong)
            if (!core::level_schedule(sched.data() + (size_t)o * core::LEVEL_SCHED_N, o)) return fail(M17HIP_ESTATE);
        if (hipMemcpy(c->level_gain, sched.data(), sched.size() * sizeof(core::Kalman2Gain), hipMemcpyHostToDevice) != hipSuccess) return fail(M17HIP_EHIP);
    }
    c->coef = build_coef();
    c->set_mode = stream_set_mode();
    bool parked = false;
    if (c->set_mode) {
        std::lock_guard<std::mutex> lk(g_sets.mu);
        for (size_t i = g_sets.parked.size(); i-- > 0;)   // (the set parked last)
            if (g_sets.parked[i].device == device) {
                const StreamSet& ss = g_sets.parked[i];
                c->own_main = ss.main; c->side = ss.side; c->side2 = ss.side2; c->side3 = ss.side3; c->copy = ss.copy;
                g_sets.parked.erase(g_sets.parked.begin() + (long)i);
                parked = true;
                break;
            }
    }
    if (!parked) {
        // (the copy stream joins the set when something is staged — stage_prepare: created here with the others, two batches measured 21.5 against 21.1 ms)
        // (the ORDER of the four inside the set does not matter: all 24 measured, two batches 20.3-20.6, continued stream 21.3-21.7 ms on one box — NOTES 6.6)
        int least = 0, greatest = 0;
        if (hipDeviceGetStreamPriorityRange(&least, &greatest) != hipSuccess) return fail(M17HIP_EHIP);
        auto make_set = [&](StreamSet& ss) -> bool {
            ss = StreamSet{nullptr, nullptr, nullptr, nullptr, nullptr, device};
            if (c->set_mode && hipStreamCreateWithFlags(&ss.main, hipStreamNonBlocking) != hipSuccess) return false;
            if (hipStreamCreateWithFlags(&ss.side, hipStreamNonBlocking) != hipSuccess) return false;
            if (hipStreamCreateWithFlags(&ss.side2, hipStreamNonBlocking) != hipSuccess) return false;
            // the replay stream outranks the others: its few workgroups must not queue behind K5's thousand.  (K1's stream at the LOWEST priority
            // was tried: a continued stream of 2 x 2048 channels 24.6 -> 22.8 ms in a process with history, but 2 x 1024 channels 11.5 -> 16.1 ms —
            // K1 starves behind the other group's kernels: NOTES 5.10.)
            return hipStreamCreateWithPriority(&ss.side3, hipStreamNonBlocking, greatest) == hipSuccess;
        };
        StreamSet mine;
        if (!make_set(mine)) return fail(M17HIP_EHIP);
        c->own_main = mine.main; c->side = mine.side; c->side2 = mine.side2; c->side3 = mine.side3;
        // (creating the sets two at a time — a spare right behind every new set, so that the sets of two contexts sit side by side whatever the host creates in
        //  between — was measured: two batches 20.6-20.9 ms in all four layouts of tools/bisect_bench.py against 20.7 / 21.6 / 20.7 / 21.3; but four more queues per
        //  process: four rank processes on one GPU, 32 + queues in all, ran tests/test_gpu_gather_ranks.py twice as long.  Not kept: NOTES 6.6)
    }
    if (c->own_main) c->stream = c->own_main;
    if (hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming) != hipSuccess) return fail(M17HIP_EHIP);
    if (hipEventCreateWithFlags(&c->ev_join, hipEventDisableTiming) != hipSuccess) return fail(M17HIP_EHIP);
    if (hipEventCreateWithFlags(&c->ev_mark, hipEventDisableTiming) != hipSuccess) return fail(M17HIP_EHIP);
    if (hipEventCreateWithFlags(&c->ev_tail, hipEventDisableTiming) != hipSuccess) return fail(M17HIP_EHIP);
    for (int q = 0; q < 2; ++q)
        if (hipEventCreateWithFlags(&c->ev_end[q], hipEventDisableTiming) != hipSuccess) return fail(M17HIP_EHIP);
    if (hipFuncSetAttribute((const void*)viterbi_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (122 + 122 + 16) * 64 * 4) != hipSuccess)
        return fail(M17HIP_EHIP);
    if (hipFuncSetAttribute((const void*)decode_deferred_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, DEFER_LDS_BYTES) != hipSuccess)
        return fail(M17HIP_EHIP);
    if (hipFuncSetAttribute((const void*)decode_frames_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (92 + 122 + 16) * 64 * 4) != hipSuccess)
        return fail(M17HIP_EHIP);
    if (hipMemset(c->overflow, 0, 32) != hipSuccess || hipDeviceSynchronize() != hipSuccess) return fail(M17HIP_EHIP);   // (default-stream work of the creation is through before the context's own streams start)
    for (hipEvent_t* e : {&c->sets[0].done, &c->sets[1].done, &c->sets[0].chain, &c->sets[1].chain, &c->ev_dst})
        if (hipEventCreateWithFlags(e, hipEventDisableTiming) != hipSuccess) return fail(M17HIP_EHIP);
    { std::lock_guard<std::mutex> lk(g_runs.mu); c->seen_overlap = g_runs.overlaps; g_runs.ctxs.push_back(c); }   // (a new context has seen no overlap yet)
    *out = c;
    const int r = m17hip_demod_reset(c);
    if (r != M17HIP_OK) { *out = nullptr; return fail(r); }
    return M17HIP_OK;
}

void m17hip_ctx_destroy(m17hip_ctx* c)
{
    if (!c) return;
    {
        std::lock_guard<std::mutex> lk(g_runs.mu);
        for (size_t i = 0; i < g_runs.ctxs.size(); ++i)
            if (g_runs.ctxs[i] == c) { g_runs.ctxs.erase(g_runs.ctxs.begin() + (long)i); break; }
    }
    DeviceGuard guard_(c);
    drain_timing(c);
    for (auto e : c->pool) hipEventDestroy(e);
    if (c->ev_fork) hipEventDestroy(c->ev_fork);
    if (c->ev_join) hipEventDestroy(c->ev_join);
    const bool any_foreign = c->foreign_streams[0] || c->foreign_streams[1] || c->foreign_streams[2] || c->foreign_streams[3];
    if (c->set_mode && !any_foreign && c->own_main && c->side && c->side2 && c->side3) {
        // the set is parked as a SET: the next context of this device gets the same five streams in the same roles
        for (hipStream_t st : {c->own_main, c->side, c->side2, c->side3, c->copy})
            if (st) (void)hipStreamSynchronize(st);
        std::lock_guard<std::mutex> lk(g_sets.mu);
        g_sets.parked.push_back(StreamSet{c->own_main, c->side, c->side2, c->side3, c->copy, c->device});
    } else {
        if (c->own_main) hipStreamDestroy(c->own_main);
        if (c->side && !c->foreign_streams[0]) hipStreamDestroy(c->side);
        if (c->side2 && !c->foreign_streams[1]) hipStreamDestroy(c->side2);
        if (c->side3 && !c->foreign_streams[2]) hipStreamDestroy(c->side3);
        if (c->copy && !c->foreign_streams[3]) hipStreamDestroy(c->copy);
    }
    if (c->ev_dst) hipEventDestroy(c->ev_dst);
    for (hipEvent_t e : {c->ev_copy, c->ev_in_ready, c->ev_end[0], c->ev_end[1], c->ev_mark, c->ev_tail, c->sets[0].done, c->sets[1].done, c->sets[0].chain, c->sets[1].chain})
        if (e) hipEventDestroy(e);
    for (int q = 0; q < 2; ++q)
        for (auto* v : {&c->ev_fir_[q], &c->ev_dcd_[q], &c->ev_gate_[q], &c->ev_redo_[q], &c->ev_seq_[q]})
            for (auto e : *v) hipEventDestroy(e);
    void* ptrs[] = {c->xbuf, c->ybuf, c->dcd_table, c->dcd_state, c->seq_state, c->sets[0].recs, c->sets[0].rec_count, c->sets[0].defer_llr,
                    c->sets[1].recs, c->sets[1].rec_count, c->sets[1].defer_llr, c->rec_offsets,
                    c->overflow, c->tables, c->taps, c->taps_skew, c->llr_edges, c->level_gain, c->compact, c->scratch, c->dbg, c->hbuf, c->final_h, c->gate_exp, c->dropped, c->bert_state, c->xstage, c->pkt_state, c->pkt_recs2[0], c->pkt_recs2[1], c->pkt_count2, c->diag_log, c->diag_count, c->defer_hist,
                    c->yalt, c->halt, c->dcd_alt, c->synth_scratch, c->bnd, c->ev_ops2[0], c->ev_ops2[1], c->ev_cur, c->ev_state, c->truth, c->first_needed};
    for (void* p : ptrs)
        if (p) (void)hipFree(p);   // (the context is going away: nothing to report to)
    delete c;
}

int m17hip_get_stream(m17hip_ctx* c, void** hip_stream)
{
    if (!c || !hip_stream) return M17HIP_EINVAL;
    *hip_stream = (void*)c->stream;
    return M17HIP_OK;
}

int m17hip_set_stream(m17hip_ctx* c, void* hip_stream)
{
    if (!c) return M17HIP_EINVAL;
    GUARD(c);
    c->stream = (hipStream_t)hip_stream;
    return M17HIP_OK;
}

// ---- staged input (streaming) -------------------------------------------------------------------------------------------
// Second slab pair, streams and events: allocated the first time input is staged.
static int stage_prepare(m17hip_ctx* c)
{
    if (!c->xstage) {
        auto alloc = [&](void** p, size_t bytes) -> int {
            const hipError_t e = hipMalloc(p, bytes);
            if (e != hipSuccess) { c->last_hip = (int)e; return e == hipErrorOutOfMemory ? M17HIP_ENOMEM : M17HIP_EHIP; }
            return M17HIP_OK;
        };
        int r;   // (a call that failed half way is picked up where it stopped: nothing is allocated twice)
        bool fresh_y = false;
        if (!c->yalt) {
            if ((r = alloc((void**)&c->yalt, (size_t)c->maxC * c->ypitch * sizeof(float)))) return r;
            fresh_y = true;
        }
        if (!c->halt && (r = alloc((void**)&c->halt, (size_t)c->maxC * c->ypitch * sizeof(float)))) return r;
        if (!c->dcd_alt && (r = alloc((void**)&c->dcd_alt, (size_t)c->maxC * c->ticks_cap * 12 * sizeof(float)))) return r;
        if (!c->copy) HIPCHK(c, hipStreamCreateWithFlags(&c->copy, hipStreamNonBlocking));
        // (zeroed as ybuf is — ON the copy stream: the staged run's K1 waits for that stream's ev_in_ready, a memset on the default stream would
        //  sit behind the run in flight and land in the middle of the staged one)
        if (fresh_y) HIPCHK(c, hipMemsetAsync(c->yalt, 0, (size_t)c->maxC * c->ypitch * sizeof(float), c->copy));
        if (!c->ev_copy) HIPCHK(c, hipEventCreateWithFlags(&c->ev_copy, hipEventDisableTiming));
        if (!c->ev_in_ready) HIPCHK(c, hipEventCreateWithFlags(&c->ev_in_ready, hipEventDisableTiming));
        if ((r = alloc((void**)&c->xstage, (size_t)c->maxC * c->xpitch * sizeof(int16_t)))) return r;   // last: its presence says "all of it is there"
    }
    return M17HIP_OK;
}

// Where an in-place producer (m17hip_upload_i16, m17hip_upload_i16_device, m17hip_synth_i16) writes: the current input slab on the
// main stream, or — tuning knob 16 — the STAGING slab on the copy stream, as soon as the run before the latest one has released it.
struct InputTarget { int16_t* x = nullptr; hipStream_t st = nullptr; bool stage = false; };
static int input_target(m17hip_ctx* c, InputTarget& t)
{
    if (!c->stage_inputs) { t.x = c->xbuf; t.st = c->stream; t.stage = false; return M17HIP_OK; }
    const int r = stage_prepare(c);
    if (r) return r;
    const int other = c->slot ^ 1;
    if (c->slot_used[other]) HIPCHK(c, hipStreamWaitEvent(c->copy, c->ev_end[other], 0));
    t.x = c->xstage; t.st = c->copy; t.stage = true;
    return M17HIP_OK;
}
static void input_done(m17hip_ctx* c, const InputTarget& t, uint32_t C, uint32_t T)
{
    if (t.stage) {
        c->staged = true; c->staged_h2d = false; c->stagedC = C; c->stagedT = T;
        c->slabC[c->slot ^ 1] = C; c->slabT[c->slot ^ 1] = T;
        return;
    }
    c->uploaded = true;
    c->slabC[c->slot] = C; c->slabT[c->slot] = T;
    c->lastC = C; c->lastT = T;
    if (c->have_run) c->inplace_after_run = true;
}

int m17hip_upload_i16(m17hip_ctx* c, const int16_t* host, uint32_t C, uint32_t T, size_t pitch)
{
    if (!c || !host || C == 0 || T == 0 || C > c->maxC || T > c->maxT || pitch < T) return M17HIP_EINVAL;
    GUARD(c);
    if (c->front_pending) return M17HIP_ESTATE;   // the slabs belong to the run m17hip_demod_front has started
    InputTarget in;
    int r = input_target(c, in);
    if (r) return r;
    HIPCHK(c, hipMemcpy2DAsync(in.x + XPRE, c->xpitch * sizeof(int16_t), host, pitch * sizeof(int16_t), (size_t)T * sizeof(int16_t), C,
                               hipMemcpyHostToDevice, in.st));
    HIPCHK(c, hipStreamSynchronize(in.st));
    input_done(c, in, C, T);
    return M17HIP_OK;
}

int m17hip_upload_i16_async(m17hip_ctx* c, const int16_t* host, uint32_t C, uint32_t T, size_t pitch)
{
    if (!c || !host || C == 0 || T == 0 || C > c->maxC || T > c->maxT || pitch < T) return M17HIP_EINVAL;
    GUARD(c);
    if (c->front_pending) return M17HIP_ESTATE;
    int r = stage_prepare(c);
    if (r) return r;
    // the staging slab was the input of the run BEFORE the one now queued / running: free once that run is done with it
    const int other = c->slot ^ 1;
    if (c->slot_used[other]) HIPCHK(c, hipStreamWaitEvent(c->copy, c->ev_end[other], 0));
    HIPCHK(c, hipMemcpy2DAsync(c->xstage + XPRE, c->xpitch * sizeof(int16_t), host, pitch * sizeof(int16_t), (size_t)T * sizeof(int16_t), C,
                               hipMemcpyHostToDevice, c->copy));
    HIPCHK(c, hipEventRecord(c->ev_copy, c->copy));
    c->staged = true; c->staged_h2d = true; c->stagedC = C; c->stagedT = T;
    c->slabC[other] = C; c->slabT[other] = T;
    return M17HIP_OK;
}

int m17hip_upload_i16_device_async(m17hip_ctx* c, const int16_t* dev, uint32_t C, uint32_t T, size_t pitch)
{
    if (!c || !dev || C == 0 || T == 0 || C > c->maxC || T > c->maxT || pitch < T) return M17HIP_EINVAL;
    GUARD(c);
    if (c->front_pending) return M17HIP_ESTATE;
    int r = stage_prepare(c);
    if (r) return r;
    const int other = c->slot ^ 1;
    if (c->slot_used[other]) HIPCHK(c, hipStreamWaitEvent(c->copy, c->ev_end[other], 0));
    dim3 grid(((T + 7) / 8 + 255) / 256, C);
    hipLaunchKernelGGL(copy_rows_i16_kernel, grid, dim3(256), 0, c->copy, dev, pitch, c->xstage, c->xpitch, T);
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipEventRecord(c->ev_copy, c->copy));
    c->staged = true; c->staged_h2d = false; c->stagedC = C; c->stagedT = T;
    c->slabC[other] = C; c->slabT[other] = T;
    return M17HIP_OK;
}

int m17hip_input_alternate(m17hip_ctx* c, uint32_t C, uint32_t T)
{
    if (!c || C == 0 || T == 0 || C > c->maxC || T > c->maxT) return M17HIP_EINVAL;
    GUARD(c);
    if (c->front_pending) return M17HIP_ESTATE;
    const int other = c->slot ^ 1;
    if (!c->xstage || c->slabC[other] != C || c->slabT[other] != T) return M17HIP_ESTATE;   // the other slab does not hold such an input
    c->staged = true; c->staged_h2d = false; c->stagedC = C; c->stagedT = T;
    return M17HIP_OK;
}

int m17hip_synth_i16(m17hip_ctx* c, const m17_synth_params* params, uint32_t C, uint32_t T, uint32_t chan0)
{
    if (!c || !params || C == 0 || T == 0 || C > c->maxC || T > c->maxT || params->n_frames < 0 || params->kind > 4) return M17HIP_EINVAL;
    GUARD(c);
    if (c->front_pending) return M17HIP_ESTATE;   // the slabs belong to the run m17hip_demod_front has started
    if (params->kind == 4 && (params->n_frames < 1 || params->n_frames > 33)) return M17HIP_EINVAL;   // 5-bit frame numbers
    static_assert(sizeof(ModParams) == sizeof(m17_synth_params), "parameter block layout");
    ModParams mp;
    std::memcpy(&mp, params, sizeof(mp));
    const size_t sym_pitch = round_up((size_t)mod_max_symbols(mp.n_frames, mp.n_preamble), 16);
    const size_t sym_bytes = round_up((size_t)C * sym_pitch, 256);
    InputTarget in;
    int r = input_target(c, in);
    if (r) return r;
    // the symbol staging lives in its own allocation: the per-operator scratch may be in use by work queued on the main stream
    if (sym_bytes + (size_t)C * 4 > c->synth_bytes) {
        if (c->synth_scratch) { HIPCHK(c, hipDeviceSynchronize()); free_dev(c->synth_scratch, &c->last_hip); c->synth_bytes = 0; }
        HIPCHK(c, hipMalloc(&c->synth_scratch, sym_bytes + (size_t)C * 4));
        c->synth_bytes = sym_bytes + (size_t)C * 4;
    }
    int8_t* sym = reinterpret_cast<int8_t*>(c->synth_scratch);
    uint32_t* nsym = reinterpret_cast<uint32_t*>(reinterpret_cast<char*>(c->synth_scratch) + sym_bytes);
    hipLaunchKernelGGL(mod_symbols_kernel, dim3((C + 63) / 64), dim3(64), 0, in.st, mp, C, chan0, sym, sym_pitch, nsym);
    HIPCHK(c, hipGetLastError());
    hipLaunchKernelGGL(mod_shape_kernel, dim3((T + 255) / 256, C), dim3(256), 0, in.st, mp, C, T, chan0, sym, sym_pitch, nsym, in.x, c->xpitch);
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipStreamSynchronize(in.st));
    input_done(c, in, C, T);
    return M17HIP_OK;
}

int m17hip_download_i16(m17hip_ctx* c, int16_t* host, uint32_t C, uint32_t T, size_t pitch)
{
    if (!c || !host || C == 0 || T == 0 || C > c->maxC || T > c->maxT || pitch < T) return M17HIP_EINVAL;
    GUARD(c);
    if (!c->uploaded) return M17HIP_ESTATE;
    HIPCHK(c, hipMemcpy2DAsync(host, pitch * sizeof(int16_t), c->xbuf + XPRE, c->xpitch * sizeof(int16_t), (size_t)T * sizeof(int16_t), C,
                               hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return M17HIP_OK;
}

int m17hip_upload_i16_device(m17hip_ctx* c, const int16_t* dev, uint32_t C, uint32_t T, size_t pitch)
{
    if (!c || !dev || C == 0 || T == 0 || C > c->maxC || T > c->maxT || pitch < T) return M17HIP_EINVAL;
    GUARD(c);
    if (c->front_pending) return M17HIP_ESTATE;   // the slabs belong to the run m17hip_demod_front has started
    InputTarget in;
    int r = input_target(c, in);
    if (r) return r;
    dim3 grid(((T + 7) / 8 + 255) / 256, C);
    hipLaunchKernelGGL(copy_rows_i16_kernel, grid, dim3(256), 0, in.st, dev, pitch, in.x, c->xpitch, T);
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipStreamSynchronize(in.st));   // `dev` belongs to the caller again when this returns
    input_done(c, in, C, T);
    return M17HIP_OK;
}

int m17hip_fir_rrc150(m17hip_ctx* c, uint32_t C, uint32_t T, uint32_t flags, float* out_host)
{
    if (!c || C == 0 || T == 0 || C > c->maxC || T > c->maxT || (flags & ~M17HIP_FLAG_INVERT)) return M17HIP_EINVAL;
    GUARD(c);
    if (c->front_pending) return M17HIP_ESTATE;
    if (!c->uploaded) return M17HIP_ESTATE;
    c->fir_latency = false;   // (an operator call: the whole chip is its own)
    int r = launch_fir(c, C, T, flags, c->stream);
    if (r) return r;
    if (out_host) {
        HIPCHK(c, hipMemcpy2DAsync(out_host, (size_t)T * sizeof(float), c->ybuf + YPRE, c->ypitch * sizeof(float), (size_t)T * sizeof(float), C,
                                   hipMemcpyDeviceToHost, c->stream));
    }
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return M17HIP_OK;
}

int m17hip_correlator(m17hip_ctx* c, uint32_t C, uint32_t T, float* limit_host, float* corr_host)
{
    if (!c || C == 0 || T == 0 || C > c->maxC || T > c->maxT) return M17HIP_EINVAL;
    GUARD(c);
    if (c->front_pending) return M17HIP_ESTATE;
    const size_t n = (size_t)C * T;
    int r = ensure_scratch(c, 5 * n * sizeof(float));
    if (r) return r;
    float* limit = (float*)c->scratch;
    float* corr = limit + n;
    {   // the limit filter is a handful of latency-bound workgroups, the correlations an HBM-bound elementwise pass: side by side
        Timed tm(c, KT_CORR);
        HIPCHK(c, hipEventRecord(c->ev_fork, c->stream));
        HIPCHK(c, hipStreamWaitEvent(c->side, c->ev_fork, 0));
        if (T % 4 == 0) hipLaunchKernelGGL(correlate4_kernel, dim3((T / 4 + 255) / 256, C), dim3(256), 0, c->side, c->ybuf, c->ypitch, corr, C, T, 0u, T);
        else hipLaunchKernelGGL(correlate_kernel, dim3((T + 255) / 256, C), dim3(256), 0, c->side, c->ybuf, c->ypitch, corr, C, T, 0u, T);
        HIPCHK(c, hipGetLastError());
        HIPCHK(c, hipEventRecord(c->ev_join, c->side));
#ifdef M17_TOOLS
        if (c->limit_form == 0 && T % LP_TILE == 0 && T >= 4 * LP_TILE && (((size_t)C * T) & 3) == 0)
            hipLaunchKernelGGL(limit_pipe_kernel, dim3((C + LP_CH - 1) / LP_CH), dim3(320), 0, c->stream, c->ybuf, c->ypitch, limit, (size_t)T, C, T, (const float*)nullptr, (float*)nullptr);
        else
#endif
        if (T % LR_TILE == 0 && (((size_t)C * T) & 3) == 0)
            hipLaunchKernelGGL(limit_relay_kernel, dim3((C + LR_CH - 1) / LR_CH), dim3(320), 0, c->stream, c->ybuf, c->ypitch, limit, (size_t)T, C, T, (const float*)nullptr, (float*)nullptr);
        else
            hipLaunchKernelGGL(limit_kernel, dim3((C + 63) / 64), dim3(64), 0, c->stream, c->ybuf, c->ypitch, limit, C, T);
        HIPCHK(c, hipGetLastError());
        HIPCHK(c, hipStreamWaitEvent(c->stream, c->ev_join, 0));
    }
    HIPCHK(c, hipGetLastError());
    if (limit_host) HIPCHK(c, hipMemcpyAsync(limit_host, limit, n * sizeof(float), hipMemcpyDeviceToHost, c->stream));
    if (corr_host) HIPCHK(c, hipMemcpyAsync(corr_host, corr, 4 * n * sizeof(float), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return M17HIP_OK;
}

int m17hip_dcd(m17hip_ctx* c, uint32_t C, uint32_t T, uint32_t flags, float* sums_host, uint32_t* ticks_out)
{
    if (!c || C == 0 || T == 0 || C > c->maxC || T > c->maxT) return M17HIP_EINVAL;
    if (flags & ~M17HIP_FLAG_INVERT) return M17HIP_EINVAL;
    GUARD(c);
    if (c->front_pending) return M17HIP_ESTATE;
    if (!c->uploaded) return M17HIP_ESTATE;
    // operator-level call: always from a fresh DFT state at stream position 0
    HIPCHK(c, hipMemsetAsync(c->dcd_state, 0, (size_t)C * sizeof(DcdState), c->stream));
    const uint64_t saved = c->pos;
    c->pos = 0;
    c->dcd_latency = c->dcd_form == 1;
    int r = launch_dcd(c, C, T, flags, c->stream);
    c->pos = saved;
    if (r) return r;
    const uint32_t ticks = T / TICK;
    if (ticks_out) *ticks_out = ticks;
    if (sums_host && ticks)
        HIPCHK(c, hipMemcpy2DAsync(sums_host, (size_t)ticks * 12 * sizeof(float), c->dcd_table, (size_t)c->ticks_cap * 12 * sizeof(float),
                                   (size_t)ticks * 12 * sizeof(float), C, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return M17HIP_OK;
}

int m17hip_viterbi(m17hip_ctx* c, const int8_t* soft_host, uint32_t n, int kind, uint8_t* bits_host, int32_t* cost_host)
{
    if (!c || !soft_host || n == 0 || kind < 0 || kind > 3) return M17HIP_EINVAL;
    GUARD(c);
    static const int IN[4] = {488, 296, 420, 402}, OUT[4] = {240, 144, 206, 197};
    const size_t in_b = (size_t)n * IN[kind], out_b = (size_t)n * OUT[kind];
    const size_t o1 = round_up(in_b, 256), o2 = o1 + round_up(out_b, 256);
    int r = ensure_scratch(c, o2 + (size_t)n * 4);
    if (r) return r;
    char* base = (char*)c->scratch;
    HIPCHK(c, hipMemcpyAsync(base, soft_host, in_b, hipMemcpyHostToDevice, c->stream));
    {
        Timed tm(c, KT_DEC);
        hipLaunchKernelGGL(viterbi_kernel, dim3((n + 63) / 64), dim3(64), (122 + 122 + 16) * 64 * 4, c->stream, (const int8_t*)base, n, kind,
                           (uint8_t*)(base + o1), (int32_t*)(base + o2), c->tables);
    }
    HIPCHK(c, hipGetLastError());
    if (bits_host) HIPCHK(c, hipMemcpyAsync(bits_host, base + o1, out_b, hipMemcpyDeviceToHost, c->stream));
    if (cost_host) HIPCHK(c, hipMemcpyAsync(cost_host, base + o2, (size_t)n * 4, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return M17HIP_OK;
}

int m17hip_slice_llr(m17hip_ctx* c, const float* sym_host, uint32_t rows, uint32_t n, int8_t* llr_host, float* evm_host)
{
    if (!c || !sym_host || rows == 0 || n == 0) return M17HIP_EINVAL;
    GUARD(c);
    const size_t cnt = (size_t)rows * n;
    const size_t o1 = round_up(cnt * 4, 256), o2 = o1 + round_up(cnt * 2, 256);
    int r = ensure_scratch(c, o2 + cnt * 4);
    if (r) return r;
    char* b = (char*)c->scratch;
    HIPCHK(c, hipMemcpyAsync(b, sym_host, cnt * 4, hipMemcpyHostToDevice, c->stream));
    hipLaunchKernelGGL(slice_kernel, dim3((rows + 63) / 64), dim3(64), 0, c->stream, (const float*)b, rows, n, (int8_t*)(b + o1),
                       (float*)(b + o2), c->llr_edges);
    HIPCHK(c, hipGetLastError());
    if (llr_host) HIPCHK(c, hipMemcpyAsync(llr_host, b + o1, cnt * 2, hipMemcpyDeviceToHost, c->stream));
    if (evm_host) HIPCHK(c, hipMemcpyAsync(evm_host, b + o2, cnt * 4, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return M17HIP_OK;
}

int m17hip_decode_frames(m17hip_ctx* c, const int8_t* llr368_host, uint32_t n, const uint8_t* sync_type, uint8_t* state_io,
                         uint8_t* lich_io, uint8_t* lsf_io, int8_t* dep401_io, int64_t* cost_io, m17_frame_rec* recs, uint8_t* nrec)
{
    if (!c || !llr368_host || !sync_type || !state_io || !lich_io || !lsf_io || !dep401_io || !cost_io || !recs || !nrec || n == 0)
        return M17HIP_EINVAL;
    GUARD(c);
    size_t off = 0;
    auto take = [&](size_t bytes) { const size_t o = off; off += round_up(bytes, 256); return o; };
    const size_t o_llr = take((size_t)n * 368), o_st = take(n), o_state = take(n), o_lich = take(n), o_lsf = take((size_t)n * 30),
                 o_dep = take(n), o_cost = take((size_t)n * 8), o_rec = take((size_t)n * 2 * sizeof(FrameRec)), o_nrec = take(n);
    int r = ensure_scratch(c, off);
    if (r) return r;
    char* b = (char*)c->scratch;
    HIPCHK(c, hipMemcpyAsync(b + o_llr, llr368_host, (size_t)n * 368, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(b + o_st, sync_type, n, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(b + o_state, state_io, n, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(b + o_lich, lich_io, n, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(b + o_lsf, lsf_io, (size_t)n * 30, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(b + o_dep, dep401_io, n, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(b + o_cost, cost_io, (size_t)n * 8, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemsetAsync(b + o_rec, 0, (size_t)n * 2 * sizeof(FrameRec), c->stream));
    DecodeFramesParams P{(const int8_t*)(b + o_llr), (const uint8_t*)(b + o_st), (uint8_t*)(b + o_state), (uint8_t*)(b + o_lich),
                         (uint8_t*)(b + o_lsf), (int8_t*)(b + o_dep), (int64_t*)(b + o_cost), (FrameRec*)(b + o_rec),
                         (uint8_t*)(b + o_nrec), n, c->tables, c->overflow};
    {
        Timed tm(c, KT_DEC);
        hipLaunchKernelGGL(decode_frames_kernel, dim3((n + 63) / 64), dim3(64), (92 + 122 + 16) * 64 * 4, c->stream, P);
    }
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipMemcpyAsync(state_io, b + o_state, n, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipMemcpyAsync(lich_io, b + o_lich, n, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipMemcpyAsync(lsf_io, b + o_lsf, (size_t)n * 30, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipMemcpyAsync(dep401_io, b + o_dep, n, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipMemcpyAsync(cost_io, b + o_cost, (size_t)n * 8, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipMemcpyAsync(recs, b + o_rec, (size_t)n * 2 * sizeof(FrameRec), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipMemcpyAsync(nrec, b + o_nrec, n, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return M17HIP_OK;
}

int m17hip_demod_reset(m17hip_ctx* c)
{
    if (!c) return M17HIP_EINVAL;
    GUARD(c);
    if (c->front_pending) {   // a front end queued by m17hip_demod_front is abandoned: let it drain, its results are not used
        HIPCHK(c, hipStreamSynchronize(c->side));
        HIPCHK(c, hipStreamSynchronize(c->side2));
        HIPCHK(c, hipStreamSynchronize(c->side3));
        c->front_pending = false;
        c->gate0_queued = false;
    }
    // the payload work of the runs before (deferred decode, consumers: their state is reset below) is waited for ON the device
    // (payload work that was never asked for goes with the stream it belonged to: the consumers' state is reset below)
    for (auto& rs : c->sets) {
        HIPCHK(c, hipStreamWaitEvent(c->stream, rs.done, 0));   // (never recorded: no wait)
        rs.pending = false;
    }
    c->fold_pending = false;   // (the EVM state it would have updated is reset below)
    hipLaunchKernelGGL(seq_reset_kernel, dim3((c->maxC + 63) / 64), dim3(64), 0, c->stream, c->seq_state, c->dcd_state, c->ev_state, c->maxC);
    HIPCHK(c, hipGetLastError());
    hipLaunchKernelGGL(zero_prefix_kernel, dim3(c->maxC), dim3(64), 0, c->stream, c->xbuf, c->xpitch, c->ybuf, c->ypitch, c->maxC);
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipMemset2DAsync(c->hbuf, c->ypitch * sizeof(float), 0, YPRE * sizeof(float), c->maxC, c->stream));
    hipLaunchKernelGGL(bert_reset_kernel, dim3((c->maxC + 63) / 64), dim3(64), 0, c->stream, (BertState*)c->bert_state, c->maxC);
    HIPCHK(c, hipGetLastError());
    if (c->pkt_cap) {
        hipLaunchKernelGGL(packet_reset_kernel, dim3((c->maxC + 63) / 64), dim3(64), 0, c->stream, (PacketState*)c->pkt_state, c->maxC);
        HIPCHK(c, hipGetLastError());
        HIPCHK(c, hipMemsetAsync(c->pkt_count2, 0, 8, c->stream));
    }
    for (auto& rs : c->sets) {
        if (rs.rec_count) HIPCHK(c, hipMemsetAsync(rs.rec_count, 0, (size_t)c->maxC * 4, c->stream));
        rs.valid = false;
    }
    HIPCHK(c, hipMemsetAsync(c->overflow, 0, 32, c->stream));
    HIPCHK(c, hipEventRecord(c->ev_mark, c->stream));   // the front end of a staged run starts on its own streams: not before this
    HIPCHK(c, hipStreamWaitEvent(c->pay(), c->ev_mark, 0));   // (nor anything a fetch queues on the payload stream)
    c->pos = 0;
    c->have_run = false;
    c->inplace_after_run = false;
    c->sel_back = 0;
    return M17HIP_OK;
}

namespace {

// How a run of T samples is cut into segments (tuning knobs 3, 4, 33): segment k covers [t0(k), t0(k + 1)).  Equal ones of seg_len samples;
// optionally a short first one (key 4), or a RAMP (key 33): segments of r, 2 r, 4 r, ... samples until seg_len is reached.  The ramp is for
// where channels leave the limit-filter replay — at the start of a transmission, while sync is being acquired: a channel that leaves it in
// segment k carries the filter itself to the end of segment k + 1, so what a drop costs the launches it falls into is twice the segment's length.
constexpr uint32_t AUTO_RAMP = 9600;   // (2400 ... 24 000 measured: NOTES 6.5)
struct SegPlan {
    uint32_t T, seg_len, nseg;
    std::vector<uint32_t> b;   // b[k] = first sample of segment k; b[nseg] = T
    SegPlan(const m17hip_ctx* c, uint32_t T_) : T(T_)
    {
        seg_len = (!c->profile && c->seg_len) ? c->seg_len : T;
        b.push_back(0);
        uint32_t pos = 0;
        if (seg_len < T && !c->profile) {
            if (c->ramp_now && c->ramp_now < seg_len) {
                for (uint32_t len = c->ramp_now; len < seg_len && pos + len < T; len *= 2) { pos += len; b.push_back(pos); }
            } else if (c->seg0_len && c->seg0_len < seg_len) {
                pos = c->seg0_len; b.push_back(pos);
            }
        }
        while (pos + seg_len < T) { pos += seg_len; b.push_back(pos); }
        // Segment starts are multiples of eight samples: the matched filter loads its input sixteen bytes at a time and stores float4s from a
        // segment's first sample on (an odd start made those accesses misaligned — the hardware splits them, C++ calls it undefined)
        for (size_t k = 1; k < b.size(); ++k) b[k] &= ~7u;
        b.erase(std::unique(b.begin(), b.end()), b.end());
        b.push_back(T);
        nseg = (uint32_t)b.size() - 1u;
    }
    uint32_t t0(uint32_t k) const { return b[std::min<size_t>(k, nseg)]; }
};

static int ensure_seg_events(m17hip_ctx* c, int q, uint32_t nseg)
{
    while (c->ev_fir_[q].size() < nseg) {
        for (auto* v : {&c->ev_fir_[q], &c->ev_dcd_[q], &c->ev_gate_[q], &c->ev_redo_[q], &c->ev_seq_[q]}) {
            hipEvent_t e;
            HIPCHK(c, hipEventCreateWithFlags(&e, hipEventDisableTiming));
            v->push_back(e);
        }
    }
    return M17HIP_OK;
}

// Is the run being queued gate-aware (m17hip_tune key 26)?  -1: yes when more than a quarter of the channel-segments of the last fetched run
// ended with the carrier off (K5 counts them: overflow[3], read with every fetch) — an always-on load keeps K1 running ahead freely.
// Needs at least three segments and the default front-end schedule.
static int gate_mode_for_run(m17hip_ctx* c, const SegPlan& sp)
{
    const bool want = c->gate_aware == 1 || (c->gate_aware < 0 && c->chan_segs_prev && 4ull * c->off_segs_prev > (uint64_t)c->chan_segs_prev);
    c->gate_run = want && sp.nseg >= 3u && !c->front_ahead && !c->profile;
    // A gate-aware run is a run on mostly idle channels: K1 skips most of its work, K5's waves jump from update point to update point — what is left of the step
    // is K3's chain, which sees every sample whatever the gate does.  Its latency form then (four waves per 32 channels, 1.1-1.9 ms per launch against
    // 2.2-3.7): the wave slots it takes are free on such input.
    if (c->gate_run && c->dcd_form < 0) c->dcd_latency = true;
    if (!c->truth) {   // (K5 leaves the gate state and counts the closed gates in every mode: the next run's choice comes from it)
        HIPCHK(c, hipMalloc((void**)&c->truth, 2 * (size_t)c->maxC * sizeof(GateTruth)));
        HIPCHK(c, hipMalloc((void**)&c->first_needed, (size_t)c->maxC * sizeof(uint32_t)));
    }
    return M17HIP_OK;
}

// K3 and K1 of segment k of the run being queued (slab pair c->slot), on the two side streams.
// The front end of segment k may be held back until K5 of segment k - front_ahead is done (tuning knob 5), to spread it over
// the step; measured, letting it run ahead freely is faster (K3 is a latency chain of 1.7 ms per segment: held back, it is
// what K5 ends up waiting for).
// `which`: bit 0 = K3, bit 1 = K1 (a gate-aware run queues K1 of segment k >= 2 later, behind K5 of segment k - 2: launch_gated_fir).
static int launch_front_seg(m17hip_ctx* c, const SegPlan& sp, uint32_t k, uint32_t C, uint32_t flags, uint32_t which = 3u)
{
    if (k >= sp.nseg) return M17HIP_OK;
    const int q = c->slot;
    const uint32_t ahead = c->front_ahead ? c->front_ahead : sp.nseg;
    const uint32_t t0 = sp.t0(k), len = sp.t0(k + 1) - t0;
    if (k >= ahead) {
        HIPCHK(c, hipStreamWaitEvent(c->side, c->ev_seq_[q][k - ahead], 0));
        HIPCHK(c, hipStreamWaitEvent(c->side2, c->ev_seq_[q][k - ahead], 0));
    }
    if (k == 1) HIPCHK(c, hipStreamWaitEvent(c->side, c->ev_fir_[q][0], 0));   // segment 0's front end first: K2/K5 wait for it
    int r2;
    if (which & 1u) {
        if ((r2 = launch_dcd(c, C, len, flags, c->side, t0))) return r2;
        HIPCHK(c, hipEventRecord(c->ev_dcd_[q][k], c->side));
    }
    if (which & 2u) {
        if ((r2 = launch_fir(c, C, len, flags, c->side2, t0))) return r2;
        HIPCHK(c, hipEventRecord(c->ev_fir_[q][k], c->side2));
    }
    return M17HIP_OK;
}

// The whole front end of a run as far as it can be queued at once: K3 and K1 of every segment — or, for a gate-aware run, K3 of every segment
// and K1 of the first two (K1 of segment k >= 2 follows K5 of segment k - 2 and its forecast: launch_gated_fir, from the K5 loop).
static int launch_front_all(m17hip_ctx* c, const SegPlan& sp, uint32_t C, uint32_t flags)
{
    int r;
    for (uint32_t k = 0; k < c->front_segs; ++k)
        if ((r = launch_front_seg(c, sp, k, C, flags, (c->gate_run && k >= 2u) ? 1u : 3u))) return r;
    return M17HIP_OK;
}

// Gate-aware run, after K5 of segment k has been queued (ev_seq[k] recorded): the forecast for segment k + 2 from K5's gate state at the
// end of segment k and the table rows of segments k + 1 and k + 2, then K1 of segment k + 2 over what the carrier can be on for.
static int launch_gated_fir(m17hip_ctx* c, const SegPlan& sp, uint32_t k, uint32_t C, uint32_t flags)
{
    if (k + 2u >= sp.nseg) return M17HIP_OK;
    const int q = c->slot;
    const uint32_t t1 = sp.t0(k + 1u), t2 = sp.t0(k + 2u), t3 = sp.t0(k + 3u);
    HIPCHK(c, hipStreamWaitEvent(c->side2, c->ev_seq_[q][k], 0));
    HIPCHK(c, hipStreamWaitEvent(c->side2, c->ev_dcd_[q][k + 2u], 0));
    hipLaunchKernelGGL(gate_forecast_kernel, dim3((C + 63) / 64), dim3(64), 0, c->side2, c->truth + (size_t)(k & 1u) * c->maxC, c->dcd_table, c->ticks_cap,
                       (uint64_t)(c->pos / TICK), (uint64_t)(c->pos + t1), t2 - t1, t3 - t1, c->first_needed, C);
    HIPCHK(c, hipGetLastError());
    int r = launch_fir(c, C, t3 - t2, flags, c->side2, t2, c->first_needed);
    if (r) return r;
    HIPCHK(c, hipEventRecord(c->ev_fir_[q][k + 2u], c->side2));
    return M17HIP_OK;
}

// One launch of K2 over segment k of the run whose slabs the context names: the whole segment from K5's state (first segment), ahead of
// K5 from K2's own state, or the redo of the channels K5 flagged in the previous segment.
static int launch_gate_seg(m17hip_ctx* c, const SegPlan& sp, uint32_t k, hipStream_t st, bool ahead, bool redo, uint32_t C, uint32_t flags, bool redo_stores = false)
{
    const uint32_t t0 = sp.t0(k), len = sp.t0(k + 1) - t0;
    TimedK tm(c, KT_GATE);
    GateParams G{};
    G.x = c->xbuf + t0; G.xpitch = c->xpitch; G.y = c->ybuf + t0; G.ypitch = c->ypitch; G.h = c->hbuf + t0;
    G.dcd_table = c->dcd_table; G.ticks_cap = c->ticks_cap; G.state = c->seq_state;
    G.final_h = c->final_h + (size_t)(k & 1u) * c->maxC * 4;
    G.chain_in = ahead ? c->gate_exp : nullptr; G.chain_out = c->gate_exp;
    G.only = redo ? c->dropped + (size_t)((k - 1u) & 1u) * c->maxC : nullptr;   // (flags by segment parity)
    G.bnd = redo ? c->bnd + (size_t)(k & 1u) * c->maxC : nullptr;   // (written by K5 of segment k - 1)
    G.taps = c->taps; G.C = C; G.T = len; G.pos0 = c->pos + t0; G.tick_row0 = c->pos / TICK; G.flags = flags | ((redo && !redo_stores) ? 2u : 0u);
    G.nblk = (C + GT_CPW - 1) / GT_CPW;
    uint32_t fold_blocks = 0;
    if (ahead && k >= 2 && c->defer_evm && c->ev_ops2[c->ev_par]) {   // (K5 of segment k - 2 is through: its EVM operations ride along, sixteen channels per block)
        G.ev = EvParams{c->ev_ops2[c->ev_par], c->ev_pitch, c->ev_state, c->diag_cap ? c->diag_log : nullptr, c->diag_cap, c->seq_state, C, c->ev_cur + (size_t)((k - 2u) & 1u) * c->maxC, 0u};
        fold_blocks = ev_fold_blocks(C);
    }
    // The LAST pass of the run before (its buffer is not this run's): beside K5 of this run's first segment, with the replay that runs ahead for the
    // second one — or, for a run of one segment, with this replay, which K5 waits for (that K5 is the one that moves the end-of-run cursors on)
    if (!redo && c->fold_pending && c->fold_ops && ((sp.nseg >= 2 && ahead && k == 1) || (sp.nseg < 2 && k == 0))) {
        // (beside the replay, last = 2: m17_diag is NOT settled by this pass — K5 of this run's first segment has the state in its hands meanwhile; found by the
        //  sweep: a run whose last segment fires no callback kept the value this pass had put over the first segment's mark.  In front of segment 0, which K5
        //  waits for, it is settled as always)
        G.ev = EvParams{c->fold_ops, c->ev_pitch, c->ev_state, c->diag_cap ? c->diag_log : nullptr, c->diag_cap, c->seq_state, c->fold_C, c->fold_end, k == 0 ? 1u : 2u};
        fold_blocks = ev_fold_blocks(c->fold_C);
        c->fold_pending = false;
    }
    tm.launch(limit_track_kernel, dim3(G.nblk + fold_blocks), dim3(64), GT_LDS_FLOATS * sizeof(float), st, G);
    HIPCHK(c, hipGetLastError());
    return M17HIP_OK;
}

// A staged run begins: the slab pairs swap, the 152-sample tail of the previous input is carried into the new slab's prefix, and the
// front end (K1, K3: nothing in them depends on the outcome of the run before) is queued on the side streams — NOT ordered behind
// the main stream, where K2 / K5 of the previous run may still have a long way to go.
static int begin_staged(m17hip_ctx* c, uint32_t C, uint32_t T, uint32_t flags, bool from_front)
{
    if (C != c->stagedC || T != c->stagedT) return M17HIP_EINVAL;
    if (c->have_run && C != c->lastC) return M17HIP_EINVAL;  // a continued stream keeps its channel count
    const int16_t* xprev = c->xbuf;
    std::swap(c->xbuf, c->xstage);
    std::swap(c->ybuf, c->yalt);
    std::swap(c->hbuf, c->halt);
    std::swap(c->dcd_table, c->dcd_alt);
    c->slot ^= 1;
    c->staged = false;
    c->uploaded = true;
    c->carryT = c->have_run ? c->runT : 0;
    // the new slab's prefix, behind the staged copy on the copy stream; the slab pair itself is free since ev_end[slot] (the copy
    // stream waited for it when the input was staged — m17hip_input_alternate stages without a copy, so wait here as well)
    if (c->slot_used[c->slot]) HIPCHK(c, hipStreamWaitEvent(c->copy, c->ev_end[c->slot], 0));
    if (c->carryT >= (uint32_t)XPRE && !c->inplace_after_run)   // the tail of the previous input, where it lies (that slab is only read while its run is in flight)
        hipLaunchKernelGGL(copy_tail_i16_kernel, dim3(C), dim3(64), 0, c->copy, xprev, c->xbuf, c->xpitch, c->carryT);
    else if (c->carryT) {              // (or its data region was overwritten in place since: the tail its last kernel carried into its prefix)              // a run shorter than the prefix: its tail reaches into its own prefix, which its last kernel rewrites — wait for that
        HIPCHK(c, hipStreamWaitEvent(c->copy, c->ev_end[c->slot ^ 1], 0));
        hipLaunchKernelGGL(copy_prefix_i16_kernel, dim3(C), dim3(64), 0, c->copy, xprev, c->xbuf, c->xpitch);
    } else HIPCHK(c, hipMemset2DAsync(c->xbuf, c->xpitch * sizeof(int16_t), 0, XPRE * sizeof(int16_t), C, c->copy));
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipEventRecord(c->ev_in_ready, c->copy));
    for (hipStream_t st : {c->side, c->side2}) {
        HIPCHK(c, hipStreamWaitEvent(st, c->ev_in_ready, 0));
        HIPCHK(c, hipStreamWaitEvent(st, c->ev_mark, 0));
    }
    if (c->front_k1_after && c->have_run && c->last_nseg) {   // K1 out of the way of the previous run's first (heaviest) K5 launches
        const uint32_t j = std::min(c->front_k1_after, c->last_nseg) - 1u;
        HIPCHK(c, hipStreamWaitEvent(c->side2, c->ev_seq_[c->slot ^ 1][j], 0));
    }
    c->ramp_now = c->seg_ramp > 0 ? (uint32_t)c->seg_ramp : 0u;   // (a continued stream: the extra launches of a ramp cost its chain more than the shorter drops save)
    const SegPlan sp(c, T);
    int r = ensure_seg_events(c, c->slot, sp.nseg);
    if (r) return r;
    const uint32_t ahead = c->front_ahead ? c->front_ahead : sp.nseg;
    c->dcd_latency = c->dcd_form < 0 ? (from_front || !runs_overlap(c)) : c->dcd_form == 1;
    c->fir_latency = c->dcd_latency;
    c->front_segs = std::min(ahead, sp.nseg);
    if ((r = gate_mode_for_run(c, sp))) return r;
    if ((r = launch_front_all(c, sp, C, flags))) return r;
    c->front_was_staged = true;
    c->frontC = C; c->frontT = T; c->front_flags = flags;
    return M17HIP_OK;
}

}  // namespace

// The latest run's last EVM fold pass on the main stream (behind that run), if nothing has taken it along yet.
static int flush_fold(m17hip_ctx* c)
{
    if (!c->fold_pending) return M17HIP_OK;
    c->fold_pending = false;
    if (!c->fold_ops) return M17HIP_OK;
    hipLaunchKernelGGL(evm_deferred_kernel, dim3(ev_fold_blocks(c->fold_C)), dim3(64), 0, c->stream,
                       EvParams{c->fold_ops, c->ev_pitch, c->ev_state, c->diag_cap ? c->diag_log : nullptr, c->diag_cap, c->seq_state, c->fold_C, c->fold_end, 1u});
    HIPCHK(c, hipGetLastError());
    return M17HIP_OK;
}

// Queue the payload work of the runs whose results nobody has asked for yet, in run order, on the payload stream: the frames K5 left to
// decode_deferred_kernel (one lane per frame), then the consumers (their state goes from run to run), then `done`.
// `selected_only`: up to the run the fetch family names (m17hip_frames_select) — a fetch of run k must not queue the work of run k + 1, whose
// chain may have just begun: everything behind it on the payload stream would wait for that run's end.
// `older_only`: the run before the latest alone (its record set is about to be written again).
static int flush_payload(m17hip_ctx* c, bool selected_only = false, bool older_only = false)
{
    for (int j = 0; j < 2; ++j) {
        const int i = j == 0 ? (c->cur ^ 1) : c->cur;   // the older run first
        if (j == 1 && (older_only || (selected_only && (c->sel_back & 1u)))) break;
        m17hip_ctx::RecSet& rs = c->sets[i];
        if (!rs.valid || !rs.pending) continue;
        const hipStream_t ps = c->pay();
        HIPCHK(c, hipStreamWaitEvent(ps, c->sets[i ^ 1].done, 0));   // (the run before it, whichever stream did its work; never recorded: no wait)
        HIPCHK(c, hipStreamWaitEvent(ps, rs.chain, 0));
        const uint32_t C = rs.C;
        if (rs.defer_llr && c->defer_hist) {   // (frames are deferred only while both stores exist: m17hip_tune key 15)
            DeferParams D{};
            D.recs = rs.recs; D.rec_cap = rs.rec_cap; D.rec_count = rs.rec_count; D.defer = rs.defer_llr; D.hist = c->defer_hist; D.tables = c->tables;
            D.state = c->seq_state; D.diag_log = c->diag_cap ? c->diag_log : nullptr; D.diag_cap = c->diag_cap; D.diag_count = c->diag_count; D.C = C;
            D.ev = EvParams{nullptr, 0, nullptr, nullptr, 0, nullptr, C, nullptr, 0u};
            uint32_t fold_blocks = 0;
            if (c->fold_with_decode && c->fold_pending && i == c->cur && c->fold_ops) {   // (the latest run's last fold pass, beside its decode on the main stream)
                D.ev = EvParams{c->fold_ops, c->ev_pitch, c->ev_state, c->diag_cap ? c->diag_log : nullptr, c->diag_cap, c->seq_state, c->fold_C, c->fold_end, 1u};
                fold_blocks = ev_fold_blocks(C);
                c->fold_pending = false;
            }
            TimedK tm(c, KT_DEC);
            tm.launch(decode_deferred_kernel, dim3(defer_blocks(C) + fold_blocks), dim3(64 * DEFER_CPB), DEFER_LDS_BYTES, ps, D);
            HIPCHK(c, hipGetLastError());
        }
        if (rs.bert && c->bert_state)   // payload consumer: PRBS9 statistics over this run's BERT records
            hipLaunchKernelGGL(bert_stats_kernel, dim3((C + 63) / 64), dim3(64), 0, ps, rs.recs, rs.rec_cap, rs.rec_count, (BertState*)c->bert_state, C);
        if (rs.pkt && c->pkt_cap) {   // payload consumer: packet reassembly over this run's packet records
            HIPCHK(c, hipMemsetAsync(c->pkt_count2 + i, 0, 4, ps));
            hipLaunchKernelGGL(packet_asm_kernel, dim3((C + 63) / 64), dim3(64), 0, ps, rs.recs, rs.rec_cap, rs.rec_count, (PacketState*)c->pkt_state, C,
                               (PacketRec*)c->pkt_recs2[i], c->pkt_cap, c->pkt_count2 + i, c->channel_base);
        }
        HIPCHK(c, hipGetLastError());
        HIPCHK(c, hipEventRecord(rs.done, ps));
        rs.pending = false;
    }
    return M17HIP_OK;
}

// BASELINE configs[1] as ONE call: matched filter, limit filter and the four correlations of `samples` samples per channel, pipelined in time.
// The limit filter is one dependent chain per channel over the whole run (12 ns per sample) and is what the call lasts; the matched filter
// (K1) of piece k + 1 and the correlations of piece k run beside the chain's piece k on two side streams, the chain's state (four history
// values per channel) carried from piece to piece.  Results identical to m17hip_fir_rrc150 followed by m17hip_correlator.
int m17hip_fir_correlator(m17hip_ctx* c, uint32_t C, uint32_t T, uint32_t flags, float* y_host, float* limit_host, float* corr_host)
{
    if (!c || C == 0 || T == 0 || C > c->maxC || T > c->maxT || (flags & ~M17HIP_FLAG_INVERT)) return M17HIP_EINVAL;
    GUARD(c);
    if (c->front_pending) return M17HIP_ESTATE;
    if (!c->uploaded) return M17HIP_ESTATE;
    const size_t n = (size_t)C * T;
    c->fir_latency = false;
    int r = ensure_scratch(c, 5 * n * sizeof(float) + (size_t)C * 4 * sizeof(float));
    if (r) return r;
    float* limit = (float*)c->scratch;
    float* corr = limit + n;
    float* lstate = corr + 4 * n;
    // pieces of whole 256-sample tiles (the limit pipeline's granule), about a tenth of the run each; anything else: one piece, the plain kernels
    const bool tiled = T % LP_TILE == 0 && T >= 8 * LP_TILE && (n & 3) == 0;
    const uint32_t piece = tiled ? std::max<uint32_t>((uint32_t)round_up((T + 9) / 10, LP_TILE), 4 * LP_TILE) : T;
    uint32_t npieces = (T + piece - 1) / piece;
    if (tiled && npieces > 1 && T - (npieces - 1) * piece < 4 * LP_TILE) --npieces;   // (a last piece of fewer than four tiles joins the one before it)
    if ((r = ensure_seg_events(c, c->slot, npieces))) return r;
    auto& ev_fir = c->ev_fir_[c->slot];
    // (Tried: the chain on compute units of its own — hipExtStreamCreateWithCUMask, a quarter of the chip — with the two throughput kernels on
    //  the rest: the chain's pieces 0.79 -> 0.74 ms, the call 8.2 -> 9.0 ms.  What stretches the chain beside them is not its SIMD: NOTES 5.4.)
    const hipStream_t st_chain = c->stream, st_fir = c->side2, st_corr = c->side;
    const uint32_t chain_wgs = (C + LP_CH - 1) / LP_CH;
    HIPCHK(c, hipEventRecord(c->ev_fork, c->stream));
    for (hipStream_t st : {st_chain, st_fir, st_corr})
        if (st != c->stream) HIPCHK(c, hipStreamWaitEvent(st, c->ev_fork, 0));
    if (tiled) HIPCHK(c, hipMemsetAsync(lstate, 0, (size_t)C * 4 * sizeof(float), st_chain));
    for (uint32_t k = 0; k < npieces; ++k) {
        const uint32_t t0 = k * piece, len = k + 1 < npieces ? piece : T - t0;
        if ((r = launch_fir(c, C, len, flags, st_fir, t0))) return r;
        HIPCHK(c, hipEventRecord(ev_fir[k], st_fir));
        HIPCHK(c, hipStreamWaitEvent(st_corr, ev_fir[k], 0));
        HIPCHK(c, hipStreamWaitEvent(st_chain, ev_fir[k], 0));
        {   // each kernel timed on the stream it runs on, under its own key: the correlations as "correlator", the limit chain as "limit_track"
            Timed tc(c, KT_CORR, st_corr);
            if (T % 4 == 0 && t0 % 4 == 0 && len % 4 == 0)
                hipLaunchKernelGGL(correlate4_kernel, dim3((len / 4 + 255) / 256, C), dim3(256), 0, st_corr, c->ybuf, c->ypitch, corr, C, len, t0, T);
            else
                hipLaunchKernelGGL(correlate_kernel, dim3((len + 255) / 256, C), dim3(256), 0, st_corr, c->ybuf, c->ypitch, corr, C, len, t0, T);
        }
        Timed tm(c, KT_GATE, st_chain);
#ifdef M17_TOOLS
        if (tiled && c->limit_form == 0)
            hipLaunchKernelGGL(limit_pipe_kernel, dim3(chain_wgs), dim3(320), 0, st_chain, c->ybuf + t0, c->ypitch, limit + t0, (size_t)T, C, len,
                               (const float*)lstate, lstate);
        else
#endif
        if (tiled)
            hipLaunchKernelGGL(limit_relay_kernel, dim3(chain_wgs), dim3(320), 0, st_chain, c->ybuf + t0, c->ypitch, limit + t0, (size_t)T, C, len,
                               (const float*)lstate, lstate);
        else
            hipLaunchKernelGGL(limit_kernel, dim3((C + 63) / 64), dim3(64), 0, st_chain, c->ybuf, c->ypitch, limit, C, T);
        HIPCHK(c, hipGetLastError());
    }
    HIPCHK(c, hipEventRecord(c->ev_join, st_corr));
    HIPCHK(c, hipStreamWaitEvent(c->stream, c->ev_join, 0));
    if (y_host) HIPCHK(c, hipMemcpy2DAsync(y_host, (size_t)T * sizeof(float), c->ybuf + YPRE, c->ypitch * sizeof(float), (size_t)T * sizeof(float), C, hipMemcpyDeviceToHost, c->stream));
    if (limit_host) HIPCHK(c, hipMemcpyAsync(limit_host, limit, n * sizeof(float), hipMemcpyDeviceToHost, c->stream));
    if (corr_host) HIPCHK(c, hipMemcpyAsync(corr_host, corr, 4 * n * sizeof(float), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return M17HIP_OK;
}

int m17hip_demod_front(m17hip_ctx* c, uint32_t C, uint32_t T, uint32_t flags)
{
    if (!c || C == 0 || T == 0 || C > c->maxC || T > c->maxT || (flags & ~M17HIP_FLAG_INVERT)) return M17HIP_EINVAL;
    GUARD(c);
    if (c->front_pending || !c->staged) return M17HIP_ESTATE;
    int r = begin_staged(c, C, T, flags, true);
    if (r) return r;
    c->front_pending = true;
    // The replay of the staged run's first segment (K2 from K5's state) needs the run in flight only up to its last K5 launch and its
    // carried tails — not its deferred decode, its consumers, the caller's fetch of its records or the launch of the next run from the
    // host: queued here, on the replay stream, it runs beside all of those.
    if (c->gate0_early && c->have_run && c->carryT && !c->profile) {
        const int q = c->slot;
        const SegPlan sp(c, T);
        for (hipEvent_t e : {c->ev_tail, c->ev_in_ready, c->ev_fir_[q][0], c->ev_dcd_[q][0]}) HIPCHK(c, hipStreamWaitEvent(c->side3, e, 0));
        hipLaunchKernelGGL(copy_prefix_f32_kernel, dim3(C), dim3(64), 0, c->side3, c->yalt, c->ybuf, c->ypitch);
        hipLaunchKernelGGL(copy_prefix_f32_kernel, dim3(C), dim3(64), 0, c->side3, c->halt, c->hbuf, c->ypitch);
        HIPCHK(c, hipGetLastError());
        if ((r = launch_gate_seg(c, sp, 0, c->side3, false, false, C, flags))) return r;
        HIPCHK(c, hipEventRecord(c->ev_gate_[q][0], c->side3));
        c->gate0_queued = true;
    }
    return M17HIP_OK;
}

int m17hip_demod_run(m17hip_ctx* c, uint32_t C, uint32_t T, uint32_t flags)
{
    if (!c || C == 0 || T == 0 || C > c->maxC || T > c->maxT || (flags & ~M17HIP_FLAG_INVERT)) return M17HIP_EINVAL;
    GUARD(c);
    int r;
    bool staged_run = false;
    const bool from_front = c->front_pending;
    if (c->front_pending) {   // the front end is already on its way (m17hip_demod_front): this call must be the run it was queued for
        if (C != c->frontC || T != c->frontT || flags != c->front_flags) return M17HIP_ESTATE;
        c->front_pending = false;
        staged_run = true;
    } else if (c->staged) {   // input staged by m17hip_upload_i16_async and friends: swap the slabs, queue the front end
        if ((r = begin_staged(c, C, T, flags, false))) return r;
        staged_run = true;
    }
    if (!c->uploaded) return M17HIP_ESTATE;
    if (c->have_run && C != c->lastC) return M17HIP_EINVAL;  // a continued stream keeps its channel count
    // The run is processed in segments.  K1 (throughput-bound, the whole chip) and K3 (latency-bound lone waves) of ALL segments
    // are queued on two side streams; the main stream runs K2 -> K5 per segment as soon as that segment's K1 and K3 are done,
    // so the front end of segment k+1 fills the issue slots K5 of segment k leaves idle (its tail above all).
    // Every K2 starts a fresh speculation from K5's own state, so a channel that had to drop it (forced unlock) carries the
    // limit filter itself only until the end of its segment.
    // Is this run the only thing in flight on the device?  (What a context can know of it: runs_overlap.)  Then a step lasts what its chain of launches lasts:
    // K3 and K1 take their latency forms, and the run starts with a ramp of short segments (key 33) — one batch at a time 23.65 -> 22.8 ms per
    // 4096 x 480 000; with batches in flight or a continued stream both cost more than they save.
    const bool alone = !staged_run && !runs_overlap(c);
    if (!staged_run) c->ramp_now = c->seg_ramp >= 0 ? (uint32_t)c->seg_ramp : (alone ? AUTO_RAMP : 0u);
    const SegPlan sp(c, T);
    const uint32_t nseg = sp.nseg;
    const int q = c->slot;
    if ((r = ensure_seg_events(c, q, nseg))) return r;
    auto& ev_fir = c->ev_fir_[q]; auto& ev_dcd = c->ev_dcd_[q]; auto& ev_gate = c->ev_gate_[q]; auto& ev_redo = c->ev_redo_[q]; auto& ev_seq = c->ev_seq_[q];
    const uint32_t ahead = c->front_ahead ? c->front_ahead : nseg;
    if (staged_run) {
        // K2 / K5 read the input slab too (snapshots, spliced-history FIR outputs); the y / h prefixes (the correlator ring and the limit
        // filter's history reach back into the previous run) come from the previous run's slabs, final now that its K5 is done
        HIPCHK(c, hipStreamWaitEvent(c->stream, c->ev_in_ready, 0));
        if (c->gate0_queued) {
            // (m17hip_demod_front has queued these copies on the replay stream, in front of the first segment's replay)
        } else if (c->carryT) {   // (the previous run ended by carrying its tails into its own prefixes: copy those)
            hipLaunchKernelGGL(copy_prefix_f32_kernel, dim3(C), dim3(64), 0, c->stream, c->yalt, c->ybuf, c->ypitch);
            hipLaunchKernelGGL(copy_prefix_f32_kernel, dim3(C), dim3(64), 0, c->stream, c->halt, c->hbuf, c->ypitch);
            HIPCHK(c, hipGetLastError());
        } else {
            HIPCHK(c, hipMemset2DAsync(c->ybuf, c->ypitch * sizeof(float), 0, YPRE * sizeof(float), C, c->stream));
            HIPCHK(c, hipMemset2DAsync(c->hbuf, c->ypitch * sizeof(float), 0, YPRE * sizeof(float), C, c->stream));
        }
    } else {
        HIPCHK(c, hipEventRecord(c->ev_fork, c->stream));
        HIPCHK(c, hipStreamWaitEvent(c->side, c->ev_fork, 0));
        HIPCHK(c, hipStreamWaitEvent(c->side2, c->ev_fork, 0));
        c->dcd_latency = c->dcd_form < 0 ? alone : c->dcd_form == 1;
        c->fir_latency = c->dcd_latency;
        c->front_segs = std::min(ahead, nseg);
        if ((r = gate_mode_for_run(c, sp))) return r;
        if ((r = launch_front_all(c, sp, C, flags))) return r;
    }
    // this run's record set: the one the run before the previous one wrote (its payload work is long through; waited for on the device)
    const int si = (c->sets[0].valid || c->sets[1].valid || c->have_run) ? (c->cur ^ 1) : c->cur;
    m17hip_ctx::RecSet& rs = c->sets[si];
    if (!rs.recs) {   // the second set appears with the second run
        HIPCHK(c, hipMalloc((void**)&rs.recs, (size_t)c->maxC * c->rec_cap_alloc * sizeof(FrameRec)));
        HIPCHK(c, hipMalloc((void**)&rs.rec_count, (size_t)c->maxC * sizeof(uint32_t)));
        HIPCHK(c, hipMemsetAsync(rs.rec_count, 0, (size_t)c->maxC * 4, c->stream));
    }
    if (c->defer_decode && !rs.defer_llr)   // the deferred-frame stores exist only where that mode is used (184 B per record slot)
        HIPCHK(c, hipMalloc((void**)&rs.defer_llr, (size_t)c->maxC * c->rec_cap_alloc * 46 * sizeof(uint32_t)));
    if (c->defer_decode && !c->defer_hist)
        HIPCHK(c, hipMalloc((void**)&c->defer_hist, (size_t)c->maxC * DEFER_HIST_WORDS * 64 * sizeof(uint32_t)));
    if (rs.valid) {   // the run that wrote it last: its payload work must be through (queued now if nobody asked for it)
        if (rs.pending && (r = flush_payload(c, false, true))) return r;
        HIPCHK(c, hipStreamWaitEvent(c->stream, rs.done, 0));
    }
    rs.valid = false;
    if (c->defer_evm) {   // the operation rows of the deferred EVM: 4 B per symbol of the longest run
        if (c->fold_pending) c->ev_par ^= 1;   // (the run before still has a fold pass to come: it keeps its rows, this run writes the other buffer)
        if (!c->ev_ops2[c->ev_par]) {
            c->ev_pitch = c->ev_pitch_override ? c->ev_pitch_override : ev_row_floats(c->maxT);
            HIPCHK(c, hipMalloc((void**)&c->ev_ops2[c->ev_par], (size_t)c->maxC * c->ev_pitch * sizeof(float)));
        }
    }
    c->dbg_waves = (c->profile || c->wave_times) ? C : 0;
    // K2 launches: the whole first segment from K5's state; every later segment AHEAD of K5 from the replay's own end state (replay
    // stream); and the REDO: for the channels that left the replay in segment k - 1 (a forced dcd.unlock()), the replay's state at the
    // end of segment k is re-derived from K5's boundary record, beside K5 of segment k on the replay stream.  It stores nothing — those
    // channels serve themselves in segment k (m17_wave_kernel.hpp) — and makes the replay of segment k + 1 good for them again; K5 never
    // waits for it.
    if (!c->bnd) {
        HIPCHK(c, hipMalloc((void**)&c->bnd, 2 * (size_t)c->maxC * sizeof(Boundary)));
        HIPCHK(c, hipMemsetAsync(c->bnd, 0, 2 * (size_t)c->maxC * sizeof(Boundary), c->stream));
    }
    uint32_t* const drop_of[2] = {c->dropped, c->dropped + c->maxC};   // by segment parity
    auto launch_gate = [&](uint32_t k, hipStream_t st, bool ahead, bool redo, bool redo_stores = false) -> int {
        if (k == 0 && c->gate0_queued) {   // m17hip_demod_front has queued this one on the replay stream, behind the prefix copies
            HIPCHK(c, hipStreamWaitEvent(st, ev_gate[0], 0));
            return M17HIP_OK;
        }
        return launch_gate_seg(c, sp, k, st, ahead, redo, C, flags, redo_stores);
    };
    // Redo policy (m17hip_tune key 20).  IN FRONT of K5: the redo stores the history of segment k for the channels that left the replay in
    // k - 1 before K5(k) starts — sixteen channels per instruction instead of one wave each carrying the filter itself through segment k
    // (3 dependent instructions per sample): fewer instructions, but 1-2 ms of replay latency on the K5 chain of every segment that follows
    // a drop.  BESIDE K5: the redo re-derives the replay's end state only, the channels concerned serve themselves through segment k, K5
    // never waits.  Measured: wherever the chain of K5 launches is what a step lasts — a continued stream, one batch at a time — the redo
    // beside K5 wins (24.7 against 28.0 ms, 26.3 against 28.9); with round 4's matched filter several independent batches in flight were
    // 1.4 % faster with the redo in front (NOTES 4.11), with round 5's they are 5 % slower (22.6 against 21.4: NOTES 5.5).  Default: beside.
    const bool redo_front = c->redo_form == 1;
    // one wave per channel, four waves per workgroup
    constexpr uint32_t wpb = 4;
    const dim3 grid((C + wpb - 1) / wpb), block(64 * wpb);
    // LDS: what the workgroup needs (31.8 KB for four waves), padded so that a CU holds FOUR of them and not five — the rest of
    // the CU (24 KB, 128 VGPRs per SIMD) is where a K2 wave or a K1 workgroup runs beside them without taking a K5 slot
    const size_t lds = std::max((size_t)wave_lds_words((int)wpb) * 4, c->seq_lds_bytes ? (size_t)c->seq_lds_bytes : (size_t)SEQ_LDS_BYTES_4);
    auto seq_params = [&](uint32_t k, uint32_t t0, uint32_t len) {
        SeqParams P{};
        P.h = c->hbuf + t0; P.final_h = c->final_h + (size_t)(k & 1u) * c->maxC * 4;
        P.dropped = drop_of[k & 1u];
        P.x = c->xbuf + t0; P.xpitch = c->xpitch; P.y = c->ybuf + t0; P.ypitch = c->ypitch;
        P.dcd_table = c->dcd_table; P.ticks_cap = c->ticks_cap; P.state = c->seq_state;
        P.recs = rs.recs; P.rec_cap = c->rec_cap; P.rec_count = rs.rec_count; P.overflow = rs.ovf;
        P.tables = c->tables; P.taps = c->taps; P.llr_edges = c->llr_edges;
        P.C = C; P.T = len; P.pos0 = c->pos + t0; P.tick_row0 = c->pos / TICK; P.flags = (flags & 1u) | (t0 ? 2u : 0u) | (std::min(k, 23u) << 8);
        P.kalman_order = c->kalman_order; P.channel_base = c->channel_base;
        P.level_gain = c->level_gain + (size_t)(c->kalman_order & 7u) * core::LEVEL_SCHED_N;
        P.diag_log = c->diag_cap ? c->diag_log : nullptr; P.diag_cap = c->diag_cap; P.diag_count = c->diag_count;
        P.defer = c->defer_decode ? rs.defer_llr : nullptr;
        if (c->defer_evm) { P.ev_ops = c->ev_ops2[c->ev_par]; P.ev_pitch = c->ev_pitch; P.ev_cursor_out = c->ev_cur + (size_t)(k + 1u == nseg ? 2u + (uint32_t)c->ev_par : (k & 1u)) * c->maxC; }
        P.bnd_out = c->bnd + (size_t)((k + 1u) & 1u) * c->maxC;
        P.truth_out = c->truth ? c->truth + (size_t)(k & 1u) * c->maxC : nullptr;
        P.dbg = (c->profile || c->wave_times) ? c->dbg : nullptr;
        return P;
    };
    HIPCHK(c, hipMemsetAsync(rs.ovf, 0, 4, c->stream));       // record overflow of THIS run
    HIPCHK(c, hipMemsetAsync(rs.ovf + 3, 0, 4, c->stream));   // channel-segments of THIS run that end with the carrier off (K5 counts)
    for (uint32_t k = 0; k < nseg; ++k) {
        const uint32_t t0 = sp.t0(k), len = sp.t0(k + 1) - t0;
        HIPCHK(c, hipStreamWaitEvent(c->stream, ev_fir[k], 0));
        HIPCHK(c, hipStreamWaitEvent(c->stream, ev_dcd[k], 0));
        if (k == 0 && c->front_first > 1 && ahead >= nseg) {   // the matched filter of the first `front_first` segments has the chip to itself
            const uint32_t last = std::min(c->front_first, nseg) - 1u;
            HIPCHK(c, hipStreamWaitEvent(c->stream, ev_fir[last], 0));
        }
        {
            // replay stream: ahead(k) -> redo(k) [after K5(k - 1): its flags and its state] -> ahead(k + 1) -> ...; the main stream only
            // ever waits for an `ahead`
            if (k == 0) {
                if ((r = launch_gate(0, c->stream, false, false))) return r;
                HIPCHK(c, hipEventRecord(ev_redo[0], c->stream));
                HIPCHK(c, hipStreamWaitEvent(c->side3, ev_redo[0], 0));
            } else if (redo_front) {
                HIPCHK(c, hipStreamWaitEvent(c->stream, ev_gate[k], 0));
                if ((r = launch_gate(k, c->stream, false, true, true))) return r;   // (flagged channels only: from their boundary records, history stored)
                HIPCHK(c, hipEventRecord(ev_redo[k], c->stream));
                HIPCHK(c, hipStreamWaitEvent(c->side3, ev_redo[k], 0));
            } else {
                HIPCHK(c, hipStreamWaitEvent(c->stream, ev_gate[k], 0));
                if (k + 1 < nseg) {
                    HIPCHK(c, hipStreamWaitEvent(c->side3, ev_seq[k - 1], 0));
                    if ((r = launch_gate(k, c->side3, false, true))) return r;
                }
            }
            if (k + 1 < nseg) {
                HIPCHK(c, hipStreamWaitEvent(c->side3, ev_fir[k + 1], 0));
                HIPCHK(c, hipStreamWaitEvent(c->side3, ev_dcd[k + 1], 0));
                if ((r = launch_gate(k + 1, c->side3, true, false))) return r;
                HIPCHK(c, hipEventRecord(ev_gate[k + 1], c->side3));
            }
        }
        TimedK tm(c, KT_SEQ);
        SeqParams P = seq_params(k, t0, len);
        P.dropped_in = (k > 0 && !redo_front) ? drop_of[(k - 1u) & 1u] : nullptr;   // (redo in front: nobody starts a segment off the replay)
#ifdef M17_TOOLS
        if (c->profile) tm.launch((demod_wave_kernel<4, true>), grid, block, lds, c->stream, P);
        else if (c->wave_times) tm.launch((demod_wave_kernel<4, false, true>), grid, block, lds, c->stream, P);
        else
#endif
        if (c->kalman_order == 3u) tm.launch((demod_wave_kernel<4, false, false, 3>), grid, block, lds, c->stream, P);   // (the default order: no call in the kernel)
        else tm.launch(demod_wave_kernel<4>, grid, block, lds, c->stream, P);
        HIPCHK(c, hipGetLastError());
        HIPCHK(c, hipEventRecord(ev_seq[k], c->stream));
        if (k + ahead < nseg && (r = launch_front_seg(c, sp, k + ahead, C, flags))) return r;
        if (c->gate_run && (r = launch_gated_fir(c, sp, k, C, flags))) return r;
    }
    HIPCHK(c, hipGetLastError());
    // the tails a run that continues in THESE slabs (input uploaded in place) will find as its prefixes; a staged run takes them from
    // here into the other slab pair itself.  (Before the deferred decode: the next staged run's first replay waits for these, not for that.)
    hipLaunchKernelGGL(carry_tail_kernel, dim3(C), dim3(64), 0, c->stream, c->xbuf, c->xpitch, c->ybuf, c->ypitch, C, T);
    hipLaunchKernelGGL(carry_tail_f32_kernel, dim3(C), dim3(64), 0, c->stream, c->hbuf, c->ypitch, T);
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipEventRecord(c->ev_tail, c->stream));
    c->gate0_queued = false;
    // ---- the end of the run on the MAIN stream: what the next run's chain needs of this one — the one or two deferred costs the state still
    //      names (settle_tail_kernel).  The rest of the run's EVM fold (the operations of its last two segments: 0.8 ms of one dependent chain
    //      per channel that nothing in the demodulator reads) is NOT made here: see fold_pending
    if (c->defer_decode) {
        SettleParams S{rs.recs, c->rec_cap, rs.defer_llr, c->tables, c->seq_state, C, EvParams{nullptr, 0, nullptr, nullptr, 0, nullptr, C, nullptr, 0u}};
        hipLaunchKernelGGL(settle_tail_kernel, dim3(C), dim3(64), SETTLE_LDS_BYTES, c->stream, S);
        HIPCHK(c, hipGetLastError());
    }
    if (c->fold_pending && (r = flush_fold(c))) return r;   // (the run before: nothing of this run took its last pass along — one segment, no replay ahead)
    c->fold_pending = c->defer_evm;
    c->fold_ops = c->ev_ops2[c->ev_par]; c->fold_C = C; c->fold_end = c->ev_cur + (2 + (size_t)c->ev_par) * c->maxC;
    HIPCHK(c, hipEventRecord(rs.chain, c->stream));
    HIPCHK(c, hipEventRecord(c->ev_end[q], c->stream));
    { std::lock_guard<std::mutex> lk(g_runs.mu); c->last_end = c->ev_end[q]; }
    c->slot_used[q] = true;
    c->pos += T;
    c->inplace_after_run = false;
    c->lastC = C; c->lastT = T; c->runT = T; c->last_nseg = nseg;
    c->have_run = true;
    rs.valid = true; rs.C = C; rs.rec_cap = c->rec_cap; rs.nseg = nseg;
    rs.pending = true; rs.bert = c->bert; rs.pkt = c->pkt_cap != 0;
    c->cur = si;
    c->sel_back = 0;
    // The run's payload work — the frames K5 did not decode itself, then the consumers — is queued when its results are asked for.  At once
    // where that costs nothing or is needed: a context that does not stream (the work goes to the main stream, behind the run, as up to round 5),
    // and with the diagnostic log on (ONE store, which the deferred decode patches: the next run waits for it).
    if (!c->streams() || c->diag_cap) {
        c->fold_with_decode = c->fold_pending && c->defer_decode && c->pay() == c->stream;   // (beside the decode, as up to round 5)
        if ((r = flush_payload(c))) return r;
        c->fold_with_decode = false;
        if (c->diag_cap) {
            HIPCHK(c, hipStreamWaitEvent(c->stream, rs.done, 0));
            if ((r = flush_fold(c))) return r;
        }
    }
    return M17HIP_OK;
}

// The record set the fetch family names (m17hip_frames_select): the latest run's, or the run's before it; nullptr when that run's
// records are not there (no such run since the reset, or the layout was changed under them: m17hip_tune key 8).
static m17hip_ctx::RecSet* selected_set(m17hip_ctx* c)
{
    m17hip_ctx::RecSet& rs = c->sets[c->cur ^ (int)(c->sel_back & 1u)];
    return rs.valid ? &rs : nullptr;
}

// Offsets + (optionally) the dense copy in one pass and ONE stream synchronisation — of the PAYLOAD stream: behind the deferred decode
// and the consumers of the run concerned, beside whatever the main stream has been given since.  *count = records the run produced;
// at most `cap` of them are written.  M17HIP_EOVERFLOW: a channel outran its record slots during the run;
// M17HIP_ETRUNC: more records than `cap`.
static int compact_into(m17hip_ctx* c, FrameRec* dev_out, uint64_t cap, uint64_t* count)
{
    m17hip_ctx::RecSet* rs = selected_set(c);
    if (!rs) return M17HIP_ESTATE;
    if (int fr = flush_payload(c, true)) return fr;
    const uint32_t C = rs->C;
    if (dev_out && dev_out != c->compact && c->sel_back == 0 && c->pay() != c->stream) {
        // a device destination of the caller's: whatever the caller queued on the main stream for it (a fill, its previous consumer) comes first —
        // for the LATEST run that costs nothing (the payload work waits for that run's end on the main stream anyway); with a newer run queued
        // (m17hip_frames_select(ctx, 1)) the destination must be ready when the call is made (include/m17hip.h)
        HIPCHK(c, hipEventRecord(c->ev_dst, c->stream));
        HIPCHK(c, hipStreamWaitEvent(c->pay(), c->ev_dst, 0));
    }
    {
        Timed tm(c, KT_COMPACT, c->pay());
        hipLaunchKernelGGL(rec_offsets_kernel, dim3(1), dim3(256), 0, c->pay(), rs->rec_count, rs->rec_cap, c->rec_offsets, C);
        HIPCHK(c, hipGetLastError());
        if (dev_out) {
            hipLaunchKernelGGL(compact_kernel, dim3(C), dim3(64), 0, c->pay(), rs->recs, rs->rec_cap, rs->rec_count, c->rec_offsets, dev_out, cap, C);
            HIPCHK(c, hipGetLastError());
        }
    }
    uint64_t total = 0;
    uint32_t ovf[4] = {0, 0, 0, 0};
    HIPCHK(c, hipMemcpyAsync(&total, c->rec_offsets + C, 8, hipMemcpyDeviceToHost, c->pay()));
    HIPCHK(c, hipMemcpyAsync(ovf, rs->ovf, 16, hipMemcpyDeviceToHost, c->pay()));
    HIPCHK(c, hipStreamSynchronize(c->pay()));
    if (count) *count = total;
    c->off_segs_prev = ovf[3]; c->chan_segs_prev = rs->C * std::max(1u, rs->nseg);   // (what the next run's gate-aware choice looks at)
    if (ovf[0]) return M17HIP_EOVERFLOW;
    if (dev_out && total > cap) return M17HIP_ETRUNC;
    return M17HIP_OK;
}

int m17hip_frames_select(m17hip_ctx* c, uint32_t back)
{
    if (!c || back > 1) return M17HIP_EINVAL;
    if (!c->sets[c->cur ^ (int)back].valid) return M17HIP_ESTATE;
    c->sel_back = back;
    return M17HIP_OK;
}

int m17hip_frames_count(m17hip_ctx* c, uint64_t* total)
{
    if (!c || !total) return M17HIP_EINVAL;
    GUARD(c);
    return compact_into(c, nullptr, 0, total);
}

int m17hip_frames_compact_device(m17hip_ctx* c, m17_frame_rec* recs_dev, uint64_t capacity, uint64_t* count)
{
    if (!c || !recs_dev) return M17HIP_EINVAL;
    GUARD(c);
    return compact_into(c, (FrameRec*)recs_dev, capacity, count);
}

int m17hip_frames_fetch(m17hip_ctx* c, m17_frame_rec* recs_host, uint64_t capacity, uint64_t* count)
{
    if (!c || !recs_host) return M17HIP_EINVAL;
    GUARD(c);
    const m17hip_ctx::RecSet* rs = selected_set(c);
    if (!rs) return M17HIP_ESTATE;
    // one compaction into the context's dense buffer; it is sized for the caller's capacity (records beyond it are not
    // wanted anyway), so a second pass is never needed
    const uint64_t want = std::max<uint64_t>(std::min<uint64_t>(capacity, (uint64_t)rs->C * rs->rec_cap), 1024);
    if (want > c->compact_cap) {
        HIPCHK(c, hipStreamSynchronize(c->pay()));   // (nothing still reads the old one)
        free_dev(c->compact, &c->last_hip); c->compact_cap = 0;
        HIPCHK(c, hipMalloc((void**)&c->compact, (size_t)want * sizeof(FrameRec)));
        c->compact_cap = want;
    }
    uint64_t total = 0;
    const int r = compact_into(c, c->compact, std::min<uint64_t>(capacity, c->compact_cap), &total);
    if (r && r != M17HIP_EOVERFLOW && r != M17HIP_ETRUNC) return r;
    const uint64_t n = std::min(total, capacity);
    if (n) HIPCHK(c, hipMemcpy(recs_host, c->compact, (size_t)n * sizeof(FrameRec), hipMemcpyDeviceToHost));
    if (count) *count = total;
    return r;
}

int m17hip_diag_fetch(m17hip_ctx* c, m17_diag* diag_host, uint32_t C)
{
    if (!c || !diag_host || C == 0 || C > c->maxC) return M17HIP_EINVAL;
    GUARD(c);
    if (int fr = flush_fold(c)) return fr;        // m17_diag::evm of the latest run: its last fold pass, if nothing has made it yet
    if (c->front_pending && c->gate0_queued)      // (... or the replay m17hip_demod_front queued for the next run is making it: wait for that launch)
        HIPCHK(c, hipStreamWaitEvent(c->stream, c->ev_gate_[c->slot][0], 0));
    HIPCHK(c, hipMemcpy2DAsync(diag_host, sizeof(Diag), &c->seq_state[0].cold.diag, sizeof(SeqState), sizeof(Diag), C, hipMemcpyDeviceToHost,
                               c->stream));
    uint32_t ovf[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    HIPCHK(c, hipMemcpyAsync(ovf, c->overflow, 32, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return (ovf[2] | ovf[6]) ? M17HIP_EOVERFLOW : M17HIP_OK;   // (a channel outran its row of deferred EVM operations: every field but `evm` is right)
}

int m17hip_diag_log_fetch(m17hip_ctx* c, m17_diag* log_host, uint32_t* counts_host, uint32_t C, uint32_t capacity)
{
    if (!c || !log_host || !counts_host || C == 0 || C > c->maxC || capacity == 0) return M17HIP_EINVAL;
    GUARD(c);
    if (!c->diag_cap || !c->sets[c->cur].valid) return M17HIP_ESTATE;
    if (int fr = flush_fold(c)) return fr;
    if (int fr = flush_payload(c)) return fr;
    HIPCHK(c, hipStreamSynchronize(c->pay()));   // (the deferred decode has put the costs into the log's entries)
    HIPCHK(c, hipMemcpyAsync(counts_host, c->diag_count, (size_t)C * 4, hipMemcpyDeviceToHost, c->stream));
    const uint32_t n = std::min(capacity, c->diag_cap);
    HIPCHK(c, hipMemcpy2DAsync(log_host, (size_t)capacity * sizeof(Diag), c->diag_log, (size_t)c->diag_cap * sizeof(Diag), (size_t)n * sizeof(Diag), C,
                               hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    uint32_t ovf[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    HIPCHK(c, hipMemcpy(ovf, c->overflow, 32, hipMemcpyDeviceToHost));
    if (ovf[2] | ovf[6]) return M17HIP_EOVERFLOW;   // (deferred EVM operations were dropped: see m17hip_diag_fetch)
    bool trunc = false;
    for (uint32_t i = 0; i < C; ++i) trunc = trunc || counts_host[i] > n;
    return trunc ? M17HIP_ETRUNC : M17HIP_OK;
}

int m17hip_lsf_info(m17hip_ctx* c, const uint8_t* lsf30_host, uint32_t n, m17_lsf_info* out_host)
{
    if (!c || !lsf30_host || !out_host || n == 0) return M17HIP_EINVAL;
    GUARD(c);
    static_assert(sizeof(LsfInfo) == sizeof(m17_lsf_info) && sizeof(LsfInfo) == 32, "m17_lsf_info layout");
    const size_t in_b = round_up((size_t)n * 30, 256);
    int r = ensure_scratch(c, in_b + (size_t)n * sizeof(LsfInfo));
    if (r) return r;
    uint8_t* din = reinterpret_cast<uint8_t*>(c->scratch);
    LsfInfo* dout = reinterpret_cast<LsfInfo*>(din + in_b);
    HIPCHK(c, hipMemcpyAsync(din, lsf30_host, (size_t)n * 30, hipMemcpyHostToDevice, c->stream));
    hipLaunchKernelGGL(lsf_info_kernel, dim3((n + 63) / 64), dim3(64), 0, c->stream, din, n, dout);
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipMemcpyAsync(out_host, dout, (size_t)n * sizeof(LsfInfo), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return M17HIP_OK;
}

int m17hip_bert_stats(m17hip_ctx* c, m17_bert_stat* stats_host, uint32_t C)
{
    if (!c || !stats_host || C == 0 || C > c->maxC) return M17HIP_EINVAL;
    GUARD(c);
    if (!c->bert) return M17HIP_ESTATE;
    if (int fr = flush_payload(c)) return fr;
    std::vector<BertState> tmp(C);
    HIPCHK(c, hipMemcpyAsync(tmp.data(), c->bert_state, (size_t)C * sizeof(BertState), hipMemcpyDeviceToHost, c->pay()));   // (behind the consumers of the runs queued so far)
    HIPCHK(c, hipStreamSynchronize(c->pay()));
    for (uint32_t i = 0; i < C; ++i) {
        stats_host[i].bits = tmp[i].bit_count; stats_host[i].errors = tmp[i].err_count;
        stats_host[i].synced = tmp[i].synced; stats_host[i].frames = tmp[i].frames;
    }
    return M17HIP_OK;
}

int m17hip_packets_feed(m17hip_ctx* c, const m17_frame_rec* recs_host, const uint32_t* counts_host, uint32_t C, uint32_t pitch)
{
    if (!c || !recs_host || !counts_host || C == 0 || C > c->maxC || pitch == 0) return M17HIP_EINVAL;
    GUARD(c);
    if (!c->pkt_cap) return M17HIP_ESTATE;
    if (int fr = flush_payload(c)) return fr;
    const size_t rec_b = round_up((size_t)C * pitch * sizeof(FrameRec), 256);
    int r = ensure_scratch(c, rec_b + (size_t)C * 4);
    if (r) return r;
    FrameRec* drec = reinterpret_cast<FrameRec*>(c->scratch);
    uint32_t* dcnt = reinterpret_cast<uint32_t*>(reinterpret_cast<char*>(c->scratch) + rec_b);
    // (the payload stream: where the runs' own packet consumer works on the same reassembly state)
    HIPCHK(c, hipMemcpyAsync(drec, recs_host, (size_t)C * pitch * sizeof(FrameRec), hipMemcpyHostToDevice, c->pay()));
    HIPCHK(c, hipMemcpyAsync(dcnt, counts_host, (size_t)C * 4, hipMemcpyHostToDevice, c->pay()));
    const int ps = c->cur;   // (the store of the latest run's set: what m17hip_packets_fetch names unless m17hip_frames_select says otherwise)
    HIPCHK(c, hipMemsetAsync(c->pkt_count2 + ps, 0, 4, c->pay()));
    hipLaunchKernelGGL(packet_asm_kernel, dim3((C + 63) / 64), dim3(64), 0, c->pay(), drec, pitch, dcnt, (PacketState*)c->pkt_state, C,
                       (PacketRec*)c->pkt_recs2[ps], c->pkt_cap, c->pkt_count2 + ps, c->channel_base);
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipStreamSynchronize(c->pay()));   // the host buffers may go away
    c->pkt_fed = true;
    return M17HIP_OK;
}

int m17hip_packets_fetch(m17hip_ctx* c, m17_packet_rec* recs_host, uint32_t capacity, uint32_t* count)
{
    if (!c || !count || (capacity && !recs_host)) return M17HIP_EINVAL;
    GUARD(c);
    static_assert(sizeof(PacketRec) == sizeof(m17_packet_rec) && sizeof(PacketRec) == 864, "m17_packet_rec layout");
    if (!c->pkt_cap || !(c->have_run || c->pkt_fed)) return M17HIP_ESTATE;
    const int ps = c->cur ^ (int)(c->sel_back & 1u);   // the packets completed by the SELECTED run (m17hip_frames_select)
    if (c->sel_back && !c->sets[ps].valid) return M17HIP_ESTATE;
    if (int fr = flush_payload(c, true)) return fr;
    uint32_t total = 0;
    HIPCHK(c, hipMemcpyAsync(&total, c->pkt_count2 + ps, 4, hipMemcpyDeviceToHost, c->pay()));
    HIPCHK(c, hipStreamSynchronize(c->pay()));
    *count = total;
    const uint32_t stored = std::min(total, c->pkt_cap);
    std::vector<PacketRec> tmp(stored);
    if (stored) HIPCHK(c, hipMemcpy(tmp.data(), c->pkt_recs2[ps], (size_t)stored * sizeof(PacketRec), hipMemcpyDeviceToHost));
    std::sort(tmp.begin(), tmp.end(), [](const PacketRec& a, const PacketRec& b) { return a.channel != b.channel ? a.channel < b.channel : a.seq < b.seq; });
    const uint32_t n = std::min(stored, capacity);
    if (n) std::memcpy(recs_host, tmp.data(), (size_t)n * sizeof(PacketRec));
    return total > c->pkt_cap ? M17HIP_EOVERFLOW : M17HIP_OK;
}

int m17hip_set_kalman_order(m17hip_ctx* c, int order)
{
    if (!c || order < 0 || order > 7) return M17HIP_EINVAL;
    c->kalman_order = (uint32_t)order;
    return M17HIP_OK;
}

int m17hip_set_channel_base(m17hip_ctx* c, uint32_t channel_base)
{
    if (!c) return M17HIP_EINVAL;
    c->channel_base = channel_base;
    return M17HIP_OK;
}

int m17hip_kalman_trace(m17hip_ctx* c, const float* z_host, const uint32_t* dt_host, uint32_t rows, uint32_t n, int wrap, float z0, int order,
                        float* out_host)
{
    if (!c || !z_host || !dt_host || !out_host || rows == 0 || n == 0 || order < 0 || order > 7 || (wrap != 0 && wrap != 10 && wrap != -1)) return M17HIP_EINVAL;
    GUARD(c);
    const size_t cnt = (size_t)rows * n;
    const size_t o1 = round_up(cnt * 4, 256), o2 = o1 + round_up(cnt * 4, 256);
    int r = ensure_scratch(c, o2 + cnt * 24);
    if (r) return r;
    char* b = (char*)c->scratch;
    HIPCHK(c, hipMemcpyAsync(b, z_host, cnt * 4, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(b + o1, dt_host, cnt * 4, hipMemcpyHostToDevice, c->stream));
    hipLaunchKernelGGL(kalman_kernel, dim3((rows + 63) / 64), dim3(64), 0, c->stream, (const float*)b, (const uint32_t*)(b + o1), rows, n, wrap, z0,
                       (uint32_t)order, (float*)(b + o2), c->level_gain + (size_t)order * core::LEVEL_SCHED_N);
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipMemcpyAsync(out_host, b + o2, cnt * 24, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return M17HIP_OK;
}

int m17hip_upload_wait(m17hip_ctx* c)
{
    if (!c) return M17HIP_EINVAL;
    GUARD(c);
    if (c->ev_copy) HIPCHK(c, hipEventSynchronize(c->ev_copy));
    return M17HIP_OK;
}

// ---- multi-GPU: gather of the frame records over RCCL ------------------------------------------------------------------
int m17hip_comm_get_id(void* id128)
{
    if (!id128) return M17HIP_EINVAL;
    const Rccl& R = rccl();
    if (!R.ok) return M17HIP_ECOMM;
    static_assert(sizeof(ncclUniqueId) == M17HIP_COMM_ID_BYTES, "RCCL unique id size");
    ncclUniqueId id;
    if (R.GetUniqueId(&id) != ncclSuccess) return M17HIP_ECOMM;
    std::memcpy(id128, &id, sizeof(id));
    return M17HIP_OK;
}

int m17hip_comm_create(m17hip_ctx* c, const void* id128, int rank, int nranks, m17hip_comm** out)
{
    if (!c || !id128 || !out || nranks < 1 || rank < 0 || rank >= nranks) return M17HIP_EINVAL;
    GUARD(c);
    const Rccl& R = rccl();
    if (!R.ok) return M17HIP_ECOMM;
    m17hip_comm* m = new (std::nothrow) m17hip_comm();
    if (!m) return M17HIP_ENOMEM;
    m->device = c->device; m->rank = rank; m->nranks = nranks;
    ncclUniqueId id;
    std::memcpy(&id, id128, sizeof(id));
    const ncclResult_t r = R.CommInitRank(&m->comm, nranks, id, rank);
    if (r != ncclSuccess) { c->last_hip = 0x10000 | (int)r; delete m; return M17HIP_ECOMM; }   // (no communicator to ask: the ncclResult_t is left in m17hip_last_hip_error, | 0x10000)
    hipError_t e = hipMalloc((void**)&m->counts_dev, 2 * (size_t)nranks * sizeof(uint64_t));
    // (no slot looks like a word of call 1.  ON the stream the gather works on, and waited for: a plain hipMemset runs on the default stream and is not waited
    //  for — with the library's own non-blocking streams it could land behind the first exchange's words: one first gather in a few hundred then read 0xFF..
    //  where a peer's word had been and answered ECOMM.  Found by tests/test_gpu_gather_ranks.py, four ranks, 2 of 25 suite runs)
    if (e == hipSuccess) e = hipMemsetAsync(m->counts_dev, 0xFF, 2 * (size_t)nranks * sizeof(uint64_t), c->pay());
    if (e == hipSuccess) e = hipStreamSynchronize(c->pay());
    if (e == hipSuccess) e = hipHostMalloc((void**)&m->words_host, (2 + 2 * (size_t)nranks) * sizeof(uint64_t), hipHostMallocDefault);
    if (e != hipSuccess) { c->last_hip = (int)e; free_dev(m->counts_dev); R.CommDestroy(m->comm); delete m; return M17HIP_ENOMEM; }
    *out = m;
    return M17HIP_OK;
}

void m17hip_comm_destroy(m17hip_comm* m)
{
    if (!m) return;
    int prev = -1;
    if (hipGetDevice(&prev) != hipSuccess) prev = -1;
    hipSetDevice(m->device);
    free_dev(m->counts_dev);
    free_dev(m->gathered);
    if (m->words_host) { (void)hipHostFree(m->words_host); m->words_host = nullptr; }
    if (m->comm && !m->dead) rccl().CommDestroy(m->comm);   // (a communicator that was given up is not waited for)
    if (prev >= 0) hipSetDevice(prev);
    delete m;
}

// Every rank makes the same sequence of collective calls whatever goes wrong locally: a failure travels as a status inside the words
// that are exchanged anyway, and all ranks return together (a rank that left early would leave its peers waiting inside RCCL).
//   exchange 1 (all-gather, two words per rank):  [0] = record count (40 bits) | call serial (16 bits) | status (8 bits: -code)
//                                                 [1] = the ROOT's staging capacity in records (0 from the other ranks)
//       the serial makes a stale word recognisable: a rank whose own word could not be written to the device (its HIP calls fail)
//       still joins the all-gather, and what its peers then read in its slot is a word of an EARLIER exchange
//   exchange 2 (all-gather, one word per rank), ALWAYS: "I have read exchange 1 and am ready for the records" — the root grows its
//       staging first when the gathered set does not fit and says whether that worked; a rank that could NOT read exchange 1 (it does
//       not know the counts and cannot take part in exchange 3) says so here, where up to round 5 it returned and left its peers
//       waiting in exchange 3.  The words carry a phase tag in the status byte's place (0xA5: no status of exchange 1 looks like it,
//       and read as one it is an error) and the serial again; EVERY rank looks at EVERY slot — a slot without the tag and this call's
//       serial is a word that could not be written, whoever's it is — so that all ranks leave together or go on together
//   exchange 3: grouped ncclSend / ncclRecv of the exact record sets, rank after rank = global (channel, seq) order; ranks without
//       records are skipped on both sides
// What no exchange of words can close (a rank that reads exchange 1 but not exchange 2; a peer process that died) is bounded in TIME:
// every wait of this function is a bounded one (m17hip_tune key 31, default 120 s); a wait that runs out gives the communicator up
// (ncclCommAbort where the library has it) and the call — and every later call through that communicator — returns M17HIP_ECOMM.
static void comm_give_up(m17hip_comm* m, int code)
{
    m->last_rccl = code;
    if (m->dead) return;
    m->dead = true;
    const Rccl& R = rccl();
    if (m->comm && R.CommAbort) { R.CommAbort(m->comm); m->comm = nullptr; }   // (without it the communicator is left alone: destroying it would wait for the exchange)
}
// hipStreamSynchronize with a deadline: 0 = done, M17HIP_ECOMM = not in time (communicator given up), M17HIP_EHIP = the stream reports an error
static int comm_wait(m17hip_ctx* c, m17hip_comm* m)
{
    if (c->gather_timeout_ms == 0) {
        const hipError_t e = hipStreamSynchronize(c->pay());
        if (e != hipSuccess) { c->last_hip = (int)e; return M17HIP_EHIP; }
        return M17HIP_OK;
    }
    timespec t0;
    clock_gettime(CLOCK_MONOTONIC, &t0);
    for (uint64_t spins = 0;; ++spins) {
        const hipError_t e = hipStreamQuery(c->pay());
        if (e == hipSuccess) return M17HIP_OK;
        if (e != hipErrorNotReady) { c->last_hip = (int)e; return M17HIP_EHIP; }
        (void)hipGetLastError();
        if ((spins & 63) == 63) {
            timespec t;
            clock_gettime(CLOCK_MONOTONIC, &t);
            const double ms = (t.tv_sec - t0.tv_sec) * 1e3 + (t.tv_nsec - t0.tv_nsec) * 1e-6;
            if (ms > (double)c->gather_timeout_ms) { comm_give_up(m, (int)ncclSystemError); return M17HIP_ECOMM; }
            if (ms > 2.0) { timespec nap{0, 50000}; nanosleep(&nap, nullptr); }   // (a gather is through in well under that: only a stuck one gets here)
        }
    }
}

static int gather_frames_impl(m17hip_ctx* c, m17hip_comm* m, int root, m17_frame_rec* recs_host, uint64_t capacity, uint64_t* counts_host,
                              uint64_t* total_out, bool dest_is_device)
{
    if (!c || !m || m->device != c->device || root < 0 || root >= m->nranks || (m->rank == root && capacity && !recs_host)) return M17HIP_EINVAL;
    GUARD(c);
    if (m->dead) return M17HIP_ECOMM;
    const Rccl& R = rccl();
    const bool is_root = m->rank == root;
    auto hip_code = [&](hipError_t e) { c->last_hip = (int)e; return e == hipErrorOutOfMemory ? M17HIP_ENOMEM : M17HIP_EHIP; };
    auto comm_failed = [&](ncclResult_t q) { comm_give_up(m, (int)q); return M17HIP_ECOMM; };   // a collective call itself failed: nothing more can be agreed on
    // 1. this rank's records, dense and (channel, seq)-ordered, in the context's compaction buffer
    uint64_t mine = 0;
    int local = selected_set(c) ? M17HIP_OK : M17HIP_ESTATE;
    bool overflow = false;
    if (local == M17HIP_OK) {
        int r = c->gather_fault == 1 ? M17HIP_EHIP : compact_into(c, c->compact, c->compact_cap, &mine);
        // the dense buffer must hold ALL of this rank's records before anything is sent from it, whatever the first pass said
        // (an overflowed run reports EOVERFLOW before the truncation is looked at)
        if ((r == M17HIP_OK || r == M17HIP_ETRUNC || r == M17HIP_EOVERFLOW) && mine > c->compact_cap) {
            free_dev(c->compact, &c->last_hip); c->compact_cap = 0;
            const uint64_t want = std::max<uint64_t>(mine + mine / 8, 1024);
            const hipError_t e = hipMalloc((void**)&c->compact, (size_t)want * sizeof(FrameRec));
            if (e != hipSuccess) r = hip_code(e);
            else { c->compact_cap = want; r = compact_into(c, c->compact, c->compact_cap, &mine); }
        }
        overflow = r == M17HIP_EOVERFLOW;
        if (r && !overflow) { local = r; mine = 0; }
    }
    if (is_root && !m->gathered && local == M17HIP_OK) {   // a first staging buffer (grown below when a gathered set outgrows it)
        const uint64_t want = std::max<uint64_t>(2 * mine * (uint64_t)m->nranks, 1024);
        if (c->gather_fault != 2 && hipMalloc((void**)&m->gathered, (size_t)want * sizeof(FrameRec)) == hipSuccess) m->gathered_cap = want;
        else { m->gathered = nullptr; m->gathered_cap = 0; }
    }
    // 2. exchange 1
    const uint64_t serial = (uint64_t)(++m->serial & 0xFFFFu);
    constexpr uint64_t COUNT_MASK = (1ull << 40) - 1;
    if (mine > COUNT_MASK && !local) { local = M17HIP_EINVAL; mine = 0; }   // (2^40 records: not a real case, but the word has no room for more)
    uint64_t* const word = m->words_host;        // (pinned, owned by the communicator: see there)
    uint64_t* const words = m->words_host + 2;
    word[0] = (mine & COUNT_MASK) | (serial << 40) | ((uint64_t)(uint8_t)(-local) << 56);
    word[1] = is_root ? m->gathered_cap : 0ull;
    const size_t words_n = 2 * (size_t)m->nranks;
    int unread = M17HIP_OK;   // this rank could not read exchange 1
    {
        hipError_t e = hipMemcpyAsync(m->counts_dev + 2 * m->rank, word, 16, hipMemcpyHostToDevice, c->pay());
        if (e != hipSuccess && !local) local = hip_code(e);     // (our slot keeps the previous call's word: its serial gives it away)
        const ncclResult_t q = R.AllGather(m->counts_dev + 2 * m->rank, m->counts_dev, 2, ncclUint64, m->comm, c->pay());
        if (q != ncclSuccess) return comm_failed(q);
        e = c->gather_fault == 4 ? hipErrorUnknown : hipMemcpyAsync(words, m->counts_dev, words_n * 8, hipMemcpyDeviceToHost, c->pay());
        if (e != hipSuccess) unread = hip_code(e);
        const int w = comm_wait(c, m);
        if (w == M17HIP_ECOMM) return w;     // not in time: the communicator is given up
        if (w && !unread) unread = w;
    }
    std::vector<uint64_t> counts((size_t)m->nranks, 0);
    uint64_t total = 0;
    int remote = M17HIP_OK;
    if (!unread) {
        for (int k = 0; k < m->nranks; ++k) {
            const uint64_t w = words[2 * (size_t)k];
            int code = -(int)(w >> 56);
            if (((w >> 40) & 0xFFFFu) != serial) code = M17HIP_ECOMM;   // a stale word: that rank could not deliver this call's
            counts[k] = code ? 0 : (w & COUNT_MASK);
            total += counts[k];
            if (code && !remote) remote = code;
        }
        if (counts_host) std::memcpy(counts_host, counts.data(), counts.size() * 8);
        if (total_out) *total_out = total;
    }
    // 3. exchange 2: every rank, whatever it knows by now.  The root grows its staging when the gathered set does not fit it
    //    (only when the records are going to travel: every rank that read exchange 1 comes to the same conclusion about that)
    int rc = unread;
    if (!unread && !local && !remote && is_root && (total > m->gathered_cap || c->gather_fault == 3)) {   // (fault 3 on the root: grown in any case)
        free_dev(m->gathered, &c->last_hip); m->gathered_cap = 0;
        const uint64_t want = std::max<uint64_t>(total + total / 8, 1024);
        const hipError_t e = c->gather_fault == 2 ? hipErrorOutOfMemory : hipMalloc((void**)&m->gathered, (size_t)want * sizeof(FrameRec));
        if (e != hipSuccess) rc = hip_code(e); else m->gathered_cap = want;
    }
    constexpr uint64_t PHASE2 = 0xA5ull << 56;
    {
        word[0] = PHASE2 | (serial << 40) | (uint64_t)(uint8_t)(-rc);
        const hipError_t ew = c->gather_fault == 3 ? hipErrorUnknown : hipMemcpyAsync(m->counts_dev + 2 * m->rank, word, 8, hipMemcpyHostToDevice, c->pay());
        if (ew != hipSuccess && !rc) rc = hip_code(ew);          // (our slot keeps a word of exchange 1: no tag — every rank sees that)
        const ncclResult_t q = R.AllGather(m->counts_dev + 2 * m->rank, m->counts_dev, 2, ncclUint64, m->comm, c->pay());
        if (q != ncclSuccess) return comm_failed(q);
        hipError_t e = c->gather_fault == 5 ? hipErrorUnknown : hipMemcpyAsync(words, m->counts_dev, words_n * 8, hipMemcpyDeviceToHost, c->pay());
        const int w = comm_wait(c, m);
        if (w == M17HIP_ECOMM) return w;
        // this rank cannot read what was agreed on: it cannot know whether the records travel, so it takes no part in exchange 3.  If they
        // do, its peers' waits run out (comm_wait) — the one case that costs them the communicator
        if (e != hipSuccess) return hip_code(e);
        if (w) return w;
    }
    if (unread) return unread;
    if (local) return local;
    if (remote) return M17HIP_ECOMM;   // some other rank could not deliver its records: nobody sends, everybody returns
    bool all_ok = true;
    for (int k = 0; k < m->nranks; ++k) {
        const uint64_t w = words[2 * (size_t)k];
        if ((w & (0xFFull << 56)) != PHASE2 || ((w >> 40) & 0xFFFFu) != serial || (w & 0xFFu)) all_ok = false;
    }
    if (rc) return rc;
    if (!all_ok) return M17HIP_ECOMM;   // the root has no room for the gathered set, a rank could not read the counts, or some rank's word did not arrive: nobody sends
    // 4. the records travel to the root with their exact sizes, rank after rank = global channel order
    if (is_root) {
        ncclResult_t q = R.GroupStart();
        uint64_t off = 0;
        for (int k = 0; k < m->nranks && q == ncclSuccess; ++k) {
            if (k != root && counts[k]) q = R.Recv(m->gathered + off, (size_t)counts[k] * sizeof(FrameRec), ncclUint8, k, m->comm, c->pay());
            off += counts[k];
        }
        const ncclResult_t qe = R.GroupEnd();   // the group is closed whatever happened inside it
        if (q == ncclSuccess) q = qe;
        if (q != ncclSuccess) return comm_failed(q);
        off = 0;
        hipError_t he = hipSuccess;
        for (int k = 0; k < m->nranks; ++k) {   // the root's own share: a plain copy, outside the group
            if (k == root && mine) he = hipMemcpyAsync(m->gathered + off, c->compact, (size_t)mine * sizeof(FrameRec), hipMemcpyDeviceToDevice, c->pay());
            off += counts[k];
        }
        if (he != hipSuccess) return hip_code(he);
        if (const int w = comm_wait(c, m)) return w;
        // the caller's buffer is written only now, when nothing on the stream depends on a peer any more: a call that ran out of time
        // never leaves a copy into the caller's memory behind
        const uint64_t n = std::min(total, capacity);
        if (n) {
            if (dest_is_device && c->sel_back == 0 && c->pay() != c->stream) {   // (a device destination: behind the caller's main-stream work on it, as in compact_into)
                HIPCHK(c, hipEventRecord(c->ev_dst, c->stream));
                HIPCHK(c, hipStreamWaitEvent(c->pay(), c->ev_dst, 0));
            }
            HIPCHK(c, hipMemcpyAsync(recs_host, m->gathered, (size_t)n * sizeof(FrameRec), dest_is_device ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost, c->pay()));
            HIPCHK(c, hipStreamSynchronize(c->pay()));
        }
        if (overflow) return M17HIP_EOVERFLOW;
        return total > capacity ? M17HIP_ETRUNC : M17HIP_OK;
    }
    if (mine) {
        ncclResult_t q = R.GroupStart();
        if (q == ncclSuccess) q = R.Send(c->compact, (size_t)mine * sizeof(FrameRec), ncclUint8, root, m->comm, c->pay());
        const ncclResult_t qe = R.GroupEnd();
        if (q == ncclSuccess) q = qe;
        if (q != ncclSuccess) return comm_failed(q);
    }
    if (const int w = comm_wait(c, m)) return w;
    return overflow ? M17HIP_EOVERFLOW : M17HIP_OK;
}

int m17hip_gather_frames(m17hip_ctx* c, m17hip_comm* m, int root, m17_frame_rec* recs_host, uint64_t capacity, uint64_t* counts_host, uint64_t* total_out)
{
    return gather_frames_impl(c, m, root, recs_host, capacity, counts_host, total_out, false);
}
int m17hip_gather_frames_device(m17hip_ctx* c, m17hip_comm* m, int root, m17_frame_rec* recs_dev, uint64_t capacity, uint64_t* counts_host, uint64_t* total_out)
{
    return gather_frames_impl(c, m, root, recs_dev, capacity, counts_host, total_out, true);
}

int m17hip_comm_last_error(const m17hip_comm* m) { return m ? m->last_rccl : 0; }

int m17hip_replay_drops(m17hip_ctx* c, uint64_t* count)
{
    if (!c || !count) return M17HIP_EINVAL;
    GUARD(c);
    uint32_t w[4] = {0, 0, 0, 0};
    HIPCHK(c, hipMemcpyAsync(w, c->overflow, 16, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    *count = w[1];
    return M17HIP_OK;
}

int m17hip_tune(m17hip_ctx* c, int key, int64_t value)
{
    if (!c) return M17HIP_EINVAL;
    GUARD(c);
    if (c->front_pending) return M17HIP_ESTATE;
    switch (key) {
    // ---- deployment knobs (include/m17hip.h) ----------------------------------------------------------------------------------
    case 3:  // samples per K2+K5 segment of a run (0 = whole run)
        if (value < 0 || value > 0x7FFFFFFF) return M17HIP_EINVAL;
        c->seg_len = (uint32_t)value;
        return M17HIP_OK;
    case 6:  // BERT statistics (m17hip_bert_stats) on/off (the runs made so far keep the setting they were made with)
        c->bert = value != 0;
        return M17HIP_OK;
    case 7: {  // packet reassembly (m17hip_packets_fetch): room for `value` completed packets per run, 0 = off
        if (value < 0 || value > (1 << 24)) return M17HIP_EINVAL;
        if (int fr = flush_payload(c)) return fr;
        HIPCHK(c, hipStreamSynchronize(c->stream));
        HIPCHK(c, hipStreamSynchronize(c->pay()));
        free_dev(c->pkt_recs2[0], &c->last_hip); free_dev(c->pkt_recs2[1], &c->last_hip); c->pkt_cap = 0; c->pkt_fed = false;
        if (value == 0) return M17HIP_OK;
        if (!c->pkt_state) {
            HIPCHK(c, hipMalloc(&c->pkt_state, (size_t)c->maxC * sizeof(PacketState)));
            HIPCHK(c, hipMalloc((void**)&c->pkt_count2, 8));
        }
        HIPCHK(c, hipMalloc(&c->pkt_recs2[0], (size_t)value * sizeof(PacketRec)));
        HIPCHK(c, hipMalloc(&c->pkt_recs2[1], (size_t)value * sizeof(PacketRec)));
        c->pkt_cap = (uint32_t)value;
        hipLaunchKernelGGL(packet_reset_kernel, dim3((c->maxC + 63) / 64), dim3(64), 0, c->stream, (PacketState*)c->pkt_state, c->maxC);
        HIPCHK(c, hipMemsetAsync(c->pkt_count2, 0, 8, c->stream));
        HIPCHK(c, hipGetLastError());
        return M17HIP_OK;
    }
    case 8:  // record slots per channel and run actually used (0 = all that were allocated): exercises M17HIP_EOVERFLOW
        if (value < 0 || value > (int64_t)c->rec_cap_alloc) return M17HIP_EINVAL;
        c->rec_cap = value ? (uint32_t)value : c->rec_cap_alloc;   // (the sets keep the layout they were written with: a finished run's records stay fetchable)
        return M17HIP_OK;
    case 9: {  // diagnostic log: room for `value` diagnostic callbacks per channel and run, 0 = off (m17hip_diag_log_fetch)
        if (value < 0 || value > (1 << 20)) return M17HIP_EINVAL;
        if (int fr = flush_fold(c)) return fr;
        if (int fr = flush_payload(c)) return fr;
        HIPCHK(c, hipStreamSynchronize(c->stream));
        HIPCHK(c, hipStreamSynchronize(c->pay()));
        free_dev(c->diag_log, &c->last_hip); c->diag_cap = 0;
        if (value == 0) return M17HIP_OK;
        if (!c->diag_count) HIPCHK(c, hipMalloc((void**)&c->diag_count, (size_t)c->maxC * 4));
        HIPCHK(c, hipMemsetAsync(c->diag_count, 0, (size_t)c->maxC * 4, c->stream));   // (on the stream the kernels that count into it run on)
        HIPCHK(c, hipMalloc((void**)&c->diag_log, (size_t)c->maxC * (size_t)value * sizeof(Diag)));
        c->diag_cap = (uint32_t)value;
        return M17HIP_OK;
    }
    case 10:  // K3 form: 0 = one wave per 32 channels (throughput), 1 = four-wave pipeline (latency), -1 (default) = the pipeline for the runs m17hip_demod_front queues
        if (value < -1 || value > 1) return M17HIP_EINVAL;
        c->dcd_form = (int)value;
        return M17HIP_OK;
    case 20:  // redo policy of the limit-filter replay: 0 (default) = beside K5, state only; 1 = in front of K5, history stored
        if (value < 0 || value > 1) return M17HIP_EINVAL;
        c->redo_form = (int)value;
        return M17HIP_OK;
    case 17:  // RunningStandardDeviation (the EVM of the diagnostic callback) folded outside K5, one lane per channel (1, default), or inside K5 (0)
        if ((value != 0) != c->defer_evm) {
            if (int fr = flush_fold(c)) return fr;
            HIPCHK(c, hipStreamSynchronize(c->stream));
            c->defer_evm = value != 0;
            hipLaunchKernelGGL(ev_move_kernel, dim3((c->maxC + 63) / 64), dim3(64), 0, c->stream, c->seq_state, c->ev_state, c->maxC, c->defer_evm ? 1 : 0);
            HIPCHK(c, hipGetLastError());
            if (!c->defer_evm) { HIPCHK(c, hipStreamSynchronize(c->stream)); free_dev(c->ev_ops2[0], &c->last_hip); free_dev(c->ev_ops2[1], &c->last_hip); c->fold_ops = nullptr; }
        }
        return M17HIP_OK;
    case 18:  // (tests) floats per channel row of deferred EVM operations, 0 = what a run of max_samples can produce: a smaller value makes the overflow flag reachable
        if (value < 0 || value > (1 << 28) || (value & 3)) return M17HIP_EINVAL;
        if (int fr = flush_fold(c)) return fr;
        HIPCHK(c, hipStreamSynchronize(c->stream));
        free_dev(c->ev_ops2[0], &c->last_hip); free_dev(c->ev_ops2[1], &c->last_hip); c->fold_ops = nullptr;
        c->ev_pitch_override = (uint32_t)value;
        return M17HIP_OK;
    case 15:  // payload frames of running stream / BERT transmissions decoded after the run, one lane per frame (1, default), or in K5 (0)
        c->defer_decode = value != 0;
        if (!c->defer_decode && c->defer_hist) {   // give the stores back (a run in flight may still use them: wait for it)
            if (int fr = flush_payload(c)) return fr;
            HIPCHK(c, hipStreamSynchronize(c->stream));
            HIPCHK(c, hipStreamSynchronize(c->pay()));
            for (auto& rs_ : c->sets) free_dev(rs_.defer_llr, &c->last_hip);
            free_dev(c->defer_hist, &c->last_hip);
        }
        return M17HIP_OK;
    case 30:  // fault injection for m17hip_gather_frames (tests): 0 = none, 1 = this rank's compaction fails, 2 = the root's staging allocation fails,
              // 3 = this rank's word of exchange 2 cannot be written (its slot keeps the word of exchange 1) and the root's staging is grown,
              // 4 = this rank cannot read exchange 1, 5 = this rank cannot read exchange 2
        if (value < 0 || value > 5) return M17HIP_EINVAL;
        c->gather_fault = (int)value;
        return M17HIP_OK;
    case 31:  // bound, in milliseconds, of every wait inside m17hip_gather_frames[_device] (default 120000; 0 = wait for ever): a wait that runs out
              // gives the communicator up (ncclCommAbort) and the call returns M17HIP_ECOMM, as does every later call through that communicator
        if (value < 0 || value > 0x7FFFFFFF) return M17HIP_EINVAL;
        c->gather_timeout_ms = (uint32_t)value;
        return M17HIP_OK;
    case 16:  // the in-place producers (m17hip_upload_i16, m17hip_upload_i16_device, m17hip_synth_i16) write the STAGING slab instead
        c->stage_inputs = value != 0;
        return M17HIP_OK;
    case 26:  // gate-aware front end: 1 = K1 of segment k >= 2 skips what the carrier cannot be on for, 0 = never, -1 (default) = per run from the previous run's share of closed gates
        if (value < -1 || value > 1) return M17HIP_EINVAL;
        c->gate_aware = (int)value;
        return M17HIP_OK;
    case 33:  // ramp of segment lengths at the start of a run: value, 2 x value, 4 x value, ... samples, then seg_len (0 = none, -1 = per run)
        if (value < -1 || value > 0x7FFFFFFF) return M17HIP_EINVAL;
        c->seg_ramp = value;
        return M17HIP_OK;
    case 13:  // workgroups of K1's bounded grid (0 = default)
        if (value < 0 || value > (1 << 20)) return M17HIP_EINVAL;
        c->fir_grid = (uint32_t)value;
        return M17HIP_OK;
#ifdef M17_TOOLS
    case 27:  // configs[1]'s limit filter: 1 (default) = limit_relay_kernel, 0 = round 4's limit_pipe_kernel
        if (value < 0 || value > 1) return M17HIP_EINVAL;
        c->limit_form = (int)value;
        return M17HIP_OK;
    case 11:  // K1 form: 1 (default) = skewed pairs on a bounded grid, 0 = round 4's rolled R = 15 form with one workgroup per tile
        if (value < 0 || value > 1) return M17HIP_EINVAL;
        c->fir_form = (int)value;
        return M17HIP_OK;
    // ---- measurement / experiment knobs: only in the tools build (make -C csrc tools -> libm17hip_tools.so), used by tools/*.py ------
    case 1:  // section timers and counters of the sequential kernel (PROF instantiation; one segment per run) -> m17hip_debug_counters
        c->profile = value != 0;
        return M17HIP_OK;
    case 19:  // per-wave working time of the sequential kernel, per segment (TIMED instantiation): m17hip_debug_counters slot = segment
        c->wave_times = value != 0;
        return M17HIP_OK;
    case 4:  // samples of the first segment of a run (0 = like the others)
        if (value < 0 || value > 0x7FFFFFFF) return M17HIP_EINVAL;
        c->seg0_len = (uint32_t)value;
        return M17HIP_OK;
    case 5:  // segments the front end may run ahead of K5 (0 = unlimited)
        if (value < 0 || value > 1000) return M17HIP_EINVAL;
        c->front_ahead = (uint32_t)value;
        return M17HIP_OK;
    case 12:  // segments of K1 complete before the first K5 (0 / 1 = its own only)
        if (value < 0 || value > 1000) return M17HIP_EINVAL;
        c->front_first = (uint32_t)value;
        return M17HIP_OK;
    case 14:  // LDS bytes per workgroup of the sequential kernel (at least its need: decides how many of them share a CU), 0 = default
        if (value < 0 || value > 65536) return M17HIP_EINVAL;
        c->seq_lds_bytes = (uint32_t)value;
        return M17HIP_OK;
    case 21:  // the matched filter of a staged run waits for K5 of this segment (1-based) of the run before it (0 = starts at once)
        if (value < 0 || value > 1000) return M17HIP_EINVAL;
        c->front_k1_after = (uint32_t)value;
        return M17HIP_OK;
    case 40: case 41: case 42: case 43: {   // pipe-layout experiments (tools/pipe_layout.py): the K3 / K1 / replay / copy stream replaced by the caller's (value = hipStream_t)
        hipStream_t* slot[4] = {&c->side, &c->side2, &c->side3, &c->copy};
        const int r = key - 40;
        HIPCHK(c, hipDeviceSynchronize());
        if (*slot[r] && !c->foreign_streams[r]) hipStreamDestroy(*slot[r]);
        *slot[r] = (hipStream_t)(uintptr_t)value;
        c->foreign_streams[r] = true;
        return M17HIP_OK;
    }
    case 25:  // 1 (default) = m17hip_demod_front also queues the replay of the staged run's first segment (beside the current run's deferred decode)
        if (value != 0 && value != 1) return M17HIP_EINVAL;
        c->gate0_early = (int)value;
        return M17HIP_OK;
#endif
    default: return M17HIP_EINVAL;
    }
}

int m17hip_debug_counters(m17hip_ctx* c, uint64_t* host, uint32_t max_waves, uint32_t* waves)
{
    if (!c || !host || !waves) return M17HIP_EINVAL;
    GUARD(c);
    const uint32_t n = std::min(max_waves, c->dbg_waves);
    HIPCHK(c, hipMemcpy(host, c->dbg, (size_t)n * DBG_SLOTS * sizeof(uint64_t), hipMemcpyDeviceToHost));
    *waves = n;
    return M17HIP_OK;
}

int m17hip_timing_enable(m17hip_ctx* c, int on)
{
    if (!c) return M17HIP_EINVAL;
    GUARD(c);
    drain_timing(c);
    c->timing = on != 0;
    return M17HIP_OK;
}
int m17hip_timing_get(m17hip_ctx* c, int which, double* total_ms, uint64_t* launches)
{
    if (!c || which < 0 || which >= KT_N) return M17HIP_EINVAL;
    GUARD(c);
    drain_timing(c);
    if (total_ms) *total_ms = c->acc_ms[which];
    if (launches) *launches = c->acc_n[which];
    return M17HIP_OK;
}
int m17hip_timing_reset(m17hip_ctx* c)
{
    if (!c) return M17HIP_EINVAL;
    GUARD(c);
    drain_timing(c);
    for (int k = 0; k < KT_N; ++k) { c->acc_ms[k] = 0; c->acc_n[k] = 0; }
    return M17HIP_OK;
}

}  // extern "C"
