// K5: the per-channel sequential glue — reference a3, a5, a6, a8 (decision half), a9-a13, a19:
// Correlator::sample / SyncWord / outer_symbol_levels (Correlator.h), DataCarrierDetect::update,
// ClockRecovery + KalmanFilter, FreqDevEstimator, SymbolEvm, llr<float,4>, M17Framer and the
// M17Demodulator state machine (M17Demodulator.h:233-753) — plus K4 (frame decode, wave-cooperative).
//
// Mapping: ONE WAVE PER CHANNEL.  Everything here is a recurrence or a state machine per channel, so there is
// no data parallelism across time; what the hardware charges for is wave instructions (four waves share a SIMD's
// 1.85 ns per instruction, uniform "scalar" float work included) and divergence.  With one wave per channel 4096
// channels are 4096 waves = 4 per SIMD (the whole chip busy, latencies of one wave hidden by its three neighbours),
// the state machine is wave-uniform (every branch is taken by all lanes, untaken code is skipped), the channel's
// state (Hot, Cold) lives in LDS, and the 64 lanes cooperate where a stretch of the stream allows it:
//   * "bulk chunks" of up to 480 samples while nothing but feeding happens (initialisation, the 77 quiet samples
//     after a frame, SYNC_WAIT, the inside of a payload frame): samples come from a coalesced LDS window prefetched
//     512 samples ahead, the <= 48 payload symbols of the chunk are normalised / sliced by 64 lanes at once, the
//     running EVM is folded sequentially, the correlator ring is refilled from the chunk's tail;
//   * search chunks (UNLOCKED) and sync-window chunks (*_SYNC states): lane k evaluates the sync-word trigger of
//     sample k, ballots find the first sample that needs the single-sample path;
//   * the frame decode uses 16 lanes as the 16 trellis states (m17_decode_device.hpp, viterbi_decode_wave).
// The massively parallel work (K1 FIR), the state-machine-independent recurrence (K3 sliding DFT) and the
// correlator's limit filter (K2, speculatively: m17_gate_kernel.hpp) run as their own passes; this kernel consumes
//   ybuf[c][t]    the matched-filter output, valid wherever the last 149 FIR inputs were consecutive samples
//   dcd table     the sequential carrier-detect sums for every possible segment (see K3)
//   hbuf[c][t]    the limit-filter history after every fed sample — until this kernel forces an unlock K2 could not
//                 foresee; from there to the end of the segment it carries the filter itself
// The reference gates the FIR and the correlator with the carrier detect: their input is the
// concatenation of gated-on runs.  Runs start and end on tick boundaries and last >= 960 samples, so only the
// first 148 outputs of a run see samples of the previous run; for those the FIR is recomputed cooperatively from a
// 149-sample snapshot taken when the previous run ended and patched into ybuf in place (rare).
//
// All 64 lanes hold the same scalar state and execute the single-sample path redundantly; stores of state are
// issued by every lane with identical values (no cross-lane ordering is relied on).
#pragma once

#include "m17_common.hpp"
#include "m17_decode_device.hpp"
#include "m17_frontend_kernels.hpp"
#include "m17_state.hpp"

namespace m17 {

#ifndef M17_WAVE_MINW
#define M17_WAVE_MINW 5  // waves per SIMD the register budget is sized for: 96 VGPRs, so that four waves of this kernel (4096 channels
                         // resident at once) leave 128 registers of every SIMD to the kernels that run beside it (K2, K1)
#endif
constexpr int WV_WIN = 1024;                                            // LDS window of upcoming matched-filter samples (circular)
constexpr int WV_PF = 512;                                              // prefetch granule: 8 samples per lane in flight
constexpr int WV_YCH = 480;                                             // largest bulk chunk (<= WV_PF)
constexpr int WV_TAB_WORDS = 64;                                        // per block: llr edges (the decoder's source maps and the FIR taps stay in global memory)
constexpr int WV_WAVE_WORDS = 80 + 40 + 92 + 122 + 8 + 8 + WV_WIN + 488 + 64 + 48; // per wave: ring, sync samples, llr, hist, outb, lsf, sample window, decoder soft bits (+ EVM terms), hot state, cold state
constexpr int wave_lds_words(int waves_per_block) { return WV_TAB_WORDS + waves_per_block * WV_WAVE_WORDS; }

// M17FrameDecoder::operator() on the wave's completed frame; returns (viterbi_cost, decoder state)
__device__ __forceinline__ uint2 nf_decode_wave(const DecodeTables* tb, DecodeLds L, int wl, uint32_t sync_type, M17_LDS Cold* cd, uint32_t cost_in,
                                             FrameRec* rec_base, uint32_t rec_cap, uint32_t channel, uint64_t pos, uint32_t* overflow, uint32_t* defer)
{
    DecoderRegs D{cd->dec_state, cd->lich_segments, cd->stale401};
    RecSink S{rec_base, rec_cap, nullptr, nullptr, channel, pos, sync_type, overflow, defer};
    uint32_t n_run = cd->n_run, seq = cd->seq;
    const uint32_t cost = decode_frame<true>(tb, L, 0, sync_type, D, cost_in, S, n_run, seq, wl);
    cd->dec_state = D.state; cd->lich_segments = D.lich_segments; cd->stale401 = D.stale401;
    cd->n_run = n_run; cd->seq = seq;
    return make_uint2(cost, D.state);
}

// Correlator::sample x (to - from) for ONE channel as a tight pass (Correlator.h:43-49, IirFilter.h:26-42): the limit filter's
// history after every sample of [from, to) goes to the channel's hbuf row, exactly where K2 would have left it.  Used by a wave whose
// gate has left K2's replay (a forced dcd.unlock(), M17Demodulator.h:396-404, 470-478 ...): instead of carrying the recurrence sample
// by sample through every chunk of the state machine, the wave serves itself one stretch of certainly-fed samples at a time and
// then runs its usual hbuf-reading paths over it.  256-sample blocks staged in LDS (B, replaced in place by the history values),
// the next block's loads in flight during the recurrence.  h[3] = history after sample from - 1 on entry, after to - 1 on exit.
struct Hist3 { float h0, h1, h2; };
typedef float m17_v4f __attribute__((ext_vector_type(4)));
__device__ __forceinline__ Hist3 nf_serve_limit(const float* yr, float* hr, M17_LDS float* B, uint32_t from, uint32_t to, float h0, float h1, float h2)
{
    const uint32_t l = threadIdx.x & 63u;
    // Both rows through buffer descriptors that end at `to`: a load beyond it returns 0, a store beyond it is dropped — three loads and three
    // stores per block on EVERY path, so that the wait for a block's samples can leave the previous block's stores in flight (with the
    // bounds as branches the wait-count insertion could only wait for everything: a store round trip per block).
    const __amdgpu_buffer_rsrc_t ysrc = __builtin_amdgcn_make_buffer_rsrc((void*)yr, 0, (int)(to * 4u), 0x00020000);
    const __amdgpu_buffer_rsrc_t hdst = __builtin_amdgcn_make_buffer_rsrc((void*)hr, 0, (int)(to * 4u), 0x00020000);
    constexpr uint32_t BLK = TICK;   // one block = what iir_tick_in_place runs over (192 samples; B holds them + 16 bytes of padding + 32 readable)
    float nx[3];
    auto load = [&](uint32_t b) {
#pragma unroll
        for (int j = 0; j < 3; ++j) nx[j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(ysrc, (int)((b + l + 64u * j) * 4u), 0, 0));
    };
    load(from);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int j = 0; j < 3; ++j) __builtin_amdgcn_raw_buffer_store_b32(0, hdst, 0x7FFF0000 + 256 * j, 0, 0);   // (three dropped stores: the first block's wait sees what every later one sees)
    for (uint32_t b = from; b < to; b += BLK) {
        const uint32_t n = min(BLK, to - b);
#pragma unroll
        for (int j = 0; j < 3; ++j) B[l + 64u * j] = nx[j];
        wave_lds_sync();
        load(b + BLK);   // (beyond `to`: zeros, not used)
        if (l == 0u) {
            // The recurrence is one value per sample for the whole wave: ONE lane runs it.  A whole block as K2's tick does (m17_frontend_kernels.hpp,
            // iir_tick_in_place: one asm statement, three instructions per sample, the samples replaced by their history values in their registers, the
            // LDS reads two groups ahead, the writes left in flight) — a wave that serves itself is what its launch waits for (up to 2.2 ms of a launch
            // whose median wave needs 0.45: gpurun_out/r6/wave_times_*.txt), and the compiler's form of this loop (eight samples per pass, waits for its
            // own LDS writes) ran at 16 ns per sample alone and 33 in the crowd.
            if (n == BLK) {
                iir_tick_in_place((uint32_t)(uintptr_t)B, h0, h1, h2);
            } else {
                float m2 = IirCoef::a2 * h1;
                for (uint32_t i = 0; i < n; ++i) {
                    const float hn = iir_advance_pk(fabsf(B[i]), h0, m2);
                    h2 = h1; h1 = h0; h0 = hn;
                    B[i] = hn;
                }
            }
        }
        wave_lds_sync();
#pragma unroll
        for (int j = 0; j < 3; ++j) __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, B[l + 64u * j]), hdst, (int)((b + l + 64u * j) * 4u), 0, 0);
        wave_lds_sync();
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");   // the history values are read back by this wave (other lanes, later loads)
    // (lane 0 holds the history: the caller's scalar registers take the first lane's values)
    return Hist3{__builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, h0))),
                 __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, h1))),
                 __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, h2)))};
}

// PROF: compile the 100 MHz section timers and counters in (diagnostics, tools/seq_ablate.py); the production
// instantiation carries none of them.
// TIMED: per-wave working time per segment (both only in the tools build, -DM17_TOOLS).
// KORDER: the Kalman evaluation order as a compile-time constant (3 = the default order: its clock update is inlined and the kernel makes no
// call at all: no stack, no scratch), or -1: the order of SeqParams at run time through the out-of-line variants (m17hip_set_kalman_order).
template <int WPB, bool PROF = false, bool TIMED = false, int KORDER = -1>
__global__ __launch_bounds__(64 * WPB, M17_WAVE_MINW) void demod_wave_kernel(SeqParams P)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
    float* edges = reinterpret_cast<float*>(lds);                        // [64] llr table edges (43 used)
    for (int k = threadIdx.x; k < 43; k += 64 * WPB) edges[k] = P.llr_edges[k];
    const float* taps = P.taps;                                         // [149] RRC taps (slow-FIR patch: rare, read where they are)
    __syncthreads();  // the only block-level barrier: the waves of a block are independent from here on

    // the wave index is wave-uniform: tell the compiler, so that the channel's state, pointers and every branch of the state
    // machine live in scalar registers / scalar branches instead of 64 identical vector copies
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int wl = threadIdx.x & 63;
    // the lane id as the COLD paths see it: opaque to the optimiser, so that what they derive from it (per-lane addresses of the
    // diagnostic log, of state rows, lane predicates) is computed where it is used instead of being hoisted out of the main loop and
    // kept — spilled — for the whole kernel
    auto cold_lane = [&]() -> int { int l = wl; asm volatile("" : "+v"(l)); return l; };
    const uint32_t c = blockIdx.x * WPB + wave;
    if (c >= P.C) return;
    const uint32_t korder = KORDER >= 0 ? (uint32_t)KORDER : P.kalman_order;
    uint32_t* wb = lds + WV_TAB_WORDS + wave * WV_WAVE_WORDS;
    float* ring = reinterpret_cast<float*>(wb);              // [80]  Correlator::buffer_
    float* swsm = ring + 80;                                 // [4][10] SyncWord::samples_
    DecodeLds DL;
    DL.llr = wb + 120;                                       // [92]  M17Framer::buffer_ (368 int8)
    DL.hist = DL.llr + 92;                                   // [122] Viterbi decisions
    DL.outb = DL.hist + 122;                                 // [8]
    DL.lsf = DL.outb + 8;                                    // [8]   output_buffer.lsf
    float* ywin = reinterpret_cast<float*>(DL.lsf + 8);      // [WV_WIN] circular window: sample t lives at ywin[t & (WV_WIN-1)]
    DL.soft = reinterpret_cast<int32_t*>(ywin + WV_WIN);     // [488] depunctured soft bits of the frame being decoded
    float* e2 = reinterpret_cast<float*>(DL.soft) + 304;     // [96]  per-symbol EVM terms of a chunk / limit-history window of the single-sample
                                                             //       path: words 304..399 of the decoder array, which nothing else uses
    Hot* hot_lds = reinterpret_cast<Hot*>(DL.soft + 488);    // [64]  the channel's hot scalars (see below)
    static_assert(sizeof(Hot) <= 64 * 4, "Hot must fit its LDS slot");
    static_assert(WV_WIN == 2 * WV_PF, "a prefetch granule is half the window");
    DL.src = &P.tables->src[0][0];
    DL.lich_src = P.tables->lich_src;
    DL.stride = 1;
    DL.prof = nullptr;
    if constexpr (PROF) {
        DL.prof = P.dbg + (size_t)c * DBG_SLOTS + 9;
        if (wl < 15) P.dbg[(size_t)c * DBG_SLOTS + 9 + wl] = 0;   // slots 9..23
    }
    uint16_t* llr16 = reinterpret_cast<uint16_t*>(DL.llr);

    const bool invert = P.flags & 1u;
    SeqState* gs = P.state + c;
    // ... and so does the cold state (Kalman filters, decoder registers, diagnostics): its users are out-of-line helpers whose
    // global round trips (~1 us each, several in a row) made a single-sample step cost 6 us
    static_assert(sizeof(Cold) <= 48 * 4, "Cold must fit its LDS slot");
    M17_LDS Cold* cd = as_lds(reinterpret_cast<Cold*>(reinterpret_cast<uint32_t*>(hot_lds) + 64));
    // The hot scalars are staged through LDS (global -> LDS -> HotRegs and back): while the kernel runs they are wave-uniform
    // values in SCALAR registers (HotRegs, m17_state.hpp) — as plain per-lane variables ~45 of them spilled to scratch under
    // the 128-VGPR budget (4 waves per SIMD), and as LDS words every test of the state machine was a 64-cycle round trip.
    {
        const uint32_t* src = reinterpret_cast<const uint32_t*>(&gs->hot);
        uint32_t* dst = reinterpret_cast<uint32_t*>(hot_lds);
        for (int k = wl; k < (int)(sizeof(Hot) / 4); k += 64) dst[k] = src[k];
    }
    {
        const uint32_t* src = reinterpret_cast<const uint32_t*>(&gs->cold);
        M17_LDS uint32_t* dst = reinterpret_cast<M17_LDS uint32_t*>(cd);
        for (int k = wl; k < (int)(sizeof(Cold) / 4); k += 64) dst[k] = src[k];
    }
    HotRegs s;
    const float* hrow = P.h + (size_t)c * P.ypitch + YPRE;  // K2's filter history for this channel (hbuf row)
    for (int k = wl; k < 80; k += 64) ring[k] = gs->ring[k];
    for (int k = wl; k < 40; k += 64) swsm[k] = gs->sw_samples[k / 10][k % 10];
    for (int k = wl; k < 92; k += 64) DL.llr[k] = gs->llr[k];
    for (int k = wl; k < 8; k += 64) DL.lsf[k] = gs->lsf[k];
    wave_lds_sync();
    s.load(as_lds(hot_lds));
    if (!(P.flags & 2u)) { cd->n_run = 0; cd->n_diag_run = 0; cd->ev_cursor = 0; }  // (flag bit 1: a later segment of the same run keeps counting its records)
    // Sample window: ybuf samples [t, avail) are in LDS; the next WV_PF samples are in flight in registers (pf) so that the
    // HBM/L2 latency of this channel's row is paid ~WV_PF samples ahead of its use instead of at the head of every step.
    float pf[WV_PF / 64];
    uint32_t avail = 0;

    const int16_t* xr = P.x + (size_t)c * P.xpitch + XPRE;
    float* yr = const_cast<float*>(P.y) + (size_t)c * P.ypitch + YPRE;  // K1's output; the first 148 samples of a gated run are patched in place
    const float* tab = P.dcd_table + (size_t)c * P.ticks_cap * 12;
    // tick arithmetic in 32 bits: sample 0 of this segment lies pos0_ph samples into tick k0 (its low word: the width seg_start_tick is kept in),
    // which is row row_k0 of the table of this run
    const uint32_t pos0_ph = (uint32_t)(P.pos0 % TICK), k0 = (uint32_t)(P.pos0 / TICK), k0_mod5 = (uint32_t)((P.pos0 / TICK) % 5u);
    const uint32_t row_k0 = (uint32_t)(P.pos0 / TICK - P.tick_row0);
    FrameRec* rec_base = P.recs + (size_t)c * P.rec_cap;
    uint32_t t = 0;  // next sample (relative to this run)
    // Loads go through a buffer resource over this channel's row [0, T): the bounds check is the hardware's (a dword at or beyond
    // T reads 0.0) and the eight rows of a granule are one address computation plus instruction offsets.  `avail` is always a
    // multiple of WV_PF, so a granule never wraps inside the (2 * WV_PF)-sample window: its LDS stores are one address too.
    const __amdgpu_buffer_rsrc_t yrsrc = __builtin_amdgcn_make_buffer_rsrc((void*)yr, 0, (int)(P.T * 4u), 0x00020000);
    auto pf_issue = [&]() {   // start loading [avail, avail + WV_PF): 8 coalesced 256-byte rows
        const uint32_t voff = (avail + (uint32_t)wl) * 4u;
#pragma unroll
        for (int k = 0; k < WV_PF / 64; ++k) pf[k] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(yrsrc, (int)(voff + 256u * k), 0, 0));
    };
    auto pf_commit = [&]() {  // the loads issued a whole granule ago have landed: move them into the window
        float* dst = ywin + (avail & (WV_WIN - 1)) + wl;
#pragma unroll
        for (int k = 0; k < WV_PF / 64; ++k) dst[64 * k] = pf[k];
        avail += WV_PF;
        wave_lds_sync();
    };
    // (on demand: while a transmission runs the window is read for the ~95 samples of a sync phase per frame — the frame chunks gather their
    //  symbols from the row — so a granule loaded ahead was thrown away more often than used, and its eight registers were live across the
    //  whole main loop)
    auto ensure = [&](uint32_t need) {  // make [t, t + need) readable from the window (need <= WV_PF)
        while (avail < t + need && avail < P.T) {
            pf_issue();
            pf_commit();
        }
    };
    // after ybuf was patched / the window was used as scratch: refill from the granule that holds t0
    auto window_reset = [&](uint32_t t0) { avail = t0 & ~(uint32_t)(WV_PF - 1); };

    // ---------------- wave-uniform helpers ------------------------------------------------------------------------
    auto corr_index = [&]() -> uint32_t { return s.prev_pos % 10u; };
    auto idx0_of = [&](uint32_t ring_pos) -> uint32_t { return ring_pos % 10u; };   // correlator index of the sample that goes into slot ring_pos
    float r8[8];  // the eight ring samples one symbol apart that end at the newest sample (shared by all sync words)
#pragma unroll
    for (int i = 0; i < 8; ++i) r8[i] = 0.f;
    auto load_r8 = [&]() {
        uint32_t p = s.prev_pos + 10u;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (p >= 80u) p -= 80u;
            r8[i] = ring[p];
            p += 10u;
        }
    };
    auto correlate = [&](int w) -> float {  // Correlator.h:51-64: oldest symbol first
        return sync_correlate(w, r8);
    };
    // Correlator::limit() after the newest sample `tt` was fed.  While the run trusts K2 the history comes from hbuf through a
    // 64-sample LDS window (the e2 array, idle outside payload chunks); otherwise from the filter K5 carries itself.
    int32_t hw_base = 0x40000000;  // first hbuf index held in the window (invalid)
    uint32_t cur_tt = 0;           // index of the newest fed sample (single-sample path)
    // The filter history the sync-word window of a *_SYNC state will need (sync_count 77 .. 86 and the single-sample step that
    // follows) is known 77 samples ahead: fetched from hbuf straight into LDS (global_load_lds) when the quiet stretch before the
    // window starts, so that neither the window chunk nor the step waits for HBM.  hpf[k] = hbuf[hpf_base + k]; the buffer sits
    // in the decoder's cost-word array (idle between frames; every decode invalidates it).
    float* hpf = reinterpret_cast<float*>(DL.soft) + 240;
    int32_t hpf_base = -0x40000000;
    SReg<uint32_t> hpf_wait; hpf_wait = 0u;   // (wave-uniform flags as scalar WORDS: as bools the long-lived ones were kept as 64-bit lane masks)
    auto hpf_issue = [&](int32_t base) {
        const int64_t i = min((int64_t)base + cold_lane(), (int64_t)P.T - 1);
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(hrow + i), (__attribute__((address_space(3))) void*)hpf, 4, 0, 0);
        hpf_base = base;
        hpf_wait = 1u;
    };
    // The DCD sums of the NEXT update point (two floats of one table row) are fetched the same way right after each update:
    // the point is 960 (carrier on) or 384 samples away and the sum it will read is the one that restarted with the next tick.
    float* dpf = reinterpret_cast<float*>(hot_lds) + 62;   // two spare words of the hot slot
    uint32_t dpf_tick = 0xFFFFFFFFu;   // relative tick (from k0) whose sums are in dpf (none)
    auto hpf_ready = [&] {
        if (hpf_wait) { __builtin_amdgcn_s_waitcnt(0x0F70); asm volatile("" ::: "memory"); hpf_wait = 0u; }   // vmcnt(0)
    };
    auto cur_lim = [&]() -> float {
        const int32_t tt = (int32_t)cur_tt;
        {
            const int32_t o2 = tt - hpf_base;
            if (o2 >= 2 && o2 < 64) { hpf_ready(); return iir_output(hpf[o2], hpf[o2 - 1], hpf[o2 - 2]); }
        }
        if (tt - 2 < hw_base || tt >= hw_base + 64) {
            hw_base = tt - 2;
            const int64_t i = (int64_t)hw_base + wl;
            e2[wl] = i < (int64_t)P.T ? hrow[i] : 0.f;
            wave_lds_sync();
        }
        const int o = tt - hw_base;
        return iir_output(e2[o], e2[o - 1], e2[o - 2]);
    };
    // A forced dcd.unlock() is the one thing K2's replay of the gate could not foresee.  The gate itself does not move before the next
    // update point (the unlock clears the trigger; dcd_ falls when update_dcd sees it, M17Demodulator.h:275-286, 742-752), and K2 feeds
    // those samples too: its history stays right up to there (h_until).  Beyond it the wave is on its own for the rest of the segment
    // (`diverged`): it serves itself — nf_serve_limit over every stretch of samples that is certain to be fed (up to its next update
    // point) — and keeps reading hbuf like everybody else.  The next segment starts from a fresh replay (K2 redoes the channel from this
    // wave's state).  While diverged, s.h0..h2 = the filter's history after the last sample served / fed.
    unsigned long long n_despec = 0;
    SReg<uint32_t> diverged; diverged = 0u;
    uint32_t h_until = P.T;   // hbuf holds this channel's true history for every fed sample below this (relative) index
    auto pick_hist = [&](uint32_t tt) {   // the history after sample tt, from hbuf (tt < h_until)
        s.h0 = hrow[(int64_t)tt]; s.h1 = hrow[(int64_t)tt - 1]; s.h2 = hrow[(int64_t)tt - 2];
    };
    SReg<uint32_t> left_replay; left_replay = 0u;   // a forced unlock fell into THIS segment: the replay that is (or was) run for it ends in a state that is not this channel's
    auto despec = [&](uint32_t tt) {      // tt: the sample being processed; s.count already counts it
        left_replay = 1u;
        if (!diverged) {
            ++n_despec;
            if (cold_lane() == 0) atomicAdd(P.overflow + 1, 1u);   // (statistics: m17hip_replay_drops)
            diverged = 1u;
            h_until = min(P.T, tt + (960u - min((uint32_t)s.count, 960u)) + 1u);
        }
    };
    auto sw_triggered = [&](int w) -> float {  // Correlator.h:150-157
        const float lim = cur_lim();
        const float l1 = lim * SW_MAG1[w];
        const float l2 = lim * SW_MAG2[w];
        const float v = correlate(w);
        return (v > l1 || v < l2) ? v : 0.0f;
    };
    auto sw_step = [&](int w) -> uint32_t {  // SyncWord::operator() :179-200 (+ find_peak :161-177)
        const float v = sw_triggered(w);
        if (v != 0.f) {
            if (!s.sw_trig[w]) {
                for (int k = 0; k < 10; ++k) swsm[w * 10 + k] = 0.f;
                s.sw_trig[w] = 1;
            }
            swsm[w * 10 + (int)corr_index()] = v;
        } else if (s.sw_trig[w]) {
            s.sw_trig[w] = 0;
            s.sw_timing[w] = 0;
            float peak = v;
            for (int k = 0; k < 10; ++k) {
                const float f = swsm[w * 10 + k];
                if (fabsf(f) > fabsf(peak)) { peak = f; s.sw_timing[w] = (uint32_t)k; }
            }
            s.sw_updated[w] = peak > 0.f ? 1 : -1;
        }
        return s.sw_timing[w];
    };
    auto sw_updated = [&](int w) -> int32_t { const int32_t r = s.sw_updated[w]; s.sw_updated[w] = 0; return r; };
    auto update_values = [&](uint32_t index) {  // M17Demodulator.h:233-241
        const float2 r = nf_update_values(cd, ring, 1, 0, s.sample_index, korder, P.level_gain);
        s.idev = r.x; s.offset = r.y;
        s.sync_sample_index = index;
    };
    auto dev_reset = [&]() { cd->dev_reset = 1; };
    auto clock_flags = [&]() {  // the index-0 prologue of operator() (:695-709)
        if (s.need_clock_reset) {
            Kal2 k;
            kal_reset(k, (float)s.sync_sample_index);  // ClockRecovery::reset :33-39
            lds_put(&cd->ck, k);
            s.ck_count = 0;
            s.ck_sample_index = (int32_t)(int8_t)(float)s.sync_sample_index;
            s.ck_clock_est = 0.f;
            s.need_clock_reset = 0;
            s.sample_index = s.sync_sample_index;
        } else if (s.need_clock_update) {
            const ClockOut o = nf_clock_update_idx<KORDER>(cd, s.sync_sample_index, s.ck_count, korder);
            s.ck_sample_est = o.sample_est; s.ck_clock_est = o.clock_est; s.ck_sample_index = o.sample_index;
            s.ck_count = 0;
            s.need_clock_update = 0;
        }
    };
    auto corr_sample = [&](float v) {  // Correlator::sample :43-49
        ring[s.ring_pos] = v;
        s.prev_pos = s.ring_pos;
        if (++s.ring_pos == 80u) s.ring_pos = 0;
        if (s.run_pos < 148) s.run_pos++;
    };
    // symbol normalisation + EVM error term (do_frame :610-614, SymbolEvm.h:31-51)
    auto normalise = [&](float filtered, float& err) -> float {
        float sample = filtered - s.offset;
        sample = sample * s.idev;
        sample = sample * 1.0f;  // polarity
        err = core::evm_error(sample);
        return sample;
    };
    // DataCarrierDetect::update at the point that ends with relative sample te, then the fetch for the next point
    auto dcd_update_at = [&](uint32_t te) {
        const uint32_t kr = (pos0_ph + te + 1u) / TICK - 1u;          // the tick that ends with sample te, counted from k0
        bool have = false;
        float l1 = 0.f, l2 = 0.f;
        if (dpf_tick == kr) { hpf_ready(); l1 = dpf[0]; l2 = dpf[1]; have = true; }
        s.dcd_trig = nf_dcd_update(cd, tab, row_k0 + kr, k0 + kr, s.dcd_trig, have, l1, l2);
        const uint32_t krn = kr + (s.dcd_on ? 5u : 2u);               // 960 / 384 samples on
        const uint32_t ten = (krn + 1u) * TICK - 1u - pos0_ph;        // relative sample of that point
        dpf_tick = 0xFFFFFFFFu;
        if (ten < P.T && row_k0 + krn < P.ticks_cap) {
            const float* rown = tab + (size_t)(row_k0 + krn) * 12 + (size_t)((k0_mod5 + kr + 1u) % 5u);
            const int l = cold_lane();
            if (l < 2) __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(rown + 6 * l), (__attribute__((address_space(3))) void*)dpf, 4, 0, 0);
            dpf_tick = krn;
            hpf_wait = 1u;
        }
    };
    const float alpha = core::EVM_ALPHA;  // RunningStandardDeviation<float,184>::alpha
    // carrier-on update point: tail of operator() (:742-752); te = relative index of the sample just processed
    // The diagnostic callback (M17Demodulator.h:681-685, 746-750): its arguments become the channel's m17_diag; with the
    // diagnostic log on, every invocation is also appended to the channel's log with the sample that fired it.
    // The running EVM deferred (m17_state.hpp, evm_deferred_kernel): one operation appended to the channel's row
    auto ev_op = [&](float v) {
        const uint32_t cur = cd->ev_cursor;
        if (cold_lane() == 0 && cur < P.ev_pitch) P.ev_ops[(size_t)c * P.ev_pitch + cur] = v;
        cd->ev_cursor = cur + 1u;
    };
    // the mark of a carrier-on diagnostic callback: where evm_deferred_kernel puts the value (read BEFORE fire_diag counts the entry)
    auto ev_mark = [&]() -> float {
        const uint32_t nd = cd->n_diag_run;
        return (P.diag_log && nd < P.diag_cap) ? -(float)(nd + 2u) : EV_EMIT;
    };
    auto fire_diag = [&](uint32_t te, float evm_arg) {
        nf_fire_diag(cd, s.dcd_on, evm_arg, s.idev, s.offset, s.st != ST_UNLOCKED, s.ck_clock_est, s.sample_index,
                     s.sync_sample_index, s.ck_sample_index, s.viterbi_cost);
        if (P.diag_log) {
            const uint32_t n = cd->n_diag_run;
            if (n < P.diag_cap) {
                wave_lds_sync();
                const M17_LDS uint32_t* src = reinterpret_cast<const M17_LDS uint32_t*>(&cd->diag);
                uint32_t* dst = reinterpret_cast<uint32_t*>(P.diag_log + ((size_t)c * P.diag_cap + n));
                const uint64_t pos = P.pos0 + te;
                const int l = cold_lane();
                if (l < 16) {
                    uint32_t w = src[l];
                    if (l == 12) w = s.st;
                    if (l == 13) w = cd->seq;
                    if (l == 14) w = (uint32_t)pos;
                    if (l == 15) w = (uint32_t)(pos >> 32);
                    dst[l] = w;
                }
            }
            cd->n_diag_run = n + 1;
        }
    };
    // The Viterbi cost of the last frame is consulted only when a sync word is NOT found (below); if that frame's decoding was
    // deferred (its cost is a tag), it is decoded here after all — from the deferred-frame store, on 92 words of scratch inside the
    // decoder's array — and its record completed, so that decode_deferred_kernel skips it.
    auto resolve_cost = [&]() {
        if (!cost_is_deferred(s.viterbi_cost)) return;
        const uint32_t slot = s.viterbi_cost & ~DEFER_TAG;
        hpf_ready(); hpf_base = -0x40000000; hw_base = 0x40000000;   // the decoder takes the cost-word array, the scratch is e2's
        uint32_t* w = reinterpret_cast<uint32_t*>(rec_base + slot);
        const int kind = kind_of_frame_type(w[5] & 0xFFu);
        const int stale = (int)w[14];
        uint32_t* sc = reinterpret_cast<uint32_t*>(DL.soft) + 304;
        const uint32_t* src = P.defer + ((size_t)c * P.rec_cap + slot) * 46;
        const int l = cold_lane();
        for (int k = l; k < 92; k += 64) sc[k] = unpack_llr_nibbles(src[k >> 1], k & 1);
        wave_lds_sync();
        DecodeLds L2 = DL;
        L2.llr = sc;
        const uint32_t cost = viterbi_decode_wave_cold(L2, l, kind, stale);   // (the opaque lane id: nothing of this copy is shared with the hot ones)
        wave_lds_sync();
        complete_record(w, cost, DL.outb, 1, 0, len_of_kind(kind));
        s.viterbi_cost = cost;
    };
    auto dcd_point_on = [&](uint32_t te) {
        if (!s.dcd_trig) {  // update_dcd -> dcd_off :260-265 (dcd_ is on here)
            if (diverged) pick_hist(te);   // the history freezes here; the next gated run of this segment starts from it
            s.st = ST_UNLOCKED;
            s.dcd_on = 0;
            nf_snapshot_hist(gs->hist, xr, te, cold_lane());
        }
        s.count = 0;
        if (P.ev_ops) { ev_op(ev_mark()); fire_diag(te, __uint_as_float(EVM_PENDING)); }
        else fire_diag(te, sqrtf(s.evm_S));
        dcd_update_at(te);
    };

    unsigned long long n_bulk = 0, n_bulk_samples = 0, n_scalar = 0, n_flip = 0, n_decode = 0;
    uint32_t n_mode[8] = {0, 0, 0, 0, 0, 0, 0, 0}, n_lim_clock = 0, n_lim_count = 0, n_lim_room = 0;   // (PROF: loop iterations per chunk kind, why frame chunks ended)
    auto now = [&]() -> unsigned long long { if constexpr (PROF) return wall_clock64(); else return 0ull; };
    const unsigned long long tk0 = now();
    // wave timing (tuning knob 19; the production code in an instantiation of its own): how long THIS wave works on its segment, in 10 ns ticks, with the launch's
    // duration = the slowest wave's.  Slot = segment index (flags bits 8..12).
    // (the start time waits in the slot itself: nothing of this stays in registers across the kernel)
    if constexpr (TIMED) if (wl == 0) P.dbg[(size_t)c * DBG_SLOTS + ((P.flags >> 8) & 31u)] = wall_clock64();
    unsigned long long tk_bulk = 0, tk_scalar = 0, tk_decode = 0, tk_patch = 0, tk_ens = 0, tk_sym = 0, tk_iir = 0, tk_search = 0, tk_off = 0, tk_sel = 0, tk_tail = 0;

    // The first 148 FIR outputs of a gated run still see the tail of the previous run (Q2): recompute them from the
    // 149-sample snapshot + the run's own samples and patch ybuf in place, 64 outputs at a time, so that every later
    // read of ybuf is exact.  r0 = samples of the run already fed (> 0 when a run continues from the previous launch).
    // The first 148 FIR outputs of a gated run still see the tail of the previous run (Q2): recompute them from the
    // 149-sample snapshot + the run's own samples and patch ybuf in place, 64 outputs at a time, so that every later
    // read of ybuf is exact.  r0 = samples of the run already fed (> 0 when a run continues from the previous launch).
    auto patch_run_start = [&](uint32_t t0) {
        const unsigned long long p0 = now();
        const int r0 = s.run_pos;
        const int64_t rs = (int64_t)t0 - r0;                       // relative index of the run's first sample (>= -148)
        const int l = cold_lane();
        for (int k = l; k < 149; k += 64) ywin[k] = scale_sample((int)gs->hist[k], invert);
        for (int k = l; k < 148; k += 64)
            if (rs + k < (int64_t)P.T) ywin[149 + k] = scale_sample((int)xr[rs + k], invert);
        wave_lds_sync();
        for (int j = r0 + l; j < 148; j += 64) {
            if (rs + j >= (int64_t)P.T) break;
            float acc = 0.f;
            for (int i = 0; i < NTAPS; ++i) {                      // FirFilter.h:36-40: newest sample first
                const float p = ywin[149 + j - i] * taps[i];
                acc = acc + p;
            }
            yr[rs + j] = acc;
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");  // the patched samples are read back by other lanes of this wave
        wave_lds_sync();
        window_reset(t0);
        tk_patch += now() - p0;
    };
    // A channel that left K2's replay in the previous segment finds nothing of its own in hbuf (the replay that ran ahead started from
    // a state that is not this channel's; K2 is re-deriving the replay's state from this wave's while we run): it serves itself from
    // its first sample on.  The next segment's replay is good again.
    if (P.dropped_in && P.dropped_in[c]) {
        diverged = 1u;
        h_until = 0;
        if ((s.initializing || s.dcd_on) && wl == 0) { float* hw = const_cast<float*>(hrow); hw[-1] = s.h0; hw[-2] = s.h1; hw[-3] = s.h2; }
    }
    if (s.run_pos < 148 && (s.initializing || s.dcd_on)) patch_run_start(0);

    // ---------------- main loop (wave-uniform control flow) ------------------------------------------------------------
    uint32_t flags_t = 0xFFFFFFFFu;   // sample whose index-0 prologue (:695-709) has already run during chunk selection
    while (t < P.T) {
        const unsigned long long l0 = now();
        bool decode_due = false, tail_dcd = false;
        uint32_t te = 0;
        if (diverged && t >= h_until && (s.initializing || s.dcd_on)) {
            // every sample up to the next update point (the end of the initialisation run) will be fed whatever happens: serve them
            const unsigned long long q0 = now();
            const uint32_t fed_end = min(P.T, t + (s.initializing ? (uint32_t)s.initializing : 960u - min((uint32_t)s.count, 959u)));
            hpf_ready(); hpf_base = -0x40000000; hw_base = 0x40000000;   // the staging block is the decoder's array; windows of hbuf are stale now
            const Hist3 r = nf_serve_limit(yr, const_cast<float*>(hrow), as_lds(reinterpret_cast<float*>(DL.soft)), t, fed_end, s.h0, s.h1, s.h2);
            s.h0 = r.h0; s.h1 = r.h1; s.h2 = r.h2;
            h_until = fed_end;
            tk_iir += now() - q0;
        }
        // ---- carrier off: nothing happens until the next DCD update point (:675-689) -> jump there ----------------------
        if (!s.initializing && !s.dcd_on) {
            const unsigned long long f0 = now();
            const uint32_t n = min(384u - s.count, P.T - t);
            s.count += n;
            t += n;
            if (avail < t) window_reset(t);  // skipped samples are never read
            if (s.count == 384u) {
                const uint32_t te = t - 1;
                if (s.dcd_trig) {   // update_dcd :275-286 -> dcd_on :244-257
                    s.dcd_on = 1;
                    if (s.st == ST_UNLOCKED) {
                        s.sync_count = 0; s.missing_sync_count = 0;
                        for (int k = wl; k < 92; k += 64) DL.llr[k] = 0;  // framer.reset()
                        s.framer_idx = 0;
                        cd->dec_state = 0;                                 // decoder.reset()
                        if (P.ev_ops) ev_op(EV_RESET); else s.evm_S = 0.f;  // evm.reset()
                        wave_lds_sync();
                    }
                    s.need_clock_reset = 1;
                    s.run_pos = 0;  // a new gated run starts with the next sample
                    if (t < P.T) patch_run_start(t);
                    if (diverged) {   // the history the run inherits, where its first samples will look for it (K2's convention)
                        if (wl == 0) { float* hw = const_cast<float*>(hrow); hw[(int64_t)te] = s.h0; hw[(int64_t)te - 1] = s.h1; hw[(int64_t)te - 2] = s.h2; }
                        h_until = t;
                    }
                }
                dcd_update_at(te);
                fire_diag(te, 0.f);
                s.count = 0;
            }
            tk_off += now() - f0;
            continue;
        }

        // ---- bulk chunk: n samples during which the state machine only feeds the correlator (and, inside a frame, slices
        //      payload symbols at a fixed sample_index) ----------------------------------------------------------------------
        enum { BULK_NONE, BULK_INIT, BULK_QUIET, BULK_FEED, BULK_FRAME, BULK_SEARCH, BULK_SYNCWIN, BULK_LSF };
        int mode = BULK_NONE;
        uint32_t n = 0, o1 = 0;
        uint32_t upd_off = 0xFFFFFFFFu;   // BULK_FRAME: offset of the index-0 sample inside the chunk on which a pending clock update is due
        bool completes = false;   // the chunk ends on the sample that completes the frame (BULK_FRAME) / leaves SYNC_WAIT (BULK_QUIET)
        {
            const uint32_t room = min(P.T - t, (uint32_t)WV_YCH);
            if (s.initializing) {
                n = min((uint32_t)s.initializing, room);
                mode = BULK_INIT;
            } else {
                uint32_t lim = min(room, 960u - s.count);
                const uint32_t idx0 = s.ring_pos % 10u;  // correlator index of sample t
                // the index-0 prologue (:695-709) does not depend on the sample values: run it now if sample t is an index-0 sample
                if (idx0 == 0u && (s.need_clock_reset | s.need_clock_update)) { clock_flags(); flags_t = t; }
                if (s.need_clock_reset | s.need_clock_update) lim = min(lim, 10u - idx0);  // stop before the next index-0 sample
                const bool is_sync = s.st == ST_STREAM_SYNC || s.st == ST_PACKET_SYNC || s.st == ST_BERT_SYNC;
                if (is_sync) {
                    if (s.sync_count < 86) {   // the samples that only count (sync_count + 1 < MIN_SYNC_COUNT = 78) and the window where the next sync
                        n = min((uint32_t)(86 - s.sync_count), lim);   // word is looked for (:420-574), up to the sample before its trigger falls / EOT / the count runs out
                        mode = BULK_SYNCWIN;
                        if (s.sync_count < 77) {   // the limit history the window will want: into LDS ahead of its use (issued behind the frame decode as a rule)
                            const int32_t target = (int32_t)t + (77 - (int32_t)s.sync_count) - 3;
                            if (hpf_base != target && (!diverged || target + 64 <= (int32_t)h_until)) hpf_issue(target);
                        }
                    }
                } else if (s.st == ST_SYNC_WAIT) {  // do_sync_wait :583-593: count up to MAX_SYNC_COUNT, then one transition sample
                    const uint32_t q = s.sync_count < 86 ? (uint32_t)(86 - s.sync_count) : 0u;
                    n = min(q + 1u, lim);
                    completes = n == q + 1u;
                    mode = BULK_QUIET;
                } else if (s.st == ST_LSF_SYNC) {   // do_lsf_sync :350-411 acts only where index() == sample_index
                    if (!s.need_clock_reset) {      // up to 48 symbols at once (below); a pending clock UPDATE is served on the way
                        o1 = (s.sample_index + 10u - idx0) % 10u;
                        n = min(room, 960u - s.count);
                        mode = BULK_LSF;
                    } else {
                        n = min((s.sample_index + 10u - idx0) % 10u, lim);
                        mode = BULK_FEED;
                    }
                } else if (s.st == ST_UNLOCKED) {   // do_unlocked :289-342 while no sync word is (or becomes) triggered
                    const bool phase_a = s.missing_sync_count < 1920;
                    const bool armed = phase_a ? !s.sw_trig[0] : !(s.sw_trig[1] | s.sw_trig[2]);
                    if (armed) {
                        n = lim;   // (up to a whole window chunk: 64 samples per pass below)
                        if (phase_a) n = min(n, (uint32_t)(1920 - s.missing_sync_count));
                        mode = BULK_SEARCH;
                    }
                } else if (s.st == ST_FRAME) {
                    o1 = (s.sample_index + 10u - idx0) % 10u;                       // offset of the first payload symbol
                    const uint32_t remaining = (368u - s.framer_idx) >> 1;          // symbols until the frame is complete
                    const uint32_t last = o1 + 10u * (remaining - 1u);              // offset of the completing symbol
                    // (a frame chunk is bounded by neither the sample window nor the carrier-detect update points: it takes its symbol
                    //  samples from the channel's row and serves the update points that fall inside it on its way)
                    uint32_t lim_f = P.T - t;
                    // (a pending clock UPDATE — every frame of a locked stream starts with one, set on leaving SYNC_WAIT — is served inside
                    //  the chunk, on the index-0 sample it is due on; a pending reset ends the chunk in front of that sample)
                    if (s.need_clock_reset) lim_f = min(lim_f, 10u - idx0);
                    else if (s.need_clock_update) upd_off = 10u - idx0;
                    if (diverged) lim_f = min(lim_f, 960u - s.count);   // (a wave that serves itself the limit filter does so from update point to update point: no sample may be passed over)
                    n = min(last + 1u, lim_f);
                    completes = n == last + 1u;
                    mode = BULK_FRAME;
                }
            }
            if (n < 1u) mode = BULK_NONE;
        }
        tk_sel += now() - l0;
        if constexpr (PROF) {
            n_mode[mode & 7]++;
            if (mode == BULK_FRAME && !completes) {
                if ((s.need_clock_reset | s.need_clock_update) && n == 10u - s.ring_pos % 10u) n_lim_clock++;
                else if (n == P.T - t) n_lim_room++;
            }
        }
        bool frame_done = false;
        if (mode == BULK_FRAME) {
            // ---- FRAME CHUNK: up to a whole frame (184 symbols, 1840 samples) at once.  do_frame (:596-654) touches one sample in ten
            // (the symbol at index() == sample_index) plus the anti-phase clock prediction (:601-606), so the chunk does not go through
            // the sample window: lane l takes symbols l, l + 64, l + 128 straight from the channel's matched-filter row (three gathers,
            // in flight during the clock checks), slices them, and the running EVM is folded in order.  Carrier-detect update points
            // inside the chunk (:742-752: every 960 samples, at most two per frame) are served where they fall — diagnostic callback with
            // the EVM as of that sample, then dcd.update() — as long as the carrier stays on; the point that would turn it off ends the
            // chunk (the common tail does the rest).  The chunk also ends before an anti-phase sample that moves sample_index.
            const unsigned long long b0 = now();
            const uint32_t S = s.sample_index, idx0 = s.ring_pos % 10u;
            const uint32_t n_asked = n;
            float ysym[3], yring[2];
            {
                const uint32_t voff = (t + o1 + 10u * (uint32_t)wl) * 4u;
#pragma unroll
                for (int j = 0; j < 3; ++j) ysym[j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(yrsrc, (int)(voff + 2560u * j), 0, 0));
                // the correlator ring after the chunk = its last 80 samples (Correlator::sample :43-49), asked for now as well
                const uint32_t r0 = (t + (n > 80u ? n - 80u : 0u) + (uint32_t)wl) * 4u;
#pragma unroll
                for (int j = 0; j < 2; ++j) yring[j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(yrsrc, (int)(r0 + 256u * j), 0, 0));
            }
            // every anti-phase clock_recovery.update() of the chunk (:601-606) must leave sample_index where it is
            const uint32_t a1 = ((S + 5u) % 10u + 10u - idx0) % 10u;  // offset of the first anti-phase sample
            // (lane l checks anti-phase samples l, l + 64, ...; the first one that moves it is the chunk's last sample: do_frame does nothing
            //  else on that sample — it is no symbol sample — so the move itself is committed with the chunk, and the next chunk starts
            //  from the new sample_index)
            // The clock update that is due on the index-0 sample at offset u0 (:695-709; ClockRecovery::update(uint8_t) :54-67 on the count
            // up to that sample): the anti-phase samples from u0 on predict from ITS estimate, with the count restarted there; the one that
            // may lie in front of it predicts from the old one.  Served here if the chunk reaches u0 the way the sample-by-sample form would:
            // no anti-phase move in front of it, no carrier-detect update point in front of it (its diagnostic record holds the clock
            // state); otherwise the chunk ends in front of u0 as it used to and the update is served during the next chunk selection.
            const float est_old = s.ck_sample_est, clk_old = s.ck_clock_est;
            const uint32_t cnt_old = s.ck_count;
            uint32_t u0 = upd_off;
            if (u0 != 0xFFFFFFFFu) {
                bool ok = u0 < n && 959u - s.count >= u0;
                if (ok && a1 < u0) ok = (uint32_t)(uint8_t)clock_predict(est_old, clk_old, cnt_old + a1 + 1u) == S;
                if (ok) {
                    const ClockOut o = nf_clock_update_idx<KORDER>(cd, s.sync_sample_index, cnt_old + u0, korder);
                    s.ck_sample_est = o.sample_est; s.ck_clock_est = o.clock_est; s.ck_sample_index = o.sample_index;
                    s.need_clock_update = 0;
                } else {
                    if (u0 < n) { n = u0; completes = false; }
                    u0 = 0xFFFFFFFFu;
                }
            }
            const uint32_t a_eff = (u0 != 0xFFFFFFFFu && a1 < u0) ? a1 + 10u : a1;   // first anti-phase sample whose result outlives the update
            auto predict_at = [&](uint32_t a) {   // ClockRecovery::update() :76-88 on the anti-phase sample at offset a
                return (u0 != 0xFFFFFFFFu && a >= u0) ? clock_predict(s.ck_sample_est, s.ck_clock_est, a - u0 + 1u) : clock_predict(est_old, clk_old, cnt_old + a + 1u);
            };
            int32_t S_moved = -1;
            for (uint32_t base = a1; base < n; base += 640u) {
                const uint32_t a = base + 10u * wl;
                const bool post = u0 != 0xFFFFFFFFu && a >= u0;
                const float v = core::clock_predict_arg(post ? (float)s.ck_sample_est : est_old, post ? (float)s.ck_clock_est : clk_old, post ? a - u0 + 1u : cnt_old + a + 1u);
                bool bad = a < n && !core::clock_predict_equals(v, (int32_t)S);
                if (__ballot(a < n && !core::clock_predict_near(v)))   // (an estimate far outside 0..10: the general form)
                    bad = a < n && (uint32_t)(uint8_t)predict_at(a) != S;
                const unsigned long long mask = __ballot(bad);
                if (mask != 0ull) {
                    const uint32_t af = base + 10u * (uint32_t)(__ffsll((long long)mask) - 1);
                    n = af + 1u;
                    completes = false;
                    S_moved = predict_at(af);
                    ++n_flip;
                    break;
                }
            }
            // the first update point inside the chunk: the sample that makes count_ 960.  If the trigger is already gone the carrier falls
            // there (update_dcd -> dcd_off :260-265): the chunk ends on that sample
            uint32_t d = 959u - s.count;
            if (d + 1u < n && !s.dcd_trig) { n = d + 1u; completes = false; S_moved = -1; }
            if (n >= 1u) {
                frame_done = true;
                const unsigned long long b1 = now();
                tk_ens += b1 - b0;
                uint32_t m = (n > o1) ? (n - o1 + 9u) / 10u : 0u;  // payload symbols inside the chunk (<= 184)
                float* ev = reinterpret_cast<float*>(DL.soft);      // [192] EVM terms of the chunk's symbols (the decoder's array: idle inside a frame; hpf sits above)
                {
                    const uint32_t wls = (uint32_t)cold_lane();   // (opaque: the per-lane bases below are recomputed here, not hoisted out of the main loop and spilled)
#pragma unroll
                    for (int j = 0; j < 3; ++j) {
                        const uint32_t k = wls + 64u * j;
                        if (k < m) {
                            float err;
                            const float sample = normalise(ysym[j], err);
                            ev[k] = (err * err) * alpha;
                            llr16[(s.framer_idx >> 1) + k] = (uint16_t)slice_llr(sample, edges);
                        }
                    }
                }
                wave_lds_sync();
                // RunningStandardDeviation::capture (StandardDeviation.h:60-72), sequential, symbol by symbol; stops where an update point wants the value
                float Sv = s.evm_S;
                uint32_t kf = 0;
                // (wave-uniform arithmetic: sixteen lanes enabled — the issue time of a VALU instruction is the same from 16 lanes up, the
                //  energy is not, and the matched filter this kernel shares the chip with is power-limited)
                auto fold_to = [&](uint32_t kend) {
                    const uint32_t k0f = kf;
                    if (wl < 16) {
                        uint32_t k = k0f;
                        for (; k < kend && (k & 3u); ++k) { Sv = Sv - Sv * alpha; Sv = Sv + ev[k]; }
                        for (; k + 4 <= kend; k += 4) {
                            const float4 g = *reinterpret_cast<const float4*>(ev + k);
                            Sv = Sv - Sv * alpha; Sv = Sv + g.x;
                            Sv = Sv - Sv * alpha; Sv = Sv + g.y;
                            Sv = Sv - Sv * alpha; Sv = Sv + g.z;
                            Sv = Sv - Sv * alpha; Sv = Sv + g.w;
                        }
                        for (; k < kend; ++k) { Sv = Sv - Sv * alpha; Sv = Sv + ev[k]; }
                    }
                    Sv = SReg<float>::uni(Sv);
                    kf = max(k0f, kend);
                };
                bool served = false;
                uint32_t d_last = 0;
                const bool evd = P.ev_ops != nullptr;   // the fold is evm_deferred_kernel's: the terms go to the channel's row, the callbacks leave marks
                uint32_t nmk = 0, mk_k0 = 0, mk_k1 = 0;
                float mk_v0 = 0.f, mk_v1 = 0.f;
                while (d + 1u < n) {   // an update point INSIDE the chunk: carrier on, trigger set (tail of operator() :742-752)
                    const uint32_t kend = min(m, d >= o1 ? (d - o1) / 10u + 1u : 0u);   // the symbols up to and including sample d
                    if (evd) {
                        if (nmk == 0) { mk_k0 = kend; mk_v0 = ev_mark(); } else { mk_k1 = kend; mk_v1 = ev_mark(); }
                        ++nmk;
                    } else {
                        fold_to(kend);
                        s.evm_S = Sv;
                    }
                    if (a_eff <= d) s.ck_sample_index = (int32_t)S;        // (the anti-phase updates up to here returned sample_index)
                    s.count = 0;
                    fire_diag(t + d, evd ? __uint_as_float(EVM_PENDING) : sqrtf(Sv));
                    dcd_update_at(t + d);
                    served = true; d_last = d;
                    d += 960u;
                    if (d + 1u < n && !s.dcd_trig) {   // the NEXT point turns the carrier off: the chunk ends on it
                        n = d + 1u; completes = false; S_moved = -1;
                        m = (n > o1) ? (n - o1 + 9u) / 10u : 0u;   // (symbols sliced beyond it are sliced again when their turn comes)
                    }
                }
                if (evd) {
                    const uint32_t cur = cd->ev_cursor, wls = (uint32_t)cold_lane();
                    float* row = P.ev_ops + (size_t)c * P.ev_pitch;
#pragma unroll
                    for (int j = 0; j < 3; ++j) {
                        const uint32_t k = wls + 64u * j;
                        const uint32_t pos = cur + k + ((nmk > 0u && k >= mk_k0) ? 1u : 0u) + ((nmk > 1u && k >= mk_k1) ? 1u : 0u);
                        if (k < m && pos < P.ev_pitch) row[pos] = ev[k];
                    }
                    if (wls == 0u) {
                        if (nmk > 0u && cur + mk_k0 < P.ev_pitch) row[cur + mk_k0] = mk_v0;
                        if (nmk > 1u && cur + mk_k1 + 1u < P.ev_pitch) row[cur + mk_k1 + 1u] = mk_v1;
                    }
                    cd->ev_cursor = cur + m + nmk;
                } else {
                    fold_to(m);
                    s.evm_S = Sv;
                }
                s.framer_idx += 2u * m;
                if (a_eff < n) s.ck_sample_index = (int32_t)S;   // the anti-phase updates of the chunk (if any) returned sample_index ...
                if (S_moved >= 0) { s.ck_sample_index = S_moved; s.sample_index = (uint32_t)(uint8_t)S_moved; }   // ... but for the last one, which moved it (:601-606)
                const unsigned long long b2 = now();
                tk_sym += b2 - b1;
                {   // Correlator::sample x n: the ring keeps the last 80 samples
                    const uint32_t first = n > 80u ? n - 80u : 0u;
                    if (n != n_asked) {   // (the chunk was cut: its tail lies elsewhere)
                        const uint32_t r0 = (t + first + (uint32_t)wl) * 4u;
#pragma unroll
                        for (int j = 0; j < 2; ++j) yring[j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(yrsrc, (int)(r0 + 256u * j), 0, 0));
                    }
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        const uint32_t o = first + (uint32_t)wl + 64u * j;
                        if (o < n) ring[(s.ring_pos + o) % 80u] = yring[j];
                    }
                    s.prev_pos = (s.ring_pos + n - 1u) % 80u;
                    s.ring_pos = (s.ring_pos + n) % 80u;
                    s.run_pos = min(148, s.run_pos + (int32_t)n);
                }
                s.count = served ? n - 1u - d_last : s.count + n;
                s.ck_count = u0 != 0xFFFFFFFFu ? n - u0 : cnt_old + n;
                wave_lds_sync();
                t += n;
                te = t - 1u;
                tail_dcd = true;
                if (avail < t) window_reset(t);   // the window was not used: the samples passed over are never read from it
                if (completes) {  // the last sample of the chunk completed the frame
                    s.framer_idx = 0;
                    s.sync_count = 0;
                    decode_due = true;
                }
                ++n_bulk; n_bulk_samples += n;
                tk_bulk += now() - b0;
            } else {
                mode = BULK_NONE;
            }
        }
        if (mode == BULK_LSF) {
            // ---- LSF_SYNC IN BULK (do_lsf_sync :350-411): the state acts on one sample in ten (index() == sample_index); lane j evaluates
            // symbol j of the chunk — the three SyncWord::triggered() tests (Correlator.h:150-157) on the correlator's contents as of that
            // sample, and the outer symbol levels update_values() would take (Correlator.h:81-114) — and the wave then walks the symbols in
            // order: a preamble hit (:357-362) counts and asks for a clock update, served at the next index-0 sample (:695-709); a quiet
            // symbol (:403-406) updates the two level filters (state arithmetic only: the gain schedule).  The first symbol that does
            // anything else — LSF / stream / BERT sync word, the 193rd quiet symbol — ends the chunk: the single-sample path takes it.
            const unsigned long long b0 = now();
            ensure(n);
            const uint32_t S = s.sample_index, rp0 = s.ring_pos;
            uint32_t m = (n > o1) ? (n - o1 + 9u) / 10u : 0u;   // symbol samples in the chunk (<= 48)
            const uint32_t q = o1 + 10u * (uint32_t)wl;         // this lane's symbol sample (offset in the chunk)
            // a sample of the correlator's 80-sample history as of offset q: from the chunk (window) or from before it (ring)
            auto at_time = [&](int32_t o) -> float {             // the sample fed at offset o (>= -80)
                return o >= 0 ? ywin[(t + (uint32_t)o) & (WV_WIN - 1)] : ring[(rp0 + 80u + (uint32_t)(o + 80)) % 80u];
            };
            bool isA = false, isE = false, isQ = false;
            float mn = 0.f, mx = 0.f;
            if ((uint32_t)wl < m) {
                float r[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) r[i] = at_time((int32_t)q - 70 + 10 * i);   // oldest symbol first (Correlator.h:51-64)
                const int64_t hq = (int64_t)t + q;
                const float lim_k = iir_output(hrow[hq], hrow[hq - 1], hrow[hq - 2]);
                auto trig = [&](int w_) -> float {                // SyncWord::triggered
                    const float v = sync_correlate(w_, r);
                    return (v > lim_k * SW_MAG1[w_] || v < lim_k * SW_MAG2[w_]) ? v : 0.0f;
                };
                const float t0 = trig(0);
                if ((double)t0 > 0.1) isA = true;
                else {
                    const float t1 = trig(1), t2 = trig(2);
                    if (t2 < 0.f || (double)fabsf(t1) > 0.1) isE = true; else isQ = true;
                }
                // Correlator::outer_symbol_levels(sample_index): buffer_[i] in SLOT order; slot i holds the newest sample fed into it
                core::outer_symbol_levels([&](uint32_t slot) -> float {
                    const uint32_t back = (rp0 + q + 800u - slot) % 80u;   // how many samples ago slot `slot` was written, as of offset q
                    return back <= q ? ywin[(t + q - back) & (WV_WIN - 1)] : ring[slot];
                }, S, mn, mx);
            }
            const unsigned long long mA = __ballot(isA), mE = __ballot(isE), mQ = __ballot(isQ);
            // the quiet symbol that takes missing_sync_count beyond 192 is a transition too (:392-402)
            const uint32_t nq_here = (uint32_t)__popcll(mQ & ((2ull << wl) - 1ull));
            const unsigned long long mO = __ballot(isQ && (uint32_t)s.missing_sync_count + nq_here > 192u);
            const unsigned long long stop = mE | mO;
            const uint32_t fs = stop ? (uint32_t)(__ffsll((long long)stop) - 1) : m;   // symbols served here
            const uint32_t nc = fs < m ? o1 + 10u * fs : n;                            // samples committed here
            if (nc >= 1u) {
                // walk the symbols: clock updates fall on index-0 samples (idx0 + offset = 0 mod 10), the one at or before symbol j first
                const uint32_t ck_entry = s.ck_count;
                int32_t ckz = -1;                  // offset of the last clock update inside the chunk
                const int32_t z_done = flags_t == t ? 0 : -1;   // the prologue of offset 0 has run already (chunk selection: a clock RESET may have left an update pending)
                bool pend = s.need_clock_update != 0;
                auto clock_at = [&](uint32_t z) {  // ClockRecovery::update(sync_sample_index) in the prologue of offset z
                    const ClockOut o = nf_clock_update_idx<KORDER>(cd, s.sync_sample_index, ckz < 0 ? ck_entry + z : z - (uint32_t)ckz, korder);
                    s.ck_sample_est = o.sample_est; s.ck_clock_est = o.clock_est; s.ck_sample_index = o.sample_index;
                    ckz = (int32_t)z; pend = false;
                };
                float a0 = cd->min_x0, a1 = cd->min_x1, b0v = cd->max_x0, b1v = cd->max_x1;
                uint32_t ln = cd->lvl_n, nupd = 0;
                float lmn = 0.f, lmx = 0.f;
                bool rst_seen = false;
                uint32_t served = 0;
                for (uint32_t j = 0; j < fs; ++j) {
                    const uint32_t qj = o1 + 10u * j;
                    if (pend && qj >= S && (int32_t)(qj - S) > z_done) clock_at(qj - S);   // the index-0 sample at or before symbol j (index(qj) = S)
                    if ((mA >> j) & 1ull) { pend = true; s.sync_count += 1; }
                    else {   // quiet: ++missing_sync_count, update_values(sample_index) (:233-241)
                        s.missing_sync_count += 1;
                        lmn = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, mn), (int)j));
                        lmx = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, mx), (int)j));
                        uint32_t rst = 0;
                        if (uniform16()) {
                            const core::Kalman2Gain g = P.level_gain[ln];
                            core::level_update(a0, a1, lmn, g, korder);
                            core::level_update(b0v, b1v, lmx, g, korder);
                            rst = (cd->dev_reset || isnan(a0) || isnan(a1) || isnan(b0v) || isnan(b1v)) ? 1u : 0u;   // FreqDevEstimator::update :40-48
                        }
                        rst = SReg<uint32_t>::uni(rst);
                        ln = min(ln + 1u, (uint32_t)core::LEVEL_SCHED_LAST);
                        ++nupd;
                        if (rst) { a0 = lmn; a1 = 0.f; b0v = lmx; b1v = 0.f; ln = 0; cd->dev_reset = 0; rst_seen = true; }
                        else rst_seen = false;
                    }
                    served = j + 1u;
                }
                if (nupd) {
                    if (uniform16()) { cd->min_x0 = a0; cd->min_x1 = a1; cd->max_x0 = b0v; cd->max_x1 = b1v; }   // (the filters' state lives in those lanes)
                    cd->lvl_n = ln;
                    if (rst_seen) { s.offset = (lmn + lmx) / 2.f; s.idev = core::freqdev_idev(lmx, lmn); }
                    else { s.offset = core::freqdev_offset(b0v, a0); s.idev = core::freqdev_idev(b0v, a0); }
                    s.sync_sample_index = S;
                }
                // a clock update still pending falls on the first index-0 sample behind the last symbol served, if the chunk reaches it
                if (pend) {
                    const uint32_t from = served ? o1 + 10u * (served - 1u) + 1u : 0u;
                    uint32_t z = from + (10u - (idx0_of(rp0) + from) % 10u) % 10u;
                    if ((int32_t)z <= z_done) z += 10u;
                    if (z < nc) clock_at(z);
                }
                s.need_clock_update = pend ? 1u : 0u;
                s.ck_count = ckz < 0 ? ck_entry + nc : nc - (uint32_t)ckz;
                {   // Correlator::sample x nc
                    const uint32_t first = nc > 80u ? nc - 80u : 0u;
#pragma clang loop vectorize(disable) interleave(disable) unroll(disable)
                    for (uint32_t o = first + wl; o < nc; o += 64) ring[(rp0 + o) % 80u] = ywin[(t + o) & (WV_WIN - 1)];
                    s.prev_pos = (rp0 + nc - 1u) % 80u;
                    s.ring_pos = (rp0 + nc) % 80u;
                    s.run_pos = min(148, s.run_pos + (int32_t)nc);
                }
                s.count += nc;
                wave_lds_sync();
                t += nc;
                if (s.count == 960u) dcd_point_on(t - 1u);
                ++n_bulk; n_bulk_samples += nc;
                tk_search += now() - b0;
                continue;
            }
            mode = BULK_NONE;   // the very next sample is a transition: the single-sample path
            tk_search += now() - b0;
        }
        if (mode == BULK_SYNCWIN) {
            // ---- *_SYNC STATES (do_stream_sync :420-482, do_packet_sync :489-530, do_bert_sync :536-574) up to the sample on which something
            // happens.  The first samples only count (sync_count + 1 < 78); from then on SyncWord::operator() (Correlator.h:179-200) runs on
            // every sample: lane j evaluates sample kw + j of the chunk (at most nine of them), the quiet samples AND the samples of the
            // trigger run (:179-186 just stores them) are committed; the sample on which the trigger falls (peak search, state change),
            // an EOT hit and the sample that exhausts the count go through the single-sample path.
            const unsigned long long b0 = now();
            ensure(n);
            const uint32_t rp0 = s.ring_pos;
            const uint32_t kw = s.sync_count < 77 ? (uint32_t)(77 - s.sync_count) : 0u;   // samples in front of the window
            const uint32_t nw = n > kw ? n - kw : 0u;                                     // window samples in the chunk (<= 9)
            // (sixteen lanes at least: a VALU instruction with fewer enabled lanes issues 2.7 x slower; the spare lanes redo the last sample)
            const uint32_t k = kw + min((uint32_t)wl, nw ? nw - 1u : 0u);
            const int wd = (s.st == ST_STREAM_SYNC) ? 1 : 2;   // the word a *_SYNC state looks for
            bool hit = false, trg = false;
            float vk = 0.f;
            if (nw && wl < 16) {
                float h0k, h1k, h2k;   // the limit filter's history after sample k
                const int32_t off = (int32_t)(t + k) - hpf_base;
                if (off >= 2 && off < 64) { hpf_ready(); h0k = hpf[off]; h1k = hpf[off - 1]; h2k = hpf[off - 2]; }
                else { const int64_t hq = (int64_t)t + k; h0k = hrow[hq]; h1k = hrow[hq - 1]; h2k = hrow[hq - 2]; }
                const float lim_k = iir_output(h0k, h1k, h2k);
                float r[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) {   // samples k - 70, k - 60, ..., k: from the chunk (window) or from before it (ring)
                    const int32_t o = (int32_t)k - 70 + 10 * i;
                    r[i] = o >= 0 ? ywin[(t + (uint32_t)o) & (WV_WIN - 1)] : ring[(rp0 + 80u + (uint32_t)(o + 80)) % 80u];
                }
                auto corr = [&](int w_) { return sync_correlate(w_, r); };
                auto beyond = [&](int w_, float v) { return v > lim_k * SW_MAG1[w_] || v < lim_k * SW_MAG2[w_]; };
                vk = corr(wd);
                trg = beyond(wd, vk) && vk != 0.f;   // SyncWord::operator() tests the RETURNED value: an exact 0 beyond a negative limit is no trigger
                if (s.st == ST_STREAM_SYNC) { const float v3 = corr(3); hit = beyond(3, v3) && v3 > 0.1f; }   // EOT :424
                if ((uint32_t)wl >= nw) { trg = false; hit = false; }
            }
            const unsigned long long hmask = __ballot(hit);
            unsigned long long mask = hmask;
            const unsigned long long tmask = __ballot(trg);
            const uint32_t was_trig = s.sw_trig[wd];
            {   // + the first sample on which the trigger falls
                unsigned long long fall = ~tmask;
                if (!was_trig) fall = tmask ? (fall & ~((2ull << (__ffsll((long long)tmask) - 1)) - 1ull)) : 0ull;
                mask |= fall & ((1ull << nw) - 1ull);
            }
            const uint32_t fw = mask ? (uint32_t)(__ffsll((long long)mask) - 1) : nw;   // window samples committed here
            const uint32_t f = min(kw, n) + fw;                                          // samples committed here
            if (f > 0u) {
                const uint32_t first = f > 80u ? f - 80u : 0u;
#pragma clang loop vectorize(disable) interleave(disable) unroll(disable)
                for (uint32_t o = first + wl; o < f; o += 64) ring[(rp0 + o) % 80u] = ywin[(t + o) & (WV_WIN - 1)];
                s.prev_pos = (rp0 + f - 1u) % 80u;
                s.ring_pos = (rp0 + f) % 80u;
                s.run_pos = min(148, s.run_pos + (int32_t)f);
                s.count += f;
                s.ck_count += f;
                s.sync_count += (int32_t)f;
                if (tmask & ((1ull << fw) - 1ull)) {   // SyncWord::operator() on the triggered samples: (clear,) store at index()
                    if (!was_trig) {
                        if (wl < 10) swsm[wd * 10 + wl] = 0.f;
                        wave_lds_sync();
                        s.sw_trig[wd] = 1;
                    }
                    if ((uint32_t)wl < fw && trg) swsm[wd * 10 + (int)((rp0 + k) % 10u)] = vk;
                }
                wave_lds_sync();
                t += f;
                if (s.count == 960u) dcd_point_on(t - 1u);
                ++n_bulk; n_bulk_samples += f;
            }
            // The sample on which the trigger falls (SyncWord::operator() :187-198: peak search over the stored samples) when it finds the
            // word the state is looking for — the way every frame of a locked stream goes: here as well, and the SYNC_WAIT samples behind
            // it (do_sync_wait :583-593) as a quiet chunk below.  A fall with the other polarity, an EOT hit, the count running out: the
            // single-sample path.
            bool fell = false, point = false;
            if (fw < nw && !((hmask >> fw) & 1ull)) {
                float peak = 0.f;
                uint32_t timing = 0;
                for (int j = 0; j < 10; ++j) {
                    const float fj = swsm[wd * 10 + j];
                    if (fabsf(fj) > fabsf(peak)) { peak = fj; timing = (uint32_t)j; }
                }
                const bool found = s.st == ST_PACKET_SYNC || !(peak > 0.f);   // updated() = peak > 0 ? 1 : -1; :440, :508, :553
                if (found) {
                    fell = true;
                    ring[s.ring_pos] = ywin[t & (WV_WIN - 1)];   // Correlator::sample
                    s.prev_pos = s.ring_pos;
                    s.ring_pos = (s.ring_pos + 1u) % 80u;
                    s.run_pos = min(148, s.run_pos + 1);
                    s.count += 1u;
                    s.ck_count += 1u;
                    s.sync_count += 1;
                    s.sw_trig[wd] = 0;
                    s.sw_timing[wd] = timing;
                    s.sw_updated[wd] = 0;
                    s.missing_sync_count = 0;
                    s.sync_word_type = s.st == ST_STREAM_SYNC ? 1u : (s.st == ST_PACKET_SYNC ? 2u : 3u);
                    if (s.st == ST_STREAM_SYNC) s.eot_flag = 0;
                    s.st = ST_SYNC_WAIT;
                    wave_lds_sync();
                    update_values(timing & 0xFFu);
                    wave_lds_sync();
                    t += 1u;
                    ++n_bulk; n_bulk_samples += 1u;
                    if (s.count == 960u) { dcd_point_on(t - 1u); point = true; }
                }
            }
            tk_search += now() - b0;
            if (fell && !point && !(s.need_clock_reset | s.need_clock_update) && t < P.T) {
                const uint32_t q = s.sync_count < 86 ? (uint32_t)(86 - s.sync_count) : 0u;
                n = min(q + 1u, min(min(P.T - t, (uint32_t)WV_YCH), 960u - s.count));
                completes = n == q + 1u;
                mode = BULK_QUIET;   // (the generic chunk below)
            } else {
                if (f > 0u || fell) continue;
                mode = BULK_NONE;  // the very next sample needs the single-sample path
            }
        }
        if (mode == BULK_SEARCH) {
            // ---- UNLOCKED (do_unlocked :289-342) while no sync word is triggered: up to a whole chunk (480 samples) at once, 64 samples per
            // pass.  The limit history of the chunk is staged from hbuf in one go, lane k of a pass evaluates SyncWord::triggered
            // (Correlator.h:150-157) for its sample; the samples before the first one that triggers are committed as quiet, the triggering
            // one goes through the single-sample path.
            const unsigned long long b0 = now();
            ensure(n);
            const uint32_t rp0 = s.ring_pos;
            float* hb = reinterpret_cast<float*>(DL.soft);   // [3 + n] h0 trajectory: hb[k] = history after sample t - 3 + k
            hpf_ready(); hpf_base = -0x40000000; hw_base = 0x40000000;   // (the staging takes the whole decoder array: the prefetched and the single-sample history windows with it)
#pragma clang loop vectorize(disable) interleave(disable)
            for (uint32_t k = wl; k < n + 3u; k += 64) hb[k] = hrow[(int64_t)t - 3 + k];
            wave_lds_sync();
            const bool phase_a = s.missing_sync_count < 1920;
            uint32_t f = n;                                   // leading samples that are committed here
            for (uint32_t base = 0; base < n; base += 64u) {
                const uint32_t k = base + (uint32_t)wl;
                bool hit = false;
                if (k < n) {
                    const float lim_k = iir_output(hb[3u + k], hb[2u + k], hb[1u + k]);
                    float r[8];
#pragma unroll
                    for (int i = 0; i < 8; ++i) {   // samples k - 70, k - 60, ..., k: from the chunk (window) or from before it (ring)
                        const int32_t o = (int32_t)k - 70 + 10 * i;
                        r[i] = o >= 0 ? ywin[(t + (uint32_t)o) & (WV_WIN - 1)] : ring[(rp0 + 80u + (uint32_t)(o + 80)) % 80u];
                    }
                    auto corr = [&](int w_) { return sync_correlate(w_, r); };
                    auto beyond = [&](int w_, float v) { return v > lim_k * SW_MAG1[w_] || v < lim_k * SW_MAG2[w_]; };
                    hit = phase_a ? beyond(0, corr(0)) : (beyond(1, corr(1)) || beyond(2, corr(2)));
                }
                const unsigned long long mask = __ballot(hit);
                if (mask) { f = base + (uint32_t)(__ffsll((long long)mask) - 1); break; }
            }
            if (f > 0u) {
                const uint32_t first = f > 80u ? f - 80u : 0u;
#pragma clang loop vectorize(disable) interleave(disable) unroll(disable)
                for (uint32_t o = first + wl; o < f; o += 64) ring[(rp0 + o) % 80u] = ywin[(t + o) & (WV_WIN - 1)];
                s.prev_pos = (rp0 + f - 1u) % 80u;
                s.ring_pos = (rp0 + f) % 80u;
                s.run_pos = min(148, s.run_pos + (int32_t)f);
                s.count += f;
                s.ck_count += f;
                if (phase_a) s.missing_sync_count += (int32_t)f;
                wave_lds_sync();
                t += f;
                if (s.count == 960u) dcd_point_on(t - 1u);
                ++n_bulk; n_bulk_samples += f;
                tk_search += now() - b0;
                continue;
            }
            mode = BULK_NONE;  // the very next sample needs the single-sample path
            tk_search += now() - b0;
        }
        if (frame_done) {
            // (the frame chunk above)
        } else if (mode != BULK_NONE) {
            const unsigned long long b0 = now();
            ensure(n);
            const unsigned long long b1 = now();
            tk_ens += b1 - b0;

            const unsigned long long b2 = now();
            tk_sym += b2 - b1;
            {   // Correlator::sample x n: the ring keeps the last 80 samples (the limit filter's history is in hbuf: K2 / nf_serve_limit)
                const uint32_t first = n > 80u ? n - 80u : 0u;
#pragma clang loop vectorize(disable) interleave(disable) unroll(disable)
                for (uint32_t o = first + wl; o < n; o += 64) ring[(s.ring_pos + o) % 80u] = ywin[(t + o) & (WV_WIN - 1)];
                s.prev_pos = (s.ring_pos + n - 1u) % 80u;
                s.ring_pos = (s.ring_pos + n) % 80u;
                s.run_pos = min(148, s.run_pos + (int32_t)n);
            }
            if (mode == BULK_INIT) {
                s.initializing -= (int32_t)n;
                s.count = 0;
                if (s.initializing == 0) nf_snapshot_hist(gs->hist, xr, t + n - 1u, cold_lane());  // the init run ends; the carrier is off
            } else {
                s.count += n;
                s.ck_count += n;
                if (mode == BULK_QUIET) {
                    if (s.st == ST_SYNC_WAIT && completes) { s.sync_count = max(s.sync_count, 86); s.need_clock_update = 1; s.st = ST_FRAME; }
                    else s.sync_count += (int32_t)n;
                }
            }
            wave_lds_sync();
            t += n;
            te = t - 1u;
            tail_dcd = mode != BULK_INIT;
            ++n_bulk; n_bulk_samples += n;
            tk_bulk += now() - b0;
        } else {

        // ---- one input sample: M17Demodulator::operator() :657-753 -----------------------------------------------------------
        const unsigned long long c0 = now();
        ++n_scalar;
        if constexpr (PROF) { if (wl == 0) P.dbg[(size_t)c * DBG_SLOTS + 17 + 1 + min((uint32_t)s.st, 5u)] += 1; }
        const uint32_t tt = t;
        cur_tt = tt;
        s.count++;
        ensure(1u);
        const float filtered = ywin[tt & (WV_WIN - 1)];
        corr_sample(filtered);
        wave_lds_sync();
        if (s.initializing) {
            --s.initializing;
            s.count = 0;
            if (s.initializing == 0) nf_snapshot_hist(gs->hist, xr, tt, cold_lane());
            ++t;
            tk_scalar += now() - c0;
            continue;
        }
        // (the index-0 prologue (:695-709) of this sample has run in chunk selection: every sample goes through it, and nothing in between sets a flag)
        s.ck_count++;
#pragma unroll
        for (int i = 0; i < 8; ++i) r8[i] = 0.f;   // (dead outside this step: without the assignment the eight registers stay live across the whole loop)
        if (s.st <= ST_BERT_SYNC && !(s.st >= ST_STREAM_SYNC && s.sync_count + 1 < 78)) load_r8();  // states that correlate
        // update_values (M17Demodulator.h:233-241) is requested from seven places below and done once behind the switch (nothing
        // in between reads what it writes; where two requests meet in one sample — both words of do_unlocked — each resets the
        // deviation estimator first, so the later one decides alone)
        bool upd_pending = false;
        uint32_t upd_index = 0;
        switch (s.st) {
        case ST_UNLOCKED: {  // do_unlocked :289-342
            if (s.missing_sync_count < 1920) {
                s.missing_sync_count += 1;
                const uint32_t si = sw_step(0);
                if (sw_updated(0)) {
                    s.sync_count = 0; s.missing_sync_count = 0; s.need_clock_reset = 1;
                    dev_reset(); s.sample_index = si; upd_pending = true; upd_index = si;
                    s.st = ST_LSF_SYNC;
                }
                break;
            }
            uint32_t si = sw_step(1);
            int32_t up = sw_updated(1);
            if (up) {
                s.sync_count = 86; s.missing_sync_count = 0; s.need_clock_reset = 1;
                dev_reset(); s.sample_index = si; upd_pending = true; upd_index = si;
                s.st = ST_FRAME;
                s.sync_word_type = up < 0 ? 1u : 0u;
            }
            si = sw_step(2);
            up = sw_updated(2);
            if (up < 0) {
                s.sync_count = 86; s.missing_sync_count = 0; s.need_clock_reset = 1;
                dev_reset(); s.sample_index = si; upd_pending = true; upd_index = si;
                s.st = ST_FRAME;
                s.sync_word_type = 3u;
            }
            break;
        }
        case ST_LSF_SYNC: {  // do_lsf_sync :350-411
            if (corr_index() != s.sample_index) break;
            float sync_triggered = sw_triggered(0);
            if ((double)sync_triggered > 0.1) { s.need_clock_update = 1; s.sync_count += 1; break; }
            sync_triggered = sw_triggered(1);
            const float bert_triggered = sw_triggered(2);
            if (bert_triggered < 0.f) {
                s.missing_sync_count = 0; s.sync_count = 86; s.need_clock_update = 1;
                upd_pending = true; upd_index = s.sample_index; s.st = ST_FRAME; s.sync_word_type = 3u;
            } else if ((double)fabsf(sync_triggered) > 0.1) {
                s.missing_sync_count = 0; s.sync_count = 86; s.need_clock_update = 1;
                upd_pending = true; upd_index = s.sample_index; s.st = ST_FRAME;
                s.sync_word_type = sync_triggered > 0.f ? 0u : 1u;
            } else if (++s.missing_sync_count > 192) {
                if (s.sync_count >= 10) { s.missing_sync_count = 0; s.need_clock_update = 1; }
                else { s.sync_count = 0; s.st = ST_UNLOCKED; s.missing_sync_count = 0; if (s.dcd_trig) despec(tt); s.dcd_trig = 0; }
            } else {
                upd_pending = true; upd_index = s.sample_index;
            }
            break;
        }
        case ST_STREAM_SYNC:   // do_stream_sync :420-482, do_packet_sync :489-530, do_bert_sync :536-574
        case ST_PACKET_SYNC:
        case ST_BERT_SYNC: {
            s.sync_count += 1;
            if (s.sync_count < 78) break;
            const uint32_t mode_st = s.st;
            if (mode_st == ST_STREAM_SYNC && sw_triggered(3) > 0.1f) {
                s.sync_word_type = 1u; s.st = ST_FRAME; s.eot_flag = 1; s.missing_sync_count = 0;
                break;
            }
            uint32_t si;
            int32_t up;
            if (mode_st == ST_STREAM_SYNC) { si = sw_step(1) & 0xFFu; up = sw_updated(1); }
            else { si = sw_step(2) & 0xFFu; up = sw_updated(2); }
            const bool hit = (mode_st == ST_PACKET_SYNC) ? (up != 0) : (up < 0);
            const uint32_t swt = (mode_st == ST_STREAM_SYNC) ? 1u : (mode_st == ST_PACKET_SYNC ? 2u : 3u);
            if (hit) {
                s.missing_sync_count = 0; upd_pending = true; upd_index = si;
                s.sync_word_type = swt; s.st = ST_SYNC_WAIT;
                if (mode_st == ST_STREAM_SYNC) s.eot_flag = 0;
            } else if (s.sync_count > 86) {
                const uint32_t limit = (mode_st == ST_PACKET_SYNC) ? 60u : 80u;
                resolve_cost();
                if (s.viterbi_cost < limit) {
                    if (!s.missing_sync_count) s.missing_sync_count = 1;
                    s.sync_word_type = swt; s.st = ST_FRAME;
                } else if (mode_st == ST_STREAM_SYNC && s.eot_flag) {
                    s.st = ST_UNLOCKED; if (s.dcd_trig) despec(tt); s.dcd_trig = 0;
                } else if (s.missing_sync_count < 10) {
                    s.missing_sync_count += 1; s.sync_word_type = swt; s.st = ST_FRAME;
                } else {
                    s.st = ST_UNLOCKED; if (s.dcd_trig) despec(tt); s.dcd_trig = 0;
                }
                if (mode_st == ST_STREAM_SYNC) s.eot_flag = 0;
            }
            break;
        }
        case ST_SYNC_WAIT:  // do_sync_wait :583-593
            if (s.sync_count < 86) { s.sync_count += 1; break; }
            s.need_clock_update = 1;
            s.st = ST_FRAME;
            break;
        default: {  // do_frame :596-654
            const int d = (int)s.sample_index - (int)corr_index();
            if (abs(d) == 5) {
                s.ck_sample_index = clock_predict(s.ck_sample_est, s.ck_clock_est, s.ck_count);
                s.sample_index = (uint32_t)(uint8_t)s.ck_sample_index;
            } else if (corr_index() == s.sample_index) {
                float err;
                const float sample = normalise(filtered, err);
                if (P.ev_ops) ev_op((err * err) * alpha);
                else {
                    s.evm_S = s.evm_S - s.evm_S * alpha;
                    s.evm_S = s.evm_S + (err * err) * alpha;
                }
                llr16[s.framer_idx >> 1] = (uint16_t)slice_llr(sample, edges);  // llr<float,4> + M17Framer :42-53
                s.framer_idx += 2;
                if (s.framer_idx == 368u) {
                    s.framer_idx = 0;
                    s.sync_count = 0;
                    decode_due = true;
                }
            }
            break;
        }
        }
        if (upd_pending) update_values(upd_index);
        wave_lds_sync();
        te = tt;
        tail_dcd = true;
        ++t;
        tk_scalar += now() - c0;
        }
        // ---- common tail: frame decode and the carrier-on update point ----------------------------------------------------
        if (decode_due) {  // decoder(...) and the rest of do_frame (:619-642)
            const unsigned long long d0 = now();
            hpf_ready(); hpf_base = -0x40000000;   // the decoder takes the cost-word array
            const uint2 r = nf_decode_wave(P.tables, DL, wl, s.sync_word_type, cd, s.viterbi_cost, rec_base, P.rec_cap, P.channel_base + c, P.pos0 + te, P.overflow,
                                           P.defer ? P.defer + (size_t)c * P.rec_cap * 46 : nullptr);
            s.viterbi_cost = r.x;
            s.st = (r.y == 1u || r.y == 0u) ? ST_STREAM_SYNC : (r.y == 4u ? ST_BERT_SYNC : ST_PACKET_SYNC);
            ++n_decode;
            {   // the limit history the next sync window (77 samples on) will want: on its way into LDS while the next chunk is set up
                const int32_t target = (int32_t)t + 77 - 3;
                if (t < P.T && (!diverged || target + 64 <= (int32_t)h_until)) hpf_issue(target);
            }
            tk_decode += now() - d0;
        }
        { const unsigned long long q0 = now(); if (tail_dcd && s.count == 960u) dcd_point_on(te); tk_tail += now() - q0; }
    }

    // ---------------- save state ------------------------------------------------------------------------------
    if (!diverged) { const float* f = P.final_h + (size_t)c * 4; s.h0 = f[0]; s.h1 = f[1]; s.h2 = f[2]; }
    else if (s.initializing || s.dcd_on) pick_hist(P.T - 1u);   // (gate off: the history was picked where it froze)
    if (P.dropped && wl == 0) P.dropped[c] = left_replay ? 1u : 0u;
    s.store(as_lds(hot_lds));
    wave_lds_sync();
    {
        const uint32_t* src = reinterpret_cast<const uint32_t*>(hot_lds);
        uint32_t* dst = reinterpret_cast<uint32_t*>(&gs->hot);
        for (int k = wl; k < (int)(sizeof(Hot) / 4); k += 64) dst[k] = src[k];
    }
    Diag d = lds_get(&cd->diag);
    d.demod_state = s.st;
    d.n_frames = cd->seq;
    d.pad[0] = s.ck_count;   // live counters at the end of the run (debugging aid, same words as the oracle's)
    d.pad[1] = ((uint32_t)s.sync_count & 0xFFFFu) | ((uint32_t)s.missing_sync_count << 16);
    lds_put(&cd->diag, d);
    wave_lds_sync();
    {
        const M17_LDS uint32_t* src = reinterpret_cast<const M17_LDS uint32_t*>(cd);
        uint32_t* dst = reinterpret_cast<uint32_t*>(&gs->cold);
        for (int k = wl; k < (int)(sizeof(Cold) / 4); k += 64) dst[k] = src[k];
    }
    for (int k = wl; k < 80; k += 64) gs->ring[k] = ring[k];
    for (int k = wl; k < 40; k += 64) gs->sw_samples[k / 10][k % 10] = swsm[k];
    for (int k = wl; k < 92; k += 64) gs->llr[k] = DL.llr[k];
    for (int k = wl; k < 8; k += 64) gs->lsf[k] = DL.lsf[k];
    P.rec_count[c] = cd->n_run;
    if (P.truth_out && wl == 0) {
        GateTruth gt;
        gt.init = s.initializing; gt.on = s.dcd_on; gt.trig = s.dcd_trig; gt.count = s.count; gt.level = cd->dcd_level; gt.seg = cd->seg_start_tick;
        P.truth_out[c] = gt;
        if (!(s.initializing > 0 || s.dcd_on)) atomicAdd(P.overflow + 3, 1u);
    }
    if (P.ev_cursor_out && wl == 0) {
        P.ev_cursor_out[c] = cd->ev_cursor;
        if (cd->ev_cursor > P.ev_pitch) atomicOr(P.overflow + 2, 1u);   // operations were dropped: the EVM of this run's diagnostics is not to be trusted (m17hip_diag_fetch says so)
    }
    if (P.diag_log && wl == 0) P.diag_count[c] = cd->n_diag_run;
    if constexpr (TIMED) if (wl == 0) {
        unsigned long long* slot = P.dbg + (size_t)c * DBG_SLOTS + ((P.flags >> 8) & 31u);
        *slot = (wall_clock64() - *slot) | (diverged ? 1ull << 62 : 0ull);
    }
    if constexpr (PROF) if (wl == 0) {
        unsigned long long* o = P.dbg + (size_t)c * DBG_SLOTS;
        o[8] = tk_patch; o[12] = tk_ens; o[13] = tk_sym; o[14] = tk_iir; o[15] = tk_search; o[16] = tk_off; o[17] = n_despec;
        o[0] = now() - tk0; o[1] = tk_bulk; o[2] = tk_scalar; o[3] = tk_decode;
        o[4] = n_bulk; o[5] = n_scalar; o[6] = n_bulk_samples; o[7] = n_flip | (n_decode << 32);
        for (int k = 0; k < 8; ++k) o[24 + k] = n_mode[k];
        o[32] = n_flip; o[33] = n_lim_clock; o[34] = n_lim_count; o[35] = n_lim_room; o[36] = tk_sel; o[37] = tk_tail;
    }
    if (P.bnd_out && left_replay) {   // what the replay needs to take this channel up again: its state at this boundary
        Boundary* b = P.bnd_out + c;
        if (cold_lane() == 0) {
            b->init = s.initializing; b->on = s.dcd_on; b->trig = s.dcd_trig; b->count = s.count; b->run_pos = s.run_pos;
            b->h0 = s.h0; b->h1 = s.h1; b->h2 = s.h2; b->level = cd->dcd_level; b->seg = cd->seg_start_tick;
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
        const uint32_t* hs = reinterpret_cast<const uint32_t*>(gs->hist);
        uint32_t* hd = reinterpret_cast<uint32_t*>(b->hist);
        for (int k = cold_lane(); k < 75; k += 64) hd[k] = hs[k];
    }
}

}  // namespace m17
