// Stand-alone operator kernels behind the per-operator C-ABI entry points (parity API): slicer/EVM, Viterbi,
// frame decoder.  They reuse the device functions of the full-chain kernel, one lane per row / frame.
#pragma once

#include "m17_common.hpp"
#include "m17_decode_device.hpp"
#include "m17_state.hpp"

namespace m17 {

// a11/a12 stand-alone (parity API): normalised symbols -> LLR pairs (Util.h:128-145) and the running EVM
// (SymbolEvm.h:31-51) of each row.  One lane per row.
__global__ __launch_bounds__(64) void slice_kernel(const float* __restrict__ sym, uint32_t rows, uint32_t n, int8_t* __restrict__ llr,
                                                   float* __restrict__ evm, const float* __restrict__ edges)
{
    const uint32_t r = blockIdx.x * 64 + threadIdx.x;
    if (r >= rows) return;
    float S = 0.f;  // evm.reset()
    const float alpha = (float)(1.0 / 184);
    for (uint32_t k = 0; k < n; ++k) {
        const float sample = sym[(size_t)r * n + k];
        const uint32_t pair = slice_llr(sample, edges);
        llr[((size_t)r * n + k) * 2] = (int8_t)(pair & 0xFF);
        llr[((size_t)r * n + k) * 2 + 1] = (int8_t)(pair >> 8);
        float e;
        if (sample > 2.f) e = sample - 3.f;
        else if (sample > 0.f) e = sample - 1.f;
        else if (sample > -2.f) e = sample + 1.f;
        else e = sample + 3.f;
        S = S - S * alpha;
        S = S + (e * e) * alpha;
        evm[(size_t)r * n + k] = sqrtf(S);
    }
}

// a9/a10 stand-alone (parity API): the 2-state Kalman update of KalmanFilter.h:41-65 / :91-107 (kal_update, the function the
// full-chain kernel calls) on `rows` independent filters, n updates each: z[r][i] after dt[r][i] samples.
// out[r][i][6] = x0, x1, P00, P01, P10, P11 after update i.  One lane per row, the filter state in LDS as in K5.
// wrap == -1: a LEVEL filter in the form K5 runs it (core::level_update with the gain schedule `gain` of this order, dt = 192 whatever
// dt[] says): out[..][0..1] = x0, x1, the covariance slots are zero (the schedule holds it).
__global__ __launch_bounds__(64) void kalman_kernel(const float* __restrict__ z, const uint32_t* __restrict__ dt, uint32_t rows, uint32_t n, int wrap,
                                                    float z0, uint32_t order, float* __restrict__ out, const core::Kalman2Gain* __restrict__ gain)
{
    __shared__ Kal2 st[64];
    const uint32_t r = blockIdx.x * 64 + threadIdx.x;
    if (r >= rows) return;
    if (wrap == -1) {
        float x0 = z0, x1 = 0.f;
        for (uint32_t i = 0; i < n; ++i) {
            core::level_update(x0, x1, z[(size_t)r * n + i], gain[min(i, (uint32_t)core::LEVEL_SCHED_LAST)], order);
            float* o = out + ((size_t)r * n + i) * 6;
            o[0] = x0; o[1] = x1; o[2] = 0.f; o[3] = 0.f; o[4] = 0.f; o[5] = 0.f;
        }
        return;
    }
    M17_LDS Kal2* kp = as_lds(&st[threadIdx.x]);
    Kal2 k;
    kal_reset(k, z0);
    lds_put(kp, k);
    for (uint32_t i = 0; i < n; ++i) {
        kal_update(kp, z[(size_t)r * n + i], dt[(size_t)r * n + i], wrap, order);
        k = lds_get(kp);
        float* o = out + ((size_t)r * n + i) * 6;
        o[0] = k.x0; o[1] = k.x1; o[2] = k.p00; o[3] = k.p01; o[4] = k.p10; o[5] = k.p11;
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
    L.stride = 64;
    L.prof = nullptr;
    L.src = &ident->src[0][0];
    L.lich_src = ident->lich_src;
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
        int stale = llr_at(L.llr, 64, lane, kind == 3 ? 401 : 0);  // BERT callers pass position 401 explicitly
        const uint32_t cst = viterbi_decode(ident, L, lane, kind + 4, stale);  // tables 4..7: identity source map
        cost[f] = (int32_t)cst;
        for (int n = 0; n < OUT; ++n) bits[(size_t)f * OUT + n] = (uint8_t)((byte_at(L.outb, 64, lane, n >> 3) >> (7 - (n & 7))) & 1u);
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
    L.stride = 64;
    L.prof = nullptr;
    L.src = &P.tables->src[0][0];
    L.lich_src = P.tables->lich_src;
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
    for (int k = 0; k < 30; ++k) P.lsf_io[(size_t)f * 30 + k] = (uint8_t)byte_at(L.lsf, 64, lane, k);
}

// Deferred frame decode (m17_wave_kernel.hpp hands payload frames of running stream / BERT transmissions over instead of decoding
// them in the channel's wave): one workgroup per channel, one LANE per frame record — viterbi_decode, the lane-per-frame form of
// Viterbi<Trellis<4,2>,4>::decode (Viterbi.h:162-239) with the same source maps — cost and payload go into the record the wave
// reserved.  Then the tags that stood for those costs are replaced in the entries of the channel's diagnostic log (the two places of the
// demodulator STATE that can hold a tag at the end of a run are settled before, by settle_tail_kernel: this kernel runs on a stream of
// its own, beside the next run of a continued stream, and works on the record set of its run only).
struct DeferParams {
    FrameRec* recs;               // [C][rec_cap]
    uint32_t rec_cap;
    const uint32_t* rec_count;    // [C] records of this run
    const uint32_t* defer;        // [C][rec_cap][46] LLR nibbles
    uint32_t* hist;               // [C][101][64] decision words of the frames a workgroup is working on (global: LDS is what limits its waves)
    const DecodeTables* tables;
    SeqState* state;
    Diag* diag_log;               // optional [C][diag_cap]
    uint32_t diag_cap;
    const uint32_t* diag_count;   // [C]
    uint32_t C;
    EvParams ev;                  // the run's deferred EVM (m17_state.hpp): folded by the blocks behind the first C of the launch (ops == nullptr: none)
};
constexpr int DEFER_HIST_WORDS = 201;                                   // 201 trellis steps (BERT), one decision word per step
// Two channels (waves) per workgroup share the source maps: 31.5 KB per two waves = ten waves per CU instead of eight (LDS is what limits this kernel's waves, and
// what it lasts is what those waves can issue).
constexpr int DEFER_CPB = 2;                                            // channels (waves) per workgroup
constexpr int DEFER_WAVE_WORDS = (46 + 8) * 64 + 512 / 2;               // per wave: LLR nibbles, output bytes, the list of deferred slots
constexpr int DEFER_LDS_BYTES = DEFER_CPB * DEFER_WAVE_WORDS * 4 + 4 * 488 * 2;   // + the source maps: 33.5 KB
constexpr uint32_t defer_blocks(uint32_t C) { return (C + DEFER_CPB - 1) / DEFER_CPB; }
__global__ __launch_bounds__(64 * DEFER_CPB) void decode_deferred_kernel(DeferParams P)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
    if (blockIdx.x >= defer_blocks(P.C)) {    // (a handful of blocks: the EVM fold runs beside the decode instead of in front of or behind it)
        if (threadIdx.x < 64) evm_fold_pass(P.ev, blockIdx.x - defer_blocks(P.C), reinterpret_cast<float*>(lds));
        return;
    }
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    uint32_t* wb = lds + wave * DEFER_WAVE_WORDS;
    DecodeLds L;
    L.llr = wb;                 // [46][64] nibbles
    L.outb = wb + 46 * 64;      // [8][64]
    uint16_t* list = reinterpret_cast<uint16_t*>(L.outb + 8 * 64);    // [<= 512] the slots whose frames are deferred, in order
    uint16_t* maps = reinterpret_cast<uint16_t*>(lds + DEFER_CPB * DEFER_WAVE_WORDS);   // [4][488] source maps (every trellis step reads two entries)
    L.lsf = nullptr;
    L.stride = 64;
    L.prof = nullptr;
    L.soft = nullptr;
    const int lane = threadIdx.x & 63;
    const uint32_t c = blockIdx.x * DEFER_CPB + wave;
    L.hist = P.hist + (size_t)c * DEFER_HIST_WORDS * 64;
    for (int k = threadIdx.x; k < 4 * 488; k += 64 * DEFER_CPB) maps[k] = P.tables->src[k / 488][k % 488];
    __syncthreads();            // the only block-level barrier: the waves of a block are independent from here on
    L.src = maps;
    L.lich_src = P.tables->lich_src;
    if (c >= P.C) return;
    FrameRec* recs = P.recs + (size_t)c * P.rec_cap;
    const uint32_t n = min(P.rec_count[c], P.rec_cap);
    for (uint32_t s0 = 0; s0 < n; s0 += 512) {   // (512 slots at a time: the list's room; a 10 s run has 508)
        // the deferred slots as a dense list: a stream's records alternate LICH / payload — with a lane per SLOT half the lanes idled
        const uint32_t s1 = min(n, s0 + 512u);
        uint32_t nd = 0;
        wave_lds_sync();
        for (uint32_t base = s0; base < s1; base += 64) {
            const uint32_t slot = base + lane;
            bool d = false;
            if (slot < s1) {
                const uint32_t* w = reinterpret_cast<const uint32_t*>(recs + slot);
                d = cost_is_deferred(w[4]) && w[15] == DEFER_MARK;
            }
            const unsigned long long mask = __ballot(d);
            if (d) list[nd + __popcll(mask & ((1ull << lane) - 1ull))] = (uint16_t)(slot - s0);
            nd += (uint32_t)__popcll(mask);
        }
        wave_lds_sync();
        for (uint32_t k0 = lane; k0 < nd; k0 += 64) {
            const uint32_t slot = s0 + as_lds(list)[k0];
            uint32_t* w = reinterpret_cast<uint32_t*>(recs + slot);
            const uint32_t* src = P.defer + ((size_t)c * P.rec_cap + slot) * 46;
            for (int k = 0; k < 46; ++k) as_lds(L.llr)[k * 64 + lane] = src[k];
            const int kind = kind_of_frame_type(w[5] & 0xFFu);
            int stale = (int)w[14];
            const uint32_t cost = viterbi_decode_pk(L, lane, kind, stale);
            complete_record(w, cost, L.outb, 64, lane, len_of_kind(kind));
        }
    }
    __threadfence_block();
    wave_lds_sync();
    auto settle = [&](int32_t& v) {
        const uint32_t u = (uint32_t)v;
        if (cost_is_deferred(u)) v = recs[u & ~DEFER_TAG].cost;
    };
    // (the channel's saved viterbi_cost and its last diagnostic callback carry no tag any more when this kernel runs: settle_tail_kernel
    //  below has decoded the one or two frames they stood for — the demodulator state is the NEXT run's by now and is not touched here)
    if (P.diag_log) {
        const uint32_t nd = min(P.diag_count[c], P.diag_cap);
        Diag* log = P.diag_log + (size_t)c * P.diag_cap;
        for (uint32_t e = lane; e < nd; e += 64) settle(log[e].viterbi_cost);
    }
}

// The end of a run on the MAIN stream, so that nothing of the demodulator state waits for the deferred decode (2 ms per 4096 x 480 000,
// which then runs beside the next run's chain, on the payload stream).  At the end of a run a channel's state can hold a deferred-cost tag
// in exactly two places: the saved viterbi_cost (M17Demodulator.h:146 `viterbi_cost`: the last frame's, consulted by the missed-sync branches
// :453,508,555 of the NEXT run) and the viterbi_cost argument of its last diagnostic callback (m17_diag) — the last frame, and the one before
// it when a frame completed after that callback.  One WAVE per channel decodes those one or two frames with the wave decoder (16 lanes = 16
// states) from the deferred-frame store, completes their records (decode_deferred_kernel then skips them) and puts the costs in place.
// The blocks behind the first C fold the rest of the run's deferred EVM operations (evm_fold_pass, the last pass of a run).
struct SettleParams {
    FrameRec* recs; uint32_t rec_cap; const uint32_t* defer; const DecodeTables* tables; SeqState* state; uint32_t C;
    EvParams ev;   // ops == nullptr: no fold blocks in the launch
};
constexpr int SETTLE_LDS_BYTES = (92 + 122 + 8 + 488) * 4 > EV_TILE_FLOATS * 4 ? (92 + 122 + 8 + 488) * 4 : EV_TILE_FLOATS * 4;
__global__ __launch_bounds__(64) void settle_tail_kernel(SettleParams P)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
    if (blockIdx.x >= P.C) {
        evm_fold_pass(P.ev, blockIdx.x - P.C, reinterpret_cast<float*>(lds));
        return;
    }
    const int wl = threadIdx.x;
    const uint32_t c = blockIdx.x;
    SeqState* st = P.state + c;
    const uint32_t th = (uint32_t)__builtin_amdgcn_readfirstlane((int)st->hot.viterbi_cost);
    const uint32_t td = (uint32_t)__builtin_amdgcn_readfirstlane(st->cold.diag.viterbi_cost);
    if (!cost_is_deferred(th) && !cost_is_deferred(td)) return;
    DecodeLds L;
    L.llr = lds; L.hist = lds + 92; L.outb = lds + 92 + 122; L.soft = reinterpret_cast<int32_t*>(lds + 92 + 122 + 8);
    L.lsf = nullptr; L.prof = nullptr; L.stride = 1;
    L.src = &P.tables->src[0][0]; L.lich_src = P.tables->lich_src;
    auto resolve = [&](uint32_t tag) -> uint32_t {
        const uint32_t slot = tag & ~DEFER_TAG;
        uint32_t* w = reinterpret_cast<uint32_t*>(P.recs + (size_t)c * P.rec_cap + slot);
        const uint32_t have = (uint32_t)__builtin_amdgcn_readfirstlane((int)w[4]);
        if (!cost_is_deferred(have) || slot >= P.rec_cap) return have;   // the wave decoded it after all (a missed sync word asked for its cost)
        const int kind = kind_of_frame_type(w[5] & 0xFFu);
        int stale = (int)w[14];
        const uint32_t* src = P.defer + ((size_t)c * P.rec_cap + slot) * 46;
        for (int k = wl; k < 92; k += 64) as_lds(L.llr)[k] = unpack_llr_nibbles(src[k >> 1], k & 1);
        wave_lds_sync();
        const uint32_t cost = viterbi_decode_wave(L, wl, kind, stale);
        wave_lds_sync();
        if (wl == 0) complete_record(w, cost, L.outb, 1, 0, len_of_kind(kind));
        wave_lds_sync();
        return cost;
    };
    uint32_t ch = th, cd = td;
    if (cost_is_deferred(th)) ch = resolve(th);
    if (cost_is_deferred(td)) cd = (td == th) ? ch : resolve(td);
    if (wl == 0) {
        st->hot.viterbi_cost = ch;
        st->cold.diag.viterbi_cost = (int32_t)cd;
    }
}

}  // namespace m17
