// K2: limit_track_kernel — the correlator's limit filter (reference a4: Correlator::sample, Correlator.h:43-49,
// BaseIirFilter<float,3>, IirFilter.h:26-42) run AHEAD of the state machine.
//
// The limit IIR is a float recurrence over every sample the correlator is fed: 3 dependent VALU instructions per
// sample that cannot be parallelised in time, and in K5 (one wave per channel) they cost a whole wave instruction
// each — 43 % of K5's time.  Which samples are fed is decided by the carrier-detect gate, and the gate
// follows the DCD table (K3) alone EXCEPT when the demodulator forces dcd.unlock() after losing sync
// (M17Demodulator.h:396-404, 470-478 ...).  So this kernel replays the gate from the table under the assumption "no
// forced unlock happens in this run", advances the IIR over exactly the samples the gate lets through — SIXTEEN CHANNELS
// PER WAVE instruction — and leaves the filter history after every fed sample in hbuf.  K5 starts every run trusting
// hbuf; the moment it forces an unlock while the gate trigger was set it drops the speculation for the rest of the
// run, picks the filter state up from hbuf at that sample and carries the recurrence itself (the pre-existing path).
// Every run starts from K5's own saved state, so a dropped speculation never outlives its run.
//
// Indexing: hbuf[c][YPRE + t] = h0 after sample t was fed (same pitch and prefix as ybuf).  Where a gated run starts
// the three slots before its first sample are overwritten with the history the run inherits (h0, h1, h2 of the
// previous run's end), so that for EVERY fed sample t: (hbuf[t], hbuf[t-1], hbuf[t-2]) is the filter history.
// The first 148 matched-filter outputs of a run see the previous run's tail (Q2): they are recomputed here exactly
// as K5's patch_run_start does, into LDS only — ybuf is never written by this kernel.
//
// Mapping: 4 lanes per channel, 16 channels per wave: loads, stores and the FIR patch are cooperative, the recurrence
// itself runs on lane 0 of each 4-lane group (the filter history lives there only) — one VALU instruction advances 16
// channels.  The kernel is latency-bound (a lone wave per SIMD); few, fat waves keep the issue slots it takes from the
// kernels running beside it small.
#pragma once

#include "m17_common.hpp"
#include "m17_decode_device.hpp"
#include "m17_frontend_kernels.hpp"
#include "m17_state.hpp"

namespace m17 {

// The replay's own state at the end of a segment: start of the NEXT segment's replay, which runs while K5 is still busy with
// this one (valid for every channel whose K5 did not drop the speculation; the others are redone from K5's state).
struct GateExport {
    int32_t init;
    uint32_t on, trig, count;
    int32_t run_pos;
    float h0, h1, h2;
    float level;
    uint32_t seg;
    int32_t end_in_run, end_t;   // end_t relative to the next segment's first sample (negative)
};

struct GateParams {
    const int16_t* x;
    size_t xpitch;
    const float* y;           // K1 output, read-only here
    size_t ypitch;
    float* h;                 // hbuf, same pitch as y
    const float* dcd_table;   // [C][ticks_cap][12]
    uint32_t ticks_cap;
    const SeqState* state;    // K5's state at the end of the previous segment (authoritative start of the replay)
    const GateExport* chain_in;   // non-null: start from the replay's own state instead (K5 has not finished the previous segment yet)
    GateExport* chain_out;        // the replay's state at the end of this segment
    const uint32_t* only;         // non-null: redo only the channels flagged here (K5 dropped the speculation in the previous segment)
    const Boundary* bnd;          // with `only`: the flagged channels start from these records [C] instead of P.state (K5 is at work on the next segment and on that state)
    float* final_h;           // [C][4]: h0, h1, h2 after the last fed sample of this run
    const float* taps;        // 149 floats
    uint32_t C, T;
    uint64_t pos0;            // absolute index of sample 0 of this (segment of a) run
    uint64_t tick_row0;       // absolute tick stored in row 0 of the DCD table
    uint32_t flags;
    uint32_t nblk;            // workgroups of the replay itself; the ones behind them fold deferred EVM operations (m17_state.hpp, evm_fold_pass)
    EvParams ev;
};

constexpr int GT_LPC = 4;             // lanes per channel (cooperative loads / stores; the recurrence runs on the first of them)
constexpr int GT_CPW = 64 / GT_LPC;   // channels per wave
// Lane l = channel l % 16 of the wave, helper l / 16 of that channel: the sixteen lanes that carry the recurrence are lanes 0..15 — ONE
// quarter of the wave.  A packed instruction whose enabled lanes lie in one quarter is through the pipeline sooner (the dependent
// chain of the filter: 7.3 ns per sample on sixteen contiguous lanes, 12.3 on sixteen lanes spread over the four quarters;
// tools/serve_bench2.hip), and with rows of TICK + 4 words the sixteen rows start in sixteen different LDS bank groups for the
// recurrence's 16-byte reads as well as for the staging (rows of TICK words: all sixteen in the same banks).
constexpr int GT_ROW = TICK + 4;      // LDS row of a channel's tick (words)
constexpr int GT_F4 = TICK / 4 / GT_LPC;  // float4 per lane per tick
constexpr int GT_LDS_FLOATS = GT_CPW * GT_ROW + GT_CPW * 148 + 298;   // 23.2 KB

// One pass over the segment P names for the sixteen channels of a wave.  Per lane: `valid` = the channel takes part (stores, exports its
// end state); a lane that does not idles.  `from_chain` (wave-uniform) = the pass starts from the replay's own state, else from K5's:
// P.state, or the boundary record `bnd_base[c]` (the redo).  `state_only`: no history value is stored (the redo).
__device__ __forceinline__ void limit_track_pass(const GateParams& P, bool state_only, uint32_t c, bool valid, bool from_chain, const Boundary* bnd_base,
                                                 float* lds_base)
{
    constexpr uint32_t t0 = 0;
    const uint32_t segT = P.T;
    const size_t fh_off = 0;
    const Boundary* bnd = bnd_base ? bnd_base + c : nullptr;   // (bnd_base: wave-uniform)
    const uint64_t pos0 = P.pos0 + t0;
    // matched-filter samples of the current piece, replaced IN PLACE by h0 after each of them (the recurrence reads a sample, or the
    // block of samples ahead of it, before it stores the history value over it): one 12 KB array instead of two
    float (&yl)[GT_CPW][GT_ROW] = *reinterpret_cast<float (*)[GT_CPW][GT_ROW]>(lds_base);
    float (&hl)[GT_CPW][GT_ROW] = yl;
    float (&pl)[GT_CPW][148] = *reinterpret_cast<float (*)[GT_CPW][148]>(lds_base + GT_CPW * GT_ROW);   // patched first outputs of the current run
    float (&pw)[298] = *reinterpret_cast<float (*)[298]>(lds_base + GT_CPW * GT_ROW + GT_CPW * 148);      // patch window: 149 snapshot + 148 run samples
    const int lane = threadIdx.x;
    const int g = lane % GT_CPW, r = lane / GT_CPW;
    const bool invert = P.flags & 1u;
    // flags bit 1: a replay whose history values nobody will read (the channels K5 serves itself, m17_wave_kernel.hpp): only the
    // replay's end state is wanted, hbuf is left alone (K5 is writing those very rows)
    const bool store = !state_only;
    const SeqState* gs = P.state + c;
    const int16_t* xr = P.x + (size_t)c * P.xpitch + XPRE + t0;
    const float* yr = P.y + (size_t)c * P.ypitch + YPRE + t0;
    float* hr = P.h + (size_t)c * P.ypitch + YPRE + t0;
    const float* tab = P.dcd_table + (size_t)c * P.ticks_cap * 12;

    // the gate's state (M17Demodulator members dcd_, count_, initializing; DataCarrierDetect level/trigger) and the filter
    // history: as K5 left them, or as the replay of the previous segment left them
    int32_t init, run_pos, end_t = 0;
    uint32_t on, trig, count, seg;
    float h0, h1, h2, level;
    bool end_in_run = false;   // the previous gated run ended inside this slab, at relative sample end_t
    if (from_chain) {
        const GateExport e = P.chain_in[c];
        init = e.init; on = e.on; trig = e.trig; count = e.count; run_pos = e.run_pos;
        h0 = e.h0; h1 = e.h1; h2 = e.h2; level = e.level; seg = e.seg;
        end_in_run = e.end_in_run != 0; end_t = e.end_t;
    } else if (bnd) {
        init = bnd->init; on = bnd->on; trig = bnd->trig; count = bnd->count; run_pos = bnd->run_pos;
        h0 = bnd->h0; h1 = bnd->h1; h2 = bnd->h2; level = bnd->level; seg = bnd->seg;
    } else {
        init = gs->hot.initializing;
        on = gs->hot.dcd_on; trig = gs->hot.dcd_trig; count = gs->hot.count;
        run_pos = gs->hot.run_pos;
        h0 = gs->hot.h0; h1 = gs->hot.h1; h2 = gs->hot.h2;
        level = gs->cold.dcd_level;
        seg = gs->cold.seg_start_tick;
    }
    // A lane whose channel takes no part in this pass (a redo pass runs for the flagged channels only; lanes beyond the last channel)
    // idles with the carrier off: it is never fed, never patched and does not keep the other channels of the wave off the steady-state
    // path.  (Its record in `bnd` / its live state must not steer anything: the record may never have been written.)
    if (!valid) { init = 0; on = 0; trig = 0; count = 0; run_pos = 148; h0 = h1 = h2 = 0.f; level = 0.f; seg = 0; end_in_run = false; end_t = 0; }
    // Started from K5's state: the three slots in front of the segment get the history it starts with.  (What is there is the
    // previous segment's or run's tail, which is void if K5 had dropped the speculation there; a replay that continues from its own
    // state finds its own output there.)
    if (!from_chain && r == 0 && valid && store) { hr[-1] = h0; hr[-2] = h1; hr[-3] = h2; }
    bool pl_valid = false;     // pl[g] holds the patched outputs of the current run
    int32_t pl_rs = 0;         // relative index of that run's first sample

    auto lds_sync = [] {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront", "local");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront", "local");
    };
    // DataCarrierDetect::update (:63-69) on the sums K3 left for the segment that ends with tick k (same arithmetic as
    // nf_dcd_update in m17_state.hpp)
    auto dcd_update = [&](uint64_t k, float l1, float l2) {
        level = (float)((double)level * 0.8 + 0.2 * (double)(l1 / l2));
        seg = (uint32_t)(k + 1);
        trig = trig ? (level > 0.1f) : (level > 4.0f);
    };

    uint32_t t = 0;
    uint32_t phase = (uint32_t)(pos0 % TICK);
    uint64_t k_cur = pos0 / TICK;   // absolute index of the tick the current piece lies in
    float4 pre[GT_F4];         // this lane's share of a tick of matched-filter samples, loaded one tick ahead
#pragma unroll
    for (int b = 0; b < GT_F4; ++b) pre[b] = make_float4(0.f, 0.f, 0.f, 0.f);
    uint32_t pre_t = 0xFFFFFFFFu;
    // whole aligned tick, no patched outputs: lane r stages float4 r, r + GT_LPC, ... (16-byte loads issued one tick ahead)
    auto stage_tick = [&]() {
        const float4* src = reinterpret_cast<const float4*>(yr + t) + r;
        if (pre_t != t) {            // not requested a tick ahead (the first tick of a streak)
            if (valid) {             // (an idle lane loads nothing: a redo pass of one channel does not read its fifteen neighbours' rows)
#pragma unroll
                for (int b = 0; b < GT_F4; ++b) pre[b] = src[GT_LPC * b];
            }
            __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0) HERE: the wait below then only has the prefetched case to cover, with its twelve stores counted
        }
        float4* dst = reinterpret_cast<float4*>(&yl[g][0]) + r;
#pragma unroll
        for (int b = 0; b < GT_F4; ++b) dst[GT_LPC * b] = pre[b];
        if (t + 2 * TICK <= segT) {
            if (valid) {
#pragma unroll
                for (int b = 0; b < GT_F4; ++b) pre[b] = src[TICK / 4 + GT_LPC * b];
            }
            pre_t = t + TICK;
        }
    };
    auto four = [&](const float4 v, float& m2) -> float4 {
        float4 o;
        o.x = iir_advance_pk(fabsf(v.x), h0, m2); h2 = h1; h1 = h0; h0 = o.x;
        o.y = iir_advance_pk(fabsf(v.y), h0, m2); h2 = h1; h1 = h0; h0 = o.y;
        o.z = iir_advance_pk(fabsf(v.z), h0, m2); h2 = h1; h1 = h0; h0 = o.z;
        o.w = iir_advance_pk(fabsf(v.w), h0, m2); h2 = h1; h1 = h0; h0 = o.w;
        return o;
    };
    auto iir_tick = [&]() {   // (m17_frontend_kernels.hpp: one asm statement, three instructions per sample)
        iir_tick_in_place((uint32_t)(uintptr_t)as_lds(&yl[g][0]), h0, h1, h2);
    };
    // The tick's history values, LDS -> hbuf: twelve 16-byte stores per lane that are ALWAYS issued — a lane with nothing to store (an idle
    // channel, a pass that keeps no history) points beyond the end of the buffer descriptor and its store is dropped.  Behind a branch the
    // compiler cannot count them, and the next tick's wait for its prefetched samples (one counter for loads and stores, in order) became a
    // wait for these stores: a store round trip, ~1.5 us, per tick of ~4 (NOTES 5.9).
    typedef int gt_v4i __attribute__((ext_vector_type(4)));
    const __amdgpu_buffer_rsrc_t hrsrc = __builtin_amdgcn_make_buffer_rsrc((void*)(P.h + (size_t)(blockIdx.x * GT_CPW) * P.ypitch), 0,
                                                                            (int)min((size_t)GT_CPW * P.ypitch * 4u, (size_t)0x7FFF0000u), 0x00020000);
    const uint32_t hvoff = (valid && store) ? (uint32_t)(((size_t)g * P.ypitch + YPRE + t0) * 4u + (size_t)r * 16u) : 0x7FFF0000u;
    auto store_tick = [&](bool fed) {   // (`fed`: this lane's channel was fed in this tick)
        const gt_v4i* i4 = reinterpret_cast<const gt_v4i*>(&hl[g][0]) + r;
        const uint32_t voff = fed ? hvoff : 0x7FFF0000u;
#pragma unroll
        for (int b = 0; b < GT_F4; ++b) {
            __builtin_amdgcn_raw_buffer_store_b128(i4[GT_LPC * b], hrsrc, (int)(voff + t * 4u + 64u * (uint32_t)b), 0, 0);
            if (b % 3 == 2) __builtin_amdgcn_sched_barrier(0);   // three 16-byte rows in flight at a time (not all twelve: registers)
        }
    };
    while (t < segT) {
        // ---- steady state: every channel of the wave is inside a gated run, past its first 148 samples — nothing but the filter, and,
        //      where the tick ends on a carrier-on update point (every fifth one), that point (tail of operator() :742-752) ------------
        {
            const bool steady = !valid || (init <= 0 && on != 0 && run_pos >= 148 && count + TICK <= 960u);
            if (phase == 0 && t + TICK <= segT && ((pos0 + t) & 3u) == 0 && __ballot(!steady) == 0ull) {
                const bool upd = valid && count + TICK == 960u;
                float l1 = 0.f, l2 = 0.f;
                if (upd) {   // the sums the point will want: in flight during the recurrence
                    const uint64_t krow = min(k_cur - P.tick_row0, (uint64_t)P.ticks_cap - 1);
                    const float* row = tab + (size_t)krow * 12;
                    const int jsum = (uint32_t)(k_cur + 1 - seg) > 5 ? 5 : (int)(seg % 5u);
                    l1 = row[jsum]; l2 = row[6 + jsum];
                }
                stage_tick();
                lds_sync();
                if (r == 0) iir_tick();
                lds_sync();
                store_tick(true);
                lds_sync();
                count += TICK;
                if (upd) {
                    if (!trig) { on = 0; end_in_run = true; end_t = (int32_t)(t + TICK - 1u); }
                    count = 0;
                    dcd_update(k_cur, l1, l2);
                }
                t += TICK;
                ++k_cur;
                continue;
            }
        }
        const uint32_t n = min(TICK - phase, segT - t);   // a piece never crosses a tick boundary
        const bool feed = init > 0 || on;
        // ---- first 148 outputs of a gated run: FIR over (snapshot of the previous run's tail | this run's samples) ------------
        const bool need_patch = feed && run_pos < 148 && !pl_valid;
        const unsigned long long pm = __ballot(need_patch);
        if (pm) {
            for (int gg = 0; gg < GT_CPW; ++gg) {
                if (!((pm >> gg) & 1ull)) continue;   // wave-uniform
                const int src = gg;
                const uint32_t cc = (uint32_t)__shfl((int)c, src);
                const int32_t rp = __shfl(run_pos, src);
                const int32_t rs = (int32_t)t - rp;           // relative index of the run's first sample (>= -148)
                const bool eir = __shfl((int)end_in_run, src);
                const int32_t et = __shfl(end_t, src);
                const int16_t* xrc = P.x + (size_t)cc * P.xpitch + XPRE + t0;
                const Boundary* bsrc = (!from_chain && bnd_base) ? bnd_base + cc : nullptr;   // (both wave-uniform)
                const int16_t* hist = bsrc ? bsrc->hist : P.state[cc].hist;
                for (int k = lane; k < 149; k += 64) {
                    const int sv = eir ? (int)xrc[(int64_t)et - 148 + k] : (int)hist[k];
                    pw[k] = scale_sample(sv, invert);
                }
                for (int k = lane; k < 148; k += 64)
                    if ((int64_t)rs + k < (int64_t)segT) pw[149 + k] = scale_sample((int)xrc[(int64_t)rs + k], invert);
                lds_sync();
                for (int j = rp + lane; j < 148; j += 64) {
                    if ((int64_t)rs + j >= (int64_t)segT) break;
                    float acc = 0.f;
                    for (int i = 0; i < NTAPS; ++i) {          // FirFilter.h:36-40: newest sample first
                        const float p = pw[149 + j - i] * P.taps[i];
                        acc = acc + p;
                    }
                    pl[gg][j] = acc;
                }
                lds_sync();
            }
            if (need_patch) { pl_valid = true; pl_rs = (int32_t)t - run_pos; }
        }
        // ---- the DCD sums this piece's update point (if it ends on one) will need: in flight during the recurrence ---------------
        const uint32_t te = t + n - 1u;                        // last sample of the piece
        const uint64_t k = k_cur;                              // tick that ends with it (if it ends a tick)
        // (loaded unconditionally — a branch here would make the compiler wait for the loads at the join)
        const uint64_t krow = min(k - P.tick_row0, (uint64_t)P.ticks_cap - 1);
        const float* row = tab + (size_t)krow * 12;
        const int jsum = (uint32_t)(k + 1 - seg) > 5 ? 5 : (int)(seg % 5u);
        const float l1 = row[jsum], l2 = row[6 + jsum];
        // ---- feed the piece -----------------------------------------------------------------------------------------------
        if (__ballot(feed)) {
            // whole aligned ticks without patched outputs: 16-byte loads, issued one tick ahead
            const bool overlay = feed && pl_valid && (int32_t)t - pl_rs < 148;
            const bool fast = n == TICK && ((pos0 + t) & 3u) == 0 && !__ballot(overlay);
            if (fast) {
                stage_tick();
            } else {
#pragma clang loop unroll(disable) vectorize(disable)
                for (uint32_t i = r; i < n; i += GT_LPC) {   // (a rare path: kept rolled so that it does not set the kernel's register count)
                    float v = 0.f;
                    if (feed) {
                        const int32_t j = (int32_t)(t + i) - pl_rs;   // position inside the run, meaningful while pl_valid
                        v = (pl_valid && j < 148) ? pl[g][j] : yr[t + i];
                    }
                    yl[g][i] = v;
                }
            }
            lds_sync();
            if (r == 0 && feed) {
                if (n == TICK) {
                    iir_tick();
                } else {
                    float m2 = IirCoef::a2 * h1;
                    if ((n & 3u) == 0) {
#pragma clang loop unroll(disable)
                        for (uint32_t i = 0; i < n; i += 4) *reinterpret_cast<float4*>(&hl[g][i]) = four(*reinterpret_cast<const float4*>(&yl[g][i]), m2);
                    } else {
#pragma clang loop unroll(disable)
                        for (uint32_t i = 0; i < n; ++i) {
                            const float hn = iir_advance_pk(fabsf(yl[g][i]), h0, m2);
                            h2 = h1; h1 = h0; h0 = hn;
                            hl[g][i] = hn;
                        }
                    }
                }
            }
            lds_sync();
            if (feed && valid && store) {
                if (fast) {
                    store_tick(true);
                } else {
#pragma clang loop unroll(disable) vectorize(disable)
                    for (uint32_t i = r; i < n; i += GT_LPC) hr[t + i] = hl[g][i];
                }
            }
            lds_sync();
            if (feed) run_pos = min(148, run_pos + (int32_t)n);
        }
        // ---- the gate: M17Demodulator::operator() :670-689 (carrier off) and :742-752 (carrier on) ----------------------------
        if (init > 0) {
            init -= (int32_t)n;
            count = 0;
            if (init == 0) { end_in_run = true; end_t = (int32_t)te; }   // the initialisation run ends; the carrier is off
        } else if (!on) {
            count += n;
            if (count == 384u) {
                if (trig) {   // update_dcd -> dcd_on: a new gated run starts with the next sample
                    on = 1;
                    run_pos = 0;
                    pl_valid = false;
                    if (r == 0 && valid && store) {   // the history the run inherits, where its first samples will look for it
                        hr[(int64_t)te] = h0; hr[(int64_t)te - 1] = h1; hr[(int64_t)te - 2] = h2;
                    }
                }
                dcd_update(k, l1, l2);
                count = 0;
            }
        } else {
            count += n;
            if (count == 960u) {
                if (!trig) { on = 0; end_in_run = true; end_t = (int32_t)te; }
                count = 0;
                dcd_update(k, l1, l2);
            }
        }
        t += n;
        if (phase + n == TICK) { phase = 0; ++k_cur; } else phase += n;
        // (this path's loads and stores sit behind branches: nothing of it stays in flight into a steady tick, whose waits are counted)
        __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0)
    }
    if (r == 0 && valid) {
        float* f = P.final_h + fh_off + (size_t)c * 4;
        f[0] = h0; f[1] = h1; f[2] = h2;
        GateExport e;
        e.init = init; e.on = on; e.trig = trig; e.count = count; e.run_pos = run_pos;
        e.h0 = h0; e.h1 = h1; e.h2 = h2; e.level = level; e.seg = seg;
        e.end_in_run = end_in_run ? 1 : 0; e.end_t = end_t - (int32_t)segT;
        P.chain_out[c] = e;
    }
}

// (at most 128 VGPRs: a wave of it has to fit into what four K5 waves leave of a SIMD)
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(4, 4))) void limit_track_kernel(GateParams P)
{
    extern __shared__ __attribute__((aligned(16))) float lds_dyn[];
    if (blockIdx.x >= P.nblk) {   // (a launch that runs beside K5 takes an earlier segment's EVM operations along: 64 channels per block)
        evm_fold_pass(P.ev, blockIdx.x - P.nblk, lds_dyn);
        return;
    }
    __builtin_amdgcn_s_setprio(3);  // K5 of the next segment waits for this kernel: issue ahead of whatever shares the SIMD
    const int lane = threadIdx.x;
    uint32_t c = blockIdx.x * GT_CPW + lane % GT_CPW;
    bool valid = c < P.C;
    if (!valid) c = P.C - 1;  // (a real row for its addresses; the lane idles)
    if (P.only) {
        valid = valid && P.only[c] != 0;
        if (!__ballot(valid)) return;   // nothing to redo for these sixteen channels
    }
    // (dynamic LDS, GT_LDS_FLOATS floats: with a static allocation the compiler sizes the register budget for the LDS-limited occupancy and
    //  ignores the waves-per-SIMD attribute above)
    limit_track_pass(P, (P.flags & 2u) != 0, c, valid, P.chain_in != nullptr, P.only ? P.bnd : nullptr, lds_dyn);
}

// =====================================================================================================
// Gate-aware front end (m17hip_tune key 26): which samples of segment k + 2 can the carrier be ON for?
// The reference runs neither the matched filter nor the correlator while its carrier detect is off (M17Demodulator.h:675-689).  The
// detector's lower threshold is 0.1 and noise gives a bin ratio of about 1, so once triggered it never falls off by itself: every off
// transition is a forced dcd.unlock() of the state machine, which only K5 knows.  But a channel that IS off stays off until an update
// point finds level > 4.0 — a function of the table and of the off state alone (no state machine runs while the gate is closed).  So:
// K5 leaves the TRUE gate state at the end of segment k (GateTruth); one lane per channel walks the update points of segments k + 1
// and k + 2 from it (the arithmetic of limit_track_pass's carrier-off branch = M17Demodulator.h:675-689, DataCarrierDetect.h:63-69)
// and writes the first sample of segment k + 2 for which the gate can be open: 0 for a channel that was on or initialising (nothing
// is known about when it closes), 0xFFFFFFFF if it cannot open.  K1 of segment k + 2 skips the tiles that end before that sample —
// K5 patches the first 148 outputs of a gated run itself and K2's replay (which may be on a trajectory of its own after a forced
// unlock: such channels are off the replay and redone from K5's boundary record) finds stale but finite values there.
// =====================================================================================================
__global__ __launch_bounds__(64) void gate_forecast_kernel(const GateTruth* __restrict__ truth, const float* __restrict__ dcd_table, uint32_t ticks_cap, uint64_t tick_row0,
                                                           uint64_t pos_start, uint32_t target_begin, uint32_t horizon, uint32_t* __restrict__ first_needed, uint32_t C)
{
    const uint32_t c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const GateTruth t = truth[c];
    if (t.init > 0 || t.on) { first_needed[c] = 0u; return; }
    const float* tab = dcd_table + (size_t)c * ticks_cap * 12;
    float level = t.level;
    uint32_t trig = t.trig, seg = t.seg, count = t.count;
    uint64_t pos = pos_start;                       // absolute index of the next sample to be counted
    uint32_t out = 0xFFFFFFFFu;
    while (true) {
        const uint64_t te = pos + (384u - count) - 1u;   // the sample that makes count == 384: an update point (always the last sample of a tick)
        if (te >= pos_start + horizon) break;
        if (trig) {                                  // update_dcd -> dcd_on: the gated run starts with the next sample
            const uint64_t on_at = te + 1u;
            out = on_at <= pos_start + target_begin ? 0u : (uint32_t)(on_at - (pos_start + target_begin));
            break;
        }
        const uint64_t k = te / TICK;
        const uint64_t krow = min(k - tick_row0, (uint64_t)ticks_cap - 1);
        const float* row = tab + (size_t)krow * 12;
        const int jsum = (uint32_t)(k + 1 - seg) > 5 ? 5 : (int)(seg % 5u);
        const float l1 = row[jsum], l2 = row[6 + jsum];
        level = (float)((double)level * 0.8 + 0.2 * (double)(l1 / l2));
        seg = (uint32_t)(k + 1);
        trig = level > 4.0f;                         // (the trigger was clear: the upper threshold applies)
        count = 0;
        pos = te + 1u;
    }
    first_needed[c] = out;
}

}  // namespace m17
