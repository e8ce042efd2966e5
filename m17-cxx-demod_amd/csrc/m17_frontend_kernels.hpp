// K1 (RRC-150 matched filter), K2 (correlator limit + sync-word correlations), K3 (sliding-DFT carrier
// detect accumulation).  gfx950 / CDNA4, wave64.  See DESIGN.md §3 for the roofline of each kernel.
#pragma once

#include "m17_common.hpp"

namespace m17 {

// =====================================================================================================
// K1  fir_rrc150_kernel  — reference a1 + a2: apps/m17-demod.cpp:486-489, FirFilter.h:28-43.
//
// y[t] = sum_{i=0}^{148} taps[i] * x[t-i], accumulated sequentially i = 0..148 in fp32, separate mul and
// add (bit-exact with the reference's loop; tap 149 is 0.0 and cannot change the value).  Time-parallel:
// one workgroup = one channel x FIR_TILE consecutive outputs.  The int16 window (tile + 148 history
// samples, taken from the XPRE prefix at the start of a run) is loaded with coalesced 8-byte loads,
// scaled to float once and staged in LDS; each lane then produces FIR_R consecutive outputs from a
// register window that slides by one LDS word per tap (1 ds_read_b32 : 2*FIR_R VALU).  FIR_R is odd so the
// per-lane LDS stride is conflict-free.  Taps are compile-time literals.  Outputs go back through LDS so
// the HBM stores are contiguous float4.
//   VALU bound: 298 flop/sample (no FMA allowed) -> 2.6e11 samples/s at 78.6 Tflop/s non-FMA fp32.
//   HBM: 2 B in + 4 B out per sample.
// =====================================================================================================
constexpr int FIR_R = 15;
constexpr int FIR_THREADS = 256;
constexpr int FIR_TILE = FIR_R * FIR_THREADS;  // 3840 outputs per workgroup (480000 = 125 tiles)
constexpr int FIR_WIN = FIR_TILE + NTAPS - 1;  // 3988 staged samples

__global__ __launch_bounds__(FIR_THREADS, 3) void fir_rrc150_kernel(const int16_t* __restrict__ x, size_t xpitch,
                                                                float* __restrict__ y, size_t ypitch, uint32_t T,
                                                                uint32_t flags)
{
    __shared__ __attribute__((aligned(16))) float win[FIR_WIN + 4];
    const int tid = threadIdx.x;
    const uint32_t c = blockIdx.y;
    const uint32_t t0 = blockIdx.x * FIR_TILE;
    const bool invert = flags & 1u;
    const int16_t* xr = x + (size_t)c * xpitch + XPRE;  // xr[t], t >= -XPRE
    float* yr = y + (size_t)c * ypitch + YPRE;

    // stage [t0 - 148, t0 + FIR_TILE): (XPRE + t0 - 148) is a multiple of 4 samples -> aligned 8-byte loads
    const int64_t w0 = (int64_t)t0 - (NTAPS - 1);
    for (int k = tid; k < (FIR_WIN + 3) / 4; k += FIR_THREADS) {
        const int64_t t = w0 + 4 * k;
        short4 v = make_short4(0, 0, 0, 0);
        if (t < (int64_t)T) v = *reinterpret_cast<const short4*>(xr + t);  // rows are padded to a multiple of 8 past T
        float4 f;
        f.x = scale_sample(v.x, invert);
        f.y = scale_sample(v.y, invert);
        f.z = scale_sample(v.z, invert);
        f.w = scale_sample(v.w, invert);
        *reinterpret_cast<float4*>(&win[4 * k]) = f;
    }
    __syncthreads();

    // win[j] <-> sample t0 - 148 + j.  Output o = tid*R + r at tap i reads win[148 + o - i].
    const float* base = win + tid * FIR_R;
    float w[FIR_R], acc[FIR_R];
#pragma unroll
    for (int r = 0; r < FIR_R; ++r) w[r] = base[(NTAPS - 1) + r];
#pragma unroll
    for (int i = 0; i < NTAPS; ++i) {
        const float tap = rrc_tap(i);
#pragma unroll
        for (int r = 0; r < FIR_R; ++r) {
            const float p = w[r] * tap;
            acc[r] = (i == 0 ? 0.0f : acc[r]) + p;
        }
        if (i < NTAPS - 1) {
#pragma unroll
            for (int r = FIR_R - 1; r > 0; --r) w[r] = w[r - 1];
            w[0] = base[(NTAPS - 1) - (i + 1)];
        }
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < FIR_R; ++r) win[tid * FIR_R + r] = acc[r];
    __syncthreads();
    for (int k = tid; k < FIR_TILE / 4; k += FIR_THREADS) {
        const uint32_t t = t0 + 4 * k;
        if (t + 3 < T) {
            *reinterpret_cast<float4*>(yr + t) = *reinterpret_cast<const float4*>(&win[4 * k]);
        } else {
            for (int q = 0; q < 4; ++q)
                if (t + q < T) yr[t + q] = win[4 * k + q];
        }
    }
}

// =====================================================================================================
// K2a  correlate_kernel — reference a4: Correlator::correlate (Correlator.h:51-64) against the four M17
// sync words for every sample: corr[w][c][t] = sum_{i=0}^{7} word[i] * y[t - 70 + 10 i] (oldest symbol
// first, fp32, mul then add).  Time-parallel, elementwise; the 8 taps are shared by the four words.
// =====================================================================================================
__global__ __launch_bounds__(256) void correlate_kernel(const float* __restrict__ y, size_t ypitch, float* __restrict__ corr,
                                                        uint32_t C, uint32_t T)
{
    const uint32_t c = blockIdx.y;
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= T) return;
    const float* yr = y + (size_t)c * ypitch + YPRE;
    float s[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) s[i] = yr[(int64_t)t - 70 + 10 * i];  // t - 70 >= -YPRE always
#pragma unroll
    for (int w = 0; w < 4; ++w) {
        float r = 0.0f;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const float p = (float)SYNC_WORDS[w][i] * s[i];
            r = r + p;
        }
        corr[((size_t)w * C + c) * T + t] = r;
    }
}

// =====================================================================================================
// K2b  limit_kernel — reference a3: Correlator::sample's limit_ = BaseIirFilter<float,3>(|y|)
// (Correlator.h:43-45, IirFilter.h:26-42, coefficients Correlator.h:38-39).  A float recurrence: strictly
// sequential per channel, one lane per channel; each lane streams its own row with 16-byte accesses.
// =====================================================================================================
struct IirCoef {
    static constexpr float b0 = 4.24433681e-05f, b1 = 8.48867363e-05f, b2 = 4.24433681e-05f;
    static constexpr float a1 = -1.98148851f, a2 = 0.98165828f;
};
// one step: returns h0; caller rotates (h2 <- h1, h1 <- h0)
__device__ __forceinline__ float iir_advance(float in_abs, float h1, float h2)
{
    float h0 = in_abs;
    h0 = h0 - IirCoef::a1 * h1;
    h0 = h0 - IirCoef::a2 * h2;
    return h0;
}
// The same recurrence over a run of samples with the two products of a sample formed by ONE packed multiply:
// (a1 * h, a2 * h) for the newest history value h gives a1*h1 for the next sample and a2*h2 for the one after
// (identical IEEE products, one VALU instruction less per sample).  m2 carries a2 * h2 between calls.
typedef float iir_v2f __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float iir_advance_pk(float in_abs, float h1, float& m2)
{
    const iir_v2f pr = iir_v2f{h1, h1} * iir_v2f{IirCoef::a1, IirCoef::a2};
    float h0 = in_abs - pr.x;
    h0 = h0 - m2;
    m2 = pr.y;
    return h0;
}
__device__ __forceinline__ float iir_output(float h0, float h1, float h2)
{
    float r = 0.0f;
    r = r + IirCoef::b0 * h0;
    r = r + IirCoef::b1 * h1;
    r = r + IirCoef::b2 * h2;
    return r;
}

__global__ __launch_bounds__(64) void limit_kernel(const float* __restrict__ y, size_t ypitch, float* __restrict__ limit,
                                                   uint32_t C, uint32_t T)
{
    const uint32_t c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const float* yr = y + (size_t)c * ypitch + YPRE;
    float* lr = limit + (size_t)c * T;
    float h1 = 0.f, h2 = 0.f;
    uint32_t t = 0;
    const bool aligned = (((size_t)c * T) & 3) == 0;
    if (aligned) {
        for (; t + 4 <= T; t += 4) {
            const float4 v = *reinterpret_cast<const float4*>(yr + t);
            float4 o;
            float h0;
            h0 = iir_advance(fabsf(v.x), h1, h2); o.x = iir_output(h0, h1, h2); h2 = h1; h1 = h0;
            h0 = iir_advance(fabsf(v.y), h1, h2); o.y = iir_output(h0, h1, h2); h2 = h1; h1 = h0;
            h0 = iir_advance(fabsf(v.z), h1, h2); o.z = iir_output(h0, h1, h2); h2 = h1; h1 = h0;
            h0 = iir_advance(fabsf(v.w), h1, h2); o.w = iir_output(h0, h1, h2); h2 = h1; h1 = h0;
            *reinterpret_cast<float4*>(lr + t) = o;
        }
    }
    for (; t < T; ++t) {
        const float h0 = iir_advance(fabsf(yr[t]), h1, h2);
        lr[t] = iir_output(h0, h1, h2);
        h2 = h1; h1 = h0;
    }
}

// =====================================================================================================
// K3  dcd_kernel — reference a7 + the accumulation half of a8:
// NSlidingDFT<float,48000,120,2>::operator() (SlidingDFT.h:118-132) and
// DataCarrierDetect::operator() (DataCarrierDetect.h:53-58).
//
//   delta = x[n] - x[n-120];  X_k = (X_k + delta) * c_k  (libstdc++ complex multiply: ac-bd, ad+bc);
//   L1 += |X_0|^2; L2 += |X_1|^2.
// The DFT sees EVERY sample and depends on nothing else, so it runs ahead of the state machine as its own
// pass.  What the state machine needs is the value of (L1, L2) at DCD update points, accumulated
// sequentially from the previous update point; update points are 384 (carrier off) or 960 (carrier on)
// samples apart and always fall on 192-sample tick boundaries (M17Demodulator.h:677-686,742-751), the
// first one 2304 samples after the stream start.  So this pass keeps six running sums per bin — five that are
// reset at the start of ticks a = 0,1,2,3,4 (mod 5) and one that runs from the stream start — and writes
// them at the end of every tick: the state machine later picks the sum whose start matches its segment,
// bit-exact with the reference's single accumulator whatever the cadence turned out to be.
//
// A float recurrence is sequential in time and a lone wave pays for every instruction it issues (~3 ns), so the kernel
// is built around the fewest instructions per sample on the recurrence:
//   * 16 LANES PER CHANNEL, 4 channels per wave.  Lane role (bin, j): the lane carries DFT bin `bin` (redundantly with
//     the 7 other lanes of that bin — a VALU instruction costs the same for 1 or 64 lanes) and ONE of the six running
//     sums, so a sample costs one accumulate instead of six.
//   * the complex recurrence on packed fp32 (v_pk_mul_f32 / v_pk_add_f32: IEEE mul/add on two floats per instruction,
//     no contraction): t = Xr + delta; (ac, ad) = (t,t)*(cr,ci); (-bd, bc) = (Xi,Xi)*(-ci,cr); X = (ac + -bd, ad + bc);
//     (p, q) = X*X; sum += p + q  — 7 VALU instructions per sample.
//   * the time-parallel part (int16 -> float scaling of x[n] and x[n-120], delta) is done by the 16 lanes for a whole
//     192-sample tick at once (8-byte loads, issued two ticks ahead of their use) and handed to the recurrence through
//     LDS.  x[n-120] comes from the carried prefix of xbuf (XPRE >= 120), so there is no delay line to maintain.
//   * the recurrence runs as straight-line 64-sample blocks in a pinned, software-pipelined issue order.
// Measured (tools/k3bench.hip): a lone wave issues one VALU instruction per ~2.6 ns whether dependent or not, so the
// time is instructions x 2.6 ns: 7 per sample on the recurrence + ~1 for conversion = 9.6 ms per 480 000 samples.
// Sum 5 runs from the stream start (it is read at the first update point, sample 2303, only).
// Table layout: [C][ticks][2 bins][6 sums].  Algorithmic bytes: 2 B/sample read (+ 48 B per 192 samples written).
// =====================================================================================================
struct DcdCoef { float c0r, c0i, c1r, c1i; };  // exp(-j 2 pi f/48000), f = 2400, 3600 — computed on the host

// apps/m17-demod.cpp:486-489 through the double-precision product (bit-identical to the division for all int16, see
// tests/test_oracle_kat.py::test_scale_identities_exhaustive); 3 instructions instead of a division expansion.
__device__ __forceinline__ float scale_sample_mul(int s, bool invert)
{
    if (invert) s = (int)(int16_t)(-s);
    return (float)((double)s * (1.0 / 41067.0));
}

typedef float v2f __attribute__((ext_vector_type(2)));
constexpr int DCD_BLK = 64;  // samples per straight-line block of the recurrence = 16 lanes x 4
constexpr int DCD_CPW = 4;   // channels per wave

struct DcdLane {  // one (bin, sum) role of one channel
    v2f X, cc, cs;  // DFT state (re, im); (cr, ci); (-ci, cr)
    float acc;
};
__device__ __forceinline__ void dcd_step(DcdLane& s, float delta)
{
    const float a = s.X.x + delta;
    const v2f m1 = v2f{a, a} * s.cc;          // (ac, ad)
    const v2f m2 = v2f{s.X.y, s.X.y} * s.cs;  // (-bd, bc): negating a factor negates the product exactly
    s.X = m1 + m2;                            // libstdc++ complex multiply: (ac - bd, ad + bc)
    const v2f p = s.X * s.X;
    const float nrm = p.x + p.y;
    s.acc = s.acc + nrm;
}
// N steps as straight-line code in a pinned issue order.  A wave alone on its SIMD waits ~8 ns for the result of the
// instruction it has just issued but can issue an independent one every ~3 ns, so the three-deep recurrence
// (add -> mul -> add) is interleaved with the norm / accumulate work of the two previous samples: same operations on
// the same values in the same order per variable, only the instruction order differs from dcd_step().
#define M17_PIN() __builtin_amdgcn_sched_barrier(0)
template <int N>
__device__ __forceinline__ void dcd_steps_pipelined(DcdLane& s, const float (&d)[N])
{
    v2f P = {0.f, 0.f};   // X*X of the previous sample, norm not yet formed
    float nrm = 0.f;      // norm of the sample before that, not yet accumulated
#pragma unroll
    for (int n = 0; n < N; ++n) {
        const float a = s.X.x + d[n];                M17_PIN();
        v2f Pn = P;
        if (n >= 1) { Pn = s.X * s.X;                M17_PIN(); }
        const v2f m2 = v2f{s.X.y, s.X.y} * s.cs;     M17_PIN();
        const v2f m1 = v2f{a, a} * s.cc;             M17_PIN();
        if (n >= 3) { s.acc = s.acc + nrm;           M17_PIN(); }
        if (n >= 2) { nrm = P.x + P.y;               M17_PIN(); }
        s.X = m1 + m2;                               M17_PIN();
        P = Pn;
    }
    // drain: norms of the last samples
    if (N >= 3) { s.acc = s.acc + nrm; M17_PIN(); }
    if (N >= 2) { nrm = P.x + P.y; M17_PIN(); }
    const v2f Pl = s.X * s.X; M17_PIN();
    if (N >= 2) { s.acc = s.acc + nrm; M17_PIN(); }
    nrm = Pl.x + Pl.y; M17_PIN();
    s.acc = s.acc + nrm; M17_PIN();
}

// pos0: absolute index (since reset) of the first sample of this run — identical for every channel.
constexpr int DCD_PF = 2;  // whole ticks of input in flight ahead of the recurrence (a tick is ~3 us of recurrence)
__global__ __launch_bounds__(64) void dcd_kernel(const int16_t* __restrict__ x, size_t xpitch, DcdState* __restrict__ state,
                                                 float* __restrict__ table, uint32_t ticks_cap, uint32_t C, uint32_t T,
                                                 uint64_t pos0, DcdCoef k, uint32_t flags)
{
    __shared__ __attribute__((aligned(16))) float dl[DCD_CPW][TICK];
    // (no s_setprio: later segments of this kernel have slack, the kernels it shares SIMDs with do not)
    const int lane = threadIdx.x;
    const int g = lane >> 4, r = lane & 15, bin = r >> 3, j = r & 7;
    uint32_t c = blockIdx.x * DCD_CPW + g;
    const bool owner = c < C;
    const bool live = owner && j < 6;  // lanes that own a sum of an existing channel (the others shadow and never store)
    if (c >= C) c = C - 1;
    const bool invert = flags & 1u;
    const int16_t* xr = x + (size_t)c * xpitch + XPRE;
    DcdState* st = state + c;
    DcdLane s;
    s.X = v2f{st->xr[bin], st->xi[bin]};
    s.cc = bin ? v2f{k.c1r, k.c1i} : v2f{k.c0r, k.c0i};
    s.cs = v2f{-s.cc.y, s.cc.x};
    s.acc = st->acc[j < 6 ? j : 0][bin];
    float* tab = table + (size_t)c * ticks_cap * 12 + bin * 6 + j;
    float* mydl = dl[g];
    uint32_t phase = (uint32_t)(pos0 % TICK);  // position inside the current tick (wave-uniform)
    uint64_t tick = pos0 / TICK;
    uint32_t row = 0;
    auto lds_sync = [] {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront", "local");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront", "local");
    };
    auto tick_begin = [&] { if ((uint32_t)j == (uint32_t)(tick % 5)) s.acc = 0.f; };  // the sum that restarts with this tick
    auto tick_end = [&] {
        if (live) tab[(size_t)row * 12] = s.acc;
        phase = 0; ++tick; ++row;
    };
    // generic path (head / tail of a run, unaligned runs): up to 64 samples at a time, never across a tick boundary
    auto slow_block = [&](uint32_t t0, uint32_t n) {
        float4 dv;
        float* d4 = &dv.x;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const uint32_t i = 4u * r + q;
            const bool in = i < n;
            const int a = in ? (int)xr[(int64_t)t0 + i] : 0, b = in ? (int)xr[(int64_t)t0 + i - 120] : 0;
            d4[q] = scale_sample_mul(a, invert) - scale_sample_mul(b, invert);
        }
        *reinterpret_cast<float4*>(mydl + 4 * r) = dv;
        lds_sync();
        if (phase == 0) tick_begin();
        for (uint32_t i = 0; i < n; ++i) dcd_step(s, mydl[i]);
        phase += n;
        if (phase == TICK) tick_end();
        lds_sync();
    };

    uint32_t t = 0;
    while (t < T && phase != 0) {  // head: up to the next tick boundary
        const uint32_t n = min(min(64u, TICK - phase), T - t);
        slow_block(t, n);
        t += n;
    }
    // whole ticks: lane r of a channel converts samples 4r..4r+3 of each of the tick's three 64-sample blocks (8-byte loads)
    if (((pos0 + t) & 3u) == 0 && (xpitch & 3u) == 0 && t + TICK <= T) {
        int2 pa[DCD_PF][3], pb[DCD_PF][3];
        auto issue = [&](int slot, uint32_t t0) {
#pragma unroll
            for (int b = 0; b < 3; ++b) {
                const int16_t* p = xr + (size_t)t0 + 64 * b + 4 * r;
                pa[slot][b] = *reinterpret_cast<const int2*>(p);
                pb[slot][b] = *reinterpret_cast<const int2*>(p - 120);
            }
        };
        const uint32_t nt = (T - t) / TICK;
#pragma unroll
        for (int q = 0; q < DCD_PF; ++q)
            if ((uint32_t)q < nt) issue(q, t + q * TICK);
        auto conv = [&](int v) { return scale_sample_mul(v, invert); };
        auto lo = [](int w) { return (int)(int16_t)(w & 0xFFFF); };
        auto hi = [](int w) { return w >> 16; };
        for (uint32_t it = 0; it < nt; it += DCD_PF) {
#pragma unroll
            for (int q = 0; q < DCD_PF; ++q) {
                if (it + q < nt) {  // wave-uniform
#pragma unroll
                    for (int b = 0; b < 3; ++b) {
                        const int2 a = pa[q][b], d = pb[q][b];
                        float4 dv;
                        dv.x = conv(lo(a.x)) - conv(lo(d.x));
                        dv.y = conv(hi(a.x)) - conv(hi(d.x));
                        dv.z = conv(lo(a.y)) - conv(lo(d.y));
                        dv.w = conv(hi(a.y)) - conv(hi(d.y));
                        *reinterpret_cast<float4*>(mydl + 64 * b + 4 * r) = dv;
                    }
                    lds_sync();
                    if (it + q + DCD_PF < nt) issue(q, t + DCD_PF * TICK);  // this slot's next tick: in flight for DCD_PF ticks
                    tick_begin();
#pragma unroll 1
                    for (int b = 0; b < 3; ++b) {
                        float d[DCD_BLK];
#pragma unroll
                        for (int u = 0; u < DCD_BLK / 4; ++u) {
                            const float4 v = *reinterpret_cast<const float4*>(mydl + 64 * b + 4 * u);
                            d[4 * u] = v.x; d[4 * u + 1] = v.y; d[4 * u + 2] = v.z; d[4 * u + 3] = v.w;
                        }
                        dcd_steps_pipelined<DCD_BLK>(s, d);
                    }
                    tick_end();
                    lds_sync();
                    t += TICK;
                }
            }
        }
    }
    while (t < T) {  // tail (or a whole unaligned run)
        const uint32_t n = min(min(64u, TICK - phase), T - t);
        slow_block(t, n);
        t += n;
    }
    if (live) st->acc[j][bin] = s.acc;
    if (owner && j == 0) { st->xr[bin] = s.X.x; st->xi[bin] = s.X.y; }
}

}  // namespace m17
