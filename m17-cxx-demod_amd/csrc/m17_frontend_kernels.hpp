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

__global__ __launch_bounds__(FIR_THREADS) void fir_rrc150_kernel(const int16_t* __restrict__ x, size_t xpitch,
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
// A float recurrence is sequential in time, and a lone wave pays ~3 ns per dependent instruction, so the kernel
// minimises instructions per sample per lane and maximises ILP: TWO LANES PER CHANNEL (lane parity = DFT bin; the
// two bins and their sums are independent), samples converted once and kept in an LDS delay line (x[n-120] is a
// ds_read_b128 away), straight-line blocks of 8 samples, next block's 16-byte global load in flight while the
// current one is consumed.  Table layout: [C][ticks][2 bins][6 sums].
// Algorithmic bytes: 2 B/sample read (+ 48 B per 192 samples written).
// =====================================================================================================
struct DcdCoef { float c0r, c0i, c1r, c1i; };  // exp(-j 2 pi f/48000), f = 2400, 3600 — computed on the host

constexpr int DCD_RING = 128;        // delay line length (>= 120), power of two
constexpr int DCD_RING_PITCH = 132;  // floats per lane: 16-lane groups of ds_*_b128 fall on distinct banks

// apps/m17-demod.cpp:486-489 through the double-precision product (bit-identical to the division for all int16, see
// tests/test_oracle_kat.py::test_scale_identities_exhaustive); 3 instructions instead of a division expansion.
__device__ __forceinline__ float scale_sample_mul(int s, bool invert)
{
    if (invert) s = (int)(int16_t)(-s);
    return (float)((double)s * (1.0 / 41067.0));
}

struct DcdLane {  // one bin of one channel
    float xr, xi, cr, ci;
    float acc[6];
};
__device__ __forceinline__ void dcd_step(DcdLane& s, float xn, float xd, bool long_acc)
{
    const float delta = xn - xd;
    const float a = s.xr + delta, b = s.xi;
    const float ac = a * s.cr, bd = b * s.ci, ad = a * s.ci, bc = b * s.cr;
    s.xr = ac - bd;
    s.xi = ad + bc;
    const float nrm = s.xr * s.xr + s.xi * s.xi;
#pragma unroll
    for (int j = 0; j < 5; ++j) s.acc[j] = s.acc[j] + nrm;
    if (long_acc) s.acc[5] = s.acc[5] + nrm;
}

// pos0: absolute index (since reset) of the first sample of this run — identical for every channel.
__global__ __launch_bounds__(64) void dcd_kernel(const int16_t* __restrict__ x, size_t xpitch, DcdState* __restrict__ state,
                                                 float* __restrict__ table, uint32_t ticks_cap, uint32_t C, uint32_t T,
                                                 uint64_t pos0, DcdCoef k, uint32_t flags)
{
    __shared__ __attribute__((aligned(16))) float ringbuf[64 * DCD_RING_PITCH];
    const int lane = threadIdx.x;
    const uint32_t c = blockIdx.x * 32 + (lane >> 1);
    const int bin = lane & 1;
    if (c >= C) return;
    const bool invert = flags & 1u;
    const int16_t* xr = x + (size_t)c * xpitch + XPRE;
    float* ring = ringbuf + lane * DCD_RING_PITCH;
    DcdState* st = state + c;
    DcdLane s;
    s.xr = st->xr[bin]; s.xi = st->xi[bin];
    s.cr = bin ? k.c1r : k.c0r; s.ci = bin ? k.c1i : k.c0i;
#pragma unroll
    for (int q = 0; q < 6; ++q) s.acc[q] = st->acc[q][bin];
    // delay line: the 120 samples before this run (zeros after a reset), at ring slot = absolute index mod 128
    for (int d = 1; d <= 120; ++d) ring[(uint32_t)(pos0 - d) & (DCD_RING - 1)] = scale_sample_mul(xr[-d], invert);
    float* tab = table + (size_t)c * ticks_cap * 12 + bin * 6;
    uint32_t phase = (uint32_t)(pos0 % TICK);  // position inside the current tick (wave-uniform)
    uint64_t tick = pos0 / TICK;
    uint32_t row = 0;

    auto tick_begin = [&]() {  // start of tick `tick`: the sum that restarts here
        const int j = (int)(tick % 5);
#pragma unroll
        for (int q = 0; q < 5; ++q) s.acc[q] = (q == j) ? 0.f : s.acc[q];
    };
    auto tick_end = [&]() {
        float* o = tab + (size_t)row * 12;
#pragma unroll
        for (int q = 0; q < 6; ++q) o[q] = s.acc[q];
        phase = 0; ++tick; ++row;
    };
    auto one_sample = [&](uint32_t t) {  // generic path (unaligned head / tail of a run)
        if (phase == 0) tick_begin();
        const uint32_t slot = (uint32_t)(pos0 + t) & (DCD_RING - 1);
        const float xn = scale_sample_mul(xr[t], invert);
        const float xd = ring[(slot + 8) & (DCD_RING - 1)];
        ring[slot] = xn;
        dcd_step(s, xn, xd, pos0 + t < 12 * TICK);
        if (++phase == TICK) tick_end();
    };

    uint32_t t = 0;
    while (t < T && (((pos0 + t) & 7u) != 0)) { one_sample(t); ++t; }  // head: up to the next multiple of 8
    if (t + 8 <= T && (pos0 & 7u) == 0) {  // (a run that starts off an 8-sample boundary stays on the generic path)
        int4 nxt = *reinterpret_cast<const int4*>(xr + t);
        for (; t + 8 <= T; t += 8) {
            const int4 cur = nxt;
            if (t + 16 <= T) nxt = *reinterpret_cast<const int4*>(xr + t + 8);  // in flight while `cur` is consumed
            if (phase == 0) tick_begin();       // ticks are multiples of 8 samples: boundaries fall between blocks
            const uint32_t slot = (uint32_t)(pos0 + t) & (DCD_RING - 1);
            const float4 d0 = *reinterpret_cast<const float4*>(ring + ((slot + 8) & (DCD_RING - 1)));
            const float4 d1 = *reinterpret_cast<const float4*>(ring + ((slot + 12) & (DCD_RING - 1)));
            float4 n0, n1;
            n0.x = scale_sample_mul((int)(int16_t)(cur.x & 0xFFFF), invert); n0.y = scale_sample_mul(cur.x >> 16, invert);
            n0.z = scale_sample_mul((int)(int16_t)(cur.y & 0xFFFF), invert); n0.w = scale_sample_mul(cur.y >> 16, invert);
            n1.x = scale_sample_mul((int)(int16_t)(cur.z & 0xFFFF), invert); n1.y = scale_sample_mul(cur.z >> 16, invert);
            n1.z = scale_sample_mul((int)(int16_t)(cur.w & 0xFFFF), invert); n1.w = scale_sample_mul(cur.w >> 16, invert);
            *reinterpret_cast<float4*>(ring + slot) = n0;
            *reinterpret_cast<float4*>(ring + slot + 4) = n1;
            const bool la = pos0 + t < 12 * TICK;  // the stream-start sum is only ever read at the first update (sample 2303)
            dcd_step(s, n0.x, d0.x, la); dcd_step(s, n0.y, d0.y, la); dcd_step(s, n0.z, d0.z, la); dcd_step(s, n0.w, d0.w, la);
            dcd_step(s, n1.x, d1.x, la); dcd_step(s, n1.y, d1.y, la); dcd_step(s, n1.z, d1.z, la); dcd_step(s, n1.w, d1.w, la);
            phase += 8;
            if (phase == TICK) tick_end();
        }
    }
    for (; t < T; ++t) one_sample(t);  // tail
    st->xr[bin] = s.xr; st->xi[bin] = s.xi;
#pragma unroll
    for (int q = 0; q < 6; ++q) st->acc[q][bin] = s.acc[q];
}

}  // namespace m17
