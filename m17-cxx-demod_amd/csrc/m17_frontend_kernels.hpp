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
// first one 2304 samples after the stream start.  So this pass keeps six running pairs — five that are
// reset at the start of ticks a = 0,1,2,3,4 (mod 5) and one that runs from the stream start — and writes
// them at the end of every tick: the state machine later picks the pair whose start matches its segment,
// bit-exact with the reference's single accumulator whatever the cadence turned out to be.
//
// Float recurrence => sequential in time; one lane per channel.  Each lane streams its row with 16-byte
// loads (8 samples) for x[n] and x[n-120] (120 = 15 * 8 keeps both aligned; the delayed block is an L1/L2 hit).
// Algorithmic bytes: 2 B/sample read (+ 48 B per 192 samples written).
// =====================================================================================================
struct DcdCoef { float c0r, c0i, c1r, c1i; };  // exp(-j 2 pi f/48000), f = 2400, 3600 — computed on the host

__device__ __forceinline__ void dcd_step(DcdState& s, const DcdCoef& k, float xn, float xd, bool long_acc)
{
    const float delta = xn - xd;
    {
        const float a = s.xr[0] + delta, b = s.xi[0];
        const float ac = a * k.c0r, bd = b * k.c0i, ad = a * k.c0i, bc = b * k.c0r;
        s.xr[0] = ac - bd;
        s.xi[0] = ad + bc;
    }
    {
        const float a = s.xr[1] + delta, b = s.xi[1];
        const float ac = a * k.c1r, bd = b * k.c1i, ad = a * k.c1i, bc = b * k.c1r;
        s.xr[1] = ac - bd;
        s.xi[1] = ad + bc;
    }
    const float n0 = s.xr[0] * s.xr[0] + s.xi[0] * s.xi[0];
    const float n1 = s.xr[1] * s.xr[1] + s.xi[1] * s.xi[1];
#pragma unroll
    for (int j = 0; j < 5; ++j) {
        s.acc[j][0] = s.acc[j][0] + n0;
        s.acc[j][1] = s.acc[j][1] + n1;
    }
    if (long_acc) {
        s.acc[5][0] = s.acc[5][0] + n0;
        s.acc[5][1] = s.acc[5][1] + n1;
    }
}

// pos0: absolute index (since reset) of the first sample of this run — identical for every channel.
__global__ __launch_bounds__(64) void dcd_kernel(const int16_t* __restrict__ x, size_t xpitch, DcdState* __restrict__ state,
                                                 float* __restrict__ table, uint32_t ticks_cap, uint32_t C, uint32_t T,
                                                 uint64_t pos0, DcdCoef k, uint32_t flags)
{
    const uint32_t c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const bool invert = flags & 1u;
    const int16_t* xr = x + (size_t)c * xpitch + XPRE;
    DcdState s = state[c];
    float* tab = table + (size_t)c * ticks_cap * 12;
    uint32_t phase = (uint32_t)(pos0 % TICK);  // position inside the current tick (wave-uniform)
    uint64_t tick = pos0 / TICK;
    uint32_t row = 0;
    // the stream-start accumulator is only ever read at the first update (sample 2303)
    auto process = [&](float xn, float xd, uint64_t abs_pos) {
        if (phase == 0) {
            const int j = (int)(tick % 5);
#pragma unroll
            for (int q = 0; q < 5; ++q)
                if (q == j) { s.acc[q][0] = 0.f; s.acc[q][1] = 0.f; }
        }
        dcd_step(s, k, xn, xd, abs_pos < 12 * TICK);
        if (++phase == TICK) {
            float* o = tab + (size_t)row * 12;
#pragma unroll
            for (int q = 0; q < 6; ++q) { o[2 * q] = s.acc[q][0]; o[2 * q + 1] = s.acc[q][1]; }
            phase = 0; ++tick; ++row;
        }
    };
    uint32_t t = 0;
    for (; t + 8 <= T; t += 8) {
        const int4 cur = *reinterpret_cast<const int4*>(xr + t);
        const int4 old = *reinterpret_cast<const int4*>(xr + (int64_t)t - 120);
        const int cw[4] = {cur.x, cur.y, cur.z, cur.w};
        const int ow[4] = {old.x, old.y, old.z, old.w};
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int a0 = (int)(int16_t)(cw[q] & 0xFFFF), a1 = cw[q] >> 16;
            const int b0 = (int)(int16_t)(ow[q] & 0xFFFF), b1 = ow[q] >> 16;
            process(scale_sample(a0, invert), scale_sample(b0, invert), pos0 + t + 2 * q);
            process(scale_sample(a1, invert), scale_sample(b1, invert), pos0 + t + 2 * q + 1);
        }
    }
    for (; t < T; ++t)
        process(scale_sample(xr[t], invert), scale_sample(xr[(int64_t)t - 120], invert), pos0 + t);
    state[c] = s;
}

}  // namespace m17
