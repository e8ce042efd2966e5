// K1 (RRC-150 matched filter), the stand-alone correlator operators (limit filter, sync-word correlations: parity
// API), K3 (sliding-DFT carrier detect accumulation).  K2 of the chain is in m17_gate_kernel.hpp.  gfx950 / CDNA4, wave64.  See DESIGN.md §3 for the roofline of each kernel.
#pragma once

#include "m17_common.hpp"

#include <type_traits>

namespace m17 {

typedef float v2f __attribute__((ext_vector_type(2)));

template <bool INVERT>
__device__ __forceinline__ v2f dcd_scale2(int a, int b)   // core::scale_i16 on two samples at once
{
    if (INVERT) { a = (int)(int16_t)(-a); b = (int)(int16_t)(-b); }
    const v2f rcp = {1.0f / 41067.0f, 1.0f / 41067.0f};
    const v2f k = {41067.0f, 41067.0f};
    const v2f fs = {(float)a, (float)b};
    const v2f q = fs * rcp;
    const v2f r = __builtin_elementwise_fma(-q, k, fs);
    return __builtin_elementwise_fma(r, rcp, q);
}

// hand-over between the roles of the pipeline kernels: LDS only.  (__syncthreads() would also fence GLOBAL memory, i.e. wait for the producer's prefetches
// of the blocks to come — vmcnt(0) at every barrier — and expose a full HBM round trip per block.)  Every role executes the
// same number of these, each in its own loop: the hardware counts arrivals per workgroup, not program counters.
__device__ __forceinline__ void dp_handover()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

#ifdef M17_TOOLS   // round 4's K1, kept in the measurement build only (same-box comparisons against the skewed-pair form below)
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

// The tap loop is ROLLED (a straight-line form with the whole 163-sample window of a lane in registers needed 167 VGPRs, three
// waves per SIMD: rounds 1-2, NOTES.md): three banks of R samples (R outputs per lane), 95 VGPRs for R = 15, so that a workgroup of
// it fits into whatever the sequential kernel (96 VGPRs per wave, four waves per SIMD) leaves free on a SIMD.
// A block of R taps reads a 2 R - 1 sample window with static offsets: output r at tap u of block b uses
// W_b[R - 1 + r - u], W_b[k] = win[148 - R b - (R - 1) + k].  W_b = (X_b | X_{b-1}): a new bank X_b of R
// samples per block in front of the previous one — the packed multiplies / adds take their operand pairs from ADJACENT
// registers, so the two banks of a block have to be neighbours: banks P0 P1 P2, even blocks use (P1, P2), odd blocks (P0, P1),
// and after an odd block P0 is copied to P2 (R - 1 moves per two blocks = 4 R (R + 1) / 2 useful instructions).  Taps come from the
// tap table with scalar loads.
template <int R, int MINW>
__global__ __launch_bounds__(FIR_THREADS, MINW) void fir_rrc150_rolled_kernel(const int16_t* __restrict__ x, size_t xpitch,
                                                                             float* __restrict__ y, size_t ypitch, uint32_t T,
                                                                             uint32_t flags, const float* __restrict__ taps)
{
    constexpr int TILE = R * FIR_THREADS, WIN = TILE + NTAPS - 1;   // outputs per workgroup, staged samples
    constexpr int PAD = (R + 3) & ~3;                               // the last bank of lane 0 starts up to R - 1 words before the window
    __shared__ __attribute__((aligned(16))) float win_[PAD + WIN + 4];
    float* win = win_ + PAD;
    const int tid = threadIdx.x;
    const uint32_t c = blockIdx.y;
    const uint32_t t0 = blockIdx.x * TILE;
    const bool invert = flags & 1u;
    const int16_t* xr = x + (size_t)c * xpitch + XPRE;
    float* yr = y + (size_t)c * ypitch + YPRE;
    const int64_t w0 = (int64_t)t0 - (NTAPS - 1);
    for (int k = tid; k < (WIN + 3) / 4; k += FIR_THREADS) {
        const int64_t t = w0 + 4 * k;
        short4 v = make_short4(0, 0, 0, 0);
        if (t < (int64_t)T) v = *reinterpret_cast<const short4*>(xr + t);
        float4 f;
        f.x = scale_sample(v.x, invert);
        f.y = scale_sample(v.y, invert);
        f.z = scale_sample(v.z, invert);
        f.w = scale_sample(v.w, invert);
        *reinterpret_cast<float4*>(&win[4 * k]) = f;
    }
    __syncthreads();
    const float* base = win + tid * R;           // win[j] <-> sample t0 - 148 + j; output o = tid*R + r at tap i reads base[148 + r - i]
    float S[3 * R], acc[R];
#pragma unroll
    for (int r = 0; r < R; ++r) acc[r] = 0.0f;
    // X_b[k] = base[148 - R b - (R - 1) + k]
    auto load_bank = [&](float* bank, int b) {
#pragma unroll
        for (int k = 0; k < R; ++k) bank[k] = base[(NTAPS - 1) - R * b - (R - 1) + k];
    };
    // taps R b .. R b + n - 1 on the window that starts at W[0]
    auto block = [&](const float* W, int b, int n) {
        const float* tp = taps + R * b;           // wave-uniform: scalar loads
#pragma unroll
        for (int u = 0; u < R; ++u) {
            if (u < n) {
                const float tap = tp[u];
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    const float p = W[R - 1 + r - u] * tap;
                    acc[r] = acc[r] + p;
                }
            }
        }
    };
    load_bank(S + R, 0);                          // X_0 -> P1
#pragma unroll
    for (int k = 0; k < R - 1; ++k) S[2 * R + k] = base[NTAPS + k];   // X_{-1}[0 .. R-2] -> P2 (the samples behind output 0's newest)
    S[3 * R - 1] = 0.0f;
    // NB blocks in all, the last one partial: whole pairs of blocks in the loop, then one (even) or two (even, odd) more
    // (R = 15: 4 pairs = taps 0..119, then blocks 8 (15 taps) and 9 (14 taps); R = 14: 5 pairs, then block 10 (9 taps))
    constexpr int NB = (NTAPS + R - 1) / R, PAIRS = (NB - 1) / 2, LAST = NTAPS - R * (NB - 1);
#pragma unroll 1
    for (int q = 0; q < PAIRS; ++q) {
        load_bank(S, 2 * q + 1);                  // X_{2q+1} -> P0 (in flight during the even block)
        block(S + R, 2 * q, R);                   // even block on (P1, P2)
        block(S, 2 * q + 1, R);                   // odd block on (P0, P1)
#pragma unroll
        for (int k = 0; k < R - 1; ++k) S[2 * R + k] = S[k];   // P0 -> P2
        load_bank(S + R, 2 * q + 2);              // X_{2q+2} -> P1 (the last bank may begin before the window: front padding)
    }
    if constexpr (NB - 2 * PAIRS == 2) {
        load_bank(S, 2 * PAIRS + 1);
        block(S + R, 2 * PAIRS, R);
        block(S, 2 * PAIRS + 1, LAST);
    } else {
        block(S + R, 2 * PAIRS, LAST);
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < R; ++r) win[tid * R + r] = acc[r];
    __syncthreads();
    for (int k = tid; k < TILE / 4; k += FIR_THREADS) {
        const uint32_t t = t0 + 4 * k;
        if (t + 3 < T) {
            *reinterpret_cast<float4*>(yr + t) = *reinterpret_cast<const float4*>(&win[4 * k]);
        } else {
            for (int q = 0; q < 4; ++q)
                if (t + q < T) yr[t + q] = win[4 * k + q];
        }
    }
}

#endif  // M17_TOOLS

// =====================================================================================================
// K1  fir_rrc150_skew_kernel — reference a1 + a2: apps/m17-demod.cpp:486-489 (scaling), FirFilter.h:28-43 (y[t] = sum_{i=0}^{148} taps[i] x[t-i],
// accumulated sequentially i = 0..148 in fp32, multiply and add rounded separately; tap 149 is 0.0 and contributes a signed zero only).
// Skewed accumulator pairs (round 5): the same products and the same additions in the same order as the reference's loop, with every VALU instruction of the tap loop a FULL packed operation and no register moves:
//
//   * a lane owns SIXTEEN consecutive outputs as eight accumulator pairs (2q, 2q + 1).  Output o at tap i reads sample o - i, so the
//     even output of a pair at tap u and the odd one at tap u + 1 read the SAME sample: one v_pk_mul_f32 forms
//     (tap_u * w, tap_{u+1} * w) from ONE window register (broadcast through op_sel) and an aligned pair of taps, one v_pk_add_f32
//     adds them to the pair.  The odd output runs one tap ahead of the even one; each output still receives its products
//     i = 0, 1, 2, ... in order (tap_{-1} = tap_149 = 0 pad the two ends: a signed zero added to a sum that is never -0).
//   * step s = 0..149 serves all eight pairs with the tap pair (tap_{s-1}, tap_s); pair q reads lane-relative window element
//     e = 153 + 2q - s.  Every element is used by the eight pairs over fifteen consecutive steps, so the window is a RING of 32
//     registers (slot = e mod 32, two elements per ds_read_b64, issued twelve steps before their first use) and the loop is
//     rolled over bodies of 32 steps with static register numbers: 150 = 22 + 4 x 32, the first pass enters its body at position 10.
//   * LDS window with two spare words per sixteen samples (a lane's base is 18 x lane words: its 8-byte reads fall into distinct
//     banks for sixteen lanes); 16-byte coalesced loads of the int16 input, one conversion per sample (INVERT is a template
//     parameter), outputs stored from the accumulator registers (64 contiguous bytes per lane).
//   * BOUNDED GRID: a workgroup loops over (channel, tile) items, tile fastest, the input of its next item in flight during the tap
//     loop — a launch of a few workgroups per CU leaves its dispatch pipe at once instead of holding it until the last of 50 000
//     workgroups has found a place (NOTES 4.14: whatever shares a pipe with such a grid waits for whole launches).
//   VALU per 16 outputs of a lane: 2400 packed + ~90 = 156 per output (rolled R = 15 form above: 181).
// =====================================================================================================
template <int E, int N, typename F>
__device__ __forceinline__ void static_for(F&& f)
{
    if constexpr (E < N) {
        f(std::integral_constant<int, E>{});
        static_for<E + 1, N>(f);
    }
}

constexpr int FS_R = 16;                              // outputs per lane
constexpr int FS_THREADS = 256;
constexpr int FS_TILE = FS_R * FS_THREADS;            // 4096 outputs per item
constexpr int FS_WOFF = 152;                          // window samples in front of the tile (= XPRE: 16-byte aligned rows)
constexpr int FS_WIN = FS_TILE + FS_WOFF;             // 4248 staged samples = 531 chunks of eight
constexpr int FS_PADF = 18;                           // words in front of the window: the ring's last loads reach below element 0
constexpr int FS_LDS_FLOATS = FS_PADF + ((FS_WIN + 15) / 16) * 18;   // 4806 floats = 19 224 bytes (<= the 20 KB a CU has beside four K5 workgroups)
constexpr int FS_STEPS = 150, FS_BODY = 32, FS_ENTRY = 10, FS_NBODY = 5;   // steps; body length = ring size; 150 = (32 - 10) + 4 x 32
static_assert(FS_BODY - FS_ENTRY + (FS_NBODY - 1) * FS_BODY == FS_STEPS, "step count");
constexpr int FS_TAB = 64;                            // floats per body in the tap table: 32 for even positions, 32 for odd ones

// word offset of lane-relative element e inside a lane's window (two spare words per sixteen samples); e may be negative
constexpr int fs_off(int e) { return e + 2 * ((e - (e < 0 ? 15 : 0)) / 16); }

// Host side: the tap table of the kernel.  Body bi, position p serves step s = 32 (bi - 1) + 22 + p with the pair (T2(s), T2(s + 1)),
// T2(s) = tap_{s-1} (zero outside taps 0..148); even positions read the pair at [p, p + 1] of the first half, odd ones at
// [p - 1, p] of the second half (the same sequence shifted by one), so that every pair is an aligned scalar register pair.
inline void fs_build_tap_table(float* tab /* FS_NBODY * FS_TAB */)
{
    auto T2 = [](int s) { return (s >= 1 && s <= 149) ? rrc_tap(s - 1) : 0.0f; };
    for (int bi = 0; bi < FS_NBODY; ++bi)
        for (int k = 0; k < 32; ++k) {
            const int s = 32 * (bi - 1) + (FS_BODY - FS_ENTRY) + k;
            tab[bi * FS_TAB + k] = T2(s);
            tab[bi * FS_TAB + 32 + k] = T2(s + 1);
        }
}

// (tap_a * w, tap_b * w) for w = the low (HALF = 0) or high (HALF = 1) element of a window register pair: one packed multiply, the window
// element broadcast through op_sel, the taps an aligned scalar register pair
template <int HALF>
__device__ __forceinline__ v2f fs_tap_pair_times(v2f taps, v2f wpair)
{
    // (written as `taps * v2f{w, w}` the compiler selects the same instruction, but then schedules the whole body freely: 128 VGPRs and
    //  spills against 106 with the statements below, and 2-4 % slower on the chip)
    v2f r;
    if constexpr (HALF) asm("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,1]" : "=v"(r) : "s"(taps), "v"(wpair));
    else asm("v_pk_mul_f32 %0, %1, %2 op_sel:[0,0] op_sel_hi:[1,0]" : "=v"(r) : "s"(taps), "v"(wpair));
    return r;
}

template <bool INVERT>
__global__ __launch_bounds__(FS_THREADS, 4) void fir_rrc150_skew_kernel(const int16_t* __restrict__ x, size_t xpitch, float* __restrict__ y, size_t ypitch,
                                                                       uint32_t T, const float* __restrict__ tab, uint32_t tiles, uint32_t items,
                                                                       const uint32_t* __restrict__ first_needed)
{
    // first_needed (may be null): per channel, the first sample of this slab the carrier can be on for (gate_forecast_kernel): tiles that end
    // before it are skipped
    __shared__ __attribute__((aligned(16))) float win[FS_LDS_FLOATS];
    const int tid = threadIdx.x;
    typedef int v4i __attribute__((ext_vector_type(4)));
    constexpr int NCH = (FS_WIN + 7) / 8;             // 531 chunks of eight samples
    constexpr int CPT = (NCH + FS_THREADS - 1) / FS_THREADS;   // 3 per thread (the third for 19 threads only)
    v4i pre[CPT];
    // the int16 input of an item: chunk k <-> window samples 8k .. 8k + 7 <-> times t0 - 152 + 8k ...; beyond the slab's end: zero
    auto fetch = [&](uint32_t item) {
        const uint32_t c = item / tiles, tile = item - c * tiles;
        const int16_t* xr = x + (size_t)c * xpitch + XPRE;
        const int64_t w0 = (int64_t)tile * FS_TILE - FS_WOFF;
#pragma unroll
        for (int q = 0; q < CPT; ++q) {
            const int k = tid + q * FS_THREADS;
            const int64_t t = w0 + 8 * k;
            v4i v = {0, 0, 0, 0};
            if (k < NCH) {
                if (t + 8 <= (int64_t)T) v = *reinterpret_cast<const v4i*>(xr + t);
                else {                                // the slab ends inside this chunk (once per channel at most): sample by sample
                    uint32_t w[4] = {0u, 0u, 0u, 0u};
#pragma unroll 1
                    for (int h = 0; h < 8; ++h)
                        if (t + h < (int64_t)T) w[h >> 1] |= (uint32_t)(uint16_t)xr[t + h] << (16 * (h & 1));
                    v = v4i{(int)w[0], (int)w[1], (int)w[2], (int)w[3]};
                }
            }
            pre[q] = v;
        }
    };
    auto stage = [&] {
#pragma unroll
        for (int q = 0; q < CPT; ++q) {
            const int k = tid + q * FS_THREADS;
            if (k < NCH) {
                float* dst = win + FS_PADF + 8 * k + 2 * (k >> 1);   // sample j = 8k at word j + 2 (j >> 4)
#pragma unroll
                for (int h = 0; h < 4; ++h) {
                    const int w = pre[q][h];
                    const v2f f = dcd_scale2<INVERT>((int)(int16_t)(w & 0xFFFF), w >> 16);
                    *reinterpret_cast<v2f*>(dst + 2 * h) = f;
                }
            }
        }
    };
    auto next_item = [&](uint32_t it) {               // the first item from `it` on (stride = the grid) that is not skipped
        if (first_needed) {
            while (it < items) {
                const uint32_t c = it / tiles, tile = it - c * tiles;
                if ((uint64_t)(tile + 1u) * FS_TILE > (uint64_t)first_needed[c]) break;
                it += gridDim.x;
            }
        }
        return it;
    };
    uint32_t item = next_item(blockIdx.x);
    if (item < items) fetch(item);
    const float* lbase = win + FS_PADF + 18 * tid;    // lane-relative element e at lbase[fs_off(e)]
    while (item < items) {
        const uint32_t c = item / tiles, tile = item - c * tiles;
        const uint32_t following = next_item(item + gridDim.x);
        stage();
        dp_handover();                                // (LDS only: no wait for the stores of the item before)
        if (following < items) fetch(following);      // in flight during the tap loop
        v2f acc[8], ring[16];
        v2f late = {0.0f, 0.0f};                      // pair 7's product of the step before (added one step late: +0 first, harmless)
#pragma unroll
        for (int q = 0; q < 8; ++q) acc[q] = v2f{0.0f, 0.0f};
        // the ring before step 0: the pairs that positions in front of the entry point would have loaded (elements 142 .. 167)
        static_for<0, 13>([&](auto kc) {
            constexpr int e = 142 + 2 * decltype(kc)::value;
            ring[(e & 31) >> 1] = *reinterpret_cast<const v2f*>(lbase + fs_off(e));
        });
        const float* lb = lbase + 36;                 // body b reads from lbase - 36 b: b = -1 first
#pragma unroll 1
        for (int bi = 0; bi < FS_NBODY; ++bi) {
            const float* tb = tab + bi * FS_TAB;      // wave-uniform: scalar loads
            auto half = [&](auto lo_c, auto hi_c) {
                static_for<decltype(lo_c)::value, decltype(hi_c)::value>([&](auto pc) {
                    constexpr int p = decltype(pc)::value;
                    // step s = 32 b + 22 + p; pair q reads element e = 153 + 2q - s = 131 - 32 b + 2q - p: slot (131 + 2q - p) mod 32
                    if constexpr ((p & 1) == 0) {     // the pair of elements first needed twelve steps from now
                        constexpr int e = 118 - p;    // (minus 32 b: folded into lb)
                        ring[(e & 31) >> 1] = *reinterpret_cast<const v2f*>(lb + fs_off(e));
                    }
                    const v2f tp = (p & 1) ? *reinterpret_cast<const v2f*>(tb + 32 + p - 1) : *reinterpret_cast<const v2f*>(tb + p);
                    v2f pr[8];                        // the eight products first, then the eight additions: no dependent neighbours
                    static_for<0, 8>([&](auto qc) {
                        constexpr int q = decltype(qc)::value;
                        constexpr int slot = (131 + 2 * q - p) & 31;
                        pr[q] = fs_tap_pair_times<slot & 1>(tp, ring[slot >> 1]);
                    });
                    // (the compiler counts an asm statement as no wait state at all and pads a reader of ANY asm result that follows a run of
                    //  them with an s_nop: pair 7's addition of the step before goes first — its product is eight real instructions old)
                    acc[7] = acc[7] + late;
#pragma unroll
                    for (int q = 0; q < 7; ++q) acc[q] = acc[q] + pr[q];
                    late = pr[7];
                });
            };
            if (bi > 0) half(std::integral_constant<int, 0>{}, std::integral_constant<int, FS_ENTRY>{});
            half(std::integral_constant<int, FS_ENTRY>{}, std::integral_constant<int, FS_BODY>{});
            lb -= 36;
        }
        acc[7] = acc[7] + late;
        // outputs 16 tid .. 16 tid + 15 of the tile, from the registers
        const uint32_t t = tile * FS_TILE + 16u * (uint32_t)tid;
        float* yo = y + (size_t)c * ypitch + YPRE + t;
        if (t + 16 <= T) {
#pragma unroll
            for (int g = 0; g < 4; ++g) *reinterpret_cast<float4*>(yo + 4 * g) = make_float4(acc[2 * g].x, acc[2 * g].y, acc[2 * g + 1].x, acc[2 * g + 1].y);
        } else {
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                if (t + 2 * q < T) yo[2 * q] = acc[q].x;
                if (t + 2 * q + 1 < T) yo[2 * q + 1] = acc[q].y;
            }
        }
        dp_handover();                                // every wave is through with the window before the next item is staged
        item = following;
    }
}

// =====================================================================================================
// correlate_kernel — reference a4: Correlator::correlate (Correlator.h:51-64) against the four M17
// sync words for every sample: corr[w][c][t] = sum_{i=0}^{7} word[i] * y[t - 70 + 10 i] (oldest symbol
// first, fp32, mul then add).  Time-parallel, elementwise; the 8 taps are shared by the four words.
// =====================================================================================================
// samples [t0, t0 + T) of rows of Ttot samples (a piece in time of the whole pass: m17hip_fir_correlator)
__global__ __launch_bounds__(256) void correlate_kernel(const float* __restrict__ y, size_t ypitch, float* __restrict__ corr,
                                                        uint32_t C, uint32_t T, uint32_t t0, uint32_t Ttot)
{
    const uint32_t c = blockIdx.y;
    const uint32_t t = t0 + blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= t0 + T) return;
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
        corr[((size_t)w * C + c) * Ttot + t] = r;
    }
}

// The same values, FOUR consecutive samples per lane (round 5: beside the matched filter's pieces the one-sample form's 8 four-byte loads, 64 VALU
// operations and 4 four-byte stores per sample took issue slots from K1): 8 sixteen-byte loads (8-byte aligned: t - 70 + 10 i is even) and 4
// sixteen-byte stores per four samples, and |word| * y formed once per tap for the four words (word = +-3: (-3) y = -(3 y) exactly, so
// r + (-3) y = r - 3 y, the same IEEE sum).  Preconditions (else correlate_kernel): t0, T, Ttot multiples of 4.
__global__ __launch_bounds__(256) void correlate4_kernel(const float* __restrict__ y, size_t ypitch, float* __restrict__ corr,
                                                         uint32_t C, uint32_t T, uint32_t t0, uint32_t Ttot)
{
    typedef float v4f __attribute__((ext_vector_type(4)));
    static constexpr int8_t W[4][8] = {{+3, -3, +3, -3, +3, -3, +3, -3}, {+3, +3, +3, +3, -3, -3, +3, -3}, {+3, -3, +3, +3, -3, -3, -3, -3}, {+3, +3, +3, +3, +3, +3, -3, +3}};
    const uint32_t c = blockIdx.y;
    const uint32_t t = t0 + 4u * (blockIdx.x * blockDim.x + threadIdx.x);
    if (t >= t0 + T) return;
    const float* yr = y + (size_t)c * ypitch + YPRE;
    v4f q[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        v4f v;
        __builtin_memcpy(&v, __builtin_assume_aligned(yr + ((int64_t)t - 70 + 10 * i), 8), 16);
        q[i] = 3.0f * v;
    }
#pragma unroll
    for (int w = 0; w < 4; ++w) {
        static_assert(W[0][0] == 3, "");
        v4f r = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
        for (int i = 0; i < 8; ++i) r = W[w][i] > 0 ? r + q[i] : r - q[i];
        *reinterpret_cast<v4f*>(corr + ((size_t)w * C + c) * Ttot + t) = r;
    }
}

// =====================================================================================================
// limit_kernel — reference a3: Correlator::sample's limit_ = BaseIirFilter<float,3>(|y|)
// (Correlator.h:43-45, IirFilter.h:26-42, coefficients Correlator.h:38-39).  A float recurrence: strictly
// sequential per channel, one lane per channel; each lane streams its own row with 16-byte accesses.
// =====================================================================================================
using IirCoef = core::LimitIir;
// one step: returns h0; caller rotates (h2 <- h1, h1 <- h0)
using core::iir_advance;
// The same recurrence over a run of samples with the two products of a sample formed by ONE packed multiply:
// (a1 * h, a2 * h) for the newest history value h gives a1*h1 for the next sample and a2*h2 for the one after
// (identical IEEE products).  m2 carries a2 * h2 between calls.  (Measured in round 2 against two plain v_mul_f32 — shorter
// dependent latency on paper, tools/chain_bench.hip — the packed form is the faster one in the chain: 40.8 against 42.5 ms/step.)
typedef float iir_v2f __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float iir_advance_pk(float in_abs, float h1, float& m2)
{
    const iir_v2f pr = iir_v2f{h1, h1} * iir_v2f{IirCoef::a1, IirCoef::a2};
    float h0 = in_abs - pr.x;
    h0 = h0 - m2;
    m2 = pr.y;
    return h0;
}
using core::iir_output;

// The recurrence over one TICK (192 samples) of an LDS row, IN PLACE (each sample replaced by its history value), as one asm statement:
// three instructions per sample — |y| - a1 h1, - a2 h2, the packed multiply (a1 h, a2 h) with op_sel picking h from the low or the high
// half of the pair the outputs are collected in — the reads two blocks of four ahead, the 16-byte writes left in flight
// (s_waitcnt lgkmcnt(2): the compiler's form of this loop waited for lgkmcnt(0), i.e. for its own write, once per block, and put a v_mov in
// the chain and another beside it per block: 23 ns per sample in K2 against 11 here — tools/issue_bench.hip, NOTES 5.9).
// Registers: v40 row address; v[48:55], v[56:63] the two blocks of eight samples, each register overwritten by its history value (a write
// has read its data when the next instruction issues, and the LDS queue is in order: the read that refills a block comes behind the
// writes that emptied it); v[80:81], v[82:83] the products of the
// last even / odd sample; s[20:21] = (a1, a2).  The row needs 16 bytes of padding behind its 192 samples (two dummy writes
// give the first wait its two writes in flight) and readable memory for 32 bytes behind that (the last read-ahead).
#define M17_IIR_STEP(X, PLO, PHI, SEL, RAlo, RBlo, RBhi) \
    "v_sub_f32_e64 v" #X ", |v" #X "|, v" #RAlo "\n v_sub_f32_e32 v" #X ", v" #X ", v" #RBhi "\n" \
    "v_pk_mul_f32 v[" #RBlo ":" #RBhi "], v[" #PLO ":" #PHI "], s[20:21] op_sel:[" #SEL ",0] op_sel_hi:[" #SEL ",1]\n"
#define M17_IIR_FOUR(X0, X1, X2, X3) \
    M17_IIR_STEP(X0, X0, X1, 0, 82, 80, 81) M17_IIR_STEP(X1, X0, X1, 1, 80, 82, 83) \
    M17_IIR_STEP(X2, X2, X3, 0, 82, 80, 81) M17_IIR_STEP(X3, X2, X3, 1, 80, 82, 83)
#define M17_IIR_RD(V0, V1, OFF) "ds_read_b128 v[" #V0 ":" #V1 "], v40 offset:" #OFF "\n"
#define M17_IIR_WR(V0, V1, OFF) "ds_write_b128 v40, v[" #V0 ":" #V1 "] offset:" #OFF "\n"
#define M17_IIR_16(RB0, RB1, RA0, RA1, W0, W1, W2, W3) \
    "s_waitcnt lgkmcnt(2)\n" M17_IIR_RD(56, 59, RB0) M17_IIR_RD(60, 63, RB1) \
    M17_IIR_FOUR(48, 49, 50, 51) M17_IIR_WR(48, 51, W0) M17_IIR_FOUR(52, 53, 54, 55) M17_IIR_WR(52, 55, W1) \
    "s_waitcnt lgkmcnt(2)\n" M17_IIR_RD(48, 51, RA0) M17_IIR_RD(52, 55, RA1) \
    M17_IIR_FOUR(56, 57, 58, 59) M17_IIR_WR(56, 59, W2) M17_IIR_FOUR(60, 61, 62, 63) M17_IIR_WR(60, 63, W3)
#define M17_IIR_TICK_ASM \
    "s_waitcnt lgkmcnt(0)\n v_mov_b32 v40, %[row]\n v_mov_b32 v62, %[ih1]\n v_mov_b32 v63, %[ih0]\n" \
    "s_mov_b32 s20, 0xbffda16a\n s_mov_b32 s21, 0x3f7b4df5\n s_mov_b32 s22, 6\n" \
    "v_pk_mul_f32 v[80:81], v[62:63], s[20:21] op_sel:[0,0] op_sel_hi:[0,1]\n" \
    "v_pk_mul_f32 v[82:83], v[62:63], s[20:21] op_sel:[1,0] op_sel_hi:[1,1]\n" \
    "ds_read_b128 v[48:51], v40\n ds_read_b128 v[52:55], v40 offset:16\n" \
    "ds_write_b128 v40, v[60:63] offset:768\n ds_write_b128 v40, v[60:63] offset:768\n" \
    "1:\n" M17_IIR_16(32, 48, 64, 80, 0, 16, 32, 48) M17_IIR_16(96, 112, 128, 144, 64, 80, 96, 112) \
    "v_add_u32_e32 v40, 128, v40\n s_sub_u32 s22, s22, 1\n s_cmp_lg_u32 s22, 0\n s_cbranch_scc1 1b\n" \
    "s_waitcnt lgkmcnt(0)\n v_mov_b32 %[h0], v63\n v_mov_b32 %[h1], v62\n v_mov_b32 %[h2], v61\n"
#define M17_IIR_TICK_CLOBBERS \
    "v40", "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55", "v56", "v57", "v58", "v59", "v60", "v61", "v62", "v63", \
    "v80", "v81", "v82", "v83", "s20", "s21", "s22", "scc", "memory"
static_assert(TICK == 192, "M17_IIR_TICK_ASM: six passes of 32 samples");
// h0, h1, h2: the filter history (newest first) before / after the tick whose samples lie at LDS byte address `row`
__device__ __forceinline__ void iir_tick_in_place(uint32_t row, float& h0, float& h1, float& h2)
{
    float n0, n1, n2;
    asm volatile(M17_IIR_TICK_ASM : [h0] "=v"(n0), [h1] "=v"(n1), [h2] "=v"(n2) : [row] "v"(row), [ih0] "v"(h0), [ih1] "v"(h1) : M17_IIR_TICK_CLOBBERS);
    h0 = n0; h1 = n1; h2 = n2;
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
// A float recurrence is sequential in time, so a wave spends its life issuing one short instruction stream; what the
// kernel costs the chip is (waves) x (instructions per sample), and a lone wave issues one instruction per ~2.6 ns
// (tools/k3bench.hip).  Mapping: TWO LANES PER CHANNEL (lane parity = DFT bin), 32 channels per wave:
//   * the complex recurrence on packed fp32 (v_pk_mul_f32 / v_pk_add_f32: IEEE mul/add on two floats per instruction,
//     no contraction): t = Xr + delta; (ac, ad) = (t,t)*(cr,ci); (-bd, bc) = (Xi,Xi)*(-ci,cr); X = (ac + -bd, ad + bc);
//     (p, q) = X*X; n = p + q; the six running sums as three packed adds — 9 VALU instructions per sample for 32 channels.
//   * the time-parallel part (int16 -> float scaling of x[n] and x[n-120], delta) is done for 64 samples of all 32
//     channels at once (each lane converts half of its channel's block, 16-byte loads issued one block ahead) and handed
//     to the recurrence through LDS (row pitch 68: conflict-free 16-byte reads).  x[n-120] comes from the carried prefix
//     of xbuf (XPRE >= 120), so there is no delay line to maintain.
// 128 waves for 4096 channels at ~0.42 wave-instructions per channel-sample: the kernel is latency-bound (~17 ms per
// 480 000 samples) but leaves the SIMDs to the kernels it runs beside; earlier mappings with 16 lanes per channel were
// faster alone (9.6 ms) and five times as expensive in issue slots.
// Sum 5 runs from the stream start (it is read at the first update point only).
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

constexpr int DCD_BLK = 32;        // samples per block: conversion granule and straight-line length of the recurrence
constexpr int DCD_CPW = 32;        // channels per wave
constexpr int DCD_PITCH = DCD_BLK + 4;  // LDS row pitch in floats

struct DcdLane {  // one bin of one channel
    v2f X, cc, cs;        // DFT state (re, im); (cr, ci); (-ci, cr)
    v2f a01, a23, a45;    // the six running sums
};
__device__ __forceinline__ void dcd_step(DcdLane& s, float delta)
{
    const float a = s.X.x + delta;
    const v2f m1 = v2f{a, a} * s.cc;          // (ac, ad)
    const v2f m2 = v2f{s.X.y, s.X.y} * s.cs;  // (-bd, bc): negating a factor negates the product exactly
    s.X = m1 + m2;                            // libstdc++ complex multiply: (ac - bd, ad + bc)
    const v2f p = s.X * s.X;
    const float nrm = p.x + p.y;
    const v2f nn = {nrm, nrm};
    s.a01 = s.a01 + nn; s.a23 = s.a23 + nn; s.a45 = s.a45 + nn;
}

// pos0: absolute index (since reset) of the first sample of this run — identical for every channel.
// DCD_WPB independent waves per workgroup: a workgroup's waves land on one CU, one per SIMD, so the 128 lone waves of a
// 4096-channel launch take a wave slot on every SIMD of 32 CUs instead of one SIMD on each of 128 CUs — K5 (whose workgroups
// need a slot on all four SIMDs of a CU) then loses 32 workgroup slots to K3 instead of 128, K1 likewise.
constexpr int DCD_WPB = 4;
__global__ __launch_bounds__(64 * DCD_WPB) void dcd_kernel(const int16_t* __restrict__ x, size_t xpitch, DcdState* __restrict__ state,
                                                 float* __restrict__ table, uint32_t ticks_cap, uint32_t C, uint32_t T,
                                                 uint64_t pos0, DcdCoef k, uint32_t flags)
{
    __shared__ __attribute__((aligned(16))) float dl_all[DCD_WPB][DCD_CPW][DCD_PITCH];
    float (*dl)[DCD_PITCH] = dl_all[threadIdx.x >> 6];
    const int lane = threadIdx.x & 63;
    const int g = lane >> 1, bin = lane & 1;
    uint32_t c = (blockIdx.x * DCD_WPB + (threadIdx.x >> 6)) * DCD_CPW + g;
    const bool live = c < C;   // lanes beyond the last channel shadow it and never store
    if (!live) c = C - 1;
    const bool invert = flags & 1u;
    const int16_t* xr = x + (size_t)c * xpitch + XPRE;
    DcdState* st = state + c;
    DcdLane s;
    s.X = v2f{st->xr[bin], st->xi[bin]};
    s.cc = bin ? v2f{k.c1r, k.c1i} : v2f{k.c0r, k.c0i};
    s.cs = v2f{-s.cc.y, s.cc.x};
    s.a01 = v2f{st->acc[0][bin], st->acc[1][bin]};
    s.a23 = v2f{st->acc[2][bin], st->acc[3][bin]};
    s.a45 = v2f{st->acc[4][bin], st->acc[5][bin]};
    float* tab = table + (size_t)c * ticks_cap * 12 + bin * 6;
    const float* mydl = dl[g];
    uint32_t phase = (uint32_t)(pos0 % TICK);  // position inside the current tick (wave-uniform)
    uint64_t tick = pos0 / TICK;
    uint32_t row = 0;
    auto lds_sync = [] {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront", "local");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront", "local");
    };
    auto tick_begin = [&] {  // the sum that restarts with this tick
        const uint32_t j = (uint32_t)(tick % 5);
        if (j == 0) s.a01.x = 0.f;
        if (j == 1) s.a01.y = 0.f;
        if (j == 2) s.a23.x = 0.f;
        if (j == 3) s.a23.y = 0.f;
        if (j == 4) s.a45.x = 0.f;
    };
    auto tick_end = [&] {
        if (live) {
            float* o = tab + (size_t)row * 12;
            *reinterpret_cast<float2*>(o) = make_float2(s.a01.x, s.a01.y);
            *reinterpret_cast<float2*>(o + 2) = make_float2(s.a23.x, s.a23.y);
            *reinterpret_cast<float2*>(o + 4) = make_float2(s.a45.x, s.a45.y);
        }
        phase = 0; ++tick; ++row;
    };
    auto conv = [&](int v) { return scale_sample_mul(v, invert); };
    auto one_sample = [&](uint32_t t) {  // generic path: head / tail of a run
        if (phase == 0) tick_begin();
        dcd_step(s, conv((int)xr[t]) - conv((int)xr[(int64_t)t - 120]));
        if (++phase == TICK) tick_end();
    };

    uint32_t t = 0;
    while (t < T && (phase % DCD_BLK) != 0) { one_sample(t); ++t; }   // head: up to a block boundary of the tick
    // whole blocks: lane (g, bin) converts samples [HALF bin, HALF bin + HALF) of its channel's block
    if (t + DCD_BLK <= T) {
        constexpr int HALF = DCD_BLK / 2, NQ = HALF / 8;
        int4 pa[NQ], pb[NQ];
        auto issue = [&](uint32_t t0) {
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
                const int16_t* p = xr + (size_t)t0 + HALF * bin + 8 * q;
                pa[q] = *reinterpret_cast<const int4*>(p);
                pb[q] = *reinterpret_cast<const int4*>(p - 120);
            }
        };
        auto lo = [](int w) { return (int)(int16_t)(w & 0xFFFF); };
        auto hi = [](int w) { return w >> 16; };
        issue(t);
        for (; t + DCD_BLK <= T; t += DCD_BLK) {
            float* wrow = dl[g] + HALF * bin;
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
                const int4 a = pa[q], d = pb[q];
                float4 u, v;
                u.x = conv(lo(a.x)) - conv(lo(d.x)); u.y = conv(hi(a.x)) - conv(hi(d.x));
                u.z = conv(lo(a.y)) - conv(lo(d.y)); u.w = conv(hi(a.y)) - conv(hi(d.y));
                v.x = conv(lo(a.z)) - conv(lo(d.z)); v.y = conv(hi(a.z)) - conv(hi(d.z));
                v.z = conv(lo(a.w)) - conv(lo(d.w)); v.w = conv(hi(a.w)) - conv(hi(d.w));
                *reinterpret_cast<float4*>(wrow + 8 * q) = u;
                *reinterpret_cast<float4*>(wrow + 8 * q + 4) = v;
            }
            lds_sync();
            if (t + 2 * DCD_BLK <= T) issue(t + DCD_BLK);   // in flight while the recurrence below runs
            if (phase == 0) tick_begin();
            float d[DCD_BLK];
#pragma unroll
            for (int u = 0; u < DCD_BLK / 4; ++u) {
                const float4 v = *reinterpret_cast<const float4*>(mydl + 4 * u);
                d[4 * u] = v.x; d[4 * u + 1] = v.y; d[4 * u + 2] = v.z; d[4 * u + 3] = v.w;
            }
#pragma unroll
            for (int u = 0; u < DCD_BLK; ++u) dcd_step(s, d[u]);
            phase += DCD_BLK;
            if (phase == TICK) tick_end();
            lds_sync();
        }
    }
    for (; t < T; ++t) one_sample(t);  // tail
    if (live) {
        st->xr[bin] = s.X.x; st->xi[bin] = s.X.y;
        st->acc[0][bin] = s.a01.x; st->acc[1][bin] = s.a01.y; st->acc[2][bin] = s.a23.x;
        st->acc[3][bin] = s.a23.y; st->acc[4][bin] = s.a45.x; st->acc[5][bin] = s.a45.y;
    }
}


// =====================================================================================================
// K3 as a PIPELINE OF FOUR WAVES — the LATENCY form of K3 (m17hip_tune key 10; default: runs whose front end was queued by m17hip_demod_front,
// i.e. a continued stream, where K3's ten-launch chain is what a run waits for; batch runs use dcd_kernel above, which costs the
// kernels beside it a quarter of the wave slots).
// The sliding-DFT recurrence is a latency chain: a lone wave issues one instruction per ~2.6 ns whatever it depends on, so the
// time of this kernel is (instructions per sample ON THE WAVE THAT CARRIES THE RECURRENCE) x 2.6 ns x samples.  In dcd_kernel
// that wave also converts the samples, squares the bins and feeds six running sums: ~13 instructions per sample, 16.8 ms per
// 480 000 samples.  Here a workgroup of four waves (32 channels, one wave per SIMD of a CU) splits the work by ROLE and hands
// blocks of 32 samples from role to role through LDS, one barrier per block:
//   P  (wave 0)  int16 -> float scaling of x[n] and x[n-120] (packed fp32: v_pk_mul / v_pk_fma), delta = x[n] - x[n-120]
//                for 32 channels x 32 samples, 16-byte loads issued one block ahead                 -> dbuf[2][32][36]
//   R  (wave 1)  the recurrence and nothing else: t = Xr + delta; X = (t,t)*(cr,ci) + (Xi,Xi)*(-ci,cr)   (4 VALU per sample,
//                2 lanes per channel = the two DFT bins, as in dcd_kernel)                          -> xb[2][16][64] (re, im pairs)
//   A0 (wave 2)  norm = re*re + im*im and the running sums restarted at ticks = 0,1,2,3 (mod 5); table columns 0..3
//   A1 (wave 3)  norm again and the sum restarted at ticks = 4 (mod 5) + the sum since the stream start; table columns 4..5
// Arithmetic, summation order and table layout are dcd_kernel's (and the reference's): bit-identical tables.
// Runs whose start and length are multiples of 32 samples take this kernel (every block whole, tick boundaries = block
// boundaries); anything else (ragged streaming chunks) takes dcd_kernel.
// =====================================================================================================
constexpr int DP_BLK = 32;               // samples per pipeline stage
constexpr int DP_CPB = 32;               // channels per workgroup
constexpr int DP_DPITCH = DP_BLK + 4;    // delta row pitch in floats (conflict-free 16-byte reads)
constexpr int DP_PF = 3;                 // blocks of input the producer keeps in flight

// preconditions (the host launches dcd_kernel otherwise): T and pos0 are multiples of DP_BLK, T >= 4 blocks, so every block
// is whole and tick boundaries (192 = 6 x 32 samples) are block boundaries
template <bool INVERT>
__global__ __launch_bounds__(256) void dcd_pipe_kernel(const int16_t* __restrict__ x, size_t xpitch, DcdState* __restrict__ state,
                                                       float* __restrict__ table, uint32_t ticks_cap, uint32_t C, uint32_t T,
                                                       uint64_t pos0, DcdCoef k, uint32_t flags)
{
    __shared__ __attribute__((aligned(16))) float dbuf[2][DP_CPB][DP_DPITCH];
    __shared__ __attribute__((aligned(16))) float4 xb[2][DP_BLK / 2][64];
    const int role = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int lane = threadIdx.x & 63;
    const int g = lane >> 1, bin = lane & 1;
    uint32_t c = blockIdx.x * DP_CPB + g;
    const bool live = c < C;   // lanes beyond the last channel shadow it and never store
    if (!live) c = C - 1;
    const int16_t* xr = x + (size_t)c * xpitch + XPRE;
    DcdState* st = state + c;
    const uint32_t NB = T / DP_BLK;       // blocks
    // every role runs NI hand-overs: NB + 2 for pipeline fill and drain, rounded up to a multiple of DP_PF so that the
    // producer's unrolled loop has no early exit (its registers stay in fixed slots and its waits stay partial)
    const uint32_t NI = (NB + 2u + DP_PF - 1u) / DP_PF * DP_PF;

    if (role == 0) {
        // ---- P: scaling and delta.  A block lasts ~0.5 us, an HBM round trip twice that: the loads of block b are issued DP_PF
        // blocks ahead (slot = b % DP_PF is a compile-time constant: the loop is unrolled DP_PF times and every body issues the
        // same loads — past the end the last block again — so the wait for a slot leaves the other slots in flight).
        int4 pa[DP_PF][2], pb[DP_PF][2];
        auto issue = [&](uint32_t b, int slot) {   // lane (g, bin) converts samples [16 bin, 16 bin + 16) of its channel's block
            const int16_t* p = xr + (size_t)min(b, NB - 1u) * DP_BLK + 16 * bin;
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                pa[slot][q] = *reinterpret_cast<const int4*>(p + 8 * q);
                pb[slot][q] = *reinterpret_cast<const int4*>(p + 8 * q - 120);
            }
        };
#pragma unroll
        for (int j = 0; j < DP_PF; ++j) issue((uint32_t)j, j);
        auto lo = [](int w) { return (int)(int16_t)(w & 0xFFFF); };
        auto hi = [](int w) { return w >> 16; };
        for (uint32_t i0 = 0; i0 < NI; i0 += DP_PF) {
#pragma unroll
            for (int slot = 0; slot < DP_PF; ++slot) {
                const uint32_t i = i0 + (uint32_t)slot;
                {
                    float4 o[4];
#pragma unroll
                    for (int q = 0; q < 2; ++q) {
                        const int4 a = pa[slot][q], d = pb[slot][q];
                        const v2f u0 = dcd_scale2<INVERT>(lo(a.x), hi(a.x)) - dcd_scale2<INVERT>(lo(d.x), hi(d.x));
                        const v2f u1 = dcd_scale2<INVERT>(lo(a.y), hi(a.y)) - dcd_scale2<INVERT>(lo(d.y), hi(d.y));
                        const v2f u2 = dcd_scale2<INVERT>(lo(a.z), hi(a.z)) - dcd_scale2<INVERT>(lo(d.z), hi(d.z));
                        const v2f u3 = dcd_scale2<INVERT>(lo(a.w), hi(a.w)) - dcd_scale2<INVERT>(lo(d.w), hi(d.w));
                        o[2 * q] = make_float4(u0.x, u0.y, u1.x, u1.y);
                        o[2 * q + 1] = make_float4(u2.x, u2.y, u3.x, u3.y);
                    }
                    issue(i + DP_PF, slot);   // this slot's registers are free again: block i + DP_PF goes in flight
                    if (i < NB && !(flags & 16u)) {
                        float* wrow = &dbuf[i & 1u][g][16 * bin];
#pragma unroll
                        for (int q = 0; q < 4; ++q) *reinterpret_cast<float4*>(wrow + 4 * q) = o[q];
                    }
                    dp_handover();
                }
            }
        }
    } else if (role == 1) {
        // ---- R: the recurrence and nothing else
        v2f X = v2f{st->xr[bin], st->xi[bin]};
        const v2f cc = bin ? v2f{k.c1r, k.c1i} : v2f{k.c0r, k.c0i};
        const v2f cs = v2f{-cc.y, cc.x};
        auto step = [&](float delta) {
            const float a = X.x + delta;
            const v2f m1 = v2f{a, a} * cc;          // (ac, ad)
            const v2f m2 = v2f{X.y, X.y} * cs;      // (-bd, bc)
            X = m1 + m2;                            // (ac - bd, ad + bc)
        };
        for (uint32_t i = 0; i < NI; ++i) {
            if (i >= 1u && i <= NB && !(flags & 32u)) {
                const uint32_t b = i - 1u;
                const float* drow = &dbuf[b & 1u][g][0];
                float4* out = &xb[b & 1u][0][lane];
                float d[DP_BLK];
#pragma unroll
                for (int u = 0; u < DP_BLK / 4; ++u) {
                    const float4 v = *reinterpret_cast<const float4*>(drow + 4 * u);
                    d[4 * u] = v.x; d[4 * u + 1] = v.y; d[4 * u + 2] = v.z; d[4 * u + 3] = v.w;
                }
#pragma unroll
                for (int u = 0; u < DP_BLK; u += 2) {
                    step(d[u]); const v2f x0 = X;
                    step(d[u + 1]);
                    out[(u / 2) * 64] = make_float4(x0.x, x0.y, X.x, X.y);
                }
            }
            dp_handover();
        }
        if (live) { st->xr[bin] = X.x; st->xi[bin] = X.y; }
    } else {
        // ---- A0 / A1: norms and running sums; A0 (role 2) owns the sums restarted at ticks = 0..3 (mod 5), A1 the one restarted
        // at ticks = 4 (mod 5) and the one that runs from the stream start
        const bool a0 = role == 2;
        v2f s01 = a0 ? v2f{st->acc[0][bin], st->acc[1][bin]} : v2f{st->acc[4][bin], st->acc[5][bin]};
        v2f s23 = a0 ? v2f{st->acc[2][bin], st->acc[3][bin]} : v2f{0.f, 0.f};
        uint32_t phase = (uint32_t)(pos0 % TICK);
        uint64_t tick = pos0 / TICK;
        uint32_t row = 0;
        float* tab = table + (size_t)c * ticks_cap * 12 + bin * 6;
        const bool skip = flags & (a0 ? 64u : 128u);
        for (uint32_t i = 0; i < NI; ++i) {
            if (i >= 2u && i < NB + 2u && !skip) {
                const uint32_t b = i - 2u;
                const float4* in = &xb[b & 1u][0][lane];
                if (phase == 0) {   // the sum that restarts with this tick
                    const uint32_t j = (uint32_t)(tick % 5);
                    if (a0) { if (j == 0) s01.x = 0.f; if (j == 1) s01.y = 0.f; if (j == 2) s23.x = 0.f; if (j == 3) s23.y = 0.f; }
                    else if (j == 4) s01.x = 0.f;
                }
                auto acc = [&](float re, float im) {
                    const v2f xx = {re, im};
                    const v2f p = xx * xx;
                    const float nrm = p.x + p.y;
                    const v2f nn = {nrm, nrm};
                    s01 = s01 + nn;
                    if (a0) s23 = s23 + nn;
                };
#pragma unroll
                for (int u = 0; u < DP_BLK / 2; ++u) {
                    const float4 v = in[u * 64];
                    acc(v.x, v.y);
                    acc(v.z, v.w);
                }
                phase += DP_BLK;
                if (phase == TICK) {
                    if (live) {
                        float* o = tab + (size_t)row * 12;
                        if (a0) {
                            *reinterpret_cast<float2*>(o) = make_float2(s01.x, s01.y);
                            *reinterpret_cast<float2*>(o + 2) = make_float2(s23.x, s23.y);
                        } else {
                            *reinterpret_cast<float2*>(o + 4) = make_float2(s01.x, s01.y);
                        }
                    }
                    phase = 0; ++tick; ++row;
                }
            }
            dp_handover();
        }
        if (live) {
            if (a0) { st->acc[0][bin] = s01.x; st->acc[1][bin] = s01.y; st->acc[2][bin] = s23.x; st->acc[3][bin] = s23.y; }
            else { st->acc[4][bin] = s01.x; st->acc[5][bin] = s01.y; }
        }
    }
}


// =====================================================================================================
// The correlator's limit filter for the per-operator entry point (m17hip_correlator, BASELINE config 2) as a pipeline of roles,
// (the pattern of round 2's K3 pipeline, NOTES.md): limit_kernel above runs one LANE per channel with 4-byte loads 64 rows apart (uncoalesced)
// and carries output arithmetic and stores on the wave that carries the recurrence — 70 ms for 1024 x 480 000 samples.  Here a
// workgroup = SIXTEEN channels x tiles of 256 samples, five waves:
//   P0, P1  16-byte coalesced loads of the matched-filter rows (a whole 1 KB row segment per instruction, 8 rows each), three tiles in flight -> ytile[2][16][260]
//   R       h0 = |y| - a1 h1 - a2 h2 for its lane's channel, nothing else (the dependent chain), on lanes 0..15: ONE quarter of the
//           wave, where the chain with a packed multiply is at its shortest (7.3 ns per sample against 12.3 on 64 lanes, NOTES 4.12); the
//           reads run two blocks of four ahead, the multiply takes h from the register pair the outputs are collected in   -> htile[2][16][4 + 260]
//   O0, O1  limit = b0 h0 + b1 h1 + b2 h2 from the trajectory, time-parallel, 16-byte coalesced stores (8 rows each)
// (Round 3's form had 64 channels per workgroup, the recurrence on 64 lanes: 17.6 ns per sample, 8.5 ms for 480 000.)
// Bit-identical to limit_kernel (same IEEE products and sums).  Preconditions (else the host launches limit_kernel): T a multiple of 256.
// =====================================================================================================
constexpr int LP_CH = 16;                 // channels per workgroup
constexpr int LP_TILE = 256;
constexpr int LP_YP = LP_TILE + 4;        // ytile row pitch (floats): sixteen rows start in sixteen different bank groups
constexpr int LP_HP = 4 + LP_TILE + 4;    // htile row pitch: 4 floats of the previous tile in front
constexpr int LP_PF = 3;

#ifdef M17_TOOLS   // round 4's form, kept in the measurement build for same-box comparisons (m17hip_tune key 27); the product runs limit_relay_kernel below
// `T` samples from y (row pitch ypitch) to limit (row pitch lpitch); state (may be null = zero history / not kept): the last four history values of a
// channel, h[-4 .. -1], so that a run can be cut into pieces in time (m17hip_fir_correlator pipelines the matched filter with this chain)
__global__ __launch_bounds__(320, 1) void limit_pipe_kernel(const float* __restrict__ y, size_t ypitch, float* __restrict__ limit, size_t lpitch, uint32_t C, uint32_t T,
                                                            const float* __restrict__ state_in, float* __restrict__ state_out)
{
    __shared__ __attribute__((aligned(16))) float ytile[2][LP_CH][LP_YP];
    __shared__ __attribute__((aligned(16))) float htile[2][LP_CH][LP_HP];
    const int role = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));   // 0, 1 = P0, P1; 2 = R; 3, 4 = O0, O1
    const int lane = threadIdx.x & 63;
    const uint32_t c0 = blockIdx.x * (uint32_t)LP_CH;
    const uint32_t NT = T / LP_TILE;
    const uint32_t NI = (NT + 2u + LP_PF - 1u) / LP_PF * LP_PF;
    auto row_of = [&](uint32_t r) -> uint32_t { return min(c0 + r, C - 1u); };   // rows beyond the last channel shadow it
    typedef float v4f __attribute__((ext_vector_type(4)));
    if (role <= 1) {
        // lane = float4 column of the tile: 64 lanes x 16 bytes = the whole 1 KB row segment; eight rows per producer
        // three tiles in flight.  (Native vector type on purpose: an array of eight HIP float4 STRUCTS is not promoted to
        // registers by this compiler and lands in scratch memory.)
        v4f pf[LP_PF][8];
        const float* ybase[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) ybase[q] = y + (size_t)row_of(8u * (uint32_t)role + (uint32_t)q) * ypitch + YPRE + 4u * (uint32_t)lane;
        auto issue = [&](uint32_t tile, int slot) {
            const uint32_t t0 = min(tile, NT - 1u) * LP_TILE;
#pragma unroll
            for (int q = 0; q < 8; ++q) pf[slot][q] = *reinterpret_cast<const v4f*>(ybase[q] + t0);
        };
#pragma unroll
        for (int j = 0; j < LP_PF; ++j) issue((uint32_t)j, j);
        for (uint32_t i0 = 0; i0 < NI; i0 += LP_PF) {
#pragma unroll
            for (int slot = 0; slot < LP_PF; ++slot) {
                const uint32_t i = i0 + (uint32_t)slot;
                v4f v[8];
#pragma unroll
                for (int q = 0; q < 8; ++q) v[q] = pf[slot][q];
                issue(i + LP_PF, slot);   // tile i leaves its registers for LDS, tile i + LP_PF takes them
                if (i < NT) {
#pragma unroll
                    for (int q = 0; q < 8; ++q)
                        *reinterpret_cast<v4f*>(&ytile[i & 1u][8u * (uint32_t)role + (uint32_t)q][4u * (uint32_t)lane]) = v[q];
                }
                dp_handover();
            }
        }
    } else if (role == 2) {
        // the filter starts from zero history; p1 = (h[-2], h[-1]), m2 = a2 h[-2]: the pair the newest outputs sit in (see nf_serve_limit, m17_wave_kernel.hpp)
        iir_v2f p0 = {0.f, 0.f}, p1 = {0.f, 0.f};
        const iir_v2f coef = {IirCoef::a1, IirCoef::a2};
        typedef float v4f_ __attribute__((ext_vector_type(4)));
        v4f_ tail = {0.f, 0.f, 0.f, 0.f};
        if (state_in && lane < LP_CH) {
            tail = *reinterpret_cast<const v4f_*>(state_in + 4 * (size_t)row_of((uint32_t)lane));
            p1 = iir_v2f{tail.z, tail.w};
        }
        float m2 = IirCoef::a2 * p1.x;                // a2 h[-2]: the product the sample before the first one left behind
        __builtin_amdgcn_s_setprio(3);                // the chain's instructions first, whatever shares the SIMD (the matched filter's packed operations, four cycles each)
        // (a1 h, a2 h) for h = the low / high element of a pair whose BOTH elements are live: the compiler selects v_pk_mul_f32 with op_sel on
        // that pair.  (As an asm statement the multiply drew an s_nop in front of its reader — the compiler takes an asm result for a forwarding
        // hazard — one more issue slot in a four-instruction dependent chain.)
        auto lo = [&](iir_v2f p) { return iir_v2f{p.x, p.x} * coef; };
        auto hi = [&](iir_v2f p) { return iir_v2f{p.y, p.y} * coef; };
        auto four = [&](const v4f v) {   // h0 = (|y| - a1 h1) - a2 h2, the two products of a history value from one packed multiply
            iir_v2f r;
            r = hi(p1); p0.x = (fabsf(v.x) - r.x) - m2; m2 = r.y;
            r = lo(p0); p0.y = (fabsf(v.y) - r.x) - m2; m2 = r.y;
            r = hi(p0); p1.x = (fabsf(v.z) - r.x) - m2; m2 = r.y;
            r = lo(p1); p1.y = (fabsf(v.w) - r.x) - m2; m2 = r.y;
            return v4f{p0.x, p0.y, p1.x, p1.y};
        };
        for (uint32_t i = 0; i < NI; ++i) {
            if (i >= 1u && i <= NT && lane < LP_CH) {
                const uint32_t b = (i - 1u) & 1u;
                const v4f* yrow = reinterpret_cast<const v4f*>(&ytile[b][lane][0]);
                v4f* hrow = reinterpret_cast<v4f*>(&htile[b][lane][0]);
                hrow[0] = tail;   // the last four values of the previous tile, for the output stage
                v4f c0v = yrow[0], c1v = yrow[1];
#pragma unroll 4
                for (int q = 0; q < LP_TILE / 4; q += 2) {
                    const v4f v0 = c0v, v1 = c1v;
                    c0v = yrow[min(q + 2, LP_TILE / 4 - 2)];
                    c1v = yrow[min(q + 3, LP_TILE / 4 - 1)];
                    __builtin_amdgcn_sched_barrier(0);
                    hrow[1 + q] = four(v0);
                    tail = four(v1);
                    hrow[2 + q] = tail;
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            dp_handover();
        }
        if (state_out && lane < LP_CH && c0 + (uint32_t)lane < C) *reinterpret_cast<v4f_*>(state_out + 4 * (size_t)(c0 + (uint32_t)lane)) = tail;
    } else {
        for (uint32_t i = 0; i < NI; ++i) {
            if (i >= 2u && i < NT + 2u) {
                const uint32_t tile = i - 2u, b = tile & 1u, col = (uint32_t)lane;
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    const uint32_t r = 8u * (uint32_t)(role - 3) + (uint32_t)q;
                    const float* hrow = &htile[b][r][0];
                    const float4 p = *reinterpret_cast<const float4*>(hrow + 4u * col);        // h[k0-4 .. k0-1]
                    const float4 h = *reinterpret_cast<const float4*>(hrow + 4u + 4u * col);   // h[k0 .. k0+3]
                    float4 o;
                    o.x = iir_output(h.x, p.w, p.z);
                    o.y = iir_output(h.y, h.x, p.w);
                    o.z = iir_output(h.z, h.y, h.x);
                    o.w = iir_output(h.w, h.z, h.y);
                    if (c0 + r < C) *reinterpret_cast<float4*>(limit + (size_t)(c0 + r) * lpitch + (size_t)tile * LP_TILE + 4u * col) = o;
                }
            }
            dp_handover();
        }
    }
}
#endif

// =====================================================================================================
// The same filter with the recurrence RELAYED between two waves (round 5, NOTES 5.9).  What a lone wave pays is the NUMBER of
// instructions it issues — one per four cycles whether dependent or not (`tools/issue_bench.hip`: the bare three-instruction chain 6.0 ns per
// sample, every further independent instruction + 1.7 ns) — and a 16-byte LDS write holds the issue for ~ 28 cycles (VGPR data read out at
// issue; two 8-byte writes cost the same): in limit_pipe_kernel's R wave the hand-over of the trajectory was a third of the time per
// sample.  Here the wave that carries the chain issues NOTHING else: a tile's 128 samples sit in 128 registers, each overwritten in place by
// its history value, three instructions per sample in one asm statement (no compiler-made v_mov / s_nop between them); then the chain's
// state (h[-2], h[-1], eight bytes through LDS) goes to the OTHER recurrence wave, which has the next tile in its registers already, and while
// that one computes, the first writes its 128 values out and reads the tile after next in.
//   step i:   P        tile i     global -> ytile[i & 1]                     (one wave: 16 rows x 512 bytes, four tiles in flight)
//             R(w)     w = parity of its tiles.  (i - w) even: the chain over tile i - 2;
//                      odd: tile i - 3's values -> htile[w], tile i - 1's samples ytile[w] -> registers
//             O0, O1   tile i - 4: limit = b0 h0 + b1 h1 + b2 h2 from htile, 16-byte coalesced stores (8 rows each)
// One workgroup barrier per step.  Bit-identical to limit_kernel (the same IEEE products and differences in the same order).
// Preconditions (else the host launches limit_kernel): T a multiple of 128.
// =====================================================================================================
constexpr int LR_CH = 16;
constexpr int LR_TILE = 128;
constexpr int LR_YP = LR_TILE + 4;        // row pitch (floats): sixteen rows start in sixteen different bank groups
constexpr int LR_HP = 4 + LR_TILE + 4;    // 4 floats of history in front ([2], [3] = h[-2], h[-1])
constexpr int LR_PF = 4;
#define LR_CLOBBERS \
    "v96", "v97", "v98", "v99", "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107", "v108", "v109", "v110", "v111", "v112", "v113", "v114", "v115", \
    "v116", "v117", "v118", "v119", "v120", "v121", "v122", "v123", "v124", "v125", "v126", "v127", "v128", "v129", "v130", "v131", "v132", "v133", "v134", "v135", \
    "v136", "v137", "v138", "v139", "v140", "v141", "v142", "v143", "v144", "v145", "v146", "v147", "v148", "v149", "v150", "v151", "v152", "v153", "v154", "v155", \
    "v156", "v157", "v158", "v159", "v160", "v161", "v162", "v163", "v164", "v165", "v166", "v167", "v168", "v169", "v170", "v171", "v172", "v173", "v174", "v175", \
    "v176", "v177", "v178", "v179", "v180", "v181", "v182", "v183", "v184", "v185", "v186", "v187", "v188", "v189", "v190", "v191", "v192", "v193", "v194", "v195", \
    "v196", "v197", "v198", "v199", "v200", "v201", "v202", "v203", "v204", "v205", "v206", "v207", "v208", "v209", "v210", "v211", "v212", "v213", "v214", "v215", \
    "v216", "v217", "v218", "v219", "v220", "v221", "v222", "v223", "v224", "v225", "v226", "v227", "v228", "v229", "v230", \
    "s20", "s21", "s22", "s23", "s24", "s25", "scc", "memory"

// the recurrence waves' whole life as ONE asm statement: registers v96 .. v223 hold a tile across steps, nothing of the compiler's runs in between
// v[224:225] = (h[-2], h[-1]) of the tile in the registers; v[226:227], v[228:229] = (a1 h, a2 h) of the last even / odd sample; v230 = |x| - a1 h1
#define LR_PAIR(E0, E1) \
    "v_sub_f32_e64 v230, |v[" E0 "+4*m17k]|, v228\n v_sub_f32_e32 v[" E0 "+4*m17k], v230, v227\n" \
    "v_pk_mul_f32 v[226:227], v[" E0 "+4*m17k:" E1 "+4*m17k], s[20:21] op_sel:[0,0] op_sel_hi:[0,1]\n" \
    "v_sub_f32_e64 v230, |v[" E1 "+4*m17k]|, v226\n v_sub_f32_e32 v[" E1 "+4*m17k], v230, v229\n" \
    "v_pk_mul_f32 v[228:229], v[" E0 "+4*m17k:" E1 "+4*m17k], s[20:21] op_sel:[1,0] op_sel_hi:[1,1]\n"

__global__ __launch_bounds__(320, 1) void limit_relay_kernel(const float* __restrict__ y, size_t ypitch, float* __restrict__ limit, size_t lpitch, uint32_t C, uint32_t T,
                                                             const float* __restrict__ state_in, float* __restrict__ state_out)
{
    __shared__ __attribute__((aligned(16))) float ytile[2][LR_CH][LR_YP];
    __shared__ __attribute__((aligned(16))) float htile[2][LR_CH][LR_HP];
    __shared__ __attribute__((aligned(16))) float relay[LR_CH][2];
    const int role = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));   // 0 = P; 1, 2 = R(0), R(1); 3, 4 = O0, O1
    const int lane = threadIdx.x & 63;
    const uint32_t c0 = blockIdx.x * (uint32_t)LR_CH;
    const uint32_t NT = T / LR_TILE;
    const uint32_t NI = (NT + 4u + LR_PF - 1u) / LR_PF * LR_PF;
    auto row_of = [&](uint32_t r) -> uint32_t { return min(c0 + r, C - 1u); };   // rows beyond the last channel shadow it
    typedef float v4f __attribute__((ext_vector_type(4)));
    const uint32_t sub = (uint32_t)lane >> 5, col = (uint32_t)lane & 31u;       // P, O: two rows per instruction, 32 lanes x 16 bytes = a row's 512-byte tile
    if (role == 0) {
        v4f pf[LR_PF][8];
        const float* ybase[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) ybase[q] = y + (size_t)row_of(2u * (uint32_t)q + sub) * ypitch + YPRE + 4u * col;
        auto issue = [&](uint32_t tile, int slot) {
            const uint32_t t0 = min(tile, NT - 1u) * LR_TILE;
#pragma unroll
            for (int q = 0; q < 8; ++q) pf[slot][q] = *reinterpret_cast<const v4f*>(ybase[q] + t0);
        };
#pragma unroll
        for (int j = 0; j < LR_PF; ++j) issue((uint32_t)j, j);
        for (uint32_t i0 = 0; i0 < NI; i0 += LR_PF) {
#pragma unroll
            for (int slot = 0; slot < LR_PF; ++slot) {
                const uint32_t i = i0 + (uint32_t)slot;
                v4f v[8];
#pragma unroll
                for (int q = 0; q < 8; ++q) v[q] = pf[slot][q];
                issue(i + LR_PF, slot);
                if (i < NT) {
#pragma unroll
                    for (int q = 0; q < 8; ++q) *reinterpret_cast<v4f*>(&ytile[i & 1u][2u * (uint32_t)q + sub][4u * col]) = v[q];
                }
                dp_handover();
            }
        }
    } else if (role <= 2) {
        const uint32_t w = (uint32_t)(role - 1), r = (uint32_t)lane & 15u;
        float h2 = 0.f, h1 = 0.f;
        if (state_in) { const v4f s4 = *reinterpret_cast<const v4f*>(state_in + 4 * (size_t)row_of(r)); h2 = s4.z; h1 = s4.w; }
        const uint32_t ya = (uint32_t)(uintptr_t)as_lds(&ytile[w][r][0]), ha = (uint32_t)(uintptr_t)as_lds(&htile[w][r][0]);
        const uint32_t sa = (uint32_t)(uintptr_t)as_lds(&relay[r][0]);
        float o0, o1, o2, o3;
        asm volatile(
            "s_setprio 3\n"
            "v_mov_b32 v224, %[h2]\n v_mov_b32 v225, %[h1]\n"
            "s_mov_b32 s20, 0xbffda16a\n s_mov_b32 s21, 0x3f7b4df5\n s_mov_b32 s22, 0\n"
            "Lstep%=:\n"
            "s_sub_u32 s24, s22, %[w]\n s_and_b32 s24, s24, 1\n s_cmp_eq_u32 s24, 0\n s_cbranch_scc0 Lidle%=\n"
            // ---- the chain over tile i - 2
            "s_cmp_lt_u32 s22, 2\n s_cbranch_scc1 Lsync%=\n s_sub_u32 s25, s22, 2\n s_cmp_ge_u32 s25, %[nt]\n s_cbranch_scc1 Lsync%=\n"
            "s_cmp_eq_u32 s25, 0\n s_cbranch_scc1 Lgo%=\n"
            "ds_read_b64 v[224:225], %[sa]\n"
            "Lgo%=:\n"
            "s_waitcnt lgkmcnt(0)\n"
            "v_pk_mul_f32 v[226:227], v[224:225], s[20:21] op_sel:[0,0] op_sel_hi:[0,1]\n"     // (a1 h[-2], a2 h[-2])
            "v_pk_mul_f32 v[228:229], v[224:225], s[20:21] op_sel:[1,0] op_sel_hi:[1,1]\n"     // (a1 h[-1], a2 h[-1])
            ".set m17k, 0\n .rept 32\n" LR_PAIR("96", "97") LR_PAIR("98", "99") ".set m17k, m17k+1\n .endr\n"
            "ds_write_b64 %[sa], v[222:223]\n"
            "s_branch Lsync%=\n"
            // ---- tile i - 3 out, tile i - 1 in
            "Lidle%=:\n"
            "s_cmp_lt_u32 s22, 3\n s_cbranch_scc1 Lpre%=\n s_sub_u32 s25, s22, 3\n s_cmp_ge_u32 s25, %[nt]\n s_cbranch_scc1 Lpre%=\n"
            "ds_write_b64 %[ha], v[224:225] offset:8\n"
            ".set m17k, 0\n .rept 32\n ds_write_b128 %[ha], v[96+4*m17k:99+4*m17k] offset:16+16*m17k\n .set m17k, m17k+1\n .endr\n"
            "Lpre%=:\n"
            "s_cmp_lt_u32 s22, 1\n s_cbranch_scc1 Lsync%=\n s_sub_u32 s25, s22, 1\n s_cmp_ge_u32 s25, %[nt]\n s_cbranch_scc1 Lsync%=\n"
            ".set m17k, 0\n .rept 32\n ds_read_b128 v[96+4*m17k:99+4*m17k], %[ya] offset:16*m17k\n .set m17k, m17k+1\n .endr\n"
            "Lsync%=:\n"
            "s_waitcnt lgkmcnt(0)\n s_barrier\n"
            "s_add_u32 s22, s22, 1\n s_cmp_lt_u32 s22, %[ni]\n s_cbranch_scc1 Lstep%=\n"
            "v_mov_b32 %[o0], v220\n v_mov_b32 %[o1], v221\n v_mov_b32 %[o2], v222\n v_mov_b32 %[o3], v223\n"
            : [o0] "=v"(o0), [o1] "=v"(o1), [o2] "=v"(o2), [o3] "=v"(o3)
            : [h2] "v"(h2), [h1] "v"(h1), [ya] "v"(ya), [ha] "v"(ha), [sa] "v"(sa), [w] "s"(w), [nt] "s"(NT), [ni] "s"(NI)
            : LR_CLOBBERS);
        if (state_out && ((NT - 1u) & 1u) == w && lane < LR_CH && c0 + r < C)
            *reinterpret_cast<v4f*>(state_out + 4 * (size_t)(c0 + r)) = v4f{o0, o1, o2, o3};
    } else {
        for (uint32_t i = 0; i < NI; ++i) {
            if (i >= 4u && i < NT + 4u) {
                const uint32_t tile = i - 4u, b = tile & 1u;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const uint32_t r = 8u * (uint32_t)(role - 3) + 2u * (uint32_t)q + sub;
                    const float* hrow = &htile[b][r][0];
                    const float4 p = *reinterpret_cast<const float4*>(hrow + 4u * col);        // h[k0-4 .. k0-1]
                    const float4 h = *reinterpret_cast<const float4*>(hrow + 4u + 4u * col);   // h[k0 .. k0+3]
                    float4 o;
                    o.x = iir_output(h.x, p.w, p.z);
                    o.y = iir_output(h.y, h.x, p.w);
                    o.z = iir_output(h.z, h.y, h.x);
                    o.w = iir_output(h.w, h.z, h.y);
                    if (c0 + r < C) *reinterpret_cast<float4*>(limit + (size_t)(c0 + r) * lpitch + (size_t)tile * LR_TILE + 4u * col) = o;
                }
            }
            dp_handover();
        }
    }
}

}  // namespace m17
