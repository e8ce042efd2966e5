// Shared definitions for the HIP kernels of the M17 demodulation hot path (gfx950 / CDNA4).
// Product code: nothing here includes, links or calls anything under oracle/.
//
// Numerics contract: every kernel is compiled with -ffp-contract=off, no fast-math,
// IEEE divide/sqrt, denormals preserved — the reference is a baseline x86-64 build where every
// fp32 multiply and add rounds separately, and decoded bits must be bit-exact.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

// the arithmetic cores shared with the scalar operator classes of include/m17cxx (one definition of every rounding step)
#include "m17cxx/detail/core.h"

namespace m17 {
namespace core = ::mobilinkd::core;

// ---- slab geometry ---------------------------------------------------------------------------
// xbuf row:  [XPRE samples carried from the previous run | T new samples]   (int16)
// ybuf row:  [YPRE samples carried from the previous run | T new samples]   (float)
constexpr int XPRE = 152;  // >= 148 (FIR history), >= 120 (sliding-DFT delay line), >= 149 (history snapshot); multiple of 8
constexpr int YPRE = 96;   // >= 80 (correlator ring); multiple of 4
constexpr int TICK = 192;  // gcd(384, 960): every DCD update point is a tick boundary (M17Demodulator.h:677,742)
constexpr int NTAPS = 149; // taps[149] == 0.0 contributes a signed zero only

// ---- RRC taps: reference M17Demodulator.h:79-118 (alpha = 0.5, 10 samples/symbol), symmetric: core::rrc_tap -----
using core::rrc_tap;

// apps/m17-demod.cpp:486-489: x = float(double(s) / 41067.0) (optionally s *= -1 first, in int16).
// (float)s / 41067.0f is bit-identical for all 65536 inputs (no double rounding: 41067 is odd and < 2^16, so s / 41067 is never
// within 2^-40 of a float midpoint), and so is one Newton step on q = s * RN(1/41067): r = fma(-q, 41067, s) is the exact
// remainder, fma(r, 1/41067, q) the correctly rounded quotient — 3 instructions instead of the 10 of an IEEE division.
// Exhaustive: tests/test_oracle_kat.py::test_scale_identities_exhaustive (host), tests/test_gpu_parity.py::test_scale_exhaustive.
__device__ __forceinline__ float scale_sample(int s, bool invert) { return core::scale_i16(s, invert); }

// Sync words M17Demodulator.h:154-157: preamble, LSF(/stream), packet(/BERT), EOT — symbol signs (x3).
// The same words as sign masks (bit i set = symbol i is -3): (float)(-3) * x == -(3.0f * x) exactly, and r + (-p) is what
// r - p computes, so a correlation is eight multiplies by the literal 3.0f and eight adds / subtracts — no coefficient table
// to keep in registers (the sequential kernel used to spill the 32 converted coefficients to scratch and reload them at every use).
using core::SYNC_NEG;
// Correlator::correlate (Correlator.h:51-64) for word w over r[0..7] (oldest symbol first)
__device__ __forceinline__ float sync_correlate(int w, const float (&r)[8])
{
    const uint32_t neg = w == 0 ? SYNC_NEG[0] : (w == 1 ? SYNC_NEG[1] : (w == 2 ? SYNC_NEG[2] : SYNC_NEG[3]));
    return core::correlate_mask(neg, r);
}
__device__ __constant__ const int8_t SYNC_WORDS[4][8] = {{+3, -3, +3, -3, +3, -3, +3, -3},
                                                          {+3, +3, +3, +3, -3, -3, +3, -3},
                                                          {+3, -3, +3, +3, -3, -3, -3, -3},
                                                          {+3, +3, +3, +3, +3, +3, -3, +3}};

// ---- frame record / diag: byte layout of include/m17hip.h -------------------------------------------
struct FrameRec {
    uint32_t channel, seq;
    uint64_t sample_pos;
    int32_t cost;
    uint8_t frame_type, sync_type, len, flags;
    uint8_t payload[32];
    uint8_t pad[8];
};
static_assert(sizeof(FrameRec) == 64, "FrameRec layout");

struct Diag {
    int32_t dcd;
    float evm, deviation, offset;
    int32_t locked;
    float clock;
    int32_t sample_index, sync_index, clock_index, viterbi_cost;
    float dcd_level;
    uint32_t n_diag, demod_state, n_frames, pad[2];
};
static_assert(sizeof(Diag) == 64, "Diag layout");

// ---- per-channel DCD recurrence state carried between runs (K3) -------------------------------------
struct DcdState {
    float xr[2], xi[2];  // sliding-DFT bins
    float acc[6][2];     // acc[j]: sums since the start of the last tick a with a % 5 == j; acc[5]: since reset
};

}  // namespace m17
