// mobilinkd::Viterbi — the reference's soft-decision Viterbi decoder (include/m17cxx/Viterbi.h:25-240) for rate 1/n trellises,
// as the M17 frame decoder uses it: K = 4 (16 states), n = 2, polynomials 031 / 027, 4-bit LLRs where 0 = erasure.
// Scalar form: decode<IN, OUT>() below, a plain restatement with the reference's decisions — branch cost |c - s| per received
// soft bit, skipped where erased; strict `>` compare so a tie keeps the path from the lower predecessor; the end state is the
// first minimum; cost = round(min / 7.0f); the last IN/2 - OUT steps (flush bits) are not emitted.
// Batched form: decode(batched::Device&, ...) = m17hip_viterbi (the DPP trellis of csrc/m17_decode_device.hpp on the GPU).
#pragma once

#include "Convolution.h"
#include "Trellis.h"
#include "Util.h"
#include "detail/batched.h"
#include "detail/core.h"

#include <array>
#include <bitset>
#include <cmath>
#include <cstddef>
#include <cstdint>
#include <limits>

namespace mobilinkd
{

/// next state of (state, input bit): the K low bits of the shifted register
template <typename Trellis_>
constexpr std::array<std::array<uint8_t, (1 << Trellis_::k)>, (1 << Trellis_::K)> makeNextState(Trellis_)
{
    constexpr size_t S = size_t(1) << Trellis_::K;
    std::array<std::array<uint8_t, (1 << Trellis_::k)>, S> table{};
    for (size_t s = 0; s != S; ++s)
        for (size_t in = 0; in != (size_t(1) << Trellis_::k); ++in)
            table[s][in] = uint8_t(((s << Trellis_::k) | in) & (S - 1));
    return table;
}

/// the two predecessors of every state: [0] from the lower half of the state space, [1] from the upper half
template <typename Trellis_>
constexpr std::array<std::array<uint8_t, (1 << Trellis_::k)>, (1 << Trellis_::K)> makePrevState(Trellis_)
{
    constexpr size_t S = size_t(1) << Trellis_::K;
    std::array<std::array<uint8_t, (1 << Trellis_::k)>, S> table{};
    for (size_t s = 0; s != S; ++s) {
        table[s][0] = uint8_t(s >> 1);
        table[s][1] = uint8_t((s >> 1) + S / 2);
    }
    return table;
}

/// expected soft value (+-(2^(LLR-1) - 1)) of each of the n coded bits when a 0 is shifted into `state`
template <typename Trellis_, size_t LLR = 2>
constexpr auto makeCost(Trellis_ trellis)
{
    constexpr size_t S = size_t(1) << Trellis_::K;
    constexpr int16_t mag = int16_t((1 << (LLR - 1)) - 1);
    std::array<std::array<int16_t, Trellis_::n>, S> table{};
    for (uint32_t s = 0; s != S; ++s)
        for (uint32_t j = 0; j != Trellis_::n; ++j)
            table[s][j] = convolve_bit(trellis.polynomials[j], s << 1) ? mag : int16_t(-mag);
    return table;
}

template <typename Trellis_, size_t LLR_ = 2>
struct Viterbi
{
    static_assert(LLR_ < 7);    // keeps the path metrics far from overflow

    static constexpr size_t K = Trellis_::K;
    static constexpr size_t k = Trellis_::k;
    static constexpr size_t n = Trellis_::n;
    static constexpr size_t InputValues = 1 << n;
    static constexpr size_t NumStates = (1 << K);
    static constexpr int32_t METRIC = ((1 << (LLR_ - 1)) - 1) << 2;

    using metrics_t = std::array<int32_t, NumStates>;
    using cost_t = std::array<std::array<int16_t, n>, NumStates>;
    using state_transition_t = std::array<std::array<uint8_t, 2>, NumStates>;

    metrics_t pathMetrics_{};
    cost_t cost_;
    state_transition_t nextState_;
    state_transition_t prevState_;

    metrics_t prevMetrics, currMetrics;

    // one decision bit per state and step; 244 steps is the longest M17 frame (LSF: 488 soft bits)
    std::array<std::bitset<NumStates>, 244> history_;

    Viterbi(Trellis_ trellis)
    : cost_(makeCost<Trellis_, LLR_>(trellis))
    , nextState_(makeNextState(trellis))
    , prevState_(makePrevState(trellis))
    {}

    /// One butterfly: predecessors j and j + NumStates/2 feed successors nextState_[j][0] and nextState_[j][1].
    void calculate_path_metric(const std::array<int16_t, NumStates / 2>& cost0, const std::array<int16_t, NumStates / 2>& cost1,
                               std::bitset<NumStates>& hist, size_t j)
    {
        const int32_t lo = prevMetrics[j], hi = prevMetrics[j + NumStates / 2];
        const uint8_t even = nextState_[j][0], odd = nextState_[j][1];
        const int32_t via_lo_even = lo + cost0[j], via_hi_even = hi + cost1[j];
        const int32_t via_lo_odd = lo + cost1[j], via_hi_odd = hi + cost0[j];
        const bool take_hi_even = via_lo_even > via_hi_even;   // strict: a tie keeps the lower predecessor
        const bool take_hi_odd = via_lo_odd > via_hi_odd;
        hist.set(even, take_hi_even);
        hist.set(odd, take_hi_odd);
        currMetrics[even] = take_hi_even ? via_hi_even : via_lo_even;
        currMetrics[odd] = take_hi_odd ? via_hi_odd : via_lo_odd;
    }

    /// Decode IN soft bits (n per step, 0 = erased) into the first OUT message bits; returns round(path metric / (2^(LLR-1) - 1)).
    template <size_t IN, size_t OUT>
    size_t decode(std::array<int8_t, IN> const& in, std::array<uint8_t, OUT>& out)
    {
        static_assert(n == 2 && IN % 2 == 0 && IN / 2 <= 244, "rate 1/2, at most 244 trellis steps");
        constexpr size_t STEPS = IN / 2, HALF = NumStates / 2;

        prevMetrics.fill(std::numeric_limits<int32_t>::max() / 2);
        prevMetrics[0] = 0;   // the coder starts in state 0

        std::array<int16_t, HALF> cost0, cost1;
        for (size_t step = 0; step != STEPS; ++step) {
            const int16_t s0 = in[2 * step], s1 = in[2 * step + 1];
            for (size_t j = 0; j != HALF; ++j) {
                int16_t c0 = 0, c1 = 0;   // distance to the branch with coded bits (c, c') and to its complement
                if (s0) { c0 = int16_t(std::abs(cost_[j][0] - s0)); c1 = int16_t(std::abs(cost_[j][0] + s0)); }
                if (s1) { c0 = int16_t(c0 + std::abs(cost_[j][1] - s1)); c1 = int16_t(c1 + std::abs(cost_[j][1] + s1)); }
                cost0[j] = c0; cost1[j] = c1;
            }
            for (size_t j = 0; j != HALF; ++j) calculate_path_metric(cost0, cost1, history_[step], j);
            std::swap(currMetrics, prevMetrics);
        }

        size_t state = 0;
        int32_t best = prevMetrics[0];
        for (size_t s = 1; s != NumStates; ++s)
            if (prevMetrics[s] < best) { best = prevMetrics[s]; state = s; }   // first minimum wins

        const size_t cost = core::viterbi_cost_of<(1 << (LLR_ - 1)) - 1>(best);

        // chain back: the message bit of step t is the low bit of the state after it; the trailing flush steps are dropped
        for (size_t step = STEPS; step-- > 0;) {
            if (step < OUT) out[step] = state & 1;
            state = prevState_[state][history_[step][state]];
        }
        return cost;
    }

    /// Batched form (GPU) for the four M17 frame shapes (<488,240> LSF, <296,144> stream, <420,206> packet, <402,197> BERT):
    /// n_frames depunctured frames in, message bits and costs out.
    template <size_t IN, size_t OUT>
    static int decode(batched::Device& dev, const std::array<int8_t, IN>* in, size_t n_frames, std::array<uint8_t, OUT>* out, int32_t* cost)
    {
        static_assert(K == 4 && n == 2 && LLR_ == 4, "the batched decoder is the M17 one: Trellis<4,2>, 4-bit LLRs");
        constexpr int kind = (IN == 488 && OUT == 240) ? 0 : (IN == 296 && OUT == 144) ? 1 : (IN == 420 && OUT == 206) ? 2 : (IN == 402 && OUT == 197) ? 3 : -1;
        static_assert(kind >= 0, "not an M17 frame shape");
        return m17hip_viterbi(dev.ctx(), reinterpret_cast<const int8_t*>(in), (uint32_t)n_frames, kind, reinterpret_cast<uint8_t*>(out), cost);
    }
};

} // mobilinkd
