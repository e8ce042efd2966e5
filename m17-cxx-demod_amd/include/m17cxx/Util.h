// Helpers of the reference's include/m17cxx/Util.h on this framework's terms: the soft-decision slicer llr<> (:128-145 with
// the table of :63-104), (de)puncturing (:147-212), bit / byte packing (:214-318) and the PRBS9 generator / receiver of the
// BERT mode (:320-413).  The slicer for the demodulator's case (float, 4-bit LLRs) is core::llr_slice — the function
// kernel K5 and m17hip_slice_llr run; batched form: llr(batched::Device&, ...).
#pragma once

#include "detail/batched.h"
#include "detail/core.h"

#include <algorithm>
#include <array>
#include <bitset>
#include <cassert>
#include <cstdint>
#include <cstdlib>
#include <limits>
#include <tuple>
#include <type_traits>
#include <utility>

namespace mobilinkd
{

namespace detail
{

/// Largest LLR magnitude for an N-bit LLR.
template <size_t N>
constexpr size_t llr_limit() { return (size_t(1) << (N - 1)) - 1; }

/// Rows of the slicer table: llr_limit() steps in each of the six unit intervals between -3 and +3, plus one.
template <size_t N>
constexpr size_t llr_size() { return llr_limit<N>() * 6 + 1; }

// The slicer table: row r = (upper edge, (llr of bit 1, llr of bit 0)).  Edges start at -3 + 1/limit and are ACCUMULATED in
// FloatType (the float rounding of the running sum decides which row a sample near an edge falls into).  Walking
// up from -3: below -1 the second LLR falls from +limit to -limit (skipping 0), between -1 and +1 the first one does, above
// +1 the second one climbs back.
template <typename FloatType, size_t LLR>
struct LlrTable {
    static constexpr size_t ROWS = llr_size<LLR>();
    std::array<FloatType, ROWS> edge;
    std::array<std::tuple<int8_t, int8_t>, ROWS> pair;
    LlrTable()
    {
        const int8_t top = int8_t(llr_limit<LLR>());
        const FloatType step = 1.0 / FloatType(top);
        auto down = [top](int8_t v) { v = int8_t(v - 1); if (v == 0) v = -1; return v < -top ? int8_t(-top) : v; };
        auto up = [top](int8_t v) { v = int8_t(v + 1); if (v == 0) v = 1; return v > top ? top : v; };
        int8_t first = top, second = top;
        FloatType e = -3.0 + step;
        for (size_t r = 0; r < ROWS; ++r) {
            edge[r] = e;
            pair[r] = std::make_tuple(first, second);
            if (e + 1.0 < 0) second = down(second);
            else if (e - 1.0 < 0) first = down(first);
            else second = up(second);
            e += step;
        }
    }
    static const LlrTable& get() { static const LlrTable t; return t; }
};

} // detail

template <class... Bools>
constexpr auto make_bitset(Bools&&... bools)
{
    std::bitset<sizeof...(Bools)> result;
    size_t i = 0;
    ((result[i++] = bool(bools)), ...);
    return result;
}

/// 4-FSK symbol -> dibit (+1 -> 00, +3 -> 01, -1 -> 10, -3 -> 11); anything else is a programming error.
inline int from_4fsk(int symbol)
{
    switch (symbol) {
    case 1: return 0;
    case 3: return 1;
    case -1: return 2;
    case -3: return 3;
    default: abort();
    }
}

/// Soft decision of one normalised symbol: (LLR of the dibit's first bit, LLR of its second bit), LLR > 0 <=> bit 1.
template <typename FloatType, size_t LLR>
auto llr(FloatType sample)
{
    const auto& table = detail::LlrTable<FloatType, LLR>::get();
    if constexpr (std::is_same_v<FloatType, float> && LLR == 4) {
        const uint32_t p = core::llr_slice(sample, table.edge.data());
        return std::make_tuple(int8_t(p & 0xFF), int8_t(p >> 8));
    } else {
        const FloatType s = std::min(FloatType(3.0), std::max(FloatType(-3.0), sample));
        const auto it = std::lower_bound(table.edge.begin(), table.edge.end(), s);
        const size_t row = it == table.edge.end() ? table.ROWS - 1 : size_t(it - table.edge.begin());
        return table.pair[row];
    }
}

/// Batched form (GPU): rows x n normalised symbols -> llr_out[rows][n][2] and, per row, the running EVM after each symbol
/// (SymbolEvm after reset()): evm_out[rows][n].  Either output may be null.
inline int llr(batched::Device& dev, const float* symbols, uint32_t rows, uint32_t n, int8_t* llr_out, float* evm_out)
{
    return m17hip_slice_llr(dev.ctx(), symbols, rows, n, llr_out, evm_out);
}

/// Re-insert erasures (0) where the puncture matrix has a 0; returns a fresh array of M values.
template <size_t M, typename T, size_t N, typename U, size_t IN>
auto depunctured(std::array<T, N> puncture_matrix, std::array<U, IN> in)
{
    static_assert(M % N == 0);
    std::array<U, M> out;
    size_t taken = 0;
    for (size_t i = 0; i != M; ++i) out[i] = puncture_matrix[i % N] ? in[taken++] : U(0);
    return out;
}

/// Same into a caller's buffer; stops when the input runs out (positions beyond keep what they held); returns the erasures written.
template <size_t IN, size_t OUT, size_t P>
size_t depuncture(const std::array<int8_t, IN>& in, std::array<int8_t, OUT>& out, const std::array<int8_t, P>& p)
{
    size_t taken = 0, erased = 0, phase = 0;
    for (size_t i = 0; i != OUT && taken < IN; ++i) {
        if (p[phase]) out[i] = in[taken++];
        else { out[i] = 0; ++erased; }
        if (++phase == P) phase = 0;
    }
    return erased;
}

/// Drop the positions where the puncture matrix has a 0; returns the values kept.
template <typename T, size_t IN, typename U, size_t OUT, size_t P>
size_t puncture(const std::array<T, IN>& in, std::array<U, OUT>& out, const std::array<int8_t, P>& p)
{
    size_t kept = 0, phase = 0;
    for (size_t i = 0; i != IN && kept != OUT; ++i) {
        if (p[phase]) out[kept++] = in[i];
        if (++phase == P) phase = 0;
    }
    return kept;
}

// ---- bit addressing in byte arrays, MSB first ----------------------------------------------------------------------------
template <size_t N>
constexpr bool get_bit_index(const std::array<uint8_t, N>& input, size_t index)
{
    assert((index >> 3) < N);
    return (input[index >> 3] >> (7 - (index & 7))) & 1;
}

template <size_t N>
void set_bit_index(std::array<uint8_t, N>& input, size_t index)
{
    assert((index >> 3) < N);
    input[index >> 3] |= uint8_t(0x80u >> (index & 7));
}

template <size_t N>
void reset_bit_index(std::array<uint8_t, N>& input, size_t index)
{
    assert((index >> 3) < N);
    input[index >> 3] &= uint8_t(~(0x80u >> (index & 7)));
}

template <size_t N>
void assign_bit_index(std::array<uint8_t, N>& input, size_t index, bool value)
{
    value ? set_bit_index(input, index) : reset_bit_index(input, index);
}

template <size_t IN, size_t OUT, size_t P>
size_t puncture_bytes(const std::array<uint8_t, IN>& in, std::array<uint8_t, OUT>& out, const std::array<int8_t, P>& p)
{
    size_t kept = 0, phase = 0;
    for (size_t i = 0; i != IN * 8 && kept != OUT * 8; ++i) {
        if (p[phase]) assign_bit_index(out, kept++, get_bit_index(in, i));
        if (++phase == P) phase = 0;
    }
    return kept;
}

/// Sign-extend the low n bits of v into T.
template <typename T, size_t n>
constexpr T to_int(uint8_t v)
{
    const unsigned low = v & ((1u << n) - 1u);
    return (low >> (n - 1)) ? T(int(low) - int(1u << n)) : T(low);
}

/// Pack bits (one per array element, MSB first) into bytes.
template <typename T, size_t N>
constexpr auto to_byte_array(std::array<T, N> in)
{
    std::array<uint8_t, (N + 7) / 8> out{};
    for (size_t i = 0; i != N; ++i) out[i >> 3] |= uint8_t(in[i] << (7 - (i & 7)));
    return out;
}

template <typename T, size_t N>
constexpr void to_byte_array(std::array<T, N> in, std::array<uint8_t, (N + 7) / 8>& out)
{
    for (size_t byte = 0; byte != out.size(); ++byte) {
        uint8_t v = 0;
        for (size_t b = 0; b != 8 && byte * 8 + b < N; ++b) v |= uint8_t(in[byte * 8 + b] << (7 - b));
        out[byte] = v;
    }
}

/// PRBS9 (x^9 + x^5 + 1) generator and self-synchronising receiver of the BERT mode: locks after 18 consecutive good bits,
/// unlocks when 25 of the last 128 validated bits were wrong.  (The consumer over decoded BERT frames runs on the GPU as
/// m17hip_bert_stats; this is the host-side object apps/m17-demod.cpp holds.)
struct PRBS9
{
    static constexpr uint16_t MASK = 0x1FF;
    static constexpr uint8_t TAP_1 = 8;         // bit 9
    static constexpr uint8_t TAP_2 = 4;         // bit 5
    static constexpr uint8_t LOCK_COUNT = 18;
    static constexpr uint8_t UNLOCK_COUNT = 25;

    uint16_t state = 1;
    bool synced = false;
    uint8_t sync_count = 0;
    uint32_t bit_count = 0;
    uint32_t err_count = 0;
    std::array<uint8_t, 16> history{};   // error flags of the last 128 validated bits
    size_t hist_count = 0;
    size_t hist_pos = 0;

    void count_errors(bool error)
    {
        uint8_t& cell = history[hist_pos >> 3];
        const uint8_t flag = uint8_t(1u << (hist_pos & 7));
        ++bit_count;
        if (cell & flag) --hist_count;     // the bit that leaves the window
        if (error) {
            cell |= flag;
            ++err_count;
            if (++hist_count >= UNLOCK_COUNT) synced = false;
        } else {
            cell &= uint8_t(~flag);
        }
        hist_pos = (hist_pos + 1) & 127;
    }

    bool generate()
    {
        const bool out = ((state >> TAP_1) ^ (state >> TAP_2)) & 1;
        state = ((state << 1) | out) & MASK;
        return out;
    }

    // feed the received bit through the register; result 0 = it was what the register predicted
    bool synchronize(bool bit)
    {
        const bool mismatch = (bit ^ (state >> TAP_1) ^ (state >> TAP_2)) & 1;
        state = ((state << 1) | bit) & MASK;
        if (mismatch) {
            sync_count = 0;
        } else if (++sync_count == LOCK_COUNT) {
            synced = true;
            bit_count += LOCK_COUNT;
            history.fill(0);
            hist_count = hist_pos = 0;
            sync_count = 0;
        }
        return mismatch;
    }

    bool validate(bool bit)
    {
        if (!synced) return synchronize(bit);
        const bool mismatch = bit ^ generate();   // free-running once locked
        count_errors(mismatch);
        return mismatch;
    }

    bool sync() const { return synced; }
    uint32_t errors() const { assert(synced); return err_count; }
    uint32_t bits() const { assert(synced); return bit_count; }

    void reset()
    {
        state = 1;
        synced = false;
        sync_count = 0;
        bit_count = err_count = 0;
        history.fill(0);
        hist_count = hist_pos = 0;
    }
};

} // mobilinkd
