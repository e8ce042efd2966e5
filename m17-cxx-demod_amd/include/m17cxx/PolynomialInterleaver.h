// mobilinkd::PolynomialInterleaver — the M17 quadratic permutation polynomial interleaver (reference
// include/m17cxx/PolynomialInterleaver.h:14-72): position i of the coded frame travels at (45 i + 92 i^2) mod 368.  The
// permutation is tabulated once at compile time; the four entry points of the reference (soft bits and packed bytes, both
// directions) are scatter / gather through that table.
#pragma once

#include "Util.h"

#include <array>
#include <cstddef>
#include <cstdint>

namespace mobilinkd
{

template <size_t F1 = 45, size_t F2 = 92, size_t K = 368>
struct PolynomialInterleaver
{
    using buffer_t = std::array<int8_t, K>;
    using bytes_t = std::array<uint8_t, K / 8>;

    static constexpr std::array<uint16_t, K> make_table()
    {
        std::array<uint16_t, K> t{};
        for (size_t i = 0; i != K; ++i) t[i] = uint16_t((F1 * i + F2 * i * i) % K);
        return t;
    }
    static constexpr std::array<uint16_t, K> table = make_table();

    size_t index(size_t i) { return table[i]; }

    void interleave(buffer_t& data)
    {
        buffer_t out{};
        for (size_t i = 0; i != K; ++i) out[table[i]] = data[i];
        data = out;
    }

    void deinterleave(buffer_t& frame)
    {
        buffer_t out{};
        for (size_t i = 0; i != K; ++i) out[i] = frame[table[i]];
        frame = out;
    }

    void interleave(bytes_t& data)
    {
        bytes_t out{};
        for (size_t i = 0; i != K; ++i) assign_bit_index(out, table[i], get_bit_index(data, i));
        data = out;
    }

    void deinterleave(bytes_t& data)
    {
        bytes_t out{};
        for (size_t i = 0; i != K; ++i) assign_bit_index(out, i, get_bit_index(data, table[i]));
        data = out;
    }
};

} // mobilinkd
