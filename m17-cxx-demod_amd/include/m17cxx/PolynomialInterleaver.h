// mobilinkd::PolynomialInterleaver — the M17 quadratic permutation polynomial interleaver (reference
// include/m17cxx/PolynomialInterleaver.h:14-72): position i of the coded frame travels at (45 i + 92 i^2) mod 368.
#pragma once

#include "Util.h"

#include <algorithm>
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

    alignas(16) buffer_t buffer_;

    size_t index(size_t i) { return (F1 * i + F2 * i * i) % K; }

    void interleave(buffer_t& data)
    {
        buffer_.fill(0);
        for (size_t i = 0; i != K; ++i) buffer_[index(i)] = data[i];
        data = buffer_;
    }

    void interleave(bytes_t& data)
    {
        bytes_t shuffled{};
        for (size_t i = 0; i != K; ++i) assign_bit_index(shuffled, index(i), get_bit_index(data, i));
        data = shuffled;
    }

    void deinterleave(buffer_t& frame)
    {
        for (size_t i = 0; i != K; ++i) buffer_[i] = frame[index(i)];
        frame = buffer_;
    }

    void deinterleave(bytes_t& data)
    {
        bytes_t restored{};
        for (size_t i = 0; i != K; ++i) assign_bit_index(restored, i, get_bit_index(data, index(i)));
        data = restored;
    }
};

} // mobilinkd
