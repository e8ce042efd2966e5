// Convolutional-coder primitives of the reference (include/m17cxx/Convolution.h:11-23): one output bit = parity of the taps
// selected by a polynomial; the shift register keeps K + 1 bits.
#pragma once

#include <cstddef>
#include <cstdint>

namespace mobilinkd
{

inline constexpr uint32_t convolve_bit(uint32_t poly, uint32_t memory)
{
    return uint32_t(__builtin_popcount(poly & memory) & 1);
}

template <size_t K, size_t k = 1>
inline constexpr uint32_t update_memory(uint32_t memory, uint32_t input)
{
    return ((memory << k) | input) & ((1u << (K + 1)) - 1u);
}

} // mobilinkd
