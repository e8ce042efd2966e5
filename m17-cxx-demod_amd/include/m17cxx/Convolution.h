// Convolutional-coder primitives of the reference (include/m17cxx/Convolution.h:11-23).  convolve_bit: one coded bit is the
// parity of the register taps a generator polynomial selects.  update_memory<K, k>: shift k input bits into a register that
// keeps K + 1 bits.  Both are the functions the Viterbi branch table (Viterbi.h, kernel K6) is built from.
#pragma once

#include <cstddef>
#include <cstdint>

namespace mobilinkd
{

namespace detail
{
constexpr uint32_t parity32(uint32_t v)
{
    v ^= v >> 16; v ^= v >> 8; v ^= v >> 4;
    return (0x6996u >> (v & 15u)) & 1u;
}
template <size_t BITS> constexpr uint32_t low_mask = BITS >= 32 ? ~0u : ((uint32_t(1) << BITS) - 1u);
} // detail

inline constexpr uint32_t convolve_bit(uint32_t poly, uint32_t memory) { return detail::parity32(poly & memory); }

template <size_t K, size_t k = 1>
inline constexpr uint32_t update_memory(uint32_t memory, uint32_t input)
{
    const uint32_t shifted = (memory << k) | input;
    return shifted & detail::low_mask<K + 1>;
}

} // mobilinkd
