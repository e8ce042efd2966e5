// mobilinkd::M17Randomizer — the M17 decorrelation sequence (reference include/m17cxx/M17Randomizer.h:12-77; the 46 bytes are
// the M17 specification's): soft bits are multiplied by +-1, hard bits XORed.  On the GPU the sequence is folded into the
// frame decoder's source map together with the interleaver and the puncture matrices (csrc/m17hip.hip frame_source).
#pragma once

#include <array>
#include <cstddef>
#include <cstdint>

namespace mobilinkd
{

namespace detail
{
inline auto DC = std::array<uint8_t, 46>{
    0xd6, 0xb5, 0xe2, 0x30, 0x82, 0xFF, 0x84, 0x62, 0xba, 0x4e, 0x96, 0x90, 0xd8, 0x98, 0xdd, 0x5d, 0x0c, 0xc8, 0x52, 0x43, 0x91, 0x1d, 0xf8,
    0x6e, 0x68, 0x2F, 0x35, 0xda, 0x14, 0xea, 0xcd, 0x76, 0x19, 0x8d, 0xd5, 0x80, 0xd1, 0x33, 0x87, 0x13, 0x57, 0x18, 0x2d, 0x29, 0x78, 0xc3};
}

template <size_t N = 368>
struct M17Randomizer
{
    std::array<int8_t, N> dc_;   // +1 where the sequence bit is 0, -1 where it is 1

    M17Randomizer()
    {
        for (size_t i = 0; i != N; ++i) dc_[i] = ((detail::DC[i >> 3] >> (7 - (i & 7))) & 1) ? -1 : 1;
    }

    /// soft bits: flip the sign where the sequence is 1 (its own inverse)
    void operator()(std::array<int8_t, N>& frame)
    {
        for (size_t i = 0; i != N; ++i) frame[i] *= dc_[i];
    }

    /// hard bits (0 / 1)
    void randomize(std::array<int8_t, N>& frame)
    {
        for (size_t i = 0; i != N; ++i) frame[i] ^= (dc_[i] < 0);
    }
};

template <size_t N = 46>
struct M17ByteRandomizer
{
    void operator()(std::array<uint8_t, N>& frame)
    {
        for (size_t i = 0; i != N; ++i) frame[i] ^= detail::DC[i];
    }
};

} // mobilinkd
