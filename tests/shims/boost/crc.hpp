// Stand-in for <boost/crc.hpp> (build check only, see ../README.md): crc_optimal as apps/m17-demod.cpp:218-222 uses it.
#pragma once
#include <cstddef>
#include <cstdint>
namespace boost
{
template <std::size_t Bits, uint32_t Poly, uint32_t Init, uint32_t XorOut, bool ReflectIn, bool ReflectRem>
class crc_optimal
{
    static_assert(Bits == 16 && ReflectIn && ReflectRem, "only the reflected 16-bit case is needed");
    uint32_t rem_ = Init;
    static constexpr uint32_t reflected_poly()
    {
        uint32_t r = 0;
        for (std::size_t i = 0; i < Bits; ++i) r |= ((Poly >> i) & 1u) << (Bits - 1 - i);
        return r;
    }
public:
    void process_bytes(const void* data, std::size_t n)
    {
        const unsigned char* p = static_cast<const unsigned char*>(data);
        for (std::size_t k = 0; k < n; ++k) {
            rem_ ^= p[k];
            for (int b = 0; b < 8; ++b) rem_ = (rem_ & 1u) ? (rem_ >> 1) ^ reflected_poly() : rem_ >> 1;
        }
    }
    uint32_t checksum() const { return (rem_ ^ XorOut) & 0xFFFFu; }
};
}
