// mobilinkd::Golay24 — the extended Golay (24,12) code of the LICH (reference include/m17cxx/Golay24.h:14-224): generator
// polynomial 0xC75, systematic 23-bit codeword = data << 11 | checks, plus one overall parity bit in the LSB.
// decode(): the syndrome of the 23-bit word selects an error pattern of weight <= 3 from a 2048-entry table (the code is
// perfect: every syndrome has exactly one), the corrected word is accepted when fewer than 3 syndrome bits were set or its
// overall parity is even.  The same table, indexed by syndrome, is what the frame decoder on the GPU uses (csrc/m17hip.hip
// build_tables -> golay_fix).
#pragma once

#include <array>
#include <cstddef>
#include <cstdint>

namespace mobilinkd {

namespace Golay24
{

constexpr uint16_t POLY = 0xC75;

/// remainder of the 23-bit word, left-aligned the way the reference returns it (syndrome << 12)
constexpr uint32_t syndrome(uint32_t codeword)
{
    codeword &= 0xFFFFFFu;
    for (int i = 0; i != 12; ++i) {
        if (codeword & 1u) codeword ^= POLY;
        codeword >>= 1;
    }
    return codeword << 12;
}

constexpr bool parity(uint32_t codeword) { return __builtin_popcount(codeword) & 1; }

constexpr uint32_t encode23(uint16_t data)
{
    uint32_t checks = data;
    for (int i = 0; i != 12; ++i) {
        if (checks & 1u) checks ^= POLY;
        checks >>= 1;
    }
    return checks | (uint32_t(data) << 11);
}

constexpr uint32_t encode24(uint16_t data)
{
    const uint32_t cw = encode23(data);
    return (cw << 1) | uint32_t(parity(cw));
}

namespace detail
{
/// error pattern (23 bits) for each 11-bit syndrome
struct CorrectionTable {
    std::array<uint32_t, 2048> pattern{};
    constexpr CorrectionTable()
    {
        auto put = [this](uint32_t e) { pattern[(syndrome(e) >> 12) & 0x7FF] = e; };
        put(0);
        for (int a = 0; a < 23; ++a) {
            put(1u << a);
            for (int b = a + 1; b < 23; ++b) {
                put((1u << a) | (1u << b));
                for (int c = b + 1; c < 23; ++c) put((1u << a) | (1u << b) | (1u << c));
            }
        }
    }
};
inline constexpr CorrectionTable CORRECTIONS{};
} // detail

inline bool decode(uint32_t input, uint32_t& output)
{
    const uint32_t syn = syndrome(input >> 1);
    const uint32_t pattern = detail::CORRECTIONS.pattern[(syn >> 12) & 0x7FF];
    output = input ^ (pattern << 1);
    return __builtin_popcount(syn) < 3 || !parity(output);
}

} // Golay24

} // mobilinkd
