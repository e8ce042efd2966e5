// The M17 puncture matrices and the trellis descriptor of the reference (include/m17cxx/Trellis.h:17-130).
//   P1: link setup frame, 61-periodic, every 4th position from the third one punctured (488 -> 368)
//   P2: stream / BERT frames, rate 6/11 matrix of 12 with the last position punctured (296 -> 272, 402 -> 368)
//   P3: packet frames, 8 with the last position punctured (420 -> 368)
#pragma once

#include "Convolution.h"
#include "Util.h"

#include <array>
#include <cstdint>
#include <cstdlib>

namespace mobilinkd
{

inline constexpr std::array<int8_t, 61> make_p1()
{
    std::array<int8_t, 61> m{};
    for (size_t i = 0; i != m.size(); ++i) m[i] = (i % 4 == 2) ? 0 : 1;   // zeros at 2, 6, ..., 58
    return m;
}

inline constexpr auto P1 = make_p1();
inline constexpr auto P2 = std::array<int8_t, 12>{1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 0};
inline constexpr auto P3 = std::array<int8_t, 8>{1, 1, 1, 1, 1, 1, 1, 0};

/// value -> N bits, most significant first
template <size_t N>
constexpr std::array<uint8_t, N> toBitArray(int8_t value)
{
    std::array<uint8_t, N> bits{};
    for (size_t i = 0; i != N; ++i) bits[i] = (value >> (N - 1 - i)) & 1;
    return bits;
}

/// Descriptor of a rate 1/n convolutional code with K memory bits (2^K states).
template <size_t K_, size_t n_>
struct Trellis
{
    static constexpr size_t K = K_;
    static constexpr size_t k = 1;
    static constexpr size_t n = n_;
    static constexpr size_t NumStates = (1 << K);

    using polynomials_t = std::array<uint32_t, n_>;

    polynomials_t polynomials;

    constexpr Trellis(polynomials_t polys) : polynomials(polys) {}
};

template <size_t K, size_t n>
constexpr Trellis<K, n> makeTrellis(std::array<uint32_t, n> polys)
{
    return Trellis<K, n>(polys);
}

} // mobilinkd
