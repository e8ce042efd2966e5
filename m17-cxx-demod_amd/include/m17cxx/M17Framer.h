// mobilinkd::M17Framer — collects the soft bits of one frame (reference include/m17cxx/M17Framer.h:12-60): two per symbol,
// N = 368 per frame; the call that completes a frame returns N, every other call 0.  `result` is accepted for signature
// compatibility and, as in the reference, not written.
#pragma once

#include <array>
#include <cstddef>
#include <cstdint>
#include <tuple>

namespace mobilinkd
{

template <size_t N = 368>
struct M17Framer
{
    using buffer_t = std::array<int8_t, N>;

    alignas(16) buffer_t buffer_;
    size_t index_ = 0;

    M17Framer() { reset(); }

    static constexpr size_t size() { return N; }

    /// hard decision: dibit -> +-1 per bit
    size_t operator()(int dibit, int8_t** result)
    {
        (void)result;
        return push((dibit & 2) ? 1 : -1, (dibit & 1) ? 1 : -1);
    }

    /// soft decision: the LLR pair of llr<>()
    size_t operator()(std::tuple<int8_t, int8_t> symbol, int8_t** result)
    {
        (void)result;
        return push(std::get<0>(symbol), std::get<1>(symbol));
    }

    void reset()
    {
        buffer_.fill(0);
        index_ = 0;
    }

private:
    size_t push(int8_t first, int8_t second)
    {
        buffer_[index_] = first;
        buffer_[index_ + 1] = second;
        index_ += 2;
        if (index_ != N) return 0;
        index_ = 0;
        return N;
    }
};

} // mobilinkd
