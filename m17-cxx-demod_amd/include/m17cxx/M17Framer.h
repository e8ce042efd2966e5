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
        return hand_out(push((dibit & 2) ? 1 : -1, (dibit & 1) ? 1 : -1), result);
    }

    /// soft decision: the LLR pair of llr<>()
    size_t operator()(std::tuple<int8_t, int8_t> symbol, int8_t** result)
    {
        return hand_out(push(std::get<0>(symbol), std::get<1>(symbol)), result);
    }

    void reset()
    {
        buffer_.fill(0);
        index_ = 0;
    }

private:
    // a full frame is handed out as a pointer into the framer's own buffer (valid until the next symbol), as the reference does
    size_t hand_out(size_t n, int8_t** result)
    {
        if (n != 0 && result) *result = buffer_.data();
        return n;
    }
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
