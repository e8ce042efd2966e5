// Host-side mirror of the reference's frame-callback types (reference include/m17cxx/M17FrameDecoder.h:40-104):
// same names, same enumerators, same buffer shapes, so that a frame handler written against the reference
// (apps/m17-demod.cpp:307-336 `handle_frame`) compiles unchanged.  The decode itself runs on the GPU
// (csrc/m17_decode_device.hpp) behind the C ABI in include/m17hip.h; this header holds types only.
#pragma once

#include <array>
#include <cstddef>
#include <cstdint>
#include <functional>

namespace mobilinkd
{

struct M17FrameDecoder
{
    enum class State { LSF, STREAM, BASIC_PACKET, FULL_PACKET, BERT };
    enum class SyncWordType { LSF, STREAM, PACKET, BERT };
    enum class DecodeResult { FAIL, OK, EOS, INCOMPLETE, PACKET_INCOMPLETE };
    enum class FrameType { LSF, LICH, STREAM, BASIC_PACKET, FULL_PACKET, BERT };

    using input_buffer_t = std::array<int8_t, 368>;
    using lsf_buffer_t = std::array<uint8_t, 30>;
    using lich_buffer_t = std::array<uint8_t, 6>;
    using audio_buffer_t = std::array<uint8_t, 18>;
    using packet_buffer_t = std::array<uint8_t, 26>;
    using bert_buffer_t = std::array<uint8_t, 25>;

    struct output_buffer_t {
        FrameType type;
        union {
            lich_buffer_t lich;
            audio_buffer_t stream;
            packet_buffer_t packet;
            bert_buffer_t bert;
        };
        lsf_buffer_t lsf;
    };

    // true = data good or unknown (only consulted for the last frame of a packet in the reference)
    using callback_t = std::function<bool(const output_buffer_t&, int)>;
};

} // mobilinkd
