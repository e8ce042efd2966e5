// mobilinkd::LinkSetupFrame — callsign coding of the M17 link setup frame (reference include/m17cxx/LinkSetupFrame.h:14-130):
// up to 9 characters of the 40-symbol alphabet " A-Z0-9-/." as a base-40 number, first character least significant, stored
// big-endian in 6 bytes; six 0xFF bytes are the broadcast address.  (Batched on the GPU: m17hip_lsf_info.)
#pragma once

#include <algorithm>
#include <array>
#include <cstdint>
#include <stdexcept>
#include <string_view>

namespace mobilinkd
{

struct LinkSetupFrame
{
    using call_t = std::array<char, 10>;             // NUL-terminated C-string
    using encoded_call_t = std::array<uint8_t, 6>;
    using frame_t = std::array<uint8_t, 30>;
    using nonce_t = std::string_view;

    static constexpr encoded_call_t BROADCAST_ADDRESS = {0xff, 0xff, 0xff, 0xff, 0xff, 0xff};
    static constexpr call_t BROADCAST_CALL = {'B', 'R', 'O', 'A', 'D', 'C', 'A', 'S', 'T', 0};

    enum TxType { PACKET, STREAM };
    enum DataType { DT_RESERVED, DATA, VOICE, MIXED };
    enum EncType { NONE, AES, LFSR, ET_RESERVED };

    call_t tocall_ = {0};   // destination
    call_t mycall_ = {0};   // source
    TxType tx_type_ = TxType::STREAM;
    DataType data_type_ = DataType::VOICE;
    EncType encryption_type_ = EncType::NONE;

    /// value of one character in the alphabet, or -1
    static int symbol_value(char c)
    {
        if (c >= 'A' && c <= 'Z') return c - 'A' + 1;
        if (c >= '0' && c <= '9') return c - '0' + 27;
        if (c == '-') return 37;
        if (c == '/') return 38;
        if (c == '.') return 39;
        return -1;
    }

    /// Unmappable characters (NUL padding included) count as 0 unless `strict`, which throws std::invalid_argument.
    static encoded_call_t encode_callsign(call_t callsign, bool strict = false)
    {
        uint64_t value = 0;
        for (size_t i = callsign.size(); i-- > 0;) {   // last character is the most significant digit
            const int v = symbol_value(callsign[i]);
            if (v < 0 && strict) throw std::invalid_argument("bad callsign");
            value = value * 40 + uint64_t(v < 0 ? 0 : v);
        }
        encoded_call_t out;
        for (size_t i = 0; i != out.size(); ++i) out[i] = uint8_t(value >> (8 * (5 - i)));
        return out;
    }

    static call_t decode_callsign(encoded_call_t callsign, bool strict = false)
    {
        (void)strict;
        if (callsign == BROADCAST_ADDRESS) return BROADCAST_CALL;
        static const char alphabet[] = "xABCDEFGHIJKLMNOPQRSTUVWXYZ0123456789-/.";
        uint64_t value = 0;
        for (uint8_t b : callsign) value = (value << 8) | b;
        call_t out;
        out.fill(0);
        for (size_t i = 0; value; value /= 40) out[i++] = alphabet[value % 40];
        return out;
    }

    LinkSetupFrame() {}

    LinkSetupFrame& myCall(const char*) { return *this; }
};

} // mobilinkd
