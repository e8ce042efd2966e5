// mobilinkd::CRC16 — the reference's bitwise CRC-16 (include/m17cxx/CRC16.h:12-70): M17 uses polynomial 0x5935, initial value
// 0xFFFF, MSB first, no reflection, no final XOR ("123456789" -> 0x772B; a message followed by its CRC gives 0).
// The register is kept in the reference's "message bits pass through 16 zero bits" form: reset() pre-conditions it so that
// get() (= 16 more zero bits) yields the conventional CRC.
#pragma once

#include <array>
#include <cstddef>
#include <cstdint>

namespace mobilinkd
{

template <uint16_t Poly = 0x5935, uint16_t Init = 0xFFFF>
struct CRC16
{
    static constexpr uint16_t MASK = 0xFFFF;
    static constexpr uint16_t LSB = 0x0001;
    static constexpr uint16_t MSB = 0x8000;

    uint16_t reg_ = Init;

    // run the register 16 steps BACKWARDS from Init, so that feeding the message and then 16 zero bits equals the usual
    // "initialise with Init, feed the message" CRC
    void reset()
    {
        uint16_t r = Init;
        for (int step = 0; step < 16; ++step) {
            const bool wrapped = r & LSB;             // a forward step that applied the polynomial left its low bit set
            if (wrapped) r ^= Poly;
            r = uint16_t((r >> 1) | (wrapped ? MSB : 0));
        }
        reg_ = r;
    }

    void operator()(uint8_t byte) { reg_ = crc(byte, reg_); }

    uint16_t crc(uint8_t byte, uint16_t reg)
    {
        for (int bit = 7; bit >= 0; --bit) {
            const bool carry = reg & MSB;
            reg = uint16_t((reg << 1) | ((byte >> bit) & 1));
            if (carry) reg ^= Poly;
        }
        return reg;
    }

    uint16_t get()
    {
        uint16_t r = reg_;
        for (int step = 0; step < 16; ++step) {
            const bool carry = r & MSB;
            r = uint16_t(r << 1);
            if (carry) r ^= Poly;
        }
        return r;
    }

    std::array<uint8_t, 2> get_bytes()
    {
        const uint16_t v = get();
        return {uint8_t(v >> 8), uint8_t(v & 0xFF)};
    }
};

} // mobilinkd
