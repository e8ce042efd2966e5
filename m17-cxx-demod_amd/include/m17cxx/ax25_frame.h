// mobilinkd::ax25_frame / write() — what apps/m17-demod.cpp:226-231 needs to present an intact packet: the AX.25 address
// field (destination, source, digipeaters: 7 bytes each, characters shifted left by one, SSID in the last byte, bit 0 of the
// last byte = end of the address field), control / PID, information field and the trailing frame check sequence.  This is
// presentation on the host, off the demodulation hot path (reference include/m17cxx/ax25_frame.h:20-262); kept so that the
// stock application builds against this directory unchanged.
#pragma once

#include <cstddef>
#include <cstdint>
#include <iomanip>
#include <iostream>
#include <optional>
#include <string>
#include <vector>

namespace mobilinkd {

struct ax25_frame
{
    using repeaters_type = std::vector<std::string>;
    using pid_type = std::optional<uint8_t>;
    enum frame_type {UNDEFINED, INFORMATION, SUPERVISORY, UNNUMBERED};

private:
    static constexpr size_t ADDRESS = 7;

    std::string destination_, source_;
    repeaters_type repeaters_;
    frame_type type_ = UNDEFINED;
    uint8_t raw_type_ = 0;
    std::string info_;
    uint16_t fcs_ = 0xFFFF;
    uint16_t crc_ = 0;
    pid_type pid_;

    // one 7-byte address -> "CALL" or "CALL-ssid"; returns whether another address follows
    static bool address(const std::string& frame, size_t at, std::string& text)
    {
        text.clear();
        for (size_t i = 0; i != 6; ++i) {
            const char c = char(uint8_t(frame[at + i]) >> 1);
            if (c == ' ') break;
            text.push_back(c);
        }
        const uint8_t last = uint8_t(frame[at + 6]);
        const int ssid = (last >> 1) & 0x0F;
        if (ssid) text += "-" + std::to_string(ssid);
        return (last & 1) == 0;
    }

    static frame_type control_type(uint8_t control)
    {
        switch (control & 3) {
        case 1: return SUPERVISORY;
        case 3: return UNNUMBERED;
        default: return INFORMATION;
        }
    }

    void parse(const std::string& frame)
    {
        if (frame.size() < 17) return;
        // the FCS travels LSB first in the last two bytes; reported bit-reversed
        const uint16_t wire = uint16_t(uint8_t(frame[frame.size() - 2]) | (uint8_t(frame[frame.size() - 1]) << 8));
        fcs_ = 0;
        for (int b = 0; b != 16; ++b) fcs_ = uint16_t((fcs_ << 1) | ((wire >> b) & 1));

        address(frame, 0, destination_);
        bool more = address(frame, ADDRESS, source_);
        size_t at = 2 * ADDRESS;
        while (more && at + ADDRESS < frame.size()) {
            std::string hop;
            more = address(frame, at, hop);
            repeaters_.push_back(hop);
            at += ADDRESS;
        }
        if (frame.size() < at + 5) return;
        raw_type_ = uint8_t(frame[at++]);
        type_ = control_type(raw_type_);
        if (type_ == UNNUMBERED) pid_ = uint8_t(frame[at++]);
        info_.assign(frame.begin() + at, frame.end() - 2);
    }

public:
    ax25_frame(const std::string& frame) { parse(frame); }

    std::string destination() const { return destination_; }
    std::string source() const { return source_; }
    repeaters_type repeaters() const { return repeaters_; }
    frame_type type() const { return type_; }
    std::string info() const { return info_; }
    uint16_t fcs() const { return fcs_; }
    uint16_t crc() const { return crc_; }
    pid_type pid() const { return pid_; }
};

inline void write(std::ostream& os, const ax25_frame& frame)
{
    os << "Dest: " << frame.destination() << std::endl << "Source: " << frame.source() << std::endl;
    const auto hops = frame.repeaters();
    if (!hops.empty()) {
        os << "Via: ";
        for (const auto& h : hops) os << h << ' ';
        os << std::endl;
    }
    if (frame.pid()) os << "PID: " << std::setbase(16) << int(*frame.pid()) << std::endl;
    os << "Info: " << std::endl << frame.info() << std::endl;
}

} // mobilinkd
