// mobilinkd::M17FrameDecoder — the reference's frame decoder (include/m17cxx/M17FrameDecoder.h:40-395): one 368-soft-bit frame
// and the sync word that announced it in; derandomise, deinterleave, depuncture, Viterbi (Golay for the LICH), CRC, and the
// LSF -> STREAM / PACKET / BERT state machine out, reported through the frame callback.  Names, enumerators, buffer types
// and the callback signature are the reference's (an application written against it compiles unchanged); the body is this
// framework's own.  Scalar form below (one frame per call); batched form = m17hip_decode_frames / the full chain on the GPU
// (BatchedDemodulator.h), whose device code follows the same rules (csrc/m17_decode_device.hpp decode_frame).
// Defined semantics for what the reference leaves indeterminate: all buffers start zeroed and the depuncture
// buffer persists between frames, so the BERT frame's never-written soft bit [401] carries what the last LSF / packet left.
#pragma once

#include "CRC16.h"
#include "Golay24.h"
#include "LinkSetupFrame.h"
#include "M17Randomizer.h"
#include "PolynomialInterleaver.h"
#include "Trellis.h"
#include "Viterbi.h"

#include <algorithm>
#include <array>
#include <cstddef>
#include <cstdio>
#include <functional>
#include <iostream>

extern bool display_lsf;   // defined by the application (reference M17FrameDecoder.h:19)

namespace mobilinkd
{

/// hex dump of a buffer, `header=XX..` (debugging aid of the reference)
template <typename C, size_t N>
void dump(const std::array<C, N>& data, char header = 'D')
{
    std::printf("%c=", header);
    for (auto c : data) std::printf("%02X", unsigned(uint8_t(c)));
    std::printf("\r\n");
}

struct M17FrameDecoder
{
    static constexpr size_t MAX_LICH_FRAGMENT = 5;

    M17Randomizer<368> derandomize_;
    PolynomialInterleaver<45, 92, 368> interleaver_;
    Trellis<4,2> trellis_{makeTrellis<4, 2>({031,027})};
    Viterbi<decltype(trellis_), 4> viterbi_{trellis_};
    CRC16<0x5935, 0xFFFF> crc_;

    enum class State { LSF, STREAM, BASIC_PACKET, FULL_PACKET, BERT };
    enum class SyncWordType { LSF, STREAM, PACKET, BERT };
    enum class DecodeResult { FAIL, OK, EOS, INCOMPLETE, PACKET_INCOMPLETE };
    enum class FrameType { LSF, LICH, STREAM, BASIC_PACKET, FULL_PACKET, BERT };

    State state_ = State::LSF;

    using input_buffer_t = std::array<int8_t, 368>;

    using lsf_conv_buffer_t = std::array<uint8_t, 46>;
    using audio_conv_buffer_t = std::array<uint8_t, 34>;

    using lsf_buffer_t = std::array<uint8_t, 30>;
    using lich_buffer_t = std::array<uint8_t, 6>;
    using audio_buffer_t = std::array<uint8_t, 18>;
    using packet_buffer_t = std::array<uint8_t, 26>;
    using bert_buffer_t = std::array<uint8_t, 25>;

    using output_buffer_t = struct {
        FrameType type;
        union {
            lich_buffer_t lich;
            audio_buffer_t stream;
            packet_buffer_t packet;
            bert_buffer_t bert;
        };
        lsf_buffer_t lsf;
    };

    using depunctured_buffer_t = union {
        std::array<int8_t, 488> lsf;
        std::array<int8_t, 296> stream;
        std::array<int8_t, 420> packet;
        std::array<int8_t, 402> bert;
    };

    using decode_buffer_t = union {
        std::array<uint8_t, 240> lsf;
        std::array<uint8_t, 144> stream;
        std::array<uint8_t, 206> packet;
        std::array<uint8_t, 197> bert;
    };

    /// bool(frame, viterbi_cost): return false only when the data is known to be bad (used for the last packet frame)
    using callback_t = std::function<bool(const output_buffer_t&, int)>;

    callback_t callback_;

    output_buffer_t output_buffer{};
    depunctured_buffer_t depuncture_buffer{};
    decode_buffer_t decode_buffer{};
    uint16_t frame_number = 0;

    uint8_t lich_segments{0};       ///< one bit per LICH fragment received since the last LSF

    M17FrameDecoder(callback_t callback) : callback_(callback) {}

    /// LSF type field (bits 109..111 of the decoded LSF: packet/stream selector and packet type) -> what follows the LSF
    void update_state(std::array<uint8_t, 240>& lsf_output)
    {
        const bool stream = lsf_output[111];
        if (stream) {
            if (lsf_output[109] != 0) state_ = State::STREAM;
            return;
        }
        const unsigned packet_type = (lsf_output[109] << 1) | lsf_output[110];
        state_ = packet_type == 1 ? State::BASIC_PACKET : State::FULL_PACKET;   // RAW : ENCAPSULATED / reserved
    }

    void reset()
    {
        state_ = State::LSF;
        frame_number = 0;
    }

    /// CRC over the 30 bytes of output_buffer.lsf (message + its CRC): 0 when intact
    uint16_t lsf_checksum()
    {
        crc_.reset();
        for (uint8_t byte : output_buffer.lsf) crc_(byte);
        return crc_.get();
    }

    DecodeResult decode_lsf(input_buffer_t& buffer, size_t& viterbi_cost)
    {
        depuncture(buffer, depuncture_buffer.lsf, P1);
        viterbi_cost = viterbi_.decode(depuncture_buffer.lsf, decode_buffer.lsf);
        to_byte_array(decode_buffer.lsf, output_buffer.lsf);
        if (lsf_checksum() != 0) {
            lich_segments = 0;
            output_buffer.lsf.fill(0);
            return DecodeResult::FAIL;
        }
        update_state(decode_buffer.lsf);
        output_buffer.type = FrameType::LSF;
        callback_(output_buffer, viterbi_cost);
        return DecodeResult::OK;
    }

    /// The LICH: 4 Golay(24,12) words = 48 bits = 6 bytes in output_buffer.lich (5 bytes of the LSF + fragment number).
    bool unpack_lich(input_buffer_t& buffer)
    {
        uint64_t bits48 = 0;
        for (size_t word = 0; word != 4; ++word) {
            uint32_t received = 0;
            for (size_t j = 0; j != 24; ++j) received = (received << 1) | (buffer[word * 24 + j] > 0);
            uint32_t corrected = 0;
            if (!Golay24::decode(received, corrected)) return false;   // (words already placed stay in output_buffer.lich)
            bits48 |= uint64_t(corrected >> 12) << (12 * (3 - word));
            for (size_t b = 0; b != 6; ++b) output_buffer.lich[b] = uint8_t(bits48 >> (8 * (5 - b)));
        }
        return true;
    }

    DecodeResult decode_lich(input_buffer_t& buffer, size_t& viterbi_cost)
    {
        output_buffer.lich.fill(0);
        if (!unpack_lich(buffer)) return DecodeResult::FAIL;

        output_buffer.type = FrameType::LICH;
        callback_(output_buffer, 0);

        const uint8_t fragment = (output_buffer.lich[5] >> 5) & 7;
        if (fragment > MAX_LICH_FRAGMENT) {
            viterbi_cost = size_t(-1);
            return DecodeResult::INCOMPLETE;
        }
        std::copy_n(output_buffer.lich.begin(), 5, output_buffer.lsf.begin() + fragment * 5);
        lich_segments |= uint8_t(1u << fragment);
        if ((lich_segments & 0x3F) != 0x3F) {
            viterbi_cost = size_t(-1);
            return DecodeResult::INCOMPLETE;    // not all six fragments yet
        }
        if (lsf_checksum() != 0) {
            viterbi_cost = 128;                 // all fragments, bad CRC: keep collecting (fragments get overwritten)
            return DecodeResult::INCOMPLETE;
        }
        lich_segments = 0;
        state_ = State::STREAM;
        viterbi_cost = 0;
        output_buffer.type = FrameType::LSF;
        callback_(output_buffer, viterbi_cost);
        return DecodeResult::OK;
    }

    DecodeResult decode_bert(input_buffer_t& buffer, size_t& viterbi_cost)
    {
        depuncture(buffer, depuncture_buffer.bert, P2);   // 368 inputs fill positions 0..400; [401] keeps its old value (Q4)
        viterbi_cost = viterbi_.decode(depuncture_buffer.bert, decode_buffer.bert);
        to_byte_array(decode_buffer.bert, output_buffer.bert);
        output_buffer.type = FrameType::BERT;
        callback_(output_buffer, viterbi_cost);
        return DecodeResult::OK;
    }

    DecodeResult decode_stream(input_buffer_t& buffer, size_t& viterbi_cost)
    {
        std::array<int8_t, 272> payload;   // the 96 LICH bits in front are skipped once the stream is known
        std::copy(buffer.begin() + 96, buffer.end(), payload.begin());
        depuncture(payload, depuncture_buffer.stream, P2);
        viterbi_cost = viterbi_.decode(depuncture_buffer.stream, decode_buffer.stream);
        to_byte_array(decode_buffer.stream, output_buffer.stream);
        output_buffer.type = FrameType::STREAM;
        callback_(output_buffer, viterbi_cost);
        return DecodeResult::OK;
    }

    DecodeResult decode_packet(input_buffer_t& buffer, size_t& viterbi_cost, FrameType type)
    {
        depuncture(buffer, depuncture_buffer.packet, P3);
        viterbi_cost = viterbi_.decode(depuncture_buffer.packet, decode_buffer.packet);
        to_byte_array(decode_buffer.packet, output_buffer.packet);
        output_buffer.type = type;
        const bool good = callback_(output_buffer, viterbi_cost);
        if (!(output_buffer.packet[25] & 0x80)) return DecodeResult::PACKET_INCOMPLETE;
        state_ = State::LSF;   // the frame with the end-of-packet bit closes the transmission
        return good ? DecodeResult::OK : DecodeResult::FAIL;
    }

    /// One frame.  The sync word decides: an LSF word always (re)starts at LSF; a stream word is a LICH fragment while no
    /// LSF is known (late entry) and a stream frame afterwards; a packet word only counts after a packet LSF; a BERT word
    /// always decodes as BERT.  A word that does not fit the state drops back to LSF and fails the frame.
    DecodeResult operator()(SyncWordType frame_type, input_buffer_t& buffer, size_t& viterbi_cost)
    {
        derandomize_(buffer);
        interleaver_.deinterleave(buffer);

        switch (frame_type) {
        case SyncWordType::LSF:
            state_ = State::LSF;
            return decode_lsf(buffer, viterbi_cost);
        case SyncWordType::BERT:
            state_ = State::BERT;
            return decode_bert(buffer, viterbi_cost);
        case SyncWordType::STREAM:
            if (state_ == State::LSF) return decode_lich(buffer, viterbi_cost);
            if (state_ == State::STREAM) return decode_stream(buffer, viterbi_cost);
            break;
        case SyncWordType::PACKET:
            if (state_ == State::BASIC_PACKET) return decode_packet(buffer, viterbi_cost, FrameType::BASIC_PACKET);
            if (state_ == State::FULL_PACKET) return decode_packet(buffer, viterbi_cost, FrameType::FULL_PACKET);
            break;
        }
        state_ = State::LSF;
        return DecodeResult::FAIL;
    }

    State state() const { return state_; }
};

} // mobilinkd
