// ORACLE — TEST INFRASTRUCTURE ONLY (see m17_oracle_dsp.hpp header).
// Seeded synthetic M17 baseband generator: a restatement of the framing of the
// reference's modulator CLI (apps/m17-mod.cpp:164-504,628-677), which is the
// only specification of test input in the reference.
//   symbols -> one sample per 10 -> 150-tap RRC in double -> x 7168 -> int16
// plus the impairments of BASELINE config 5 (AWGN, DC offset, gain), a random
// timing phase (0..9 samples) and a loud lead-in (SURVEY §9-Q13 shape ii).
//
// Noise is an integer-hash sum-of-uniforms Gaussian (8 x u16 per sample from
// splitmix64) so that a device-side generator can reproduce it bit-for-bit.
#pragma once

#include "m17_oracle_dsp.hpp"
#include "m17_oracle_fec.hpp"

#include <string>
#include <vector>

namespace m17o {

inline uint64_t splitmix64(uint64_t x)
{
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}

// Zero-mean unit-variance noise for (stream, n): sum of 8 uniform u16, exact in double.
inline double unit_noise(uint64_t stream, uint64_t n)
{
    uint64_t h = splitmix64(stream ^ (n * 0xD1342543DE82EF95ull));
    uint64_t g = splitmix64(h);
    int64_t sum = 0;
    for (int k = 0; k < 4; ++k) { sum += (int64_t)((h >> (16 * k)) & 0xFFFF); sum += (int64_t)((g >> (16 * k)) & 0xFFFF); }
    // mean 8*32767.5, variance 8*(65536^2-1)/12
    const double inv_sigma = 1.0 / 53510.38419625641;  // 1/sqrt(8*(65536^2-1)/12)
    return (double)(2 * sum - 8 * 65535) * 0.5 * inv_sigma;
}

inline int8_t dibit_symbol(uint8_t bits)  // m17-mod.cpp:164-174
{
    static const int8_t map[4] = {+1, +3, -1, -3};
    return map[bits & 3];
}

struct SymbolStream {
    std::vector<int8_t> sym;
    void bytes(const uint8_t* b, size_t n)
    {
        for (size_t i = 0; i < n; ++i)
            for (int k = 0; k < 4; ++k) sym.push_back(dibit_symbol((uint8_t)(b[i] >> (6 - 2 * k))));
    }
    void bits368(const int8_t* f)
    {
        for (size_t i = 0; i < 368; i += 2) sym.push_back(dibit_symbol((uint8_t)((f[i] << 1) | f[i + 1])));
    }
    void preamble() { uint8_t p[48]; std::memset(p, 0x77, 48); bytes(p, 48); }  // m17-mod.cpp:264-280
    void zeros(size_t n) { sym.insert(sym.end(), n, 0); }
};

// rate-1/2 K=5 encoder + 4 flush bits (m17-mod.cpp:348-368, 416-436, 470-498)
inline size_t conv_encode_bits(const uint8_t* in_bits, size_t n, uint8_t* out)
{
    size_t idx = 0;
    uint32_t mem = 0;
    for (size_t i = 0; i < n + 4; ++i) {
        uint32_t x = i < n ? in_bits[i] : 0;
        mem = update_memory4(mem, x);
        out[idx++] = (uint8_t)convolve_bit(031, mem);
        out[idx++] = (uint8_t)convolve_bit(027, mem);
    }
    return idx;
}
inline void bytes_to_bits(const uint8_t* b, size_t nbits, uint8_t* bits)
{
    for (size_t i = 0; i < nbits; ++i) bits[i] = (uint8_t)((b[i >> 3] >> (7 - (i & 7))) & 1);
}

inline void encode_callsign(const std::string& call, uint8_t out[6])  // LinkSetupFrame.h:46-86
{
    uint64_t enc = 0;
    char cs[10] = {0};
    for (size_t i = 0; i < call.size() && i < 9; ++i) cs[i] = call[i];
    for (int i = 9; i >= 0; --i) {
        char c = cs[i];
        enc *= 40;
        if (c >= 'A' && c <= 'Z') enc += (uint64_t)(c - 'A' + 1);
        else if (c >= '0' && c <= '9') enc += (uint64_t)(c - '0' + 27);
        else if (c == '-') enc += 37;
        else if (c == '/') enc += 38;
        else if (c == '.') enc += 39;
    }
    for (int i = 0; i < 6; ++i) out[5 - i] = (uint8_t)(enc >> (8 * i));
}

// LinkSetupFrame::decode_callsign (LinkSetupFrame.h:95-121): 6 bytes big-endian base 40 -> up to 9 characters, NUL padded;
// the all-ones address is "BROADCAST"
inline void decode_callsign(const uint8_t enc[6], char out[10])
{
    static const char map[] = "xABCDEFGHIJKLMNOPQRSTUVWXYZ0123456789-/.";
    std::memset(out, 0, 10);
    bool bc = true;
    for (int i = 0; i < 6; ++i) bc = bc && enc[i] == 0xFF;
    if (bc) { std::memcpy(out, "BROADCAST", 9); return; }
    uint64_t v = 0;
    for (int i = 0; i < 6; ++i) v = (v << 8) | enc[i];
    size_t idx = 0;
    while (v) { out[idx++] = map[v % 40]; v /= 40; }
}

static const uint8_t SYNC_LSF[2] = {0x55, 0xF7}, SYNC_STREAM[2] = {0xFF, 0x5D}, SYNC_PACKET[2] = {0x75, 0xFF},
                     SYNC_BERT[2] = {0xDF, 0x55}, SYNC_EOT[2] = {0x55, 0x5D};

// LSF: 30 bytes -> 488 coded -> P1 -> 368 -> interleave -> randomize (m17-mod.cpp:310-386)
inline void make_lsf(const std::string& src, const std::string& dst, uint16_t type_field, uint8_t lsf[30])
{
    std::memset(lsf, 0, 30);
    if (dst.empty()) std::memset(lsf, 0xFF, 6); else encode_callsign(dst, lsf);
    encode_callsign(src, lsf + 6);
    lsf[12] = (uint8_t)(type_field >> 8);
    lsf[13] = (uint8_t)(type_field & 0xFF);
    uint16_t c = crc16_m17(lsf, 28);
    lsf[28] = (uint8_t)(c >> 8);
    lsf[29] = (uint8_t)(c & 0xFF);
}
inline void finish_frame(int8_t f[368])
{
    interleave(f);
    randomize_bits(f);
}
inline void lsf_frame_bits(const uint8_t lsf[30], int8_t f[368])
{
    uint8_t bits[240], enc[488];
    bytes_to_bits(lsf, 240, bits);
    conv_encode_bits(bits, 240, enc);
    puncture(enc, 488, f, 368, punct_matrix(1));
    finish_frame(f);
}
inline void lich_segment_bits(const uint8_t seg[5], uint8_t n, uint8_t out[96])  // m17-mod.cpp:509-548
{
    uint16_t w[4] = {(uint16_t)((seg[0] << 4) | ((seg[1] >> 4) & 0x0F)), (uint16_t)(((seg[1] & 0x0F) << 8) | seg[2]),
                     (uint16_t)((seg[3] << 4) | ((seg[4] >> 4) & 0x0F)), (uint16_t)(((seg[4] & 0x0F) << 8) | (n << 5))};
    for (int k = 0; k < 4; ++k) {
        uint32_t e = golay::encode24(w[k]);
        for (int i = 0; i < 24; ++i) out[k * 24 + i] = (uint8_t)((e >> (23 - i)) & 1);
    }
}
inline void stream_frame_bits(const uint8_t lich96[96], uint16_t fn, const uint8_t payload[16], int8_t f[368])
{
    uint8_t data[18], bits[144], enc[296];
    data[0] = (uint8_t)(fn >> 8); data[1] = (uint8_t)(fn & 0xFF);
    std::memcpy(data + 2, payload, 16);
    bytes_to_bits(data, 144, bits);
    conv_encode_bits(bits, 144, enc);
    for (int i = 0; i < 96; ++i) f[i] = (int8_t)lich96[i];
    puncture(enc, 296, f + 96, 272, punct_matrix(2));
    finish_frame(f);
}
inline void bert_frame_bits(Prbs9& prbs, int8_t f[368], uint8_t payload_out[25])  // m17-mod.cpp:442-504
{
    uint8_t bits[197], enc[402];
    for (int i = 0; i < 197; ++i) bits[i] = prbs.generate();
    to_bytes(bits, 197, payload_out);
    conv_encode_bits(bits, 197, enc);
    puncture(enc, 402, f, 368, punct_matrix(2));
    finish_frame(f);
}
inline void packet_frame_bits(const uint8_t data26[26], int8_t f[368])  // 206 bits -> 420 -> P3 -> 368
{
    uint8_t bits[206], enc[420];
    bytes_to_bits(data26, 206, bits);
    conv_encode_bits(bits, 206, enc);
    puncture(enc, 420, f, 368, punct_matrix(3));
    finish_frame(f);
}

// CRC-16/X.25 (poly 0x1021 reflected = 0x8408, init 0xFFFF, xorout 0xFFFF; check "123456789" -> 0x906E): the
// boost::crc_optimal<16, 0x1021, 0xFFFF, 0xFFFF, true, true> of apps/m17-demod.cpp:218.  boost is absent here, so this is
// the published algorithm; one byte of the register update, without the final xor.
inline uint16_t crc16_x25_update(uint16_t crc, uint8_t b)
{
    crc ^= b;
    for (int i = 0; i < 8; ++i) crc = (crc & 1) ? (uint16_t)((crc >> 1) ^ 0x8408) : (uint16_t)(crc >> 1);
    return crc;
}
inline uint16_t crc16_x25(const uint8_t* d, size_t n)
{
    uint16_t crc = 0xFFFF;
    for (size_t i = 0; i < n; ++i) crc = crc16_x25_update(crc, d[i]);
    return (uint16_t)~crc;
}

struct GenParams {
    uint64_t seed = 1;
    int kind = 0;             // 0 = BERT, 1 = voice-like stream, 2 = packet (RAW), 3 = noise only, 4 = packet with FCS
    int n_frames = 8;         // payload frames (BERT / stream / packet)
    int lead_in = 0;          // samples of loud noise before the burst (0 = start at the preamble)
    double lead_sigma = 20000.0;
    double noise_sigma = 0.0; // AWGN over the whole stream, LSB
    double dc_offset = 0.0;   // LSB
    double gain = 1.0;
    int phase = -1;           // extra delay 0..9 samples; -1 = derive from seed
    int tail = 0;             // samples of trailing noise (sigma = max(noise_sigma, tail_sigma))
    double tail_sigma = 0.0;
    int total = 0;            // if > 0: pad with tail noise / truncate to exactly this many samples
    int invert = 0;
    int n_preamble = 0;       // 0 = default (2 for BERT as m17-mod does, else 1)
};

struct GenTruth {               // what was sent, for end-to-end checks
    std::vector<std::vector<uint8_t>> payloads;  // per payload frame
    uint8_t lsf[30];
    int burst_start = 0;        // sample index of the first preamble sample
};

inline int16_t sat16(double v)
{
    double r = std::nearbyint(v);
    if (r > 32767.0) r = 32767.0;
    if (r < -32768.0) r = -32768.0;
    return (int16_t)r;
}

inline std::vector<int16_t> generate(const GenParams& p, GenTruth* truth = nullptr)
{
    SymbolStream ss;
    GenTruth tr;
    std::memset(tr.lsf, 0, 30);
    uint64_t rs = splitmix64(p.seed ^ 0xA5A5A5A5ull);
    auto rnd = [&rs]() { rs = splitmix64(rs); return rs; };
    int8_t f[368];
    if (p.kind == 0) {  // BERT: two preambles then frames (m17-mod.cpp:664-677)
        for (int k = 0; k < (p.n_preamble > 0 ? p.n_preamble : 2); ++k) ss.preamble();
        Prbs9 prbs;
        for (int i = 0; i < p.n_frames; ++i) {
            uint8_t pl[25];
            bert_frame_bits(prbs, f, pl);
            ss.bytes(SYNC_BERT, 2); ss.bits368(f);
            tr.payloads.emplace_back(pl, pl + 25);
        }
    } else if (p.kind == 1) {  // voice-like stream: preamble, LSF, N stream frames, EOT
        for (int k = 0; k < (p.n_preamble > 0 ? p.n_preamble : 1); ++k) ss.preamble();
        uint8_t can = (uint8_t)(rnd() & 15);
        make_lsf("N0CALL", "", (uint16_t)(((can >> 1) << 8) | (5 | ((can & 1) << 7))), tr.lsf);
        lsf_frame_bits(tr.lsf, f);
        ss.bytes(SYNC_LSF, 2); ss.bits368(f);
        uint8_t lich[6][96];
        for (uint8_t i = 0; i < 6; ++i) lich_segment_bits(tr.lsf + 5 * i, i, lich[i]);
        for (int i = 0; i < p.n_frames; ++i) {
            uint8_t pl[16];
            for (int k = 0; k < 16; k += 8) { uint64_t r = rnd(); std::memcpy(pl + k, &r, 8); }
            uint16_t fn = (uint16_t)(i & 0x7FFF);
            if (i == p.n_frames - 1) fn |= 0x8000;
            stream_frame_bits(lich[i % 6], fn, pl, f);
            ss.bytes(SYNC_STREAM, 2); ss.bits368(f);
            std::vector<uint8_t> v(18);
            v[0] = (uint8_t)(fn >> 8); v[1] = (uint8_t)fn; std::memcpy(v.data() + 2, pl, 16);
            tr.payloads.push_back(v);
        }
        ss.bytes(SYNC_EOT, 2); ss.zeros(40);  // m17-mod.cpp:289-308
    } else if (p.kind == 2) {  // RAW packet: preamble, LSF (type: packet, RAW), N packet frames
        for (int k = 0; k < (p.n_preamble > 0 ? p.n_preamble : 1); ++k) ss.preamble();
        make_lsf("N0CALL", "", (uint16_t)0x0002, tr.lsf);  // bit0 = 0 packet, bits 2..1 = 01 RAW
        lsf_frame_bits(tr.lsf, f);
        ss.bytes(SYNC_LSF, 2); ss.bits368(f);
        for (int i = 0; i < p.n_frames; ++i) {
            uint8_t d[26];
            for (int k = 0; k < 24; k += 8) { uint64_t r = rnd(); std::memcpy(d + k, &r, 8); }
            d[24] = (uint8_t)rnd();
            bool last = (i == p.n_frames - 1);
            d[25] = (uint8_t)((last ? 0x80 : 0x00) | ((last ? 25 : i) << 2));  // EOF flag + count, low 2 bits unused
            packet_frame_bits(d, f);
            ss.bytes(SYNC_PACKET, 2); ss.bits368(f);
            d[25] &= 0xFC;  // only 206 bits are carried
            tr.payloads.emplace_back(d, d + 26);
        }
        ss.bytes(SYNC_EOT, 2); ss.zeros(40);
    } else if (p.kind == 4) {  // RAW packet whose last two bytes are the CRC-16/X.25 frame check sequence the packet consumer
                               // verifies (apps/m17-demod.cpp:207-253: residue 0x0f47 over contents + FCS)
        for (int k = 0; k < (p.n_preamble > 0 ? p.n_preamble : 1); ++k) ss.preamble();
        make_lsf("N0CALL", "", (uint16_t)0x0002, tr.lsf);
        lsf_frame_bits(tr.lsf, f);
        ss.bytes(SYNC_LSF, 2); ss.bits368(f);
        uint16_t crc = 0xFFFF;
        for (int i = 0; i < p.n_frames; ++i) {
            uint8_t d[26];
            for (int k = 0; k < 24; k += 8) { uint64_t r = rnd(); std::memcpy(d + k, &r, 8); }
            d[24] = (uint8_t)rnd();
            const bool last = (i == p.n_frames - 1);
            if (!last) {
                for (int k = 0; k < 25; ++k) crc = crc16_x25_update(crc, d[k]);
                d[25] = (uint8_t)(i << 2);
            } else {
                const int len = 2 + (int)(rnd() % 24);  // 2..25 bytes in the last frame, FCS included
                for (int k = 0; k < len - 2; ++k) crc = crc16_x25_update(crc, d[k]);
                const uint16_t fcs = (uint16_t)~crc;
                d[len - 2] = (uint8_t)(fcs & 0xFF); d[len - 1] = (uint8_t)(fcs >> 8);
                for (int k = len; k < 25; ++k) d[k] = 0;
                d[25] = (uint8_t)(0x80 | (len << 2));
            }
            packet_frame_bits(d, f);
            ss.bytes(SYNC_PACKET, 2); ss.bits368(f);
            tr.payloads.emplace_back(d, d + 26);
        }
        ss.bytes(SYNC_EOT, 2); ss.zeros(40);
    }

    // pulse shaping: one symbol per 10 samples, continuous 150-tap RRC in double (m17-mod.cpp:204-224)
    const size_t nsym = ss.sym.size();
    const size_t nburst = nsym * 10;
    int phase = p.phase >= 0 ? p.phase : (int)(splitmix64(p.seed ^ 0x1234567ull) % 10);
    if (p.kind == 3) phase = 0;
    std::vector<double> burst(nburst + 150, 0.0);
    {
        double taps[150];
        for (int i = 0; i < 150; ++i) taps[i] = rrc_tap_d(i);
        // y[n] = sum_i taps[i] * x[n-i], x nonzero only at multiples of 10
        for (size_t n = 0; n < nburst + 150; ++n) {
            double acc = 0.0;
            // same accumulation order as the reference FIR (i ascending), zero products skipped
            // only change the sign of zero, never the value.
            size_t i0 = n % 10;
            for (size_t i = i0; i < 150 && i <= n; i += 10) {
                size_t k = (n - i) / 10;
                if (k < nsym) acc += (double)ss.sym[k] * taps[i];
            }
            burst[n] = acc * 7168.0 * (p.invert ? -1.0 : 1.0);
        }
    }
    size_t start = (size_t)p.lead_in + (size_t)phase;
    size_t natural = start + (p.kind == 3 ? 0 : nburst + 150) + (size_t)p.tail;
    size_t total = p.total > 0 ? (size_t)p.total : natural;
    std::vector<int16_t> out(total);
    tr.burst_start = (int)start;
    const uint64_t ns = splitmix64(p.seed ^ 0x5EEDull);
    double tail_sigma = std::max(p.noise_sigma, p.tail_sigma);
    for (size_t n = 0; n < total; ++n) {
        double v;
        bool in_burst = p.kind != 3 && n >= start && n < start + nburst + 150;
        if (n < (size_t)p.lead_in) {
            v = unit_noise(ns, n) * p.lead_sigma;
        } else if (in_burst) {
            // the reference casts the shaped sample to int16 (truncation) before anything else
            double s = (double)(int16_t)burst[n - start];
            v = s * p.gain + p.dc_offset + unit_noise(ns, n) * p.noise_sigma;
        } else {
            v = p.dc_offset + unit_noise(ns, n) * (n < start ? p.noise_sigma : tail_sigma);
        }
        out[n] = sat16(v);
    }
    // never emit long runs of exact zeros (Q1): a zero-noise gap is replaced by +/-1 dither
    if (p.noise_sigma == 0.0 && tail_sigma == 0.0)
        for (size_t n = 0; n < total; ++n)
            if (out[n] == 0 && !(p.kind != 3 && n >= start && n < start + nburst + 150)) out[n] = (int16_t)((splitmix64(ns + n) & 1) ? 1 : -1);
    if (truth) *truth = tr;
    return out;
}

}  // namespace m17o
