// ORACLE — TEST INFRASTRUCTURE ONLY (see m17_oracle_dsp.hpp header).
// Integer back end of the M17 receive chain: derandomise, deinterleave,
// depuncture, soft Viterbi (K=5, polys 031/027), CRC-16, Golay(24,12), the
// frame-type state machine, plus the TX-side encoders needed by the synthetic
// signal generator.  Every function is pinned by the reference's own
// known-answer tests (tests/test_oracle_kat.py) and by oracle/_ref.
#pragma once

#include <algorithm>
#include <array>
#include <cmath>
#include <cstddef>
#include <cstdint>
#include <cstring>

namespace m17o {

// a14: decorrelation sequence (reference M17Randomizer.h:16-22; M17 spec table).
static const uint8_t DC_SEQ[46] = {
    0xd6, 0xb5, 0xe2, 0x30, 0x82, 0xFF, 0x84, 0x62, 0xba, 0x4e, 0x96, 0x90, 0xd8, 0x98, 0xdd, 0x5d,
    0x0c, 0xc8, 0x52, 0x43, 0x91, 0x1d, 0xf8, 0x6e, 0x68, 0x2F, 0x35, 0xda, 0x14, 0xea, 0xcd, 0x76,
    0x19, 0x8d, 0xd5, 0x80, 0xd1, 0x33, 0x87, 0x13, 0x57, 0x18, 0x2d, 0x29, 0x78, 0xc3};

inline int dc_bit(size_t i) { return (DC_SEQ[i >> 3] >> (7 - (i & 7))) & 1; }

// M17Randomizer<368>::operator() (M17Randomizer.h:43-49): llr *= (bit ? -1 : +1)
inline void derandomize(int8_t* f)
{
    for (size_t i = 0; i != 368; ++i) f[i] = (int8_t)(f[i] * (dc_bit(i) ? -1 : 1));
}
// randomize() (TX, bit XOR) M17Randomizer.h:51-57
inline void randomize_bits(int8_t* f)
{
    for (size_t i = 0; i != 368; ++i) f[i] = (int8_t)(f[i] ^ dc_bit(i));
}

// a15: PolynomialInterleaver<45,92,368> (PolynomialInterleaver.h:21-58)
inline size_t qpp(size_t i) { return (45 * i + 92 * i * i) % 368; }
inline void deinterleave(int8_t* f)
{
    int8_t t[368];
    for (size_t i = 0; i != 368; ++i) t[i] = f[qpp(i)];
    std::memcpy(f, t, 368);
}
inline void interleave(int8_t* f)
{
    int8_t t[368];
    std::memset(t, 0, 368);
    for (size_t i = 0; i != 368; ++i) t[qpp(i)] = f[i];
    std::memcpy(f, t, 368);
}

// a16: puncture matrices (Trellis.h:17-40) and depuncture (Util.h:169-190)
struct Punct { const int8_t* p; size_t n; };
inline Punct punct_matrix(int which)  // 1 = P1 (61, zeros at 2,6,..,58), 2 = P2 (12), 3 = P3 (8)
{
    static int8_t p1[61], p2[12], p3[8];
    static bool init = false;
    if (!init) {
        for (size_t i = 0, j = 2; i != 61; ++i) { if (i == j) { p1[i] = 0; j += 4; } else p1[i] = 1; }
        for (int i = 0; i < 12; ++i) p2[i] = (i == 11) ? 0 : 1;
        for (int i = 0; i < 8; ++i) p3[i] = (i == 7) ? 0 : 1;
        init = true;
    }
    switch (which) { case 1: return {p1, 61}; case 2: return {p2, 12}; default: return {p3, 8}; }
}
// out[i] untouched once the input is exhausted (this is what leaves bert[401] stale, Q4).
inline size_t depuncture(const int8_t* in, size_t IN, int8_t* out, size_t OUT, Punct pm)
{
    size_t index = 0, pindex = 0, erased = 0;
    for (size_t i = 0; i != OUT && index < IN; ++i) {
        if (!pm.p[pindex++]) { out[i] = 0; erased++; }
        else out[i] = in[index++];
        if (pindex == pm.n) pindex = 0;
    }
    return erased;
}
// puncture (TX) Util.h:193-211
template <typename T, typename U>
inline size_t puncture(const T* in, size_t IN, U* out, size_t OUT, Punct pm)
{
    size_t index = 0, pindex = 0, kept = 0;
    for (size_t i = 0; i != IN && index != OUT; ++i) {
        if (pm.p[pindex++]) { out[index++] = (U)in[i]; kept++; }
        if (pindex == pm.n) pindex = 0;
    }
    return kept;
}

// Convolution.h:12-21
inline uint32_t parity32(uint32_t v) { return (uint32_t)__builtin_popcount(v) & 1u; }
inline uint32_t convolve_bit(uint32_t poly, uint32_t mem) { return parity32(poly & mem); }
inline uint32_t update_memory4(uint32_t mem, uint32_t in) { return ((mem << 1) | in) & 31u; }

// ---------------------------------------------------------------------------
// a17: Viterbi<Trellis<4,2>,4> (Viterbi.h:94-240).  16 states, int32 metrics,
// history of decision bits, first-minimum end state, cost = round(min/7.f).
// ---------------------------------------------------------------------------
struct Viterbi {
    int16_t cost[16][2];
    uint8_t prevState[16][2];
    uint16_t history[244];
    Viterbi(int llr_bits = 4)
    {
        const uint32_t polys[2] = {031, 027};
        const int amp = (1 << (llr_bits - 1)) - 1;  // 7 for LLR=4
        for (uint32_t i = 0; i != 16; ++i)
            for (uint32_t j = 0; j != 2; ++j) {
                int bit = (int)convolve_bit(polys[j], i << 1);
                cost[i][j] = (int16_t)(((bit << 1) - 1) * amp);
            }
        for (uint32_t i = 0; i != 16; ++i) {
            uint32_t k = i >= 8;
            for (uint32_t j = 0; j != 2; ++j) prevState[update_memory4(i, j) & 15][k] = (uint8_t)i;
        }
        amp_ = amp;
    }
    int amp_;
    // in: IN soft bits (0 = erasure); out: OUT hard bits.  Returns cost.
    size_t decode(const int8_t* in, size_t IN, uint8_t* out, size_t OUT)
    {
        const int32_t MAXM = std::numeric_limits<int32_t>::max() / 2;
        int32_t prev[16], cur[16];
        for (auto& p : prev) p = MAXM;
        prev[0] = 0;
        size_t h = 0;
        for (size_t i = 0; i != IN; i += 2, ++h) {
            int16_t s0 = in[i], s1 = in[i + 1];
            int16_t c0[8], c1[8];
            for (int j = 0; j < 8; ++j) {
                c0[j] = 0; c1[j] = 0;
                if (s0) { c0[j] = (int16_t)std::abs(cost[j][0] - s0); c1[j] = (int16_t)std::abs(cost[j][0] + s0); }
                if (s1) { c0[j] = (int16_t)(c0[j] + std::abs(cost[j][1] - s1)); c1[j] = (int16_t)(c1[j] + std::abs(cost[j][1] + s1)); }
            }
            uint16_t bits = 0;
            for (int j = 0; j < 8; ++j) {
                int i0 = 2 * j, i1 = 2 * j + 1;
                int32_t m0 = prev[j] + c0[j], m1 = prev[j] + c1[j];
                int32_t m2 = prev[j + 8] + c1[j], m3 = prev[j + 8] + c0[j];
                bool d0 = m0 > m2, d1 = m1 > m3;
                if (d0) bits |= (uint16_t)(1u << i0);
                if (d1) bits |= (uint16_t)(1u << i1);
                cur[i0] = d0 ? m2 : m0;
                cur[i1] = d1 ? m3 : m1;
            }
            history[h] = bits;
            std::memcpy(prev, cur, sizeof(prev));
        }
        size_t best = 0;
        int32_t best_cost = prev[0];
        for (size_t i = 0; i != 16; ++i)
            if (prev[i] < best_cost) { best_cost = prev[i]; best = i; }
        size_t cost_out = (size_t)std::round((float)best_cost / float(amp_));
        // chainback (Viterbi.h:226-236): emit state&1, dropping the flush steps
        size_t steps = IN / 2;
        size_t o = OUT;          // writes out[o-1] downwards
        size_t index = steps;
        size_t state = best;
        size_t hi = steps;       // reads history[hi-1] downwards
        while (o != 0 && hi != 0) {
            int v = (history[--hi] >> state) & 1;
            if (index-- <= OUT) out[--o] = (uint8_t)(state & 1);
            state = prevState[state][v];
        }
        return cost_out;
    }
};

// a18: to_byte_array (Util.h:300-318)
inline void to_bytes(const uint8_t* bits, size_t n, uint8_t* out)
{
    size_t i = 0, b = 0;
    uint8_t tmp = 0;
    for (size_t k = 0; k != n; ++k) {
        tmp |= (uint8_t)(bits[k] << (7 - b));
        if (++b == 8) { out[i] = tmp; tmp = 0; ++i; b = 0; }
    }
    if (i < (n + 7) / 8) out[i] = tmp;
}

// CRC16<0x5935,0xFFFF> (CRC16.h:12-70)
struct Crc16 {
    uint16_t reg = 0xFFFF;
    void reset()
    {
        reg = 0xFFFF;
        for (size_t i = 0; i != 16; ++i) {
            uint16_t bit = reg & 1;
            if (bit) reg ^= 0x5935;
            reg >>= 1;
            if (bit) reg |= 0x8000;
        }
    }
    void add(uint8_t byte)
    {
        uint16_t r = reg;
        for (size_t i = 0; i != 8; ++i) {
            uint16_t msb = r & 0x8000;
            r = (uint16_t)(((r << 1) & 0xFFFF) | ((byte >> (7 - i)) & 1));
            if (msb) r ^= 0x5935;
        }
        reg = r;
    }
    uint16_t get() const
    {
        uint16_t r = reg;
        for (size_t i = 0; i != 16; ++i) {
            uint16_t msb = r & 0x8000;
            r = (uint16_t)((r << 1) & 0xFFFF);
            if (msb) r ^= 0x5935;
        }
        return r;
    }
};
inline uint16_t crc16_m17(const uint8_t* d, size_t n)
{
    Crc16 c; c.reset();
    for (size_t i = 0; i < n; ++i) c.add(d[i]);
    return c.get();
}

// Golay(24,12) (Golay24.h:88-222)
namespace golay {
constexpr uint32_t POLY = 0xC75;
inline uint32_t syndrome(uint32_t cw)
{
    cw &= 0xffffffu;
    for (size_t i = 0; i != 12; ++i) { if (cw & 1) cw ^= POLY; cw >>= 1; }
    return cw << 12;
}
inline uint32_t encode23(uint16_t data)
{
    uint32_t cw = data;
    for (size_t i = 0; i != 12; ++i) { if (cw & 1) cw ^= POLY; cw >>= 1; }
    return cw | ((uint32_t)data << 11);
}
inline uint32_t encode24(uint16_t data)
{
    uint32_t cw = encode23(data);
    return (cw << 1) | parity32(cw);
}
struct Lut {
    uint64_t e[2048];  // (syndrome << 24) | error-bits, sorted ascending
    Lut()
    {
        size_t n = 0;
        e[n++] = 0;
        for (int i = 0; i < 23; ++i) { uint32_t v = 1u << i; e[n++] = ((uint64_t)syndrome(v) << 24) | v; }
        for (int i = 0; i < 22; ++i)
            for (int j = i + 1; j < 23; ++j) { uint32_t v = (1u << i) | (1u << j); e[n++] = ((uint64_t)syndrome(v) << 24) | v; }
        for (int i = 0; i < 21; ++i)
            for (int j = i + 1; j < 22; ++j)
                for (int k = j + 1; k < 23; ++k) { uint32_t v = (1u << i) | (1u << j) | (1u << k); e[n++] = ((uint64_t)syndrome(v) << 24) | v; }
        std::sort(e, e + 2048);
    }
};
inline const Lut& lut() { static const Lut l; return l; }
inline bool decode(uint32_t input, uint32_t& output)  // Golay24.h:203-222
{
    uint32_t syn = syndrome(input >> 1);
    const Lut& L = lut();
    size_t lo = 0, hi = 2048;  // lower_bound on the syndrome key
    while (lo < hi) {
        size_t mid = (lo + hi) / 2;
        if ((uint32_t)(L.e[mid] >> 24) < syn) lo = mid + 1; else hi = mid;
    }
    // reference dereferences `it` unconditionally; every 11-bit syndrome is present
    // (perfect code: 1+23+253+1771 = 2048), so lo < 2048 always.
    if (lo < 2048 && (uint32_t)(L.e[lo] >> 24) == syn) {
        uint32_t correction = (uint32_t)((L.e[lo] & 0xFFFFFF) << 1);
        output = input ^ correction;
        return __builtin_popcount(syn) < 3 || !parity32(output);
    }
    return false;
}
}  // namespace golay

// ---------------------------------------------------------------------------
// a18: M17FrameDecoder (M17FrameDecoder.h:40-395).
// ---------------------------------------------------------------------------
enum class DecState : uint8_t { LSF, STREAM, BASIC_PACKET, FULL_PACKET, BERT };
enum class SyncType : uint8_t { LSF, STREAM, PACKET, BERT };
enum class FrameType : uint8_t { LSF, LICH, STREAM, BASIC_PACKET, FULL_PACKET, BERT };
enum class DecodeResult : uint8_t { FAIL, OK, EOS, INCOMPLETE, PACKET_INCOMPLETE };

struct FrameOut {  // what the reference callback would see
    FrameType type;
    int cost;
    uint8_t len;
    uint8_t data[30];
};

template <typename Sink>  // Sink: void(const FrameOut&)
struct FrameDecoder {
    Viterbi vit{4};
    DecState state_ = DecState::LSF;
    uint8_t lich_segments = 0;
    uint8_t lsf[30];
    int8_t dep[488];     // ONE persistent depuncture buffer shared by all frame types (Q4)
    uint8_t bits[240];
    Sink sink;
    explicit FrameDecoder(Sink s) : sink(s) { std::memset(lsf, 0, 30); std::memset(dep, 0, 488); std::memset(bits, 0, 240); }
    void reset() { state_ = DecState::LSF; }
    DecState state() const { return state_; }

    void emit(FrameType t, int cost, const uint8_t* d, size_t n)
    {
        FrameOut f; f.type = t; f.cost = cost; f.len = (uint8_t)n;
        std::memset(f.data, 0, 30); std::memcpy(f.data, d, n);
        sink(f);
    }
    void update_state()  // M17FrameDecoder.h:113-136
    {
        if (bits[111]) { if (bits[109] != 0) state_ = DecState::STREAM; }
        else {
            uint8_t pt = (uint8_t)((bits[109] << 1) | bits[110]);
            state_ = (pt == 1) ? DecState::BASIC_PACKET : DecState::FULL_PACKET;
        }
    }
    DecodeResult decode_lsf(int8_t* buf, size_t& cost)
    {
        depuncture(buf, 368, dep, 488, punct_matrix(1));
        cost = vit.decode(dep, 488, bits, 240);
        to_bytes(bits, 240, lsf);
        if (crc16_m17(lsf, 30) == 0) {
            update_state();
            emit(FrameType::LSF, (int)cost, lsf, 30);
            return DecodeResult::OK;
        }
        lich_segments = 0;
        std::memset(lsf, 0, 30);
        return DecodeResult::FAIL;
    }
    DecodeResult decode_lich(int8_t* buf, size_t& cost)
    {
        uint8_t lich[6] = {0, 0, 0, 0, 0, 0};
        size_t index = 0;
        for (size_t i = 0; i != 4; ++i) {
            uint32_t cw = 0;
            for (size_t j = 0; j != 24; ++j) { cw <<= 1; cw |= (buf[i * 24 + j] > 0); }
            uint32_t dec = 0;
            if (!golay::decode(cw, dec)) return DecodeResult::FAIL;
            dec >>= 12;
            if (i & 1) { lich[index++] |= (uint8_t)(dec >> 8); lich[index++] = (uint8_t)(dec & 0xFF); }
            else { lich[index++] |= (uint8_t)(dec >> 4); lich[index] = (uint8_t)((dec & 0x0F) << 4); }
        }
        emit(FrameType::LICH, 0, lich, 6);
        uint8_t frag = (uint8_t)((lich[5] >> 5) & 7);
        if (frag > 5) { cost = (size_t)-1; return DecodeResult::INCOMPLETE; }
        std::memcpy(lsf + frag * 5, lich, 5);
        lich_segments |= (uint8_t)(1 << frag);
        if ((lich_segments & 0x3F) != 0x3F) { cost = (size_t)-1; return DecodeResult::INCOMPLETE; }
        if (crc16_m17(lsf, 30) == 0) {
            lich_segments = 0;
            state_ = DecState::STREAM;
            cost = 0;
            emit(FrameType::LSF, 0, lsf, 30);
            return DecodeResult::OK;
        }
        cost = 128;
        return DecodeResult::INCOMPLETE;
    }
    DecodeResult decode_bert(int8_t* buf, size_t& cost)
    {
        depuncture(buf, 368, dep, 402, punct_matrix(2));  // dep[401] keeps its old value (Q4)
        cost = vit.decode(dep, 402, bits, 197);
        uint8_t out[25];
        to_bytes(bits, 197, out);
        emit(FrameType::BERT, (int)cost, out, 25);
        return DecodeResult::OK;
    }
    DecodeResult decode_stream(int8_t* buf, size_t& cost)
    {
        depuncture(buf + 96, 272, dep, 296, punct_matrix(2));
        cost = vit.decode(dep, 296, bits, 144);
        uint8_t out[18];
        to_bytes(bits, 144, out);
        emit(FrameType::STREAM, (int)cost, out, 18);
        return DecodeResult::OK;
    }
    DecodeResult decode_packet(int8_t* buf, size_t& cost, FrameType t)
    {
        depuncture(buf, 368, dep, 420, punct_matrix(3));
        cost = vit.decode(dep, 420, bits, 206);
        uint8_t out[26];
        to_bytes(bits, 206, out);
        emit(t, (int)cost, out, 26);  // callback result is `true` for every consumer we model
        if (out[25] & 0x80) { state_ = DecState::LSF; return DecodeResult::OK; }
        return DecodeResult::PACKET_INCOMPLETE;
    }
    DecodeResult run(SyncType st, int8_t* buf, size_t& cost)  // operator(), :353-392
    {
        derandomize(buf);
        deinterleave(buf);
        switch (st) {
        case SyncType::LSF:
            state_ = DecState::LSF;
            return decode_lsf(buf, cost);
        case SyncType::STREAM:
            if (state_ == DecState::LSF) return decode_lich(buf, cost);
            if (state_ == DecState::STREAM) return decode_stream(buf, cost);
            state_ = DecState::LSF;
            break;
        case SyncType::PACKET:
            if (state_ == DecState::BASIC_PACKET) return decode_packet(buf, cost, FrameType::BASIC_PACKET);
            if (state_ == DecState::FULL_PACKET) return decode_packet(buf, cost, FrameType::FULL_PACKET);
            state_ = DecState::LSF;
            break;
        case SyncType::BERT:
            state_ = DecState::BERT;
            return decode_bert(buf, cost);
        }
        return DecodeResult::FAIL;
    }
};

// PRBS9 (Util.h:320-413) — BERT generator / validator used by the payload consumer.
struct Prbs9 {
    uint16_t state = 1;
    bool synced = false;
    uint8_t sync_count = 0;
    uint32_t bit_count = 0, err_count = 0;
    uint8_t history[16] = {0};
    size_t hist_count = 0, hist_pos = 0;
    bool generate()
    {
        bool r = ((state >> 8) ^ (state >> 4)) & 1;
        state = (uint16_t)(((state << 1) | r) & 0x1FF);
        return r;
    }
    void count_errors(bool error)
    {
        bit_count += 1;
        hist_count -= (history[hist_pos >> 3] & (1 << (hist_pos & 7))) != 0;
        if (error) {
            err_count += 1; hist_count += 1;
            history[hist_pos >> 3] |= (uint8_t)(1 << (hist_pos & 7));
            if (hist_count >= 25) synced = false;
        } else history[hist_pos >> 3] &= (uint8_t)~(1 << (hist_pos & 7));
        if (++hist_pos == 128) hist_pos = 0;
    }
    bool synchronize(bool bit)
    {
        bool r = (bit ^ (state >> 8) ^ (state >> 4)) & 1;
        state = (uint16_t)(((state << 1) | bit) & 0x1FF);
        if (r) sync_count = 0;
        else if (++sync_count == 18) {
            synced = true; bit_count += 18;
            std::memset(history, 0, 16); hist_count = 0; hist_pos = 0; sync_count = 0;
        }
        return r;
    }
    bool validate(bool bit)
    {
        bool r;
        if (!synced) r = synchronize(bit);
        else { r = bit ^ generate(); count_errors(r); }
        return r;
    }
};

}  // namespace m17o
