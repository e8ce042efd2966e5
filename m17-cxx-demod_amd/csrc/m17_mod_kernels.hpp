// Synthetic M17 baseband on the device: the framing of the reference's modulator CLI
// (apps/m17-mod.cpp:164-504,628-677) — preamble, LSF, stream / BERT / packet frames, EOT — as symbols, pulse shaping
// (one symbol per 10 samples through the 150-tap RRC in double, x 7168, truncation to int16, m17-mod.cpp:204-224) and the
// impairments of BASELINE config 5 (AWGN, DC offset, gain, timing phase, loud lead-in), written straight into the
// context's input slab so that large configurations need no host-generated input over PCIe.
//
// Two kernels:
//   mod_symbols_kernel  one lane per channel builds that channel's symbol stream (integer work: PRBS9, K=5 convolutional
//                       encoder with the P1/P2/P3 puncture, QPP interleaver, decorrelator, Golay(24,12) LICH, CRC-16);
//   mod_shape_kernel    time-parallel: every sample is at most 15 multiply-adds in double (exact accumulation order),
//                       noise is an integer-hash sum-of-uniforms Gaussian, so the slab is reproducible bit for bit.
// The parameter block and the per-channel seeding are those of the test generator the parity tests compare with
// (tests/: oracle m17o_generate_batch); every int16 must match.
#pragma once

#include "m17_common.hpp"

namespace m17 {

struct ModParams {   // layout of m17_synth_params (include/m17hip.h)
    uint64_t seed;
    int32_t kind, n_frames, lead_in, phase, tail, total, invert, n_preamble;
    double lead_sigma, noise_sigma, dc_offset, gain, tail_sigma;
};

__host__ __device__ inline uint64_t mod_splitmix64(uint64_t x)
{
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}
__device__ inline uint64_t mod_channel_seed(uint64_t base, uint64_t cc) { return base ^ mod_splitmix64(cc * 0x9E3779B97F4A7C15ull + 1); }

// M17Randomizer.h:16-22 (decorrelation sequence)
__device__ __constant__ const uint8_t MOD_DC_SEQ[46] = {
    0xd6, 0xb5, 0xe2, 0x30, 0x82, 0xFF, 0x84, 0x62, 0xba, 0x4e, 0x96, 0x90, 0xd8, 0x98, 0xdd, 0x5d,
    0x0c, 0xc8, 0x52, 0x43, 0x91, 0x1d, 0xf8, 0x6e, 0x68, 0x2F, 0x35, 0xda, 0x14, 0xea, 0xcd, 0x76,
    0x19, 0x8d, 0xd5, 0x80, 0xd1, 0x33, 0x87, 0x13, 0x57, 0x18, 0x2d, 0x29, 0x78, 0xc3};

struct ModFrame {   // the 368 bits of a frame after the interleaver (bit i of the frame = bit i & 31 of w[i >> 5])
    uint32_t w[12];
    __device__ void clear() { for (int i = 0; i < 12; ++i) w[i] = 0; }
    __device__ void set_interleaved(uint32_t k, uint32_t bit)   // PolynomialInterleaver<45,92,368>::interleave: out[(45 k + 92 k^2) % 368] = in[k]
    {
        const uint32_t q = (45u * k + 92u * k * k) % 368u;
        // (dynamic word index: a small select chain keeps w[] in registers)
        for (int i = 0; i < 12; ++i) if ((q >> 5) == (uint32_t)i) w[i] |= bit << (q & 31u);
    }
    __device__ uint32_t get(uint32_t i) const
    {
        uint32_t v = 0;
        for (int j = 0; j < 12; ++j) if ((i >> 5) == (uint32_t)j) v = w[j];
        return (v >> (i & 31u)) & 1u;
    }
};

struct SymWriter {
    int8_t* p;
    uint32_t n;
    __device__ void dibit(uint32_t bits) { const int8_t map[4] = {+1, +3, -1, -3}; p[n++] = map[bits & 3u]; }   // m17-mod.cpp:164-174
    __device__ void byte(uint32_t b) { for (int k = 0; k < 4; ++k) dibit(b >> (6 - 2 * k)); }
    __device__ void preamble() { for (int i = 0; i < 48; ++i) byte(0x77); }                                    // m17-mod.cpp:264-280
    __device__ void frame(const ModFrame& f)   // randomize (M17Randomizer.h:51-57) + dibits
    {
        for (uint32_t i = 0; i < 368; i += 2) {
            const uint32_t b0 = f.get(i) ^ ((MOD_DC_SEQ[i >> 3] >> (7 - (i & 7))) & 1u);
            const uint32_t b1 = f.get(i + 1) ^ ((MOD_DC_SEQ[(i + 1) >> 3] >> (7 - ((i + 1) & 7))) & 1u);
            dibit((b0 << 1) | b1);
        }
    }
    __device__ void zeros(uint32_t k) { for (uint32_t i = 0; i < k; ++i) p[n++] = 0; }
};

// rate-1/2 K=5 encoder (polys 031 / 027) + 4 flush bits, punctured (Trellis.h:17-40, Util.h:193-211), written through the
// interleaver into frame positions first .. first + kept - 1.   which: 1 = P1 (61, zeros at 2, 6, .., 58), 2 = P2 (12, zero at
// 11), 3 = P3 (8, zero at 7).  `bit_at(i)` yields message bit i.
template <typename F>
__device__ inline void mod_encode(ModFrame& f, F bit_at, uint32_t nbits, int which, uint32_t first, uint32_t out_max)
{
    uint32_t mem = 0, pidx = 0, k = 0;
    const uint32_t plen = which == 1 ? 61u : (which == 2 ? 12u : 8u);
    for (uint32_t i = 0; i < nbits + 4u && k < out_max; ++i) {
        const uint32_t x = i < nbits ? bit_at(i) : 0u;
        mem = ((mem << 1) | x) & 31u;
        for (int j = 0; j < 2 && k < out_max; ++j) {
            const uint32_t c = (uint32_t)__popc((j ? 027u : 031u) & mem) & 1u;
            const bool keep = which == 1 ? !((pidx & 3u) == 2u && pidx <= 58u) : (which == 2 ? pidx != 11u : pidx != 7u);
            if (keep) { f.set_interleaved(first + k, c); ++k; }
            if (++pidx == plen) pidx = 0;
        }
    }
}

__device__ inline uint32_t mod_crc16(const uint8_t* d, int n)   // CRC16<0x5935, 0xFFFF> (CRC16.h:12-70)
{
    uint32_t reg = 0xFFFFu;
    for (int i = 0; i != 16; ++i) { const uint32_t bit = reg & 1u; if (bit) reg ^= 0x5935u; reg >>= 1; if (bit) reg |= 0x8000u; }
    for (int k = 0; k < n; ++k)
        for (int i = 0; i != 8; ++i) {
            const uint32_t msb = reg & 0x8000u;
            reg = ((reg << 1) & 0xFFFFu) | ((d[k] >> (7 - i)) & 1u);
            if (msb) reg ^= 0x5935u;
        }
    for (int i = 0; i != 16; ++i) { const uint32_t msb = reg & 0x8000u; reg = (reg << 1) & 0xFFFFu; if (msb) reg ^= 0x5935u; }
    return reg;
}
// CRC-16/X.25 register update for one byte (reflected 0x1021, no final xor): the checksum of apps/m17-demod.cpp:218
__host__ __device__ inline uint32_t mod_crc16_x25_update(uint32_t crc, uint32_t b)
{
    crc ^= b;
    for (int i = 0; i != 8; ++i) crc = (crc & 1u) ? ((crc >> 1) ^ 0x8408u) : (crc >> 1);
    return crc;
}
__device__ inline uint32_t mod_golay24(uint32_t data)   // Golay24.h:100-129
{
    uint32_t cw = data;
    for (int i = 0; i != 12; ++i) { if (cw & 1u) cw ^= 0xC75u; cw >>= 1; }
    cw |= data << 11;
    return (cw << 1) | ((uint32_t)__popc(cw) & 1u);
}

// symbols per channel for a parameter block (upper bound used to size the symbol buffer)
__host__ __device__ inline uint32_t mod_max_symbols(int n_frames, int n_preamble) { return 192u * (uint32_t)(n_frames + (n_preamble > 2 ? n_preamble : 2) + 2) + 48u; }

__global__ __launch_bounds__(64) void mod_symbols_kernel(ModParams base, uint32_t C, uint32_t chan0, int8_t* sym, size_t sym_pitch, uint32_t* nsym_out)
{
    const uint32_t c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const uint64_t cc = (uint64_t)chan0 + c;
    const uint64_t seed = mod_channel_seed(base.seed, cc);
    const int kind = base.kind < 0 ? (int)(cc % 2) : base.kind;
    SymWriter ss{sym + (size_t)c * sym_pitch, 0};
    uint64_t rs = mod_splitmix64(seed ^ 0xA5A5A5A5ull);
    auto rnd = [&rs]() { rs = mod_splitmix64(rs); return rs; };
    ModFrame f;
    const uint8_t call[6] = {0x00, 0x00, 0x4B, 0x13, 0xD1, 0x06};   // "N0CALL" base-40 (LinkSetupFrame.h:46-86)
    uint8_t lsf[30];
    auto make_lsf = [&](uint32_t type_field) {   // m17-mod.cpp:310-347
        for (int i = 0; i < 30; ++i) lsf[i] = 0;
        for (int i = 0; i < 6; ++i) { lsf[i] = 0xFF; lsf[6 + i] = call[i]; }
        lsf[12] = (uint8_t)(type_field >> 8); lsf[13] = (uint8_t)(type_field & 0xFF);
        const uint32_t crc = mod_crc16(lsf, 28);
        lsf[28] = (uint8_t)(crc >> 8); lsf[29] = (uint8_t)(crc & 0xFF);
    };
    auto send_lsf = [&]() {   // 240 bits -> 488 -> P1 -> 368 (m17-mod.cpp:348-386); sync 0x55F7
        f.clear();
        mod_encode(f, [&](uint32_t i) { return (uint32_t)(lsf[i >> 3] >> (7 - (i & 7))) & 1u; }, 240, 1, 0, 368);
        ss.byte(0x55); ss.byte(0xF7); ss.frame(f);
    };
    if (kind == 0) {   // BERT: two preambles then frames of 197 PRBS9 bits (m17-mod.cpp:442-504,664-677); sync 0xDF55
        for (int k = 0; k < (base.n_preamble > 0 ? base.n_preamble : 2); ++k) ss.preamble();
        uint32_t prbs = 1;
        for (int i = 0; i < base.n_frames; ++i) {
            uint32_t bits[7] = {0, 0, 0, 0, 0, 0, 0};
            for (uint32_t b = 0; b < 197; ++b) {   // PRBS9::generate (Util.h:353-358)
                const uint32_t r = ((prbs >> 8) ^ (prbs >> 4)) & 1u;
                prbs = ((prbs << 1) | r) & 0x1FFu;
                for (int j = 0; j < 7; ++j) if ((b >> 5) == (uint32_t)j) bits[j] |= r << (b & 31u);
            }
            f.clear();
            mod_encode(f, [&](uint32_t b) { uint32_t v = 0; for (int j = 0; j < 7; ++j) if ((b >> 5) == (uint32_t)j) v = bits[j]; return (v >> (b & 31u)) & 1u; }, 197, 2, 0, 368);
            ss.byte(0xDF); ss.byte(0x55); ss.frame(f);
        }
    } else if (kind == 1) {   // voice-like stream: preamble, LSF, N stream frames, EOT (m17-mod.cpp:407-440,509-564)
        for (int k = 0; k < (base.n_preamble > 0 ? base.n_preamble : 1); ++k) ss.preamble();
        const uint32_t can = (uint32_t)(rnd() & 15u);
        make_lsf(((can >> 1) << 8) | (5u | ((can & 1u) << 7)));
        send_lsf();
        for (int i = 0; i < base.n_frames; ++i) {
            uint8_t data[18];
            for (int k = 0; k < 16; k += 8) { const uint64_t r = rnd(); for (int q = 0; q < 8; ++q) data[2 + k + q] = (uint8_t)(r >> (8 * q)); }
            uint32_t fn = (uint32_t)(i & 0x7FFF);
            if (i == base.n_frames - 1) fn |= 0x8000u;
            data[0] = (uint8_t)(fn >> 8); data[1] = (uint8_t)(fn & 0xFF);
            f.clear();
            {   // LICH: four Golay(24,12) words of LSF fragment i % 6 (m17-mod.cpp:509-548), frame bits 0..95
                const uint32_t n = (uint32_t)(i % 6);
                const uint8_t* seg = lsf + 5 * n;
                const uint32_t w[4] = {((uint32_t)seg[0] << 4) | ((seg[1] >> 4) & 0x0Fu), (((uint32_t)seg[1] & 0x0Fu) << 8) | seg[2],
                                       ((uint32_t)seg[3] << 4) | ((seg[4] >> 4) & 0x0Fu), (((uint32_t)seg[4] & 0x0Fu) << 8) | (n << 5)};
                for (int k = 0; k < 4; ++k) {
                    const uint32_t e = mod_golay24(w[k]);
                    for (int b = 0; b < 24; ++b) f.set_interleaved((uint32_t)(k * 24 + b), (e >> (23 - b)) & 1u);
                }
            }
            mod_encode(f, [&](uint32_t b) { return (uint32_t)(data[b >> 3] >> (7 - (b & 7))) & 1u; }, 144, 2, 96, 272);
            ss.byte(0xFF); ss.byte(0x5D); ss.frame(f);
        }
        ss.byte(0x55); ss.byte(0x5D); ss.zeros(40);   // EOT (m17-mod.cpp:289-308)
    } else if (kind == 2) {   // RAW packet: preamble, LSF (type 0x0002), N packet frames of 206 bits -> 420 -> P3 -> 368
        for (int k = 0; k < (base.n_preamble > 0 ? base.n_preamble : 1); ++k) ss.preamble();
        make_lsf(0x0002u);
        send_lsf();
        for (int i = 0; i < base.n_frames; ++i) {
            uint8_t d[26];
            for (int k = 0; k < 24; k += 8) { const uint64_t r = rnd(); for (int q = 0; q < 8; ++q) d[k + q] = (uint8_t)(r >> (8 * q)); }
            d[24] = (uint8_t)rnd();
            const bool last = i == base.n_frames - 1;
            d[25] = (uint8_t)((last ? 0x80 : 0x00) | ((last ? 25 : i) << 2));
            f.clear();
            mod_encode(f, [&](uint32_t b) { return (uint32_t)(d[b >> 3] >> (7 - (b & 7))) & 1u; }, 206, 3, 0, 368);
            ss.byte(0x75); ss.byte(0xFF); ss.frame(f);
        }
        ss.byte(0x55); ss.byte(0x5D); ss.zeros(40);
    } else if (kind == 4) {   // RAW packet closed by its CRC-16/X.25 frame check sequence (what apps/m17-demod.cpp:207-253 verifies)
        for (int k = 0; k < (base.n_preamble > 0 ? base.n_preamble : 1); ++k) ss.preamble();
        make_lsf(0x0002u);
        send_lsf();
        uint32_t crc = 0xFFFFu;
        for (int i = 0; i < base.n_frames; ++i) {
            uint8_t d[26];
            for (int k = 0; k < 24; k += 8) { const uint64_t r = rnd(); for (int q = 0; q < 8; ++q) d[k + q] = (uint8_t)(r >> (8 * q)); }
            d[24] = (uint8_t)rnd();
            const bool last = i == base.n_frames - 1;
            if (!last) {
                for (int k = 0; k < 25; ++k) crc = mod_crc16_x25_update(crc, d[k]);
                d[25] = (uint8_t)(i << 2);
            } else {
                const int len = 2 + (int)(rnd() % 24u);
                for (int k = 0; k < 25; ++k) if (k < len - 2) crc = mod_crc16_x25_update(crc, d[k]);
                const uint32_t fcs = ~crc & 0xFFFFu;
                for (int k = 0; k < 25; ++k) {
                    if (k == len - 2) d[k] = (uint8_t)(fcs & 0xFFu);
                    else if (k == len - 1) d[k] = (uint8_t)(fcs >> 8);
                    else if (k >= len) d[k] = 0;
                }
                d[25] = (uint8_t)(0x80 | (len << 2));
            }
            f.clear();
            mod_encode(f, [&](uint32_t b) { return (uint32_t)(d[b >> 3] >> (7 - (b & 7))) & 1u; }, 206, 3, 0, 368);
            ss.byte(0x75); ss.byte(0xFF); ss.frame(f);
        }
        ss.byte(0x55); ss.byte(0x5D); ss.zeros(40);
    }
    nsym_out[c] = ss.n;
}

// Zero-mean unit-variance noise for (stream, n): sum of 8 uniform u16 from two splitmix64 words, exact in double
__device__ inline double mod_unit_noise(uint64_t stream, uint64_t n)
{
    const uint64_t h = mod_splitmix64(stream ^ (n * 0xD1342543DE82EF95ull));
    const uint64_t g = mod_splitmix64(h);
    int64_t sum = 0;
    for (int k = 0; k < 4; ++k) { sum += (int64_t)((h >> (16 * k)) & 0xFFFF); sum += (int64_t)((g >> (16 * k)) & 0xFFFF); }
    const double inv_sigma = 1.0 / 53510.38419625641;
    return (double)(2 * sum - 8 * 65535) * 0.5 * inv_sigma;
}
__device__ inline int16_t mod_sat16(double v)
{
    double r = rint(v);
    if (r > 32767.0) r = 32767.0;
    if (r < -32768.0) r = -32768.0;
    return (int16_t)r;
}

__global__ __launch_bounds__(256) void mod_shape_kernel(ModParams base, uint32_t C, uint32_t T, uint32_t chan0, const int8_t* sym, size_t sym_pitch,
                                                        const uint32_t* nsym_in, int16_t* x, size_t xpitch)
{
    const uint32_t c = blockIdx.y;
    const uint32_t n = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C || n >= T) return;
    const uint64_t cc = (uint64_t)chan0 + c;
    const uint64_t seed = mod_channel_seed(base.seed, cc);
    const int kind = base.kind < 0 ? (int)(cc % 2) : base.kind;
    const uint32_t nsym = kind == 3 ? 0u : nsym_in[c];
    const uint32_t nburst = nsym * 10u;
    uint32_t phase = base.phase >= 0 ? (uint32_t)base.phase : (uint32_t)(mod_splitmix64(seed ^ 0x1234567ull) % 10u);
    if (kind == 3) phase = 0;
    const uint32_t start = (uint32_t)base.lead_in + phase;
    const uint64_t ns = mod_splitmix64(seed ^ 0x5EEDull);
    const double tail_sigma = base.noise_sigma > base.tail_sigma ? base.noise_sigma : base.tail_sigma;
    const bool in_burst = kind != 3 && n >= start && n < start + nburst + 150u;
    double v;
    if (n < (uint32_t)base.lead_in) {
        v = mod_unit_noise(ns, n) * base.lead_sigma;
    } else if (in_burst) {
        // y[m] = sum_i taps[i] * u[m - i], u nonzero only at multiples of 10 (m17-mod.cpp:204-224): i ascending, exact order
        const uint32_t m = n - start;
        const int8_t* sr = sym + (size_t)c * sym_pitch;
        double acc = 0.0;
        for (uint32_t i = m % 10u; i < 150u && i <= m; i += 10u) {
            const uint32_t k = (m - i) / 10u;
            if (k < nsym) {
                const double tap = i == 149u ? 0.0 : (i <= 74u ? core::RRC_HALF[i] : core::RRC_HALF[148u - i]);
                const double p = (double)sr[k] * tap;
                acc = acc + p;
            }
        }
        double b = acc * 7168.0;
        b = b * (base.invert ? -1.0 : 1.0);
        const double s = (double)(int16_t)(int32_t)b;   // the reference casts the shaped sample to int16 (truncation)
        double t = s * base.gain;
        t = t + base.dc_offset;
        const double q = mod_unit_noise(ns, n) * base.noise_sigma;
        v = t + q;
    } else {
        const double q = mod_unit_noise(ns, n) * (n < start ? base.noise_sigma : tail_sigma);
        v = base.dc_offset + q;
    }
    int16_t o = mod_sat16(v);
    // never emit exact zeros outside the burst when there is no noise at all (NaN poisoning of the DCD)
    if (base.noise_sigma == 0.0 && tail_sigma == 0.0 && o == 0 && !in_burst) o = (int16_t)((mod_splitmix64(ns + n) & 1ull) ? 1 : -1);
    x[(size_t)c * xpitch + XPRE + n] = o;
}

}  // namespace m17
