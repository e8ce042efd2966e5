// K4: per-frame FEC back end on the device — derandomise, deinterleave, depuncture, soft Viterbi
// (K = 5, polys 031/027, LLR = 4), CRC-16, Golay(24,12) LICH recovery and the frame-type state machine.
// Reference: M17FrameDecoder.h:40-395, Viterbi.h:94-240, Util.h:169-190,300-318, CRC16.h, Golay24.h.
//
// Mapping: ONE LANE PER FRAME.  The 16 path metrics live in registers (the 8 ACS butterflies are fully
// unrolled), the 16 decision bits of each trellis step go to an LDS column private to the lane, the frame's
// 368 LLRs are read from an LDS column private to the lane.  64 frames decode in lock step per wave; there
// is no cross-lane traffic at all, so the only LDS cost is conflict-free column access ([word][lane] layout).
// Integer work throughout: bit-exact by construction.
#pragma once

#include "m17_common.hpp"

namespace m17 {

// ---- host-built constant tables (uploaded once per context) ------------------------------------------------
// soft-bit source map for the four depunctured layouts: entry for trellis input position i
//   0x8000            erasure (puncture matrix 0)                      Util.h:176-180
//   0x4000            never written by depuncture: keeps the previous frame's value (BERT [401])
//   else bits 0..8    index into the received 368-LLR frame (after the QPP interleaver, PolynomialInterleaver.h:21-24)
//        bit 9        multiply by -1 (decorrelator bit, M17Randomizer.h:43-49)
struct DecodeTables {
    uint16_t src[8][488];    // kind 0 LSF(488) 1 stream(296) 2 packet(420) 3 BERT(402); 4..7: identity (already depunctured input)
    uint16_t lich_src[96];   // first 96 deinterleaved positions (hard bits for Golay)
    uint32_t golay_fix[2048];  // 11-bit syndrome -> 23-bit error pattern (Golay24.h:131-177: every syndrome occurs once)
};

__device__ __constant__ const int DEC_IN[4] = {488, 296, 420, 402};
__device__ __constant__ const int DEC_OUT[4] = {240, 144, 206, 197};

// LDS columns: element k of lane l lives at base[k * stride + l]
struct DecodeLds {
    uint32_t* llr;    // [92][64]   368 int8 LLRs packed little-endian
    uint32_t* hist;   // [122][64]  16 decision bits per step, two steps per word
    uint32_t* outb;   // [8][64]    decoded bytes (<= 30), little-endian packed
    uint32_t* lsf;    // [8][64]    M17FrameDecoder::output_buffer.lsf (persistent across frames)
    const uint16_t* src;       // [kinds][488] soft-bit source maps (LDS copy in K5, global in the stand-alone kernels)
    const uint16_t* lich_src;  // [96]
    int stride;                // lanes per column (= active lanes of the wave)
    int32_t* soft;             // [488] branch-cost words of the frame being decoded (wave-cooperative decoder only)
    unsigned long long* prof;  // optional 100 MHz tick accumulators {depuncture, trellis, chainback} (diagnostics)
};

// The wave decoder only ever runs on LDS-resident buffers: say so, so that the accesses are ds_* instructions instead of
// flat ones (DecodeLds carries generic pointers because the stand-alone kernels keep their tables in global memory).
#define M17_LDS __attribute__((address_space(3)))
typedef uint32_t v4u __attribute__((ext_vector_type(4)));
template <typename T> __device__ __forceinline__ M17_LDS T* as_lds(T* p) { return (M17_LDS T*)p; }
template <typename T> __device__ __forceinline__ const M17_LDS T* as_lds(const T* p) { return (const M17_LDS T*)p; }
#define M17_GLOBAL __attribute__((address_space(1)))
template <typename T> __device__ __forceinline__ M17_GLOBAL T* as_global(T* p) { return (M17_GLOBAL T*)p; }
// whole-struct copies out of / into LDS (C++ copy constructors do not take address-space qualified objects)
template <typename T> __device__ __forceinline__ T lds_get(const M17_LDS T* p)
{
    static_assert(sizeof(T) % 4 == 0, "word-sized structs only");
    T v;
    const M17_LDS uint32_t* s = reinterpret_cast<const M17_LDS uint32_t*>(p);
    uint32_t* d = reinterpret_cast<uint32_t*>(&v);
#pragma unroll
    for (int k = 0; k < (int)(sizeof(T) / 4); ++k) d[k] = s[k];
    return v;
}
template <typename T> __device__ __forceinline__ void lds_put(M17_LDS T* p, const T& v)
{
    static_assert(sizeof(T) % 4 == 0, "word-sized structs only");
    M17_LDS uint32_t* d = reinterpret_cast<M17_LDS uint32_t*>(p);
    const uint32_t* s = reinterpret_cast<const uint32_t*>(&v);
#pragma unroll
    for (int k = 0; k < (int)(sizeof(T) / 4); ++k) d[k] = s[k];
}

__device__ __forceinline__ int llr_at(const uint32_t* llr, int stride, int lane, int idx)
{
    const uint32_t w = llr[(idx >> 2) * stride + lane];
    return (int)(int8_t)(w >> (8 * (idx & 3)));
}
// the same frame with the LLRs (-7 .. 7) packed two to a byte: element k of lane l at base[(k >> 3) * stride + l], nibble k & 7
// (LQ: the LLR column and the source maps are KNOWN to be in LDS — ds_read instead of flat_load, whose s_waitcnt also waits for the
// kernel's global stores)
template <bool LQ = false>
__device__ __forceinline__ int llr_at_nib(const uint32_t* llr, int stride, int lane, int idx)
{
    const uint32_t w = LQ ? as_lds(llr)[(idx >> 3) * stride + lane] : llr[(idx >> 3) * stride + lane];
    return (int)(((w >> (4 * (idx & 7))) & 0xFu) ^ 8u) - 8;
}
template <bool NIB = false, bool LQ = false>
__device__ __forceinline__ int soft_at(const uint16_t* src, const uint32_t* llr, int stride, int lane, int kind, int i, int stale)
{
    const uint32_t e = LQ ? as_lds(src)[kind * 488 + i] : src[kind * 488 + i];
    if (LQ) {   // branch-free (the frame is in LDS: reading element 0 for an erasure costs nothing)
        const int v = NIB ? llr_at_nib<LQ>(llr, stride, lane, (int)(e & 0x1FFu)) : llr_at(llr, stride, lane, (int)(e & 0x1FFu));
        const int w = (e & 0x200u) ? -v : v;
        return (e & 0x8000u) ? 0 : ((e & 0x4000u) ? stale : w);
    }
    if (e & 0x8000u) return 0;
    if (e & 0x4000u) return stale;
    const int v = NIB ? llr_at_nib<LQ>(llr, stride, lane, (int)(e & 0x1FFu)) : llr_at(llr, stride, lane, (int)(e & 0x1FFu));
    return (e & 0x200u) ? -v : v;
}
// 368 LLR bytes (92 words) <-> 46 words of nibbles
__device__ __forceinline__ uint32_t pack_llr_nibbles(uint32_t lo, uint32_t hi)   // bytes b0..b3 of lo, b4..b7 of hi -> eight nibbles
{
    auto four = [](uint32_t w) { return (w & 0xFu) | ((w >> 4) & 0xF0u) | ((w >> 8) & 0xF00u) | ((w >> 12) & 0xF000u); };
    return four(lo) | (four(hi) << 16);
}
__device__ __forceinline__ uint32_t unpack_llr_nibbles(uint32_t packed, int half)    // half 0: b0..b3, 1: b4..b7, sign-extended bytes
{
    const uint32_t h = half ? packed >> 16 : packed & 0xFFFFu;
    uint32_t w = 0;
    for (int q = 0; q < 4; ++q) {
        const uint32_t n = (h >> (4 * q)) & 0xFu;
        w |= ((uint32_t)(uint8_t)(int8_t)((int)(n ^ 8u) - 8)) << (8 * q);
    }
    return w;
}
__device__ __forceinline__ uint32_t byte_at(const uint32_t* col, int stride, int lane, int b)
{
    return (col[(b >> 2) * stride + lane] >> (8 * (b & 3))) & 0xFFu;
}

// CRC16<0x5935,0xFFFF> over n bytes of an LDS column (CRC16.h:12-70).
__device__ __forceinline__ uint32_t crc16_col(const uint32_t* col, int stride, int lane, int n)
{
    uint32_t reg = 0xFFFFu;
    for (int i = 0; i != 16; ++i) {  // reset()
        const uint32_t bit = reg & 1u;
        if (bit) reg ^= 0x5935u;
        reg >>= 1;
        if (bit) reg |= 0x8000u;
    }
    for (int k = 0; k < n; ++k) {
        const uint32_t byte = byte_at(col, stride, lane, k);
        for (int i = 0; i != 8; ++i) {
            const uint32_t msb = reg & 0x8000u;
            reg = ((reg << 1) & 0xFFFFu) | ((byte >> (7 - i)) & 1u);
            if (msb) reg ^= 0x5935u;
        }
    }
    for (int i = 0; i != 16; ++i) {  // get()
        const uint32_t msb = reg & 0x8000u;
        reg = (reg << 1) & 0xFFFFu;
        if (msb) reg ^= 0x5935u;
    }
    return reg;
}

__device__ __forceinline__ uint32_t golay_syndrome(uint32_t cw)  // Golay24.h:88-98
{
    cw &= 0xFFFFFFu;
    for (int i = 0; i != 12; ++i) {
        if (cw & 1u) cw ^= 0xC75u;
        cw >>= 1;
    }
    return cw;  // 11 bits (the reference returns this << 12)
}
__device__ __forceinline__ bool golay_decode(const DecodeTables* tb, uint32_t input, uint32_t& output)  // Golay24.h:203-222
{
    const uint32_t syn = golay_syndrome(input >> 1);
    const uint32_t correction = tb->golay_fix[syn & 0x7FFu] << 1;
    output = input ^ correction;
    return __popc(syn) < 3 || !(__popc(output) & 1);
}

// Viterbi<Trellis<4,2>,4>::decode (Viterbi.h:162-239).  Returns cost; decoded bytes (to_byte_array, Util.h:300-318)
// are left in L.outb.  `stale_io`: value of depunctured position 401 left by the previous frame (Q4); updated to this
// frame's position-401 value when the layout writes it (LSF, packet).
template <bool NIB = false, bool LQ = false>
__device__ __forceinline__ uint32_t viterbi_decode(const DecodeTables* tb, const DecodeLds& L, int lane, int kind, int& stale_io)
{
    const int IN = DEC_IN[kind & 3], OUT = DEC_OUT[kind & 3];
    const int steps = IN >> 1;
    constexpr int32_t MAXM = 0x7FFFFFFF / 2;
    int32_t m[16];
#pragma unroll
    for (int s = 0; s < 16; ++s) m[s] = MAXM;
    m[0] = 0;
    uint32_t prev_bits = 0;
    // (the two soft bits of step h + 1 are fetched — map entry, then LLR: two dependent LDS round trips — while step h's sixteen
    // add-compare-selects run; they do not depend on the metrics)
    int ns0 = soft_at<NIB, LQ>(L.src, L.llr, L.stride, lane, kind, 0, stale_io);
    int ns1 = soft_at<NIB, LQ>(L.src, L.llr, L.stride, lane, kind, 1, stale_io);
    for (int h = 0; h < steps; ++h) {
        const int s0 = ns0, s1 = ns1;
        if (2 * h == 400 && (kind & 3) != 3) stale_io = s1;  // this layout writes position 401
        if (h + 1 < steps) {
            ns0 = soft_at<NIB, LQ>(L.src, L.llr, L.stride, lane, kind, 2 * h + 2, stale_io);
            ns1 = soft_at<NIB, LQ>(L.src, L.llr, L.stride, lane, kind, 2 * h + 3, stale_io);
        }
        // branch metrics (Viterbi.h:181-200): an erased bit contributes 0
        const int a = s0 ? abs(-7 - s0) : 0, b = s0 ? abs(7 - s0) : 0;  // |c - s0| for c = -7 / +7
        const int d = s1 ? abs(-7 - s1) : 0, e = s1 ? abs(7 - s1) : 0;
        // |c + s| = |(-c) - s| : the complement costs swap a<->b, d<->e.  The four distinct costs, doubled (see below):
        // nn = (-7,-7), np = (-7,+7), pn = (+7,-7), pp = (+7,+7); the complement of nn is pp, of np is pn
        const uint32_t nn = 2u * (uint32_t)(a + d), np = 2u * (uint32_t)(a + e), pn = 2u * (uint32_t)(b + d), pp = 2u * (uint32_t)(b + e);
        // cost_[0..7] = nn np np nn pn pp pp pn   (polys 031/027); cost1 of a butterfly is the complement of its cost0
        const uint32_t c0[8] = {nn, np, np, nn, pn, pp, pp, pn};
        const uint32_t c1[8] = {pp, pn, pn, pp, np, nn, nn, np};
        // Add-compare-select on DOUBLED metrics: the candidate through the lower predecessor is even, the other one odd
        // (+ 1), so their minimum is the survivor (a tie keeps the lower predecessor, Viterbi.h:143-158: `m0 > m2` is strict) AND
        // carries the decision in its lowest bit — no compare, no select.  v_alignbit shifts that bit into the decision word
        // (state 0 first: it ends up in bit 16, state 15 in bit 31), a shift gives the metric back.  Unsigned: MAXM doubled is
        // beyond INT32_MAX.
        uint32_t n[16];
        uint32_t bits = 0;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const uint32_t lo = (uint32_t)m[j] << 1, hi = ((uint32_t)m[j + 8] << 1) + 1u;
            const uint32_t n0 = min(lo + c0[j], hi + c1[j]);   // new state 2j:   m0 = m[j] + cost0, m2 = m[j+8] + cost1
            const uint32_t n1 = min(lo + c1[j], hi + c0[j]);   // new state 2j+1: m1 = m[j] + cost1, m3 = m[j+8] + cost0
            bits = __builtin_amdgcn_alignbit(n0, bits, 1);
            bits = __builtin_amdgcn_alignbit(n1, bits, 1);
            n[2 * j] = n0 >> 1;
            n[2 * j + 1] = n1 >> 1;
        }
#pragma unroll
        for (int s = 0; s < 16; ++s) m[s] = (int32_t)n[s];
        if (h & 1) {
            const uint32_t hw = prev_bits | (bits & 0xFFFF0000u);
            if (LQ) as_global(L.hist)[(h >> 1) * L.stride + lane] = hw;   // (LQ: the decision words are known to be in GLOBAL memory — a flat store would make every later LDS wait wait for it)
            else L.hist[(h >> 1) * L.stride + lane] = hw;
        } else prev_bits = bits >> 16;
    }
    if (steps & 1) {
        if (LQ) as_global(L.hist)[(steps >> 1) * L.stride + lane] = prev_bits;
        else L.hist[(steps >> 1) * L.stride + lane] = prev_bits;
    }
    // end state: first strict minimum scanning 0 -> 15 (Viterbi.h:211-221)
    int best = 0;
    int32_t best_cost = m[0];
#pragma unroll
    for (int s = 1; s < 16; ++s)
        if (m[s] < best_cost) { best_cost = m[s]; best = s; }
    const uint32_t cost = (uint32_t)roundf((float)best_cost / 7.0f);
    // chainback (Viterbi.h:226-236) fused with to_byte_array: bit n of the message -> byte n>>3, bit 7-(n&7)
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        if (LQ) as_lds(L.outb)[q * L.stride + lane] = 0;
        else L.outb[q * L.stride + lane] = 0;
    }
    uint32_t state = (uint32_t)best;
    uint32_t word = 0;
    int o = OUT;
    int index = steps;
    for (int hi = steps; hi > 0 && o > 0;) {
        --hi;
        const uint32_t hw = LQ ? as_global(L.hist)[(hi >> 1) * L.stride + lane] : L.hist[(hi >> 1) * L.stride + lane];
        const uint32_t hb = (hi & 1) ? (hw >> 16) : (hw & 0xFFFFu);
        const uint32_t v = (hb >> state) & 1u;
        if (index-- <= OUT) {
            --o;
            const int byte = o >> 3;
            word |= (state & 1u) << (8 * (byte & 3) + (7 - (o & 7)));
            if ((o & 31) == 0) {
                if (LQ) as_lds(L.outb)[(byte >> 2) * L.stride + lane] = word;
                else L.outb[(byte >> 2) * L.stride + lane] = word;
                word = 0;
            }
        }
        state = (state >> 1) + (v ? 8u : 0u);  // prevState_[s] = (s>>1, (s>>1)+8)
    }
    return cost;
}

// The lane-per-frame decoder with the sixteen path metrics as EIGHT packed 16-bit pairs P[k] = (m[2k], m[2k + 1]) — one v_pk_add_u16 /
// v_pk_min_u16 serves both new states of a butterfly, the predecessor's metric broadcast into both halves by the instruction's op_sel
// (no move).  The candidates through the lower predecessor j, x = (m[j] + cost0, m[j] + cost1), and through the upper one, y = (m[j + 8] +
// cost1, m[j + 8] + cost0): the survivor is their packed minimum, the decision the SIGN of y - x in each half (v_pk_sub_u16: set exactly when
// y < x — a tie keeps the lower predecessor, as Viterbi.h:143-158's strict `m0 > m2` does; metrics stay below 2^15, so the 16-bit
// difference cannot wrap), collected by a shift and a masked or.  (Up to the middle of round 6 the metrics were kept doubled with the
// decision in the survivor's lowest bit: seven instructions per butterfly where this form needs six, and four more per step for the doubled
// and the odd costs.)  Unreachable states start at the sentinel 20000: any value that stays above every reachable metric (<= 4 x 28 while a
// sentinel is still in play: all sixteen states are reachable from step 4 on) and below 2^15 - 244 x 28 orders exactly like the reference's
// INT_MAX / 2 — the same sentinel is on both sides of every comparison it takes part in.  A step's decision word (ONE 32-bit word per step):
// bit 8 + j = new state 2j, bit 24 + j = new state 2j + 1.
// LDS frame as nibbles, source maps in LDS, decision words in GLOBAL memory (hist: [steps][stride]).  Same results as viterbi_decode.
__device__ __forceinline__ uint32_t pk_add_bcast_lo(uint32_t p, uint32_t c)   // (p.lo + c.lo, p.lo + c.hi)
{
    uint32_t r;
    asm("v_pk_add_u16 %0, %1, %2 op_sel:[0,0] op_sel_hi:[0,1]" : "=v"(r) : "v"(p), "v"(c));
    return r;
}
__device__ __forceinline__ uint32_t pk_add_bcast_hi(uint32_t p, uint32_t c)   // (p.hi + c.lo, p.hi + c.hi)
{
    uint32_t r;
    asm("v_pk_add_u16 %0, %1, %2 op_sel:[1,0] op_sel_hi:[1,1]" : "=v"(r) : "v"(p), "v"(c));
    return r;
}
__device__ __forceinline__ uint32_t pk_min_u16(uint32_t a, uint32_t b)
{
    uint32_t r;
    asm("v_pk_min_u16 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ uint32_t sad_u32(uint32_t a, uint32_t b, uint32_t c)   // |a - b| + c (unsigned)
{
    uint32_t r;
    asm("v_sad_u32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
__device__ __forceinline__ uint32_t pk_sub_u16(uint32_t a, uint32_t b)   // (a.lo - b.lo, a.hi - b.hi), each modulo 2^16
{
    uint32_t r;
    asm("v_pk_sub_u16 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ uint32_t viterbi_decode_pk(const DecodeLds& L, int lane, int kind, int& stale_io)
{
    const int IN = DEC_IN[kind & 3], OUT = DEC_OUT[kind & 3];
    const int steps = IN >> 1;
    constexpr uint32_t SENT = 20000u;
    uint32_t P[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) P[k] = SENT | (SENT << 16);
    P[0] = SENT << 16;   // state 0 sits at 0 before step 0
    // The two soft bits of a step: a map entry each (soft_at: where the bit comes from — a position of the frame, negated or not; an erasure; position 401 as the
    // frame before left it), then the LLR word that holds it.  Two dependent LDS reads per step: in a pipeline of depth two — the entries of step h + 2 and the
    // words of step h + 1 are requested while step h is computed — so that no step waits for a read it has just issued (the compiler's order of the plain form:
    // read the entries, wait, read the words, wait, and only then the add-compare-selects).
    const M17_LDS uint32_t* map2 = as_lds(reinterpret_cast<const uint32_t*>(L.src)) + kind * 244;   // two 16-bit entries per step
    const M17_LDS uint32_t* frame = as_lds(L.llr);
    auto ld_e = [&](int h) -> uint32_t { return h < steps ? map2[h] : 0x80008000u; };
    auto ld_w = [&](uint32_t e16) -> uint32_t { return frame[((e16 & 0x1FFu) >> 3) * L.stride + lane]; };
    auto mk_s = [&](uint32_t e16, uint32_t w) -> int {
        const int v = (int)(((w >> (4u * (e16 & 7u))) & 0xFu) ^ 8u) - 8;
        const int wv = (e16 & 0x200u) ? -v : v;
        return (e16 & 0x8000u) ? 0 : ((e16 & 0x4000u) ? stale_io : wv);
    };
    uint32_t e_cur = ld_e(0), e_nxt = ld_e(1);
    uint32_t wa = ld_w(e_cur & 0xFFFFu), wb = ld_w(e_cur >> 16);
    for (int h = 0; h < steps; ++h) {
        const int s0 = mk_s(e_cur & 0xFFFFu, wa), s1 = mk_s(e_cur >> 16, wb);
        if (2 * h == 400 && (kind & 3) != 3) stale_io = s1;  // this layout writes position 401 (the steps behind it read the new value: mk_s runs at their turn)
        const uint32_t e_nn2 = ld_e(h + 2);
        wa = ld_w(e_nxt & 0xFFFFu); wb = ld_w(e_nxt >> 16);
        e_cur = e_nxt; e_nxt = e_nn2;
        // branch metrics (Viterbi.h:181-200): |c - s| for c = -7 / +7, 0 for an erased bit (s == 0), summed over the step's two bits.  With u = s + 8
        // (s in [-8, 8]: a sign-extended nibble, possibly negated): |-7 - s| = |u - 1| and |7 - s| = |u - 15| — one v_sad_u32 each, the second bit's
        // added to the first's by the same instruction; an erased bit compares u with itself (|u - u| = 0).  Fourteen instructions where the
        // abs / select / add form took twenty-two.
        const uint32_t u0 = (uint32_t)(s0 + 8), u1 = (uint32_t)(s1 + 8);
        const uint32_t r0n = s0 ? 1u : u0, r0p = s0 ? 15u : u0, r1n = s1 ? 1u : u1, r1p = s1 ? 15u : u1;
        const uint32_t a = sad_u32(u0, r0n, 0u), b = sad_u32(u0, r0p, 0u);
        const uint32_t nn = sad_u32(u1, r1n, a), np = sad_u32(u1, r1p, a), pn = sad_u32(u1, r1n, b), pp = sad_u32(u1, r1p, b);
        // cost_[0..7] = nn np np nn pn pp pp pn (polys 031/027); cost1 of a butterfly is the complement of its cost0.
        // Per butterfly j: the pair (cost0, cost1) for the lower predecessor, (cost1, cost0) for the upper one — four distinct pairs in all
        const uint32_t e_nn = nn | (pp << 16), e_np = np | (pn << 16), e_pn = pn | (np << 16), e_pp = pp | (nn << 16);
        const uint32_t ce[8] = {e_nn, e_np, e_np, e_nn, e_pn, e_pp, e_pp, e_pn};
        const uint32_t co[8] = {e_pp, e_pn, e_pn, e_pp, e_np, e_nn, e_nn, e_np};
        uint32_t N[8];
        uint32_t acc = 0;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            // new states (2j, 2j + 1): through j: (m[j] + cost0, m[j] + cost1); through j + 8: (m[j+8] + cost1, m[j+8] + cost0)
            const uint32_t x = (j & 1) ? pk_add_bcast_hi(P[j >> 1], ce[j]) : pk_add_bcast_lo(P[j >> 1], ce[j]);
            const uint32_t y = (j & 1) ? pk_add_bcast_hi(P[(j + 8) >> 1], co[j]) : pk_add_bcast_lo(P[(j + 8) >> 1], co[j]);
            N[j] = pk_min_u16(x, y);
            acc = (acc >> 1) | (pk_sub_u16(y, x) & 0x80008000u);   // after the eighth: butterfly j's two decisions at bits 8 + j and 24 + j
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) P[k] = N[k];
        as_global(L.hist)[h * L.stride + lane] = acc;
    }
    // end state: first strict minimum scanning 0 -> 15 (Viterbi.h:211-221)
    auto metric = [&](int s) { return (int32_t)((s & 1) ? P[s >> 1] >> 16 : P[s >> 1] & 0xFFFFu); };
    int best = 0;
    int32_t best_cost = metric(0);
#pragma unroll
    for (int s = 1; s < 16; ++s) { const int32_t v = metric(s); if (v < best_cost) { best_cost = v; best = s; } }
    const uint32_t cost = (uint32_t)roundf((float)best_cost / 7.0f);
    // chainback (Viterbi.h:226-236) fused with to_byte_array: bit n of the message -> byte n>>3, bit 7-(n&7)
#pragma unroll
    for (int q = 0; q < 8; ++q) as_lds(L.outb)[q * L.stride + lane] = 0;
    // (the decision words come back from global memory SIXTEEN steps at a time: one word per step behind a wait of its own made the walk a
    //  chain of 200 load latencies per frame — what the kernel lasted, whatever the trellis cost)
    uint32_t state = (uint32_t)best, word = 0;
    int o = OUT, index = steps;
    for (int top = steps; top > 0 && o > 0; top -= 16) {
        uint32_t hwv[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) hwv[i] = as_global(L.hist)[max(top - 1 - i, 0) * L.stride + lane];
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            if (top - 1 - i < 0 || o <= 0) break;
            const uint32_t v = (hwv[i] >> (8u + (state >> 1) + ((state & 1u) << 4))) & 1u;
            if (index-- <= OUT) {
                --o;
                const int byte = o >> 3;
                word |= (state & 1u) << (8 * (byte & 3) + (7 - (o & 7)));
                if ((o & 31) == 0) { as_lds(L.outb)[(byte >> 2) * L.stride + lane] = word; word = 0; }
            }
            state = (state >> 1) + (v ? 8u : 0u);  // prevState_[s] = (s>>1, (s>>1)+8)
        }
    }
    return cost;
}

// Wave-cooperative form of the same decoder (one frame per WAVE; used by the one-wave-per-channel demodulator, where
// a frame completes on one channel at a time): sixteen lanes hold the sixteen trellis states, the 16 decision bits of a step
// are one ballot, the add-compare-select exchanges metrics between lanes with DPP row operations (see below), and the
// chainback walks four blocks x sixteen hypothetical entry states on all 64 lanes.  Columns have stride 1 here (per-wave
// LDS arrays).  `wl` = lane id in the wave.
// LDS-only ordering between the lanes of one wave: the fences are restricted to the local address space so that global
// loads in flight (window prefetches) are NOT waited for here.
__device__ __forceinline__ void wave_lds_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront", "local");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront", "local");
}
// Lane <-> trellis state mapping of the wave decoder.  The 16 states live in lanes 0..15 (one DPP row; the other lanes idle
// through the trellis loop).  Before step h the metric of state s sits at POSITION rotr4(s, h mod 4); the two predecessors
// j and j+8 of the new states (2j, 2j+1) then differ in position bit 3 - (h mod 4), and the new states land on the SAME two
// lanes (in-place butterfly).  Position p sits in lane p ^ (p & 4 ? 3 : 0), which makes every one of the four exchanges a
// DPP row operation that folds into the add that consumes it: position bit 3 = lane ^ 8 (row_ror:8), bit 2 = lane ^ 7
// (row_half_mirror), bit 1 = lane ^ 2 and bit 0 = lane ^ 1 (quad_perm) for h mod 4 = 0, 1, 2, 3.
__device__ __forceinline__ int vit_pos_of_lane(int wl) { return (wl & 15) ^ ((wl & 4) ? 3 : 0); }
__device__ __forceinline__ int vit_lane_of_pos(int p) { return p ^ ((p & 4) ? 3 : 0); }
__device__ __forceinline__ int rotl4(int v, int r) { r &= 3; return ((v << r) | (v >> (4 - r))) & 15; }
__device__ __forceinline__ int rotr4(int v, int r) { return rotl4(v, 4 - (r & 3)); }

template <int R>  // R = h mod 4
__device__ __forceinline__ int32_t vit_partner(int32_t m)
{
    if constexpr (R == 3) return __builtin_amdgcn_update_dpp(0, m, 0xB1, 0xF, 0xF, true);        // quad_perm [1,0,3,2]: lane ^ 1
    else if constexpr (R == 2) return __builtin_amdgcn_update_dpp(0, m, 0x4E, 0xF, 0xF, true);   // quad_perm [2,3,0,1]: lane ^ 2
    else if constexpr (R == 1) return __builtin_amdgcn_update_dpp(0, m, 0x141, 0xF, 0xF, true);  // row_half_mirror: lane ^ 7
    else return __builtin_amdgcn_update_dpp(0, m, 0x128, 0xF, 0xF, true);                        // row_ror:8: lane ^ 8
}

// One trellis step.  W (wave-uniform) = the four branch costs of the step, one byte each: (-7,-7) | (-7,+7) << 8 |
// (+7,-7) << 16 | (+7,+7) << 24 (Viterbi.h:181-200, an erased bit costs 0).  sh = 8 * index of cost_[j] of this lane's
// butterfly in that word; the complementary cost (Viterbi.h:143-146: m1/m2 use it) sits at 24 - sh.  A lane always adds
// cost_[j] to its own metric and the complement to its partner's; which of the two sums is "candidate A" depends on
// whether the lane held state j or j + 8 (`upper`, as a wave mask): dec = A > B, new metric = min (ties: equal values).
template <int R>
__device__ __forceinline__ uint32_t vit_step(uint32_t W, int wl, int32_t& m, uint32_t sh, unsigned long long upper)
{
    const int32_t c_own = (int32_t)__builtin_amdgcn_ubfe(W, sh, 8u);
    const int32_t c_oth = (int32_t)__builtin_amdgcn_ubfe(W, 24u - sh, 8u);
    const int32_t mp = vit_partner<R>(m);
    const int32_t own = m + c_own, oth = mp + c_oth;
    // decision = (candidate through the lower predecessor) > (the other one): that candidate is `own` on the lane that held
    // state j and `oth` on the lane that held j + 8, and it loses exactly when the minimum differs from it (a tie keeps it)
    const int32_t low = ((upper >> wl) & 1ull) ? oth : own;
    m = min(own, oth);
    // (the trellis loop runs with lanes 0..15 enabled only: the ballot is the step's 16 decisions, in LANE order)
    return (uint32_t)__ballot(m != low);
}

__device__ __forceinline__ uint32_t viterbi_decode_wave(const DecodeLds& L, int wl, int kind, int& stale_io)
{
    const int IN = DEC_IN[kind & 3], OUT = DEC_OUT[kind & 3];
    const int steps = IN >> 1;
    constexpr int32_t MAXM = 0x7FFFFFFF / 2;
    const unsigned long long tp0 = L.prof ? wall_clock64() : 0ull;
    // depuncture / deinterleave / derandomise the whole frame and form the branch-cost word of every trellis step, 64 steps
    // at a time (lane = step); L.soft[h] = cost word of step h, L.soft[480] = depunctured position 401 (Q4)
    M17_LDS uint32_t* cw = as_lds(reinterpret_cast<uint32_t*>(L.soft));
    M17_LDS uint32_t* hist = as_lds(L.hist);
    M17_LDS uint32_t* outb = as_lds(L.outb);
    {
        // (the source maps stay in global memory: one frame reads <= 8 entries per lane, and the 2 KB LDS copy per workgroup is part of what kept a CU from holding anything beside four of the
        // sequential kernel's workgroups)
        const uint16_t* src = L.src + kind * 488;
        const M17_LDS uint32_t* llr = as_lds(L.llr);
        auto soft = [&](uint32_t e) -> int {  // soft_at() on the wave's own LDS frame
            if (e & 0x8000u) return 0;
            if (e & 0x4000u) return stale_io;
            const uint32_t idx = e & 0x1FFu;
            const int v = (int)(int8_t)(llr[idx >> 2] >> (8 * (idx & 3)));
            return (e & 0x200u) ? -v : v;
        };
        // 64 steps per round (lane = step), the next round's two map entries in flight while this round's are used
        // (the lane id is made opaque here: the per-lane global addresses are otherwise computed once per kernel and kept, spilled)
        int wlo = wl;
        asm volatile("" : "+v"(wlo));
        uint32_t e0 = wlo < steps ? src[2 * wlo] : 0x8000u, e1 = wlo < steps ? src[2 * wlo + 1] : 0x8000u;
        for (int h = wlo; h < steps; h += 64) {
            const int hn = h + 64;
            const uint32_t n0 = hn < steps ? src[2 * hn] : 0x8000u, n1 = hn < steps ? src[2 * hn + 1] : 0x8000u;
            const int s0 = soft(e0), s1 = soft(e1);
            const int a = s0 ? abs(-7 - s0) : 0, b = s0 ? abs(7 - s0) : 0;  // |c - s0| for c = -7 / +7
            const int d = s1 ? abs(-7 - s1) : 0, e = s1 ? abs(7 - s1) : 0;
            cw[h] = (uint32_t)(a + d) | ((uint32_t)(a + e) << 8) | ((uint32_t)(b + d) << 16) | ((uint32_t)(b + e) << 24);
            if (h == 200) cw[480] = (uint32_t)s1;
            e0 = n0; e1 = n1;
        }
    }
    for (int h = steps + wl; h < ((steps + 3) & ~3); h += 64) cw[h] = 0;  // the group load below reads whole groups of four
    wave_lds_sync();
    if ((kind & 3) != 3 && IN > 401) stale_io = (int)cw[480];  // this layout writes position 401
    const int pos = vit_pos_of_lane(wl);
    // per phase r = h mod 4: which butterfly j this lane serves, whether it is the upper lane of its pair, and where cost_[j]
    // sits in the cost word (polys 031/027): cost_[j][0] == -7 <=> j < 4, cost_[j][1] == -7 <=> j in {0,3,4,7}
    uint32_t sh[4];
    unsigned long long upper[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int pb = 3 - r;
        upper[r] = __ballot((pos >> pb) & 1);
        const int j = rotl4(pos & ~(1 << pb), r);
        const bool c0neg = j < 4, c1neg = (((j ^ (j >> 1)) & 1) == 0);
        sh[r] = 8u * ((c0neg ? 0u : 2u) | (c1neg ? 0u : 1u));
    }
    int32_t m = (pos == 0) ? 0 : MAXM;  // state 0 sits at position 0 before step 0
    const unsigned long long tp1 = L.prof ? wall_clock64() : 0ull;
    const int groups = (steps + 3) >> 2;
    v4u Wn = *reinterpret_cast<const M17_LDS v4u*>(cw);
    // the loop runs on the sixteen lanes that hold the states: a step's ballot IS its decision set
    if (wl < 16)
    for (int g = 0; g < groups; ++g) {
        const v4u W = Wn;
        if (g + 1 < groups) Wn = *reinterpret_cast<const M17_LDS v4u*>(cw + 4 * (g + 1));  // next group in flight
        const int h = 4 * g;
        const uint32_t b0 = vit_step<0>(W.x, wl, m, sh[0], upper[0]);
        int32_t m1 = m, m2, m3;
        const uint32_t b1 = vit_step<1>(W.y, wl, m1, sh[1], upper[1]);
        m2 = m1;
        const uint32_t b2 = vit_step<2>(W.z, wl, m2, sh[2], upper[2]);
        m3 = m2;
        const uint32_t b3 = vit_step<3>(W.w, wl, m3, sh[3], upper[3]);
        // a trailing partial group computes steps that do not exist: keep the metric of the last real step
        const int left = steps - h;
        m = left >= 4 ? m3 : (left == 3 ? m2 : (left == 2 ? m1 : m));
        hist[2 * g] = b0 | (b1 << 16);       // two steps per word, the odd one in the upper half
        hist[2 * g + 1] = b2 | (b3 << 16);
    }
    const unsigned long long tp2 = L.prof ? wall_clock64() : 0ull;
    // end state: first strict minimum scanning 0 -> 15 (Viterbi.h:211-221); state s now sits at position rotr4(s, steps)
    int best = 0;
    int32_t best_cost = __builtin_amdgcn_readlane(m, __builtin_amdgcn_readfirstlane(vit_lane_of_pos(rotr4(0, steps))));
#pragma unroll
    for (int s = 1; s < 16; ++s) {
        const int32_t v = __builtin_amdgcn_readlane(m, __builtin_amdgcn_readfirstlane(vit_lane_of_pos(rotr4(s, steps))));
        if (v < best_cost) { best_cost = v; best = s; }
    }
    const uint32_t cost = (uint32_t)roundf((float)best_cost / 7.0f);
    wave_lds_sync();
    for (int q = wl; q < 8; q += 64) outb[q] = 0;
    // ---- chainback (Viterbi.h:226-236) fused with to_byte_array (Util.h:300-318), lane-parallel ---------------------------
    // The walk goes in POSITION space: the current state after step hi sits at position P = rotr4(state, hi + 1).  State bit 0
    // (the decoded bit) is position bit k = -(hi + 1) mod 4, and stepping back to (state >> 1) + 8 v replaces exactly that
    // position bit by the decision v — no rotation per step; P is also the bit index of its own decision inside a
    // (position-ordered) decision set.  The bit a step decodes is the one written into that position four steps earlier:
    // message bit n IS the decision read at step n + 4, so the output is the decision stream itself.
    // A walk of up to 240 dependent steps is cut into four blocks of 64 message bits; lane (b, e) walks block b from the
    // hypothetical entry position e (16 x 4 = 64 lanes, all blocks in lock step: 64 = 0 mod 4, so k and the odd/even half are
    // the same for every block), keeping the position it ends in and the 64 bits it decoded.  Chaining the four blocks is then
    // four lane reads.  Steps beyond the real ones (a partly filled or empty top block) get NEUTRAL decision sets — bit p =
    // position bit k of p — that leave the position unchanged.
    // Decision words: lane order -> position order (positions 4-7 and 12-15 sit in lanes p ^ 3, i.e. the odd nibbles of a set
    // are bit-reversed), neutral words behind them; into the cost-word array, which the trellis no longer needs.
    {
        auto to_pos_order = [](uint32_t x) -> uint32_t {
            const uint32_t odd = x & 0xF0F0F0F0u;
            return (x & 0x0F0F0F0Fu) | ((odd & 0x10101010u) << 3) | ((odd & 0x20202020u) << 1) | ((odd >> 1) & 0x20202020u) | ((odd >> 3) & 0x10101010u);
        };
        const int nreal = (steps + 1) >> 1;
        for (int q = wl; q < 130; q += 64) {
            const uint32_t neutral = (q & 1) ? 0xAAAACCCCu : 0xF0F0FF00u;   // steps 2q (k = 3 or 1) and 2q + 1 (k = 2 or 0)
            uint32_t x = neutral;
            if (q < nreal) {
                x = to_pos_order(hist[q]);
                if (2 * q + 1 >= steps) x = (x & 0xFFFFu) | (neutral & 0xFFFF0000u);   // odd step count: the upper half is not a step
            }
            cw[q] = x;
        }
    }
    wave_lds_sync();
    // One step: the decision of position I sits at bit I (+ 16 for an odd step) of the step's word and becomes bit k of I.  The
    // word is rotated beforehand (off the dependent chain) so that a rotation by I itself brings that bit to bit k; the chain
    // is then two instructions per step, v_alignbit (rotate by I) and v_bfi (insert bit k).  After the four steps of a group
    // I IS the four bits just decoded, first one in bit 0: the groups are collected as nibbles (w4) and the bit order inside
    // the nibbles is put right once, at the end (nib_rev below).
    uint32_t I = (uint32_t)wl & 15u, w4 = 0, w4_hi = 0;
    {
        // block b = wl >> 4 decodes message bits 64 b + 63 .. 64 b, i.e. steps hi = 64 b + 67 .. 64 b + 4: words 32 b + 33 .. 32 b + 2
        const M17_LDS uint32_t* hp = cw + 32 * (wl >> 4) + 2;
        auto rotr = [](uint32_t x, uint32_t n) -> uint32_t { return __builtin_amdgcn_alignbit(x, x, n); };
        auto take = [&](uint32_t prerot, uint32_t bit) {   // I <- I with `bit` replaced by that bit of (prerot rotated right by I)
            const uint32_t r = rotr(prerot, I);
            uint32_t out;
            asm("v_bfi_b32 %0, %1, %2, %3" : "=v"(out) : "v"(bit), "v"(r), "v"(I));   // (bit & r) | (~bit & I)
            I = out;
        };
#pragma unroll
        for (int g = 0; g < 16; ++g) {   // four steps: hi = 3, 2, 1, 0 (mod 4) -> k = 0, 1, 2, 3
            const uint32_t wa = hp[31 - 2 * g], wb = hp[30 - 2 * g];
            // bit k of rotr(rotr(word, c), I) = bit (k + c + I) of word: c = (16 or 0) - k
            const uint32_t a0 = rotr(wa, 16u), a1 = rotr(wa, 31u), b2 = rotr(wb, 14u), b3 = rotr(wb, 29u);
            take(a0, 1u);
            take(a1, 2u);
            take(b2, 4u);
            take(b3, 8u);
            w4 = (w4 << 4) | I;
            if (g == 7) { w4_hi = w4; w4 = 0; }
        }
    }
    auto nib_rev = [](uint32_t x) -> uint32_t {   // reverse the bit order inside every nibble
        return ((x & 0x11111111u) << 3) | ((x & 0x22222222u) << 1) | ((x >> 1) & 0x22222222u) | ((x >> 3) & 0x11111111u);
    };
    // chain the blocks from the top: the walk enters block 3 at the position of the best end state
    {
        uint32_t e = (uint32_t)__builtin_amdgcn_readfirstlane(rotr4(best, steps));
#pragma unroll
        for (int b = 3; b >= 0; --b) {
            const int lane = 16 * b + (int)e;
            const uint32_t hi32 = nib_rev((uint32_t)__builtin_amdgcn_readlane((int)w4_hi, lane)), lo32 = nib_rev((uint32_t)__builtin_amdgcn_readlane((int)w4, lane));
            e = (uint32_t)__builtin_amdgcn_readlane((int)I, lane);
            // message bit n -> byte n >> 3, bit 7 - (n & 7); four bytes per little-endian word; bits at and beyond OUT stay zero
            const int nh = OUT - (64 * b + 32), nl = OUT - 64 * b;
            if (nh > 0) outb[2 * b + 1] = __builtin_bswap32(__builtin_bitreverse32(nh >= 32 ? hi32 : (hi32 & ((1u << nh) - 1u))));
            if (nl > 0) outb[2 * b] = __builtin_bswap32(__builtin_bitreverse32(nl >= 32 ? lo32 : (lo32 & ((1u << nl) - 1u))));
        }
    }
    wave_lds_sync();
    if (L.prof && wl == 0) {
        const unsigned long long tp3 = wall_clock64();
        L.prof[0] += tp1 - tp0; L.prof[1] += tp2 - tp1; L.prof[2] += tp3 - tp2;
    }
    return cost;
}

// The wave decoder once more, with a run-time layout: the on-demand decode of a deferred frame (rare).
__device__ __forceinline__ uint32_t viterbi_decode_wave_cold(const DecodeLds& L, int wl, int kind, int stale)
{
    return viterbi_decode_wave(L, wl, kind, stale);   // (inlined: an out-of-line call costs the kernel a stack and spills around the call)
}
// cost and payload of a (formerly deferred) record: the payload bytes as emit_record leaves them (zero at and beyond len)
__device__ __forceinline__ void complete_record(uint32_t* w, uint32_t cost, const uint32_t* col, int stride, int lane, uint32_t len)
{
    w[4] = cost;
    for (int q = 0; q < 8; ++q) {
        uint32_t v = col[q * stride + lane];
        const int lo = 4 * q;
        if (lo + 4 > (int)len) v = (lo >= (int)len) ? 0u : (v & (0xFFFFFFFFu >> (8 * (lo + 4 - (int)len))));
        w[6 + q] = v;
    }
    w[14] = 0;
    w[15] = 0;
}

// Per-lane frame-decoder state (M17FrameDecoder members that persist between frames).
struct DecoderRegs {
    uint32_t state;          // State enum: 0 LSF 1 STREAM 2 BASIC_PACKET 3 FULL_PACKET 4 BERT
    uint32_t lich_segments;
    int stale401;            // depuncture_buffer[401]
};

struct RecSink {  // where callbacks go
    FrameRec* base;   // this channel's record slots for this run
    uint32_t cap;
    uint32_t* n_run;  // records written this run (register copy owned by caller)
    uint32_t* seq;    // records since reset
    uint32_t channel;
    uint64_t sample_pos;
    uint32_t sync_type;
    uint32_t* overflow;
    uint32_t* defer;  // wave decoder only: this channel's deferred-frame store [cap][46] (LLR nibbles; nullptr: every frame is decoded where it completes)
};

// A payload frame whose decoding is deferred (one-wave-per-channel demodulator, m17_wave_kernel.hpp): its 368 LLRs go to the
// channel's store, its record gets the cost DEFER_TAG | slot and the marker below, and decode_deferred_kernel fills cost and
// payload in after the run (one LANE per frame: a seventh of the instructions the wave decoder spends on it).  Real costs are
// small, 128 or 0xFFFFFFFF ("size_t(-1)"), so bit 31 set with any other bit clear identifies a tag.
constexpr uint32_t DEFER_TAG = 0x80000000u, DEFER_MARK = 0xDEFE77EDu;
__device__ __forceinline__ bool cost_is_deferred(uint32_t cost) { return (cost & DEFER_TAG) && cost != 0xFFFFFFFFu; }
__device__ __forceinline__ int kind_of_frame_type(uint32_t frame_type) { return frame_type == 5u ? 3 : 1; }   // BERT : STREAM
__device__ __forceinline__ uint32_t len_of_kind(int kind) { return kind == 3 ? 25u : 18u; }

__device__ __forceinline__ void emit_record(const RecSink& S, uint32_t& n_run, uint32_t& seq, uint32_t frame_type, int32_t cost,
                                            const uint32_t* col, int stride, int lane, uint32_t len)
{
    if (n_run < S.cap) {
        uint32_t* w = reinterpret_cast<uint32_t*>(S.base + n_run);
        w[0] = S.channel;
        w[1] = seq;
        w[2] = (uint32_t)S.sample_pos;
        w[3] = (uint32_t)(S.sample_pos >> 32);
        w[4] = (uint32_t)cost;
        w[5] = frame_type | (S.sync_type << 8) | (len << 16);
        for (int q = 0; q < 8; ++q) {
            uint32_t v = col[q * stride + lane];
            const int lo = 4 * q;  // zero bytes at and beyond len
            if (lo + 4 > (int)len) v = (lo >= (int)len) ? 0u : (v & (0xFFFFFFFFu >> (8 * (lo + 4 - (int)len))));
            w[6 + q] = v;
        }
        w[14] = 0;
        w[15] = 0;
    } else {
        *S.overflow = 1;
    }
    ++n_run;
    ++seq;
}

// Defer a payload frame (wave decoder): store its LLRs, reserve its record.  Returns the tag that stands for its cost, or 0 if the
// record store is full (the caller then decodes in place so that the overflow path keeps its exact cost).
__device__ __forceinline__ uint32_t defer_frame(const RecSink& S, uint32_t& n_run, uint32_t& seq, uint32_t frame_type, const DecodeLds& L,
                                                int stale401, int wl)
{
    if (n_run >= S.cap) return 0u;
    const uint32_t slot = n_run;
    const M17_LDS uint32_t* llr = as_lds(L.llr);
    uint32_t* dst = S.defer + (size_t)slot * 46;
    int l = wl;
    asm volatile("" : "+v"(l));   // (opaque: the per-lane addresses below are not worth registers across the kernel's main loop)
    if (l < 46) dst[l] = pack_llr_nibbles(llr[2 * l], llr[2 * l + 1]);
    uint32_t* w = reinterpret_cast<uint32_t*>(S.base + slot);
    w[0] = S.channel;
    w[1] = seq;
    w[2] = (uint32_t)S.sample_pos;
    w[3] = (uint32_t)(S.sample_pos >> 32);
    w[4] = DEFER_TAG | slot;
    w[5] = frame_type | (S.sync_type << 8) | (len_of_kind(kind_of_frame_type(frame_type)) << 16);
    for (int q = 6; q < 14; ++q) w[q] = 0;
    w[14] = (uint32_t)stale401;   // depunctured position 401 as this frame sees it (BERT reads it, Q4)
    w[15] = DEFER_MARK;
    ++n_run;
    ++seq;
    return DEFER_TAG | slot;
}

// M17FrameDecoder::operator() (M17FrameDecoder.h:353-392).  Returns the new viterbi_cost (unchanged when the
// reference leaves its by-reference argument untouched).  0xFFFFFFFF stands for size_t(-1).
template <bool WAVE = false>
__device__ __forceinline__ uint32_t decode_frame(const DecodeTables* tb, const DecodeLds& L, int lane, uint32_t sync_type,
                                                 DecoderRegs& D, uint32_t cost_in, const RecSink& S, uint32_t& n_run, uint32_t& seq, int wl = 0)
{
    uint32_t cost = cost_in;
    auto run_viterbi = [&](int kind) {
        if constexpr (WAVE) return viterbi_decode_wave(L, wl, kind, D.stale401);
        else return viterbi_decode(tb, L, lane, kind, D.stale401);
    };
    switch (sync_type) {
    case 0: {  // LSF: decode_lsf :154-178
        D.state = 0;
        cost = run_viterbi(0);
#pragma unroll
        for (int q = 0; q < 8; ++q) L.lsf[q * L.stride + lane] = L.outb[q * L.stride + lane];
        if (crc16_col(L.lsf, L.stride, lane, 30) == 0) {
            const uint32_t b13 = byte_at(L.lsf, L.stride, lane, 13);  // update_state :113-136 on bits 109..111
            const uint32_t bit109 = (b13 >> 2) & 1u, bit110 = (b13 >> 1) & 1u, bit111 = b13 & 1u;
            if (bit111) { if (bit109) D.state = 1; }
            else D.state = (((bit109 << 1) | bit110) == 1u) ? 2u : 3u;
            emit_record(S, n_run, seq, 0 /*LSF*/, (int32_t)cost, L.lsf, L.stride, lane, 30);
        } else {
            D.lich_segments = 0;
#pragma unroll
            for (int q = 0; q < 8; ++q) L.lsf[q * L.stride + lane] = 0;
        }
        break;
    }
    case 1:  // STREAM
        if (D.state == 0) {  // decode_lich :214-262
            uint32_t lich[6] = {0, 0, 0, 0, 0, 0};
            bool ok = true;
            for (int i = 0; i < 4 && ok; ++i) {
                uint32_t cw = 0;
                for (int j = 0; j < 24; ++j) {
                    const uint32_t e = L.lich_src[i * 24 + j];
                    int v = llr_at(L.llr, L.stride, lane, (int)(e & 0x1FFu));
                    if (e & 0x200u) v = -v;
                    cw = (cw << 1) | (v > 0 ? 1u : 0u);
                }
                uint32_t dec = 0;
                if (!golay_decode(tb, cw, dec)) { ok = false; break; }
                dec >>= 12;
                // unpack_lich :181-212 — i = 0: bytes 0,1hi; 1: 1lo,2; 2: 3,4hi; 3: 4lo,5
                if (i == 0) { lich[0] |= (dec >> 4) & 0xFFu; lich[1] = (dec & 0xFu) << 4; }
                else if (i == 1) { lich[1] |= (dec >> 8) & 0xFFu; lich[2] = dec & 0xFFu; }
                else if (i == 2) { lich[3] |= (dec >> 4) & 0xFFu; lich[4] = (dec & 0xFu) << 4; }
                else { lich[4] |= (dec >> 8) & 0xFFu; lich[5] = dec & 0xFFu; }
            }
            if (!ok) break;  // FAIL, cost untouched
            L.outb[0 * L.stride + lane] = lich[0] | (lich[1] << 8) | (lich[2] << 16) | (lich[3] << 24);
            L.outb[1 * L.stride + lane] = lich[4] | (lich[5] << 8);
            emit_record(S, n_run, seq, 1 /*LICH*/, 0, L.outb, L.stride, lane, 6);
            const uint32_t frag = (lich[5] >> 5) & 7u;
            if (frag > 5) { cost = 0xFFFFFFFFu; break; }
            {  // copy 5 bytes into lsf[frag*5 ..]
                for (int k = 0; k < 5; ++k) {
                    const int b = (int)frag * 5 + k;
                    uint32_t w = L.lsf[(b >> 2) * L.stride + lane];
                    w = (w & ~(0xFFu << (8 * (b & 3)))) | (lich[k] << (8 * (b & 3)));
                    L.lsf[(b >> 2) * L.stride + lane] = w;
                }
            }
            D.lich_segments |= (1u << frag);
            if ((D.lich_segments & 0x3Fu) != 0x3Fu) { cost = 0xFFFFFFFFu; break; }
            if (crc16_col(L.lsf, L.stride, lane, 30) == 0) {
                D.lich_segments = 0;
                D.state = 1;
                cost = 0;
                emit_record(S, n_run, seq, 0 /*LSF*/, 0, L.lsf, L.stride, lane, 30);
            } else {
                cost = 128;
            }
        } else if (D.state == 1) {  // decode_stream :276-289
            if constexpr (WAVE) {
                if (S.defer) {
                    const uint32_t tag = defer_frame(S, n_run, seq, 2 /*STREAM*/, L, D.stale401, wl);
                    if (tag) { cost = tag; break; }
                }
            }
            cost = run_viterbi(1);
            emit_record(S, n_run, seq, 2 /*STREAM*/, (int32_t)cost, L.outb, L.stride, lane, 18);
        } else {
            D.state = 0;
        }
        break;
    case 2:  // PACKET: decode_packet :299-315
        if (D.state == 2 || D.state == 3) {
            cost = run_viterbi(2);
            emit_record(S, n_run, seq, D.state == 2 ? 3u : 4u, (int32_t)cost, L.outb, L.stride, lane, 26);
            if (byte_at(L.outb, L.stride, lane, 25) & 0x80u) D.state = 0;
        } else {
            D.state = 0;
        }
        break;
    default:  // BERT: decode_bert :264-274
        D.state = 4;
        if constexpr (WAVE) {
            if (S.defer) {
                const uint32_t tag = defer_frame(S, n_run, seq, 5 /*BERT*/, L, D.stale401, wl);
                if (tag) { cost = tag; break; }
            }
        }
        cost = run_viterbi(3);
        emit_record(S, n_run, seq, 5 /*BERT*/, (int32_t)cost, L.outb, L.stride, lane, 25);
        break;
    }
    return cost;
}

}  // namespace m17
