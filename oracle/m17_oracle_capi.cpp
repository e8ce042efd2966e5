// ORACLE — TEST INFRASTRUCTURE ONLY (see m17_oracle_dsp.hpp header).
// C entry points (ctypes) over the scalar restatement: per-operator functions
// for the parity tests, the full chain, the synthetic generator, and a
// multi-threaded batch runner used only as bench.py's `cpu_baseline` leg.
#include "m17_oracle_demod.hpp"
#include "m17_oracle_gen.hpp"

#include <atomic>
#include <memory>
#include <thread>

using namespace m17o;

extern "C" {

// 64-byte POD, same layout as include/m17hip.h `m17_frame_rec`.
struct m17o_frame_rec {
    uint32_t channel;
    uint32_t seq;
    uint64_t sample_pos;
    int32_t cost;
    uint8_t frame_type, sync_type, len, flags;
    uint8_t payload[32];
    uint8_t pad[8];
};
static_assert(sizeof(m17o_frame_rec) == 64, "record must be 64 bytes");

struct m17o_diag {  // same layout as include/m17hip.h `m17_diag`
    int32_t dcd;
    float evm, deviation, offset;
    int32_t locked;
    float clock;
    int32_t sample_index, sync_index, clock_index, viterbi_cost;
    float dcd_level;
    uint32_t n_diag;
    uint32_t demod_state, n_frames;
    uint32_t pad[2];
};
static_assert(sizeof(m17o_diag) == 64, "diag must be 64 bytes");

// ---- front-end operators ---------------------------------------------------
void m17o_scale(const int16_t* s, size_t n, int invert, float* out)
{
    for (size_t i = 0; i < n; ++i) out[i] = scale_sample(s[i], invert != 0);
}
void m17o_taps(float* out150) { for (int i = 0; i < 150; ++i) out150[i] = rrc_tap_f(i); }

// ungated FIR over a float stream (state starts at zero)
void m17o_fir_f32(const float* x, size_t n, float* y)
{
    Fir150 f;
    for (size_t i = 0; i < n; ++i) y[i] = f.step(x[i]);
}
void m17o_fir_i16(const int16_t* s, size_t n, int invert, float* y)
{
    Fir150 f;
    for (size_t i = 0; i < n; ++i) y[i] = f.step(scale_sample(s[i], invert != 0));
}
// correlator over a filtered stream: limit[n], corr[4][n] (preamble, lsf, packet, eot raw correlations)
void m17o_correlator(const float* y, size_t n, float* limit, float* corr)
{
    Correlator c;
    const int8_t w[4][8] = {{+3, -3, +3, -3, +3, -3, +3, -3}, {+3, +3, +3, +3, -3, -3, +3, -3},
                            {3, -3, 3, 3, -3, -3, -3, -3}, {+3, +3, +3, +3, +3, +3, -3, +3}};
    for (size_t i = 0; i < n; ++i) {
        c.sample(y[i]);
        limit[i] = c.limit();
        for (int k = 0; k < 4; ++k) corr[k * n + i] = c.correlate(w[k]);
    }
}
// SyncWord trace: which=0 preamble(29), 1 lsf(31,-31), 2 packet(31,-31), 3 eot(31); per sample: timing index, updated
void m17o_syncword(const float* y, size_t n, int which, uint8_t* timing, int8_t* updated, float* trig)
{
    Correlator c;
    SyncWord sw[4] = {SyncWord({+3, -3, +3, -3, +3, -3, +3, -3}, 29.f), SyncWord({+3, +3, +3, +3, -3, -3, +3, -3}, 31.f, -31.f),
                      SyncWord({3, -3, 3, 3, -3, -3, -3, -3}, 31.f, -31.f), SyncWord({+3, +3, +3, +3, +3, +3, -3, +3}, 31.f)};
    SyncWord& s = sw[which];
    for (size_t i = 0; i < n; ++i) {
        c.sample(y[i]);
        trig[i] = s.triggered(c);
        timing[i] = (uint8_t)s.step(c);
        updated[i] = s.updated();
    }
}
void m17o_outer_levels(const float* y, size_t n, size_t si, float* mn, float* mx)
{
    Correlator c;
    for (size_t i = 0; i < n; ++i) c.sample(y[i]);
    c.outer_symbol_levels(si, *mn, *mx);
}
void m17o_dcd_coeffs(float* out4)
{
    Dcd d;
    out4[0] = d.cr[0]; out4[1] = d.ci[0]; out4[2] = d.cr[1]; out4[3] = d.ci[1];
}
// DCD with a fixed update cadence: per update k: level[k], trig[k]
size_t m17o_dcd_trace_cfg(const float* x, size_t n, size_t period, size_t N, size_t f1, size_t f2, float lt, float ht,
                          float* level, uint8_t* trig)
{
    Dcd d(N, f1, f2, lt, ht);
    size_t k = 0;
    for (size_t i = 0; i < n; ++i) {
        d.step(x[i]);
        if ((i + 1) % period == 0) { d.update(); level[k] = d.level(); trig[k] = d.dcd(); ++k; }
    }
    return k;
}
size_t m17o_dcd_trace(const float* x, size_t n, size_t period, float* level, uint8_t* trig)
{
    Dcd d;
    size_t k = 0;
    for (size_t i = 0; i < n; ++i) {
        d.step(x[i]);
        if ((i + 1) % period == 0) { d.update(); level[k] = d.level(); trig[k] = d.dcd(); ++k; }
    }
    return k;
}
// sequential sums of norm(X0), norm(X1) over [start, start+len) with the DFT run from sample 0
void m17o_dcd_sums(const float* x, size_t start, size_t len, float* l1, float* l2)
{
    Dcd d;
    for (size_t i = 0; i < start + len; ++i) {
        if (i == start) { d.level_1 = 0.f; d.level_2 = 0.f; }
        d.step(x[i]);
    }
    *l1 = d.level_1; *l2 = d.level_2;
}
void m17o_evm_trace(const float* sym, size_t n, int do_reset, float* out)
{
    SymbolEvm e;
    if (do_reset) e.reset();
    for (size_t i = 0; i < n; ++i) { e.update(sym[i]); out[i] = e.evm(); }
}
void m17o_llr(const float* sym, size_t n, int8_t* out2n)
{
    for (size_t i = 0; i < n; ++i) llr_table().lookup(sym[i], out2n[2 * i], out2n[2 * i + 1]);
}
void m17o_llr_table(float* edges43, int8_t* l0, int8_t* l1)
{
    const LlrTable& t = llr_table();
    for (int i = 0; i < 43; ++i) { edges43[i] = t.edge[i]; l0[i] = t.l0[i]; l1[i] = t.l1[i]; }
}
// Kalman-based estimators (parity unpinned) — exposed so the HIP path can be compared with them.
// Evaluation order of the blaze expressions of KalmanFilter.h:49-64 (bit set, see m17_oracle_dsp.hpp); process-wide,
// set it before starting a batch.
void m17o_set_kalman_order(int order) { kalman_order() = order & 7; }
int m17o_get_kalman_order(void) { return kalman_order(); }
// One filter, n updates: z[i] after dt[i] samples; wrap = 10 (KalmanFilter<float,10>) or 0 (SymbolKalmanFilter).
// out[i][6] = x0, x1, P00, P01, P10, P11 after update i.
void m17o_kalman_trace(const float* z, const uint32_t* dt, size_t n, int wrap, float z0, float* out)
{
    Kalman2 k;
    k.reset(z0);
    for (size_t i = 0; i < n; ++i) {
        k.update(z[i], dt[i], wrap);
        float* o = out + 6 * i;
        o[0] = k.x[0]; o[1] = k.x[1]; o[2] = k.P[0][0]; o[3] = k.P[0][1]; o[4] = k.P[1][0]; o[5] = k.P[1][1];
    }
}
void m17o_freqdev(const float* mn, const float* mx, size_t n, const uint8_t* reset_before, float* idev, float* offset)
{
    FreqDevEstimator d;
    for (size_t i = 0; i < n; ++i) {
        if (reset_before && reset_before[i]) d.reset();
        d.update(mn[i], mx[i]);
        idev[i] = d.idev(); offset[i] = d.offset();
    }
}
// ops: 0 = reset(index), 1 = update(index) after `count` ticks, 2 = update() after `count` ticks
void m17o_clock(const uint8_t* op, const uint8_t* index, const uint32_t* count, size_t n, uint8_t* sample_index, float* clock_est)
{
    ClockRecovery c;
    for (size_t i = 0; i < n; ++i) {
        for (uint32_t k = 0; k < count[i]; ++k) c.tick();
        if (op[i] == 0) c.reset((float)index[i]);
        else if (op[i] == 1) c.update(index[i]);
        else c.update();
        sample_index[i] = c.sample_index(); clock_est[i] = c.clock_estimate();
    }
}

// ---- FEC operators -----------------------------------------------------------
uint16_t m17o_crc16(const uint8_t* d, size_t n) { return crc16_m17(d, n); }
void m17o_decode_callsign(const uint8_t* enc6, char* out10) { decode_callsign(enc6, out10); }
void m17o_encode_callsign(const char* call, uint8_t* out6) { encode_callsign(std::string(call), out6); }
uint32_t m17o_golay_encode24(uint16_t v) { return golay::encode24(v); }
int m17o_golay_decode(uint32_t in, uint32_t* out) { return golay::decode(in, *out) ? 1 : 0; }
void m17o_interleave(int8_t* f368) { interleave(f368); }
void m17o_deinterleave(int8_t* f368) { deinterleave(f368); }
void m17o_derandomize(int8_t* f368) { derandomize(f368); }
void m17o_randomize_bits(int8_t* f368) { randomize_bits(f368); }
size_t m17o_qpp(size_t i) { return qpp(i); }
size_t m17o_puncture(const uint8_t* in, size_t IN, int8_t* out, size_t OUT, int which) { return puncture(in, IN, out, OUT, punct_matrix(which)); }
size_t m17o_depuncture(const int8_t* in, size_t IN, int8_t* out, size_t OUT, int which) { return depuncture(in, IN, out, OUT, punct_matrix(which)); }
size_t m17o_conv_encode(const uint8_t* bits, size_t n, uint8_t* out) { return conv_encode_bits(bits, n, out); }
size_t m17o_viterbi(const int8_t* in, size_t IN, uint8_t* out, size_t OUT, int llr_bits)
{
    Viterbi v(llr_bits);
    return v.decode(in, IN, out, OUT);
}
void m17o_viterbi_tables(int16_t* cost32, uint8_t* prev32)
{
    Viterbi v(4);
    for (int i = 0; i < 16; ++i) for (int j = 0; j < 2; ++j) { cost32[i * 2 + j] = v.cost[i][j]; prev32[i * 2 + j] = v.prevState[i][j]; }
}
void m17o_to_bytes(const uint8_t* bits, size_t n, uint8_t* out) { to_bytes(bits, n, out); }
void m17o_prbs9(uint16_t* state, uint8_t* bits, size_t n)
{
    Prbs9 p; p.state = *state;
    for (size_t i = 0; i < n; ++i) bits[i] = p.generate();
    *state = p.state;
}
// BER over 25-byte BERT payloads, as apps/m17-demod.cpp:287-305 feeds PRBS9::validate
void m17o_bert_count(const uint8_t* payloads25, size_t n_frames, uint32_t* bits, uint32_t* errs, int* synced)
{
    Prbs9 p;
    for (size_t f = 0; f < n_frames; ++f) {
        const uint8_t* b = payloads25 + 25 * f;
        for (int j = 0; j < 24; ++j) { uint8_t v = b[j]; for (int i = 0; i < 8; ++i) { p.validate(v & 0x80); v <<= 1; } }
        uint8_t v = b[24];
        for (int i = 0; i < 5; ++i) { p.validate(v & 0x80); v <<= 1; }
    }
    *bits = p.bit_count; *errs = p.err_count; *synced = p.synced;
}

// Packet reassembly consumer: decode_packet (apps/m17-demod.cpp:207-253) with dump_lsf's reset (:154-155) over the callback
// records of ONE channel in order (types[n], payloads[n][32]).  `cur`/`st` carry current_packet and {frame counter, sequence
// errors, frames accepted} between calls (st[0] = current_packet.size()).  Per closed packet k: size[k], checksum[k],
// frames[k], seq_errors[k], rec_index[k] and data[k][840].  The checksum is CRC-16/X.25 (boost::crc_optimal<16, 0x1021,
// 0xFFFF, 0xFFFF, true, true>, :218 — boost is absent, restated from the published parameters; "123456789" -> 0x906E).
// dump_lsf's ENCAPSULATED branch (:157-171) indexes a 30-byte array at 109..111 (out of bounds) and is not modelled.
size_t m17o_packet_reassemble(const uint8_t* types, const uint8_t* payloads32, size_t n, uint8_t* cur, uint32_t* st, size_t cap,
                              uint16_t* size, uint16_t* checksum, uint8_t* frames, uint8_t* seq_errors, uint32_t* rec_index, uint8_t* data840)
{
    std::vector<uint8_t> current_packet(cur, cur + st[0]);
    size_t packet_frame_counter = st[1];
    uint32_t errs = st[2], nfr = st[3];
    size_t done = 0;
    for (size_t r = 0; r < n; ++r) {
        const uint8_t* seg = payloads32 + 32 * r;
        if (types[r] == (uint8_t)FrameType::LSF) { current_packet.clear(); packet_frame_counter = 0; errs = 0; nfr = 0; continue; }
        if (types[r] != (uint8_t)FrameType::BASIC_PACKET && types[r] != (uint8_t)FrameType::FULL_PACKET) continue;
        if (seg[25] & 0x80) {
            size_t packet_size = (seg[25] & 0x7F) >> 2;
            packet_size = std::min(packet_size, size_t(25));
            for (size_t i = 0; i != packet_size; ++i) current_packet.push_back(seg[i]);
            ++nfr;
            const uint16_t sum = crc16_x25(current_packet.data(), current_packet.size());
            if (done < cap) {
                size[done] = (uint16_t)current_packet.size(); checksum[done] = sum; frames[done] = (uint8_t)nfr;
                seq_errors[done] = (uint8_t)std::min<uint32_t>(errs, 255); rec_index[done] = (uint32_t)r;
                std::memset(data840 + 840 * done, 0, 840);
                std::memcpy(data840 + 840 * done, current_packet.data(), std::min<size_t>(current_packet.size(), 840));
            }
            ++done;
            continue;
        }
        const size_t frame_number = (seg[25] & 0x7F) >> 2;
        if (frame_number != packet_frame_counter) { ++errs; continue; }
        packet_frame_counter += 1; ++nfr;
        for (size_t i = 0; i != 25; ++i) current_packet.push_back(seg[i]);
    }
    st[0] = (uint32_t)current_packet.size(); st[1] = (uint32_t)packet_frame_counter; st[2] = errs; st[3] = nfr;
    std::memcpy(cur, current_packet.data(), std::min<size_t>(current_packet.size(), 832));
    return done;
}
uint16_t m17o_crc16_x25(const uint8_t* d, size_t n) { return crc16_x25(d, n); }

// One frame through the frame decoder.  state_io: decoder state in/out; lich_io: lich_segments;
// lsf_io[30]; dep401_io: the stale depuncture byte (Q4).  Returns number of callbacks (0..2) written to recs.
struct VecSink {
    std::vector<FrameOut>* v;
    void operator()(const FrameOut& f) const { v->push_back(f); }
};
int m17o_decode_frame(int sync_type, const int8_t* llr368, uint8_t* state_io, uint8_t* lich_io, uint8_t* lsf_io,
                      int8_t* dep401_io, int64_t* cost_io, m17o_frame_rec* recs)
{
    std::vector<FrameOut> outs;
    FrameDecoder<VecSink> d{VecSink{&outs}};
    d.state_ = (DecState)*state_io; d.lich_segments = *lich_io;
    std::memcpy(d.lsf, lsf_io, 30); d.dep[401] = *dep401_io;
    int8_t buf[368];
    std::memcpy(buf, llr368, 368);
    size_t cost = (size_t)*cost_io;
    d.run((SyncType)sync_type, buf, cost);
    *state_io = (uint8_t)d.state_; *lich_io = d.lich_segments;
    std::memcpy(lsf_io, d.lsf, 30); *dep401_io = d.dep[401]; *cost_io = (int64_t)cost;
    int n = 0;
    for (auto& f : outs) {
        m17o_frame_rec& r = recs[n++];
        std::memset(&r, 0, sizeof(r));
        r.cost = f.cost; r.frame_type = (uint8_t)f.type; r.sync_type = (uint8_t)sync_type; r.len = f.len;
        std::memcpy(r.payload, f.data, 30);
    }
    return n;
}

// ---- full chain ----------------------------------------------------------------
static size_t run_channel(const int16_t* s, size_t n, int invert, uint32_t channel, m17o_frame_rec* recs, size_t cap,
                          m17o_diag* diag, float* sym_out, size_t sym_cap, size_t* n_sym)
{
    return run_channel_t<Demodulator>(s, n, invert, channel, recs, cap, diag, sym_out, sym_cap, n_sym);
}

size_t m17o_demod(const int16_t* s, size_t n, int invert, m17o_frame_rec* recs, size_t cap, m17o_diag* diag)
{
    return run_channel(s, n, invert, 0, recs, cap, diag, nullptr, 0, nullptr);
}
size_t m17o_demod_symbols(const int16_t* s, size_t n, int invert, float* sym_out, size_t sym_cap)
{
    size_t ns = 0;
    run_channel(s, n, invert, 0, nullptr, 0, nullptr, sym_out, sym_cap, &ns);
    return ns;
}
// Every diagnostic callback of one channel, in order (same layout as the log entries of m17hip_diag_log_fetch: demod_state and
// n_frames at that moment, pad[0] | pad[1] << 32 = the sample that fired it).  Returns the number of callbacks.
size_t m17o_demod_diag_log(const int16_t* s, size_t n, int invert, m17o_diag* log, size_t cap)
{
    return diag_log_t<Demodulator>(s, n, invert, log, cap);
}

// Batch: samples[C][T] (row pitch = pitch samples); recs[C][cap]; counts[C]; diags[C].  `threads` host threads.
void m17o_demod_batch(const int16_t* s, size_t C, size_t T, size_t pitch, int invert, int threads, m17o_frame_rec* recs,
                      size_t cap, uint32_t* counts, m17o_diag* diags)
{
    std::atomic<size_t> next{0};
    auto worker = [&]() {
        for (;;) {
            size_t c = next.fetch_add(1);
            if (c >= C) break;
            size_t n = run_channel(s + c * pitch, T, invert, (uint32_t)c, recs ? recs + c * cap : nullptr, recs ? cap : 0,
                                   diags ? diags + c : nullptr, nullptr, 0, nullptr);
            if (counts) counts[c] = (uint32_t)n;
        }
    };
    if (threads <= 1) { worker(); return; }
    std::vector<std::thread> th;
    for (int i = 0; i < threads; ++i) th.emplace_back(worker);
    for (auto& t : th) t.join();
}

// ---- generator -------------------------------------------------------------------
struct m17o_gen_params {
    uint64_t seed;
    int32_t kind, n_frames, lead_in, phase, tail, total, invert, n_preamble;
    double lead_sigma, noise_sigma, dc_offset, gain, tail_sigma;
};
static GenParams to_params(const m17o_gen_params* q)
{
    GenParams p;
    p.seed = q->seed; p.kind = q->kind; p.n_frames = q->n_frames; p.lead_in = q->lead_in; p.phase = q->phase;
    p.tail = q->tail; p.total = q->total; p.invert = q->invert; p.n_preamble = q->n_preamble; p.lead_sigma = q->lead_sigma;
    p.noise_sigma = q->noise_sigma; p.dc_offset = q->dc_offset; p.gain = q->gain; p.tail_sigma = q->tail_sigma;
    return p;
}
// Returns the natural length; writes min(len, cap) samples.  payloads: n_frames x 32 bytes (optional), lsf30 optional.
size_t m17o_generate(const m17o_gen_params* q, int16_t* out, size_t cap, uint8_t* payloads, uint8_t* lsf30, int32_t* burst_start)
{
    GenTruth tr;
    std::vector<int16_t> v = generate(to_params(q), &tr);
    size_t n = std::min(cap, v.size());
    if (out) std::memcpy(out, v.data(), n * sizeof(int16_t));
    if (payloads)
        for (size_t i = 0; i < tr.payloads.size(); ++i) {
            std::memset(payloads + 32 * i, 0, 32);
            std::memcpy(payloads + 32 * i, tr.payloads[i].data(), tr.payloads[i].size());
        }
    if (lsf30) std::memcpy(lsf30, tr.lsf, 30);
    if (burst_start) *burst_start = tr.burst_start;
    return v.size();
}
// The on-air 368-bit frames (after interleave + randomize) of a synthetic burst, with the sync type of each:
// kind 0: BERT x n; kind 1: LSF + n stream; kind 2: LSF + n packet.  Returns the number of frames written.
size_t m17o_make_frames(int kind, uint64_t seed, int n_frames, int8_t* bits /* [n+1][368] */, uint8_t* sync_types)
{
    uint64_t rs = splitmix64(seed ^ 0xA5A5A5A5ull);
    auto rnd = [&rs]() { rs = splitmix64(rs); return rs; };
    size_t n = 0;
    uint8_t lsf[30];
    if (kind == 0) {
        Prbs9 prbs;
        for (int i = 0; i < n_frames; ++i) { uint8_t pl[25]; bert_frame_bits(prbs, bits + 368 * n, pl); sync_types[n++] = 3; }
    } else if (kind == 1) {
        uint8_t can = (uint8_t)(rnd() & 15);
        make_lsf("N0CALL", "", (uint16_t)(((can >> 1) << 8) | (5 | ((can & 1) << 7))), lsf);
        lsf_frame_bits(lsf, bits + 368 * n); sync_types[n++] = 0;
        uint8_t lich[6][96];
        for (uint8_t i = 0; i < 6; ++i) lich_segment_bits(lsf + 5 * i, i, lich[i]);
        for (int i = 0; i < n_frames; ++i) {
            uint8_t pl[16];
            for (int k = 0; k < 16; k += 8) { uint64_t r = rnd(); std::memcpy(pl + k, &r, 8); }
            uint16_t fn = (uint16_t)(i & 0x7FFF);
            if (i == n_frames - 1) fn |= 0x8000;
            stream_frame_bits(lich[i % 6], fn, pl, bits + 368 * n); sync_types[n++] = 1;
        }
    } else {
        make_lsf("N0CALL", "", (uint16_t)0x0002, lsf);
        lsf_frame_bits(lsf, bits + 368 * n); sync_types[n++] = 0;
        for (int i = 0; i < n_frames; ++i) {
            uint8_t d[26];
            for (int k = 0; k < 24; k += 8) { uint64_t r = rnd(); std::memcpy(d + k, &r, 8); }
            d[24] = (uint8_t)rnd();
            bool last = (i == n_frames - 1);
            d[25] = (uint8_t)((last ? 0x80 : 0x00) | ((last ? 25 : i) << 2));
            packet_frame_bits(d, bits + 368 * n); sync_types[n++] = 2;
        }
    }
    return n;
}

// Batch generator for benches: channel c uses seed base.seed ^ splitmix64(c), kind alternates per `kind_mask`.
void m17o_generate_batch(const m17o_gen_params* base, size_t C, size_t T, size_t pitch, int threads, int16_t* out, uint32_t chan0)
{
    std::atomic<size_t> next{0};
    auto worker = [&]() {
        for (;;) {
            size_t c = next.fetch_add(1);
            if (c >= C) break;
            m17o_gen_params q = *base;
            uint64_t cc = (uint64_t)chan0 + c;
            q.seed = base->seed ^ splitmix64(cc * 0x9E3779B97F4A7C15ull + 1);
            if (base->kind < 0) q.kind = (int)(cc % 2);  // mixed: even BERT, odd voice-like
            q.total = (int)T;
            GenParams p = to_params(&q);
            std::vector<int16_t> v = generate(p, nullptr);
            std::memcpy(out + c * pitch, v.data(), T * sizeof(int16_t));
        }
    };
    if (threads <= 1) { worker(); return; }
    std::vector<std::thread> th;
    for (int i = 0; i < threads; ++i) th.emplace_back(worker);
    for (auto& t : th) t.join();
}

}  // extern "C"
