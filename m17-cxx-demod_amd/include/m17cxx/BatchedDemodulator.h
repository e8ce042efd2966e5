// C++ RAII face of the C ABI (include/m17hip.h) for C independent channels on one GPU.  What the reference does with
// one M17Demodulator<float> object per channel and one operator() call per sample (apps/m17-demod.cpp:455,484-490),
// this does with one context and one run() per block of samples.  Errors become std::runtime_error.
#pragma once

#include "../../../include/m17hip.h"

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <stdexcept>
#include <string>
#include <vector>

namespace mobilinkd
{

class BatchedDemodulator
{
    m17hip_ctx* ctx_ = nullptr;
    uint32_t channels_ = 0, samples_ = 0, room_ = 0, diag_room_ = 0;
    uint64_t last_frames_ = 0;

    static void check(int code, const char* what)
    {
        if (code != M17HIP_OK) throw std::runtime_error(std::string(what) + ": " + m17hip_strerror(code));
    }

public:
    BatchedDemodulator(uint32_t max_channels, uint32_t max_samples, int device = 0)
    {
        // The HIP runtime reads GPU_MAX_HW_QUEUES when it initialises (a context's five streams must not share hardware queues,
        // include/m17hip.h m17hip_advice): ask for 16 unless the host has decided otherwise.  Effective when this is the process's
        // first contact with the GPU — the case of a host like apps/m17-demod.cpp; a host that has already used HIP sets it itself.
        ::setenv("GPU_MAX_HW_QUEUES", "16", /*overwrite*/0);
        check(m17hip_ctx_create(device, max_channels, max_samples, &ctx_), "m17hip_ctx_create");
        static bool warned = false;
        if ((m17hip_advice(ctx_) & M17HIP_ADVICE_HW_QUEUES) && !warned) {
            warned = true;
            std::fprintf(stderr, "m17hip: GPU_MAX_HW_QUEUES is below 8 (M17HIP_FEW_HW_QUEUES_OK was given): the context's streams will share hardware queues and serialise; "
                                 "export GPU_MAX_HW_QUEUES=16 before the process touches the GPU (include/m17hip.h)\n");
        }
    }
    ~BatchedDemodulator() { m17hip_ctx_destroy(ctx_); }
    BatchedDemodulator(const BatchedDemodulator&) = delete;
    BatchedDemodulator& operator=(const BatchedDemodulator&) = delete;

    m17hip_ctx* handle() const { return ctx_; }
    // the context's main stream (a hipStream_t; the library's own, non-blocking): what a host orders its own device work against
    void* stream() const { void* s = nullptr; check(m17hip_get_stream(ctx_, &s), "m17hip_get_stream"); return s; }

    // [channels][samples] int16, row pitch in samples
    void upload(const int16_t* host, uint32_t channels, uint32_t samples, size_t pitch)
    {
        check(m17hip_upload_i16(ctx_, host, channels, samples, pitch), "m17hip_upload_i16");
        channels_ = channels; samples_ = samples;
    }
    void reset() { check(m17hip_demod_reset(ctx_), "m17hip_demod_reset"); }
    void run(uint32_t flags = 0) { check(m17hip_demod_run(ctx_, channels_, samples_, flags), "m17hip_demod_run"); }

    // records of the last run, ordered by (channel, seq): one compaction and one synchronisation when the guessed capacity suffices
    // (M17HIP_ETRUNC reports the real count; the fetch is then repeated once)
    std::vector<m17_frame_rec> frames()
    {
        std::vector<m17_frame_rec> out(last_frames_ + last_frames_ / 4 + 1024);
        uint64_t got = 0;
        int code = m17hip_frames_fetch(ctx_, out.data(), out.size(), &got);
        if (code == M17HIP_ETRUNC && got > out.size()) {
            out.resize(got);
            code = m17hip_frames_fetch(ctx_, out.data(), out.size(), &got);
        }
        check(code, "m17hip_frames_fetch");
        out.resize(got);
        last_frames_ = got;
        return out;
    }
    // Which run's records frames() / packets() name: 0 = the latest run (every run() selects it again), 1 = the run before it — what a live
    // feed asks for after it has queued the next run (m17hip_frames_select).
    void select(uint32_t back) { check(m17hip_frames_select(ctx_, back), "m17hip_frames_select"); }
    // Streaming (include/m17hip.h, m17hip_demod_front): stage the next run's input from pinned host memory while the current run
    // computes, start its front end beside the current run's state-machine half, queue its run() behind the current one, then select(1) and
    // collect the current run's frames() (or, as up to round 5: frames() first, then run()).
    void stage(const int16_t* pinned_host, uint32_t channels, uint32_t samples, size_t pitch)
    {
        check(m17hip_upload_i16_async(ctx_, pinned_host, channels, samples, pitch), "m17hip_upload_i16_async");
        channels_ = channels; samples_ = samples;
    }
    void front(uint32_t flags = 0) { check(m17hip_demod_front(ctx_, channels_, samples_, flags), "m17hip_demod_front"); }
    void stage_wait() { check(m17hip_upload_wait(ctx_), "m17hip_upload_wait"); }
    // BERT statistics per channel (apps/m17-demod.cpp:286-304 + PRBS9): counted over the runs made after enable_bert(true)
    void enable_bert(bool on) { check(m17hip_tune(ctx_, 6, on ? 1 : 0), "m17hip_tune"); }
    std::vector<m17_bert_stat> bert_stats()
    {
        std::vector<m17_bert_stat> st(channels_);
        check(m17hip_bert_stats(ctx_, st.data(), channels_), "m17hip_bert_stats");
        return st;
    }
    // Packet reassembly per channel (decode_packet, apps/m17-demod.cpp:207-253): enable with room for `room` packets per run (0 = off),
    // then packets() returns what the last run completed, ordered by (channel, seq)
    void enable_packets(uint32_t room) { check(m17hip_tune(ctx_, 7, room), "m17hip_tune"); room_ = room; }
    std::vector<m17_packet_rec> packets()
    {
        std::vector<m17_packet_rec> out(room_);
        uint32_t n = 0;
        check(m17hip_packets_fetch(ctx_, out.data(), room_, &n), "m17hip_packets_fetch");
        out.resize(n < room_ ? n : room_);
        return out;
    }
    std::vector<m17_diag> diagnostics()
    {
        std::vector<m17_diag> d(channels_);
        check(m17hip_diag_fetch(ctx_, d.data(), channels_), "m17hip_diag_fetch");
        return d;
    }
    // Every diagnostic callback of a run, not only the last: enable with room for `room` callbacks per channel and run
    // (samples / 384 + 2 is always enough), then diag_log(channel) returns them in stream order; entry.pad[0] | pad[1] << 32
    // is the sample that fired it.
    void enable_diag_log(uint32_t room) { check(m17hip_tune(ctx_, 9, room), "m17hip_tune"); diag_room_ = room; }
    std::vector<m17_diag> diag_log(uint32_t channel = 0)
    {
        std::vector<m17_diag> all((size_t)channels_ * diag_room_);
        std::vector<uint32_t> counts(channels_);
        check(m17hip_diag_log_fetch(ctx_, all.data(), counts.data(), channels_, diag_room_), "m17hip_diag_log_fetch");
        const auto first = all.begin() + (size_t)channel * diag_room_;
        return std::vector<m17_diag>(first, first + counts[channel]);
    }
    // Multi-GPU (one BatchedDemodulator = one shard = one rank): records carry channel = base + local index
    void set_channel_base(uint32_t base) { check(m17hip_set_channel_base(ctx_, base), "m17hip_set_channel_base"); }
    // Evaluation order of the Kalman updates (include/m17hip.h); default 3
    void set_kalman_order(int order) { check(m17hip_set_kalman_order(ctx_, order), "m17hip_set_kalman_order"); }

    // batched counterparts of the reference's operators (parity API)
    std::vector<float> fir(uint32_t flags = 0)  // BaseFirFilter<float,150> with the RRC taps: FirFilter.h:28-43
    {
        std::vector<float> y((size_t)channels_ * samples_);
        check(m17hip_fir_rrc150(ctx_, channels_, samples_, flags, y.data()), "m17hip_fir_rrc150");
        return y;
    }
    void correlator(std::vector<float>& limit, std::vector<float>& corr)  // Correlator::sample/correlate: Correlator.h:43-64
    {
        limit.resize((size_t)channels_ * samples_);
        corr.resize((size_t)4 * channels_ * samples_);
        check(m17hip_correlator(ctx_, channels_, samples_, limit.data(), corr.data()), "m17hip_correlator");
    }
    std::vector<float> dcd_sums(uint32_t flags = 0)  // NSlidingDFT + DataCarrierDetect accumulation: DataCarrierDetect.h:53-58
    {
        std::vector<float> s((size_t)channels_ * (samples_ / 192) * 12);
        uint32_t ticks = 0;
        check(m17hip_dcd(ctx_, channels_, samples_, flags, s.data(), &ticks), "m17hip_dcd");
        return s;
    }
    // Viterbi<Trellis<4,2>,4>::decode<IN,OUT>: kind 0 = <488,240>, 1 = <296,144>, 2 = <420,206>, 3 = <402,197> (Viterbi.h:162-239)
    void viterbi(const int8_t* soft, uint32_t n_frames, int kind, uint8_t* bits, int32_t* cost)
    {
        check(m17hip_viterbi(ctx_, soft, n_frames, kind, bits, cost), "m17hip_viterbi");
    }
};

} // mobilinkd
