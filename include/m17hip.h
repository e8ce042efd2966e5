/* m17hip.h — C ABI of the MI355X-native M17 4-FSK demodulation hot path.
 *
 * The reference (mobilinkd/m17-cxx-demod) has no FFI layer: its hot path is the header-only
 * operator surface in include/m17cxx driven one sample at a time by apps/m17-demod.cpp:484-490.
 * This library is what a batched replacement of that path binds to: plain pointers and sizes,
 * no C++ or torch types, every function returns 0 on success or a negative M17HIP_E* code
 * (m17hip_strerror), no exceptions cross the boundary.  One context per (host thread, GPU).
 * All channels of a context advance together: a "run" consumes T new samples of each of C
 * independent 48 kSPS int16 baseband channels and continues from the state the previous run left
 * (m17hip_demod_reset starts over), exactly as C fresh reference processes would.
 *
 * Each entry point names the reference interface it replaces (file:line in /root/reference).
 */
#ifndef M17HIP_H
#define M17HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define M17HIP_OK 0
#define M17HIP_EINVAL (-1)   /* bad argument / size beyond what the context was created for */
#define M17HIP_EHIP (-2)     /* a HIP runtime call failed (m17hip_last_hip_error) */
#define M17HIP_ENOMEM (-3)
#define M17HIP_ESTATE (-4)   /* call sequence error (e.g. fetch before run) */
#define M17HIP_EOVERFLOW (-5) /* a per-channel frame-record buffer overflowed */
#define M17HIP_ETRUNC (-6)   /* more results than the caller's capacity: *count says how many exist, `capacity` were written */
#define M17HIP_ECOMM (-7)    /* an RCCL call failed (m17hip_gather_*) */
#define M17HIP_ECONFIG (-8)  /* m17hip_ctx_create: the process runs with fewer than 8 hardware queues (GPU_MAX_HW_QUEUES; see m17hip_advice) */
/* (-8 was M17HIP_ETIMEOUT of ABI 301's persistent kernel form, removed in ABI 400) */

/* Frame-type / sync-type codes = the reference enums M17FrameDecoder.h:52-55. */
enum { M17_FRAME_LSF = 0, M17_FRAME_LICH = 1, M17_FRAME_STREAM = 2, M17_FRAME_BASIC_PACKET = 3, M17_FRAME_FULL_PACKET = 4, M17_FRAME_BERT = 5 };
enum { M17_SYNC_LSF = 0, M17_SYNC_STREAM = 1, M17_SYNC_PACKET = 2, M17_SYNC_BERT = 3 };

/* One record per invocation of the reference's frame callback
 * `bool(const output_buffer_t&, int viterbi_cost)` (M17FrameDecoder.h:98, called at :171,221,253,271,286,306).
 * payload = lsf[30] | lich[6] | stream[18] | packet[26] | bert[25] according to frame_type. */
typedef struct m17_frame_rec {
    uint32_t channel;
    uint32_t seq;        /* per-channel callback counter since the last reset */
    uint64_t sample_pos; /* 0-based index (since reset) of the input sample whose processing fired the callback */
    int32_t cost;        /* the callback's viterbi_cost argument */
    uint8_t frame_type;  /* M17_FRAME_* */
    uint8_t sync_type;   /* M17_SYNC_* that selected the decode */
    uint8_t len;         /* valid payload bytes */
    uint8_t flags;
    uint8_t payload[32];
    uint8_t pad[8];
} m17_frame_rec; /* 64 bytes */

/* Arguments of the most recent diagnostic callback (M17Demodulator.h:144, fired at :681-685 and :746-750),
 * plus the demodulator state at the end of the run. */
typedef struct m17_diag {
    int32_t dcd;
    float evm, deviation, offset;
    int32_t locked;
    float clock;
    int32_t sample_index, sync_index, clock_index, viterbi_cost;
    float dcd_level;
    uint32_t n_diag;
    uint32_t demod_state; /* M17Demodulator.h:146 DemodState */
    uint32_t n_frames;    /* frame callbacks since reset */
    uint32_t pad[2];
} m17_diag; /* 64 bytes */

typedef struct m17hip_ctx m17hip_ctx;

#define M17HIP_FLAG_INVERT 1u /* apps/m17-demod.cpp:488 `if (invert_input) sample *= -1`; the only flag: any other bit is M17HIP_EINVAL */

const char* m17hip_strerror(int code);
int m17hip_last_hip_error(const m17hip_ctx* ctx);
int m17hip_version(void);

/* Context: device slabs for `max_channels` x `max_samples` (per run).  Replaces constructing one
 * M17Demodulator<float> per channel (apps/m17-demod.cpp:455, M17Demodulator.h:180-182).
 * max_samples <= M17HIP_MAX_SAMPLES_PER_RUN (11.6 minutes of 48 kSPS in ONE run; a stream of any length is continued run after run):
 * the kernels address a channel row — sixteen rows in the limit-filter replay — with 32-bit byte offsets.  Beyond it: M17HIP_EINVAL. */
#define M17HIP_MAX_SAMPLES_PER_RUN 33553152u
int m17hip_ctx_create(int device, uint32_t max_channels, uint32_t max_samples, m17hip_ctx** out);
void m17hip_ctx_destroy(m17hip_ctx* ctx);
/* Hardware queues.  A context uses five streams (main, matched filter, carrier detect, limit-filter replay, copy); with streams
 * sharing a hardware queue a kernel queued behind another stream's event wait waits with it, and the overlap the streams exist for
 * is lost (measured: two contexts 42 instead of 26 ms per step).  The HIP runtime reads GPU_MAX_HW_QUEUES when it initialises (its
 * default is 4): export GPU_MAX_HW_QUEUES=16 before the first HIP call of the process (the Python binding and the C++ host classes
 * set it at load time when it is unset — effective when that is before the process's first contact with the GPU).
 * The slow state is never entered silently: with the variable unset or below 8 m17hip_ctx_create returns M17HIP_ECONFIG, unless the
 * host says it accepts that (environment variable M17HIP_FEW_HW_QUEUES_OK=1).
 * m17hip_advice (bit set; 0 = nothing to say): bit 0 (M17HIP_ADVICE_HW_QUEUES) fewer than 8 queues were asked for, bit 1
 * (M17HIP_ADVICE_HW_QUEUES_16) fewer than 16 — enough for one context, not for two batches in flight or a stream in two groups. */
#define M17HIP_ADVICE_HW_QUEUES 1
#define M17HIP_ADVICE_HW_QUEUES_16 2
int m17hip_advice(const m17hip_ctx* ctx);
/* Operational statistic: how many times since the last m17hip_demod_reset a channel left the limit-filter replay (the demodulator
 * forced dcd.unlock() after losing sync, M17Demodulator.h:396-404, 470-478: the one gate event the replay cannot foresee) and computed
 * its own limit-filter history up to the end of the next segment — the slow path of the sequential kernel.  Waits for the queued work. */
int m17hip_replay_drops(m17hip_ctx* ctx, uint64_t* count);
/* The context's main stream (a hipStream_t).  From round 6 on the library creates it — together with the context's other streams, in one go and in a
 * fixed role order; a destroyed context's streams are parked as a set and handed to the next context of that device — because which streams share a
 * hardware pipe decides 10-40 % of a continued stream's step time and the runtime maps streams to pipes in creation order: a process that creates and
 * destroys contexts around a host's own main streams walks through the slow layouts (24 -> 26 -> 32 ms per step over three create / use / destroy
 * cycles, tools/stream_history.py; 22.1-22.5 ms in every cycle with the library's own).  It is a NON-BLOCKING stream: it does not synchronise with the
 * default stream.  A host that produces input on, or consumes device-side results from, a stream of its own orders the two with events (or, in
 * PyTorch, wraps this one: torch.cuda.ExternalStream(handle)); results fetched into host memory are complete when the fetch returns, as ever. */
int m17hip_get_stream(m17hip_ctx* ctx, void** hip_stream);
/* For hosts that insist: launch all work of this context on `hip_stream` instead (a hipStream_t; NULL = the default stream — what a context ran on
 * up to round 5 when this was not called).  The library's main stream stays with the context, idle.  M17HIP_STREAM_SETS=0 in the environment restores
 * the older behaviour altogether (streams per context, destroyed with it, main = the default stream): a diagnostic, not a deployment mode. */
int m17hip_set_stream(m17hip_ctx* ctx, void* hip_stream);

/* Input: [C][T] int16, row pitch in samples.  Replaces the stdin read loop apps/m17-demod.cpp:484-488. */
int m17hip_upload_i16(m17hip_ctx* ctx, const int16_t* host, uint32_t channels, uint32_t samples, size_t pitch);
/* Same, from a DEVICE pointer (e.g. a tensor that already lives in HBM); the copy is complete when the call returns, so
 * `dev` may be freed or refilled at once. */
int m17hip_upload_i16_device(m17hip_ctx* ctx, const int16_t* dev, uint32_t channels, uint32_t samples, size_t pitch);
/* Streaming ingest: stage the input of the NEXT m17hip_demod_run in a second slab while the current run is
 * computing.  The copy is queued on the context's own copy stream and starts as soon as the run before the current one has
 * released that slab; it should come from pinned memory (hipHostMalloc / hipHostRegister) to overlap.  The next
 * m17hip_demod_run (same channel / sample counts) makes the device wait for the copy, swaps the slabs (the carried 152-sample
 * tail is moved over) and runs on the staged input.  (The first staging call of a context allocates the second set of per-run
 * slabs: input, matched-filter output, limit-filter history, carrier-detect table.)  m17hip_demod_run does NOT wait on the host: the copy may still be in
 * flight when it returns.  The host buffer must stay valid AND unmodified until m17hip_upload_wait has returned (or until a
 * frames / diag fetch of the run that consumed it has returned — those synchronise with the run, which waited for the copy). */
int m17hip_upload_i16_async(m17hip_ctx* ctx, const int16_t* host, uint32_t channels, uint32_t samples, size_t pitch);
/* Block until the copy queued by the last m17hip_upload_i16_async / m17hip_upload_i16_device_async has left its source buffer. */
int m17hip_upload_wait(m17hip_ctx* ctx);
/* The same staging from DEVICE memory (a producer on the GPU hands a chunk over): a device-to-device copy on the context's copy
 * stream, NOT complete when the call returns — `dev` must stay valid and unmodified until m17hip_upload_wait has returned.  The copy
 * stream does not wait for the producer's stream: `dev` must be COMPLETE when the call is made (the producer synchronised, or the hand-over
 * ordered by the caller). */
int m17hip_upload_i16_device_async(m17hip_ctx* ctx, const int16_t* dev, uint32_t channels, uint32_t samples, size_t pitch);
/* Staging without a copy: declares that the context's second input slab — the one the run BEFORE the latest run consumed, or
 * that an earlier m17hip_upload_i16_async filled — already holds the next run's `channels` x `samples` input (two resident slabs
 * that alternate, e.g. a replayed capture).  M17HIP_ESTATE if that slab holds no input of this shape. */
int m17hip_input_alternate(m17hip_ctx* ctx, uint32_t channels, uint32_t samples);

/* ---- per-operator batched entry points (config 2 parity) ---------------------------------------- */
/* K1: sample scaling + BaseFirFilter<float,150> with the RRC taps, ungated, over the uploaded slab
 * (apps/m17-demod.cpp:489 scaling; FirFilter.h:28-43; taps M17Demodulator.h:79-118).
 * out_host may be NULL (result stays on the device for the next operator). out: [C][T] float. */
int m17hip_fir_rrc150(m17hip_ctx* ctx, uint32_t channels, uint32_t samples, uint32_t flags, float* out_host);
/* K2: Correlator::sample (limit_ = IIR LPF of |y|) and Correlator::correlate against the four M17 sync words
 * (preamble, LSF, packet, EOT — M17Demodulator.h:154-157) for every sample of the FIR output left on the
 * device by m17hip_fir_rrc150 (Correlator.h:43-64, IirFilter.h:26-42).  limit: [C][T]; corr: [4][C][T]. */
int m17hip_correlator(m17hip_ctx* ctx, uint32_t channels, uint32_t samples, float* limit_host, float* corr_host);
/* K1 + K2 in one call (BASELINE configs[1]: "FIR + Correlator only"): the matched filter, the limit filter and the four correlations over
 * the uploaded slab, pipelined in time — the limit filter is one dependent chain per channel over the whole run and is what the call
 * lasts; the matched filter of the next piece and the correlations of this one run beside it.  Results are identical to
 * m17hip_fir_rrc150 followed by m17hip_correlator (FirFilter.h:28-43, Correlator.h:38-64, IirFilter.h:26-42).
 * y: [C][T], limit: [C][T], corr: [4][C][T]; any of the host pointers may be NULL (the result stays on the device). */
int m17hip_fir_correlator(m17hip_ctx* ctx, uint32_t channels, uint32_t samples, uint32_t flags, float* y_host, float* limit_host, float* corr_host);
/* K3: NSlidingDFT<float,48000,120,2> + DataCarrierDetect accumulation (SlidingDFT.h:118-132,
 * DataCarrierDetect.h:53-58) over the uploaded slab.  For every 192-sample tick k the table holds the
 * sequential sums of norm(X0) (bin 0) and norm(X1) (bin 1) for segments that started 1..5 ticks ago (index a%5,
 * a = start tick) and since the stream start (index 5): sums[C][ticks][2][6]. */
int m17hip_dcd(m17hip_ctx* ctx, uint32_t channels, uint32_t samples, uint32_t flags, float* sums_host, uint32_t* ticks_out);
/* K4: Viterbi<Trellis<4,2>,4>::decode (Viterbi.h:162-239) on n depunctured soft-bit frames.
 * kind: 0 = LSF 488->240, 1 = stream 296->144, 2 = packet 420->206, 3 = BERT 402->197.
 * soft: [n][IN] int8 (0 = erasure); bits: [n][OUT] uint8; cost: [n]. */
int m17hip_viterbi(m17hip_ctx* ctx, const int8_t* soft_host, uint32_t n_frames, int kind, uint8_t* bits_host, int32_t* cost_host);
/* a11/a12: llr<float,4> (Util.h:63-104,128-145) and SymbolEvm::update (SymbolEvm.h:31-51, after reset()) on `rows`
 * independent sequences of n normalised symbols.  llr: [rows][n][2] int8; evm: [rows][n] running evm(). */
int m17hip_slice_llr(m17hip_ctx* ctx, const float* sym_host, uint32_t rows, uint32_t n, int8_t* llr_host, float* evm_host);
/* K4': M17FrameDecoder::operator() (M17FrameDecoder.h:353-392: derandomize, deinterleave, depuncture,
 * Viterbi / Golay, CRC, frame-type state machine) on n independent 368-LLR frames, each with its own decoder
 * state in/out.  sync_type[n]; state_io[n] (State enum), lich_io[n], lsf_io[n][30], dep401_io[n], cost_io[n];
 * recs: [n][2] records, nrec[n] callbacks per frame. */
int m17hip_decode_frames(m17hip_ctx* ctx, const int8_t* llr368_host, uint32_t n_frames, const uint8_t* sync_type,
                         uint8_t* state_io, uint8_t* lich_io, uint8_t* lsf_io, int8_t* dep401_io, int64_t* cost_io,
                         m17_frame_rec* recs, uint8_t* nrec);

/* ---- the full chain ------------------------------------------------------------------------------- */
/* Fresh demodulators for every channel (M17Demodulator ctor + zero-initialised storage). */
int m17hip_demod_reset(m17hip_ctx* ctx);
/* M17Demodulator<float>::operator() (M17Demodulator.h:657-753) for `samples` new samples of each channel of
 * the uploaded slab; frame callbacks become records, the last diagnostic callback becomes m17_diag. */
int m17hip_demod_run(m17hip_ctx* ctx, uint32_t channels, uint32_t samples, uint32_t flags);
/* Streaming: pipelining of consecutive runs of the SAME channels.  apps/m17-demod.cpp:484-490 is one endless
 * stream per channel; here the stream arrives in runs, and most of a run's arithmetic does not depend on how the run before it
 * ended: the matched filter (FirFilter.h:28-43) needs only the carried 152-sample input tail, the sliding DFT
 * (SlidingDFT.h:118-132) only its own end state.  With the next run's input STAGED (m17hip_upload_i16_async,
 * m17hip_upload_i16_device_async or m17hip_input_alternate), m17hip_demod_front swaps the context's two sets of per-run slabs
 * and queues that front end at once, on the context's side streams — while the state-machine half of the latest run (limit
 * filter, clock recovery, slicer, Viterbi: everything that waits for M17Demodulator's state) is still at work.  The call
 * sequence of a live feed:
 *     stage(k + 1); m17hip_demod_front(k + 1); m17hip_demod_run(k + 1);
 *     m17hip_frames_select(ctx, 1); m17hip_frames_fetch / _compact_device / m17hip_gather_frames (run k);  ...
 * i.e. the state-machine half of run k + 1 is queued BEFORE the host collects run k's records: nothing of run k + 1 waits for what only
 * run k's consumers need.  A run ends, on the context's main stream, with its demodulator state settled (M17Demodulator.h:146-176:
 * everything operator() reads the next time); the payload frames whose Viterbi decode K5 deferred, the payload consumers (BERT
 * statistics, packet reassembly), the compaction of the records and the gather work on the context's PAYLOAD stream, beside the next
 * run, on the record set of their own run — two record sets alternate run by run, so a run's records stay fetchable until the run
 * after the next is queued.  (The older order — fetch run k between m17hip_demod_front(k + 1) and m17hip_demod_run(k + 1) — works as
 * before; the next run's chain then starts a host round trip and the deferred decode, 2 ms per 4096 x 480 000, later.)
 * The m17hip_demod_run that follows m17hip_demod_front must name the same channels / samples / flags (M17HIP_ESTATE otherwise) and
 * queues the rest.  Between the two calls the context's results are still those of run k; in-place uploads, per-operator entry points and
 * m17hip_tune return M17HIP_ESTATE; m17hip_demod_reset abandons the queued front end.  Results are bit-identical to the same runs
 * made one after the other (tests/test_gpu_streaming.py).  M17HIP_ESTATE if nothing is staged.
 * (m17hip_demod_run on staged input without this call queues the same front end itself — then nothing is gained unless the host
 * calls it before it has fetched the previous run's records, which it thereby gives up.) */
int m17hip_demod_front(m17hip_ctx* ctx, uint32_t channels, uint32_t samples, uint32_t flags);
/* Which run's records m17hip_frames_count / _fetch / _compact_device and m17hip_gather_frames[_device] name: back = 0 the latest run
 * (the default; every m17hip_demod_run selects it again), back = 1 the run before it — what a live feed asks for after it has queued
 * the next run (above).  M17HIP_ESTATE if that run's records are not there (no such run since the last reset). */
int m17hip_frames_select(m17hip_ctx* ctx, uint32_t back);
/* Number of records produced by the selected run (all channels).  The calls of this family wait for the run's payload work only —
 * not for anything queued after it. */
int m17hip_frames_count(m17hip_ctx* ctx, uint64_t* total);
/* Records of the last run, ordered by (channel, seq).  Host destination.  *count = records the run produced; when that is
 * more than `capacity` only `capacity` are written and M17HIP_ETRUNC is returned. */
int m17hip_frames_fetch(m17hip_ctx* ctx, m17_frame_rec* recs_host, uint64_t capacity, uint64_t* count);
/* Same, compacted into caller-provided DEVICE memory (so a collective can ship it without a host hop); same
 * truncation rule.  The copy is made on the context's payload stream and is complete when the call returns.  For the latest run it is
 * ordered behind whatever the caller has queued on the context's main stream so far (a fill of the destination, say); with a newer run
 * queued (m17hip_frames_select(ctx, 1)) it waits for nothing but the selected run: the destination must be ready when the call is made.
 * The same holds for the device destination of m17hip_gather_frames_device. */
int m17hip_frames_compact_device(m17hip_ctx* ctx, m17_frame_rec* recs_dev, uint64_t capacity, uint64_t* count);
/* Per-channel diagnostics after the last run: diag_host[C]. */
int m17hip_diag_fetch(m17hip_ctx* ctx, m17_diag* diag_host, uint32_t channels);

/* Every diagnostic callback of the last run, not only the last one: after m17hip_tune(ctx, 9, room) each invocation of the
 * reference's diagnostic callback (every 960 samples while the carrier is on, every 384 while it is off) appends its
 * arguments to the channel's log: log_host[channels][capacity], counts_host[channels] entries valid per row, in stream order.
 * In a log entry demod_state / n_frames are the values at that moment and pad[0] | pad[1] << 32 is the index (since reset) of
 * the sample whose processing fired the callback — frame callbacks of the same sample come first (M17Demodulator.h:729-752).
 * M17HIP_ETRUNC if a channel fired more than min(capacity, room) callbacks. */
int m17hip_diag_log_fetch(m17hip_ctx* ctx, m17_diag* log_host, uint32_t* counts_host, uint32_t channels, uint32_t capacity);

/* Synthetic input on the device: the framing of the reference's modulator CLI (apps/m17-mod.cpp:164-504,
 * 628-677: preamble, LSF, stream / BERT / packet frames, EOT), its pulse shaping (one symbol per 10 samples through the 150-tap
 * RRC in double, x 7168, truncation to int16, :204-224) and the impairments of BASELINE config 5, written straight into the
 * context's input slab: [channels][samples], ready for m17hip_demod_run (no upload).  Channel c of the batch uses
 * seed ^ splitmix64((chan0 + c) * 0x9E3779B97F4A7C15 + 1); kind < 0 = even channels BERT, odd channels voice-like streams.
 * Reproducible bit for bit (integer-hash noise); the parity tests compare every int16 with the test generator. */
typedef struct m17_synth_params {
    uint64_t seed;
    int32_t kind;          /* 0 BERT, 1 voice-like stream, 2 RAW packet, 3 noise only, 4 RAW packet closed by a CRC-16/X.25 FCS (1..33 frames), < 0 mixed */
    int32_t n_frames;      /* payload frames */
    int32_t lead_in;       /* samples of lead-in noise (sigma lead_sigma) before the burst */
    int32_t phase;         /* extra delay 0..9 samples, < 0 = derived from the seed */
    int32_t tail, total;   /* unused here (the slab length is `samples`) */
    int32_t invert;        /* transmit inverted polarity */
    int32_t n_preamble;    /* 0 = as m17-mod does (2 for BERT, else 1) */
    double lead_sigma, noise_sigma, dc_offset, gain, tail_sigma;   /* LSB */
} m17_synth_params;
int m17hip_synth_i16(m17hip_ctx* ctx, const m17_synth_params* params, uint32_t channels, uint32_t samples, uint32_t chan0);
/* Read the input slab back: out[channels][samples] (row pitch in samples). */
int m17hip_download_i16(m17hip_ctx* ctx, int16_t* host, uint32_t channels, uint32_t samples, size_t pitch);

/* Payload consumer: BERT statistics — decode_bert + PRBS9::validate (apps/m17-demod.cpp:286-304,
 * Util.h:320-441: LFSR x^9 + x^5 + 1, lock after 18 good bits, unlock at 25 errors in the last 128 bits) over the BERT
 * frame records of every run since the last m17hip_demod_reset, per channel.  Enabled with m17hip_tune(ctx, 6, 1) BEFORE the
 * runs to be counted (the records of a run are accounted at its end); M17HIP_ESTATE otherwise. */
typedef struct m17_bert_stat {
    uint32_t bits;    /* PRBS9::bits()   */
    uint32_t errors;  /* PRBS9::errors() */
    uint32_t synced;  /* PRBS9::sync()   */
    uint32_t frames;  /* BERT frames seen */
} m17_bert_stat;
int m17hip_bert_stats(m17hip_ctx* ctx, m17_bert_stat* stats_host, uint32_t channels);

/* Payload consumer: packet reassembly — decode_packet (apps/m17-demod.cpp:207-253) with the per-transmission
 * reset of dump_lsf (:154-155), per channel, over the packet frame records of every run since the last m17hip_demod_reset:
 * an LSF record starts a new packet; numbered frames (payload[25] = frame number << 2) must arrive in sequence, one that does
 * not is dropped and counted; the frame with the EOF bit (payload[25] & 0x80) contributes its first min(count, 25) bytes and
 * closes the packet, whose CRC-16/X.25 over contents + FCS must leave 0x0f47 (boost::crc_optimal<16, 0x1021, 0xFFFF, 0xFFFF,
 * true, true>, :218-222).  dump_lsf's ENCAPSULATED branch reads lsf[109..111] of a 30-byte array (:157-171, out of bounds);
 * that prefix is not reproduced — every packet is assembled as RAW.  Enabled with m17hip_tune(ctx, 7, capacity) BEFORE the
 * runs; m17hip_packets_fetch returns the packets the LAST run completed, ordered by (channel, seq); *count = how many
 * there were (M17HIP_EOVERFLOW if more than the capacity set by m17hip_tune). */
typedef struct m17_packet_rec {
    uint32_t channel;
    uint32_t seq;          /* packets this channel completed before this one, since reset */
    uint64_t sample_pos;   /* of the closing frame's callback */
    uint16_t size;         /* bytes in data, FCS included */
    uint16_t checksum;     /* CRC-16/X.25 of data[0..size): 0x0f47 for an intact packet */
    uint8_t crc_ok;
    uint8_t frames;        /* packet frames accepted, the closing one included */
    uint8_t seq_errors;    /* frames dropped by the sequence check since the LSF */
    uint8_t reserved;
    uint8_t data[840];     /* at most 32 x 25 + 25 bytes are ever used */
} m17_packet_rec; /* 864 bytes */
int m17hip_packets_fetch(m17hip_ctx* ctx, m17_packet_rec* recs_host, uint32_t capacity, uint32_t* count);
/* The same consumer over frame records supplied by the caller (fetched earlier, or gathered from other GPUs) instead of the
 * last run's: recs_host[channels][pitch] with counts_host[c] records used in row c.  The per-channel assembly state advances
 * exactly as it does at the end of a run; the packets completed are then returned by m17hip_packets_fetch. */
int m17hip_packets_feed(m17hip_ctx* ctx, const m17_frame_rec* recs_host, const uint32_t* counts_host, uint32_t channels, uint32_t pitch);

/* Payload consumer: link setup frames as text — LinkSetupFrame::decode_callsign (LinkSetupFrame.h:95-121: 6 bytes
 * big-endian base 40 -> up to 9 characters, all ones = "BROADCAST"), the 16-bit type field and the CRC check of dump_lsf
 * (apps/m17-demod.cpp:124-200), for a batch of n 30-byte LSFs (e.g. the payloads of the frame_type 0 records). */
typedef struct m17_lsf_info {
    char dst[10], src[10];   /* NUL-padded callsigns */
    uint16_t type;           /* LSF bytes 12..13 */
    uint8_t crc_ok;          /* CRC16 over the 30 bytes == 0 */
    uint8_t reserved[9];
} m17_lsf_info;
int m17hip_lsf_info(m17hip_ctx* ctx, const uint8_t* lsf30_host, uint32_t n, m17_lsf_info* out_host);

/* ---- numerics switch: evaluation order of the Kalman updates --------------------------------------------------------- */
/* KalmanFilter.h:41-65,91-107 (used by ClockRecovery.h:54-67 and FreqDevEstimator.h:31-48) keeps the innovation covariance and
 * the gain as lazy expressions of the blaze library (`auto S`, `auto K`), so the association and rounding of `x += K*y` and
 * `P = P - K*H*P` are decided by blaze's restructuring operators.  blaze is not part of the reference tree (empty submodule),
 * so the order is selectable: bit 0: x += double(fl32(P(:,0)*y)) * (1/S) instead of x += (double(P(:,0))/S) * y;
 * bit 1: P -= double(fl32(P(i,0)*P(0,j))) * (1/S) instead of P -= (double(P(i,0))/S) * P(0,j); bit 2: F*(P*F^T) instead of
 * (F*P)*F^T.  Default 3 (blaze's documented restructuring rules; DESIGN.md §4.4 has the measured sensitivity).  Applies to the
 * runs that follow. */
int m17hip_set_kalman_order(m17hip_ctx* ctx, int order);
/* a9/a10 per-operator parity entry: `rows` independent filters reset to z0, n updates each with measurement z[r][i] after
 * dt[r][i] samples; wrap = 10: KalmanFilter<float,10> (KalmanFilter.h:41-65), 0: SymbolKalmanFilter (:91-107).
 * out[rows][n][6] = x[0], x[1], P(0,0), P(0,1), P(1,0), P(1,1) after each update. */
int m17hip_kalman_trace(m17hip_ctx* ctx, const float* z_host, const uint32_t* dt_host, uint32_t rows, uint32_t n, int wrap, float z0, int order,
                        float* out_host);

/* ---- multi-GPU: contiguous channel shards, one context (rank) per GPU, no data-path collective ------------ */
/* Records of this context carry channel = channel_base + local channel index (default 0), so the union of the shards'
 * record sets is the record set of one big run.  Applies to the runs that follow (also to m17hip_packets_fetch). */
int m17hip_set_channel_base(m17hip_ctx* ctx, uint32_t channel_base);
/* The one exchange of the path: the gather of the frame records of the last run to `root` over RCCL (xGMI inside a node).
 * Rank 0 obtains an id (ncclGetUniqueId) and hands its 128 bytes to the other ranks by any host-side channel; every rank then
 * creates its communicator (ncclCommInitRank; collective; any context of the same device may then gather through it, one call
 * at a time).  m17hip_gather_frames is collective: every rank compacts its
 * records on the device, the counts are all-gathered, the records travel to `root` with their exact sizes (grouped
 * ncclSend / ncclRecv) and land in recs_host[capacity] rank after rank — with contiguous shards and channel bases set that is
 * global (channel, seq) order.  counts[nranks] (optional) and *total are filled on every rank; recs_host is only used on
 * `root`.  M17HIP_ETRUNC if total > capacity, M17HIP_ECOMM if RCCL is missing or fails (m17hip_comm_last_error).
 * Failure behaviour: a rank-local failure (its compaction, an allocation, a word it cannot write or read) travels inside the two
 * all-gathers every call makes before any record moves — the failing rank returns its own code, every other rank M17HIP_ECOMM, all of
 * them from the same call, and the communicator stays usable.  What words cannot settle (a peer that died, a rank that lost its device
 * between the second all-gather and the records) is bounded in time: every wait of the call has a deadline (m17hip_tune key 31, default
 * 120 s); when it passes the communicator is given up (ncclCommAbort), the call and every later call through it return M17HIP_ECOMM,
 * and the ranks create a new communicator to go on. */
typedef struct m17hip_comm m17hip_comm;
#define M17HIP_COMM_ID_BYTES 128
int m17hip_comm_get_id(void* id128);
int m17hip_comm_create(m17hip_ctx* ctx, const void* id128, int rank, int nranks, m17hip_comm** out);
void m17hip_comm_destroy(m17hip_comm* comm);
int m17hip_comm_last_error(const m17hip_comm* comm);
int m17hip_gather_frames(m17hip_ctx* ctx, m17hip_comm* comm, int root, m17_frame_rec* recs_host, uint64_t capacity, uint64_t* counts,
                         uint64_t* total);
/* Same, with the root's destination in DEVICE memory on the root's GPU (the gathered set stays in HBM for a device-side consumer). */
int m17hip_gather_frames_device(m17hip_ctx* ctx, m17hip_comm* comm, int root, m17_frame_rec* recs_dev, uint64_t capacity, uint64_t* counts,
                                uint64_t* total);

/* Knobs of a context (never results).  M17HIP_EINVAL for a key the library does not have.
 * key 3: samples per segment a run is processed in (default 48000; 0 = one segment): the granule of the K2 / K5 alternation — shorter
 *        segments bound how long a channel that lost sync computes its own limit-filter history, longer ones mean fewer launches.
 * key 6: BERT statistics on/off (m17hip_bert_stats; default off).
 * key 7: packet reassembly, value = packets of room per run (m17hip_packets_fetch; default 0 = off).
 * key 8: record slots per channel and run actually used, 0 = all that were allocated (2 per 1920 samples + 8, which a run cannot
 *        outgrow) — a smaller value makes M17HIP_EOVERFLOW reachable for tests.
 * key 9: diagnostic log, value = diagnostic callbacks of room per channel and run (m17hip_diag_log_fetch; default 0 = off).
 * key 10: form of the carrier-detect kernel K3: 0 = one wave per 32 channels (least wave slots: batch throughput), 1 = four-wave pipeline per
 *        32 channels (1.8x shorter chain, four times the wave slots: stream latency), -1 (default) = the pipeline for runs whose front end
 *        was queued by m17hip_demod_front (a continued stream waits for K3's chain) and for runs of a process that has not overlapped runs
 *        of different contexts on this device lately (one batch at a time: 29 -> 26 ms per 4096 x 480 000), the one-wave form while it does.
 *        Same table either way.  NOTE: -1 consults PROCESS-WIDE state (a registry of the process's contexts on the device and their last
 *        end-of-run events): the form, and with it a run's latency, depends on what other contexts of the process did lately — never a
 *        result.  A host that needs the same latency whatever else the process runs pins 0 or 1.
 * key 13: workgroups of the matched filter's grid (a K1 workgroup loops over (channel, tile) items), 0 (default) = about three items per workgroup
 *        for runs that get K3's latency form (key 10), eight for the others, at least five workgroups per compute unit.
 * key 26: gate-aware front end.  The reference runs neither the matched filter nor the correlator while its carrier detect is off
 *        (M17Demodulator.h:675-689).  1 = the matched filter of segment k >= 2 of a run skips what the carrier cannot be on for: the sequential
 *        kernel leaves the TRUE gate state at the end of every segment, a forecast walks the carrier-off update points of the next two segments
 *        from it (a closed gate reopens only when an update finds level > 4.0: a function of the table and the off state alone) and K1 follows
 *        K5 of segment k - 2 instead of running ahead — on input that is idle most of the time 1.35 x the throughput, on always-on input 5 %
 *        less (K1 on the chain); 0 = never; -1 (default) = per run: on when more than a quarter of the channel-segments of the last FETCHED run
 *        ended with the carrier off.  Same results either way.  NOTE: -1 consults the context's HISTORY (what the last fetched run looked like):
 *        the schedule of a run, and with it its latency, depends on the runs before — never a result.  A host that needs the same latency
 *        whatever came before pins 0 or 1.
 * key 33: a RAMP of segment lengths at the start of a run: value, 2 x value, 4 x value, ... samples until key 3's length is reached (0 = none).
 *        A channel that leaves the limit-filter replay in a segment carries the filter itself to the end of the NEXT one, and channels leave
 *        it mostly while sync is being acquired: short first segments bound what that costs the launches concerned.  One batch at a time
 *        23.65 -> 22.8 ms per 4096 x 480 000 with 9600; with batches in flight or a continued stream the extra launches cost 1-3 % (NOTES 6.5).
 *        -1 (default) = per run: 9600 when the run is the only thing in flight on the device as far as the library can see (no other context's
 *        run unfinished when this one is queued, none since this context's previous run — the rule key 10 = -1 uses) and was not staged, else none.
 *        NOTE: like keys 10 and 26, -1 consults process HISTORY: the schedule of a run, and with it its latency, depends on what was in flight
 *        before it — never a result.  A host that needs the same latency whatever came before pins a value.
 * key 20: what happens after a forced dcd.unlock() took a channel off the limit-filter replay: 0 (default) = the replay's state is re-derived
 *        beside the sequential kernel and the channel computes its own filter history through the next segment (the sequential kernel never
 *        waits: best wherever its chain of launches is what a step lasts — a continued stream, one batch at a time); 1 = the replay of the next
 *        segment is redone for those channels, history stored, IN FRONT of the sequential kernel (1-2 ms of replay latency on that chain, fewer
 *        instructions in all).  With round 4's matched filter 1 was 1.4 % faster when several independent batches were in flight; with
 *        round 5's it is slower in every regime (two batches 22.6 against 21.4 ms per 4096 x 480 000, a continued stream 28.0 against 24.7,
 *        one batch at a time 28.9 against 26.3: NOTES 5.5) and stays only as the other side of that comparison.
 * key 15: 1 (default) = the sequential kernel leaves the payload frames of running stream / BERT transmissions undecoded (LLRs to a
 *        store, the record reserved) and a lane-per-frame kernel decodes them after the run; 0 = every frame is decoded where it completes.
 * key 17: 1 (default) = the running EVM of the diagnostic callback (RunningStandardDeviation: three dependent operations per payload symbol
 *        that nothing in the demodulator reads) is folded OUTSIDE the sequential kernel: that kernel writes one 4-byte operation per symbol
 *        (4 B x samples / 10 per channel of device memory), one lane per channel folds them in the reference's order as extra workgroups of
 *        the limit-filter replay / the deferred decode, and m17_diag::evm and the diagnostic log carry the same values as with 0 = folded in
 *        the sequential kernel, symbol by symbol.  Can be changed between runs.
 * key 18 (tests): floats per channel row of deferred EVM operations (key 17), 0 (default) = what a run of max_samples can produce; with a
 *        smaller row a channel outruns it, the operations beyond are dropped and m17hip_diag_fetch / m17hip_diag_log_fetch return
 *        M17HIP_EOVERFLOW (every diagnostic field but `evm` is still right; the frame records are not affected).
 * key 16: 1 = m17hip_upload_i16, m17hip_upload_i16_device and m17hip_synth_i16 write the context's STAGING slab (as
 *        m17hip_upload_i16_async does, but complete when they return) and stage it for the next run; 0 (default) = the current slab.
 * key 30 (tests): fault injection for m17hip_gather_frames*: 1 = this rank's compaction fails inside the call, 2 = the root's staging
 *        allocation fails, 3 = this rank's word of the second exchange cannot be written (and, on the root, the staging is grown whether
 *        it has to be or not), 4 = this rank cannot read the first exchange, 5 = this rank cannot read the second exchange; 0 = none.
 *        Under 1-4 every rank makes all its collective calls and all return from the same call; 5 is the case key 31 bounds.
 * key 31: deadline in milliseconds of every wait inside m17hip_gather_frames* (default 120000, 0 = none).
 * The measurement build of the library (make -C m17-cxx-demod_amd/csrc tools -> libm17hip_tools.so, -DM17_TOOLS; tools/ only) adds
 * key 1 / key 19 (section timers / per-wave working times of the sequential kernel -> m17hip_debug_counters) and the schedule
 * experiments 4, 5, 12, 14, 21, 25 (csrc/m17hip.hip, m17hip_tune). */
int m17hip_tune(m17hip_ctx* ctx, int key, int64_t value);

/* Measurement build only (otherwise *waves = 0).  Diagnostic counters of the last sequential-kernel launch (after m17hip_tune(ctx, 1, 1)): host[channels][40] =
 * {total, bulk chunks, single-sample steps, frame decodes} in 10 ns ticks, {#chunks, #single steps, samples in chunks,
 * #chunks cut by a clock move | #decodes << 32}, then [8..14] ticks and [16..22] counts of single-sample steps per
 * DemodState. */
int m17hip_debug_counters(m17hip_ctx* ctx, uint64_t* host, uint32_t max_waves, uint32_t* waves);

/* ---- measurement ----------------------------------------------------------------------------------- */
/* When enabled, every kernel launch of the context is timed by HIP events: the five kernels of a run's chain (matched filter, carrier detect, limit filter,
 * sequential kernel, deferred decode) by events BOUND to the launch (hipExtLaunchKernelGGL: the dispatch's own time stamps, what rocprofv3 reports), the others by
 * a pair recorded around them on the stream they run on.  Not free on a chain of dependent launches: a continued stream 0.45 ms of a 21.7 ms step, two batches in
 * flight 1.5 % (tools/stream_only.py, tools/stream_history.py: TIMING=1). */
int m17hip_timing_enable(m17hip_ctx* ctx, int on);
/* Accumulated device time (ms) and launch count per kernel since the last m17hip_timing_reset:
 * which: 0 = fir_rrc150, 1 = dcd, 2 = demod_seq, 3 = viterbi/decode_frames, 4 = correlator, 5 = compaction,
 * 6 = limit_track (the limit filter run ahead of demod_seq; in m17hip_fir_correlator the limit filter's chain, 4 = its correlations). */
int m17hip_timing_get(m17hip_ctx* ctx, int which, double* total_ms, uint64_t* launches);
int m17hip_timing_reset(m17hip_ctx* ctx);

#ifdef __cplusplus
}
#endif
#endif /* M17HIP_H */
