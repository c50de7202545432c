/*
 * shennong_amd.h — C ABI of libshennong_hip.so, the MI355X (gfx950) speech-features backend.
 *
 * This is the drop-in boundary for the hot path of bootphon/shennong: the seam where the
 * reference's Python processors hand one utterance to pykaldi/Kaldi
 *   - MelFeaturesProcessor._process:  cls(options).compute(SubVector(int16 wave), vtln_warp)
 *         (reference shennong/processor/base.py:408-436; filterbank.py:84; mfcc.py:86)
 *   - SpectrogramProcessor.process:   Spectrogram(options).compute(wave, 1.0)
 *         (reference shennong/processor/spectrogram.py:134-140)
 *   - PlpProcessor._compute:          per-frame Python loop over pykaldi primitives
 *         (reference shennong/processor/plp.py:510-626)
 *   - KaldiPitchProcessor.process:    compute_kaldi_pitch(options, wave)
 *         (reference shennong/processor/pitch_kaldi.py:296-299)
 *   - KaldiPitchPostProcessor.process: process_pitch(options, matrix)
 *         (reference shennong/processor/pitch_kaldi.py:535-537)
 *   - DeltaPostProcessor.process:     compute_deltas(options, matrix)
 *         (reference shennong/postprocessor/delta.py:129-131)
 *   - EnergyProcessor.process:        per-frame extract_window + sum of squares
 *         (reference shennong/processor/energy.py:148-186)
 *   - Frames.nframes / window():      num_frames / FeatureWindowFunction
 *         (reference shennong/frames.py:137; shennong/window.py:107-114)
 *   - VadPostProcessor.process:       kaldi.ivector.compute_vad_energy
 *         (reference shennong/postprocessor/vad.py:182-185)
 *   - CmvnPostProcessor.accumulate / process: kaldi.transform.cmvn.Cmvn.accumulate / apply
 *         (reference shennong/postprocessor/cmvn.py:216-219, :277-278)
 *   - SlidingWindowCmvnPostProcessor.process: kaldi.feat.functions.sliding_window_cmn
 *         (reference shennong/postprocessor/cmvn.py:493-495)
 *
 * The ABI is batch-first: one call covers N utterances given as one concatenated int16 buffer
 * plus an offsets table (the reference's process_all / joblib loop, base.py:56-107, becomes a
 * single launch).  Plain pointers and sizes only; no torch/numpy types.
 *
 * Threading: a plan is immutable after creation; run calls on the same plan serialise on an
 * internal mutex, calls on different plans may run concurrently.  The library never keeps a
 * caller pointer after a call returns.  Errors: every entry point returns 0 on success and a
 * negative SNF_E_* code otherwise; snf_last_error() returns a thread-local message.
 *
 * Option-struct field meanings and defaults are exactly Kaldi's FrameExtractionOptions /
 * MelBanksOptions / FbankOptions / MfccOptions / PlpOptions / SpectrogramOptions /
 * PitchExtractionOptions / ProcessPitchOptions / DeltaFeaturesOptions, which the reference
 * wraps one-to-one (reference shennong/processor/base.py:122-374, filterbank.py:48-55,
 * mfcc.py:48-56, plp.py:265-273, pitch_kaldi.py:86-91,321-327, delta.py:54).
 */
#ifndef SHENNONG_AMD_H_
#define SHENNONG_AMD_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- error codes ------------------------------------------------------------------------- */
#define SNF_OK 0
#define SNF_E_INVALID (-1)  /* bad argument (Python raises ValueError before reaching here)   */
#define SNF_E_RUNTIME (-2)  /* Kaldi-class option error (KALDI_ERR) -> Python RuntimeError     */
#define SNF_E_HIP (-3)      /* HIP runtime failure                                             */
#define SNF_E_NODEVICE (-4) /* no usable gfx950 device                                         */

/* ---- plan kinds --------------------------------------------------------------------------- */
#define SNF_KIND_SPECTROGRAM 0
#define SNF_KIND_FBANK 1
#define SNF_KIND_MFCC 2
#define SNF_KIND_PLP 3
#define SNF_KIND_PITCH 4
#define SNF_KIND_PITCH_POST 5
#define SNF_KIND_DELTA 6
#define SNF_KIND_ENERGY 7
#define SNF_KIND_VAD 8
#define SNF_KIND_CMVN 9
#define SNF_KIND_SLIDING_CMVN 10

/* ---- window types (reference shennong/processor/base.py:215-221) --------------------------- */
#define SNF_WINDOW_HAMMING 0
#define SNF_WINDOW_HANNING 1
#define SNF_WINDOW_POVEY 2
#define SNF_WINDOW_RECTANGULAR 3
#define SNF_WINDOW_BLACKMAN 4

/* energy compression of EnergyProcessor (reference shennong/processor/energy.py:81-84) */
#define SNF_COMPRESS_OFF 0
#define SNF_COMPRESS_LOG 1
#define SNF_COMPRESS_SQRT 2

/* Kaldi FrameExtractionOptions (reference shennong/processor/base.py:122-262). */
typedef struct snf_frame_options {
  float samp_freq;               /* 16000 */
  float frame_shift_ms;          /* 10    */
  float frame_length_ms;         /* 25    */
  float dither;                  /* 1.0 (0 = off; != 0 uses a counter-based GPU RNG)           */
  float preemph_coeff;           /* 0.97  */
  int32_t remove_dc_offset;      /* 1     */
  int32_t window_type;           /* SNF_WINDOW_POVEY */
  int32_t round_to_power_of_two; /* 1     */
  float blackman_coeff;          /* 0.42  */
  int32_t snip_edges;            /* 1     */
} snf_frame_options;

/* Kaldi MelBanksOptions (reference shennong/processor/base.py:288-374). */
typedef struct snf_mel_options {
  int32_t num_bins; /* 23 */
  float low_freq;   /* 20 */
  float high_freq;  /* 0 => Nyquist, <0 => offset from Nyquist */
  float vtln_low;   /* 100 */
  float vtln_high;  /* -500 */
} snf_mel_options;

/* Kaldi PitchExtractionOptions (reference shennong/processor/pitch_kaldi.py:86-91). */
typedef struct snf_pitch_options {
  float samp_freq;               /* 16000 */
  float frame_shift_ms;          /* 10 */
  float frame_length_ms;         /* 25 */
  float preemph_coeff;           /* 0 */
  float min_f0;                  /* 50 */
  float max_f0;                  /* 400 */
  float soft_min_f0;             /* 10 */
  float penalty_factor;          /* 0.1 */
  float lowpass_cutoff;          /* 1000 */
  float resample_freq;           /* 4000 */
  float delta_pitch;             /* 0.005 */
  float nccf_ballast;            /* 7000 */
  int32_t lowpass_filter_width;  /* 1 */
  int32_t upsample_filter_width; /* 5 */
  int32_t recompute_frame;       /* 500 (not exposed by the reference; Kaldi default)           */
  int32_t snip_edges;            /* 1   (not exposed by the reference; Kaldi default)           */
} snf_pitch_options;

/* Kaldi ProcessPitchOptions (reference shennong/processor/pitch_kaldi.py:321-327). */
typedef struct snf_pitch_post_options {
  float pitch_scale;                   /* 2.0 */
  float pov_scale;                     /* 2.0 */
  float pov_offset;                    /* 0.0 */
  float delta_pitch_scale;             /* 10.0 */
  float delta_pitch_noise_stddev;      /* 0.005 (0 = deterministic)                             */
  int32_t normalization_left_context;  /* 75 */
  int32_t normalization_right_context; /* 75 */
  int32_t delta_window;                /* 2 */
  int32_t delay;                       /* 0 */
  int32_t add_pov_feature;             /* 1 */
  int32_t add_normalized_log_pitch;    /* 1 */
  int32_t add_delta_pitch;             /* 1 */
  int32_t add_raw_log_pitch;           /* 0 */
} snf_pitch_post_options;

/* Kaldi VadEnergyOptions (reference shennong/postprocessor/vad.py:77-78). */
typedef struct snf_vad_options {
  float energy_threshold;     /* 5.0 */
  float energy_mean_scale;    /* 0.5 */
  int32_t frames_context;     /* 0 */
  float proportion_threshold; /* 0.6 */
} snf_vad_options;

/* Kaldi SlidingWindowCmnOptions (reference shennong/postprocessor/cmvn.py:407-408). */
typedef struct snf_sliding_cmvn_options {
  int32_t center;             /* 1 */
  int32_t cmn_window;         /* 600 */
  int32_t min_window;         /* 100 */
  int32_t normalize_variance; /* 0 */
} snf_sliding_cmvn_options;

/* One flat options record; `kind` selects which fields are read. */
typedef struct snf_options {
  int32_t kind; /* SNF_KIND_* */
  snf_frame_options frame;
  snf_mel_options mel;
  /* FbankOptions / MfccOptions / PlpOptions / SpectrogramOptions / EnergyProcessor */
  int32_t use_energy;
  float energy_floor;
  int32_t raw_energy;
  int32_t htk_compat;
  int32_t use_log_fbank; /* fbank only */
  int32_t use_power;     /* fbank only */
  int32_t num_ceps;      /* mfcc, plp */
  float cepstral_lifter; /* mfcc, plp */
  int32_t lpc_order;     /* plp */
  float compress_factor; /* plp */
  float cepstral_scale;  /* plp */
  int32_t rasta;         /* plp (shennong extension, reference plp.py:64-146) */
  int32_t compression;   /* energy: SNF_COMPRESS_* */
  /* DeltaFeaturesOptions */
  int32_t delta_order;  /* 2 */
  int32_t delta_window; /* 2 */
  /* mfcc only: non-zero appends the deltas of order 2 / window 2 to every row in the same call,
     [cepstra | delta | delta-delta] = what DeltaPostProcessor().process(MfccProcessor().process(audio))
     returns (reference postprocessor/delta.py:129-131 chained after processor/mfcc.py:86): the plan runs
     the MFCC kernel into a scratch in HBM and the delta kernel on it (any MFCC configuration).  The
     one-launch form of round 2 (512-point path only, slower) is selected by SNF_FUSED_DELTA=1 in the
     environment of snf_plan_create.  Other orders / windows are refused at plan creation. */
  int32_t append_deltas;
  snf_pitch_options pitch;
  snf_pitch_post_options pitch_post;
  snf_vad_options vad;
  snf_sliding_cmvn_options sliding_cmvn;
  uint64_t seed; /* RNG seed for dither / delta-pitch noise */
} snf_options;

typedef struct snf_plan snf_plan; /* opaque */

/* ---- library / device --------------------------------------------------------------------- */
const char* snf_version(void);
const char* snf_last_error(void);
int snf_device_count(void);
int snf_set_device(int device_id);
int snf_device_name(int device_id, char* buf, int buflen);
int snf_device_synchronize(void);

/* ---- host-side helpers that replace pykaldi free functions -------------------------------- */
/* kaldi.feat.window.num_frames(nsamples, opts, flush=True)   (reference frames.py:137) */
int64_t snf_num_frames(const snf_frame_options* o, int64_t num_samples);
/* kaldi.feat.window.first_sample_of_frame(frame, opts)        (reference plp.py:218) */
int64_t snf_first_sample_of_frame(const snf_frame_options* o, int64_t frame);
int32_t snf_window_size(const snf_frame_options* o);
int32_t snf_window_shift(const snf_frame_options* o);
int32_t snf_padded_window_size(const snf_frame_options* o);
/* FeatureWindowFunction.from_options(opts).window             (reference window.py:107-114);
   writes snf_window_size() floats. */
int snf_window_function(const snf_frame_options* o, float* out);
/* Number of frames Kaldi's pitch extractor emits for `num_samples` input samples. */
int64_t snf_pitch_num_frames(const snf_pitch_options* o, int64_t num_samples);

/* ---- plans ---------------------------------------------------------------------------------- */
/* Precomputes window, mel banks (per distinct vtln warp, on demand), DCT, lifter, IDFT bases,
   resampler taps... and uploads them to `device_id`.  Immutable afterwards. */
int snf_plan_create(const snf_options* opts, int device_id, snf_plan** out);
void snf_plan_destroy(snf_plan* plan);
/* output dimension (columns); for DELTA/PITCH_POST/ENERGY see the dedicated entry points */
int32_t snf_plan_ndims(const snf_plan* plan);
/* 1 when the plan runs on a register-resident kernel (even frames that pad to 512, 256 or 128 samples
   with up to 64 mel bins, 16 cepstra and a power spectrum: 8 and 16 kHz audio; or to 2048 / 1024 samples
   with up to 128 mel bins: 32, 44.1 and 48 kHz audio), 0 when its option combination falls back to the
   generic wave-per-frame kernel (5 to 8x slower per frame; e.g. 4096-sample frames, odd window lengths,
   magnitude filterbanks on short frames): hosts should say so instead of being silently slow. */
int32_t snf_plan_fast_path(const snf_plan* plan);
/* rows produced for an utterance of `num_samples` samples (bit-exact Kaldi NumFrames) */
int64_t snf_plan_num_frames(const snf_plan* plan, int64_t num_samples);

/*
 * Audio -> Features for a batch of utterances (kinds SPECTROGRAM, FBANK, MFCC, PLP, PITCH, ENERGY).
 *   wave            concatenated int16 PCM of all utterances            [sample_offsets[n_utts]]
 *   sample_offsets  n_utts+1 offsets into `wave`
 *   vtln_warp       n_utts warp factors or NULL (=1.0; ignored by SPECTROGRAM/PITCH/ENERGY)
 *   out             concatenated row-major float32 [frame_offsets[n_utts], ndims]
 *   frame_offsets   n_utts+1 row offsets; frame_offsets[u+1]-frame_offsets[u] must equal
 *                   snf_plan_num_frames(plan, samples of u)
 * The rows of an utterance are the same bits whatever else is in the batch (the reference processes one
 * utterance at a time: shennong/processor/base.py:150-180): the kernel an utterance runs on depends on the
 * plan, on its own warp factor and on its own length only - a batch may therefore take more than one
 * launch (unwarped / warped utterances of a two-frames-per-row plan; utterances shorter than one window
 * with snip_edges = 0 go to the generic kernel).  Exception: dither != 0 (a random stream keyed by the
 * frame's position in the batch).
 * Host-pointer variant: stages through device memory owned by the plan (H2D, kernels, D2H).
 */
int snf_plan_run_batch(snf_plan* plan, const int16_t* wave, const int64_t* sample_offsets,
                       int64_t n_utts, const float* vtln_warp, float* out,
                       const int64_t* frame_offsets);

/* Device-pointer variant: `d_wave`/`d_out` are device buffers on the plan's device; the offsets
   tables and vtln_warp stay host pointers (they are small and are uploaded by the call).
   `stream` is a hipStream_t (NULL = the plan's own stream).  Asynchronous w.r.t. the host when
   `stream` != NULL; with NULL the call returns after the plan's stream has been synchronised.
   A plan owns scratch in HBM - its offsets and frame tables, the lists of the two-frames-per-transform kernels,
   the mel / log-mel rows of PLP and of MFCC with more than 16 cepstra, the tracker's buffers: asynchronous
   calls of ONE plan on DIFFERENT streams must not overlap on the device unless they pass the same offsets
   tables and the plan writes no intermediate (the 512-sample filterbank / MFCC-13 / spectrogram kernels).  One
   plan per stream otherwise (what the Python host does for the pieces of a large batch: `Plan._clones`). */
int snf_plan_run_batch_device(snf_plan* plan, const int16_t* d_wave, const int64_t* sample_offsets,
                              int64_t n_utts, const float* vtln_warp, float* d_out,
                              const int64_t* frame_offsets, void* stream);

/*
 * Features -> Features post-processors (kinds DELTA, PITCH_POST, VAD, SLIDING_CMVN).
 * VAD writes one column of 0.0 / 1.0 (the reference casts it to uint8).
 *   in   concatenated row-major float32 [frame_offsets[n_utts], in_cols]
 *   out  concatenated row-major float32 [frame_offsets[n_utts], snf_post_ndims(plan, in_cols)]
 */
int32_t snf_post_ndims(const snf_plan* plan, int32_t in_cols);
int snf_post_run_batch(snf_plan* plan, const float* in, int32_t in_cols,
                       const int64_t* frame_offsets, int64_t n_utts, float* out);
int snf_post_run_batch_device(snf_plan* plan, const float* d_in, int32_t in_cols,
                              const int64_t* frame_offsets, int64_t n_utts, float* d_out,
                              void* stream);

/*
 * CMVN (plan kind SNF_KIND_CMVN).  Statistics are Kaldi's double [2, cols+1] blocks
 * (row 0: sums and, last, the weighted frame count; row 1: sums of squares).
 *   accumulate: stats[group[u]] += stats of utterance u (weights: per-frame, NULL = 1.0;
 *               group: per-utterance speaker index, NULL = one group 0).  `stats` is
 *               [n_groups, 2, cols+1] on the host and is accumulated INTO.
 *   apply:      out = Kaldi ApplyCmvn / ApplyCmvnReverse of each utterance with stats[group[u]];
 *               fails with SNF_E_INVALID when a used group has count < 1.
 */
int snf_cmvn_accumulate(snf_plan* plan, const float* in, int32_t cols, const int64_t* frame_offsets,
                        int64_t n_utts, const float* weights, const int32_t* group,
                        int32_t n_groups, double* stats);
int snf_cmvn_apply(snf_plan* plan, const float* in, int32_t cols, const int64_t* frame_offsets,
                   int64_t n_utts, const double* stats, const int32_t* group, int32_t n_groups,
                   int32_t norm_vars, int32_t reverse, float* out);
/* the same with the feature blocks (and the optional per-frame weights) resident in HBM; offsets,
 * group ids and statistics stay on the host */
int snf_cmvn_accumulate_device(snf_plan* plan, const float* d_in, int32_t cols,
                               const int64_t* frame_offsets, int64_t n_utts, const float* d_weights,
                               const int32_t* group, int32_t n_groups, double* stats);
int snf_cmvn_apply_device(snf_plan* plan, const float* d_in, int32_t cols,
                          const int64_t* frame_offsets, int64_t n_utts, const double* stats,
                          const int32_t* group, int32_t n_groups, int32_t norm_vars, int32_t reverse,
                          float* d_out);

/*
 * Column-wise concatenation of two device-resident feature blocks, utterance by utterance (reference
 * shennong/features.py:386-437 `Features.concatenate`, used by pipeline.py:636-641 to append the pitch
 * columns): out[u] = [a[u][:rows], b[u][:rows]] with rows = offsets_out[u+1] - offsets_out[u] (the
 * caller trims the longer side within its tolerance).  All offsets tables are host arrays [n_utts+1].
 */
int snf_concat_columns_device(int device_id, const float* d_a, int32_t cols_a,
                              const int64_t* offsets_a, const float* d_b, int32_t cols_b,
                              const int64_t* offsets_b, int64_t n_utts, float* d_out,
                              const int64_t* offsets_out);

/*
 * Random terms (dither, reference processor/base.py:122; delta-pitch noise, pitch_kaldi.py:321-327).  A
 * draw is keyed by (options seed, noise call, the frame's index inside its utterance, the utterance's
 * length and a hash of 64 samples spread over it): never by the frame's position in the batch.  The noise call is the plan's own
 * count of calls that draw - every call a new stream, like the reference's global rand() - unless the
 * calling thread names it for its NEXT such call with snf_set_noise_call(call != 0): a caller that must see
 * the same noise for the same utterance in two passes over a corpus (streamed CMVN by speaker: statistics
 * pass, then apply pass) names the same call in both.
 */
int snf_set_noise_call(uint64_t call);

/*
 * Number of NaN / +-Inf among n floats of a device-resident block (16-byte aligned): the data part of
 * reference shennong/features.py:170-215 `Features.is_valid` ("data contains non-finite numbers"), run on
 * the whole batch before its only device -> host copy.
 */
int snf_count_nonfinite_device(int device_id, const float* d_data, uint64_t n, uint64_t* count);

/* ---- device memory + timing (so hosts without torch can keep data resident in HBM) ---------- */
int snf_malloc(void** dptr, uint64_t bytes);
int snf_free(void* dptr);
/* Free and total bytes of HBM on the calling thread's current device (hipMemGetInfo): the streamed pipeline
 * sizes its batches from it (shennong_amd/pipeline.py extract_features_streamed) and bench.py reports the peak
 * use of BASELINE config 5.  No counterpart in the reference (CPU only). */
int snf_mem_info(uint64_t* free_bytes, uint64_t* total_bytes);
/* A host that pools freed device buffers (shennong_amd/_backend.py DEVICE_POOL) registers a callback that
 * releases them: an allocation of the library's own scratch that fails with out-of-memory calls it and tries
 * once more (NULL removes the hook).  No counterpart in the reference (CPU only). */
typedef void (*snf_oom_hook)(void);
int snf_set_oom_hook(snf_oom_hook hook);
int snf_memcpy_h2d(void* dst, const void* src, uint64_t bytes);
int snf_memcpy_d2h(void* dst, const void* src, uint64_t bytes);
int snf_memset(void* dst, int value, uint64_t bytes);  /* complete when it returns (the plans use their own streams) */
/* Streams and stream-ordered copies for callers that overlap the transfers of one batch with the
   kernels of another (the *_device entry points take the stream; page-locked host memory from
   snf_host_malloc is required for the copies to be asynchronous). */
int snf_stream_create(void** stream);
int snf_stream_destroy(void* stream);
int snf_stream_synchronize(void* stream);
/* 0 = everything enqueued on `stream` has completed, 1 = not yet (hipStreamQuery; never waits), < 0 = error.
   A host that must not wait for ever on work that depends on OTHER processes - the exchange steps below: a
   peer that died or stalled never completes them - polls this against a deadline instead of synchronising
   (bench.py's watchdog).  No counterpart in the reference (its pool is one process). */
int snf_stream_query(void* stream);
int snf_memcpy_h2d_async(void* dst, const void* src, uint64_t bytes, void* stream);
int snf_memcpy_d2h_async(void* dst, const void* src, uint64_t bytes, void* stream);
/* Timing marks on a caller's stream (hipEvent_t behind a plain pointer): a caller that enqueues several
   *_device calls on its own stream without waiting for each - the way a pipeline uses the path, and what
   bench.py times - brackets every call with two marks and reads the elapsed device time of each pair after
   ONE synchronisation (snf_plan_last_kernel_ms is for calls on the plan's own stream and waits for the call).
   snf_event_elapsed_ms waits for `stop`.  Marks belong to the calling thread's current device (the one of the
   stream they are recorded on). */
int snf_event_create(void** event);
int snf_event_destroy(void* event);
int snf_event_record(void* event, void* stream);
/* The host waits until `event` has happened (hipEventSynchronize): ONE copy of a stream that already carries the
   next one - the streamed pipeline sends the audio of batch k + 1 ahead on the stream that carried batch k's
   (shennong_amd/pipeline.py _Prefetch). */
int snf_event_synchronize(void* event);
/* Work enqueued on `stream` after this call starts only when `event` (recorded on another stream) has happened:
   the upload stream of a large batch runs ahead of the stream that launches the kernels and downloads the rows
   (shennong_amd/_backend.py Plan._run_large).  hipStreamWaitEvent; the host does not wait. */
int snf_stream_wait_event(void* stream, void* event);
int snf_event_elapsed_ms(void* start, void* stop, float* ms);
/* Page-locked host staging memory for the host-pointer entry points (snf_plan_run_batch, ...): a
   batch assembled in it (the reference hands over one numpy array per utterance, processor/base.py:428)
   crosses the link at full rate and its pages are faulted in once, not once per call.  Plain host
   memory as far as the caller is concerned. */
int snf_host_malloc(void** hptr, uint64_t bytes);
int snf_host_free(void* hptr);
/* ---- WAV ingest (host only) ---------------------------------------------------------
   The reference decodes one file per job in Python (shennong/audio.py:243-286: scipy, sox for other containers)
   and forces the signal to int16 before Kaldi sees it (processor/base.py:428).  These read RIFF / WAVE files
   natively: snf_wav_scan is `Audio.scan` (channels, sample rate, samples per channel, bits, format tag: 1 = PCM,
   3 = IEEE float); snf_wav_read_pcm16 copies samples [first_sample[i], first_sample[i] + n_samples[i]) of file i
   to dst + dst_offsets[i] for n_files files on `threads` threads - straight into the (page-locked) block a batch
   is uploaded from.  status[i]: 0 = read; 1 = not 16-bit mono PCM (the caller's general reader takes that file);
   2 = I/O error; 3 = the file holds fewer samples than asked for.  The call itself fails only for bad arguments. */
int snf_wav_scan(const char* path, int32_t* channels, int32_t* sample_rate, int64_t* nsamples, int32_t* bits,
                 int32_t* format_tag);
int snf_wav_read_pcm16(const char* const* paths, int64_t n_files, const int64_t* first_sample,
                       const int64_t* n_samples, int16_t* dst, const int64_t* dst_offsets, int32_t threads,
                       int32_t* status);

/* ---- multi-GPU exchange steps (RCCL over xGMI; one process per GPU) --------------------------------
   The features path shards by utterance and needs no data-path collective; what crosses GPUs is (a)
   the final gather of the per-rank feature blocks to a root - the counterpart of the reference's
   thread pool returning every utterance's matrix to one dict (processor/base.py:104-107) - and (b) the
   sum of the by-speaker CMVN statistics (postprocessor/cmvn.py:145-164).  RCCL is loaded on the first
   call.  Bootstrap: rank 0 calls snf_comm_unique_id and hands the 128 bytes to the other ranks by any
   side channel (a socket, a file, MPI ...), then every rank calls snf_comm_init. */
#define SNF_COMM_ID_BYTES 128
typedef struct snf_comm snf_comm; /* opaque */
int snf_comm_unique_id(void* id128);
int snf_comm_init(const void* id128, int32_t world_size, int32_t rank, int32_t device_id, snf_comm** out);
int snf_comm_rank(const snf_comm* comm);
int snf_comm_world_size(const snf_comm* comm);
int snf_comm_destroy(snf_comm* comm);
/* Variable-length gather of float blocks to `root`, device pointers in and out: rank r contributes
   `send_count` floats; on the root they land in `d_recv` in rank order (`recv_counts[world]`, host, root
   only).  Point-to-point ncclSend / ncclRecv in one group: every peer uses its own xGMI link to the root.
   `stream` NULL: the communicator's stream, synchronised before returning. */
int snf_comm_gatherv(snf_comm* comm, const float* d_send, int64_t send_count, float* d_recv,
                     const int64_t* recv_counts, int32_t root, void* stream);
/* In-place all-reduce of float64 on the device; op 0 = sum, 1 = max. */
int snf_comm_allreduce_f64(snf_comm* comm, double* d_buf, int64_t count, int32_t op, void* stream);

/* Test aid: fills the LDS of every CU of the current device with `pattern` (e.g. 0xFFFFFFFF, a NaN).
   LDS is not cleared between workgroups, so a kernel that reads a word it did not write sees whatever
   the previous kernel left there; the parity tests poison it before they compare against the oracle. */
int snf_debug_fill_lds(uint32_t pattern);
/* Test aid: device pointers of the intermediates the last run of a pitch plan left in its scratch
   (resampled signal [total_down], NCCF at the lag of every state [frames, states], NCCF without
   ballast at the integer lags [frames, lags], Viterbi states [frames]); the parity tests compare
   them stage by stage with the oracle's.  Valid until the next call on the plan. */
int snf_debug_pitch_scratch(snf_plan* plan, void** down, void** nccf_res, void** pov_nccf,
                            void** states);
/* Duration in milliseconds of the kernels launched by the last run call on this plan, measured
   with HIP events on the stream the kernels were launched on.  `which` selects a kernel slot:
   0 = whole call, 1.. = per-kernel; returns <0 if the slot was not recorded. */
float snf_plan_last_kernel_ms(const snf_plan* plan, int which);
/* Name of the kernel recorded in slot `which` (NULL if none). */
const char* snf_plan_kernel_name(const snf_plan* plan, int which);

#ifdef __cplusplus
}
#endif
#endif /* SHENNONG_AMD_H_ */
