// Internal declarations of libshennong_hip.so (MI355X / gfx950 speech-features backend).
// Not part of the public ABI (that is include/shennong_amd.h).
#ifndef SNF_INTERNAL_H_
#define SNF_INTERNAL_H_

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <map>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/shennong_amd.h"

namespace snf {

// ---------------------------------------------------------------------------------------------
// error plumbing
// ---------------------------------------------------------------------------------------------
int set_error(int code, const std::string& msg);
const char* last_error();

#define SNF_HIP_CHECK(expr)                                                                  \
  do {                                                                                       \
    hipError_t _e = (expr);                                                                  \
    if (_e != hipSuccess)                                                                    \
      return snf::set_error(SNF_E_HIP, std::string(#expr) + ": " + hipGetErrorString(_e));   \
  } while (0)

// ---------------------------------------------------------------------------------------------
// host-side tables (host_tables.cpp): the product's own restatement of the tables Kaldi
// precomputes in FeatureWindowFunction / MelBanks / MfccComputer / PlpComputer / LinearResample /
// ArbitraryResample (called by the reference at processor/base.py:429-431, plp.py:443-508,
// pitch_kaldi.py:296-299)
// ---------------------------------------------------------------------------------------------
int32_t window_shift(const snf_frame_options& o);
int32_t window_size(const snf_frame_options& o);
int32_t padded_window_size(const snf_frame_options& o);
int64_t num_frames(const snf_frame_options& o, int64_t num_samples);
int64_t first_sample_of_frame(const snf_frame_options& o, int64_t frame);
int make_window(const snf_frame_options& o, std::vector<float>* w);

struct MelBanksHost {
  int num_bins = 0, num_fft_bins = 0;
  std::vector<int> first, size, offset;  // per bin: first fft bin, support length, offset in w
  std::vector<float> w;                  // concatenated supports
  std::vector<float> center_freqs;
};
int make_mel_banks(const snf_mel_options& mo, const snf_frame_options& fo, float vtln_warp,
                   MelBanksHost* out);
// structurally valid banks with zero weights: what a PLP plan holds in place of unwarped banks that the
// options do not allow, until an utterance asks for them (capi.hip: snf_plan::base_banks_error)
void make_placeholder_banks(const snf_mel_options& mo, const snf_frame_options& fo, MelBanksHost* out);
void make_dct_matrix(int num_rows, int num_cols, std::vector<float>* m);  // first rows of NxN DCT
void make_lifter(float q, int n, std::vector<float>* c);
void make_equal_loudness(const MelBanksHost& mb, std::vector<float>* out);
void make_idft_bases(int n_bases, int dim, std::vector<float>* m);
void make_delta_scales(int order, int window, std::vector<float>* scales, std::vector<int>* dims);

struct LinearResampleHost {
  int rate_in = 0, rate_out = 0, in_unit = 0, out_unit = 0, num_zeros = 0, max_taps = 0;
  float cutoff = 0;
  std::vector<int> first;       // [out_unit]
  std::vector<int> ntaps;       // [out_unit]
  std::vector<float> weights;   // [out_unit][max_taps] zero padded
  int64_t num_output(int64_t n_in, bool flush) const;
};
void make_linear_resample(int rate_in, int rate_out, float cutoff, int num_zeros,
                          LinearResampleHost* out);

struct PitchTablesHost {
  int first_lag = 0, last_lag = 0, num_lags = 0, num_states = 0;
  int win_size = 0, win_shift = 0, full_len = 0, max_taps = 0;
  std::vector<float> lags;        // [num_states]
  std::vector<int> ar_first;      // [num_states]
  std::vector<int> ar_n;          // [num_states]
  std::vector<float> ar_w;        // [num_states][max_taps]
  LinearResampleHost resample;
  int64_t frames_available(int64_t n_down, bool input_finished, bool snip_edges) const;
};
int make_pitch_tables(const snf_pitch_options& o, PitchTablesHost* out);

// ---------------------------------------------------------------------------------------------
// device parameter blocks (passed by value to kernels)
// ---------------------------------------------------------------------------------------------
struct MelParams {
  // framing
  int win_len, win_shift, padded, half, log2_half;  // half = padded/2 (complex FFT size)
  int snip_edges, remove_dc, pow2;
  float preemph, dither;
  unsigned long long seed;
  const float* window;       // [win_len]
  const float2* tw_fft;      // [half/2]     exp(-2 pi i k / half)
  const float2* tw_unpack;   // [half/2 + 1] exp(-2 pi i k / padded)
  const float2* tw_dft;      // [padded]     exp(-2 pi i k / padded) (non power-of-two path)
  // epilogue
  int kind, ndims;
  int use_energy, raw_energy, htk_compat, use_log, use_power, need_raw, need_post;
  int has_floor;
  float log_energy_floor;
  int num_bins, num_ceps, compression;
  const int* mel_first;      // [n_warps][num_bins]
  const int* mel_size;       // [n_warps][num_bins]
  const int* mel_offset;     // [n_warps][num_bins] offset into mel_w
  const float* mel_w;
  const float* mel_w32;      // long-frame kernel: the weights with every filter zero-padded to 32-tap slices
  const int* mel_off32;      // [n_warps][num_bins] offset into mel_w32 (slice 0 of mel_w32 is all zeros)
  const float* dct;          // [num_ceps][num_bins]
  const float* lifter;       // [num_ceps] or nullptr
};

// fbank256x2_kernel: one record per frame pair.  Pairs are formed INSIDE an utterance (frames 2 m and
// 2 m + 1 counted from the utterance's first frame; an odd last frame is paired with itself), so the
// features of an utterance do not depend on what else is in the batch.
struct PairRec {
  int64_t start_a, start_b;  // first sample of the two frames (clamped into the utterance)
  int64_t frame_a;           // batch-wide frame index of frame a (frame b, if any, is frame_a + 1)
  int32_t utt1;              // utterance + 1
  int32_t flags;             // bit 0 / 1: frame a / b reflects; bit 2: frame b exists
};

struct BatchArgs {
  const int16_t* wave;
  const int64_t* sample_offsets;  // [n_utts+1]
  const int64_t* frame_offsets;   // [n_utts+1]
  const int32_t* utt_warp;        // [n_utts] index into the plan's warp tables, or nullptr
  const int64_t* frame_start;     // [total_frames] first sample of every frame (fast path only)
  const int32_t* frame_edge;      // [total_frames] snip_edges = false: utterance + 1 of edge frames
  const int32_t* frame_utt;       // [total_frames] utterance of every frame (generic kernel)
  const int32_t* blk_utt;         // [n_blocks] fast path with VTLN warps: utterance of every workgroup
  const int32_t* blk_set0;        // [n_blocks] ... and its first frame set inside that utterance
  const uint8_t* utt_mask;        // [n_utts] generic kernel: when set, only the utterances marked 1 are computed
  const uint64_t* frame_noise;    // [total_frames] fbank512b_kernel with dither: wave_noise_id of every frame
  const uint32_t* utt_noise;      // [n_utts] dither: a hash of 64 samples spread over every utterance (per call)
  const PairRec* pair_tab;        // [n_pairs] fbank256x2_kernel / fbank1024x2_kernel only
  int64_t n_pairs;
  // fbank256x2_kernel: a pair of frames shares the roundings of ONE complex transform - the louder frame sets the
  // error floor of both.  Pairs whose windowed energies differ by more than `split_ratio` are appended here, as two
  // records of one frame each (a frame paired with itself), `fix_count` counting the records; a second launch
  // of the kernel over that list (`n_pairs_dev` = the count, read on the device) rewrites their rows.
  PairRec* fix_tab;
  unsigned int* fix_count;
  const unsigned int* n_pairs_dev;
  float split_ratio;
  int64_t n_blocks;
  int64_t n_utts;
  int64_t total_frames;
};

// The noise stream of a frame (dither, delta-pitch noise) is keyed by what the frame IS, not by where it
// sits in the batch: its index inside its utterance, the utterance's length and 32 bits that stand for the
// utterance (a hash of 64 samples spread over the waveform / the bits of the first pitch value).  An utterance therefore draws the same noise alone and in
// any batch of the same call count (the per-call stream key is in `seed`), and the utterances of a batch draw
// different noise.  The reference draws from C rand(): no stream is "the" Kaldi one.
#if defined(__HIPCC__)
__device__ __forceinline__ uint64_t frame_noise_id(int64_t local_frame, int64_t utt_len, uint32_t first_bits) {
  uint64_t x = static_cast<uint64_t>(local_frame) * 0x9E3779B97F4A7C15ull +
               static_cast<uint64_t>(utt_len) * 0xC2B2AE3D27D4EB4Full + first_bits;
  x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
  x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
  return x ^ (x >> 31);
}
// ... of the frame `local_frame` of utterance `u` of a waveform batch.  What stands for "the utterance" is a
// 32-bit hash of 64 of its samples, spread evenly from its first to its last (launch_build_utt_noise, once
// per call; round 4 - the first two samples alone, as before, made every equal-length utterance that begins
// with digital silence draw the same dither: ADVICE r03).
__device__ __forceinline__ uint64_t wave_noise_id(const BatchArgs& b, int64_t u, int64_t local_frame) {
  const int64_t n = b.sample_offsets[u + 1] - b.sample_offsets[u];
  return frame_noise_id(local_frame, n, b.utt_noise[u]);
}
#endif

struct PlpParams {
  int num_bins, lpc_order, num_ceps, use_energy, htk_compat, has_floor, rasta;
  int exact_pow;  // SNF_PLP_EXACT_POW=1: the correctly-rounded powf in plp_tail_exact_kernel (A/B runs)
  float compress_factor, cepstral_scale;
  double log_energy_floor;
  const float* eql;     // [n_warps][num_bins]
  const float* idft;    // [lpc_order+1][num_bins+2]
  const float* lifter;  // [num_ceps] or nullptr
};

struct DeltaParams {
  int order, window, n_scales;
  const float* scales;   // concatenated
  const int* dims;       // [order+1]
};

// ---------------------------------------------------------------------------------------------
// kernel launchers (kernels_*.hip)
// ---------------------------------------------------------------------------------------------
// Fused frame extraction -> window -> FFT -> power -> epilogue.  `out` is [total_frames, out_cols];
// for kind PLP it receives linear mel energies [total_frames, num_bins] and `energy_out`
// [total_frames] the linear frame energy (double; plp_tail_kernel floors it and takes the double log).
int launch_mel_features(const MelParams& p, const BatchArgs& b, float* out, int out_cols,
                        double* energy_out, hipStream_t stream);
int launch_rasta(float* mel, const BatchArgs& b, int num_bins, hipStream_t stream);
int launch_plp_tail(const PlpParams& p, const BatchArgs& b, const float* mel, const double* energy,
                    float* out, hipStream_t stream);
// `tile_info`: scratch of 4 (total_frames / 32 + 2) int64 (nullptr: per-element kernels only), rebuilt
// from the offsets table when `build_info` is set (the caller keeps it while the table stays the same)
int launch_deltas(const DeltaParams& p, const float* in, int in_cols, const int64_t* frame_offsets,
                  int64_t n_utts, int64_t total_frames, float* out, int64_t* tile_info, bool build_info,
                  hipStream_t stream);


// ---- register-resident 512-point fast path (kernels_fbank512.hip) ----------------------------------
constexpr int kFast512MaxBins = 64;    // mel bins: 16 MFMA blocks of 4 bins
constexpr int kFast512FusedSets = 82;  // fused deltas: frame sets per workgroup (kernels_fbank512.hip)
struct Fast512Params {
  int win_len, win_shift, remove_dc, snip_edges;
  float preemph, dither;
  unsigned long long seed;
  int kind, out_cols, use_energy, need_raw, need_post, htk_compat, use_log, has_floor;
  float log_energy_floor;
  int num_bins, num_ceps, compression;
  // mel filterbank as a chain of v_mfma_f32_4x4x1_16b_f32: 16 blocks of (4 mel bins x 4 frames), one
  // FFT bin per block and instruction; `mm_quads` = chain length / 4, `mm_levels` = the most blocks a
  // group of 4 bins is split into along the FFT bins (partial sums are added through DPP)
  int mm_quads, mm_levels;
  int dd_quads;              // MFCC: DCT-II chain length / 4 (mel bins per K partition / 4)
  int dd_groups;             // MFCC, vector-pipe DCT: groups of 4 mel bins (ceil(num_bins / 4))
  int dct_mfma;              // MFCC: 1 = DCT-II as the MFMA chain (SNF_DCT_MFMA=1; measured slower), 0 = vector pipe
  int dual;                  // frames pad to 256 samples: two frames per 16-lane row (fbank256x2_kernel)
  int fused_delta;           // MFCC: rows are [cepstra | delta | delta-delta] (order 2, window 2)
  const float* delta_scales; // ... composite scales of the three orders, 1 + 5 + 9 floats (device)
  int table_floats;          // total floats of the packed table blob below (warp 1.0)
  int table_stride;          // floats between the blobs of consecutive warp factors (VTLN)
  // one packed blob, copied to LDS at kernel start:
  //   header[16] | float2 win[16][18] | float2 tw16[16][18] | float2 tw512[16][10] |
  //   float4 mm_a[mm_quads][64] | mm_lane[5][64] | float4 dd_a[dd_quads][64] | float lifter[16] |
  //   float4 dd_v[dd_groups][16]
  const float* tables;
  int off_mm_a, off_mm_lane, off_dd_a, off_lifter, off_dd_v;  // float offsets into the blob
};

bool fast512_eligible(const MelParams& mp, bool any_warp);
// frames that pad to 256 samples: the two-frames-per-row form (flat batches without VTLN warps)
bool fast512_dual_eligible(const MelParams& mp);
int fast512_build(const MelParams& mp, const std::vector<float>& window, const MelBanksHost& mb,
                  const std::vector<float>& dct, const std::vector<float>& lifter, bool dual,
                  std::vector<float>* blob, Fast512Params* out);
int launch_build_pair_table(const int64_t* d_frame_offsets, const int64_t* d_sample_offsets,
                            const int64_t* d_pair_offsets, int64_t n_utts, int64_t n_pairs, int win_shift,
                            int win_len, int snip_edges, PairRec* d_pairs, hipStream_t stream);
int launch_build_frame_start(const int64_t* d_frame_offsets, const int64_t* d_sample_offsets,
                             int64_t n_utts, int64_t total_frames, int64_t total_samples, int win_shift,
                             int win_len, int snip_edges, int64_t* d_frame_start, int32_t* d_frame_edge,
                             int32_t* d_frame_utt, hipStream_t stream);
int launch_fbank512(const Fast512Params& p, const BatchArgs& b, float* out, int out_cols,
                    double* energy_out, hipStream_t stream);
// per-call table of the frames' noise keys (wave_noise_id: the dither of fbank512b_kernel; the keys hold
// the first two samples of the utterance, so they are rebuilt with every batch)
int launch_build_frame_noise(const BatchArgs& b, uint64_t* d_keys, hipStream_t stream);
// per-call table of the utterances' noise words (every dithering kernel reads it through wave_noise_id)
int launch_build_utt_noise(const BatchArgs& b, uint32_t* d_words, hipStream_t stream);
// the occupancy-first form of the same kernel (kernels_fbank512b.hip): flat, snip_edges batches
bool fbank512b_eligible(const Fast512Params& p, const BatchArgs& b);
int launch_fbank512b(const Fast512Params& p, const BatchArgs& b, float* out, int out_cols,
                     double* energy_out, hipStream_t stream);

// ---- register-resident 2048-point path (kernels_fbank2048.hip): 44.1 / 48 kHz frames, 32 kHz zero-extended
bool fbank2048_eligible(const MelParams& mp);
void fbank2048_tables(const MelParams& mp, const std::vector<float>& window, std::vector<float>* blob);
int launch_fbank2048(const MelParams& p, const BatchArgs& b, const float* tables, float* out, int out_cols,
                     double* energy_out, hipStream_t stream);

// ---- two 1024-sample frames per complex transform (kernels_fbank1024x2.hip): 22.05 / 32 kHz frames; `b` carries
// the pair table of launch_build_pair_table
bool fbank1024x2_eligible(const MelParams& mp);
void fbank1024x2_tables(const MelParams& mp, const std::vector<float>& window, std::vector<float>* blob);
int launch_fbank1024x2(const MelParams& p, const BatchArgs& b, const float* tables, float* out, int out_cols,
                       double* energy_out, hipStream_t stream);

// MFCC tail behind the filterbank kernels (kernels_dct.hip): DCT, lifter, energy and htk conventions on rows of
// [log energy |] log-mel energies
int launch_mfcc_dct(const float* in, int in_cols, int num_bins, int num_ceps, const float* dct_t, const float* lifter,
                    int use_energy, int htk_compat, int64_t total_frames, float* out, int out_cols,
                    hipStream_t stream);

struct PitchDevTables {
  int first_lag, last_lag, num_lags, num_states, win_size, win_shift, full_len;
  int ar_max_taps, rs_in_unit, rs_out_unit, rs_max_taps;
  int snip_edges, recompute_frame;
  float soft_min_f0, inter_frame_factor, nccf_ballast;
  const float* lags;      // [num_states]
  const int* ar_first;    // [num_states]
  const int* ar_n;        // [num_states]
  const float* ar_w;      // [num_states][ar_max_taps]
  // the same taps laid out for the matrix pipe (pitch_nccf_kernel): states in groups of 64 = 16 quads;
  // a quad shares a window of ar_quad_taps lags starting at ar_quad_base (zero weight where a state has
  // no tap); ar_groups is even (a zero-weight group pads an odd count)
  int ar_groups, ar_quad_taps;
  const float* ar_quad_w;   // [ar_groups][ar_quad_taps / 4][64 = quad * 4 + state][4]
  const int* ar_quad_base;  // [ar_groups][16]
  const int* rs_first;    // [rs_out_unit]
  const int* rs_ntaps;    // [rs_out_unit]
  const float* rs_w;      // [rs_out_unit][rs_max_taps]
};
struct PitchBatch {
  const int16_t* wave;
  const int64_t* sample_offsets;  // [n_utts+1]
  const int64_t* frame_offsets;   // [n_utts+1]
  const int64_t* down_offsets;    // [n_utts+1] offsets into the downsampled scratch
  const int64_t* down_phase1;     // [n_utts] samples available before the flush
  const int64_t* frames_phase1;   // [n_utts] frames processed before the flush
  const int32_t* order;           // [n_utts] utterance handled by every tracker slot (longest first), or nullptr
  int64_t n_utts, total_frames, total_down;
  int64_t max_down;               // longest downsampled utterance
};
// scratch of one pitch batch (all in HBM, owned by the plan)
struct PitchScratch {
  float* down;       // [total_down]                  signal resampled to resample_freq
  float* ub;         // [n_utts][6]                   NCCF ballasts of every utterance (pitch_stats_kernel)
  float* nccf_res;   // [total_frames][num_states]    NCCF (with ballast) at the lag of every state
  float* pov_nccf;   // [total_frames][num_lags]      NCCF without ballast at the integer lags
  float* anp;        // [total_frames]                average norm product (RecomputeBacktraces)
  int16_t* backptr;  // [total_frames][num_states]
  int32_t* states;   // [total_frames]                traceback
  int4* frame_meta;  // [total_frames]                window start / existing samples / ballast of every frame
};
// resample -> signal statistics -> frame-parallel NCCF + lag resampling -> Viterbi per utterance ->
// traceback + POV output
int launch_pitch(const PitchDevTables& t, const PitchBatch& b, const PitchScratch& w, float* out,
                 hipStream_t stream);

int launch_vad(const snf_vad_options& o, const float* in, int in_cols, const int64_t* frame_offsets,
               int64_t n_utts, int64_t total_frames, float* thr_scratch, float* out,
               hipStream_t stream);
int launch_cmvn_stats(const float* in, int in_cols, const int64_t* frame_offsets,
                      const float* weights, int64_t n_utts, double* stats, hipStream_t stream);
int launch_cmvn_apply(const float* in, int in_cols, const int64_t* frame_offsets, int64_t n_utts,
                      int64_t max_frames, const int32_t* group, const float* norm, int scale_it,
                      float* out, hipStream_t stream);
int launch_concat_columns(const float* a, int cols_a, const int64_t* d_off_a, const float* b, int cols_b,
                          const int64_t* d_off_b, int64_t n_utts, float* out, const int64_t* d_off_out,
                          int64_t total_rows, hipStream_t stream);
int launch_count_nonfinite(const float* x, uint64_t n, unsigned long long* d_count, hipStream_t stream);
int launch_sliding_cmvn(const snf_sliding_cmvn_options& o, const float* in, int in_cols,
                        const int64_t* frame_offsets, int64_t n_utts, float* out,
                        hipStream_t stream);

struct PitchPostParams {
  snf_pitch_post_options o;
  int ndims;
  unsigned long long seed;
};
int launch_pitch_post(const PitchPostParams& p, const float* in, const int64_t* frame_offsets,
                      int64_t n_utts, int64_t total_frames, float* out, hipStream_t stream);

}  // namespace snf

#endif  // SNF_INTERNAL_H_
