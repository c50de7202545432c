// C ABI of libshennong_hip.so (include/shennong_amd.h): plans, device tables, batch orchestration.
//
// A plan owns (a) the immutable tables Kaldi would rebuild per utterance (window, FFT twiddles, mel
// banks per VTLN warp, DCT, lifter, IDFT bases, resampler taps), resident in HBM, (b) grow-only
// device scratch for the host-pointer entry points, (c) one HIP stream and the events that time the
// kernels on that stream.
#include <atomic>
#include <cmath>
#include <cstring>
#include <memory>

#include "snf_internal.h"

using namespace snf;

namespace {

// Out-of-memory hook (snf_set_oom_hook): the host side parks freed device buffers in a pool of its own
// (shennong_amd/_backend.py, up to 8 GiB); an allocation of the library that fails asks it to give them
// back and tries once more.
std::atomic<snf_oom_hook> g_oom_hook{nullptr};
hipError_t malloc_with_hook(void** p, size_t bytes) {
  hipError_t e = hipMalloc(p, bytes);
  if (e == hipErrorOutOfMemory) {
    if (snf_oom_hook hook = g_oom_hook.load()) {
      (void)hipGetLastError();
      // the hook frees pooled blocks of EVERY device and binds each one to do it: the retry (and the
      // launches of the plan call we are in the middle of) must find the calling thread on its own device
      int dev = -1;
      const bool have_dev = hipGetDevice(&dev) == hipSuccess;
      hook();
      if (have_dev) (void)hipSetDevice(dev);
      (void)hipGetLastError();
      e = hipMalloc(p, bytes);
    }
  }
  return e;
}

struct DevBuf {
  void* p = nullptr;
  size_t cap = 0;
  int ensure(size_t bytes) {
    if (bytes <= cap) return SNF_OK;
    if (p) (void)hipFree(p);
    p = nullptr;
    cap = 0;
    size_t want = bytes + bytes / 8 + 256;
    SNF_HIP_CHECK(malloc_with_hook(&p, want));
    cap = want;
    return SNF_OK;
  }
  template <typename T>
  int upload(const std::vector<T>& v, hipStream_t s) {
    const size_t bytes = sizeof(T) * v.size();
    int rc = ensure(bytes > 0 ? bytes : 16);
    if (rc) return rc;
    if (bytes) {
      // the source is a short-lived pageable host vector: finish the copy before returning
      SNF_HIP_CHECK(hipMemcpyAsync(p, v.data(), bytes, hipMemcpyHostToDevice, s));
      SNF_HIP_CHECK(hipStreamSynchronize(s));
    }
    return SNF_OK;
  }
  template <typename T>
  T* as() const { return static_cast<T*>(p); }
  ~DevBuf() {
    if (p) (void)hipFree(p);
  }
};

constexpr int kMaxSlots = 6;
constexpr float kPairSplitRatio = 8.0f;   // fbank256x2_kernel: windowed-energy ratio beyond which a pair is redone
                                          // one frame at a time (kernels_fbank1024x2.hip holds the same number)

// Scratch and a stream of its own for the plan-less device helpers (snf_concat_columns_device,
// snf_count_nonfinite_device), per calling thread and device: no hipMalloc / hipFree per call (both wait for the
// whole device) and no device-wide wait at the end - the batches a pipeline keeps in flight on other threads, and
// the pitch tracker beside this thread, go on undisturbed.  Lives as long as the thread.
struct ThreadScratch {
  hipStream_t stream = nullptr;
  DevBuf buf;
};
ThreadScratch* thread_scratch(int device_id) {
  thread_local std::vector<std::pair<int, ThreadScratch*>> mine;
  for (auto& e : mine)
    if (e.first == device_id) return e.second;
  ThreadScratch* t = new ThreadScratch;
  if (hipStreamCreateWithFlags(&t->stream, hipStreamNonBlocking) != hipSuccess) {
    delete t;
    snf::set_error(SNF_E_HIP, "hipStreamCreate failed");
    return nullptr;
  }
  mine.emplace_back(device_id, t);
  return t;
}

}  // namespace

struct snf_plan {
  snf_options o{};
  int device = 0;
  hipStream_t stream = nullptr;
  std::mutex mu;
  std::mutex host_mu;  // a host-pointer call owns the plan's staging scratch (s_wave / s_in / s_out)
                       // from its upload to its download: whole-call lock, taken before `mu`
  int kind = 0, ndims = 0;

  // mel family
  MelParams mp{};
  DevBuf d_window, d_tw_fft, d_tw_unpack, d_tw_dft, d_dct, d_lifter, d_idft;
  std::vector<float> warps;  // distinct VTLN warp factors seen so far (index = warp id)
  std::string base_banks_error;  // PLP: why the unwarped banks (id 0) cannot be built; empty = they can
  std::vector<MelBanksHost> banks;
  DevBuf d_mel_first, d_mel_size, d_mel_off, d_mel_w, d_eql, d_mel_w32, d_mel_off32;
  bool warps_dirty = true;
  PlpParams pp{};
  // register-resident fast path for the 512-point configuration
  bool fast512 = false;
  Fast512Params fp{};
  DevBuf d_fast_tables;
  // filterbanks of 65 ... 128 bins (fbank-80): the 64-bin kernel twice, over the two halves of the bank - the
  // second launch with `fp_hi`, writing `wide_offset` floats into every row
  bool wide = false;
  Fast512Params fp_hi{};
  DevBuf d_fast_tables_hi;
  int wide_offset = 0;
  // MFCC through the filterbank kernel + mfcc_dct_kernel (more than 16 cepstra, or more than 64 bins)
  bool mfcc_via_fbank = false;
  DevBuf d_dct_t;
  // ... and its per-warp-factor tables (VTLN): one blob per warp id, `fp_warp.table_stride` apart
  std::vector<float> h_window, h_dct, h_lifter;
  Fast512Params fp_warp{};
  DevBuf d_fast_warp_tables, s_blk_utt, s_blk_set0, s_noise, s_unoise;
  size_t fast_warps_built = 0;   // number of warp ids covered by d_fast_warp_tables
  bool fast_warps_ok = true;     // false: some warp's banks do not fit the fast kernel
  // register-resident 2048-point path (frames that pad to 2048 or 1024 samples)
  bool fast2048 = false;
  bool pair1024 = false;      // frames that pad to 1024 samples: two per transform (kernels_fbank1024x2.hip)
  DevBuf d_long_tables;

  // delta (post-processor plans, and MFCC plans with append_deltas)
  DeltaParams dp{};
  DevBuf d_scales, d_dims;
  // append_deltas: true = the MFCC kernel writes [T, num_ceps] to a scratch and the delta kernel forms the
  // rows (two launches: 1.17 + 0.11 ms per 2.98 M frames); false = fbank512_kernel's fused mode (one
  // launch, 1.46 ms: it loses to the chain, profiles/NOTEBOOK.md 4.4; SNF_FUSED_DELTA=1 selects it)
  bool chain_deltas = false;
  DevBuf s_cep, s_tile;
  bool tile_valid = false;  // s_tile describes the cached frame offsets table

  // pitch
  PitchTablesHost pt;
  PitchDevTables pd{};
  DevBuf d_lags, d_ar_first, d_ar_n, d_ar_w, d_ar_quad_w, d_ar_quad_base, d_rs_first, d_rs_ntaps, d_rs_w;
  PitchPostParams ppost{};

  // scratch (host-pointer entry points and intermediates)
  DevBuf s_wave, s_out, s_in, s_soff, s_foff, s_uwarp, s_mel, s_energy;
  DevBuf s_down, s_stats, s_bp, s_doff, s_dp1, s_fp1, s_states, s_setidx, s_edge, s_futt, s_poff, s_pairs, s_umask, s_fix, s_fixcount;
  DevBuf s_pres, s_anp;  // pitch: NCCF at the lag of every state [frames, states], norm average [frames]
  bool setidx_valid = false;
  bool pairs_valid = false;   // s_pairs / n_pairs describe the cached offsets tables
  int64_t n_pairs = 0;
  // calls that draw random numbers (dither, delta-pitch noise) so far: every call gets its own noise
  // stream (the reference draws from one global rand(): two calls never repeat the same samples)
  uint64_t noise_calls = 0;

  // last uploaded offsets tables (re-validated / re-uploaded only when they change)
  std::vector<int64_t> h_soff, h_foff;
  int post_table_cols = -1;  // post plans: input width the cached tile records of h_foff were built for

  // timing
  hipEvent_t ev[kMaxSlots + 1] = {};
  const char* slot_name[kMaxSlots + 1] = {};
  int n_slots = 0;
  bool events_valid = false;

  ~snf_plan() {
    for (auto& e : ev)
      if (e) (void)hipEventDestroy(e);
    if (stream) (void)hipStreamDestroy(stream);
  }
};

namespace {

int guard_device(const snf_plan* plan) {
  SNF_HIP_CHECK(hipSetDevice(plan->device));
  return SNF_OK;
}

int ilog2(int v) {
  int l = 0;
  while ((1 << l) < v) ++l;
  return l;
}

// ---- mel-family plan -----------------------------------------------------------------------------
int build_mel_plan(snf_plan* plan) {
  const snf_options& o = plan->o;
  snf_frame_options fo = o.frame;
  if (plan->kind == SNF_KIND_ENERGY && o.raw_energy) {
    // reference processor/energy.py:150-154: raw energy = no pre-emphasis, rectangular window
    fo.preemph_coeff = 0.0f;
    fo.window_type = SNF_WINDOW_RECTANGULAR;
  }
  MelParams& p = plan->mp;
  p.win_len = window_size(fo);
  p.win_shift = window_shift(fo);
  p.padded = padded_window_size(fo);
  if (p.win_shift <= 0) return set_error(SNF_E_RUNTIME, "frame shift is shorter than one sample");
  if (p.win_len < 2) return set_error(SNF_E_RUNTIME, "frame length must be at least 2 samples");
  // (the frame energy has no spectrum: an odd window - e.g. 25 ms at 22.05 kHz without rounding to a
  // power of two - is fine for it; Kaldi's RealFft asserts an even size for everything else)
  if (p.padded % 2 != 0 && plan->kind != SNF_KIND_ENERGY)
    return set_error(SNF_E_RUNTIME, "padded window size must be even (real FFT)");
  p.half = p.padded / 2;
  p.pow2 = (p.padded & (p.padded - 1)) == 0;
  p.log2_half = p.pow2 ? ilog2(p.half) : 0;
  p.snip_edges = fo.snip_edges;
  p.remove_dc = fo.remove_dc_offset;
  p.preemph = fo.preemph_coeff;
  p.dither = fo.dither;
  p.seed = o.seed;
  p.kind = plan->kind;

  std::vector<float> window;
  int rc = make_window(fo, &window);
  if (rc) return rc;
  if ((rc = plan->d_window.upload(window, plan->stream))) return rc;
  p.window = plan->d_window.as<float>();

  constexpr double kTwoPi = 6.283185307179586476925286766559005;
  std::vector<float2> tw;
  if (p.pow2) {
    tw.resize(p.half / 2 > 0 ? p.half / 2 : 1);
    for (int k = 0; k < static_cast<int>(tw.size()); ++k) {
      const double a = -kTwoPi * k / p.half;
      tw[k] = make_float2(static_cast<float>(std::cos(a)), static_cast<float>(std::sin(a)));
    }
    if ((rc = plan->d_tw_fft.upload(tw, plan->stream))) return rc;
    tw.resize(p.half / 2 + 1);
    for (int k = 0; k < static_cast<int>(tw.size()); ++k) {
      const double a = -kTwoPi * k / p.padded;
      tw[k] = make_float2(static_cast<float>(std::cos(a)), static_cast<float>(std::sin(a)));
    }
    if ((rc = plan->d_tw_unpack.upload(tw, plan->stream))) return rc;
    p.tw_fft = plan->d_tw_fft.as<float2>();
    p.tw_unpack = plan->d_tw_unpack.as<float2>();
    p.tw_dft = nullptr;
  } else {
    tw.resize(p.padded);
    for (int k = 0; k < p.padded; ++k) {
      const double a = -kTwoPi * k / p.padded;
      tw[k] = make_float2(static_cast<float>(std::cos(a)), static_cast<float>(std::sin(a)));
    }
    if ((rc = plan->d_tw_dft.upload(tw, plan->stream))) return rc;
    p.tw_dft = plan->d_tw_dft.as<float2>();
    p.tw_fft = nullptr;
    p.tw_unpack = nullptr;
  }

  p.use_energy = o.use_energy;
  p.raw_energy = o.raw_energy;
  p.htk_compat = o.htk_compat;
  p.use_log = o.use_log_fbank;
  p.use_power = o.use_power;
  p.num_bins = o.mel.num_bins;
  p.num_ceps = o.num_ceps;
  p.compression = o.compression;
  p.has_floor = o.energy_floor > 0.0f;
  p.log_energy_floor = p.has_floor ? logf(o.energy_floor) : 0.0f;
  p.dct = nullptr;
  p.lifter = nullptr;

  switch (plan->kind) {
    case SNF_KIND_SPECTROGRAM:
      plan->ndims = p.half + 1;
      p.need_raw = o.raw_energy ? 1 : 0;
      p.need_post = o.raw_energy ? 0 : 1;
      p.num_bins = 0;
      break;
    case SNF_KIND_FBANK:
      plan->ndims = o.mel.num_bins + (o.use_energy ? 1 : 0);
      break;
    case SNF_KIND_MFCC:
      plan->ndims = o.append_deltas ? 3 * o.num_ceps : o.num_ceps;
      if (o.append_deltas && (o.delta_order != 2 || o.delta_window != 2))
        return set_error(SNF_E_INVALID, "append_deltas supports delta_order 2 / delta_window 2 only");
      break;
    case SNF_KIND_PLP:
      plan->ndims = o.num_ceps;
      break;
    case SNF_KIND_ENERGY:
      if (o.compression != SNF_COMPRESS_OFF && o.compression != SNF_COMPRESS_LOG &&
          o.compression != SNF_COMPRESS_SQRT)
        return set_error(SNF_E_INVALID, "compression must be in off, log, sqrt");
      plan->ndims = 1;
      p.need_raw = 0;
      p.need_post = 0;
      p.num_bins = 0;
      break;
    default:
      return set_error(SNF_E_INVALID, "not a mel-family kind");
  }
  if (plan->kind != SNF_KIND_SPECTROGRAM && plan->kind != SNF_KIND_ENERGY) {
    p.need_raw = (o.use_energy && o.raw_energy) ? 1 : 0;
    p.need_post = (o.use_energy && !o.raw_energy) ? 1 : 0;
    // Kaldi builds the warp-1.0 banks in the computer's constructor: option errors surface here.  Not for
    // PLP: the reference's own recipe builds the banks of a warp factor when the first frame asks for them
    // (shennong/processor/plp.py:482-494, :559) - an utterance without frames, or a batch in which every
    // utterance carries another warp factor, never sees the errors of the unwarped banks.  The plan then
    // holds zero-weight placeholders as bank 0 and reports the error when an utterance with frames needs it.
    MelBanksHost mb;
    if ((rc = make_mel_banks(o.mel, fo, 1.0f, &mb))) {
      if (plan->kind != SNF_KIND_PLP || rc != SNF_E_RUNTIME || o.mel.num_bins < 3 ||
          padded_window_size(fo) % 2 != 0)
        return rc;
      plan->base_banks_error = last_error();
      make_placeholder_banks(o.mel, fo, &mb);
    }
    plan->warps.assign(1, 1.0f);
    plan->banks.assign(1, mb);
    plan->warps_dirty = true;
  }
  if (plan->kind == SNF_KIND_MFCC) {
    if (o.num_ceps > o.mel.num_bins)
      return set_error(SNF_E_RUNTIME, "num-ceps cannot be larger than num-mel-bins. It should be "
                                      "smaller or equal. You provided num-ceps: " +
                                          std::to_string(o.num_ceps) + "  and num-mel-bins: " +
                                          std::to_string(o.mel.num_bins));
    if (o.num_ceps <= 0) return set_error(SNF_E_RUNTIME, "num-ceps must be strictly positive");
    std::vector<float> dct, lifter;
    make_dct_matrix(o.num_ceps, o.mel.num_bins, &dct);
    if ((rc = plan->d_dct.upload(dct, plan->stream))) return rc;
    p.dct = plan->d_dct.as<float>();
    if (o.cepstral_lifter != 0.0f) {
      make_lifter(o.cepstral_lifter, o.num_ceps, &lifter);
      if ((rc = plan->d_lifter.upload(lifter, plan->stream))) return rc;
      p.lifter = plan->d_lifter.as<float>();
    }
  }
  if (plan->kind == SNF_KIND_PLP) {
    if (o.num_ceps <= 0 || o.num_ceps > o.lpc_order + 1)
      return set_error(SNF_E_INVALID, "We must have 0 < num_ceps <= lpc_order+1");
    PlpParams& q = plan->pp;
    q.num_bins = o.mel.num_bins;
    q.lpc_order = o.lpc_order;
    q.num_ceps = o.num_ceps;
    q.use_energy = o.use_energy;
    q.htk_compat = o.htk_compat;
    q.has_floor = o.energy_floor > 0.0f;
    q.log_energy_floor = q.has_floor ? std::log(static_cast<double>(o.energy_floor)) : 0.0;
    q.rasta = o.rasta;
    q.compress_factor = o.compress_factor;
    q.exact_pow = getenv("SNF_PLP_EXACT_POW") != nullptr ? 1 : 0;
    q.cepstral_scale = o.cepstral_scale;
    std::vector<float> idft, lifter;
    make_idft_bases(o.lpc_order + 1, o.mel.num_bins + 2, &idft);
    if ((rc = plan->d_idft.upload(idft, plan->stream))) return rc;
    q.idft = plan->d_idft.as<float>();
    q.lifter = nullptr;
    if (o.cepstral_lifter != 0.0f) {
      make_lifter(o.cepstral_lifter, o.num_ceps, &lifter);
      if ((rc = plan->d_lifter.upload(lifter, plan->stream))) return rc;
      q.lifter = plan->d_lifter.as<float>();
    }
  }
  p.ndims = plan->ndims;
  if (plan->kind == SNF_KIND_MFCC && o.append_deltas) {
    const char* knob = getenv("SNF_FUSED_DELTA");
    plan->chain_deltas = !(knob && knob[0] == '1');
    std::vector<float> scales;
    std::vector<int> dims;
    make_delta_scales(2, 2, &scales, &dims);
    if ((rc = plan->d_scales.upload(scales, plan->stream))) return rc;
    if ((rc = plan->d_dims.upload(dims, plan->stream))) return rc;
    plan->dp.order = 2;
    plan->dp.window = 2;
    plan->dp.n_scales = static_cast<int>(scales.size());
    plan->dp.scales = plan->d_scales.as<float>();
    plan->dp.dims = plan->d_dims.as<int>();
  }
  const bool want_fused = plan->kind == SNF_KIND_MFCC && o.append_deltas && !plan->chain_deltas;
  if (want_fused && (!fast512_eligible(p, false) || o.num_ceps > 16))
    return set_error(SNF_E_INVALID, "append_deltas needs frames that pad to 512 samples (the register-"
                                    "resident path); chain a delta plan for this configuration");
  if (fast512_eligible(p, false)) {
    std::vector<float> dct_h, lifter_h, blob;
    if (plan->kind == SNF_KIND_MFCC) {
      make_dct_matrix(o.num_ceps, o.mel.num_bins, &dct_h);
      if (o.cepstral_lifter != 0.0f) make_lifter(o.cepstral_lifter, o.num_ceps, &lifter_h);
    }
    const MelBanksHost no_banks;
    const bool dual = !want_fused && fast512_dual_eligible(p);
    rc = fast512_build(p, window, plan->banks.empty() ? no_banks : plan->banks[0], dct_h, lifter_h, dual,
                       &blob, &plan->fp);
    if (rc < 0) return rc;
    if (rc == 0) {  // rc > 0: shape not covered by the fast kernel, keep the generic one
      if ((rc = plan->d_fast_tables.upload(blob, plan->stream))) return rc;
      plan->fp.tables = plan->d_fast_tables.as<float>();
      plan->fast512 = true;
      if (want_fused) {
        // (the fused form keeps 14 waves' tiles + the cepstra of 336 frames in LDS beside the tables)
        if (((static_cast<size_t>(plan->fp.table_floats) * 4 + 255) & ~static_cast<size_t>(255)) +
                14 * 4 * 2176 + sizeof(float) * 4 * (kFast512FusedSets + 2) * 16 > 160 * 1024)
          return set_error(SNF_E_INVALID, "append_deltas: the mel / DCT tables of this configuration leave no "
                                          "room for the fused form in LDS; chain a delta plan");
        plan->fp.fused_delta = 1;
        plan->fp.delta_scales = plan->d_scales.as<float>();
      }
      plan->h_window = window;
      plan->h_dct = dct_h;
      plan->h_lifter = lifter_h;
    }
  }
  // Filterbank plans the 64-bin kernel covers in two launches, and MFCC plans it covers up to the log-mel energies
  // (round 6; the generic wave-per-frame kernel until then, 7 x slower per frame):
  //  * a filterbank of 65 ... 128 bins (fbank-80 at 16 kHz is a common front end): the kernel's matrix-pipe mel
  //    chain holds 16 blocks of 4 bins; a wider bank runs it TWICE, over the lower and the upper half of the bins
  //    (twice the transform arithmetic, still 3.5 x faster than the generic kernel): two parameter sets, the
  //    second one writing behind the columns of the first.  The energy column goes with the half it is adjacent to;
  //  * MFCC with more than 16 cepstra (Kaldi's "hires" MFCC: 40 bins, 40 cepstra) or more than 64 bins: the
  //    filterbank kernel writes [log energy |] log-mel rows to a scratch, mfcc_dct_kernel forms the cepstra.
  // -> 0: plan->fp (and fp_hi) are built, 1: not covered, < 0: error
  auto build_fbank_fast = [&](const MelParams& pf) -> int {
    if (plan->banks.empty() || pf.padded != 512) return 1;
    const MelBanksHost& mb = plan->banks[0];
    std::vector<float> none;
    if (pf.num_bins <= kFast512MaxBins) {
      if (!fast512_eligible(pf, false)) return 1;
      std::vector<float> blob;
      const int rc2 = fast512_build(pf, window, mb, none, none, false, &blob, &plan->fp);
      if (rc2 != 0) return rc2;
      if (int rc3 = plan->d_fast_tables.upload(blob, plan->stream)) return rc3;
      plan->fp.tables = plan->d_fast_tables.as<float>();
      return 0;
    }
    if (pf.num_bins > 2 * kFast512MaxBins || getenv("SNF_DISABLE_WIDE512")) return 1;
    const int nb = pf.num_bins, lo_n = ((nb + 1) / 2 + 3) & ~3, hi_n = nb - lo_n;
    auto half_of = [&](int first_bin, int count, MelBanksHost* out) {
      out->num_bins = count;
      out->num_fft_bins = mb.num_fft_bins;
      for (int m = first_bin; m < first_bin + count; ++m) {
        out->first.push_back(mb.first[m]);
        out->size.push_back(mb.size[m]);
        out->offset.push_back(static_cast<int>(out->w.size()));
        out->w.insert(out->w.end(), mb.w.begin() + mb.offset[m], mb.w.begin() + mb.offset[m] + mb.size[m]);
        out->center_freqs.push_back(mb.center_freqs[m]);
      }
    };
    MelBanksHost mb_lo, mb_hi;
    half_of(0, lo_n, &mb_lo);
    half_of(lo_n, hi_n, &mb_hi);
    MelParams p_lo = pf, p_hi = pf;
    p_lo.num_bins = lo_n;
    p_hi.num_bins = hi_n;
    const bool energy_first = pf.use_energy && !pf.htk_compat;   // column 0; otherwise (htk) the last column
    MelParams& bare = energy_first ? p_hi : p_lo;                // the half that does not write the energy
    if (pf.use_energy) bare.use_energy = bare.need_raw = bare.need_post = 0;
    std::vector<float> blob_lo, blob_hi;
    if (hi_n < 3 || !fast512_eligible(p_lo, false) || !fast512_eligible(p_hi, false)) return 1;
    int rc2 = fast512_build(p_lo, window, mb_lo, none, none, false, &blob_lo, &plan->fp);
    if (rc2 == 0) rc2 = fast512_build(p_hi, window, mb_hi, none, none, false, &blob_hi, &plan->fp_hi);
    if (rc2 != 0) return rc2;
    if (int rc3 = plan->d_fast_tables.upload(blob_lo, plan->stream)) return rc3;
    if (int rc3 = plan->d_fast_tables_hi.upload(blob_hi, plan->stream)) return rc3;
    plan->fp.tables = plan->d_fast_tables.as<float>();
    plan->fp_hi.tables = plan->d_fast_tables_hi.as<float>();
    plan->wide_offset = lo_n + (energy_first ? 1 : 0);
    plan->wide = true;
    return 0;
  };
  if (!plan->fast512 && plan->kind == SNF_KIND_FBANK && p.num_bins > kFast512MaxBins) {
    const int rc2 = build_fbank_fast(p);
    if (rc2 < 0) return rc2;
    if (rc2 == 0) {
      plan->fast512 = true;
      plan->h_window = window;
    }
  }
  if (!plan->fast512 && plan->kind == SNF_KIND_MFCC && !want_fused && !getenv("SNF_DISABLE_MFCC_VIA_FBANK")) {
    MelParams pf = p;
    pf.kind = SNF_KIND_FBANK;
    pf.use_log = 1;
    pf.use_power = 1;
    pf.htk_compat = 0;     // (the energy in column 0 of the scratch rows, whatever the cepstra's layout)
    pf.num_ceps = 0;
    pf.dct = nullptr;
    pf.lifter = nullptr;
    const int rc2 = build_fbank_fast(pf);
    if (rc2 < 0) return rc2;
    if (rc2 == 0) {
      // the DCT matrix transposed, rows of num_ceps rounded up to 16 (mfcc_dct_kernel reads sixteen cepstra at a time)
      std::vector<float> dct_h, dct_t;
      make_dct_matrix(o.num_ceps, o.mel.num_bins, &dct_h);
      const int nc8 = (o.num_ceps + 15) & ~15;
      dct_t.assign(static_cast<size_t>(o.mel.num_bins) * nc8, 0.0f);
      for (int c = 0; c < o.num_ceps; ++c)
        for (int m = 0; m < o.mel.num_bins; ++m) dct_t[static_cast<size_t>(m) * nc8 + c] = dct_h[c * o.mel.num_bins + m];
      if ((rc = plan->d_dct_t.upload(dct_t, plan->stream))) return rc;
      plan->fast512 = plan->mfcc_via_fbank = true;
      plan->h_window = window;
    }
  }
  if (want_fused && !plan->fast512)
    return set_error(SNF_E_INVALID, "append_deltas: this configuration is not covered by the register-resident "
                                    "512-point kernel (its tables do not fit); chain a delta plan");
  if (!plan->fast512 && fbank1024x2_eligible(p)) {
    std::vector<float> blob;
    fbank1024x2_tables(p, window, &blob);
    if ((rc = plan->d_long_tables.upload(blob, plan->stream))) return rc;
    plan->pair1024 = true;
  } else if (!plan->fast512 && fbank2048_eligible(p)) {
    std::vector<float> blob;
    fbank2048_tables(p, window, &blob);
    if ((rc = plan->d_long_tables.upload(blob, plan->stream))) return rc;
    plan->fast2048 = true;
  }
  return SNF_OK;
}

// (re)upload the per-warp mel tables after a new warp factor appeared
int sync_warp_tables(snf_plan* plan) {
  if (!plan->warps_dirty || plan->kind == SNF_KIND_SPECTROGRAM) return SNF_OK;
  const int nb = plan->o.mel.num_bins;
  std::vector<int> first, size, off;
  std::vector<float> w, eql;
  for (const MelBanksHost& mb : plan->banks) {
    const int base = static_cast<int>(w.size());
    for (int b = 0; b < nb; ++b) {
      first.push_back(mb.first[b]);
      size.push_back(mb.size[b]);
      off.push_back(base + mb.offset[b]);
    }
    w.insert(w.end(), mb.w.begin(), mb.w.end());
    if (plan->kind == SNF_KIND_PLP) {
      std::vector<float> e;
      make_equal_loudness(mb, &e);
      eql.insert(eql.end(), e.begin(), e.end());
    }
  }
  int rc;
  if ((rc = plan->d_mel_first.upload(first, plan->stream))) return rc;
  if ((rc = plan->d_mel_size.upload(size, plan->stream))) return rc;
  if ((rc = plan->d_mel_off.upload(off, plan->stream))) return rc;
  if ((rc = plan->d_mel_w.upload(w, plan->stream))) return rc;
  if (plan->fast2048 || plan->pair1024) {
    // the long-frame kernels read a filter in 32-tap slices of 16-byte vectors: a copy of the weights in
    // which every filter is zero-padded to whole slices, behind one all-zero slice (for the lanes whose
    // filter has fewer slices than the widest one of their round)
    std::vector<float> w32(32, 0.0f);
    std::vector<int> off32;
    // ... every filter starts at a multiple of 4 bins (leading zeros) and each group of 4 taps is rotated
    // by the bin index modulo 4 (= the team of 8 lanes that reads it: kernels_fbank2048.hip)
    for (const MelBanksHost& mb : plan->banks)
      for (int b = 0; b < nb; ++b) {
        off32.push_back(static_cast<int>(w32.size()));
        const int lead = mb.first[b] & 3, taps = lead + mb.size[b], rot = b & 3;
        const size_t base32 = w32.size();
        w32.resize(base32 + ((taps + 31) & ~31), 0.0f);
        for (int t = 0; t < ((taps + 3) & ~3); ++t) {
          const int src = (t & ~3) + (((t & 3) + rot) & 3);  // tap stored at position t of its group
          if (src >= lead && src < taps) w32[base32 + t] = mb.w[mb.offset[b] + src - lead];
        }
      }
    if ((rc = plan->d_mel_w32.upload(w32, plan->stream))) return rc;
    if ((rc = plan->d_mel_off32.upload(off32, plan->stream))) return rc;
    plan->mp.mel_w32 = plan->d_mel_w32.as<float>();
    plan->mp.mel_off32 = plan->d_mel_off32.as<int>();
  }
  plan->mp.mel_first = plan->d_mel_first.as<int>();
  plan->mp.mel_size = plan->d_mel_size.as<int>();
  plan->mp.mel_offset = plan->d_mel_off.as<int>();
  plan->mp.mel_w = plan->d_mel_w.as<float>();
  if (plan->kind == SNF_KIND_PLP) {
    if ((rc = plan->d_eql.upload(eql, plan->stream))) return rc;
    plan->pp.eql = plan->d_eql.as<float>();
  }
  // the uploads read from host vectors that die at scope exit
  SNF_HIP_CHECK(hipStreamSynchronize(plan->stream));
  plan->warps_dirty = false;
  return SNF_OK;
}

// fast-kernel tables of every warp factor seen so far (rebuilt when a new one appeared)
int sync_fast_warp_tables(snf_plan* plan) {
  if (!plan->fast512 || !plan->fast_warps_ok) return SNF_OK;
  if (plan->fast_warps_built == plan->banks.size()) return SNF_OK;
  std::vector<std::vector<float>> blobs(plan->banks.size());
  size_t stride = 0;
  Fast512Params fp0{};
  for (size_t w = 0; w < plan->banks.size(); ++w) {
    Fast512Params fp{};
    // (per-utterance tables: always the 512-point form, also for plans whose flat batches run dual)
    const int rc = fast512_build(plan->mp, plan->h_window, plan->banks[w], plan->h_dct, plan->h_lifter,
                                 false, &blobs[w], &fp);
    if (rc < 0) return rc;
    if (rc > 0) {  // this warp's banks need more taps per slot than the kernel unrolls
      plan->fast_warps_ok = false;
      return SNF_OK;
    }
    if (w == 0) fp0 = fp;
    stride = std::max(stride, blobs[w].size());
  }
  stride = (stride + 3) & ~static_cast<size_t>(3);
  std::vector<float> all(stride * blobs.size(), 0.0f);
  for (size_t w = 0; w < blobs.size(); ++w)
    std::copy(blobs[w].begin(), blobs[w].end(), all.begin() + w * stride);
  int rc;
  if ((rc = plan->d_fast_warp_tables.upload(all, plan->stream))) return rc;
  plan->fp_warp = fp0;
  plan->fp_warp.fused_delta = plan->fp.fused_delta;
  plan->fp_warp.delta_scales = plan->fp.delta_scales;
  plan->fp_warp.tables = plan->d_fast_warp_tables.as<float>();
  plan->fp_warp.table_stride = static_cast<int>(stride);
  plan->fast_warps_built = plan->banks.size();
  return SNF_OK;
}

// map per-utterance warp factors to table ids, creating tables on demand
int resolve_warps(snf_plan* plan, const float* vtln_warp, const int64_t* frame_offsets, int64_t n_utts,
                  std::vector<int32_t>* ids, bool* any) {
  *any = false;
  if (!vtln_warp || plan->kind == SNF_KIND_SPECTROGRAM) return SNF_OK;
  ids->assign(n_utts, 0);
  for (int64_t u = 0; u < n_utts; ++u) {
    // Kaldi builds the banks of a warp factor when the first frame asks for them
    // ([KALDI-UPSTREAM] MfccComputer::GetMelBanks): an utterance without frames never does, and
    // never sees the option errors of its warp factor
    if (frame_offsets[u + 1] == frame_offsets[u]) continue;
    const float wf = vtln_warp[u];
    int id = -1;
    for (size_t k = 0; k < plan->warps.size(); ++k)
      if (plan->warps[k] == wf) {
        id = static_cast<int>(k);
        break;
      }
    if (id < 0) {
      MelBanksHost mb;
      int rc = make_mel_banks(plan->o.mel, plan->o.frame, wf, &mb);
      if (rc) return rc;
      plan->warps.push_back(wf);
      plan->banks.push_back(mb);
      plan->warps_dirty = true;
      id = static_cast<int>(plan->warps.size()) - 1;
    }
    (*ids)[u] = id;
    if (id != 0) *any = true;
  }
  return SNF_OK;
}

int build_delta_plan(snf_plan* plan) {
  const snf_options& o = plan->o;
  if (o.delta_order < 0 || o.delta_order >= 1000)
    return set_error(SNF_E_RUNTIME, "delta order must be in [0, 999]");
  if (o.delta_window <= 0 || o.delta_window >= 1000)
    return set_error(SNF_E_INVALID, "window must be in [1, 999]");
  std::vector<float> scales;
  std::vector<int> dims;
  make_delta_scales(o.delta_order, o.delta_window, &scales, &dims);
  int rc;
  if ((rc = plan->d_scales.upload(scales, plan->stream))) return rc;
  if ((rc = plan->d_dims.upload(dims, plan->stream))) return rc;
  plan->dp.order = o.delta_order;
  plan->dp.window = o.delta_window;
  plan->dp.n_scales = static_cast<int>(scales.size());
  plan->dp.scales = plan->d_scales.as<float>();
  plan->dp.dims = plan->d_dims.as<int>();
  plan->ndims = -1;
  return SNF_OK;
}

int build_pitch_plan(snf_plan* plan) {
  const snf_pitch_options& o = plan->o.pitch;
  int rc = make_pitch_tables(o, &plan->pt);
  if (rc) return rc;
  const PitchTablesHost& t = plan->pt;
  if ((rc = plan->d_lags.upload(t.lags, plan->stream))) return rc;
  if ((rc = plan->d_ar_first.upload(t.ar_first, plan->stream))) return rc;
  if ((rc = plan->d_ar_n.upload(t.ar_n, plan->stream))) return rc;
  if ((rc = plan->d_ar_w.upload(t.ar_w, plan->stream))) return rc;
  if ((rc = plan->d_rs_first.upload(t.resample.first, plan->stream))) return rc;
  if ((rc = plan->d_rs_ntaps.upload(t.resample.ntaps, plan->stream))) return rc;
  if ((rc = plan->d_rs_w.upload(t.resample.weights, plan->stream))) return rc;
  PitchDevTables& d = plan->pd;
  {
    // ArbitraryResample as 4 x 4 outer-product steps (see PitchDevTables): quad windows and weights
    const int S = t.num_states, groups = ((S + 63) / 64 + 1) & ~1;
    std::vector<int> qbase(static_cast<size_t>(groups) * 16, 0);
    int kmax = 1;
    for (int qd = 0; qd < groups * 16; ++qd) {
      const int s0 = qd * 4;
      if (s0 >= S) continue;
      int lo = t.ar_first[s0], hi = lo;
      for (int s = s0; s < s0 + 4 && s < S; ++s) {
        lo = std::min(lo, t.ar_first[s]);
        hi = std::max(hi, t.ar_first[s] + std::max(t.ar_n[s], 0));
      }
      qbase[qd] = lo;
      kmax = std::max(kmax, hi - lo);
    }
    const int taps = (kmax + 3) & ~3;
    std::vector<float> qw(static_cast<size_t>(groups) * taps * 64, 0.0f);
    for (int g = 0; g < groups; ++g)
      for (int k = 0; k < taps; ++k)
        for (int ln = 0; ln < 64; ++ln) {
          const int s = 64 * g + ln;
          if (s >= S) continue;
          const int j = qbase[g * 16 + ln / 4] + k - t.ar_first[s];
          if (j >= 0 && j < t.ar_n[s])
            qw[((static_cast<size_t>(g) * (taps / 4) + k / 4) * 64 + ln) * 4 + (k & 3)] =
                t.ar_w[static_cast<size_t>(s) * t.max_taps + j];
        }
    if ((rc = plan->d_ar_quad_w.upload(qw, plan->stream))) return rc;
    if ((rc = plan->d_ar_quad_base.upload(qbase, plan->stream))) return rc;
    d.ar_groups = groups;
    d.ar_quad_taps = taps;
    d.ar_quad_w = plan->d_ar_quad_w.as<float>();
    d.ar_quad_base = plan->d_ar_quad_base.as<int>();
  }
  d.first_lag = t.first_lag;
  d.last_lag = t.last_lag;
  d.num_lags = t.num_lags;
  d.num_states = t.num_states;
  d.win_size = t.win_size;
  d.win_shift = t.win_shift;
  d.full_len = t.full_len;
  d.ar_max_taps = t.max_taps;
  d.rs_in_unit = t.resample.in_unit;
  d.rs_out_unit = t.resample.out_unit;
  d.rs_max_taps = t.resample.max_taps;
  d.snip_edges = o.snip_edges;
  d.recompute_frame = o.recompute_frame;
  d.soft_min_f0 = o.soft_min_f0;
  const float delta_pitch_sq =
      static_cast<float>(std::pow(static_cast<double>(logf(static_cast<float>(1.0 + o.delta_pitch))), 2.0));
  d.inter_frame_factor = delta_pitch_sq * o.penalty_factor;
  d.nccf_ballast = o.nccf_ballast;
  d.lags = plan->d_lags.as<float>();
  d.ar_first = plan->d_ar_first.as<int>();
  d.ar_n = plan->d_ar_n.as<int>();
  d.ar_w = plan->d_ar_w.as<float>();
  d.rs_first = plan->d_rs_first.as<int>();
  d.rs_ntaps = plan->d_rs_ntaps.as<int>();
  d.rs_w = plan->d_rs_w.as<float>();
  plan->ndims = 2;
  return SNF_OK;
}

int64_t pitch_frames_for(const snf_plan* plan, int64_t n, int64_t* n_down, int64_t* n_down_p1,
                         int64_t* frames_p1) {
  const PitchTablesHost& t = plan->pt;
  const int64_t nd = t.resample.num_output(n, true), nd1 = t.resample.num_output(n, false);
  const bool snip = plan->o.pitch.snip_edges != 0;
  const int64_t T = t.frames_available(nd, true, snip);
  int64_t T1 = t.frames_available(nd1, false, snip);
  if (T1 > T) T1 = T;
  if (n_down) *n_down = nd;
  if (n_down_p1) *n_down_p1 = nd1;
  if (frames_p1) *frames_p1 = T1;
  return T;
}

void begin_timing(snf_plan* plan) {
  plan->n_slots = 0;
  plan->events_valid = false;
  (void)hipEventRecord(plan->ev[0], plan->stream);
}
void mark_kernel(snf_plan* plan, const char* name) {
  if (plan->n_slots >= kMaxSlots) return;
  ++plan->n_slots;
  plan->slot_name[plan->n_slots] = name;
  (void)hipEventRecord(plan->ev[plan->n_slots], plan->stream);
  plan->events_valid = true;
}

int check_offsets(const snf_plan* plan, const int64_t* sample_offsets, const int64_t* frame_offsets,
                  int64_t n_utts) {
  if (n_utts < 0) return set_error(SNF_E_INVALID, "n_utts < 0");
  if (n_utts == 0) return SNF_OK;
  if (!sample_offsets || !frame_offsets) return set_error(SNF_E_INVALID, "null offsets table");
  // (rows below offsets[0] would resolve to utterance 0 with a negative local frame)
  if (sample_offsets[0] != 0 || frame_offsets[0] != 0)
    return set_error(SNF_E_INVALID, "offsets tables must start at 0");
  for (int64_t u = 0; u < n_utts; ++u) {
    const int64_t n = sample_offsets[u + 1] - sample_offsets[u];
    const int64_t f = frame_offsets[u + 1] - frame_offsets[u];
    if (n < 0 || f < 0) return set_error(SNF_E_INVALID, "offsets tables must be non-decreasing");
    if (f != snf_plan_num_frames(plan, n))
      return set_error(SNF_E_INVALID, "frame_offsets do not match snf_plan_num_frames for utterance " +
                                          std::to_string(u));
  }
  return SNF_OK;
}

int run_pitch_device(snf_plan* plan, const int16_t* d_wave, const int64_t* sample_offsets,
                     int64_t n_utts, float* d_out, const int64_t* frame_offsets, hipStream_t s) {
  const int64_t total_frames = frame_offsets[n_utts];
  if (total_frames == 0) return SNF_OK;
  std::vector<int64_t> doff(n_utts + 1, 0), dp1(n_utts, 0), fp1(n_utts, 0);
  for (int64_t u = 0; u < n_utts; ++u) {
    int64_t nd, nd1, t1;
    pitch_frames_for(plan, sample_offsets[u + 1] - sample_offsets[u], &nd, &nd1, &t1);
    doff[u + 1] = doff[u] + nd;
    dp1[u] = nd1;
    fp1[u] = t1;
  }
  const int64_t total_down = doff[n_utts];
  int64_t max_down = 0;
  for (int64_t u = 0; u < n_utts; ++u) max_down = std::max(max_down, doff[u + 1] - doff[u]);
  int rc;
  std::vector<int64_t> soff(sample_offsets, sample_offsets + n_utts + 1);
  std::vector<int64_t> foff(frame_offsets, frame_offsets + n_utts + 1);
  if ((rc = plan->s_soff.upload(soff, s))) return rc;
  if ((rc = plan->s_foff.upload(foff, s))) return rc;
  if ((rc = plan->s_doff.upload(doff, s))) return rc;
  if ((rc = plan->s_dp1.upload(dp1, s))) return rc;
  if ((rc = plan->s_fp1.upload(fp1, s))) return rc;
  // ragged batches: the tracker walks one utterance per wavefront, so a workgroup lasts as long as
  // its longest utterance - hand the utterances out longest first (ties keep the batch order)
  std::vector<int32_t> order;
  bool ragged = false;
  for (int64_t u = 1; u < n_utts && !ragged; ++u)
    ragged = (foff[u + 1] - foff[u]) != (foff[1] - foff[0]);
  if (ragged && n_utts < (int64_t{1} << 31)) {
    order.resize(static_cast<size_t>(n_utts));
    for (int64_t u = 0; u < n_utts; ++u) order[u] = static_cast<int32_t>(u);
    std::stable_sort(order.begin(), order.end(), [&](int32_t x, int32_t y) {
      return foff[x + 1] - foff[x] > foff[y + 1] - foff[y];
    });
    if ((rc = plan->s_uwarp.upload(order, s))) return rc;
  }
  const size_t nf = static_cast<size_t>(total_frames);
  if ((rc = plan->s_down.ensure(sizeof(float) * static_cast<size_t>(total_down > 0 ? total_down : 1)))) return rc;
  if ((rc = plan->s_stats.ensure(sizeof(float) * 6 * static_cast<size_t>(n_utts)))) return rc;
  if ((rc = plan->s_bp.ensure(sizeof(int16_t) * nf * plan->pd.num_states))) return rc;
  if ((rc = plan->s_states.ensure(sizeof(int32_t) * nf))) return rc;
  if ((rc = plan->s_mel.ensure(sizeof(float) * nf * plan->pd.num_lags))) return rc;
  if ((rc = plan->s_pres.ensure(sizeof(float) * nf * plan->pd.num_states))) return rc;
  if ((rc = plan->s_anp.ensure(sizeof(float) * nf))) return rc;
  if ((rc = plan->s_futt.ensure(sizeof(int4) * nf))) return rc;
  SNF_HIP_CHECK(hipStreamSynchronize(s));  // host vectors above go out of scope after launch setup
  PitchBatch b{};
  b.wave = d_wave;
  b.sample_offsets = plan->s_soff.as<int64_t>();
  b.frame_offsets = plan->s_foff.as<int64_t>();
  b.down_offsets = plan->s_doff.as<int64_t>();
  b.down_phase1 = plan->s_dp1.as<int64_t>();
  b.order = order.empty() ? nullptr : plan->s_uwarp.as<int32_t>();
  b.frames_phase1 = plan->s_fp1.as<int64_t>();
  b.n_utts = n_utts;
  b.total_frames = total_frames;
  b.total_down = total_down;
  b.max_down = max_down;
  PitchScratch w{};
  w.down = plan->s_down.as<float>();
  w.ub = plan->s_stats.as<float>();
  w.nccf_res = plan->s_pres.as<float>();
  w.pov_nccf = plan->s_mel.as<float>();
  w.anp = plan->s_anp.as<float>();
  w.backptr = plan->s_bp.as<int16_t>();
  w.states = plan->s_states.as<int32_t>();
  w.frame_meta = plan->s_futt.as<int4>();
  return launch_pitch(plan->pd, b, w, d_out, s);
}

}  // namespace

// =================================================================================================
extern "C" {

const char* snf_version(void) { return "shennong_amd 0.1 (gfx950)"; }
const char* snf_last_error(void) { return last_error(); }

namespace {
// noise stream of the next call of THIS thread that draws random numbers (0: the plan's own call count)
thread_local uint64_t t_noise_call = 0;
// every run entry point takes it first thing: a name given to a call that draws nothing is not left behind
// for the thread's next call
uint64_t take_noise_call() {
  const uint64_t pinned = t_noise_call;
  t_noise_call = 0;
  return pinned;
}
}  // namespace

int snf_set_noise_call(uint64_t call) {
  t_noise_call = call;
  return SNF_OK;
}

int snf_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}
int snf_set_device(int device_id) {
  SNF_HIP_CHECK(hipSetDevice(device_id));
  return SNF_OK;
}
int snf_device_name(int device_id, char* buf, int buflen) {
  hipDeviceProp_t prop;
  SNF_HIP_CHECK(hipGetDeviceProperties(&prop, device_id));
  snprintf(buf, buflen, "%s (%s)", prop.name, prop.gcnArchName);
  return SNF_OK;
}
int snf_device_synchronize(void) {
  SNF_HIP_CHECK(hipDeviceSynchronize());
  return SNF_OK;
}

int64_t snf_num_frames(const snf_frame_options* o, int64_t n) { return num_frames(*o, n); }
int64_t snf_first_sample_of_frame(const snf_frame_options* o, int64_t f) {
  return first_sample_of_frame(*o, f);
}
int32_t snf_window_size(const snf_frame_options* o) { return window_size(*o); }
int32_t snf_window_shift(const snf_frame_options* o) { return window_shift(*o); }
int32_t snf_padded_window_size(const snf_frame_options* o) { return padded_window_size(*o); }
int snf_window_function(const snf_frame_options* o, float* out) {
  std::vector<float> w;
  int rc = make_window(*o, &w);
  if (rc) return rc;
  std::memcpy(out, w.data(), sizeof(float) * w.size());
  return SNF_OK;
}
int64_t snf_pitch_num_frames(const snf_pitch_options* o, int64_t n) {
  PitchTablesHost t;
  if (make_pitch_tables(*o, &t)) return -1;
  return t.frames_available(t.resample.num_output(n, true), true, o->snip_edges != 0);
}

int snf_plan_create(const snf_options* opts, int device_id, snf_plan** out) {
  if (!opts || !out) return set_error(SNF_E_INVALID, "null argument");
  *out = nullptr;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
    return set_error(SNF_E_NODEVICE, "no HIP device visible: libshennong_hip needs an MI355X (gfx950)");
  if (device_id < 0 || device_id >= ndev) return set_error(SNF_E_INVALID, "bad device id");
  SNF_HIP_CHECK(hipSetDevice(device_id));
  std::unique_ptr<snf_plan> plan(new snf_plan);
  plan->o = *opts;
  plan->device = device_id;
  plan->kind = opts->kind;
  SNF_HIP_CHECK(hipStreamCreateWithFlags(&plan->stream, hipStreamNonBlocking));
  for (auto& e : plan->ev) SNF_HIP_CHECK(hipEventCreate(&e));
  int rc;
  switch (opts->kind) {
    case SNF_KIND_SPECTROGRAM:
    case SNF_KIND_FBANK:
    case SNF_KIND_MFCC:
    case SNF_KIND_PLP:
    case SNF_KIND_ENERGY:
      rc = build_mel_plan(plan.get());
      break;
    case SNF_KIND_VAD: {
      const snf_vad_options& v = opts->vad;
      plan->ndims = 1;
      rc = SNF_OK;
      if (v.frames_context < 0)
        rc = set_error(SNF_E_RUNTIME, "vad-frames-context must be >= 0");
      else if (!(v.proportion_threshold > 0.0f && v.proportion_threshold < 1.0f))
        rc = set_error(SNF_E_RUNTIME, "vad-proportion-threshold must be in (0, 1)");
      break;
    }
    case SNF_KIND_SLIDING_CMVN: {
      const snf_sliding_cmvn_options& c = opts->sliding_cmvn;
      rc = SNF_OK;
      if (c.cmn_window <= 0) rc = set_error(SNF_E_RUNTIME, "cmn_window must be positive");
      else if (c.min_window <= 0) rc = set_error(SNF_E_RUNTIME, "min_window must be positive");
      break;
    }
    case SNF_KIND_CMVN:
      rc = SNF_OK;
      break;
    case SNF_KIND_DELTA:
      rc = build_delta_plan(plan.get());
      break;
    case SNF_KIND_PITCH:
      rc = build_pitch_plan(plan.get());
      break;
    case SNF_KIND_PITCH_POST: {
      const snf_pitch_post_options& q = opts->pitch_post;
      plan->ppost.o = q;
      plan->ppost.seed = opts->seed;
      plan->ppost.ndims = (q.add_pov_feature ? 1 : 0) + (q.add_normalized_log_pitch ? 1 : 0) +
                          (q.add_delta_pitch ? 1 : 0) + (q.add_raw_log_pitch ? 1 : 0);
      plan->ndims = plan->ppost.ndims;
      rc = SNF_OK;
      if (plan->ndims <= 0)
        rc = set_error(SNF_E_INVALID, "at least one of the pitch post-processing features must be selected");
      else if (q.delay != 0)
        rc = set_error(SNF_E_RUNTIME, "pitch post-processing delay != 0 is not supported");
      else if (q.delta_window <= 0 || q.delta_window >= 1000)
        rc = set_error(SNF_E_RUNTIME, "delta_window must be in [1, 999]");
      break;
    }
    default:
      rc = set_error(SNF_E_INVALID, "unknown or unsupported plan kind");
  }
  if (rc) return rc;
  SNF_HIP_CHECK(hipStreamSynchronize(plan->stream));
  *out = plan.release();
  return SNF_OK;
}

void snf_plan_destroy(snf_plan* plan) {
  if (!plan) return;
  (void)hipSetDevice(plan->device);
  (void)hipStreamSynchronize(plan->stream);
  delete plan;
}

int32_t snf_plan_ndims(const snf_plan* plan) { return plan ? plan->ndims : -1; }
int32_t snf_plan_fast_path(const snf_plan* plan) {
  if (!plan) return -1;
  switch (plan->kind) {
    case SNF_KIND_SPECTROGRAM:
    case SNF_KIND_FBANK:
    case SNF_KIND_MFCC:
    case SNF_KIND_PLP:
    case SNF_KIND_ENERGY:
      return (plan->fast512 || plan->fast2048 || plan->pair1024) ? 1 : 0;
    default:
      return 1;  // (no slower alternative exists for this kind)
  }
}

int64_t snf_plan_num_frames(const snf_plan* plan, int64_t n) {
  if (!plan) return -1;
  switch (plan->kind) {
    case SNF_KIND_SPECTROGRAM:
    case SNF_KIND_FBANK:
    case SNF_KIND_MFCC:
    case SNF_KIND_PLP:
    case SNF_KIND_ENERGY:
      return num_frames(plan->o.frame, n);
    case SNF_KIND_PITCH:
      return pitch_frames_for(plan, n, nullptr, nullptr, nullptr);
    default:
      return -1;
  }
}

int snf_plan_run_batch_device(snf_plan* plan, const int16_t* d_wave, const int64_t* sample_offsets,
                              int64_t n_utts, const float* vtln_warp, float* d_out,
                              const int64_t* frame_offsets, void* stream) {
  const uint64_t named_call = take_noise_call();
  if (!plan) return set_error(SNF_E_INVALID, "null plan");
  std::lock_guard<std::mutex> lock(plan->mu);
  int rc = guard_device(plan);
  if (rc) return rc;
  if (n_utts < 0) return set_error(SNF_E_INVALID, "n_utts < 0");
  if (n_utts == 0) return SNF_OK;
  if (!sample_offsets || !frame_offsets) return set_error(SNF_E_INVALID, "null offsets table");
  const bool same_tables =
      plan->h_soff.size() == static_cast<size_t>(n_utts + 1) &&
      std::memcmp(plan->h_soff.data(), sample_offsets, sizeof(int64_t) * (n_utts + 1)) == 0 &&
      std::memcmp(plan->h_foff.data(), frame_offsets, sizeof(int64_t) * (n_utts + 1)) == 0;
  if (!same_tables) {
    plan->h_soff.clear();
    if ((rc = check_offsets(plan, sample_offsets, frame_offsets, n_utts))) return rc;
  }
  const int64_t total_frames = frame_offsets[n_utts];
  if (total_frames == 0) return SNF_OK;
  hipStream_t s = stream ? static_cast<hipStream_t>(stream) : plan->stream;
  const bool own_stream = (stream == nullptr);

  if (plan->kind == SNF_KIND_PITCH) {
    plan->h_soff.clear();
    if (own_stream) begin_timing(plan);
    rc = run_pitch_device(plan, d_wave, sample_offsets, n_utts, d_out, frame_offsets, s);
    if (rc) return rc;
    if (own_stream) {
      mark_kernel(plan, "pitch");
      SNF_HIP_CHECK(hipStreamSynchronize(s));
    }
    return SNF_OK;
  }
  if (plan->kind != SNF_KIND_SPECTROGRAM && plan->kind != SNF_KIND_FBANK &&
      plan->kind != SNF_KIND_MFCC && plan->kind != SNF_KIND_PLP && plan->kind != SNF_KIND_ENERGY)
    return set_error(SNF_E_INVALID, "plan kind does not take audio input");

  std::vector<int32_t> warp_ids;
  bool any_warp = false;
  if ((rc = resolve_warps(plan, vtln_warp, frame_offsets, n_utts, &warp_ids, &any_warp))) return rc;
  if (!plan->base_banks_error.empty()) {
    // (PLP, see snf_plan_create) an utterance with frames that needs the unwarped banks
    for (int64_t u = 0; u < n_utts; ++u)
      if (frame_offsets[u + 1] > frame_offsets[u] && (warp_ids.empty() || warp_ids[u] == 0))
        return set_error(SNF_E_RUNTIME, plan->base_banks_error);
  }
  if ((rc = sync_warp_tables(plan))) return rc;

  if (!same_tables) {
    std::vector<int64_t> soff(sample_offsets, sample_offsets + n_utts + 1);
    std::vector<int64_t> foff(frame_offsets, frame_offsets + n_utts + 1);
    if ((rc = plan->s_soff.upload(soff, s))) return rc;
    if ((rc = plan->s_foff.upload(foff, s))) return rc;
    plan->h_soff.swap(soff);
    plan->h_foff.swap(foff);
    plan->setidx_valid = false;
    plan->pairs_valid = false;
    plan->tile_valid = false;
  }
  if (any_warp && (rc = plan->s_uwarp.upload(warp_ids, s))) return rc;
  BatchArgs b{};
  b.wave = d_wave;
  b.sample_offsets = plan->s_soff.as<int64_t>();
  b.frame_offsets = plan->s_foff.as<int64_t>();
  b.utt_warp = any_warp ? plan->s_uwarp.as<int32_t>() : nullptr;
  b.n_utts = n_utts;
  b.total_frames = total_frames;
  if (plan->mp.dither != 0.0f) {
    // what the dither streams know about an utterance: a hash of 64 of its samples (wave_noise_id)
    if ((rc = plan->s_unoise.ensure(sizeof(uint32_t) * static_cast<size_t>(n_utts > 0 ? n_utts : 1)))) return rc;
    if ((rc = launch_build_utt_noise(b, plan->s_unoise.as<uint32_t>(), s))) return rc;
    b.utt_noise = plan->s_unoise.as<uint32_t>();
  }

  if (plan->kind == SNF_KIND_PLP) {
    const int nb = plan->o.mel.num_bins;
    if ((rc = plan->s_mel.ensure(sizeof(float) * static_cast<size_t>(total_frames) * nb))) return rc;
    if ((rc = plan->s_energy.ensure(sizeof(double) * static_cast<size_t>(total_frames)))) return rc;
  }
  bool use_fast = plan->fast512;
  bool use_long = plan->fast2048 || plan->pair1024;  // (per-utterance VTLN warps included: they read the plan's bank tables)
  if (use_fast && any_warp && (plan->wide || plan->mfcc_via_fbank))
    use_fast = false;   // (VTLN batches of a wide bank / of a filterbank-first MFCC plan: the generic kernel)
  if (use_fast && any_warp) {
    if ((rc = sync_fast_warp_tables(plan))) return rc;
    use_fast = plan->fast_warps_ok;
  }
  // snip_edges = false: the clamped bulk loads of the centred frames need an utterance that holds one full
  // window.  A shorter utterance ALWAYS runs on the generic kernel, whatever else is in the batch (the
  // register-resident kernels leave it out, or compute its frames from some window inside the batch and
  // have them overwritten by the masked generic launch below); a batch shorter than one window holds
  // nothing but such utterances.
  std::vector<uint8_t> short_mask;
  bool any_short = false;
  if ((use_fast || use_long) && !plan->mp.snip_edges) {
    if (sample_offsets[n_utts] < plan->mp.win_len) {
      use_fast = use_long = false;
    } else {
      short_mask.assign(static_cast<size_t>(n_utts), 0);
      for (int64_t u = 0; u < n_utts; ++u) {
        const int64_t n = sample_offsets[u + 1] - sample_offsets[u];
        if (n > 0 && n < plan->mp.win_len && frame_offsets[u + 1] > frame_offsets[u]) {
          short_mask[u] = 1;
          any_short = true;
        }
      }
    }
  }
  const bool fused = plan->fp.fused_delta != 0;
  if (fused && (!use_fast || any_short))
    return set_error(SNF_E_RUNTIME, "append_deltas: this batch cannot run on the 512-point path");
  // Two-frames-per-row plans (fbank256x2_kernel) in a batch with VTLN warps: the kernel an utterance runs
  // on must not depend on its neighbours (the two forms round differently in the last bits), so the
  // unwarped utterances keep the two-frame kernel and only the warped ones take the zero-extended
  // 512-point form with their per-warp tables - two launches over disjoint sets of utterances.
  const bool split_dual = use_fast && plan->fp.dual && any_warp && !fused;
  if (use_fast && (any_warp || fused)) {
    // workgroup -> (utterance, first frame set) list: every workgroup stages the tables of one warp
    // (fused deltas: a workgroup owns a run of frames of one utterance + their delta halo)
    const int kSetsPerBlock = fused ? kFast512FusedSets : 64;  // (kernels_fbank512.hip)
    std::vector<int32_t> blk_utt, blk_set0;
    for (int64_t u = 0; u < n_utts; ++u) {
      if (split_dual && warp_ids[u] == 0) continue;  // (runs on fbank256x2_kernel: below)
      if (any_short && short_mask[u]) continue;       // (runs on the generic kernel: below)
      const int64_t sets = (frame_offsets[u + 1] - frame_offsets[u] + 3) / 4;
      for (int64_t s0 = 0; s0 < sets; s0 += kSetsPerBlock) {
        blk_utt.push_back(static_cast<int32_t>(u));
        blk_set0.push_back(static_cast<int32_t>(s0));
      }
    }
    if ((rc = plan->s_blk_utt.upload(blk_utt, s))) return rc;
    if ((rc = plan->s_blk_set0.upload(blk_set0, s))) return rc;
    b.blk_utt = plan->s_blk_utt.as<int32_t>();
    b.blk_set0 = plan->s_blk_set0.as<int32_t>();
    b.n_blocks = static_cast<int64_t>(blk_utt.size());
  }
  BatchArgs b_dual = b;  // the arguments of the two-frame kernel (all utterances, or the unwarped ones)
  const bool use_pair = use_long && plan->pair1024;   // (every utterance, warped or not: fbank1024x2_kernel)
  if ((use_fast && plan->fp.dual && (!any_warp || split_dual)) || use_pair) {
    // fbank256x2_kernel: frame pairs formed inside every utterance (PairRec), built once per offsets table
    // (a batch split by warp factor: pairs of the unwarped utterances only, rebuilt on every call)
    if (split_dual || any_short) plan->pairs_valid = false;
    if (!plan->pairs_valid) {
      std::vector<int64_t> poff(static_cast<size_t>(n_utts) + 1, 0);
      for (int64_t u = 0; u < n_utts; ++u)
        poff[u + 1] = poff[u] + (((split_dual && warp_ids[u] != 0) || (any_short && short_mask[u]))
                                     ? 0 : (frame_offsets[u + 1] - frame_offsets[u] + 1) / 2);
      plan->n_pairs = poff[n_utts];
      if ((rc = plan->s_poff.upload(poff, s))) return rc;
      if ((rc = plan->s_pairs.ensure(sizeof(PairRec) * static_cast<size_t>(plan->n_pairs)))) return rc;
      if ((rc = launch_build_pair_table(plan->s_foff.as<int64_t>(), plan->s_soff.as<int64_t>(),
                                        plan->s_poff.as<int64_t>(), n_utts, plan->n_pairs,
                                        plan->mp.win_shift, plan->mp.win_len, plan->mp.snip_edges,
                                        plan->s_pairs.as<PairRec>(), s)))
        return rc;
      if (!own_stream) SNF_HIP_CHECK(hipStreamSynchronize(s));  // (cached: see below)
      plan->pairs_valid = !split_dual && !any_short;
    }
    b_dual.pair_tab = plan->s_pairs.as<PairRec>();
    b_dual.n_pairs = plan->n_pairs;
    if (!use_pair && plan->n_pairs > 0 && !getenv("SNF_DUAL_NO_FIXUP")) {
      // fbank256x2_kernel: pairs of very different energies are redone one frame at a time by a second launch
      // (BatchArgs::fix_tab): room for every pair, the count zeroed in stream order
      if ((rc = plan->s_fix.ensure(sizeof(PairRec) * 2 * static_cast<size_t>(plan->n_pairs)))) return rc;
      if ((rc = plan->s_fixcount.ensure(sizeof(unsigned int)))) return rc;
      SNF_HIP_CHECK(hipMemsetAsync(plan->s_fixcount.p, 0, sizeof(unsigned int), s));
      b_dual.fix_tab = plan->s_fix.as<PairRec>();
      b_dual.fix_count = plan->s_fixcount.as<unsigned int>();
      static const float ratio = [] {
        const char* e = getenv("SNF_PAIR_SPLIT_RATIO");   // (experiments: 0 redoes every pair)
        return e ? static_cast<float>(atof(e)) : kPairSplitRatio;
      }();
      b_dual.split_ratio = ratio;
    }
    b_dual.blk_utt = nullptr;
    b_dual.blk_set0 = nullptr;
    b_dual.n_blocks = 0;
    if (!use_pair) b_dual.utt_warp = nullptr;
  } else if (!(use_fast && (any_warp || fused)) && !plan->setidx_valid) {
    // frame -> first-sample index, edge marks and utterance index: built once per offsets table,
    // reused by later calls (fast kernel: bulk loads; generic kernel: no per-frame binary search)
    if ((rc = plan->s_setidx.ensure(sizeof(int64_t) * static_cast<size_t>(total_frames)))) return rc;
    if ((rc = plan->s_edge.ensure(sizeof(int32_t) * static_cast<size_t>(total_frames)))) return rc;
    if ((rc = plan->s_futt.ensure(sizeof(int32_t) * static_cast<size_t>(total_frames)))) return rc;
    if ((rc = launch_build_frame_start(plan->s_foff.as<int64_t>(), plan->s_soff.as<int64_t>(), n_utts,
                                       total_frames, sample_offsets[n_utts], plan->mp.win_shift, plan->mp.win_len,
                                       plan->mp.snip_edges, plan->s_setidx.as<int64_t>(),
                                       plan->s_edge.as<int32_t>(), plan->s_futt.as<int32_t>(), s)))
      return rc;
    // (the tables are cached: a later call may come in on another stream, so they must be complete)
    if (!own_stream) SNF_HIP_CHECK(hipStreamSynchronize(s));
    plan->setidx_valid = true;
  }
  b.frame_start = plan->s_setidx.as<int64_t>();
  b.frame_edge = plan->s_edge.as<int32_t>();
  b.frame_utt = plan->setidx_valid ? plan->s_futt.as<int32_t>() : nullptr;
  if (own_stream) begin_timing(plan);
  if (plan->mp.dither != 0.0f) {
    const unsigned long long stream_key = plan->o.seed + 0x9E3779B97F4A7C15ull * (named_call ? named_call : ++plan->noise_calls);
    plan->mp.seed = plan->fp.seed = plan->fp_hi.seed = plan->fp_warp.seed = stream_key;
    // fbank512b_kernel reads the noise key of a frame from a table (the keys hold the utterance's noise
    // word: made with every batch; 8 bytes per frame)
    b.frame_noise = nullptr;
    if (use_fast && !any_warp && !fused && !plan->fp.dual && b.frame_utt != nullptr && plan->mp.snip_edges &&
        !getenv("SNF_FBANK512_OLD")) {
      if ((rc = plan->s_noise.ensure(sizeof(uint64_t) * static_cast<size_t>(total_frames)))) return rc;
      if ((rc = launch_build_frame_noise(b, plan->s_noise.as<uint64_t>(), s))) return rc;
      b.frame_noise = plan->s_noise.as<uint64_t>();
    }
  }
  // the register-resident 512-point family: one launch, or two over disjoint utterances (split_dual)
  auto run_fast = [&](float* out, int cols, double* energy) -> int {
    int rc2;
    if (plan->fp.dual && (!any_warp || split_dual)) {
      if (b_dual.n_pairs > 0) {
        if ((rc2 = launch_fbank512(plan->fp, b_dual, out, cols, energy, s))) return rc2;
        if (own_stream) mark_kernel(plan, "fbank256x2_kernel");
      }
      if (!split_dual) return SNF_OK;
    }
    if (b.blk_utt != nullptr && b.n_blocks == 0) return SNF_OK;  // (no utterance left for this form)
    const Fast512Params& fpx = any_warp ? plan->fp_warp : plan->fp;
    if ((rc2 = launch_fbank512(fpx, b, out, cols, energy, s))) return rc2;
    if (own_stream) mark_kernel(plan, fbank512b_eligible(fpx, b) ? "fbank512b_kernel" : "fbank512_kernel");
    if (plan->wide) {   // the upper half of a wide bank, behind the columns of the lower one
      if ((rc2 = launch_fbank512(plan->fp_hi, b, out + plan->wide_offset, cols, energy, s))) return rc2;
      if (own_stream) mark_kernel(plan, fbank512b_eligible(plan->fp_hi, b) ? "fbank512b_kernel" : "fbank512_kernel");
    }
    return SNF_OK;
  };
  // the long-frame family: one frame per wave (2048-sample frames), or a pair of frames (1024-sample frames)
  auto run_long = [&](float* out, int cols, double* energy) -> int {
    int rc2;
    if (use_pair) {
      BatchArgs bp = b_dual;
      bp.utt_noise = b.utt_noise;
      if ((rc2 = launch_fbank1024x2(plan->mp, bp, plan->d_long_tables.as<float>(), out, cols, energy, s))) return rc2;
      if (own_stream) mark_kernel(plan, "fbank1024x2_kernel");
      return SNF_OK;
    }
    if ((rc2 = launch_fbank2048(plan->mp, b, plan->d_long_tables.as<float>(), out, cols, energy, s))) return rc2;
    if (own_stream) mark_kernel(plan, "fbank2048_kernel");
    return SNF_OK;
  };
  // utterances shorter than a window (snip_edges = false), after the register-resident kernels
  auto run_short = [&](float* out, int cols, double* energy) -> int {
    if (!any_short || !(use_fast || use_long)) return SNF_OK;
    int rc2;
    if ((rc2 = plan->s_umask.upload(short_mask, s))) return rc2;
    BatchArgs bm = b;
    bm.utt_mask = plan->s_umask.as<uint8_t>();
    if ((rc2 = launch_mel_features(plan->mp, bm, out, cols, energy, s))) return rc2;
    if (own_stream) mark_kernel(plan, "mel_features_generic_kernel");
    return SNF_OK;
  };
  if (plan->kind == SNF_KIND_PLP) {
    const int nb = plan->o.mel.num_bins;
    if (use_fast) {
      if ((rc = run_fast(plan->s_mel.as<float>(), nb, plan->s_energy.as<double>()))) return rc;
    } else if (use_long) {
      if ((rc = run_long(plan->s_mel.as<float>(), nb, plan->s_energy.as<double>()))) return rc;
    } else {
      if ((rc = launch_mel_features(plan->mp, b, plan->s_mel.as<float>(), nb,
                                    plan->s_energy.as<double>(), s)))
        return rc;
      if (own_stream) mark_kernel(plan, "mel_features_generic_kernel");
    }
    if ((rc = run_short(plan->s_mel.as<float>(), nb, plan->s_energy.as<double>()))) return rc;
    if (plan->o.rasta) {
      if ((rc = launch_rasta(plan->s_mel.as<float>(), b, nb, s))) return rc;
      if (own_stream) mark_kernel(plan, "rasta_kernel");
    }
    if ((rc = launch_plp_tail(plan->pp, b, plan->s_mel.as<float>(), plan->s_energy.as<double>(),
                              d_out, s)))
      return rc;
    if (own_stream) mark_kernel(plan, "plp_tail_kernel");
  } else {
    // append_deltas as two launches: the cepstra go to a scratch, the delta kernel forms the rows
    float* feat_out = d_out;
    int feat_cols = plan->ndims;
    if (plan->chain_deltas) {
      feat_cols = plan->o.num_ceps;
      if ((rc = plan->s_cep.ensure(sizeof(float) * static_cast<size_t>(total_frames) * feat_cols))) return rc;
      feat_out = plan->s_cep.as<float>();
    }
    if (use_fast && plan->mfcc_via_fbank) {
      // [log energy |] log-mel rows from the filterbank kernel, then DCT / lifter / energy / htk conventions
      const int nb = plan->o.mel.num_bins, in_cols = nb + (plan->o.use_energy ? 1 : 0);
      if ((rc = plan->s_mel.ensure(sizeof(float) * static_cast<size_t>(total_frames) * in_cols))) return rc;
      if ((rc = run_fast(plan->s_mel.as<float>(), in_cols, nullptr))) return rc;
      if ((rc = launch_mfcc_dct(plan->s_mel.as<float>(), in_cols, nb, plan->o.num_ceps, plan->d_dct_t.as<float>(),
                                plan->mp.lifter, plan->o.use_energy ? 1 : 0, plan->o.htk_compat ? 1 : 0,
                                total_frames, feat_out, feat_cols, s)))
        return rc;
      if (own_stream) mark_kernel(plan, "mfcc_dct_kernel");
    } else if (use_fast) {
      if ((rc = run_fast(feat_out, feat_cols, nullptr))) return rc;
    } else if (use_long) {
      if ((rc = run_long(feat_out, feat_cols, nullptr))) return rc;
    } else {
      if ((rc = launch_mel_features(plan->mp, b, feat_out, feat_cols, nullptr, s))) return rc;
      if (own_stream) mark_kernel(plan, "mel_features_generic_kernel");
    }
    if ((rc = run_short(feat_out, feat_cols, nullptr))) return rc;
    if (plan->chain_deltas) {
      if ((rc = plan->s_tile.ensure(4 * sizeof(int64_t) * static_cast<size_t>(total_frames / 32 + 2)))) return rc;
      if ((rc = launch_deltas(plan->dp, feat_out, feat_cols, plan->s_foff.as<int64_t>(), n_utts, total_frames,
                              d_out, plan->s_tile.as<int64_t>(), !plan->tile_valid, s)))
        return rc;
      // (the tile records are cached: complete before a later call on another stream may use them)
      if (!plan->tile_valid && !own_stream) SNF_HIP_CHECK(hipStreamSynchronize(s));
      plan->tile_valid = true;
      if (own_stream) mark_kernel(plan, "delta_kernel");
    }
  }
  if (own_stream) SNF_HIP_CHECK(hipStreamSynchronize(s));
  return SNF_OK;
}

int snf_plan_run_batch(snf_plan* plan, const int16_t* wave, const int64_t* sample_offsets,
                       int64_t n_utts, const float* vtln_warp, float* out,
                       const int64_t* frame_offsets) {
  if (!plan) return set_error(SNF_E_INVALID, "null plan");
  std::lock_guard<std::mutex> host_lock(plan->host_mu);
  if (n_utts <= 0) return n_utts == 0 ? SNF_OK : set_error(SNF_E_INVALID, "n_utts < 0");
  if (!sample_offsets || !frame_offsets) return set_error(SNF_E_INVALID, "null offsets table");
  const int64_t total_samples = sample_offsets[n_utts] - sample_offsets[0];
  const int64_t total_frames = frame_offsets[n_utts];
  if (sample_offsets[0] != 0 || frame_offsets[0] != 0)
    return set_error(SNF_E_INVALID, "offsets tables must start at 0");
  int16_t* d_wave;
  float* d_out;
  {
    std::lock_guard<std::mutex> lock(plan->mu);
    int rc = guard_device(plan);
    if (rc) return rc;
    if ((rc = plan->s_wave.ensure(sizeof(int16_t) * static_cast<size_t>(total_samples > 0 ? total_samples : 1)))) return rc;
    if ((rc = plan->s_out.ensure(sizeof(float) * static_cast<size_t>(total_frames > 0 ? total_frames : 1) *
                                 (plan->ndims > 0 ? plan->ndims : 1))))
      return rc;
    d_wave = plan->s_wave.as<int16_t>();
    d_out = plan->s_out.as<float>();
    if (total_samples > 0)
      SNF_HIP_CHECK(hipMemcpyAsync(d_wave, wave, sizeof(int16_t) * total_samples, hipMemcpyHostToDevice,
                                   plan->stream));
  }
  int rc = snf_plan_run_batch_device(plan, d_wave, sample_offsets, n_utts, vtln_warp, d_out,
                                     frame_offsets, nullptr);
  if (rc) return rc;
  if (total_frames > 0) {
    std::lock_guard<std::mutex> lock(plan->mu);
    SNF_HIP_CHECK(hipMemcpyAsync(out, d_out, sizeof(float) * total_frames * plan->ndims,
                                 hipMemcpyDeviceToHost, plan->stream));
    SNF_HIP_CHECK(hipStreamSynchronize(plan->stream));
  }
  return SNF_OK;
}

int32_t snf_post_ndims(const snf_plan* plan, int32_t in_cols) {
  if (!plan) return -1;
  if (plan->kind == SNF_KIND_DELTA) return in_cols * (plan->o.delta_order + 1);
  if (plan->kind == SNF_KIND_PITCH_POST) return plan->ndims;
  if (plan->kind == SNF_KIND_VAD) return 1;
  if (plan->kind == SNF_KIND_SLIDING_CMVN) return in_cols;
  return -1;
}

int snf_post_run_batch_device(snf_plan* plan, const float* d_in, int32_t in_cols,
                              const int64_t* frame_offsets, int64_t n_utts, float* d_out,
                              void* stream) {
  const uint64_t named_call = take_noise_call();
  if (!plan) return set_error(SNF_E_INVALID, "null plan");
  if (n_utts <= 0) return n_utts == 0 ? SNF_OK : set_error(SNF_E_INVALID, "n_utts < 0");
  if (!frame_offsets) return set_error(SNF_E_INVALID, "null offsets table");
  if (frame_offsets[0] != 0) return set_error(SNF_E_INVALID, "offsets tables must start at 0");
  for (int64_t u = 0; u < n_utts; ++u)
    if (frame_offsets[u + 1] < frame_offsets[u])
      return set_error(SNF_E_INVALID, "offsets tables must be non-decreasing");
  std::lock_guard<std::mutex> lock(plan->mu);
  int rc = guard_device(plan);
  if (rc) return rc;
  const int64_t total_frames = frame_offsets[n_utts];
  if (total_frames == 0) return SNF_OK;
  hipStream_t s = stream ? static_cast<hipStream_t>(stream) : plan->stream;
  const bool own_stream = (stream == nullptr);
  // the offsets table (and what the delta kernel derives from it) stays on the device between calls
  // with the same table: a pipeline runs the same batch layout call after call
  const bool same_table = plan->post_table_cols == in_cols &&
                          plan->h_foff.size() == static_cast<size_t>(n_utts) + 1 &&
                          std::equal(plan->h_foff.begin(), plan->h_foff.end(), frame_offsets);
  if (!same_table) {
    plan->h_foff.assign(frame_offsets, frame_offsets + n_utts + 1);
    plan->post_table_cols = -1;
    if ((rc = plan->s_foff.upload(plan->h_foff, s))) return rc;
    SNF_HIP_CHECK(hipStreamSynchronize(s));
  }
  if (own_stream) begin_timing(plan);
  if (plan->kind == SNF_KIND_DELTA) {
    if (in_cols <= 0) return set_error(SNF_E_INVALID, "in_cols must be positive");
    if ((rc = plan->s_futt.ensure(4 * sizeof(int64_t) * static_cast<size_t>(total_frames / 32 + 2)))) return rc;
    if ((rc = launch_deltas(plan->dp, d_in, in_cols, plan->s_foff.as<int64_t>(), n_utts,
                            total_frames, d_out, plan->s_futt.as<int64_t>(), !same_table, s)))
      return rc;
    // (the tile records are complete before a later call on another stream may use them)
    if (!same_table && !own_stream) SNF_HIP_CHECK(hipStreamSynchronize(s));
    plan->post_table_cols = in_cols;
    if (own_stream) mark_kernel(plan, "delta_kernel");
  } else if (plan->kind == SNF_KIND_PITCH_POST) {
    if (in_cols != 2)
      return set_error(SNF_E_INVALID, "data shape must be (_, 2), but it is (_, " +
                                          std::to_string(in_cols) + ")");
    if (plan->ppost.o.delta_pitch_noise_stddev != 0.0f)
      plan->ppost.seed = plan->o.seed + 0x9E3779B97F4A7C15ull * (named_call ? named_call : ++plan->noise_calls);
    if ((rc = launch_pitch_post(plan->ppost, d_in, plan->s_foff.as<int64_t>(), n_utts,
                                total_frames, d_out, s)))
      return rc;
    if (own_stream) mark_kernel(plan, "pitch_post_kernel");
  } else if (plan->kind == SNF_KIND_VAD) {
    if (in_cols <= 0) return set_error(SNF_E_INVALID, "in_cols must be positive");
    if ((rc = plan->s_stats.ensure(sizeof(float) * static_cast<size_t>(n_utts)))) return rc;
    if ((rc = launch_vad(plan->o.vad, d_in, in_cols, plan->s_foff.as<int64_t>(), n_utts,
                         total_frames, plan->s_stats.as<float>(), d_out, s)))
      return rc;
    if (own_stream) mark_kernel(plan, "vad_kernel");
  } else if (plan->kind == SNF_KIND_SLIDING_CMVN) {
    if (in_cols <= 0) return set_error(SNF_E_INVALID, "in_cols must be positive");
    if ((rc = launch_sliding_cmvn(plan->o.sliding_cmvn, d_in, in_cols, plan->s_foff.as<int64_t>(),
                                  n_utts, d_out, s)))
      return rc;
    if (own_stream) mark_kernel(plan, "sliding_cmvn_kernel");
  } else {
    return set_error(SNF_E_INVALID, "plan kind is not a post-processor");
  }
  if (own_stream) SNF_HIP_CHECK(hipStreamSynchronize(s));
  return SNF_OK;
}

int snf_post_run_batch(snf_plan* plan, const float* in, int32_t in_cols,
                       const int64_t* frame_offsets, int64_t n_utts, float* out) {
  if (!plan) return set_error(SNF_E_INVALID, "null plan");
  std::lock_guard<std::mutex> host_lock(plan->host_mu);
  if (n_utts <= 0) return n_utts == 0 ? SNF_OK : set_error(SNF_E_INVALID, "n_utts < 0");
  if (!frame_offsets) return set_error(SNF_E_INVALID, "null offsets table");
  const int64_t total_frames = frame_offsets[n_utts];
  const int32_t out_cols = snf_post_ndims(plan, in_cols);
  if (out_cols <= 0 || in_cols <= 0) return set_error(SNF_E_INVALID, "bad column count");
  if (total_frames == 0) return SNF_OK;
  float *d_in, *d_out;
  {
    std::lock_guard<std::mutex> lock(plan->mu);
    int rc = guard_device(plan);
    if (rc) return rc;
    if ((rc = plan->s_in.ensure(sizeof(float) * static_cast<size_t>(total_frames) * in_cols))) return rc;
    if ((rc = plan->s_out.ensure(sizeof(float) * static_cast<size_t>(total_frames) * out_cols))) return rc;
    d_in = plan->s_in.as<float>();
    d_out = plan->s_out.as<float>();
    SNF_HIP_CHECK(hipMemcpyAsync(d_in, in, sizeof(float) * total_frames * in_cols, hipMemcpyHostToDevice,
                                 plan->stream));
  }
  int rc = snf_post_run_batch_device(plan, d_in, in_cols, frame_offsets, n_utts, d_out, nullptr);
  if (rc) return rc;
  std::lock_guard<std::mutex> lock(plan->mu);
  SNF_HIP_CHECK(hipMemcpyAsync(out, d_out, sizeof(float) * total_frames * out_cols, hipMemcpyDeviceToHost,
                               plan->stream));
  SNF_HIP_CHECK(hipStreamSynchronize(plan->stream));
  return SNF_OK;
}

namespace {
int cmvn_check(const snf_plan* plan, int32_t cols, const int64_t* frame_offsets, int64_t n_utts,
               const int32_t* group, int32_t n_groups) {
  if (!plan) return set_error(SNF_E_INVALID, "null plan");
  if (plan->kind != SNF_KIND_CMVN) return set_error(SNF_E_INVALID, "plan kind is not CMVN");
  if (n_utts < 0) return set_error(SNF_E_INVALID, "n_utts < 0");
  if (cols <= 0) return set_error(SNF_E_INVALID, "dimension must be a strictly positive integer");
  if (n_groups <= 0) return set_error(SNF_E_INVALID, "n_groups must be positive");
  if (n_utts > 0 && !frame_offsets) return set_error(SNF_E_INVALID, "null offsets table");
  if (n_utts > 0 && frame_offsets[0] != 0) return set_error(SNF_E_INVALID, "offsets tables must start at 0");
  for (int64_t u = 0; u < n_utts; ++u) {
    if (frame_offsets[u + 1] < frame_offsets[u])
      return set_error(SNF_E_INVALID, "offsets tables must be non-decreasing");
    const int32_t g = group ? group[u] : 0;
    if (g < 0 || g >= n_groups) return set_error(SNF_E_INVALID, "group index out of range");
  }
  return SNF_OK;
}
}  // namespace

int snf_cmvn_accumulate_device(snf_plan* plan, const float* d_in, int32_t cols,
                               const int64_t* frame_offsets, int64_t n_utts, const float* d_weights,
                               const int32_t* group, int32_t n_groups, double* stats) {
  int rc = cmvn_check(plan, cols, frame_offsets, n_utts, group, n_groups);
  if (rc) return rc;
  if (n_utts == 0) return SNF_OK;
  if (!stats) return set_error(SNF_E_INVALID, "null stats");
  std::lock_guard<std::mutex> lock(plan->mu);
  if ((rc = guard_device(plan))) return rc;
  const int64_t total_frames = frame_offsets[n_utts];
  if (total_frames == 0) return SNF_OK;
  if (!d_in) return set_error(SNF_E_INVALID, "null input");
  hipStream_t s = plan->stream;
  const size_t blk = 2 * static_cast<size_t>(cols + 1);
  if ((rc = plan->s_stats.ensure(sizeof(double) * blk * static_cast<size_t>(n_utts)))) return rc;
  std::vector<int64_t> foff(frame_offsets, frame_offsets + n_utts + 1);
  if ((rc = plan->s_foff.upload(foff, s))) return rc;
  begin_timing(plan);
  if ((rc = launch_cmvn_stats(d_in, cols, plan->s_foff.as<int64_t>(), d_weights, n_utts,
                              plan->s_stats.as<double>(), s)))
    return rc;
  mark_kernel(plan, "cmvn_stats_kernel");
  std::vector<double> per_utt(blk * static_cast<size_t>(n_utts));
  SNF_HIP_CHECK(hipMemcpyAsync(per_utt.data(), plan->s_stats.p, sizeof(double) * per_utt.size(),
                               hipMemcpyDeviceToHost, s));
  SNF_HIP_CHECK(hipStreamSynchronize(s));
  // the per-speaker sum runs over a handful of [2, cols+1] blocks: host, in utterance order
  for (int64_t u = 0; u < n_utts; ++u) {
    double* dst = stats + blk * static_cast<size_t>(group ? group[u] : 0);
    const double* src = per_utt.data() + blk * static_cast<size_t>(u);
    for (size_t i = 0; i < blk; ++i) dst[i] += src[i];
  }
  return SNF_OK;
}

int snf_cmvn_accumulate(snf_plan* plan, const float* in, int32_t cols, const int64_t* frame_offsets,
                        int64_t n_utts, const float* weights, const int32_t* group,
                        int32_t n_groups, double* stats) {
  int rc = cmvn_check(plan, cols, frame_offsets, n_utts, group, n_groups);
  if (rc) return rc;
  if (n_utts == 0) return SNF_OK;
  const int64_t total_frames = frame_offsets[n_utts];
  if (total_frames == 0) return SNF_OK;
  if (!in) return set_error(SNF_E_INVALID, "null input");
  std::lock_guard<std::mutex> host_lock(plan->host_mu);
  const float *d_in, *d_w = nullptr;
  {
    std::lock_guard<std::mutex> lock(plan->mu);
    if ((rc = guard_device(plan))) return rc;
    if ((rc = plan->s_in.ensure(sizeof(float) * static_cast<size_t>(total_frames) * cols))) return rc;
    SNF_HIP_CHECK(hipMemcpyAsync(plan->s_in.p, in, sizeof(float) * total_frames * cols,
                                 hipMemcpyHostToDevice, plan->stream));
    d_in = plan->s_in.as<float>();
    if (weights) {
      if ((rc = plan->s_energy.ensure(sizeof(float) * static_cast<size_t>(total_frames)))) return rc;
      SNF_HIP_CHECK(hipMemcpyAsync(plan->s_energy.p, weights, sizeof(float) * total_frames,
                                   hipMemcpyHostToDevice, plan->stream));
      d_w = plan->s_energy.as<float>();
    }
  }
  return snf_cmvn_accumulate_device(plan, d_in, cols, frame_offsets, n_utts, d_w, group, n_groups, stats);
}

int snf_cmvn_apply_device(snf_plan* plan, const float* d_in, int32_t cols,
                          const int64_t* frame_offsets, int64_t n_utts, const double* stats,
                          const int32_t* group, int32_t n_groups, int32_t norm_vars, int32_t reverse,
                          float* d_out) {
  int rc = cmvn_check(plan, cols, frame_offsets, n_utts, group, n_groups);
  if (rc) return rc;
  if (n_utts == 0) return SNF_OK;
  if (!stats) return set_error(SNF_E_INVALID, "null stats");
  // [KALDI-UPSTREAM] transform/cmvn.cc ApplyCmvn / ApplyCmvnReverse: float (offset, scale) per column
  const size_t blk = 2 * static_cast<size_t>(cols + 1);
  std::vector<float> norm(static_cast<size_t>(n_groups) * 2 * cols, 0.0f);
  std::vector<char> used(n_groups, 0);
  for (int64_t u = 0; u < n_utts; ++u) used[group ? group[u] : 0] = 1;
  for (int32_t g = 0; g < n_groups; ++g) {
    if (!used[g]) continue;
    const double* st = stats + blk * static_cast<size_t>(g);
    const double count = st[cols];
    if (count < 1.0)
      return set_error(SNF_E_INVALID, "Insufficient stats for cepstral mean and variance "
                                      "normalization: count = " + std::to_string(count));
    float* off = norm.data() + static_cast<size_t>(g) * 2 * cols;
    float* scl = off + cols;
    for (int d = 0; d < cols; ++d) {
      const double mean = st[d] / count;
      double offset, scale;
      if (!reverse) {
        // without variance normalisation Kaldi adds offset.AddVec(-1.0 / count, mean_stats), whose
        // alpha is a BaseFloat
        offset = static_cast<double>(static_cast<float>(-1.0 / count)) * st[d];
        scale = 1.0;
        if (norm_vars) {
          double var = st[(cols + 1) + d] / count - mean * mean;
          const double floor = 1.0e-20;
          if (var < floor) var = floor;
          scale = 1.0 / std::sqrt(var);
          if (scale != scale || 1.0 / scale == 0.0)
            return set_error(SNF_E_RUNTIME, "NaN or infinity in cepstral mean/variance computation");
          offset = -(mean * scale);
        }
      } else {
        offset = mean;
        scale = 1.0;
        if (norm_vars) {
          double var = st[(cols + 1) + d] / count - mean * mean;
          const double floor = 1.0e-20;
          if (var < floor) var = floor;
          scale = std::sqrt(var);
        }
      }
      off[d] = static_cast<float>(offset);
      scl[d] = static_cast<float>(scale);
    }
  }
  std::lock_guard<std::mutex> lock(plan->mu);
  if ((rc = guard_device(plan))) return rc;
  const int64_t total_frames = frame_offsets[n_utts];
  if (total_frames == 0) return SNF_OK;
  if (!d_in || !d_out) return set_error(SNF_E_INVALID, "null buffer");
  hipStream_t s = plan->stream;
  std::vector<int64_t> foff(frame_offsets, frame_offsets + n_utts + 1);
  if ((rc = plan->s_foff.upload(foff, s))) return rc;
  if ((rc = plan->s_mel.upload(norm, s))) return rc;
  const int32_t* d_group = nullptr;
  if (group) {
    std::vector<int32_t> gv(group, group + n_utts);
    if ((rc = plan->s_uwarp.upload(gv, s))) return rc;
    d_group = plan->s_uwarp.as<int32_t>();
  }
  begin_timing(plan);
  int64_t max_frames = 0;
  for (int64_t k = 0; k < n_utts; ++k)
    max_frames = std::max(max_frames, frame_offsets[k + 1] - frame_offsets[k]);
  if ((rc = launch_cmvn_apply(d_in, cols, plan->s_foff.as<int64_t>(), n_utts, max_frames, d_group,
                              plan->s_mel.as<float>(), norm_vars ? 1 : 0, d_out, s)))
    return rc;
  mark_kernel(plan, "cmvn_apply_kernel");
  SNF_HIP_CHECK(hipStreamSynchronize(s));
  return SNF_OK;
}

int snf_cmvn_apply(snf_plan* plan, const float* in, int32_t cols, const int64_t* frame_offsets,
                   int64_t n_utts, const double* stats, const int32_t* group, int32_t n_groups,
                   int32_t norm_vars, int32_t reverse, float* out) {
  int rc = cmvn_check(plan, cols, frame_offsets, n_utts, group, n_groups);
  if (rc) return rc;
  if (n_utts == 0) return SNF_OK;
  const int64_t total_frames = frame_offsets[n_utts];
  if (total_frames == 0) return SNF_OK;
  if (!in || !out) return set_error(SNF_E_INVALID, "null buffer");
  std::lock_guard<std::mutex> host_lock(plan->host_mu);
  const size_t bytes = sizeof(float) * static_cast<size_t>(total_frames) * cols;
  float *d_in, *d_out;
  {
    std::lock_guard<std::mutex> lock(plan->mu);
    if ((rc = guard_device(plan))) return rc;
    if ((rc = plan->s_in.ensure(bytes))) return rc;
    if ((rc = plan->s_out.ensure(bytes))) return rc;
    SNF_HIP_CHECK(hipMemcpyAsync(plan->s_in.p, in, bytes, hipMemcpyHostToDevice, plan->stream));
    d_in = plan->s_in.as<float>();
    d_out = plan->s_out.as<float>();
  }
  rc = snf_cmvn_apply_device(plan, d_in, cols, frame_offsets, n_utts, stats, group, n_groups, norm_vars,
                             reverse, d_out);
  if (rc) return rc;
  std::lock_guard<std::mutex> lock(plan->mu);
  SNF_HIP_CHECK(hipMemcpyAsync(out, d_out, bytes, hipMemcpyDeviceToHost, plan->stream));
  SNF_HIP_CHECK(hipStreamSynchronize(plan->stream));
  return SNF_OK;
}

int snf_concat_columns_device(int device_id, const float* d_a, int32_t cols_a,
                              const int64_t* offsets_a, const float* d_b, int32_t cols_b,
                              const int64_t* offsets_b, int64_t n_utts, float* d_out,
                              const int64_t* offsets_out) {
  if (n_utts < 0) return set_error(SNF_E_INVALID, "n_utts < 0");
  if (n_utts == 0) return SNF_OK;
  if (!offsets_a || !offsets_b || !offsets_out) return set_error(SNF_E_INVALID, "null offsets table");
  if (cols_a <= 0 || cols_b <= 0) return set_error(SNF_E_INVALID, "bad column count");
  for (int64_t u = 0; u < n_utts; ++u) {
    const int64_t na = offsets_a[u + 1] - offsets_a[u], nb = offsets_b[u + 1] - offsets_b[u];
    const int64_t no = offsets_out[u + 1] - offsets_out[u];
    if (na < 0 || nb < 0 || no < 0 || no > na || no > nb)
      return set_error(SNF_E_INVALID, "concatenation rows exceed an input for utterance " +
                                          std::to_string(u));
  }
  const int64_t total = offsets_out[n_utts];
  if (total == 0) return SNF_OK;
  if (!d_a || !d_b || !d_out) return set_error(SNF_E_INVALID, "null buffer");
  SNF_HIP_CHECK(hipSetDevice(device_id));
  ThreadScratch* t = thread_scratch(device_id);
  if (!t) return SNF_E_HIP;
  int rc = t->buf.ensure(sizeof(int64_t) * 3 * (n_utts + 1));
  if (rc) return rc;
  int64_t* d_off = t->buf.as<int64_t>();
  // (pageable sources: each copy has read its source when it returns; everything on the thread's own stream,
  // waited for alone - a device-wide wait would also wait for the tracker and the copies of other batches)
  SNF_HIP_CHECK(hipMemcpyAsync(d_off, offsets_a, sizeof(int64_t) * (n_utts + 1), hipMemcpyHostToDevice, t->stream));
  SNF_HIP_CHECK(hipMemcpyAsync(d_off + (n_utts + 1), offsets_b, sizeof(int64_t) * (n_utts + 1),
                               hipMemcpyHostToDevice, t->stream));
  SNF_HIP_CHECK(hipMemcpyAsync(d_off + 2 * (n_utts + 1), offsets_out, sizeof(int64_t) * (n_utts + 1),
                               hipMemcpyHostToDevice, t->stream));
  rc = launch_concat_columns(d_a, cols_a, d_off, d_b, cols_b, d_off + (n_utts + 1), n_utts, d_out,
                             d_off + 2 * (n_utts + 1), total, t->stream);
  if (hipStreamSynchronize(t->stream) != hipSuccess && !rc) rc = set_error(SNF_E_HIP, "concat kernel failed");
  return rc;
}

int snf_count_nonfinite_device(int device_id, const float* d_data, uint64_t n, uint64_t* count) {
  if (!count) return set_error(SNF_E_INVALID, "null count");
  *count = 0;
  if (n == 0) return SNF_OK;
  if (!d_data) return set_error(SNF_E_INVALID, "null buffer");
  if (reinterpret_cast<uintptr_t>(d_data) & 15) return set_error(SNF_E_INVALID, "buffer is not 16-byte aligned");
  SNF_HIP_CHECK(hipSetDevice(device_id));
  ThreadScratch* t = thread_scratch(device_id);
  if (!t) return SNF_E_HIP;
  int rc = t->buf.ensure(sizeof(unsigned long long));
  if (rc) return rc;
  unsigned long long* d_count = t->buf.as<unsigned long long>();
  unsigned long long host = 0;
  SNF_HIP_CHECK(hipMemsetAsync(d_count, 0, sizeof(unsigned long long), t->stream));
  rc = launch_count_nonfinite(d_data, n, d_count, t->stream);
  if (!rc && (hipMemcpyAsync(&host, d_count, sizeof(host), hipMemcpyDeviceToHost, t->stream) != hipSuccess ||
              hipStreamSynchronize(t->stream) != hipSuccess))
    rc = set_error(SNF_E_HIP, "non-finite count kernel failed");
  *count = host;
  return rc;
}

int snf_malloc(void** dptr, uint64_t bytes) {
  SNF_HIP_CHECK(hipMalloc(dptr, bytes));
  return SNF_OK;
}
int snf_mem_info(uint64_t* free_bytes, uint64_t* total_bytes) {
  size_t f = 0, t = 0;
  SNF_HIP_CHECK(hipMemGetInfo(&f, &t));
  if (free_bytes) *free_bytes = f;
  if (total_bytes) *total_bytes = t;
  return SNF_OK;
}
int snf_set_oom_hook(snf_oom_hook hook) {
  g_oom_hook.store(hook);
  return SNF_OK;
}
int snf_free(void* dptr) {
  SNF_HIP_CHECK(hipFree(dptr));
  return SNF_OK;
}
int snf_memcpy_h2d(void* dst, const void* src, uint64_t bytes) {
  SNF_HIP_CHECK(hipMemcpy(dst, src, bytes, hipMemcpyHostToDevice));
  return SNF_OK;
}
int snf_memcpy_d2h(void* dst, const void* src, uint64_t bytes) {
  SNF_HIP_CHECK(hipMemcpy(dst, src, bytes, hipMemcpyDeviceToHost));
  return SNF_OK;
}
int snf_stream_create(void** stream) {
  if (!stream) return set_error(SNF_E_INVALID, "null pointer");
  hipStream_t s;
  SNF_HIP_CHECK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  *stream = s;
  return SNF_OK;
}
int snf_stream_destroy(void* stream) {
  if (stream) SNF_HIP_CHECK(hipStreamDestroy(static_cast<hipStream_t>(stream)));
  return SNF_OK;
}
int snf_stream_synchronize(void* stream) {
  SNF_HIP_CHECK(hipStreamSynchronize(static_cast<hipStream_t>(stream)));
  return SNF_OK;
}
int snf_stream_query(void* stream) {
  const hipError_t e = hipStreamQuery(static_cast<hipStream_t>(stream));
  if (e == hipSuccess) return 0;
  if (e == hipErrorNotReady) {
    (void)hipGetLastError();   // (not an error: nothing to leave behind for the next call's check)
    return 1;
  }
  return set_error(SNF_E_HIP, std::string("hipStreamQuery: ") + hipGetErrorString(e));
}
int snf_event_create(void** event) {
  if (!event) return set_error(SNF_E_INVALID, "null pointer");
  hipEvent_t e;
  SNF_HIP_CHECK(hipEventCreate(&e));
  *event = e;
  return SNF_OK;
}
int snf_event_destroy(void* event) {
  if (event) SNF_HIP_CHECK(hipEventDestroy(static_cast<hipEvent_t>(event)));
  return SNF_OK;
}
int snf_event_record(void* event, void* stream) {
  if (!event) return set_error(SNF_E_INVALID, "null event");
  SNF_HIP_CHECK(hipEventRecord(static_cast<hipEvent_t>(event), static_cast<hipStream_t>(stream)));
  return SNF_OK;
}
int snf_event_synchronize(void* event) {
  if (!event) return set_error(SNF_E_INVALID, "null event");
  SNF_HIP_CHECK(hipEventSynchronize(static_cast<hipEvent_t>(event)));
  return SNF_OK;
}
int snf_stream_wait_event(void* stream, void* event) {
  if (!event) return set_error(SNF_E_INVALID, "null event");
  SNF_HIP_CHECK(hipStreamWaitEvent(static_cast<hipStream_t>(stream), static_cast<hipEvent_t>(event), 0));
  return SNF_OK;
}
int snf_event_elapsed_ms(void* start, void* stop, float* ms) {
  if (!start || !stop || !ms) return set_error(SNF_E_INVALID, "null pointer");
  SNF_HIP_CHECK(hipEventSynchronize(static_cast<hipEvent_t>(stop)));
  SNF_HIP_CHECK(hipEventElapsedTime(ms, static_cast<hipEvent_t>(start), static_cast<hipEvent_t>(stop)));
  return SNF_OK;
}
int snf_memcpy_h2d_async(void* dst, const void* src, uint64_t bytes, void* stream) {
  SNF_HIP_CHECK(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, static_cast<hipStream_t>(stream)));
  return SNF_OK;
}
int snf_memcpy_d2h_async(void* dst, const void* src, uint64_t bytes, void* stream) {
  SNF_HIP_CHECK(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, static_cast<hipStream_t>(stream)));
  return SNF_OK;
}
int snf_memset(void* dst, int value, uint64_t bytes) {
  // hipMemset on device memory returns before the fill has run, and the plans' streams are non-blocking:
  // they do not wait for the null stream.  A caller that fills a buffer and then hands it to a plan expects
  // the fill to be over (found by the pipeline fuzzer: the fill landed on top of a kernel's output).
  SNF_HIP_CHECK(hipMemset(dst, value, bytes));
  SNF_HIP_CHECK(hipStreamSynchronize(nullptr));
  return SNF_OK;
}
namespace {
__global__ __launch_bounds__(256) void lds_fill_kernel(unsigned pattern, int words, unsigned* sink) {
  extern __shared__ unsigned fill[];
  for (int i = threadIdx.x; i < words; i += blockDim.x) fill[i] = pattern;
  __syncthreads();
  // (a dependent read keeps the stores from being optimised away)
  if (fill[(threadIdx.x * 97) % words] != pattern) sink[0] = 1;
}
}  // namespace

int snf_debug_pitch_scratch(snf_plan* plan, void** down, void** nccf_res, void** pov_nccf,
                            void** states) {
  if (!plan || plan->kind != SNF_KIND_PITCH) return set_error(SNF_E_INVALID, "not a pitch plan");
  if (down) *down = plan->s_down.p;
  if (nccf_res) *nccf_res = plan->s_pres.p;
  if (pov_nccf) *pov_nccf = plan->s_mel.p;
  if (states) *states = plan->s_states.p;
  return SNF_OK;
}
int snf_debug_fill_lds(uint32_t pattern) {
  // two 80 KB workgroups cover the 160 KB of a CU; many more workgroups than CUs so that every CU
  // (and both halves of its LDS) is visited
  const int bytes = 80 * 1024 - 256;
  unsigned* sink = nullptr;
  SNF_HIP_CHECK(hipMalloc(&sink, sizeof(unsigned)));
  SNF_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(lds_fill_kernel),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
  hipLaunchKernelGGL(lds_fill_kernel, dim3(256 * 32), dim3(256), bytes, nullptr, pattern, bytes / 4, sink);
  hipError_t e = hipGetLastError();
  if (e == hipSuccess) e = hipDeviceSynchronize();
  (void)hipFree(sink);
  if (e != hipSuccess) return snf::set_error(SNF_E_HIP, std::string("lds fill: ") + hipGetErrorString(e));
  return SNF_OK;
}

int snf_host_malloc(void** hptr, uint64_t bytes) {
  if (!hptr) return set_error(SNF_E_INVALID, "null pointer");
  SNF_HIP_CHECK(hipHostMalloc(hptr, bytes > 0 ? bytes : 1, hipHostMallocDefault));
  return SNF_OK;
}
int snf_host_free(void* hptr) {
  if (hptr) SNF_HIP_CHECK(hipHostFree(hptr));
  return SNF_OK;
}

float snf_plan_last_kernel_ms(const snf_plan* plan, int which) {
  if (!plan || !plan->events_valid || which < 0 || which > plan->n_slots) return -1.0f;
  float ms = -1.0f;
  if (hipEventSynchronize(plan->ev[plan->n_slots]) != hipSuccess) return -1.0f;
  hipError_t e = which == 0 ? hipEventElapsedTime(&ms, plan->ev[0], plan->ev[plan->n_slots])
                            : hipEventElapsedTime(&ms, plan->ev[which - 1], plan->ev[which]);
  return e == hipSuccess ? ms : -1.0f;
}
const char* snf_plan_kernel_name(const snf_plan* plan, int which) {
  if (!plan || which <= 0 || which > plan->n_slots) return nullptr;
  return plan->slot_name[which];
}

}  // extern "C"
