// Host-side precomputed tables of the MI355X speech-features backend.
//
// These are the quantities Kaldi precomputes once per computer object and the reference rebuilds
// for every utterance (reference shennong/processor/base.py:429-431 constructs `cls(options)` per
// call; plp.py:443-508 rebuilds mel banks / IDFT bases / lifter per call).  Here they are built once
// per plan and kept resident in HBM.
//
// Formulas follow the published Kaldi sources ([KALDI-UPSTREAM], not in /root/reference):
// feature-window.cc (framing, window), mel-computations.cc (MelBanks, VTLN warp, lifter, equal
// loudness), matrix-functions.cc (ComputeDctMatrix), feature-functions.cc (InitIdftBases,
// DeltaFeatures), resample.cc (LinearResample/ArbitraryResample), pitch-functions.cc (SelectLags).
#include <cmath>
#include <cstring>

#include "snf_internal.h"

namespace snf {

static thread_local std::string g_error;
int set_error(int code, const std::string& msg) {
  g_error = msg;
  return code;
}
const char* last_error() { return g_error.c_str(); }

static constexpr double kPi = 3.14159265358979323846;
static constexpr double kTwoPi = 6.283185307179586476925286766559005;

// ---- framing ---------------------------------------------------------------------------------
int32_t window_shift(const snf_frame_options& o) {
  return static_cast<int32_t>(static_cast<double>(o.samp_freq) * 0.001 * o.frame_shift_ms);
}
int32_t window_size(const snf_frame_options& o) {
  return static_cast<int32_t>(static_cast<double>(o.samp_freq) * 0.001 * o.frame_length_ms);
}
int32_t padded_window_size(const snf_frame_options& o) {
  int32_t w = window_size(o);
  if (!o.round_to_power_of_two) return w;
  int32_t p = 1;
  while (p < w) p <<= 1;
  return p;
}
int64_t num_frames(const snf_frame_options& o, int64_t n) {
  const int64_t shift = window_shift(o), len = window_size(o);
  if (shift <= 0) return 0;
  if (o.snip_edges) return n < len ? 0 : 1 + (n - len) / shift;
  return (n + shift / 2) / shift;
}
int64_t first_sample_of_frame(const snf_frame_options& o, int64_t frame) {
  const int64_t shift = window_shift(o);
  if (o.snip_edges) return frame * shift;
  return shift * frame + shift / 2 - window_size(o) / 2;
}

int make_window(const snf_frame_options& o, std::vector<float>* w) {
  const int32_t n = window_size(o);
  w->assign(n > 0 ? n : 0, 0.0f);
  const double a = kTwoPi / (n - 1);
  for (int32_t i = 0; i < n; ++i) {
    const double c = std::cos(a * i);
    double v;
    switch (o.window_type) {
      case SNF_WINDOW_HANNING: v = 0.5 - 0.5 * c; break;
      case SNF_WINDOW_HAMMING: v = 0.54 - 0.46 * c; break;
      case SNF_WINDOW_POVEY: v = std::pow(0.5 - 0.5 * c, 0.85); break;
      case SNF_WINDOW_RECTANGULAR: v = 1.0; break;
      case SNF_WINDOW_BLACKMAN:
        v = static_cast<double>(o.blackman_coeff) - 0.5 * c +
            (0.5 - static_cast<double>(o.blackman_coeff)) * std::cos(2 * a * i);
        break;
      default: return set_error(SNF_E_INVALID, "invalid window type");
    }
    (*w)[i] = static_cast<float>(v);
  }
  return SNF_OK;
}

// ---- mel banks ---------------------------------------------------------------------------------
static inline float mel_of(float hz) { return 1127.0f * logf(1.0f + hz / 700.0f); }
static inline float hz_of(float mel) { return 700.0f * (expf(mel / 1127.0f) - 1.0f); }

namespace {
struct VtlnWarp {
  float low_cut, high_cut, low, high, factor;
  float freq(float f) const {
    if (f < low || f > high) return f;
    const float l = low_cut * (factor > 1.0f ? factor : 1.0f);
    const float h = high_cut * (factor < 1.0f ? factor : 1.0f);
    const float scale = 1.0f / factor;
    const float fl = scale * l, fh = scale * h;
    const float slope_left = (fl - low) / (l - low);
    const float slope_right = (high - fh) / (high - h);
    if (f < l) return low + slope_left * (f - low);
    if (f < h) return scale * f;
    return high + slope_right * (f - high);
  }
  float mel(float m) const { return mel_of(freq(hz_of(m))); }
};
}  // namespace

int make_mel_banks(const snf_mel_options& mo, const snf_frame_options& fo, float vtln_warp,
                   MelBanksHost* out) {
  const int nb = mo.num_bins;
  if (nb < 3) return set_error(SNF_E_RUNTIME, "Must have at least 3 mel bins");
  const float sf = fo.samp_freq;
  const int padded = padded_window_size(fo);
  if (padded % 2 != 0)
    return set_error(SNF_E_RUNTIME, "padded window size must be even for the real FFT");
  const int nfft = padded / 2;
  const float nyquist = 0.5f * sf;
  const float low = mo.low_freq;
  const float high = mo.high_freq > 0.0f ? mo.high_freq : nyquist + mo.high_freq;
  if (low < 0.0f || low >= nyquist || high <= 0.0f || high > nyquist || high <= low)
    return set_error(SNF_E_RUNTIME, "Bad values in options: low-freq " + std::to_string(low) +
                                        " and high-freq " + std::to_string(high) +
                                        " vs. nyquist " + std::to_string(nyquist));
  const float bin_width = sf / padded;
  const float mel_low = mel_of(low), mel_high = mel_of(high);
  const float delta = (mel_high - mel_low) / (nb + 1);
  float vtln_low = mo.vtln_low, vtln_high = mo.vtln_high;
  if (vtln_high < 0.0f) vtln_high += nyquist;
  const bool warped = vtln_warp != 1.0f;
  if (warped && (vtln_low < 0.0f || vtln_low <= low || vtln_low >= high || vtln_high <= 0.0f ||
                 vtln_high >= high || vtln_high <= vtln_low))
    return set_error(SNF_E_RUNTIME, "Bad values in options: vtln-low " + std::to_string(vtln_low) +
                                        " and vtln-high " + std::to_string(vtln_high) +
                                        ", versus low-freq " + std::to_string(low) +
                                        " and high-freq " + std::to_string(high));
  const VtlnWarp vw{vtln_low, vtln_high, low, high, vtln_warp};

  out->num_bins = nb;
  out->num_fft_bins = nfft;
  out->first.assign(nb, 0);
  out->size.assign(nb, 0);
  out->offset.assign(nb, 0);
  out->center_freqs.assign(nb, 0.0f);
  out->w.clear();
  std::vector<float> fft_mel(nfft);
  for (int i = 0; i < nfft; ++i) fft_mel[i] = mel_of(bin_width * i);
  for (int b = 0; b < nb; ++b) {
    float left = mel_low + b * delta, center = mel_low + (b + 1) * delta,
          right = mel_low + (b + 2) * delta;
    if (warped) {
      left = vw.mel(left);
      center = vw.mel(center);
      right = vw.mel(right);
    }
    out->center_freqs[b] = hz_of(center);
    int first = -1, last = -1;
    std::vector<float> row(nfft, 0.0f);
    for (int i = 0; i < nfft; ++i) {
      const float mel = fft_mel[i];
      if (mel > left && mel < right) {
        row[i] = mel <= center ? (mel - left) / (center - left) : (right - mel) / (right - center);
        if (first < 0) first = i;
        last = i;
      }
    }
    if (first < 0)
      return set_error(SNF_E_RUNTIME, "You may have set num_bins too large (a mel bin is empty)");
    out->first[b] = first;
    out->size[b] = last + 1 - first;
    out->offset[b] = static_cast<int>(out->w.size());
    out->w.insert(out->w.end(), row.begin() + first, row.begin() + last + 1);
  }
  return SNF_OK;
}

void make_placeholder_banks(const snf_mel_options& mo, const snf_frame_options& fo, MelBanksHost* out) {
  const int nb = mo.num_bins > 0 ? mo.num_bins : 1;
  const int padded = padded_window_size(fo);
  const int nfft = padded / 2 > 0 ? padded / 2 : 1;
  out->num_bins = nb;
  out->num_fft_bins = nfft;
  out->first.assign(nb, 0);
  out->size.assign(nb, 1);
  out->offset.resize(nb);
  for (int b = 0; b < nb; ++b) {
    out->first[b] = b < nfft ? b : nfft - 1;
    out->offset[b] = b;
  }
  out->w.assign(nb, 0.0f);
  out->center_freqs.assign(nb, 1000.0f);
}

void make_dct_matrix(int num_rows, int num_cols, std::vector<float>* m) {
  m->assign(static_cast<size_t>(num_rows) * num_cols, 0.0f);
  const float n_f = static_cast<float>(num_cols);
  float normalizer = static_cast<float>(std::sqrt(1.0 / n_f));
  for (int j = 0; j < num_cols && num_rows > 0; ++j) (*m)[j] = normalizer;
  normalizer = static_cast<float>(std::sqrt(2.0 / n_f));
  for (int k = 1; k < num_rows; ++k)
    for (int n = 0; n < num_cols; ++n)
      (*m)[static_cast<size_t>(k) * num_cols + n] =
          static_cast<float>(normalizer * std::cos(kPi / num_cols * (n + 0.5) * k));
}

void make_lifter(float q, int n, std::vector<float>* c) {
  c->assign(n, 1.0f);
  for (int i = 0; i < n; ++i)
    (*c)[i] = static_cast<float>(1.0 + 0.5 * q * std::sin(kPi * i / q));
}

void make_equal_loudness(const MelBanksHost& mb, std::vector<float>* out) {
  out->assign(mb.num_bins, 0.0f);
  for (int i = 0; i < mb.num_bins; ++i) {
    const float fsq = mb.center_freqs[i] * mb.center_freqs[i];
    const float fsub = static_cast<float>(fsq / (fsq + 1.6e5));
    (*out)[i] = static_cast<float>((fsub * fsub) * ((fsq + 1.44e6) / (fsq + 9.61e6)));
  }
}

void make_idft_bases(int n_bases, int dim, std::vector<float>* m) {
  m->assign(static_cast<size_t>(n_bases) * dim, 0.0f);
  const float angle = static_cast<float>(kPi / static_cast<float>(dim - 1));
  const float scale = static_cast<float>(1.0f / (2.0 * static_cast<float>(dim - 1)));
  for (int i = 0; i < n_bases; ++i) {
    float* row = m->data() + static_cast<size_t>(i) * dim;
    row[0] = static_cast<float>(1.0 * scale);
    const float i_fl = static_cast<float>(i);
    for (int j = 1; j < dim - 1; ++j)
      row[j] = static_cast<float>(2.0 * scale *
                                  std::cos(static_cast<double>(angle * i_fl * static_cast<float>(j))));
    row[dim - 1] = static_cast<float>(
        scale * std::cos(static_cast<double>(angle * i_fl * static_cast<float>(dim - 1))));
  }
}

void make_delta_scales(int order, int window, std::vector<float>* scales, std::vector<int>* dims) {
  scales->clear();
  dims->clear();
  std::vector<float> prev{1.0f};
  scales->push_back(1.0f);
  dims->push_back(1);
  for (int i = 1; i <= order; ++i) {
    std::vector<float> cur(prev.size() + 2 * window, 0.0f);
    const int prev_offset = (static_cast<int>(prev.size()) - 1) / 2;
    const int cur_offset = prev_offset + window;
    float normalizer = 0.0f;
    for (int j = -window; j <= window; ++j) {
      normalizer += static_cast<float>(j * j);
      for (int k = -prev_offset; k <= prev_offset; ++k)
        cur[j + k + cur_offset] += static_cast<float>(j) * prev[k + prev_offset];
    }
    const float s = static_cast<float>(1.0 / normalizer);
    for (float& v : cur) v *= s;
    scales->insert(scales->end(), cur.begin(), cur.end());
    dims->push_back(static_cast<int>(cur.size()));
    prev.swap(cur);
  }
}

// ---- resamplers (pitch) ------------------------------------------------------------------------
static int gcd_int(int a, int b) {
  while (b) {
    const int t = a % b;
    a = b;
    b = t;
  }
  return a;
}

// windowed-sinc low-pass: Hann window of half-width num_zeros/(2 cutoff) times sinc
static float lowpass_filter(float t, float cutoff, int num_zeros) {
  const double td = t;
  float window = 0.0f;
  if (std::fabs(td) < num_zeros / (2.0 * cutoff))
    window = static_cast<float>(0.5 * (1 + std::cos(kTwoPi * cutoff / num_zeros * td)));
  const float filter = t != 0.0f ? static_cast<float>(std::sin(kTwoPi * cutoff * td) / (kPi * td))
                                 : static_cast<float>(2.0 * cutoff);
  return filter * window;
}

void make_linear_resample(int rate_in, int rate_out, float cutoff, int num_zeros,
                          LinearResampleHost* r) {
  r->rate_in = rate_in;
  r->rate_out = rate_out;
  r->cutoff = cutoff;
  r->num_zeros = num_zeros;
  const int base = gcd_int(rate_in, rate_out);
  r->in_unit = rate_in / base;
  r->out_unit = rate_out / base;
  r->first.assign(r->out_unit, 0);
  r->ntaps.assign(r->out_unit, 0);
  const double half_width = num_zeros / (2.0 * cutoff);
  std::vector<std::vector<float>> rows(r->out_unit);
  r->max_taps = 0;
  for (int i = 0; i < r->out_unit; ++i) {
    const double t_out = i / static_cast<double>(rate_out);
    const int lo = static_cast<int>(std::ceil((t_out - half_width) * rate_in));
    const int hi = static_cast<int>(std::floor((t_out + half_width) * rate_in));
    r->first[i] = lo;
    r->ntaps[i] = hi - lo + 1;
    rows[i].resize(r->ntaps[i]);
    for (int j = 0; j < r->ntaps[i]; ++j) {
      const double dt = (lo + j) / static_cast<double>(rate_in) - t_out;
      rows[i][j] = lowpass_filter(static_cast<float>(dt), cutoff, num_zeros) / rate_in;
    }
    if (r->ntaps[i] > r->max_taps) r->max_taps = r->ntaps[i];
  }
  r->weights.assign(static_cast<size_t>(r->out_unit) * r->max_taps, 0.0f);
  for (int i = 0; i < r->out_unit; ++i)
    std::memcpy(r->weights.data() + static_cast<size_t>(i) * r->max_taps, rows[i].data(),
                sizeof(float) * rows[i].size());
}

int64_t LinearResampleHost::num_output(int64_t n_in, bool flush) const {
  const int64_t tick_freq = static_cast<int64_t>(rate_in) / gcd_int(rate_in, rate_out) * rate_out;
  const int64_t ticks_per_in = tick_freq / rate_in;
  int64_t interval = n_in * ticks_per_in;
  if (!flush) {
    // [KALDI-UPSTREAM] resample.cc GetNumOutputSamples: BaseFloat window_width; floor(window_width * tick_freq) -
    // a float product (float x int32) rounded to float before the floor
    const float half_width = static_cast<float>(num_zeros / (2.0 * cutoff));
    interval -= static_cast<int>(std::floor(half_width * static_cast<float>(tick_freq)));
  }
  if (interval <= 0) return 0;
  const int64_t ticks_per_out = tick_freq / rate_out;
  int64_t last = interval / ticks_per_out;
  if (last * ticks_per_out == interval) --last;
  return last + 1;
}

int make_pitch_tables(const snf_pitch_options& o, PitchTablesHost* t) {
  if (!(o.samp_freq > 0 && o.resample_freq > 0 && o.lowpass_cutoff > 0 &&
        o.lowpass_cutoff * 2 <= o.samp_freq && o.lowpass_cutoff * 2 <= o.resample_freq &&
        o.lowpass_filter_width > 0 && o.upsample_filter_width > 0 && o.min_f0 > 0 &&
        o.max_f0 > o.min_f0 && o.delta_pitch > 0))
    return set_error(SNF_E_RUNTIME, "bad pitch extraction options");
  if (o.preemph_coeff != 0.0f)
    return set_error(SNF_E_RUNTIME, "pitch preemph_coeff != 0 is not supported");
  // the Viterbi kernels order (cost, index) pairs by the bit pattern of the float cost, which is the float
  // order for non-negative costs only; the costs they compare are fwd[j] >= 0 plus (j - k)^2 times a factor
  // with the sign of penalty_factor (Kaldi's own bounded search also presumes a convex transition cost)
  if (!(o.penalty_factor >= 0.0f))
    return set_error(SNF_E_INVALID, "pitch penalty_factor must be >= 0");
  const double rf = o.resample_freq;
  const double pad = o.upsample_filter_width / (2.0 * rf);
  t->first_lag = static_cast<int>(std::ceil(rf * (1.0 / o.max_f0 - pad)));
  t->last_lag = static_cast<int>(std::floor(rf * (1.0 / o.min_f0 + pad)));
  t->num_lags = t->last_lag + 1 - t->first_lag;
  t->win_size = static_cast<int>(rf * o.frame_length_ms / 1000.0);
  t->win_shift = static_cast<int>(rf * o.frame_shift_ms / 1000.0);
  t->full_len = t->win_size + t->last_lag;
  if (t->win_size <= 0 || t->win_shift <= 0 || t->num_lags <= 0)
    return set_error(SNF_E_RUNTIME, "bad pitch frame options");
  // log-spaced lags (SelectLags)
  t->lags.clear();
  const float min_lag = static_cast<float>(1.0 / o.max_f0), max_lag = static_cast<float>(1.0 / o.min_f0);
  for (float lag = min_lag; lag <= max_lag; lag = static_cast<float>(lag * (1.0 + o.delta_pitch)))
    t->lags.push_back(lag);
  t->num_states = static_cast<int>(t->lags.size());
  // ArbitraryResample(num_lags, resample_freq, resample_freq/2, lags - first_lag/resample_freq, width)
  const float cutoff = static_cast<float>(rf * 0.5);
  const float half_width = static_cast<float>(o.upsample_filter_width / (2.0 * cutoff));
  const float offset = -static_cast<float>(t->first_lag) / o.resample_freq;
  t->ar_first.assign(t->num_states, 0);
  t->ar_n.assign(t->num_states, 0);
  std::vector<std::vector<float>> rows(t->num_states);
  t->max_taps = 0;
  for (int i = 0; i < t->num_states; ++i) {
    const float tp = t->lags[i] + offset;
    const float t_min = tp - half_width, t_max = tp + half_width;
    int lo = static_cast<int>(std::ceil(static_cast<double>(o.resample_freq * t_min)));
    int hi = static_cast<int>(std::floor(static_cast<double>(o.resample_freq * t_max)));
    if (lo < 0) lo = 0;
    if (hi >= t->num_lags) hi = t->num_lags - 1;
    t->ar_first[i] = lo;
    t->ar_n[i] = hi - lo + 1;
    rows[i].resize(t->ar_n[i] > 0 ? t->ar_n[i] : 0);
    for (int j = 0; j < t->ar_n[i]; ++j) {
      const float dt = tp - static_cast<float>(lo + j) / o.resample_freq;
      rows[i][j] = lowpass_filter(dt, cutoff, o.upsample_filter_width) / o.resample_freq;
    }
    if (t->ar_n[i] > t->max_taps) t->max_taps = t->ar_n[i];
  }
  t->ar_w.assign(static_cast<size_t>(t->num_states) * t->max_taps, 0.0f);
  for (int i = 0; i < t->num_states; ++i)
    if (!rows[i].empty())
      std::memcpy(t->ar_w.data() + static_cast<size_t>(i) * t->max_taps, rows[i].data(),
                  sizeof(float) * rows[i].size());
  make_linear_resample(static_cast<int>(o.samp_freq), static_cast<int>(o.resample_freq),
                       o.lowpass_cutoff, o.lowpass_filter_width, &t->resample);
  return SNF_OK;
}

int64_t PitchTablesHost::frames_available(int64_t n_down, bool input_finished,
                                          bool snip_edges) const {
  const int64_t shift = win_shift;
  int64_t len = win_size;
  if (!input_finished) len += last_lag;
  if (n_down < len) return 0;
  if (!snip_edges) {
    if (input_finished) return static_cast<int64_t>(n_down * 1.0f / shift + 0.5f);
    return static_cast<int64_t>((n_down - len / 2) * 1.0f / shift + 0.5f);
  }
  return (n_down - len) / shift + 1;
}

}  // namespace snf
