// A/B harness for the 512-point kernels through the C ABI (no Python): the benchmark workload
// (N utterances x 3 s of synthetic int16 audio), every variant selected by environment variables that
// the launcher reads at every call, timed with the library's own HIP events and compared with the
// output of the round-2 kernel (SNF_FBANK512_OLD=1) element by element.
//
//   g++ -O2 -std=c++17 tools/ab_fbank512.cpp -Iinclude -Lshennong_amd -lshennong_hip \
//       -Wl,-rpath,'$ORIGIN/../shennong_amd' -o scratch/ab512
//   scratch/ab512 [n_utts] [kind: fbank|mfcc|spec|plp] [reps] -- name=ENV1=v,ENV2=v ...
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "shennong_amd.h"

static uint64_t rng_state = 0x9E3779B97F4A7C15ull;
static inline uint32_t rnd() {
  rng_state ^= rng_state << 13;
  rng_state ^= rng_state >> 7;
  rng_state ^= rng_state << 17;
  return static_cast<uint32_t>(rng_state >> 32);
}
static inline double gauss() {
  const double u1 = (rnd() + 1.0) / 4294967297.0, u2 = rnd() / 4294967296.0;
  return std::sqrt(-2.0 * std::log(u1)) * std::cos(6.283185307179586 * u2);
}

#define CK(x)                                                          \
  do {                                                                 \
    int rc_ = (x);                                                     \
    if (rc_ != 0) {                                                    \
      printf("%s -> %d: %s\n", #x, rc_, snf_last_error());             \
      return 1;                                                        \
    }                                                                  \
  } while (0)

static void apply_env(const std::string& spec, bool set) {
  // spec: name=ENV1=v,ENV2=v   (the first field is the label)
  size_t pos = spec.find('=');
  if (pos == std::string::npos) return;
  std::string rest = spec.substr(pos + 1);
  while (!rest.empty()) {
    const size_t comma = rest.find(',');
    const std::string item = rest.substr(0, comma);
    const size_t eq = item.find('=');
    if (eq != std::string::npos) {
      if (set) setenv(item.substr(0, eq).c_str(), item.substr(eq + 1).c_str(), 1);
      else unsetenv(item.substr(0, eq).c_str());
    }
    if (comma == std::string::npos) break;
    rest = rest.substr(comma + 1);
  }
}

int main(int argc, char** argv) {
  int64_t n_utts = 10000;
  std::string kind = "fbank";
  int reps = 12;
  std::vector<std::string> variants;
  int i = 1;
  for (; i < argc && std::strcmp(argv[i], "--") != 0; ++i) {
    if (i == 1) n_utts = atoll(argv[i]);
    if (i == 2) kind = argv[i];
    if (i == 3) reps = atoi(argv[i]);
  }
  for (++i; i < argc; ++i) variants.push_back(argv[i]);
  if (variants.empty()) variants.push_back("default=");

  snf_options o;
  std::memset(&o, 0, sizeof(o));
  o.frame.samp_freq = 16000;
  o.frame.frame_shift_ms = 10;
  o.frame.frame_length_ms = 25;
  o.frame.dither = 0;
  o.frame.preemph_coeff = 0.97f;
  o.frame.remove_dc_offset = 1;
  o.frame.window_type = SNF_WINDOW_POVEY;
  o.frame.round_to_power_of_two = 1;
  o.frame.blackman_coeff = 0.42f;
  o.frame.snip_edges = 1;
  o.mel.num_bins = 40;
  o.mel.low_freq = 20;
  o.mel.high_freq = 0;
  o.mel.vtln_low = 100;
  o.mel.vtln_high = -500;
  o.energy_floor = 0;
  o.raw_energy = 1;
  o.use_log_fbank = 1;
  o.use_power = 1;
  o.kind = SNF_KIND_FBANK;
  if (kind == "mfcc") {
    o.kind = SNF_KIND_MFCC;
    o.mel.num_bins = 23;
    o.num_ceps = 13;
    o.cepstral_lifter = 22;
    o.use_energy = 1;
  } else if (kind == "spec") {
    o.kind = SNF_KIND_SPECTROGRAM;
  } else if (kind == "plp") {
    o.kind = SNF_KIND_PLP;
    o.mel.num_bins = 23;
    o.num_ceps = 13;
    o.cepstral_lifter = 22;
    o.use_energy = 1;
    o.lpc_order = 12;
    o.compress_factor = 1.0f / 3.0f;
    o.cepstral_scale = 1.0f;
  }
  o.delta_order = 2;
  o.delta_window = 2;

  snf_plan* plan = nullptr;
  CK(snf_plan_create(&o, 0, &plan));
  const int cols = snf_plan_ndims(plan);
  // ragged tail: a few utterances of other lengths so that sets straddle utterance boundaries
  std::vector<int64_t> soff(n_utts + 1, 0), foff(n_utts + 1, 0);
  for (int64_t u = 0; u < n_utts; ++u) {
    int64_t n = 48000;
    if (!getenv("AB_UNIFORM")) {   // (AB_UNIFORM=1: the benchmark's own batch, every utterance 3 s)
      if (u % 97 == 5) n = 16000 + 37 * (u % 1000);
      if (u % 211 == 7) n = 399;  // shorter than a window: no frame
    }
    soff[u + 1] = soff[u] + n;
    foff[u + 1] = foff[u] + snf_plan_num_frames(plan, n);
  }
  const int64_t total_samples = soff[n_utts], total_frames = foff[n_utts];
  std::vector<int16_t> wave(static_cast<size_t>(total_samples));
  {
    // harmonic source + noise, generated once for 64 utterances and tiled with a per-utterance shift
    std::vector<int16_t> proto(64 * 48000);
    for (int u = 0; u < 64; ++u) {
      const double f0 = 80 + 220.0 * (rnd() / 4294967296.0);
      for (int t = 0; t < 48000; ++t) {
        double v = 3000.0 * gauss();
        for (int h = 1; h <= 5; ++h) v += 8000.0 * std::sin(6.283185307179586 * h * f0 * t / 16000.0) / h;
        v = std::max(-32767.0, std::min(32767.0, std::nearbyint(v)));
        proto[static_cast<size_t>(u) * 48000 + t] = static_cast<int16_t>(v);
      }
    }
    for (int64_t u = 0; u < n_utts; ++u) {
      const int64_t n = soff[u + 1] - soff[u];
      const size_t src = static_cast<size_t>(u % 64) * 48000;
      std::memcpy(&wave[soff[u]], &proto[src], static_cast<size_t>(n) * 2);
    }
  }
  void *d_wave = nullptr, *d_out = nullptr;
  CK(snf_malloc(&d_wave, static_cast<uint64_t>(total_samples) * 2));
  CK(snf_malloc(&d_out, static_cast<uint64_t>(total_frames) * cols * 4));
  CK(snf_memcpy_h2d(d_wave, wave.data(), static_cast<uint64_t>(total_samples) * 2));
  printf("workload: %lld utterances, %lld frames, %d columns, kind %s\n", (long long)n_utts, (long long)total_frames,
         cols, kind.c_str());

  // reference: the round-2 kernel
  std::vector<float> ref(static_cast<size_t>(total_frames) * cols), got(ref.size());
  setenv("SNF_FBANK512_OLD", "1", 1);
  CK(snf_memset(d_out, 0xff, static_cast<uint64_t>(total_frames) * cols * 4));
  CK(snf_plan_run_batch_device(plan, static_cast<const int16_t*>(d_wave), soff.data(), n_utts, nullptr,
                               static_cast<float*>(d_out), foff.data(), nullptr));
  CK(snf_memcpy_d2h(ref.data(), d_out, ref.size() * 4));
  unsetenv("SNF_FBANK512_OLD");

  for (const std::string& v : variants) {
    const std::string label = v.substr(0, v.find('='));
    apply_env(v, true);
    snf_debug_fill_lds(0xFFFFFFFFu);
    CK(snf_memset(d_out, 0xff, static_cast<uint64_t>(total_frames) * cols * 4));
    int rc = snf_plan_run_batch_device(plan, static_cast<const int16_t*>(d_wave), soff.data(), n_utts, nullptr,
                                       static_cast<float*>(d_out), foff.data(), nullptr);
    if (rc != 0) {
      printf("%-28s FAILED: %s\n", label.c_str(), snf_last_error());
      apply_env(v, false);
      continue;
    }
    CK(snf_memcpy_d2h(got.data(), d_out, got.size() * 4));
    double max_abs = 0.0;
    size_t n_diff = 0, n_nan = 0, worst = 0;
    for (size_t k = 0; k < got.size(); ++k) {
      if (std::memcmp(&got[k], &ref[k], 4) != 0) ++n_diff;
      if (std::isnan(got[k])) {
        ++n_nan;
        continue;
      }
      const double d = std::fabs(static_cast<double>(got[k]) - ref[k]);
      if (d > max_abs) {
        max_abs = d;
        worst = k;
      }
    }
    if (getenv("SNF_FBANK512B_ABL") && (atoi(getenv("SNF_FBANK512B_ABL")) & 1024)) {
      // timing experiment: s_memtime stamps of one wave's 20th iteration in the first output row
      long long st[10];
      std::memcpy(st, got.data(), sizeof(st));
      printf("%-28s stamps (clocks since the top of the iteration):", label.c_str());
      for (int k = 1; k < 10; ++k) printf(" %lld", st[k] - st[0]);
      printf("\n");
    }
    std::vector<float> ms;
    for (int r = 0; r < reps; ++r) {
      CK(snf_plan_run_batch_device(plan, static_cast<const int16_t*>(d_wave), soff.data(), n_utts, nullptr,
                                   static_cast<float*>(d_out), foff.data(), nullptr));
      ms.push_back(snf_plan_last_kernel_ms(plan, 1));
    }
    std::sort(ms.begin(), ms.end());
    printf("%-28s kernel ms min %.4f med %.4f max %.4f | vs round-2 kernel: %zu of %zu values differ bitwise, "
           "max abs %.3g (row %zu col %zu: %g vs %g), NaN %zu\n",
           label.c_str(), ms.front(), ms[ms.size() / 2], ms.back(), n_diff, got.size(), max_abs, worst / cols,
           worst % cols, got[worst], ref[worst], n_nan);
    fflush(stdout);
    apply_env(v, false);
  }
  snf_free(d_wave);
  snf_free(d_out);
  snf_plan_destroy(plan);
  return 0;
}
