// Native ingest of PCM WAV files ("int16 wav ingest"): the samples of many files - or of
// stretches of them - read side by side straight into one caller-provided int16 block, e.g. the page-locked block an
// upload starts from.  The reference decodes one file per joblib job in Python (shennong/audio.py:243-286, scipy /
// sox) and forces the signal to int16 before Kaldi sees it (processor/base.py:428); at GPU rates that loop is what
// a corpus on disk waits for (46 us per cached 3 s file against 5 us of pipeline).  Handled here: RIFF / WAVE,
// format tag 1 (PCM) or the extensible form of it, 16 bits, one channel - what speech corpora are stored in.
// Anything else is reported per file (status 1) and stays with the Python reader, which knows every sample type.
#include <fcntl.h>
#include <sys/stat.h>
#include <unistd.h>

#include <atomic>
#include <cstdint>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "snf_internal.h"

namespace {

struct WavInfo {
  int32_t channels = 0, rate = 0, bits = 0, tag = 0;
  int64_t data_offset = 0, data_bytes = 0;
};

uint32_t le32(const unsigned char* p) { return p[0] | (p[1] << 8) | (p[2] << 16) | (static_cast<uint32_t>(p[3]) << 24); }
uint16_t le16(const unsigned char* p) { return static_cast<uint16_t>(p[0] | (p[1] << 8)); }

// 0 = parsed, 1 = not a RIFF / WAVE file this reader understands, 2 = I/O error
int parse_wav(int fd, WavInfo* w) {
  struct stat st;
  if (fstat(fd, &st) != 0) return 2;
  const int64_t size = st.st_size;
  unsigned char head[12];
  if (pread(fd, head, 12, 0) != 12) return 1;
  if (std::memcmp(head, "RIFF", 4) != 0 || std::memcmp(head + 8, "WAVE", 4) != 0) return 1;
  int64_t pos = 12;
  bool have_fmt = false;
  while (pos + 8 <= size) {
    unsigned char ck[8];
    if (pread(fd, ck, 8, pos) != 8) return 2;
    const int64_t len = le32(ck + 4);
    if (std::memcmp(ck, "fmt ", 4) == 0) {
      unsigned char f[40];
      const int64_t want = len < 40 ? len : 40;
      if (want < 16 || pread(fd, f, static_cast<size_t>(want), pos + 8) != want) return 1;
      w->tag = le16(f);
      w->channels = le16(f + 2);
      w->rate = static_cast<int32_t>(le32(f + 4));
      w->bits = le16(f + 14);
      if (w->tag == 0xFFFE && want >= 26) w->tag = le16(f + 24);  // WAVE_FORMAT_EXTENSIBLE: the sub-format's tag
      have_fmt = true;
    } else if (std::memcmp(ck, "data", 4) == 0) {
      if (!have_fmt) return 1;
      w->data_offset = pos + 8;
      // (a streamed file may say 0 or 0xFFFFFFFF: the data then runs to the end of the file)
      w->data_bytes = (len == 0 || len == 0xFFFFFFFFll || pos + 8 + len > size) ? size - (pos + 8) : len;
      return 0;
    }
    pos += 8 + len + (len & 1);  // chunks are word aligned
  }
  return 1;
}

}  // namespace

extern "C" {

int snf_wav_scan(const char* path, int32_t* channels, int32_t* sample_rate, int64_t* nsamples, int32_t* bits,
                 int32_t* format_tag) {
  if (!path) return snf::set_error(SNF_E_INVALID, "null path");
  const int fd = open(path, O_RDONLY | O_CLOEXEC);
  if (fd < 0) return snf::set_error(SNF_E_INVALID, std::string(path) + ": file not found");
  WavInfo w;
  const int rc = parse_wav(fd, &w);
  close(fd);
  if (rc != 0 || w.channels <= 0 || w.bits <= 0)
    return snf::set_error(SNF_E_INVALID, std::string(path) + ": not a RIFF / WAVE file");
  if (channels) *channels = w.channels;
  if (sample_rate) *sample_rate = w.rate;
  if (bits) *bits = w.bits;
  if (format_tag) *format_tag = w.tag;
  if (nsamples) *nsamples = w.data_bytes / (static_cast<int64_t>(w.channels) * (w.bits / 8 > 0 ? w.bits / 8 : 1));
  return SNF_OK;
}

int snf_wav_read_pcm16(const char* const* paths, int64_t n_files, const int64_t* first_sample,
                       const int64_t* n_samples, int16_t* dst, const int64_t* dst_offsets, int32_t threads,
                       int32_t* status) {
  if (n_files < 0) return snf::set_error(SNF_E_INVALID, "n_files < 0");
  if (n_files == 0) return SNF_OK;
  if (!paths || !first_sample || !n_samples || !dst || !dst_offsets || !status)
    return snf::set_error(SNF_E_INVALID, "null argument");
  std::atomic<int64_t> next{0};
  auto work = [&]() {
    for (;;) {
      const int64_t i = next.fetch_add(1);
      if (i >= n_files) return;
      status[i] = 2;
      if (!paths[i] || first_sample[i] < 0 || n_samples[i] < 0) continue;
      const int fd = open(paths[i], O_RDONLY | O_CLOEXEC);
      if (fd < 0) continue;
      WavInfo w;
      const int rc = parse_wav(fd, &w);
      if (rc != 0 || w.tag != 1 || w.bits != 16 || w.channels != 1) {
        status[i] = rc == 2 ? 2 : 1;   // another sample type or layout: the caller's general reader takes it
        close(fd);
        continue;
      }
      const int64_t have = w.data_bytes / 2;
      if (first_sample[i] + n_samples[i] > have) {   // (the caller sized its block from a header that has changed)
        status[i] = 3;
        close(fd);
        continue;
      }
      char* out = reinterpret_cast<char*>(dst + dst_offsets[i]);
      int64_t left = 2 * n_samples[i], at = w.data_offset + 2 * first_sample[i];
      bool ok = true;
      while (left > 0) {
        const ssize_t got = pread(fd, out, static_cast<size_t>(left), at);
        if (got <= 0) {
          ok = false;
          break;
        }
        out += got;
        at += got;
        left -= got;
      }
      close(fd);
      status[i] = ok ? 0 : 2;
    }
  };
  int count = threads < 1 ? 1 : (threads > 64 ? 64 : threads);
  if (count > n_files) count = static_cast<int>(n_files);
  std::vector<std::thread> pool;
  for (int t = 1; t < count; ++t) pool.emplace_back(work);
  work();
  for (auto& t : pool) t.join();
  return SNF_OK;
}

}  // extern "C"
