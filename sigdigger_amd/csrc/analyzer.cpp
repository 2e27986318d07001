// analyzer.cpp -- live path behind the suscan_analyzer_* C ABI (include/suscan_amd.h): message
// queue, source reader, analyzer worker thread and per-inspector chains.  All arithmetic is
// delegated to the GPU entry points of include/sigdigger_amd.h; this file only moves blocks
// (source -> pinned host -> HBM), sequences the calls and packages results as malloc'd messages
// with the reference's ownership protocol.
//
// Worker loop per block (what libsuscan's source worker does):
//   pending requests -> read L samples -> PSD (every window, Welch average -> one psd_msg per
//   psd_update_int) -> every open inspector: channel bank -> [AGC] -> [Costas | quad demod] ->
//   [Gardner] -> sample_batch_msg.
#include <hip/hip_runtime.h>

#include <atomic>
#include <chrono>
#include <cmath>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <strings.h>
#include <dlfcn.h>
#include <deque>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <type_traits>
#include <vector>

#include "../../include/suscan_amd.h"
#include "analyzer_internal.hpp"
#include "tuning.hpp"

// a gang call's rows must lie within this many bytes of each other (32-bit lane offsets): one slot's three slabs do, up to here
#ifndef SUAMD_SLAB_NEAR_BYTES
#define SUAMD_SLAB_NEAR_BYTES ((size_t)1 << 32)
#endif

using suan::cfg_get;
using suan::dupstr;


namespace {

// source reader: IQ file (raw float32 / u8 / s8 / s16, or a WAV / SigMF container around one of
// them -- Default/SourceConfig/FileSourcePage.cpp:80-104), or a tone generator ("tonegen": params
// signal / noise in dB, Default/SourceConfig/ToneGenSourcePage.cpp:81-90,125-126).  Compact formats
// stay compact across PCIe: read() returns raw bytes, the worker expands them with suamd_ingest_iq.
struct Source {
  suscan_source_config cfg;
  FILE *fp = nullptr;
  uint64_t n = 0;
  uint32_t lcg = 12345u;
  int raw_format = SUAMD_FORMAT_RAW_FLOAT32;   // payload format after container / AUTO resolution
  long data_start = 0;                         // payload offset (WAV header)
  long data_bytes = -1;                        // payload length, -1 = to end of file

  static bool ends_with(const std::string &s, const char *suf)
  {
    const size_t l = std::strlen(suf);
    return s.size() >= l && strcasecmp(s.c_str() + s.size() - l, suf) == 0;
  }
  static uint32_t le32(const unsigned char *b) { return b[0] | (b[1] << 8) | (b[2] << 16) | ((uint32_t)b[3] << 24); }

  bool open_wav(std::string &err)
  {
    unsigned char h[12];
    if (std::fread(h, 1, 12, fp) != 12 || std::memcmp(h, "RIFF", 4) || std::memcmp(h + 8, "WAVE", 4)) { err = "not a RIFF/WAVE file"; return false; }
    bool have_fmt = false;
    for (;;) {
      unsigned char c[8];
      if (std::fread(c, 1, 8, fp) != 8) { err = "WAV: no data chunk"; return false; }
      const uint32_t len = le32(c + 4);
      if (!std::memcmp(c, "fmt ", 4)) {
        unsigned char f[16];
        if (len < 16 || std::fread(f, 1, 16, fp) != 16) { err = "WAV: short fmt chunk"; return false; }
        const unsigned tag = f[0] | (f[1] << 8), ch = f[2] | (f[3] << 8), bits = f[14] | (f[15] << 8);
        const uint32_t rate = le32(f + 4);
        if (ch != 2) { err = "WAV: need 2 channels (I/Q)"; return false; }
        if (tag == 1 && bits == 8) raw_format = SUAMD_FORMAT_RAW_UNSIGNED8;
        else if (tag == 1 && bits == 16) raw_format = SUAMD_FORMAT_RAW_SIGNED16;
        else if (tag == 3 && bits == 32) raw_format = SUAMD_FORMAT_RAW_FLOAT32;
        else { err = "WAV: unsupported sample type"; return false; }
        if (rate) cfg.samp_rate = rate;
        have_fmt = true;
        std::fseek(fp, (long)(len - 16 + (len & 1)), SEEK_CUR);
      } else if (!std::memcmp(c, "data", 4)) {
        if (!have_fmt) { err = "WAV: data before fmt"; return false; }
        data_start = std::ftell(fp);
        data_bytes = len == 0xffffffffu ? -1 : (long)len;
        return true;
      } else {
        std::fseek(fp, (long)(len + (len & 1)), SEEK_CUR);
      }
    }
  }

  // SigMF: datatype (and sample rate) from the .sigmf-meta next to the .sigmf-data
  bool open_sigmf(std::string &err)
  {
    std::string base = cfg.path;
    const size_t dot = base.rfind('.');
    if (dot != std::string::npos) base.resize(dot);
    const std::string meta = base + ".sigmf-meta", data = base + ".sigmf-data";
    FILE *m = std::fopen(meta.c_str(), "rb");
    if (!m) { err = "cannot open " + meta; return false; }
    std::string js;
    char buf[4096]; size_t r;
    while ((r = std::fread(buf, 1, sizeof buf, m)) > 0) js.append(buf, r);
    std::fclose(m);
    auto value_of = [&](const char *key) -> std::string {
      size_t k = js.find(key);
      if (k == std::string::npos) return "";
      k = js.find(':', k + std::strlen(key));
      if (k == std::string::npos) return "";
      size_t b = js.find_first_not_of(" \t\r\n\"", k + 1);
      if (b == std::string::npos) return "";               // the meta file ends right after the ':'
      size_t e = js.find_first_of(",\"}\r\n", b);
      return js.substr(b, e == std::string::npos ? e : e - b);
    };
    const std::string dt = value_of("\"core:datatype\"");
    if (dt == "cf32_le" || dt == "cf32") raw_format = SUAMD_FORMAT_RAW_FLOAT32;
    else if (dt == "ci16_le" || dt == "ci16") raw_format = SUAMD_FORMAT_RAW_SIGNED16;
    else if (dt == "ci8") raw_format = SUAMD_FORMAT_RAW_SIGNED8;
    else if (dt == "cu8") raw_format = SUAMD_FORMAT_RAW_UNSIGNED8;
    else { err = "SigMF: unsupported core:datatype '" + dt + "'"; return false; }
    const std::string sr = value_of("\"core:sample_rate\"");
    if (!sr.empty() && std::atof(sr.c_str()) >= 1) cfg.samp_rate = (unsigned)std::atof(sr.c_str());
    if (fp) std::fclose(fp);
    fp = std::fopen(data.c_str(), "rb");
    if (!fp) { err = "cannot open " + data; return false; }
    return true;
  }

  bool open(std::string &err)
  {
    if (cfg.type == "tonegen") return true;
    if (cfg.type != "file") { err = "unsupported source type '" + cfg.type + "' (file, tonegen)"; return false; }
    fp = std::fopen(cfg.path.c_str(), "rb");
    if (!fp) { err = "cannot open " + cfg.path; return false; }
    int f = cfg.format;
    if (f == SUSCAN_SOURCE_FORMAT_AUTO) {                  // by extension, as the file dialog filters do
      const std::string &p = cfg.path;
      if (ends_with(p, ".wav")) f = SUSCAN_SOURCE_FORMAT_WAV;
      else if (ends_with(p, ".sigmf-data") || ends_with(p, ".sigmf-meta")) f = SUSCAN_SOURCE_FORMAT_SIGMF;
      else if (ends_with(p, ".u8") || ends_with(p, ".cu8")) f = SUSCAN_SOURCE_FORMAT_RAW_UNSIGNED8;
      else if (ends_with(p, ".s8") || ends_with(p, ".cs8")) f = SUSCAN_SOURCE_FORMAT_RAW_SIGNED8;
      else if (ends_with(p, ".s16") || ends_with(p, ".cs16")) f = SUSCAN_SOURCE_FORMAT_RAW_SIGNED16;
      else f = SUSCAN_SOURCE_FORMAT_RAW_FLOAT32;
    }
    switch (f) {
      case SUSCAN_SOURCE_FORMAT_RAW_FLOAT32:   raw_format = SUAMD_FORMAT_RAW_FLOAT32; return true;
      case SUSCAN_SOURCE_FORMAT_RAW_UNSIGNED8: raw_format = SUAMD_FORMAT_RAW_UNSIGNED8; return true;
      case SUSCAN_SOURCE_FORMAT_RAW_SIGNED8:   raw_format = SUAMD_FORMAT_RAW_SIGNED8; return true;
      case SUSCAN_SOURCE_FORMAT_RAW_SIGNED16:  raw_format = SUAMD_FORMAT_RAW_SIGNED16; return true;
      case SUSCAN_SOURCE_FORMAT_WAV:           return open_wav(err);
      case SUSCAN_SOURCE_FORMAT_SIGMF:         return open_sigmf(err);
      default: err = "unsupported sample format"; return false;
    }
  }
  size_t bytes_per_sample() const { return cfg.type == "tonegen" ? 8 : suamd_format_bytes_per_sample(raw_format); }

  // the worker reads one block ahead while the GPU works; when a request changes the block size the
  // prefetched block is pushed back
  long mark_pos = 0; uint64_t mark_n = 0; uint32_t mark_lcg = 0;
  void mark() { mark_pos = fp ? std::ftell(fp) : 0; mark_n = n; mark_lcg = lcg; }
  void rewind_to_mark() { if (fp) std::fseek(fp, mark_pos, SEEK_SET); n = mark_n; lcg = mark_lcg; }

  // Suscan::Analyzer::seek: the next read starts at sample `pos` (clamped to the payload)
  void seek(uint64_t pos)
  {
    if (fp) {
      const size_t bps = bytes_per_sample();
      if (data_bytes >= 0 && pos * bps > (uint64_t)data_bytes) pos = (uint64_t)data_bytes / bps;
      std::fseek(fp, data_start + (long)(pos * bps), SEEK_SET);
    }
    n = pos;
  }

  // fills dst with `want` samples in the payload format (bytes_per_sample() each); returns the
  // samples read (< want only at end of stream)
  size_t read(void *dst, size_t want, bool *looped)
  {
    if (cfg.type == "tonegen") {
      suamd_complex *o = static_cast<suamd_complex *>(dst);
      const double sig = std::pow(10.0, std::atof(cfg.params.count("signal") ? cfg.params["signal"].c_str() : "0") / 20);
      const double noi = std::pow(10.0, std::atof(cfg.params.count("noise") ? cfg.params["noise"].c_str() : "-40") / 20);
      const double w = 2 * M_PI * 0.05;
      for (size_t i = 0; i < want; ++i, ++n) {
        lcg = lcg * 1664525u + 1013904223u; const float a = ((lcg >> 8) & 0xffff) / 32768.0f - 1.0f;
        lcg = lcg * 1664525u + 1013904223u; const float b = ((lcg >> 8) & 0xffff) / 32768.0f - 1.0f;
        o[i].re = (float)(sig * std::cos(w * (double)n) + noi * a);
        o[i].im = (float)(sig * std::sin(w * (double)n) + noi * b);
      }
      return want;
    }
    const size_t bps = bytes_per_sample();
    unsigned char *o = static_cast<unsigned char *>(dst);
    size_t got = 0;
    while (got < want) {
      size_t ask = want - got;
      if (data_bytes >= 0) {
        const long left = data_start + data_bytes - std::ftell(fp);
        if ((long)(ask * bps) > left) ask = left > 0 ? (size_t)left / bps : 0;
      }
      const size_t r = ask ? std::fread(o + got * bps, bps, ask, fp) : 0;
      got += r;
      if (got < want) {
        if (cfg.loop && std::ftell(fp) > data_start) { std::fseek(fp, data_start, SEEK_SET); if (looped) *looped = true; continue; }
        break;
      }
    }
    n += got;
    return got;
  }
  ~Source() { if (fp) std::fclose(fp); }
};

// The inspectors' sample rows (channel samples, gain-controlled, carrier-corrected, symbols: four per slot and inspector) come
// out of slabs of one allocation each, per shard: the narrow-channel kernels address every row of a launch with 32-bit offsets
// from the lowest one when they all start within ~1.75 GiB (suamd_specttuner_feed_rows_near), and rows that are hipMalloc'ed
// one by one are only that close while nothing else allocates on the device (bench.py's live64 line behind 16 Mi-sample
// pipelines, eight shards on one device: spans beyond 2 GiB, the 64-bit kernels, 2.9 instead of 2.0 ms per block).  Freed rows
// go to a free list by size (an analyzer's inspectors use a handful of sizes); slabs go back to the device with the shard.
struct RowArena {
  struct Slab { char *base; size_t size, head; };
  std::vector<Slab> slabs;
  std::map<size_t, std::vector<void *>> spare;
  // Rows come in size classes -- eight per octave (>= 4 KiB; at most 12.5 % over the request: the rows of hundreds of
  // inspectors must stay within the 1.75 GiB span of suamd_specttuner_feed_rows_near) --: a freed row serves any later request
  // of its class, so an inspector that is resized back and forth (set_bandwidth: another decimation, another row length)
  // reuses what it gave back instead of carving the slabs further.  The first slab is sized from the first request (32 MiB or 8 rows), later ones grow
  // geometrically up to 256 MiB: one raw inspector on a small block does not reserve 256 MiB per shard any more.
  // Memory floor: slabs are only returned when the analyzer is destroyed (release()).
  static size_t rounded(size_t bytes)
  {
    if (bytes <= 4096) return 4096;
    size_t p = 4096;
    while ((p << 1) <= bytes) p <<= 1;                        // p <= bytes < 2 p
    const size_t g = p >> 3;
    return (bytes + g - 1) / g * g;
  }
  void *take(size_t bytes)
  {
    bytes = rounded(bytes);
    auto it = spare.find(bytes);
    if (it != spare.end() && !it->second.empty()) { void *p = it->second.back(); it->second.pop_back(); return p; }
    for (Slab &sl : slabs)
      if (sl.head + bytes <= sl.size) { void *p = sl.base + sl.head; sl.head += bytes; return p; }
    const size_t grow = slabs.empty() ? ((size_t)32 << 20) : std::min<size_t>((size_t)256 << 20, 2 * slabs.back().size);
    const size_t want = std::max<size_t>(grow, slabs.empty() ? 8 * bytes : 4 * bytes);
    char *base = nullptr;
    if (hipMalloc((void **)&base, want) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    slabs.push_back(Slab{base, want, bytes});
    return base;
  }
  // (the caller has made sure nothing in flight still touches the row: free_rows synchronises, as hipFree did)
  void give(void *p, size_t bytes) { if (p) spare[rounded(bytes)].push_back(p); }
  void release()
  {
    for (Slab &sl : slabs) (void)hipFree(sl.base);
    slabs.clear(); spare.clear();
  }
};

struct Inspector {
  SUHANDLE handle;
  RowArena *arena = nullptr;                  // the shard's row slabs (null: plain allocations)
  uint32_t inspector_id = 0;
  std::string cls;
  struct sigutils_channel channel;
  suscan_config_t *config = nullptr;
  unsigned D = 1;
  double equiv_fs = 0, fnor = 0;
  SUSCOUNT watermark = 0;                      // > 0: SAMPLES batches of exactly this many samples (Suscan/Analyzer.cpp:528-537)
  std::vector<suamd_complex> wm_buf;           // what has not filled a batch yet (flushed at EOS / close)
  bool dirty = true;                          // chain must be (re)built
  suamd_chanbank_t *bank = nullptr;           // channeliser "fir": translate + 255-tap low-pass + decimate (SPEC.md C)
  suamd_specttuner_t *st = nullptr;           // channeliser "fft": a channel of the analyzer's su_specttuner (SPEC.md C2)
  int st_chan = -1;
  double st_f0 = 0, st_bw = 0, st_guard = 0; bool st_precise = false;   // what the open channel was opened with
  bool precise = false;                       // open_ex_async's flag: the FFT channel corrects its centre-bin rounding
  suamd_agc_bank_t *agc = nullptr;
  suamd_costas_bank_t *costas = nullptr;
  suamd_clock_bank_t *clock = nullptr;
  suamd_nco_bank_t *nco = nullptr;            // afc.costas-order = 0 with afc.offset
  suamd_pll_bank_t *pll = nullptr;            // ask.use-pll
  suamd_fir_bank_t *mf = nullptr;             // mf.type = MANUAL
  suamd_cma_bank_t *cma = nullptr;            // equalizer.type = CMA
  float fixed_gain = 0;                       // agc.enabled = false: linear agc.gain (0 = none)
  uint32_t spectsrc_id = 0;                   // 0 = none (Suscan/Analyzer.cpp:539-547)
  suamd_power_bank_t *power = nullptr;         // class "power"
  suamd_audio_t *audio = nullptr;              // class "audio"
  static constexpr int NEST = 3;
  suamd_baud_estimator_t *est[NEST] = {nullptr, nullptr, nullptr};   // "baud-fac", "baud-nonlinear", "carrier" (estimator_list of the OPEN message)
  bool est_on[NEST] = {false, false, false}, est_fed[NEST] = {false, false, false};
  SUFLOAT last_est[NEST] = {0, 0, 0};                           // a block too short for the analysis window repeats the last estimate
  suamd_psd_t *spect_psd = nullptr;           // spectrum of the channel samples, one frame set per block
  unsigned spect_n = 0;
  suamd_complex *d_spre = nullptr;            // transformed samples
  float *d_spec = nullptr;
  bool spect_have_prev = false;               // the sample before the block: the other slot's last channel sample
  int last_slot = 0; SUSCOUNT last_fir_m = 0;
  // all inspectors work on the analyzer's inspector stream, stage by stage (enqueue_inspectors); results
  // land in pinned memory and become messages after one synchronisation (collect_inspectors)
  hipStream_t stream = nullptr;
  struct Pinned { uint32_t count; float est[NEST]; float spec[8192]; } *pin = nullptr;   // D2H landing zone
  suamd_complex *h_out = nullptr;             // samples / symbols of the block, written by the device (mapped, cap long)
  SUSCOUNT pend_m = 0;                        // channel samples of the block in flight
  bool pend_samples = false, pend_spectrum = false, pend_symbols = false;
  const suamd_complex *pend_src = nullptr;
  unsigned pend_spec_n = 0;
  bool quad = false, first = true;
  suamd_complex *d_y = nullptr, *d_a = nullptr, *d_z = nullptr, *d_sym = nullptr, *d_prev = nullptr;
  uint32_t *d_count = nullptr;
  size_t cap = 0;
  // The narrow channels of the FFT filter bank (<= 64 bins: the kernels that store one channel per lane) leave the
  // channeliser as COLUMNS of the shard's time-major slab (suscan_analyzer::slab; column = lane = the channel's index in the
  // tuner).  in_slab: the chain lives there too -- d_y / d_a / d_z are columns of the three slabs, element stride ts = the
  // pitch, and the gangs stream them where they lie (suamd_*_gang_*_slab).  copy_out: the class needs contiguous rows
  // throughout (audio, power): the column is copied into the inspector's own row right behind the channeliser.
  bool in_slab = false, copy_out = false;
  int lane = -1;
  size_t ts = 1;                              // element stride of d_y / d_a / d_z
  suamd_complex *d_lin = nullptr;             // in_slab: the block's channel samples as a contiguous row, when a consumer needs one (spectrum, estimators)
  // Two blocks are in flight at a time (block k+1 is enqueued before block k's results are collected), so every
  // per-block buffer and every "pending" note exists twice; the names above are the aliases of the slot in use.
  struct Slot {
    suamd_complex *d_y = nullptr, *d_a = nullptr, *d_z = nullptr, *d_sym = nullptr, *h_out = nullptr, *d_lin = nullptr;
    uint32_t *d_count = nullptr;
    Pinned *pin = nullptr;
    SUSCOUNT pend_m = 0;
    bool pend_samples = false, pend_spectrum = false, pend_symbols = false, est_fed[NEST] = {false, false, false};
    const suamd_complex *pend_src = nullptr;
    unsigned pend_spec_n = 0;
  } slot[2];
  void use(int p)
  {
    Slot &s = slot[p];
    d_y = s.d_y; d_a = s.d_a; d_z = s.d_z; d_sym = s.d_sym; h_out = s.h_out; d_count = s.d_count; pin = s.pin; d_lin = s.d_lin;
  }
  void stash(int p)
  {
    Slot &s = slot[p];
    s.pend_m = pend_m; s.pend_samples = pend_samples; s.pend_spectrum = pend_spectrum; s.pend_symbols = pend_symbols;
    s.pend_src = pend_src; s.pend_spec_n = pend_spec_n; for (int k = 0; k < NEST; ++k) s.est_fed[k] = est_fed[k];
  }
  void recall(int p)
  {
    use(p);
    const Slot &s = slot[p];
    pend_m = s.pend_m; pend_samples = s.pend_samples; pend_spectrum = s.pend_spectrum; pend_symbols = s.pend_symbols;
    pend_src = s.pend_src; pend_spec_n = s.pend_spec_n; for (int k = 0; k < NEST; ++k) est_fed[k] = s.est_fed[k];
  }
  void close_channel()
  {
    if (st && st_chan >= 0) (void)suamd_specttuner_close_channel(st, st_chan);
    st_chan = -1;
  }
  // keep_channel: the FFT channel survives a rebuild of the stages behind it (a configuration change does not touch
  // the channeliser: its stream goes on without a seam)
  void free_chain(bool keep_channel = false)
  {
    if (bank) suamd_chanbank_destroy(bank);
    if (!keep_channel) close_channel();
    if (agc) suamd_agc_bank_destroy(agc);
    if (costas) suamd_costas_bank_destroy(costas);
    if (clock) suamd_clock_bank_destroy(clock);
    if (nco) suamd_nco_bank_destroy(nco);
    if (pll) suamd_pll_bank_destroy(pll);
    if (mf) suamd_fir_bank_destroy(mf);
    if (cma) suamd_cma_bank_destroy(cma);
    if (power) suamd_power_bank_destroy(power);
    power = nullptr;
    if (audio) suamd_audio_destroy(audio);
    audio = nullptr;
    bank = nullptr; agc = nullptr; costas = nullptr; clock = nullptr; nco = nullptr; pll = nullptr; mf = nullptr; cma = nullptr;
    fixed_gain = 0;
  }
  void free_spectrum()
  {
    if (spect_psd) suamd_psd_destroy(spect_psd);
    if (d_spre) (void)hipFree(d_spre);
    if (d_spec) (void)hipFree(d_spec);
    spect_psd = nullptr; d_spre = nullptr; d_spec = nullptr; spect_n = 0;
  }
  void free_estimators()
  {
    for (auto &e : est) { if (e) suamd_baud_estimator_destroy(e); e = nullptr; }
  }
  void free_rows()
  {
    if (arena && cap) (void)hipDeviceSynchronize();            // launches still in flight read and write these rows
    for (Slot &s : slot) {
      // (columns of the analyzer's slabs are not this inspector's to give back)
      for (void *p : {in_slab ? nullptr : (void *)s.d_y, in_slab ? nullptr : (void *)s.d_a, in_slab ? nullptr : (void *)s.d_z, (void *)s.d_sym, (void *)s.d_lin}) {
        if (!p) continue;
        if (arena) arena->give(p, cap * 8); else (void)hipFree(p);
      }
      if (s.d_count) (void)hipFree(s.d_count);
      if (s.h_out) (void)hipHostFree(s.h_out);
      s.d_y = s.d_a = s.d_z = s.d_sym = s.h_out = s.d_lin = nullptr; s.d_count = nullptr;
    }
    d_y = d_a = d_z = d_sym = h_out = d_lin = nullptr; d_count = nullptr; cap = 0;
  }
  void free_all()
  {
    free_spectrum();
    free_estimators();
    free_chain();
    free_rows();
    for (Slot &s : slot) { if (s.pin) (void)hipHostFree(s.pin); s.pin = nullptr; }
    pin = nullptr;
    if (d_prev) (void)hipFree(d_prev);
    d_prev = nullptr;
    if (config) suscan_config_destroy(config);
    config = nullptr;
  }
};

struct Request {
  enum Kind { OPEN, CLOSE, SET_ID, SET_CONFIG, SET_WATERMARK, SET_FREQ, SET_BW, SET_PARAMS, SET_THROTTLE, SET_SPECTRUM,
              SEEK, SOURCE_INFO, ESTIMATOR, SET_TLE } kind;
  uint32_t req_id = 0;
  SUHANDLE handle = -1;
  std::string cls;
  struct sigutils_channel channel{};
  bool precise = false;
  uint32_t inspector_id = 0;
  suscan_config_t *config = nullptr;
  SUSCOUNT value = 0;
  double fvalue = 0;
  struct suscan_analyzer_params params{};
};

}  // namespace

// Multi-GPU: inspector channels are sharded over the devices of SUAMD_DEVICES (handle h lives on
// shard h mod G); shard 0 -- the analyzer the caller holds -- owns the source, the PSD and the channel detector and
// PUBLISHES every block (after the baseband filters) on this bus; every other shard is a worker thread bound to its
// GPU that takes the block from the publisher's pinned host buffer over its own PCIe link (or, with
// SUAMD_ANALYZER_BCAST=rccl, from GPU 0 over xGMI with one ncclBroadcast per block), runs its inspectors and posts their
// messages to the same queue.  No other exchange: the chains are independent after the shared input.
struct BlockBus {
  std::mutex m;
  std::condition_variable cv;
  uint64_t seq = 0;                             // blocks published so far; block k sits in entry k & 1
  struct Entry { const void *host = nullptr; size_t samples = 0; size_t valid = 0;   // valid < samples: the stream's tail (no reallocation, no rebuild)
                 bool via_bcast = false;          // decided by the publisher per block: every shard takes part in ONE ncclBroadcast, or none does
                  int raw_format = 0; unsigned bytes_per_sample = 8; uint64_t position = 0; unsigned samp_rate = 0; } e[2];
  bool closed = false;                          // the publisher is done (end of stream, halt, error)
  std::vector<uint64_t> done;                   // per subscriber: blocks whose host buffer it no longer needs
  // RCCL variant (opt-in): one communicator per shard, created by shard 0
  void *rccl_lib = nullptr;
  std::vector<void *> comm;
  int (*bcast)(const void *, void *, size_t, int, int, void *, hipStream_t) = nullptr;
  int (*comm_destroy)(void *) = nullptr;
  int (*comm_abort)(void *) = nullptr;          // ncclCommAbort: releases a collective whose peer never came
  int bcast_timeout_ms = 2000;                  // SUAMD_ANALYZER_BCAST_TIMEOUT_MS: how long a broadcast may sit on a stream ONCE A SHARD IS KNOWN TO BE GONE
  int bcast_ceiling_ms = 120000;                // ... and with every shard alive as far as anyone knows (a wedged peer that never said so)
  std::vector<bool> lost_reported;              // per subscriber: its loss has been posted as READ_ERROR
  bool bcast_off = false;                       // a broadcast failed, or a shard gave up: per-GPU host copies from here on
  uint64_t bcast_blocks = 0;                    // blocks that went out through ncclBroadcast (diagnostic, SOURCE_INFO-independent)
};

struct suscan_analyzer {
  int device = 0;                               // the GPU this shard is bound to
  int shard = 0, nshards = 1;
  int fault_shard = -1; long long fault_block = -1;   // SUAMD_ANALYZER_FAULT=shard_dies:<shard>:<block> (tests): that shard stops at that block
  suscan_analyzer *primary = nullptr;           // shard 0 (itself for shard 0)
  std::vector<suscan_analyzer *> secondaries;   // shard 0 only: shards 1 .. G-1
  std::shared_ptr<BlockBus> bus;
  std::atomic<uint32_t> open_rr{0};             // shard 0: round-robin placement of OPEN requests
  struct suscan_analyzer_params params;
  suscan_source_config source_cfg;
  struct suscan_mq *mq;
  std::thread worker;
  std::mutex req_m;
  std::deque<Request> requests;
  std::atomic<bool> halt{false};
  std::atomic<float> measured_rate{0.f};
  std::atomic<uint64_t> throttle{0};
  struct suscan_source_info info{};
  SUHANDLE next_handle = 0;
  // baseband filters (registered from the GUI thread, called on the worker thread): under req_m
  struct Filter { suscan_analyzer_baseband_filter_func_t func; void *priv; int64_t prio; uint64_t seq; };
  std::vector<Filter> filters;
  uint64_t filter_seq = 0;
  // source controls
  std::atomic<bool> iq_reverse{false}, dc_remove{false};
  std::atomic<uint64_t> position{0};            // samples delivered so far (get_source_time, filter offsets)
  std::map<std::string, float> gains;
  std::string antenna;
  SUSCOUNT history_size = 0;
  bool replay = false;
  // wide-spectrum (panoramic) mode, under req_m: every block of the source is one dwell of the sweep
  int sweep_strategy = 0, partitioning = 0;
  double hop_min = 0, hop_max = 0;
  float rel_bw = 1.0f;
  uint64_t hop_k = 0;                          // dwells so far
  uint32_t hop_lcg = 0x2545f491u;              // the stochastic strategy's generator
  SUSCOUNT buffering_size = 0;
  // worker-owned
  suamd_ctx_t *ctx = nullptr;
  suamd_psd_t *psd = nullptr;
  // the inspectors' channeliser: the FFT filter bank (su_specttuner semantics: one forward FFT of the block shared by all
  // inspectors -- what libsuscan itself runs) whenever the block is a whole number of half windows, else -- or with
  // SUAMD_ANALYZER_CHANNELISER=fir -- one translate + 255-tap FIR per inspector
  suamd_chandet_t *chandet = nullptr;         // su_channel_detector on the block spectra -> CHANNEL messages
  unsigned chan_every = 1, chan_phase = 0;    // a list every chan_every blocks (channel_update_int of signal time)
  bool want_fft = true, use_fft = false;
  suamd_specttuner_t *st = nullptr;
  bool st_idle = false;                       // the tuner was reset when the last inspector went away (no stale history at the next open)
  std::atomic<bool> st_reset_pending{false};  // a SEEK moved the source: every shard forgets the tuner's stream position
  suamd_complex **d_rowptr[2] = {nullptr, nullptr};   // per slot: where each FFT channel's row starts (device table)
  suamd_complex **h_rowptr[2] = {nullptr, nullptr};   // pinned staging, and what the device table holds
  size_t rowptr_cap = 0;
  RowArena rows;                              // the inspectors' sample rows (one slab: rows near each other)
  // Time-major slabs of the narrow FFT channels, per slot: y = what the channeliser writes ([time][pitch], column = channel
  // index), a / z = the chain's ping-pong partners, work = the AGC's magnitudes and levels (2 x rows x pitch floats).  The
  // three complex slabs of a slot are ONE allocation (gang items of one call then lie within 4 GiB of each other).
  struct SlabSet { suamd_complex *y = nullptr, *a = nullptr, *z = nullptr; float *work = nullptr; } slab[2];
  size_t slab_rows = 0, slab_pitch = 0;       // rows include 128 of slack: the gang kernels look a tile ahead
  uint32_t *d_sink = nullptr;                 // where the column -> row copies write their counts
  std::map<SUHANDLE, std::unique_ptr<Inspector>> inspectors;
  hipStream_t stream = nullptr;
  static constexpr int NISTREAMS = 4;          // gain control / carrier control / clock recovery / channeliser (+ spectra, estimators)
  static constexpr int NSUB = 16;             // at most this many sub-ranges of a block pipelined through those stages
  int nsub = 4;                               // sub-ranges of the block being enqueued (nsub_env, or chosen by the inspector count)
  int nsub_env = 0;                           // SUAMD_ANALYZER_SUBRANGES; 0: automatic
  bool slab_on = true;                        // tuning().analyzer_slab as it was when the shard started: the narrow channels' layout is one decision per analyzer
  bool trace = false;                         // SUAMD_ANALYZER_TRACE: per-block host timeline on stderr
  double t_chains_done = 0;
  hipEvent_t ev_t0 = nullptr, ev_tfir = nullptr, ev_tpre = nullptr, ev_tdone = nullptr, ev_tstage[3][NSUB] = {};   // timed, trace only
  std::chrono::steady_clock::time_point t_block0;
  hipEvent_t ev_stage[3][NSUB] = {};
  hipStream_t istream[NISTREAMS] = {};
  hipEvent_t ev_input = nullptr;              // the block is in d_x
  hipEvent_t ev_xfree = nullptr;              // ... and every kernel that reads it has run
  hipEvent_t ev_fir = nullptr;                // the channel samples of the block are in the inspectors' rows
  bool xfree_set = false;
  hipEvent_t ev_done[2][NISTREAMS] = {};      // the inspector work of the block in slot p is through stream k
  hipEvent_t ev_psd[2] = {};                  // the PSD of the block in slot p is in h_psd[p]
  float *h_psd[2] = {nullptr, nullptr};       // pinned landing zones of the PSD frames
  hipEvent_t ev_h2d[2] = {};                  // the host half h has been copied out: it may take the next read
  hipEvent_t ev_bcast = nullptr;              // shard 0 (the root): behind its latest ncclBroadcast; settled lazily (bcast_settle)
  bool bcast_pending = false;
  bool h2d_set[2] = {false, false};
  bool pipelined = true;                      // two blocks in flight (SUAMD_ANALYZER_PIPELINE=0 or the trace knob: one)
  suamd_complex *h_x = nullptr, *d_x = nullptr;
  suamd_complex *h_flt = nullptr;              // a compact-format block expanded on the host, for the baseband filters
  float *d_dc = nullptr;                       // tracked DC level (suamd_source_fix)
  bool dc_first = true;
  void *d_raw = nullptr;                       // compact-format payload before suamd_ingest_iq
  float *d_psd = nullptr;
  size_t block = 0;
  unsigned navg = 1;
};

namespace {

void push(suscan_analyzer *a, uint32_t type, void *msg) { suscan_mq_write(a->mq, type, msg); }

void push_status(suscan_analyzer *a, uint32_t type, int code, const std::string &text)
{
  auto *m = static_cast<suscan_analyzer_status_msg *>(std::calloc(1, sizeof(suscan_analyzer_status_msg)));
  m->code = code;
  m->err_msg = text.empty() ? nullptr : dupstr(text.c_str());
  push(a, type, m);
}

suscan_analyzer_inspector_msg *new_insp_msg(suscan_analyzer_inspector_msgkind kind, uint32_t req_id)
{
  auto *m = static_cast<suscan_analyzer_inspector_msg *>(std::calloc(1, sizeof(suscan_analyzer_inspector_msg)));
  m->kind = kind;
  m->req_id = req_id;
  m->handle = -1;
  m->signal_name = strdup("");                              // InspectorMessage.cpp:239 builds a std::string from it
  m->signal_value = std::nan("");
  return m;
}

unsigned pow2floor(double v) { unsigned d = 1; while ((double)(d * 2) <= v && d < 4096) d *= 2; return d; }

// (re)builds the GPU chain of one inspector from its channel + config
bool build_chain(suscan_analyzer *a, Inspector &in, std::string &err)
{
  in.free_chain(/* keep_channel = */ a->use_fft);
  const double fs = a->source_cfg.samp_rate;
  const double bw = in.channel.bw > 0 ? in.channel.bw : fs / 4;
  in.D = pow2floor(fs / (2.0 * bw));
  in.equiv_fs = fs / in.D;
  in.fnor = 2.0 * in.channel.fc / fs;                    // channel centre relative to the tuner
  if (a->use_fft) {
    if (in.D > 2048) { in.D = 2048; in.equiv_fs = fs / in.D; }   // the filter bank's smallest channel is 2 of its 4096 bins
    // a channel of the shared FFT filter bank: f0, bw as angular frequencies; the guard band sizes the channel for
    // exactly W / D bins (guard = fs / (D bw) >= 2), so the channel decimates by D like the FIR path does
    if (!a->st) a->st = suamd_specttuner_new(a->ctx, 4096);
    if (!a->st) { err = suamd_last_error(); return false; }
    constexpr double kTwoPi = 6.283185307179586476925286766559;
    double f0 = std::fmod(kTwoPi * in.channel.fc / fs, kTwoPi);
    if (f0 < 0) f0 += kTwoPi;
    const double bwa = kTwoPi * bw / fs, guard = fs / ((double)in.D * bw);
    if (in.st_chan >= 0 && (in.st != a->st || in.st_f0 != f0 || in.st_bw != bwa || in.st_guard != guard || in.st_precise != in.precise)) in.close_channel();
    in.st = a->st;
    if (in.st_chan < 0) {
      in.st_chan = suamd_specttuner_open_channel(a->st, f0, bwa, guard, in.precise ? SU_TRUE : SU_FALSE);
      if (in.st_chan < 0) { err = suamd_last_error(); return false; }
      in.st_f0 = f0; in.st_bw = bwa; in.st_guard = guard; in.st_precise = in.precise;
    }
    if (suamd_specttuner_channel_decimation(a->st, in.st_chan) != in.D) {
      err = "FFT channeliser: unexpected decimation";
      return false;
    }
  }
  {
    // narrow channels of the filter bank are columns of the shard's slab (the feed sends every channel of <= 64 bins there);
    // classes whose stages want contiguous rows throughout copy their column out instead of living in it
    const bool narrow = a->use_fft && a->slab_on && suamd_specttuner_channel_size(a->st, in.st_chan) <= 64;
    const bool lives = narrow && in.cls != "audio" && in.cls != "power";
    if (lives != in.in_slab) in.free_rows();                 // (rows <-> columns: the other kind of buffers)
    in.in_slab = lives;
    in.copy_out = narrow && !lives;
    in.lane = narrow ? in.st_chan : -1;
    if (!lives) in.ts = 1;
  }
  if (!a->use_fft) {
    float taps[255];
    suamd_lpf_design(taps, 255, bw / fs);                // cut-off bw/2 in Hz = (bw/fs) of Nyquist
    const double fn = in.fnor;
    in.bank = suamd_chanbank_new(a->ctx, 1, &fn, in.D, taps, 255);
    if (!in.bank) { err = suamd_last_error(); return false; }
  }
  const size_t need = a->block / in.D + 8;
  if (need > in.cap) {
    in.free_rows();
    bool ok = true;
    in.arena = &a->rows;
    in.cap = need;                                       // (free_rows hands rows back by this size)
    const bool poison = sdk::tuning().analyzer_poison_rows != 0;   // debug: rows start as NaNs, not as whatever was there
    auto row = [&](suamd_complex **p) {
      *p = static_cast<suamd_complex *>(a->rows.take(need * 8));
      if (*p && poison) (void)hipMemset(*p, 0xff, need * 8);
      return *p != nullptr;
    };
    for (Inspector::Slot &sl : in.slot)
      ok = ok && (in.in_slab || (row(&sl.d_y) && row(&sl.d_a) && row(&sl.d_z))) && row(&sl.d_sym) &&
           hipMalloc((void **)&sl.d_count, 4) == hipSuccess &&
           hipHostMalloc((void **)&sl.h_out, need * 8, hipHostMallocMapped) == hipSuccess;
    if (ok && !in.d_prev) ok = hipMalloc((void **)&in.d_prev, 8) == hipSuccess;
    if (!ok) { in.free_rows(); err = "device allocation failed"; return false; }
  }
  if (!in.stream) in.stream = a->istream[0];
  for (Inspector::Slot &sl : in.slot)
    if (!sl.pin && hipHostMalloc((void **)&sl.pin, sizeof(Inspector::Pinned), hipHostMallocDefault) != hipSuccess) {
      err = "pinned allocation failed"; return false;
    }
  in.spect_have_prev = false;                            // a new chain starts its channel from scratch
  (void)hipMemsetAsync(in.d_prev, 0, 8, in.stream);
  for (Inspector::Slot &sl : in.slot) (void)hipMemsetAsync(sl.d_count, 0, 4, in.stream);   // from here on cleared by every hand-off (suamd_rows_deliver)
  in.first = true;
  in.quad = false;
  if (in.cls == "raw") return true;
  if (in.cls == "power") {
    const double n = cfg_get(in.config, "power.integrate-samples", 1000);
    in.power = suamd_power_bank_new(a->ctx, n >= 1 ? (SUSCOUNT)n : 1);
    if (!in.power) { err = suamd_last_error(); return false; }
    return true;
  }
  if (in.cls == "audio") {
    // gain control (optional) -> demodulator -> cut-off low-pass + resampler to the sound card's rate (SPEC.md section Q)
    if (cfg_get(in.config, "agc.enabled", 0) != 0) {
      struct suamd_agc_params prm;
      suamd_agc_params_from_tau(&prm, (float)std::fmax(8.0, cfg_get(in.config, "agc.ts", 0.2) * in.equiv_fs * 1e-2));
      in.agc = suamd_agc_bank_new(a->ctx, 1, &prm);
      if (!in.agc) { err = suamd_last_error(); return false; }
    }
    in.audio = suamd_audio_new(a->ctx, (SUFLOAT)in.equiv_fs, (SUFLOAT)bw);
    if (!in.audio || !suamd_audio_configure(in.audio, (int)cfg_get(in.config, "audio.demodulator", 2),
                                            (SUFLOAT)cfg_get(in.config, "audio.sample-rate", 44100), (SUFLOAT)cfg_get(in.config, "audio.cutoff", 15000),
                                            (SUFLOAT)cfg_get(in.config, "audio.volume", 1), cfg_get(in.config, "audio.squelch", 0) != 0 ? SU_TRUE : SU_FALSE,
                                            (SUFLOAT)cfg_get(in.config, "audio.squelch-level", 0.5))) { err = suamd_last_error(); return false; }
    return true;
  }
  const double baud = cfg_get(in.config, "clock.baud", 0);
  const double sps = baud > 0 ? in.equiv_fs / baud : 8.0;
  // stage order of the generic inspector: gain control -> carrier control -> matched filter ->
  // clock recovery -> equalizer (Default/GenericInspector/InspectorCtl/*.cpp, one control each)
  if (cfg_get(in.config, "agc.enabled", 1) != 0) {
    struct suamd_agc_params prm;
    suamd_agc_params_from_tau(&prm, (float)sps);
    in.agc = suamd_agc_bank_new(a->ctx, 1, &prm);
    if (!in.agc) { err = suamd_last_error(); return false; }
  } else {
    in.fixed_gain = (float)std::pow(10.0, cfg_get(in.config, "agc.gain", 0) / 20.0);     // dB spin box
  }
  if (in.cls == "psk") {
    const int order = (int)cfg_get(in.config, "afc.costas-order", 0);
    if (order >= 1 && order <= 3) {
      const double loop_bw = cfg_get(in.config, "afc.loop-bw", 100);
      in.costas = suamd_costas_bank_new(a->ctx, 1, order, 0.0f, (float)std::fmin(2.0 / sps, 0.95), 3,
                                        (float)(2.0 * loop_bw / in.equiv_fs));
      if (!in.costas) { err = suamd_last_error(); return false; }
    } else if (cfg_get(in.config, "afc.offset", 0) != 0) {  // manual: mix the offset away
      const double fn = -2.0 * cfg_get(in.config, "afc.offset", 0) / in.equiv_fs;
      in.nco = suamd_nco_bank_new(a->ctx, 1, &fn);
      if (!in.nco) { err = suamd_last_error(); return false; }
    }
  } else if (in.cls == "ask") {
    if (cfg_get(in.config, "ask.use-pll", 0) != 0) {      // su_pll_init(fhint = offset, fc = loop bandwidth)
      in.pll = suamd_pll_bank_new(a->ctx, 1, (float)(2.0 * cfg_get(in.config, "ask.offset", 0) / in.equiv_fs),
                                  (float)(2.0 * cfg_get(in.config, "ask.loop-bw", 100) / in.equiv_fs));
      if (!in.pll) { err = suamd_last_error(); return false; }
    }
  } else {                                                // fsk
    in.quad = cfg_get(in.config, "fsk.quad-demod", 1) != 0;
  }
  if ((int)cfg_get(in.config, "mf.type", 0) == 1 && baud > 0) {
    const unsigned nt = suamd_rrc_ntaps(sps);
    std::vector<float> h(nt);
    suamd_rrc_design(h.data(), nt, sps, cfg_get(in.config, "mf.roll-off", .35));
    in.mf = suamd_fir_bank_new(a->ctx, 1, h.data(), nt);
    if (!in.mf) { err = suamd_last_error(); return false; }
  }
  if (baud > 0) {
    const bool gardner = (int)cfg_get(in.config, "clock.type", 0) == 1;
    in.clock = suamd_clock_bank_new(a->ctx, 1, gardner ? (float)cfg_get(in.config, "clock.gain", .2) : 0.0f,
                                    (float)(baud / in.equiv_fs));
    if (!in.clock) { err = suamd_last_error(); return false; }
    if (!gardner) suamd_clock_bank_set_phase(in.clock, 0.5f * (float)cfg_get(in.config, "clock.phase", 0), in.stream);
    if ((int)cfg_get(in.config, "equalizer.type", 0) == 1) {
      in.cma = suamd_cma_bank_new(a->ctx, 1, 8, (float)cfg_get(in.config, "equalizer.rate", 1e-3));
      if (!in.cma) { err = suamd_last_error(); return false; }
      suamd_cma_bank_set_locked(in.cma, cfg_get(in.config, "equalizer.locked", 0) != 0);
    }
  }
  return true;
}

void push_samples(suscan_analyzer *a, const Inspector &in, const suamd_complex *src, size_t count)
{
  auto *m = static_cast<suscan_analyzer_sample_batch_msg *>(std::calloc(1, sizeof(suscan_analyzer_sample_batch_msg)));
  m->inspector_id = in.inspector_id;
  m->sample_count = count;
  m->samples = static_cast<suamd_complex *>(std::malloc(count * sizeof(suamd_complex)));
  std::memcpy(m->samples, src, count * sizeof(suamd_complex));
  push(a, SUSCAN_ANALYZER_MESSAGE_TYPE_SAMPLES, m);
}

// One block's output of an inspector.  No watermark: one SAMPLES message per block.  Watermark w
// (Analyzer::setInspectorWatermark, Suscan/Analyzer.cpp:528-537): batches of exactly w samples; what does not fill one
// waits for the next block.  A batch is shorter only where the stream itself is cut: CLOSE / EOS, a retune or a
// reconfiguration (flush_watermark: the remainder goes out before samples of the new tuning follow).  The stream is the same either way.
void emit_samples(suscan_analyzer *a, Inspector &in, size_t count)
{
  if (count == 0 || count > in.cap) return;
  if (in.watermark == 0 && in.wm_buf.empty()) { push_samples(a, in, in.h_out, count); return; }   // delivered by the device before the sync
  in.wm_buf.insert(in.wm_buf.end(), in.h_out, in.h_out + count);
  const size_t w = in.watermark ? (size_t)in.watermark : in.wm_buf.size();    // (the watermark was just cleared: everything goes)
  size_t off = 0;
  for (; in.wm_buf.size() - off >= w && w > 0; off += w) push_samples(a, in, in.wm_buf.data() + off, w);
  in.wm_buf.erase(in.wm_buf.begin(), in.wm_buf.begin() + (long)off);
}

void flush_watermark(suscan_analyzer *a, Inspector &in)
{
  if (!in.wm_buf.empty()) push_samples(a, in, in.wm_buf.data(), in.wm_buf.size());
  in.wm_buf.clear();
}

// INSPECTOR/SPECTRUM: the selected source's transform of this block's channel samples, then every whole
// frame of the block Welch-averaged into one spectrum (linear power, natural order: the tab takes dB and
// rotates, GenericInspector.cpp:231-247); frame = the largest power of two <= min(block, 8192)
void enqueue_spectrum(suscan_analyzer *a, Inspector &in, SUSCOUNT m)
{
  unsigned n = 512;
  while (n * 2 <= m && n < 8192) n *= 2;
  if (m < n) return;
  if (n != in.spect_n) {
    in.free_spectrum();
    in.spect_psd = suamd_psd_new(a->ctx, n, a->params.detector_params.window);
    if (!in.spect_psd || hipMalloc((void **)&in.d_spre, in.cap * sizeof(suamd_complex)) != hipSuccess ||
        hipMalloc((void **)&in.d_spec, n * sizeof(float)) != hipSuccess) {
      push_status(a, SUSCAN_ANALYZER_MESSAGE_TYPE_INTERNAL, -1, "inspector spectrum: allocation failed");
      in.free_spectrum();
      in.spectsrc_id = 0;
      return;
    }
    in.spect_n = n;
  }
  const suamd_complex *d_before = in.spect_have_prev && in.last_fir_m ? in.slot[in.last_slot].d_y + (in.last_fir_m - 1) * in.ts : nullptr;
  if (!suamd_spectsrc_preproc_from(a->ctx, in.spectsrc_id, in.in_slab ? in.d_lin : in.d_y, m, d_before, in.d_spre, in.stream)) return;
  const unsigned frames = (unsigned)(m / n);
  if (!suamd_psd_feed(in.spect_psd, in.d_spre, frames, n, frames, 1.0f / (float)n, SUAMD_PSD_LINEAR, in.d_spec, in.stream)) return;
  (void)hipMemcpyAsync(in.pin->spec, in.d_spec, n * sizeof(float), hipMemcpyDeviceToHost, in.stream);
  in.pend_spectrum = true;
  in.pend_spec_n = n;
}

// One block through every open inspector.  The per-inspector kernels (channel FIR, spectrum, the parallel
// parts of the AGC, matched filter) are short and queue back to back; the serial recurrences -- AGC level
// trackers, Costas loops / PLLs, Gardner detectors, equalizers -- of ALL inspectors run as gangs (one lane per
// inspector, its own parameters and length), so their cost is that of one inspector, not the sum.  The three
// gang stages sit on three streams and the block is pushed through them in NSUB sub-ranges: while the Costas
// gang works on sub-range j the level trackers are already on j+1 and the Gardner gang on j-1 (every stage
// is invariant under splitting its input, which the parity tests pin).  Bit-identical to running every chain
// on its own, whole block at once.
void enqueue_inspectors_slot(suscan_analyzer *a, size_t len, int slot);

// The slabs of the narrow FFT channels: pitch = the tuner's channel table rounded up to 64 columns, rows = the longest
// narrow channel's samples per block (D = 64) + slack.  Grown when the table or the block grows (contents carried over:
// the previous block's last samples are still looked at); every inspector's column pointers are set from here.
bool ensure_slabs(suscan_analyzer *a)
{
  bool any = false;
  for (auto &kv : a->inspectors) any = any || kv.second->in_slab || kv.second->copy_out;
  if (!any) return true;
  const size_t cap = a->st ? suamd_specttuner_channel_capacity(a->st) : 0;
  const size_t pitch = std::max<size_t>(64, (cap + 63) / 64 * 64);
  const size_t rows = a->block / 64 + 8 + 128;
  if (pitch > a->slab_pitch || rows > a->slab_rows) {
    (void)hipDeviceSynchronize();                            // launches in flight read and write the old slabs
    const size_t np = std::max(pitch, a->slab_pitch), nr = std::max(rows, a->slab_rows);
    const size_t one = nr * np;                              // elements of one complex slab
    suscan_analyzer::SlabSet fresh[2];
    bool ok = true;
    for (auto &f : fresh) {
      ok = ok && hipMalloc((void **)&f.y, 3 * one * sizeof(suamd_complex)) == hipSuccess &&
           hipMalloc((void **)&f.work, 2 * one * sizeof(float)) == hipSuccess;
      if (!ok) break;
      f.a = f.y + one; f.z = f.a + one;
      const int fill = sdk::tuning().analyzer_poison_rows != 0 ? 0xff : 0;
      ok = hipMemset(f.y, fill, 3 * one * sizeof(suamd_complex)) == hipSuccess && hipMemset(f.work, 0, 2 * one * sizeof(float)) == hipSuccess;
    }
    if (ok && !a->d_sink) ok = hipMalloc((void **)&a->d_sink, 4) == hipSuccess;
    if (!ok) {
      (void)hipGetLastError();
      for (auto &f : fresh) { if (f.y) (void)hipFree(f.y); if (f.work) (void)hipFree(f.work); }
      return false;
    }
    for (int p = 0; p < 2; ++p) {
      suscan_analyzer::SlabSet &o = a->slab[p];
      if (o.y) {
        const suamd_complex *src[3] = {o.y, o.a, o.z};
        suamd_complex *dst[3] = {fresh[p].y, fresh[p].a, fresh[p].z};
        for (int k = 0; k < 3; ++k)
          (void)hipMemcpy2D(dst[k], np * sizeof(suamd_complex), src[k], a->slab_pitch * sizeof(suamd_complex),
                            a->slab_pitch * sizeof(suamd_complex), a->slab_rows, hipMemcpyDeviceToDevice);
        (void)hipFree(o.y);
        (void)hipFree(o.work);
      }
      o = fresh[p];
    }
    (void)hipDeviceSynchronize();
    a->slab_pitch = np; a->slab_rows = nr;
  }
  for (auto &kv : a->inspectors) {
    Inspector &in = *kv.second;
    if (!in.in_slab) continue;
    in.ts = a->slab_pitch;
    for (int p = 0; p < 2; ++p) {
      in.slot[p].d_y = a->slab[p].y + in.lane;
      in.slot[p].d_a = a->slab[p].a + in.lane;
      in.slot[p].d_z = a->slab[p].z + in.lane;
    }
  }
  return true;
}

// Block k goes into slot k & 1 while block k-1 (the other slot) may still be on the device: every stream takes the
// blocks in order, so the loop states carry over by themselves, and the two slots share no per-block buffer.
void enqueue_inspectors(suscan_analyzer *a, size_t len, int slot)
{
  enqueue_inspectors_slot(a, len, slot);
  for (auto &kv : a->inspectors) kv.second->stash(slot);
  for (int k = 0; k < suscan_analyzer::NISTREAMS; ++k) (void)hipEventRecord(a->ev_done[slot][k], a->istream[k]);
}

void enqueue_inspectors_slot(suscan_analyzer *a, size_t len, int slot)
{
  if (a->st_reset_pending.exchange(false) && a->st) (void)suamd_specttuner_reset(a->st, a->istream[3]);
  // Sub-ranges per stage: four let the three serial stages of a block overlap when a handful of wavefronts carry all
  // inspectors; with hundreds of inspectors the worker's own enqueue time is what bounds the rate, and two sub-ranges
  // halve it (2 Mi-sample blocks, MS/s at the consumer with 1 / 2 / 4 sub-ranges: 64 inspectors 797 / 1048 / 1055,
  // 128: 779 / 1022 / 1011, 256: 1033 / 1180 / 950, 512: 1030 / 1017 / 891).
  if (!a->nsub_env) a->nsub = a->inspectors.size() > 128 ? 2 : 4;
  const int P = a->nsub;
  hipStream_t sA = a->istream[0], sC = a->istream[1], sK = a->istream[2], sF = a->istream[3];
  // the channeliser has its own stream: with hundreds of inspectors it is as long as a recurrence stage, and block
  // k+1's may run beside block k's gain control.  It overwrites this slot's channel rows: their readers two blocks ago
  // (every other stream) must be through
  (void)hipStreamWaitEvent(sF, a->ev_input, 0);
  for (int k = 0; k < 3; ++k) (void)hipStreamWaitEvent(sF, a->ev_done[slot][k], 0);
  std::vector<Inspector *> live;
  for (auto &kv : a->inspectors) {
    Inspector &in = *kv.second;
    in.use(slot);
    in.pend_samples = in.pend_spectrum = in.pend_symbols = false;
    for (bool &f : in.est_fed) f = false;
    in.stream = sA;
    if (in.dirty) {
      std::string err;
      flush_watermark(a, in);                                 // (a rebuild for any other reason -- the block size changed -- as well)
      if (!build_chain(a, in, err)) {
        // the inspector sits this block out; its channel must not stay a member of the filter bank, whose kernel would
        // store that channel's samples through a row pointer nobody maintains
        in.close_channel();
        push_status(a, SUSCAN_ANALYZER_MESSAGE_TYPE_INTERNAL, -1, err);
        continue;
      }
      in.dirty = false;
      in.use(slot);                                           // the rebuild may have re-allocated the slot's rows
    }
    live.push_back(&in);
  }
  if (live.empty()) {
    // nobody to feed: the tuner's half-window history would be from an unrelated stream position when feeding resumes
    if (a->use_fft && a->st && !a->st_idle) { (void)suamd_specttuner_reset(a->st, sF); a->st_idle = true; }
    return;
  }
  a->st_idle = false;
  auto fail = [&](const char *what) { push_status(a, SUSCAN_ANALYZER_MESSAGE_TYPE_INTERNAL, -1, std::string(what) + ": " + suamd_last_error()); };
  if (!ensure_slabs(a)) { push_status(a, SUSCAN_ANALYZER_MESSAGE_TYPE_INTERNAL, -1, "inspector slabs: device allocation failed"); return; }
  for (Inspector *pi : live) pi->use(slot);                   // (the slabs may have moved)
  const suscan_analyzer::SlabSet &sb = a->slab[slot];
  const SUSCOUNT pitch = (SUSCOUNT)a->slab_pitch;
  {
    // every inspector channelises the same wideband block: one launch for all of them
    std::vector<suamd_chanbank_t *> fb; std::vector<suamd_complex *> fy; std::vector<SUSCOUNT> fm(live.size());
    if (a->use_fft) {
      // the FFT filter bank: ONE forward transform of the block for all inspectors (per size group); every channel's row
      // of this slot is looked up in a device table indexed by channel
      size_t nrows = 0;
      for (Inspector *pi : live) nrows = std::max(nrows, (size_t)pi->st_chan + 1);
      if (nrows > a->rowptr_cap) {
        const size_t cap = std::max<size_t>(64, 2 * nrows);
        for (int p = 0; p < 2; ++p) {
          if (a->d_rowptr[p]) (void)hipFree(a->d_rowptr[p]);
          if (a->h_rowptr[p]) (void)hipHostFree(a->h_rowptr[p]);
          a->d_rowptr[p] = a->h_rowptr[p] = nullptr;
          if (hipMalloc((void **)&a->d_rowptr[p], cap * sizeof(void *)) != hipSuccess ||
              hipHostMalloc((void **)&a->h_rowptr[p], cap * sizeof(void *), hipHostMallocDefault) != hipSuccess) { fail("channeliser tables"); return; }
          std::memset(a->h_rowptr[p], 0, cap * sizeof(void *));
        }
        a->rowptr_cap = cap;
        for (int p = 0; p < 2; ++p) (void)hipMemcpyAsync(a->d_rowptr[p], a->h_rowptr[p], cap * sizeof(void *), hipMemcpyHostToDevice, sF);
      }
      bool changed = false, any_col = false;
      for (Inspector *pi : live) {
        if (pi->lane >= 0) { any_col = true; continue; }      // a column of the slab: the view serves it, not the table
        if (a->h_rowptr[slot][pi->st_chan] != pi->d_y) { a->h_rowptr[slot][pi->st_chan] = pi->d_y; changed = true; }
      }
      // (the slot's previous block has been collected by now: its table and staging are free to change)
      if (changed) (void)hipMemcpyAsync(a->d_rowptr[slot], a->h_rowptr[slot], nrows * sizeof(void *), hipMemcpyHostToDevice, sF);
      // one count per slot of the tuner's channel table: it never shrinks, so after a close (or a retune, which closes
      // and reopens) it can be longer than the highest live channel
      std::vector<SUSCOUNT> counts(std::max(nrows, (size_t)suamd_specttuner_channel_capacity(a->st)), 0);
      // the inspectors' rows are separate allocations, but of one arena in practice: when they all start within ~1.75 GiB the
      // narrow-channel kernels address them with 32-bit offsets from the lowest (suamd_specttuner_feed_rows_near)
      uintptr_t lo = ~(uintptr_t)0, hi = 0;
      size_t nrow = 0;
      for (Inspector *pi : live) {
        if (pi->lane >= 0) continue;
        const uintptr_t p = reinterpret_cast<uintptr_t>(pi->d_y); lo = std::min(lo, p); hi = std::max(hi, p); ++nrow;
      }
      const bool near = nrow > 0 && hi - lo < ((uintptr_t)1 << 31) - ((uintptr_t)1 << 28);   // the library checks the feed's own extent on top
      if (sdk::tuning().analyzer_debug) { static int once = 0; if (!once++) std::fprintf(stderr, "[worker] %zu inspectors: %zu in rows spanning %.1f MiB (%s), %zu in slab columns (pitch %zu)\n", live.size(), nrow, nrow ? (double)(hi - lo) / 1048576.0 : 0.0, near ? "near" : "64-bit", live.size() - nrow, (size_t)pitch); }
      const void *base = near ? reinterpret_cast<const void *>(lo) : nullptr;
      const size_t span = near ? (size_t)(hi - lo) + 8 : 0;
      if (any_col) {
        // narrow channels (every channel of <= 64 bins is a column) through the time-major view, wide ones to their rows
        if (!suamd_specttuner_feed_mixed(a->st, a->d_x, len, sb.y, suamd_view{1, pitch}, 64, a->d_rowptr[slot], base, span, counts.data(), sF)) { fail("channeliser"); return; }
      } else if (!(near ? suamd_specttuner_feed_rows_near(a->st, a->d_x, len, a->d_rowptr[slot], base, span, counts.data(), sF)
                        : suamd_specttuner_feed_rows(a->st, a->d_x, len, a->d_rowptr[slot], counts.data(), sF))) { fail("channeliser"); return; }
      for (size_t i = 0; i < live.size(); ++i) fm[i] = counts[live[i]->st_chan];
      {
        // columns that are wanted as rows: a copy_out inspector's d_y, the d_lin of a slab inspector with a spectrum or an estimator on
        std::vector<const suamd_complex *> src; std::vector<SUSCOUNT> stride, fixed; std::vector<suamd_complex *> dst; std::vector<uint32_t *> sink;
        for (size_t i = 0; i < live.size(); ++i) {
          Inspector &in = *live[i];
          if (in.lane < 0 || fm[i] == 0) continue;
          suamd_complex *to = nullptr;
          if (in.copy_out) to = in.d_y;
          else if (in.spectsrc_id || in.est_on[0] || in.est_on[1] || in.est_on[2]) {
            if (!in.d_lin) {
              for (Inspector::Slot &sl : in.slot) if (!sl.d_lin) sl.d_lin = static_cast<suamd_complex *>(a->rows.take(in.cap * 8));
              in.d_lin = in.slot[slot].d_lin;
            }
            to = in.d_lin;
          }
          if (!to) continue;
          src.push_back(sb.y + in.lane); stride.push_back(pitch); fixed.push_back(fm[i]); dst.push_back(to); sink.push_back(a->d_sink);
        }
        if (!src.empty() && !suamd_rows_deliver_strided(a->ctx, (unsigned)src.size(), src.data(), stride.data(), nullptr, fixed.data(), dst.data(), sink.data(), sF)) fail("channel rows");
      }
    } else {
      for (Inspector *pi : live) { fb.push_back(pi->bank); fy.push_back(pi->d_y); }
      if (!suamd_chanbank_gang_feed(a->ctx, fb.data(), (unsigned)fb.size(), a->d_x, len, fy.data(), fm.data(), sF)) { fail("channeliser"); return; }
    }
    (void)hipEventRecord(a->ev_xfree, sF);                    // the wideband block may be overwritten from here on (the PSD is on the input stream itself)
    a->xfree_set = true;
    (void)hipEventRecord(a->ev_fir, sF);
    (void)hipStreamWaitEvent(sA, a->ev_fir, 0);               // every other stage is downstream of the gain-control stream
    if (a->trace) (void)hipEventRecord(a->ev_tfir, sF);
    for (size_t i = 0; i < live.size(); ++i) {
      Inspector &in = *live[i];
      in.pend_m = fm[i];
      in.pend_src = in.d_y;
      in.stream = sF;                                         // spectra and estimators read the channel samples: beside the chain
      const suamd_complex *yrow = in.in_slab ? in.d_lin : in.d_y;   // the block's channel samples as a contiguous row (null: none was made)
      if (in.spectsrc_id && yrow) enqueue_spectrum(a, in, fm[i]);
      in.spect_have_prev = true; in.last_slot = slot; in.last_fir_m = fm[i];   // for the next block's "sample before"
      for (int k = 0; k < Inspector::NEST; ++k) {             // enabled estimators look at the channel samples too
        if (!in.est_on[k]) continue;
        unsigned want = 512;
        while (want * 2 <= fm[i] && want < 8192) want *= 2;
        if (fm[i] < want) continue;                           // fewer than 512 channel samples per block: no estimate
        if (in.est[k] && suamd_baud_estimator_size(in.est[k]) != want) { suamd_baud_estimator_destroy(in.est[k]); in.est[k] = nullptr; }
        if (!in.est[k]) in.est[k] = suamd_baud_estimator_new(a->ctx, k, want);   // (the estimator ids ARE the kinds: fac, nonlinear, carrier)
        if (!yrow) continue;
        if (!in.est[k] || !suamd_baud_estimator_feed_to(in.est[k], yrow, fm[i], &in.pin->est[k], sF)) { fail("estimator"); continue; }
        in.est_fed[k] = true;
      }
    }
  }
  auto sub = [&](const Inspector &in, int j) { return (SUSCOUNT)((unsigned long long)in.pend_m * (unsigned)j / P); };
  // per inspector: the row every stage reads and the one it writes (ping-pong d_a / d_z, nothing runs in place)
  struct Route { const suamd_complex *agc_in, *car_in, *mf_in, *clk_in; suamd_complex *agc_out, *car_out, *mf_out; };
  std::vector<Route> rt(live.size());
  // gain control of the inspectors that own rows (gb ...) and of those that live in the slabs (sgb ...)
  std::vector<suamd_agc_bank_t *> gb; std::vector<const suamd_complex *> gx; std::vector<suamd_complex *> gy; std::vector<SUSCOUNT> gl;
  std::vector<suamd_agc_bank_t *> sgb; std::vector<const suamd_complex *> sgx; std::vector<suamd_complex *> sgy; std::vector<SUSCOUNT> sgl;
  for (size_t i = 0; i < live.size(); ++i) {
    Inspector &in = *live[i];
    Route &r = rt[i];
    const suamd_complex *cur = in.d_y;
    auto other = [&](const suamd_complex *p) { return p == in.d_a ? in.d_z : in.d_a; };
    r.agc_in = cur; r.agc_out = nullptr;
    if (in.agc || in.fixed_gain > 0) { r.agc_out = in.d_a; cur = in.d_a; }
    r.car_in = cur; r.car_out = nullptr;
    if (in.costas || in.nco || in.pll || in.quad) { r.car_out = other(cur); cur = r.car_out; }
    r.mf_in = cur; r.mf_out = nullptr;
    if (in.mf) { r.mf_out = other(cur); cur = r.mf_out; }
    r.clk_in = cur;
    in.pend_src = cur;
    if (in.agc && in.in_slab) { sgb.push_back(in.agc); sgx.push_back(r.agc_in); sgy.push_back(r.agc_out); sgl.push_back(in.pend_m); }
    else if (in.agc) { gb.push_back(in.agc); gx.push_back(r.agc_in); gy.push_back(r.agc_out); gl.push_back(in.pend_m); }
    if (in.clock) in.pend_symbols = true; else in.pend_samples = true;
  }
  if (!gb.empty() && !suamd_agc_gang_pre(a->ctx, gb.data(), (unsigned)gb.size(), gx.data(), gl.data(), sA)) fail("gain control");
  // (a slab inspector's gain control always reads its column of y and writes its column of a)
  if (!sgb.empty() && !suamd_agc_gang_pre_slab(a->ctx, sgb.data(), (unsigned)sgb.size(), sb.y, pitch, sgx.data(), sgl.data(), sb.work, a->slab_rows, sA)) fail("gain control");
  if (a->trace) (void)hipEventRecord(a->ev_tpre, sA);
  for (int j = 0; j < P; ++j) {
    // ---- gain control on sA ----
    {
      std::vector<SUSCOUNT> m0, m1, sm0, sm1;
      for (size_t i = 0; i < live.size(); ++i) {
        Inspector &in = *live[i];
        if (in.agc && in.in_slab) { sm0.push_back(sub(in, j)); sm1.push_back(sub(in, j + 1)); }
        else if (in.agc) { m0.push_back(sub(in, j)); m1.push_back(sub(in, j + 1)); }
        else if (in.fixed_gain > 0 && sub(in, j + 1) > sub(in, j)) {
          const suamd_view row = {(SUSCOUNT)in.cap, (SUSCOUNT)in.ts};
          suamd_rows_scale(a->ctx, rt[i].agc_in + sub(in, j) * in.ts, row, rt[i].agc_out + sub(in, j) * in.ts, row, 1, sub(in, j + 1) - sub(in, j), in.fixed_gain, sA);
        }
      }
      if (!gb.empty()) {
        if (!suamd_agc_gang_level(a->ctx, gb.data(), (unsigned)gb.size(), gl.data(), m0.data(), m1.data(), sA) ||
            !suamd_agc_gang_apply(a->ctx, gb.data(), (unsigned)gb.size(), gx.data(), gy.data(), gl.data(), m0.data(), m1.data(), sA)) fail("gain control");
      }
      if (!sgb.empty()) {
        if (!suamd_agc_gang_level_slab(a->ctx, sgb.data(), (unsigned)sgb.size(), sb.y, pitch, sgx.data(), sgl.data(), sm0.data(), sm1.data(), sb.work, a->slab_rows, sA) ||
            !suamd_agc_gang_apply_slab(a->ctx, sgb.data(), (unsigned)sgb.size(), sb.y, pitch, sgx.data(), sb.a, pitch, sgy.data(), sgl.data(), sm0.data(), sm1.data(),
                                       sb.work, a->slab_rows, sA)) fail("gain control");
      }
      (void)hipEventRecord(a->ev_stage[0][j], sA);
      if (a->trace) (void)hipEventRecord(a->ev_tstage[0][j], sA);
    }
    // ---- carrier control on sC ----
    {
      (void)hipStreamWaitEvent(sC, a->ev_stage[0][j], 0);
      std::vector<suamd_costas_bank_t *> cb, scb; std::vector<const suamd_complex *> cx, scx; std::vector<suamd_complex *> cy, scy; std::vector<SUSCOUNT> cl, scl;
      std::vector<suamd_pll_bank_t *> pb, spb; std::vector<const suamd_complex *> px, spx; std::vector<suamd_complex *> py, spy; std::vector<SUSCOUNT> pl, spl;
      for (size_t i = 0; i < live.size(); ++i) {
        Inspector &in = *live[i];
        const SUSCOUNT b0 = sub(in, j), n = sub(in, j + 1) - b0;
        if (!rt[i].car_out || n == 0) continue;
        const suamd_view row = {(SUSCOUNT)in.cap, (SUSCOUNT)in.ts};
        const suamd_complex *xi = rt[i].car_in + b0 * in.ts;
        suamd_complex *yo = rt[i].car_out + b0 * in.ts;
        if (in.costas && in.in_slab) { scb.push_back(in.costas); scx.push_back(xi); scy.push_back(yo); scl.push_back(n); }
        else if (in.costas) { cb.push_back(in.costas); cx.push_back(xi); cy.push_back(yo); cl.push_back(n); }
        else if (in.pll && in.in_slab) { spb.push_back(in.pll); spx.push_back(xi); spy.push_back(yo); spl.push_back(n); }
        else if (in.pll) { pb.push_back(in.pll); px.push_back(xi); py.push_back(yo); pl.push_back(n); }
        else if (in.nco) suamd_nco_bank_feed(in.nco, xi, row, yo, row, n, sC);
        else if (in.quad) {
          // the sample before a later sub-range is still in the row; only the block's first one needs the carry
          suamd_quad_demod_batch(a->ctx, xi, row, yo, row, 1, n, b0 ? xi - in.ts : in.d_prev, in.first ? SU_TRUE : SU_FALSE, nullptr, sC);
          if (b0 + n == in.pend_m) (void)hipMemcpyAsync(in.d_prev, xi + (n - 1) * in.ts, 8, hipMemcpyDeviceToDevice, sC);
          in.first = false;
        }
      }
      if (!cb.empty() && !suamd_costas_gang_feed(a->ctx, cb.data(), (unsigned)cb.size(), cx.data(), cy.data(), cl.data(), sC)) fail("carrier control");
      if (!pb.empty() && !suamd_pll_gang_feed(a->ctx, pb.data(), (unsigned)pb.size(), px.data(), py.data(), pl.data(), sC)) fail("carrier control");
      // (the three complex slabs of a slot are one allocation: whichever of them an item reads and writes, a call's items are
      // near each other -- while one slab stays below 4 GiB / 3; beyond that the items go out slab pair by slab pair)
      const size_t slab_bytes = a->slab_rows * a->slab_pitch * sizeof(suamd_complex);
      const bool by_pair = 3 * slab_bytes >= SUAMD_SLAB_NEAR_BYTES;
      auto which = [&](const suamd_complex *p) { return p >= sb.z ? 2 : (p >= sb.a ? 1 : 0); };
      auto gangs_by_pair = [&](auto &banks, auto &xs, auto &ys, auto &ls, auto feed) {
        if (banks.empty()) return true;
        if (!by_pair) return feed(banks, xs, ys, ls);
        bool ok = true;
        for (int key = 0; key < 9; ++key) {
          std::remove_reference_t<decltype(banks)> b2; std::remove_reference_t<decltype(xs)> x2; std::remove_reference_t<decltype(ys)> y2; std::remove_reference_t<decltype(ls)> l2;
          for (size_t q = 0; q < banks.size(); ++q)
            if (which(xs[q]) * 3 + which(ys[q]) == key) { b2.push_back(banks[q]); x2.push_back(xs[q]); y2.push_back(ys[q]); l2.push_back(ls[q]); }
          if (!b2.empty()) ok = feed(b2, x2, y2, l2) && ok;
        }
        return ok;
      };
      if (!gangs_by_pair(scb, scx, scy, scl, [&](auto &b, auto &x, auto &y, auto &l) {
            return suamd_costas_gang_feed_slab(a->ctx, b.data(), (unsigned)b.size(), x.data(), pitch, y.data(), pitch, l.data(), sC) != SU_FALSE; })) fail("carrier control");
      if (!gangs_by_pair(spb, spx, spy, spl, [&](auto &b, auto &x, auto &y, auto &l) {
            return suamd_pll_gang_feed_slab(a->ctx, b.data(), (unsigned)b.size(), x.data(), pitch, y.data(), pitch, l.data(), sC) != SU_FALSE; })) fail("carrier control");
      (void)hipEventRecord(a->ev_stage[1][j], sC);
      if (a->trace) (void)hipEventRecord(a->ev_tstage[1][j], sC);
    }
    // ---- matched filter, clock recovery on sK ----
    {
      (void)hipStreamWaitEvent(sK, a->ev_stage[1][j], 0);
      std::vector<suamd_clock_bank_t *> kb, skb; std::vector<const suamd_complex *> kx, skx; std::vector<SUSCOUNT> kl, skl;
      std::vector<suamd_complex *> ks, sks; std::vector<uint32_t *> kc, skc;
      for (size_t i = 0; i < live.size(); ++i) {
        Inspector &in = *live[i];
        const SUSCOUNT b0 = sub(in, j), n = sub(in, j + 1) - b0;
        if (n == 0) continue;
        const suamd_view row = {(SUSCOUNT)in.cap, (SUSCOUNT)in.ts};
        if (in.mf) suamd_fir_bank_feed(in.mf, rt[i].mf_in + b0 * in.ts, row, rt[i].mf_out + b0 * in.ts, row, n, sK);
        if (in.clock && in.in_slab) { skb.push_back(in.clock); skx.push_back(rt[i].clk_in + b0 * in.ts); skl.push_back(n); sks.push_back(in.d_sym); skc.push_back(in.d_count); }
        else if (in.clock) { kb.push_back(in.clock); kx.push_back(rt[i].clk_in + b0); kl.push_back(n); ks.push_back(in.d_sym); kc.push_back(in.d_count); }
      }
      if (!kb.empty() && !suamd_clock_gang_feed(a->ctx, kb.data(), (unsigned)kb.size(), kx.data(), kl.data(), ks.data(), kc.data(), sK)) fail("clock recovery");
      {
        const size_t slab_bytes = a->slab_rows * a->slab_pitch * sizeof(suamd_complex);
        const bool by_slab = 3 * slab_bytes >= SUAMD_SLAB_NEAR_BYTES;    // (as for the carrier gangs: inputs from one slab per call then)
        for (int key = 0; key < (by_slab ? 3 : 1) && !skb.empty(); ++key) {
          std::vector<suamd_clock_bank_t *> b2; std::vector<const suamd_complex *> x2; std::vector<SUSCOUNT> l2; std::vector<suamd_complex *> s2; std::vector<uint32_t *> c2;
          for (size_t q = 0; q < skb.size(); ++q)
            if (!by_slab || (skx[q] >= sb.z ? 2 : (skx[q] >= sb.a ? 1 : 0)) == key) { b2.push_back(skb[q]); x2.push_back(skx[q]); l2.push_back(skl[q]); s2.push_back(sks[q]); c2.push_back(skc[q]); }
          if (!b2.empty() && !suamd_clock_gang_feed_slab(a->ctx, b2.data(), (unsigned)b2.size(), x2.data(), pitch, l2.data(), s2.data(), c2.data(), sK)) fail("clock recovery");
        }
      }
      (void)hipEventRecord(a->ev_stage[2][j], sK);
      if (a->trace) (void)hipEventRecord(a->ev_tstage[2][j], sK);
    }
  }
  // ---- tails: AGC state carry on sA; equalizers and the symbol counts on sK ----
  if (!gb.empty() && !suamd_agc_gang_finish(a->ctx, gb.data(), (unsigned)gb.size(), gx.data(), gl.data(), sA)) fail("gain control");
  if (!sgb.empty() && !suamd_agc_gang_finish_slab(a->ctx, sgb.data(), (unsigned)sgb.size(), sb.y, pitch, sgx.data(), sgl.data(), sb.work, a->slab_rows, sA)) fail("gain control");
  {
    std::vector<suamd_cma_bank_t *> eq; std::vector<const suamd_complex *> ex; std::vector<suamd_complex *> ey; std::vector<const uint32_t *> ec;
    for (Inspector *pi : live) {
      Inspector &in = *pi;
      if (in.pend_symbols && in.cma) { eq.push_back(in.cma); ex.push_back(in.d_sym); ey.push_back(in.d_sym); ec.push_back(in.d_count); }
    }
    // the equalizers take their symbol counts from the device: no host round trip inside the chain
    if (!eq.empty() && !suamd_cma_gang_feed(a->ctx, eq.data(), (unsigned)eq.size(), ex.data(), ec.data(), nullptr, ey.data(), sK)) fail("equalizer");
    // "power" inspectors: the block's channel samples fold into their integration windows
    for (Inspector *pi : live) {
      Inspector &in = *pi;
      if (!in.power) continue;
      SUSCOUNT k = 0;
      if (!suamd_power_bank_feed(in.power, in.d_y, in.pend_m, in.d_sym, &k, sK)) fail("power");
      in.pend_src = in.d_sym;
      in.pend_m = k;
    }
    // "audio" inspectors: demodulate + resample what the (optional) gain control left
    for (Inspector *pi : live) {
      Inspector &in = *pi;
      if (!in.audio) continue;
      SUSCOUNT k = 0;
      if (!suamd_audio_feed(in.audio, in.pend_src, in.pend_m, in.d_sym, &k, sK)) fail("audio");
      in.pend_src = in.d_sym;
      in.pend_m = k;
    }
    // hand-off: every inspector's batch goes to its mapped landing zone in one launch (sK is downstream of all stages)
    std::vector<const suamd_complex *> src; std::vector<uint32_t *> cnt; std::vector<SUSCOUNT> fixed, stride;
    std::vector<suamd_complex *> dst; std::vector<uint32_t *> cout;
    for (Inspector *pi : live) {
      Inspector &in = *pi;
      src.push_back(in.pend_symbols ? in.d_sym : in.pend_src);
      // (symbol rows, and what power / audio left in d_sym, are contiguous; a slab inspector's sample stream is its column)
      stride.push_back(in.pend_symbols || in.pend_src == in.d_sym ? 1 : (SUSCOUNT)in.ts);
      cnt.push_back(in.pend_symbols ? in.d_count : nullptr);
      fixed.push_back(in.pend_symbols ? 0 : in.pend_m);
      dst.push_back(in.h_out);
      cout.push_back(&in.pin->count);
    }
    if (!suamd_rows_deliver_strided(a->ctx, (unsigned)live.size(), src.data(), stride.data(), cnt.data(), fixed.data(), dst.data(), cout.data(), sK)) fail("hand-off");
    if (a->trace) (void)hipEventRecord(a->ev_tdone, sK);
  }
}

// after the PSD message of the block in `slot` went out: wait for that block's work (its events -- the streams may
// already hold the next block) and turn the results into messages
void collect_inspectors(suscan_analyzer *a, int slot)
{
  bool any = false;
  for (auto &kv : a->inspectors) {
    kv.second->recall(slot);
    any = any || kv.second->pend_samples || kv.second->pend_spectrum || kv.second->pend_symbols || kv.second->est_on[0] || kv.second->est_on[1] || kv.second->est_on[2];
  }
  if (!any) return;
  for (int k = 0; k < suscan_analyzer::NISTREAMS; ++k) (void)hipEventSynchronize(a->ev_done[slot][k]);
  if (a->trace) a->t_chains_done = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - a->t_block0).count();
  for (auto &kv : a->inspectors) {
    Inspector &in = *kv.second;
    if (in.pend_spectrum) {
      auto *msg = new_insp_msg(SUSCAN_ANALYZER_INSPECTOR_MSGKIND_SPECTRUM, 0);
      msg->handle = in.handle;
      msg->inspector_id = in.inspector_id;
      msg->spectsrc_id = in.spectsrc_id;
      msg->spectrum_size = in.pend_spec_n;
      msg->samp_rate = (SUSCOUNT)in.equiv_fs;
      msg->spectrum_data = static_cast<SUFLOAT *>(std::malloc(in.pend_spec_n * sizeof(SUFLOAT)));
      std::memcpy(msg->spectrum_data, in.pin->spec, in.pend_spec_n * sizeof(float));
      push(a, SUSCAN_ANALYZER_MESSAGE_TYPE_INSPECTOR, msg);
    }
    for (int k = 0; k < Inspector::NEST; ++k) {
      if (!in.est_on[k]) continue;
      auto *m = new_insp_msg(SUSCAN_ANALYZER_INSPECTOR_MSGKIND_ESTIMATOR, 0);
      m->handle = in.handle;
      m->inspector_id = in.inspector_id;
      m->estimator_id = (uint32_t)k;
      m->enabled = SU_TRUE;
      m->value = in.est_fed[k] ? in.pin->est[k] * (SUFLOAT)in.equiv_fs : in.last_est[k];   // Hz, what clock.baud / afc.offset take
      in.last_est[k] = m->value;
      push(a, SUSCAN_ANALYZER_MESSAGE_TYPE_INSPECTOR, m);
    }
    if (in.pend_symbols || in.pend_samples) emit_samples(a, in, in.pin->count);
    in.pend_samples = in.pend_spectrum = in.pend_symbols = false;
  }
}

const struct suscan_estimator_class kEstimators[3] = {
  {"baud-fac", "Fast autocorrelation baud estimator", "clock.baud"},
  {"baud-nonlinear", "Non-linear baud estimator", "clock.baud"},
  {"carrier", "Carrier offset estimator (spectral centroid)", "afc.offset"},
};

void push_source_info(suscan_analyzer *a)
{
  auto *si = static_cast<suscan_source_info *>(std::malloc(sizeof(suscan_source_info)));
  {
    std::lock_guard<std::mutex> lk(a->req_m);               // the setters write a->info under req_m
    (void)suscan_source_info_init_copy(si, &a->info);
  }
  push(a, SUSCAN_ANALYZER_MESSAGE_TYPE_SOURCE_INFO, si);
}

void handle_request(suscan_analyzer *a, Request &r)
{
  if (r.kind == Request::SOURCE_INFO) { push_source_info(a); return; }
  switch (r.kind) {
    case Request::OPEN: {
      const suscan_config_desc_t *desc = suscan_inspector_config_desc(r.cls.c_str());
      if (!desc) {
        auto *m = new_insp_msg(SUSCAN_ANALYZER_INSPECTOR_MSGKIND_WRONG_KIND, r.req_id);
        push(a, SUSCAN_ANALYZER_MESSAGE_TYPE_INSPECTOR, m);
        return;
      }
      const double fs = a->source_cfg.samp_rate;
      if (!(r.channel.bw > 0) || std::fabs(r.channel.fc) > fs / 2 || r.channel.bw > fs) {
        auto *m = new_insp_msg(SUSCAN_ANALYZER_INSPECTOR_MSGKIND_INVALID_CHANNEL, r.req_id);
        push(a, SUSCAN_ANALYZER_MESSAGE_TYPE_INSPECTOR, m);
        return;
      }
      auto in = std::make_unique<Inspector>();
      in->handle = a->next_handle;                            // shard s hands out s, s + G, s + 2G ...: a handle names its shard
      a->next_handle += a->nshards;
      in->cls = r.cls;
      in->channel = r.channel;
      in->precise = r.precise;
      in->config = suscan_config_new(desc);
      std::string err;
      if (!build_chain(a, *in, err)) {
        push_status(a, SUSCAN_ANALYZER_MESSAGE_TYPE_INTERNAL, -1, err);
        in->free_all();
        return;
      }
      in->dirty = false;
      auto *m = new_insp_msg(SUSCAN_ANALYZER_INSPECTOR_MSGKIND_OPEN, r.req_id);
      m->handle = in->handle;
      m->class_name = dupstr(r.cls.c_str());
      m->channel = r.channel;
      m->config = suscan_config_dup(in->config);
      m->fs = (uint32_t)fs;
      m->equiv_fs = (SUFLOAT)in->equiv_fs;
      m->bandwidth = (SUFLOAT)r.channel.bw;
      m->lo = (SUFLOAT)in->fnor;
      m->spectsrc_count = suamd_spectsrc_count();             // names borrowed from the library (static storage)
      m->spectsrc_list = static_cast<char **>(std::calloc(m->spectsrc_count, sizeof(char *)));
      for (unsigned k = 0; k < m->spectsrc_count; ++k) m->spectsrc_list[k] = const_cast<char *>(suamd_spectsrc_name(k + 1));
      if (r.cls != "raw" && r.cls != "power" && r.cls != "audio") {   // the baud estimators (names static, like the sources')
        m->estimator_count = 3;
        m->estimator_list = static_cast<char **>(std::calloc(3, sizeof(char *)));
        for (int k = 0; k < 3; ++k) m->estimator_list[k] = const_cast<char *>(kEstimators[k].name);
      }
      a->inspectors[in->handle] = std::move(in);
      push(a, SUSCAN_ANALYZER_MESSAGE_TYPE_INSPECTOR, m);
      return;
    }
    default: break;
  }
  auto it = a->inspectors.find(r.handle);
  if (r.kind != Request::SET_PARAMS && r.kind != Request::SET_THROTTLE && r.kind != Request::SEEK && it == a->inspectors.end()) {
    auto *m = new_insp_msg(SUSCAN_ANALYZER_INSPECTOR_MSGKIND_WRONG_HANDLE, r.req_id);
    m->handle = r.handle;
    push(a, SUSCAN_ANALYZER_MESSAGE_TYPE_INSPECTOR, m);
    if (r.config) suscan_config_destroy(r.config);
    return;
  }
  switch (r.kind) {
    case Request::CLOSE: {
      auto *m = new_insp_msg(SUSCAN_ANALYZER_INSPECTOR_MSGKIND_CLOSE, r.req_id);
      m->handle = r.handle;
      m->inspector_id = it->second->inspector_id;
      flush_watermark(a, *it->second);                         // what had not filled a batch goes out before the CLOSE
      it->second->free_all();
      a->inspectors.erase(it);
      push(a, SUSCAN_ANALYZER_MESSAGE_TYPE_INSPECTOR, m);
      break;
    }
    case Request::SET_ID: {
      it->second->inspector_id = r.inspector_id;
      auto *m = new_insp_msg(SUSCAN_ANALYZER_INSPECTOR_MSGKIND_SET_ID, r.req_id);
      m->handle = r.handle;
      m->inspector_id = r.inspector_id;
      push(a, SUSCAN_ANALYZER_MESSAGE_TYPE_INSPECTOR, m);
      break;
    }
    case Request::SET_CONFIG: {
      Inspector &in = *it->second;
      // the chain behind the channel is rebuilt at the next block: what the old one has not delivered yet (a watermark's
      // unfilled batch) goes out BEFORE the acknowledgement -- a SAMPLES batch never mixes two configurations (found by
      // tests/test_gpu_analyzer_fuzz.py: the remainder used to surface inside the new chain's first batch)
      flush_watermark(a, in);
      // copy the values of every field the inspector knows
      for (unsigned i = 0; r.config && i < r.config->desc->field_count; ++i) {
        const suscan_field_value *v = r.config->values[i];
        suscan_field_value *dst = suscan_config_get_value(in.config, v->field->name);
        if (dst && dst->field->type == v->field->type) { dst->as_int = v->as_int; dst->set = SU_TRUE; }
      }
      in.dirty = true;
      auto *m = new_insp_msg(SUSCAN_ANALYZER_INSPECTOR_MSGKIND_SET_CONFIG, r.req_id);
      m->handle = r.handle;
      m->inspector_id = in.inspector_id;
      m->config = suscan_config_dup(in.config);
      push(a, SUSCAN_ANALYZER_MESSAGE_TYPE_INSPECTOR, m);
      if (r.config) suscan_config_destroy(r.config);
      break;
    }
    case Request::SET_WATERMARK: {
      it->second->watermark = r.value;
      auto *m = new_insp_msg(SUSCAN_ANALYZER_INSPECTOR_MSGKIND_SET_WATERMARK, r.req_id);
      m->handle = r.handle;
      m->watermark = r.value;
      push(a, SUSCAN_ANALYZER_MESSAGE_TYPE_INSPECTOR, m);
      break;
    }
    case Request::SET_SPECTRUM: {
      const bool ok = r.value <= suamd_spectsrc_count();
      if (ok) { it->second->spectsrc_id = (uint32_t)r.value; it->second->spect_have_prev = false; }
      auto *m = new_insp_msg(ok ? SUSCAN_ANALYZER_INSPECTOR_MSGKIND_SPECTRUM : SUSCAN_ANALYZER_INSPECTOR_MSGKIND_INVALID_ARGUMENT, r.req_id);
      m->handle = r.handle;
      m->spectsrc_id = (uint32_t)r.value;
      push(a, SUSCAN_ANALYZER_MESSAGE_TYPE_INSPECTOR, m);   // acknowledgement: no spectrum_data yet
      break;
    }
    // A retune cuts the watermark's current batch short: the remainder goes out as it is, so that no SAMPLES message ever mixes
    // samples of two tunings (the one place a batch is shorter than the watermark besides CLOSE / EOS; a retune has no reply
    // message, the short batch is the client's only marker).  Round 6 tried carrying the remainder over: the
    // fuzz test's replay (tests/test_gpu_analyzer_fuzz.py) showed what that means -- up to w - 1 samples of the OLD frequency
    // at the head of the first batch after the retune -- and it was taken back.
    case Request::SET_FREQ: flush_watermark(a, *it->second); it->second->channel.fc = r.fvalue; it->second->dirty = true; break;
    case Request::SET_BW:   flush_watermark(a, *it->second); it->second->channel.bw = (SUFLOAT)r.fvalue; it->second->dirty = true; break;   // (another sample rate: see SET_CONFIG)
    case Request::SET_PARAMS: {
      // only the PSD parameters matter on this path; applied at the next block boundary by the worker
      a->params = r.params;
      auto *m = static_cast<suscan_analyzer_params *>(std::malloc(sizeof(suscan_analyzer_params)));
      *m = a->params;
      push(a, SUSCAN_ANALYZER_MESSAGE_TYPE_PARAMS, m);
      break;
    }
    case Request::SET_THROTTLE: a->throttle = r.value; break;
    case Request::ESTIMATOR: {
      Inspector &in = *it->second;
      if (r.value >= Inspector::NEST || in.cls == "raw" || in.cls == "power") {
        auto *m = new_insp_msg(SUSCAN_ANALYZER_INSPECTOR_MSGKIND_WRONG_OBJECT, r.req_id);
        m->handle = r.handle;
        push(a, SUSCAN_ANALYZER_MESSAGE_TYPE_INSPECTOR, m);
        break;
      }
      in.est_on[r.value] = r.fvalue != 0;
      auto *m = new_insp_msg(SUSCAN_ANALYZER_INSPECTOR_MSGKIND_ESTIMATOR, r.req_id);   // acknowledgement: no value yet
      m->handle = r.handle;
      m->inspector_id = in.inspector_id;
      m->estimator_id = (uint32_t)r.value;
      m->enabled = in.est_on[r.value] ? SU_TRUE : SU_FALSE;
      push(a, SUSCAN_ANALYZER_MESSAGE_TYPE_INSPECTOR, m);
      break;
    }
    case Request::SET_TLE: {
      // Doppler correction from orbital elements needs an orbit propagator: outside this path
      auto *m = new_insp_msg(r.fvalue != 0 ? SUSCAN_ANALYZER_INSPECTOR_MSGKIND_INVALID_ARGUMENT : SUSCAN_ANALYZER_INSPECTOR_MSGKIND_SET_TLE,
                             r.req_id);
      m->handle = r.handle;
      m->inspector_id = it->second->inspector_id;
      m->enabled = SU_FALSE;
      m->tle_enable = SU_FALSE;
      push(a, SUSCAN_ANALYZER_MESSAGE_TYPE_INSPECTOR, m);
      break;
    }
    default: break;
  }
}

void bus_wait_done(suscan_analyzer *a, uint64_t upto);

// (re)creates the channel detector from the analyzer parameters (window size, alpha / beta / gamma / snr, channel_update_int)
void setup_chandet(suscan_analyzer *a)
{
  const unsigned n = (unsigned)a->params.detector_params.window_size;
  const size_t block = (size_t)n * a->navg;
  if (a->chandet) { suamd_chandet_destroy(a->chandet); a->chandet = nullptr; }
  {
    const auto &dp = a->params.detector_params;
    if (a->params.mode == SUSCAN_ANALYZER_MODE_CHANNEL && n >= 512 && n <= 16384 && dp.alpha > 0 && dp.alpha <= 1 &&
        dp.gamma > 0 && dp.gamma <= 1 && dp.snr > 0 && a->params.channel_update_int > 0)
      a->chandet = suamd_chandet_new(a->ctx, n, dp.alpha, dp.beta, dp.gamma, dp.snr);    // (out-of-range parameters: no lists)
    const double blocks = (double)a->params.channel_update_int * a->source_cfg.samp_rate / (double)block;
    a->chan_every = blocks < 1 ? 1u : (unsigned)(blocks + 0.5);
    a->chan_phase = 0;
  }
}

bool setup_psd(suscan_analyzer *a, std::string &err)
{
  if (a->psd) { suamd_psd_destroy(a->psd); a->psd = nullptr; }
  const unsigned n = (unsigned)a->params.detector_params.window_size;
  a->psd = suamd_psd_new(a->ctx, n, a->params.detector_params.window);
  if (!a->psd) { err = suamd_last_error(); return false; }
  const double per = a->params.psd_update_int > 0 ? a->params.psd_update_int : 0.04;
  double frames = std::floor(a->source_cfg.samp_rate * per / n + 0.5);
  if (frames < 1) frames = 1;
  if (frames > 4096) frames = 4096;
  a->navg = (unsigned)frames;
  const size_t block = (size_t)n * a->navg;
  for (int k = 0; k < 2; ++k) {                              // the PSD landing zones follow the frame size
    if (a->h_psd[k]) (void)hipHostFree(a->h_psd[k]);
    a->h_psd[k] = nullptr;
    if (hipHostMalloc((void **)&a->h_psd[k], n * sizeof(float), hipHostMallocDefault) != hipSuccess) { err = "pinned allocation failed"; return false; }
  }
  if (a->d_psd) (void)hipFree(a->d_psd);
  a->d_psd = nullptr;
  if (hipMalloc((void **)&a->d_psd, n * sizeof(float)) != hipSuccess) { err = "device allocation failed"; return false; }
  if (block != a->block) {
    if (a->bus) bus_wait_done(a, a->bus->seq);               // no shard reads the old host buffers any more
    if (a->h_x) (void)hipHostFree(a->h_x);
    if (a->h_flt) (void)hipHostFree(a->h_flt);               // the filters' expansion buffer is a block long too
    a->h_flt = nullptr;
    if (a->d_x) (void)hipFree(a->d_x);
    if (a->d_raw) (void)hipFree(a->d_raw);
    a->h_x = nullptr; a->d_x = nullptr; a->d_raw = nullptr;
    a->h2d_set[0] = a->h2d_set[1] = false; a->xfree_set = false;
    if (hipHostMalloc((void **)&a->h_x, 2 * block * sizeof(suamd_complex), hipHostMallocPortable) != hipSuccess ||   // two halves
        hipMalloc((void **)&a->d_x, block * sizeof(suamd_complex)) != hipSuccess ||
        hipMalloc(&a->d_raw, block * 4) != hipSuccess) {
      err = "allocation of the block buffers failed";
      return false;
    }
    a->block = block;
    for (auto &kv : a->inspectors) kv.second->dirty = true;      // buffer capacities depend on the block
  }
  setup_chandet(a);
  const bool fft = a->want_fft && block % 2048 == 0;             // whole half windows of the 4096-point filter bank per block
  if (fft != a->use_fft) {
    for (auto &kv : a->inspectors) { kv.second->free_chain(); kv.second->dirty = true; }
    a->use_fft = fft;
    if (!fft && a->st) { suamd_specttuner_destroy(a->st); a->st = nullptr; }
  }
  return true;
}

// The next block comes off the source on a helper thread while the worker issues the current one: reading 16 MB out
// of the page cache takes longer (1.6 ms) than everything else the worker does for a block.  One read at a time; the
// worker touches the source itself (mark, rewind, seek) only while no read is pending.
struct AsyncRead {
  Source &src;
  std::thread th;
  std::mutex m;
  std::condition_variable cv;
  void *dst = nullptr; size_t want = 0, got = 0; bool looped = false, pending = false, quit = false;
  explicit AsyncRead(Source &s) : src(s), th([this] { run(); }) {}
  ~AsyncRead()
  {
    { std::lock_guard<std::mutex> lk(m); quit = true; }
    cv.notify_all();
    th.join();
  }
  void run()
  {
    std::unique_lock<std::mutex> lk(m);
    for (;;) {
      cv.wait(lk, [this] { return pending || quit; });
      if (quit) return;
      void *d = dst; const size_t w = want;
      lk.unlock();
      bool lp = false;
      const size_t g = src.read(d, w, &lp);
      lk.lock();
      got = g; looped = lp; pending = false;
      cv.notify_all();
    }
  }
  void start(void *d, size_t w)
  {
    { std::lock_guard<std::mutex> lk(m); dst = d; want = w; pending = true; }
    cv.notify_all();
  }
  size_t wait(bool *lp)
  {
    std::unique_lock<std::mutex> lk(m);
    cv.wait(lk, [this] { return !pending; });
    *lp = looped;
    return got;
  }
};

// WIDE_SPECTRUM analyzers (Panoramic/Scanner.cpp:295-370): the tuner hops over [hop_min, hop_max] and every PSD frame
// carries the frequency it was taken at (Scanner::onPSDMessage feeds it to the SpectrumView there, :503-523).  A
// file / generator source has no tuner: the capture IS the sweep -- consecutive blocks are consecutive dwells -- and the
// analyzer labels them the way the sweep strategy visits the range (SPEC.md section P):
//   step = rel_bw * fs, nsteps = max(1, ceil((max - min) / step));
//   PROGRESSIVE: dwell k at min + (k mod nsteps + 1/2) step;  STOCHASTIC: a pseudo-random slot (DISCRETE partitioning:
//   the same slot centres; CONTINUOUS: anywhere in the range);  min == max (the Scanner's noHop): that frequency.
// Called under req_m.
double next_hop_frequency(suscan_analyzer *a)
{
  double lo = a->hop_min, hi = a->hop_max;
  if (!(hi > lo)) { lo = a->params.min_freq; hi = a->params.max_freq; }     // set_hop_range not called yet: the analyzer parameters
  const uint64_t k = a->hop_k++;
  if (!(hi > lo)) return hi == lo && hi > 0 ? hi : a->source_cfg.freq;
  const double step = (double)a->rel_bw * (double)a->source_cfg.samp_rate;
  const uint64_t nsteps = std::max<uint64_t>(1, (uint64_t)std::ceil((hi - lo) / step - 1e-9));
  if (a->sweep_strategy == SUSCAN_ANALYZER_SWEEP_STRATEGY_PROGRESSIVE) return lo + ((double)(k % nsteps) + 0.5) * step;
  a->hop_lcg = a->hop_lcg * 1664525u + 1013904223u;
  const double u = (double)(a->hop_lcg >> 8) / 16777216.0;                    // [0, 1)
  if (a->partitioning == SUSCAN_ANALYZER_SPECTRUM_PARTITIONING_DISCRETE) return lo + ((double)(uint64_t)(u * (double)nsteps) + 0.5) * step;
  return lo + u * (hi - lo);
}

// CHANNEL message (Suscan/Analyzer.cpp:75-98 lets it through to ChannelMessage): the detector's list of the block in `slot`
void push_channels(suscan_analyzer *a, int slot)
{
  struct suamd_channel list[256];
  const int n = suamd_chandet_collect(a->chandet, slot, (SUFLOAT)a->source_cfg.samp_rate, list, 256);
  if (n < 0) return;
  auto *m = static_cast<suscan_analyzer_channel_msg *>(std::calloc(1, sizeof(suscan_analyzer_channel_msg)));
  m->channel_count = (unsigned)n;
  m->channel_list = static_cast<sigutils_channel **>(std::calloc(n ? n : 1, sizeof(void *)));
  double ft;
  { std::lock_guard<std::mutex> lk(a->req_m); ft = a->source_cfg.freq; }
  for (int k = 0; k < n; ++k) {
    auto *c = static_cast<sigutils_channel *>(std::calloc(1, sizeof(sigutils_channel)));
    c->fc = list[k].fc; c->f_lo = list[k].f_lo; c->f_hi = list[k].f_hi;      // relative to the tuner, like Analyzer::open's channels
    c->bw = list[k].bw; c->snr = list[k].snr; c->S0 = list[k].S0; c->N0 = list[k].N0;
    c->ft = ft; c->age = list[k].age; c->present = 1;
    m->channel_list[k] = c;
  }
  push(a, SUSCAN_ANALYZER_MESSAGE_TYPE_CHANNEL, m);
}

void free_device(suscan_analyzer *a)
{
  (void)hipDeviceSynchronize();
  for (auto &kv : a->inspectors) kv.second->free_all();
  a->inspectors.clear();
  a->rows.release();
  if (a->st) suamd_specttuner_destroy(a->st);
  a->st = nullptr;
  if (a->chandet) suamd_chandet_destroy(a->chandet);
  a->chandet = nullptr;
  for (int p = 0; p < 2; ++p) {
    if (a->d_rowptr[p]) (void)hipFree(a->d_rowptr[p]);
    if (a->h_rowptr[p]) (void)hipHostFree(a->h_rowptr[p]);
    a->d_rowptr[p] = a->h_rowptr[p] = nullptr;
  }
  a->rowptr_cap = 0;
  for (auto &sl : a->slab) {
    if (sl.y) (void)hipFree(sl.y);                          // (a and z are parts of the same allocation)
    if (sl.work) (void)hipFree(sl.work);
    sl = suscan_analyzer::SlabSet();
  }
  a->slab_rows = a->slab_pitch = 0;
  if (a->d_sink) (void)hipFree(a->d_sink);
  a->d_sink = nullptr;
  if (a->psd) suamd_psd_destroy(a->psd);
  for (int p = 0; p < 2; ++p) {
    if (a->h_psd[p]) (void)hipHostFree(a->h_psd[p]);
    a->h_psd[p] = nullptr;
    if (a->ev_psd[p]) (void)hipEventDestroy(a->ev_psd[p]);
    if (a->ev_h2d[p]) (void)hipEventDestroy(a->ev_h2d[p]);
    if (p == 0 && a->ev_bcast) { (void)hipEventDestroy(a->ev_bcast); a->ev_bcast = nullptr; }
    a->ev_psd[p] = a->ev_h2d[p] = nullptr;
    for (int k = 0; k < suscan_analyzer::NISTREAMS; ++k) { if (a->ev_done[p][k]) (void)hipEventDestroy(a->ev_done[p][k]); a->ev_done[p][k] = nullptr; }
  }
  if (a->ev_xfree) (void)hipEventDestroy(a->ev_xfree);
  if (a->ev_fir) (void)hipEventDestroy(a->ev_fir);
  a->ev_xfree = a->ev_fir = nullptr;
  if (a->h_x) (void)hipHostFree(a->h_x);
  if (a->h_flt) (void)hipHostFree(a->h_flt);
  if (a->d_dc) (void)hipFree(a->d_dc);
  a->h_flt = nullptr; a->d_dc = nullptr;
  if (a->d_x) (void)hipFree(a->d_x);
  if (a->d_raw) (void)hipFree(a->d_raw);
  if (a->d_psd) (void)hipFree(a->d_psd);
  if (a->stream) (void)hipStreamDestroy(a->stream);
  for (int k = 0; k < suscan_analyzer::NISTREAMS; ++k) { if (a->istream[k]) (void)hipStreamDestroy(a->istream[k]); a->istream[k] = nullptr; }
  if (a->ev_input) (void)hipEventDestroy(a->ev_input);
  a->ev_input = nullptr;
  for (int g = 0; g < 3; ++g)
    for (int j = 0; j < suscan_analyzer::NSUB; ++j) { if (a->ev_stage[g][j]) (void)hipEventDestroy(a->ev_stage[g][j]); a->ev_stage[g][j] = nullptr; }
  if (a->ctx) suamd_ctx_destroy(a->ctx);
  a->psd = nullptr; a->h_x = nullptr; a->d_x = nullptr; a->d_raw = nullptr; a->d_psd = nullptr; a->stream = nullptr; a->ctx = nullptr;
}

// streams, events and knobs of one shard, on its device (the calling thread stays bound to that device)
bool init_device(suscan_analyzer *a, std::string &err)
{
  a->ctx = suamd_ctx_new(a->device);
  bool ok = a->ctx != nullptr;
  if (!ok) err = suamd_last_error();
  if (ok && hipStreamCreate(&a->stream) != hipSuccess) { ok = false; err = "hipStreamCreate failed"; }
  // The three recurrence stages (gain control, carrier control, clock recovery) run concurrently on streams of their own, and
  // that only holds while their streams sit on different hardware queues: HIP deals a process's streams of one priority over
  // four queues by head count, so with streams of the host application (or of torch, in bench.py) already there two stages
  // could end up behind one queue -- measured 3.8-4.4 instead of 2.0 ms per block, depending on nothing but how many streams
  // the process had created before (tools/live_queue_probe.py).  Streams of another priority come out of another set of
  // queues: the three stage streams ask for SUAMD_ANALYZER_STAGE_PRIORITY (default: the highest), which nobody else uses.
  int prio_lo = 0, prio_hi = 0, prio = 0;
  bool prio_ok = hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi) == hipSuccess && prio_lo != prio_hi;
  if (!prio_ok) (void)hipGetLastError();
  prio = prio_hi;
  if (const long long v = sdk::tuning().analyzer_stage_priority; v == -2) prio_ok = false; else if (v != 99) prio = (int)v;
  for (int k = 0; ok && k < suscan_analyzer::NISTREAMS; ++k) {
    hipError_t e = (prio_ok && k < 3) ? hipStreamCreateWithPriority(&a->istream[k], hipStreamNonBlocking, prio)
                                      : hipStreamCreateWithFlags(&a->istream[k], hipStreamNonBlocking);
    if (e != hipSuccess) { ok = false; err = "hipStreamCreate failed"; }
  }
  if (ok && hipEventCreateWithFlags(&a->ev_input, hipEventDisableTiming) != hipSuccess) { ok = false; err = "hipEventCreate failed"; }
  a->trace = sdk::tuning().analyzer_trace != 0;
  if (const int v = (int)sdk::tuning().analyzer_subranges; v >= 1 && v <= suscan_analyzer::NSUB) a->nsub = a->nsub_env = v;   // 1 = whole block per stage
  a->slab_on = sdk::tuning().analyzer_slab != 0;            // (every narrow channel of a tuner goes one way: not re-read while inspectors are open)
  for (int g = 0; ok && g < 3; ++g)
    for (int j = 0; ok && j < suscan_analyzer::NSUB; ++j)
      if (hipEventCreateWithFlags(&a->ev_stage[g][j], hipEventDisableTiming) != hipSuccess) { ok = false; err = "hipEventCreate failed"; }
  if (ok) {
    bool e = hipEventCreateWithFlags(&a->ev_bcast, hipEventDisableTiming) == hipSuccess &&
             hipEventCreateWithFlags(&a->ev_xfree, hipEventDisableTiming) == hipSuccess &&
             hipEventCreateWithFlags(&a->ev_fir, hipEventDisableTiming) == hipSuccess;
    for (int p = 0; p < 2; ++p) {
      e = e && hipEventCreateWithFlags(&a->ev_psd[p], hipEventDisableTiming) == hipSuccess &&
          hipEventCreateWithFlags(&a->ev_h2d[p], hipEventDisableTiming) == hipSuccess;
      for (int k = 0; k < suscan_analyzer::NISTREAMS; ++k) e = e && hipEventCreateWithFlags(&a->ev_done[p][k], hipEventDisableTiming) == hipSuccess;
    }
    if (!e) { ok = false; err = "hipEventCreate failed"; }
    const char *ce = std::getenv("SUAMD_ANALYZER_CHANNELISER");
    a->want_fft = !(ce && !strcasecmp(ce, "fir"));
    a->pipelined = !a->trace && sdk::tuning().analyzer_pipeline != 0;
  }
  if (ok && a->trace) {
    (void)hipEventCreate(&a->ev_t0); (void)hipEventCreate(&a->ev_tfir); (void)hipEventCreate(&a->ev_tpre); (void)hipEventCreate(&a->ev_tdone);
    for (int g = 0; g < 3; ++g) for (int j = 0; j < suscan_analyzer::NSUB; ++j) (void)hipEventCreate(&a->ev_tstage[g][j]);
  }
  return ok;
}

void secondary_main(suscan_analyzer *a);

// ---- the block bus (multi-GPU) -------------------------------------------------------------------------------------
// shard 0, before it reuses host memory that block `upto - 1` lived in: every subscriber has copied blocks < upto
void bus_wait_done(suscan_analyzer *a, uint64_t upto)
{
  if (!a->bus || a->secondaries.empty()) return;
  BlockBus &b = *a->bus;
  std::unique_lock<std::mutex> lk(b.m);
  b.cv.wait(lk, [&] { for (uint64_t d : b.done) if (d < upto) return false; return true; });
}

bool bus_publish(suscan_analyzer *a, const void *host, size_t samples, size_t valid, int raw_format, unsigned bps, uint64_t position)
{
  if (!a->bus || a->secondaries.empty()) return false;
  BlockBus &b = *a->bus;
  {
    std::lock_guard<std::mutex> lk(b.m);
    BlockBus::Entry &e = b.e[b.seq & 1];
    e.host = host; e.samples = samples; e.valid = valid; e.raw_format = raw_format; e.bytes_per_sample = bps; e.position = position;
    e.samp_rate = a->source_cfg.samp_rate;
    // a collective needs every rank: if a shard has given up (its `done` is pinned at ~0) the broadcast is over for good
    for (uint64_t d : b.done) if (d == ~0ull) b.bcast_off = true;
    e.via_bcast = b.bcast != nullptr && !b.bcast_off;
    if (e.via_bcast) ++b.bcast_blocks;
    ++b.seq;
  }
  b.cv.notify_all();
  return b.e[(b.seq - 1) & 1].via_bcast;                     // (only the publisher writes the entries)
}

// A collective needs every rank.  A shard that dies between the publisher's decision ("this block goes out by broadcast")
// and its own ncclBroadcast call leaves the other ranks' calls on their streams with a peer that never comes: librccl's
// kernel spins for ever and every later synchronisation of that stream hangs with it.  So a stream is never waited for
// blindly behind a broadcast: the event recorded after the call is polled, and a DEADLINE RUNS ONLY WHILE A SHARD IS KNOWN TO
// BE GONE (its `done` pinned at ~0, or the bus closed): a peer that is merely slow -- building the chains of hundreds of
// inspectors on its first block, loading HIP modules, RCCL setting its channels up on the first collective -- is waited
// for as long as it takes (round 5 gave every block 2 s whatever the reason and declared a slow peer dead: ADVICE r5).
// `bcast_ceiling_ms` bounds the wait when nobody has reported anything (a peer wedged without saying so).  When the
// deadline runs out the rank aborts its communicator (ncclCommAbort releases the kernel), the bus switches to per-GPU host
// copies for good, and the dead shard is reported.  Returns false when the deadline passed.
bool wait_event_deadline(hipEvent_t ev, int timeout_ms)
{
  const auto t_end = std::chrono::steady_clock::now() + std::chrono::milliseconds(timeout_ms);
  for (int spins = 0;; ++spins) {
    const hipError_t q = hipEventQuery(ev);
    if (q == hipSuccess) return true;
    if (q != hipErrorNotReady) { (void)hipGetLastError(); return true; }   // a broken event is not a hung broadcast
    if (std::chrono::steady_clock::now() >= t_end) return false;
    if (spins < 2000) std::this_thread::yield(); else std::this_thread::sleep_for(std::chrono::microseconds(100));
  }
}

bool bcast_wait(suscan_analyzer *a, hipEvent_t ev)
{
  BlockBus &b = *a->bus;
  const auto t0 = std::chrono::steady_clock::now();
  bool armed = false;
  std::chrono::steady_clock::time_point t_armed;
  for (int spins = 0;; ++spins) {
    const hipError_t q = hipEventQuery(ev);
    if (q == hipSuccess) return true;
    if (q != hipErrorNotReady) { (void)hipGetLastError(); return true; }
    const auto now = std::chrono::steady_clock::now();
    if ((spins & 63) == 63 || spins < 4) {                      // (the mutex every 64th poll: ~6 ms once the polls sleep)
      bool lost = false;
      {
        std::lock_guard<std::mutex> lk(b.m);
        lost = b.closed;
        for (uint64_t d : b.done) lost = lost || d == ~0ull;
      }
      if (lost && !armed) { armed = true; t_armed = now; }
    }
    if (armed && now - t_armed >= std::chrono::milliseconds(b.bcast_timeout_ms)) return false;
    if (now - t0 >= std::chrono::milliseconds(b.bcast_ceiling_ms)) return false;
    if (spins < 2000) std::this_thread::yield(); else std::this_thread::sleep_for(std::chrono::microseconds(100));
  }
}

// this rank's broadcast did not complete: release its stream, stop broadcasting, say which shard is gone
void bus_broadcast_stuck(suscan_analyzer *a, hipEvent_t ev)
{
  BlockBus &b = *a->bus;
  void *comm = nullptr;
  std::vector<int> lost;
  {
    std::lock_guard<std::mutex> lk(b.m);
    b.bcast_off = true;
    if ((size_t)a->shard < b.comm.size()) { comm = b.comm[a->shard]; b.comm[a->shard] = nullptr; }
    if (b.lost_reported.size() < b.done.size()) b.lost_reported.resize(b.done.size(), false);
    for (size_t i = 0; i < b.done.size(); ++i)
      if (b.done[i] == ~0ull && !b.lost_reported[i]) { b.lost_reported[i] = true; lost.push_back((int)i + 1); }
  }
  if (comm && b.comm_abort) (void)b.comm_abort(comm);         // (the communicator is gone with it: never destroyed twice)
  if (!wait_event_deadline(ev, 20000))                         // without ncclCommAbort (an old library) the kernel cannot be released
    push_status(a, SUSCAN_ANALYZER_MESSAGE_TYPE_INTERNAL, -1, "GPU shard " + std::to_string(a->shard) + ": a stuck ncclBroadcast could not be aborted");
  bool any_gone = false;
  { std::lock_guard<std::mutex> lk(b.m); for (uint64_t d : b.done) any_gone = any_gone || d == ~0ull; }
  if (any_gone)
    push_status(a, SUSCAN_ANALYZER_MESSAGE_TYPE_INTERNAL, 0, "GPU shard " + std::to_string(a->shard) + ": ncclBroadcast did not complete within " +
                std::to_string(b.bcast_timeout_ms) + " ms of a shard going away (a rank is missing): aborted, per-GPU host copies from here on");
  else
    push_status(a, SUSCAN_ANALYZER_MESSAGE_TYPE_INTERNAL, 0, "GPU shard " + std::to_string(a->shard) + ": ncclBroadcast did not complete within " +
                std::to_string(b.bcast_ceiling_ms) + " ms although no shard has reported a failure: broadcast disabled, per-GPU host copies from here on");
  for (int sh : lost)
    push_status(a->primary ? a->primary : a, SUSCAN_ANALYZER_MESSAGE_TYPE_READ_ERROR, -1, "GPU shard " + std::to_string(sh) + " is gone: its inspectors deliver nothing any more");
}

void bus_close(suscan_analyzer *a)
{
  if (!a->bus) return;
  { std::lock_guard<std::mutex> lk(a->bus->m); a->bus->closed = true; }
  a->bus->cv.notify_all();
  for (suscan_analyzer *s : a->secondaries) { s->halt = true; if (s->worker.joinable()) s->worker.join(); }
  if (a->bus->rccl_lib) {                                      // every shard's thread is gone: the communicators can go
    (void)hipDeviceSynchronize();
    if (a->bus->comm_destroy) for (void *c : a->bus->comm) if (c) (void)a->bus->comm_destroy(c);
    a->bus->comm.clear(); a->bus->bcast = nullptr;
    dlclose(a->bus->rccl_lib);
    a->bus->rccl_lib = nullptr;
  }
}

// shards 1 .. G-1: no source, no PSD, no detector -- the published blocks through this GPU's inspectors
void secondary_main(suscan_analyzer *a)
{
  std::string err;
  BlockBus &bus = *a->bus;
  const size_t me = (size_t)a->shard - 1;
  auto give_up = [&] { { std::lock_guard<std::mutex> lk(bus.m); bus.done[me] = ~0ull; } bus.cv.notify_all(); };
  if (!init_device(a, err)) {
    push_status(a, SUSCAN_ANALYZER_MESSAGE_TYPE_INTERNAL, -1, "GPU shard " + std::to_string(a->shard) + " (device " + std::to_string(a->device) + "): " + err);
    give_up();
    return;
  }
  struct { bool on = false; int slot = 0; } flight;
  auto finish = [&] { if (flight.on) { collect_inspectors(a, flight.slot); flight.on = false; } };
  uint64_t k = 0;
  int slot = 0;
  bool failed = false;
  while (!failed) {
    for (;;) {                                                // requests routed to this shard's inspectors
      Request r;
      {
        std::lock_guard<std::mutex> lk(a->req_m);
        if (a->requests.empty()) break;
        r = std::move(a->requests.front());
        a->requests.pop_front();
      }
      finish();
      handle_request(a, r);
    }
    BlockBus::Entry e;
    {
      std::unique_lock<std::mutex> lk(bus.m);
      bus.cv.wait_for(lk, std::chrono::milliseconds(20), [&] { return bus.seq > k || bus.closed || a->halt.load(); });
      if (bus.seq <= k) {
        if (bus.closed || a->halt) break;
        continue;                                             // nothing yet: look at the request queue again
      }
      e = bus.e[k & 1];
    }
    if (e.samples != a->block || e.samp_rate != a->source_cfg.samp_rate) {   // first block, or the analyzer parameters changed
      finish();
      (void)hipDeviceSynchronize();
      if (a->d_x) (void)hipFree(a->d_x);
      if (a->d_raw) (void)hipFree(a->d_raw);
      a->d_x = nullptr; a->d_raw = nullptr; a->xfree_set = false;
      if (hipMalloc((void **)&a->d_x, e.samples * sizeof(suamd_complex)) != hipSuccess || hipMalloc(&a->d_raw, e.samples * 4) != hipSuccess) {
        push_status(a, SUSCAN_ANALYZER_MESSAGE_TYPE_INTERNAL, -1, "GPU shard " + std::to_string(a->shard) + ": allocation of the block buffers failed");
        failed = true;
        break;
      }
      a->block = e.samples;
      a->source_cfg.samp_rate = e.samp_rate;
      const bool fft = a->want_fft && a->block % 2048 == 0;
      for (auto &kv : a->inspectors) { kv.second->free_chain(); kv.second->dirty = true; }
      if (fft != a->use_fft && !fft && a->st) { suamd_specttuner_destroy(a->st); a->st = nullptr; }
      a->use_fft = fft;
    }
    if (a->fault_shard == a->shard && a->fault_block == (long long)k) {   // SUAMD_ANALYZER_FAULT (tests): this shard dies here,
      push_status(a, SUSCAN_ANALYZER_MESSAGE_TYPE_INTERNAL, -1, "GPU shard " + std::to_string(a->shard) + ": fault injection at block " + std::to_string(k));
      failed = true;                                                       // after the publisher has counted on it for block k
      break;
    }
    if (a->xfree_set) (void)hipStreamWaitEvent(a->stream, a->ev_xfree, 0);
    const bool compact = e.bytes_per_sample != sizeof(suamd_complex);
    void *dst = compact ? a->d_raw : (void *)a->d_x;
    const size_t bytes = e.valid * e.bytes_per_sample;
    bool sent = false;
    if (e.via_bcast) {
      void *comm = nullptr;
      { std::lock_guard<std::mutex> lk(bus.m); comm = bus.comm[a->shard]; }
      sent = comm && bus.bcast(nullptr, dst, bytes, 0 /* ncclInt8 */, 0, comm, a->stream) == 0;
      if (!sent) { std::lock_guard<std::mutex> lk(bus.m); bus.bcast_off = true; }
      else {
        // nothing that reads the block is enqueued before the broadcast is known to have completed (see wait_event_deadline)
        (void)hipEventRecord(a->ev_h2d[0], a->stream);
        if (!bcast_wait(a, a->ev_h2d[0])) { bus_broadcast_stuck(a, a->ev_h2d[0]); sent = false; }
      }
    }
    if (!sent) (void)hipMemcpyAsync(dst, e.host, bytes, hipMemcpyHostToDevice, a->stream);
    (void)hipEventRecord(a->ev_h2d[0], a->stream);
    if (compact && !suamd_ingest_iq(a->ctx, e.raw_format, a->d_raw, e.valid, a->d_x, a->stream))
      push_status(a, SUSCAN_ANALYZER_MESSAGE_TYPE_INTERNAL, -1, std::string("GPU shard ingest: ") + suamd_last_error());
    {
      // the same source conditioning as shard 0 (same input, same arithmetic: the shards see the same samples)
      const bool rev = a->primary->iq_reverse, dcr = a->primary->dc_remove;
      if (!dcr) a->dc_first = true;
      if (rev || dcr) {
        if (dcr && !a->d_dc && hipMalloc((void **)&a->d_dc, 2 * sizeof(float)) != hipSuccess) a->d_dc = nullptr;
        if (suamd_source_fix(a->ctx, a->d_x, e.valid, rev ? SU_TRUE : SU_FALSE, dcr ? a->d_dc : nullptr, 0.1f, a->dc_first ? SU_TRUE : SU_FALSE, a->stream) && dcr)
          a->dc_first = false;
      }
    }
    (void)hipEventRecord(a->ev_input, a->stream);
    enqueue_inspectors(a, e.valid, slot);
    // the publisher may reuse its host buffer once this shard's copy is out
    (void)hipEventSynchronize(a->ev_h2d[0]);
    { std::lock_guard<std::mutex> lk(bus.m); bus.done[me] = k + 1; }
    bus.cv.notify_all();
    finish();
    flight.on = true; flight.slot = slot;
    if (!a->pipelined) finish();
    slot ^= 1;
    ++k;
  }
  finish();
  for (auto &kv : a->inspectors) flush_watermark(a, *kv.second);   // before the publisher's EOS (bus_close joins this thread first)
  give_up();
  free_device(a);
}

// SUAMD_ANALYZER_BCAST=rccl (opt-in; the default is one host-to-device copy per GPU over its own PCIe link): the block goes
// host -> GPU 0 once and from there to every other shard with one ncclBroadcast per block over xGMI (SURVEY.md 8e;
// 16 MiB per 2 Mi-sample block against ~153 GB/s per link: ~0.1 ms).  librccl is loaded on demand, one communicator per
// shard from ncclCommInitAll (single process, one thread per device).  Needs distinct devices.
void setup_rccl(suscan_analyzer *a)
{
  if (!a->bus || a->secondaries.empty()) return;
  const char *mode = std::getenv("SUAMD_ANALYZER_BCAST");
  if (!mode || strcasecmp(mode, "rccl")) return;
  std::vector<int> devs{a->device};
  for (suscan_analyzer *s : a->secondaries) devs.push_back(s->device);
  // RCCL refuses two ranks on one device; SUAMD_RCCL_ALLOW_SAME_DEVICE=1 is for the test stand-in (tests/rccl_standin.cpp),
  // which lets the one-GPU test box run this branch with the device list 0,0
  const char *same = std::getenv("SUAMD_RCCL_ALLOW_SAME_DEVICE");
  if (!(same && std::atoi(same) != 0))
    for (size_t i = 0; i < devs.size(); ++i) for (size_t j = i + 1; j < devs.size(); ++j) if (devs[i] == devs[j]) return;
  // SUAMD_RCCL_LIB: an explicit library path (site builds of RCCL; the test stand-in)
  const char *path = std::getenv("SUAMD_RCCL_LIB");
  void *lib = path && *path ? dlopen(path, RTLD_NOW | RTLD_LOCAL) : nullptr;
  if (!lib && !(path && *path)) lib = dlopen("librccl.so.1", RTLD_NOW | RTLD_LOCAL);
  if (!lib && !(path && *path)) lib = dlopen("librccl.so", RTLD_NOW | RTLD_LOCAL);
  if (!lib) { push_status(a, SUSCAN_ANALYZER_MESSAGE_TYPE_INTERNAL, 0, "SUAMD_ANALYZER_BCAST=rccl: librccl not found, using per-GPU host copies"); return; }
  using InitAll = int (*)(void **, int, const int *);
  using Bcast = int (*)(const void *, void *, size_t, int, int, void *, hipStream_t);
  auto init_all = reinterpret_cast<InitAll>(dlsym(lib, "ncclCommInitAll"));
  auto bcast = reinterpret_cast<Bcast>(dlsym(lib, "ncclBroadcast"));
  std::vector<void *> comm(devs.size(), nullptr);
  if (!init_all || !bcast || init_all(comm.data(), (int)devs.size(), devs.data()) != 0) {
    push_status(a, SUSCAN_ANALYZER_MESSAGE_TYPE_INTERNAL, 0, "SUAMD_ANALYZER_BCAST=rccl: ncclCommInitAll failed, using per-GPU host copies");
    dlclose(lib);
    return;
  }
  (void)hipSetDevice(a->device);                             // ncclCommInitAll walks the devices
  std::lock_guard<std::mutex> lk(a->bus->m);
  a->bus->rccl_lib = lib; a->bus->comm = comm; a->bus->bcast = bcast;
  a->bus->comm_destroy = reinterpret_cast<int (*)(void *)>(dlsym(lib, "ncclCommDestroy"));
  a->bus->comm_abort = reinterpret_cast<int (*)(void *)>(dlsym(lib, "ncclCommAbort"));
  if (const char *t = std::getenv("SUAMD_ANALYZER_BCAST_TIMEOUT_MS")) { const int v = std::atoi(t); if (v >= 10 && v <= 600000) a->bus->bcast_timeout_ms = v; }
  if (const char *t = std::getenv("SUAMD_ANALYZER_BCAST_CEILING_MS")) { const int v = std::atoi(t); if (v >= 10 && v <= 3600000) a->bus->bcast_ceiling_ms = v; }
}

void worker_main(suscan_analyzer *a)
{
  std::string err;
  Source src;
  src.cfg = a->source_cfg;
  struct BusGuard { suscan_analyzer *a; ~BusGuard() { bus_close(a); } } bus_guard{a};   // the other shards stop with this one
  bool ok = init_device(a, err);
  if (ok) setup_rccl(a);
  if (ok) ok = src.open(err);
  if (ok && src.cfg.samp_rate != a->source_cfg.samp_rate) {       // a WAV / SigMF header carries its own rate
    a->source_cfg.samp_rate = src.cfg.samp_rate;
    a->info.source_samp_rate = a->info.effective_samp_rate = src.cfg.samp_rate;
    a->info.bandwidth = (SUFLOAT)src.cfg.samp_rate;
  }
  if (ok) ok = setup_psd(a, err);
  if (!ok) {
    push_status(a, SUSCAN_ANALYZER_MESSAGE_TYPE_SOURCE_INIT, SUSCAN_ANALYZER_INIT_FAILURE, err);
    push(a, SUSCAN_WORKER_MSG_TYPE_HALT, nullptr);
    return;
  }
  push_status(a, SUSCAN_ANALYZER_MESSAGE_TYPE_SOURCE_INIT, SUSCAN_ANALYZER_INIT_SUCCESS, "");
  push_source_info(a);
  AsyncRead reader(src);
  auto t_prev = std::chrono::steady_clock::now();
  uint64_t consumed = 0;
  // throttle anchor: the pace is (consumed - thr_c0) / rate seconds after thr_t0; re-anchored by SET_THROTTLE and SEEK
  auto thr_t0 = t_prev;
  uint64_t thr_c0 = 0;
  // double-buffered reading: block k+1 is read from the source while the GPU works on block k
  bool have_next = false, looped_next = false;
  size_t got_next = 0;
  int cur = 0;
  // Two blocks in flight: block k is enqueued (copy, PSD, every inspector stage) before block k-1's results are turned
  // into messages, so the device never waits for the host between blocks.  `flight` is the enqueued, uncollected block.
  struct InFlight { bool on = false; int slot = 0; suscan_analyzer_psd_msg *msg = nullptr; unsigned n = 0; bool chan = false; } flight;
  int slot = 0;
  double tmark[8] = {};
  const bool dbg = sdk::tuning().analyzer_debug != 0;
#define DBG(...) do { if (dbg) { std::fprintf(stderr, "[worker] " __VA_ARGS__); std::fputc('\n', stderr); std::fflush(stderr); } } while (0)
  // before the worker blocks on the device: the latest broadcast has completed, or its communicator is aborted (bcast_wait)
  auto bcast_settle = [&] {
    if (!a->bcast_pending) return;
    a->bcast_pending = false;
    if (!bcast_wait(a, a->ev_bcast)) bus_broadcast_stuck(a, a->ev_bcast);
  };
  auto finish = [&](InFlight &f) {                          // PSD message first, then the inspectors' messages, as ever
    bcast_settle();
    if (!f.on) return;
    DBG("finish slot %d: wait psd", f.slot);
    (void)hipEventSynchronize(a->ev_psd[f.slot]);
    DBG("finish slot %d: psd ok, collect", f.slot);
    if (f.msg) {                                            // (the stream's tail may hold no whole PSD frame)
      std::memcpy(f.msg->psd_data, a->h_psd[f.slot], f.n * sizeof(float));
      gettimeofday(&f.msg->rt_time, nullptr);
      push(a, SUSCAN_ANALYZER_MESSAGE_TYPE_PSD, f.msg);
    }
    if (f.chan && a->chandet) push_channels(a, f.slot);
    collect_inspectors(a, f.slot);
    DBG("finish slot %d: done", f.slot);
    f.on = false; f.msg = nullptr;
  };
  while (!a->halt) {
    // ---- requests posted by the GUI thread (they touch what a block in flight is using: let it finish first) ----
    for (;;) {
      Request r;
      {
        std::lock_guard<std::mutex> lk(a->req_m);
        if (a->requests.empty()) break;
        r = std::move(a->requests.front());
        a->requests.pop_front();
      }
      finish(flight);                                         // no request is handled under a block in flight
      const unsigned old_n = (unsigned)a->params.detector_params.window_size;
      const int old_w = a->params.detector_params.window;
      const float old_i = a->params.psd_update_int;
      if (r.kind == Request::SEEK) {                          // Suscan/Analyzer.cpp:150-154
        if (have_next) have_next = false;                     // the block read ahead is from the old position
        src.seek(r.value);
        // the channeliser's half-window history and cross-fade partners belong to the old position (all shards)
        if (a->bus) bus_wait_done(a, a->bus->seq);            // every shard has enqueued the blocks from before the seek
        a->st_reset_pending = true;
        for (suscan_analyzer *sh : a->secondaries) sh->st_reset_pending = true;
        consumed = r.value;
        a->position = consumed;
        thr_t0 = std::chrono::steady_clock::now(); thr_c0 = consumed;       // the throttle paces from here
        continue;
      }
      handle_request(a, r);
      if (r.kind == Request::SET_THROTTLE) { thr_t0 = std::chrono::steady_clock::now(); thr_c0 = consumed; }
      if (r.kind == Request::SET_PARAMS && (old_n != a->params.detector_params.window_size ||
                                           old_w != a->params.detector_params.window ||
                                           old_i != a->params.psd_update_int)) {
        if (have_next) { src.rewind_to_mark(); have_next = false; }        // it was read with the old block size
        if (!setup_psd(a, err)) {
          push_status(a, SUSCAN_ANALYZER_MESSAGE_TYPE_READ_ERROR, -1, err);
          a->halt = true;
        }
      } else if (r.kind == Request::SET_PARAMS) {
        setup_chandet(a);                                     // alpha / beta / gamma / snr / channel_update_int may have changed
      }
    }
    if (a->halt) break;
    // ---- one block ----
    if (a->bus) bus_wait_done(a, a->bus->seq);               // the other shards have copied every published block: the host halves are free
    const auto tb0 = std::chrono::steady_clock::now();
    a->t_block0 = tb0;
    auto tick = [&](int i) { if (a->trace) tmark[i] = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tb0).count(); };
    bool looped = looped_next;
    suamd_complex *h_cur = a->h_x + (size_t)cur * a->block;
    if (!have_next) {
      looped = false; src.mark();
      bcast_settle();
      if (a->h2d_set[cur]) (void)hipEventSynchronize(a->ev_h2d[cur]);
      got_next = src.read(h_cur, a->block, &looped);
    }
    have_next = false;
    const size_t got = got_next;
    // The stream ends inside this block: what is left still goes through the inspectors (a file-source consumer must
    // not lose the tail of every channel) -- whole half windows for the FFT filter bank, every sample for the FIR
    // bank -- and through the PSD as far as whole frames go; then EOS.
    const bool last = got < a->block;
    const size_t blen = !last ? a->block : (a->use_fft ? got / 2048 * 2048 : got);
    if (blen == 0) {
      finish(flight);
      for (auto &kv : a->inspectors) flush_watermark(a, *kv.second);
      bus_close(a);                                          // the other GPU shards deliver what they still hold: before EOS
      push_status(a, SUSCAN_ANALYZER_MESSAGE_TYPE_EOS, 0, "end of stream");
      break;
    }
    if (!last) {
      // the block after this one starts coming off the source now, on the helper thread
      src.mark();
      DBG("block at %llu: wait h2d of the other half", (unsigned long long)consumed);
      bcast_settle();
      if (a->h2d_set[cur ^ 1]) (void)hipEventSynchronize(a->ev_h2d[cur ^ 1]);   // that half's copy (the previous block) is out
      reader.start(a->h_x + (size_t)(cur ^ 1) * a->block, a->block);
      DBG("read-ahead started");
    }
    // ---- baseband filters: on this thread, on SUCOMPLEX samples, before anything else sees the block ----
    std::vector<suscan_analyzer::Filter> filters;
    {
      std::lock_guard<std::mutex> lk(a->req_m);
      filters = a->filters;
    }
    suamd_complex *h_flt = nullptr;                        // what the filters saw (and may have rewritten)
    std::string fatal;
    if (!filters.empty()) {
      if (src.bytes_per_sample() == sizeof(suamd_complex)) h_flt = h_cur;
      else {                                               // compact formats are expanded on the host for them (same arithmetic as suamd_ingest_iq)
        if (!a->h_flt && hipHostMalloc((void **)&a->h_flt, a->block * sizeof(suamd_complex), hipHostMallocPortable) != hipSuccess) fatal = "pinned allocation failed";
        else {
          h_flt = a->h_flt;
          const size_t nv = 2 * blen;
          float *o = reinterpret_cast<float *>(h_flt);
          switch (src.raw_format) {
            case SUAMD_FORMAT_RAW_UNSIGNED8: { const uint8_t *r = reinterpret_cast<const uint8_t *>(h_cur); for (size_t i = 0; i < nv; ++i) o[i] = (float)((int)r[i] - 128) * 0.0078125f; break; }
            case SUAMD_FORMAT_RAW_SIGNED8:   { const int8_t *r = reinterpret_cast<const int8_t *>(h_cur); for (size_t i = 0; i < nv; ++i) o[i] = (float)r[i] * 0.0078125f; break; }
            default:                         { const int16_t *r = reinterpret_cast<const int16_t *>(h_cur); for (size_t i = 0; i < nv; ++i) o[i] = (float)r[i] * 3.0517578125e-05f; break; }
          }
        }
      }
      for (const auto &f : filters) {
        if (!fatal.empty()) break;
        if (!f.func(f.priv, a, h_flt, blen, consumed)) fatal = "a baseband filter failed";
      }
    }
    if (!fatal.empty()) { (void)reader.wait(&looped_next); finish(flight); push_status(a, SUSCAN_ANALYZER_MESSAGE_TYPE_READ_ERROR, -1, fatal); break; }
    // every other GPU shard takes the block from here (what the filters left)
    const bool via_bcast = h_flt ? bus_publish(a, h_flt, a->block, blen, SUAMD_FORMAT_RAW_FLOAT32, sizeof(suamd_complex), consumed)
                                 : bus_publish(a, h_cur, a->block, blen, src.raw_format, src.bytes_per_sample(), consumed);
    // the previous block's channeliser must be done with d_x before this block lands in it (its PSD is on this stream)
    if (a->xfree_set) (void)hipStreamWaitEvent(a->stream, a->ev_xfree, 0);
    if (h_flt) {
      (void)hipMemcpyAsync(a->d_x, h_flt, blen * sizeof(suamd_complex), hipMemcpyHostToDevice, a->stream);
      if (h_flt == a->h_flt) { bcast_settle(); (void)hipStreamSynchronize(a->stream); }   // one expansion buffer: the copy must be out before the next block
    } else if (src.bytes_per_sample() == sizeof(suamd_complex)) {
      (void)hipMemcpyAsync(a->d_x, h_cur, blen * sizeof(suamd_complex), hipMemcpyHostToDevice, a->stream);
    } else {                                               // 2-4 B/sample over PCIe, expanded on the GPU
      (void)hipMemcpyAsync(a->d_raw, h_cur, blen * src.bytes_per_sample(), hipMemcpyHostToDevice, a->stream);
      if (!suamd_ingest_iq(a->ctx, src.raw_format, a->d_raw, blen, a->d_x, a->stream)) fatal = suamd_last_error();
    }
    if (via_bcast) {                                         // SUAMD_ANALYZER_BCAST=rccl: GPU 0 is the root of one broadcast per block
      void *root = src.bytes_per_sample() == sizeof(suamd_complex) || h_flt ? (void *)a->d_x : a->d_raw;
      const size_t bytes = blen * (h_flt ? sizeof(suamd_complex) : src.bytes_per_sample());
      void *comm = nullptr;
      { std::lock_guard<std::mutex> lk(a->bus->m); comm = a->bus->comm[0]; }
      if (!comm || a->bus->bcast(root, root, bytes, 0, 0, comm, a->stream) != 0) {
        std::lock_guard<std::mutex> lk(a->bus->m);
        a->bus->bcast_off = true;                              // (the shards that could not take part fall back to their host copy themselves)
      } else {
        // The root's stream is never waited for blindly behind a collective: a shard that died after this block was
        // published would otherwise hang the analyzer.  But the root only SENDS -- its own copy of the block is whole
        // whatever becomes of the broadcast -- so it does not stop here: the event is settled (bcast_settle) right before
        // the worker next blocks on the device, after this block's own work has been enqueued behind the collective.
        (void)hipEventRecord(a->ev_bcast, a->stream);
        a->bcast_pending = true;
      }
    }
    (void)hipEventRecord(a->ev_h2d[cur], a->stream);
    a->h2d_set[cur] = true;
    if (!fatal.empty()) { (void)reader.wait(&looped_next); finish(flight); push_status(a, SUSCAN_ANALYZER_MESSAGE_TYPE_READ_ERROR, -1, fatal); break; }
    {
      // source conditioning (Suscan/Analyzer.cpp:240-256): I/Q reversal, DC removal -- on the device, in place
      const bool rev = a->iq_reverse, dcr = a->dc_remove;
      if (!dcr) a->dc_first = true;
      if (rev || dcr) {
        if (dcr && !a->d_dc && hipMalloc((void **)&a->d_dc, 2 * sizeof(float)) != hipSuccess) a->d_dc = nullptr;
        if (!suamd_source_fix(a->ctx, a->d_x, blen, rev ? SU_TRUE : SU_FALSE, dcr ? a->d_dc : nullptr, 0.1f, a->dc_first ? SU_TRUE : SU_FALSE, a->stream))
          push_status(a, SUSCAN_ANALYZER_MESSAGE_TYPE_INTERNAL, -1, std::string("source conditioning: ") + suamd_last_error());
        else if (dcr) a->dc_first = false;
      }
    }
    (void)hipEventRecord(a->ev_input, a->stream);
    if (a->trace) (void)hipEventRecord(a->ev_t0, a->stream);
    tick(0);
    // the inspectors' chains start as soon as the block is on the device, next to the PSD
    DBG("enqueue slot %d", slot);
    enqueue_inspectors(a, blen, slot);
    DBG("enqueued");
    tick(1);
    const unsigned n = (unsigned)a->params.detector_params.window_size;
    const unsigned navg = last ? (unsigned)(blen / n) : a->navg;     // the tail: the whole frames it holds (none: no PSD message)
    // The main spectrum goes BEHIND the block's channeliser, not beside it.  Side by side the two transform launches share
    // every CU's LDS (a PSD workgroup takes 64 KB, a channeliser workgroup 50): not all of the channeliser's one round of
    // workgroups is resident, its runs wait out their bounded seam polls for successors that have not started, and the
    // launch takes 120 - 150 us per 2 Mi block where it takes 33 alone (rocprofv3 trace of the 64-inspector analyzer, round 6).
    // The chains wait for the channeliser; nothing waits for the PSD before the block's messages go out.
    (void)hipStreamWaitEvent(a->stream, a->ev_fir, 0);
    if (navg && !suamd_psd_feed(a->psd, a->d_x, navg, n, navg, 1.0f / (float)n, SUAMD_PSD_LINEAR, a->d_psd, a->stream)) {
      fatal = suamd_last_error();
      (void)reader.wait(&looped_next);                        // the helper thread is off the pinned buffer before it is freed
      bcast_settle();
      for (int k = 0; k < suscan_analyzer::NISTREAMS; ++k) (void)hipStreamSynchronize(a->istream[k]);
      finish(flight);
      push_status(a, SUSCAN_ANALYZER_MESSAGE_TYPE_READ_ERROR, -1, fatal);
      break;
    }
    if (navg) (void)hipMemcpyAsync(a->h_psd[slot], a->d_psd, n * sizeof(float), hipMemcpyDeviceToHost, a->stream);
    bool chan_now = false;
    if (a->chandet && navg) {                                        // the detector follows every block's spectrum; a list now and then
      if (!suamd_chandet_feed(a->chandet, a->d_psd, a->stream)) push_status(a, SUSCAN_ANALYZER_MESSAGE_TYPE_INTERNAL, -1, std::string("channel detector: ") + suamd_last_error());
      else if (++a->chan_phase >= a->chan_every) { a->chan_phase = 0; chan_now = suamd_chandet_find(a->chandet, slot, a->stream) != 0; }
    }
    (void)hipEventRecord(a->ev_psd[slot], a->stream);
    InFlight now_f;
    now_f.on = true; now_f.slot = slot;
    if (navg) {
      auto *m = static_cast<suscan_analyzer_psd_msg *>(std::calloc(1, sizeof(suscan_analyzer_psd_msg)));
      m->psd_size = n;
      m->psd_data = static_cast<SUFLOAT *>(std::malloc(n * sizeof(SUFLOAT)));
      {
        std::lock_guard<std::mutex> lk(a->req_m);
        m->fc = (int64_t)(a->params.mode == SUSCAN_ANALYZER_MODE_WIDE_SPECTRUM ? next_hop_frequency(a) : a->source_cfg.freq);
      }
      m->samp_rate = (SUFLOAT)a->source_cfg.samp_rate;
      m->measured_samp_rate = a->measured_rate;
      m->looped = looped ? SU_TRUE : SU_FALSE;
      const double ts = (double)consumed / a->source_cfg.samp_rate;
      m->timestamp.tv_sec = (time_t)ts;
      m->timestamp.tv_usec = (suseconds_t)((ts - std::floor(ts)) * 1e6);
      now_f.on = true; now_f.slot = slot; now_f.msg = m; now_f.n = n; now_f.chan = chan_now;
    }
    finish(flight);                                        // the block before this one: its messages go out now
    flight = now_f;
    if (!a->pipelined) finish(flight);
    if (last) {
      finish(flight);
      for (auto &kv : a->inspectors) flush_watermark(a, *kv.second);
      consumed += blen;
      a->position = consumed;
      bus_close(a);                                          // the other GPU shards deliver what they still hold: before EOS
      push_status(a, SUSCAN_ANALYZER_MESSAGE_TYPE_EOS, 0, "end of stream");
      break;
    }
    cur ^= 1;
    looped_next = false;
    DBG("wait for the read-ahead");
    got_next = reader.wait(&looped_next);                  // the next block is in the other half by now
    DBG("read-ahead got %zu", got_next);
    have_next = true;
    tick(2);
    slot ^= 1;
    tick(5);
    if (a->trace && (consumed / a->block) % 8 == 7)
    {
      auto el = [&](hipEvent_t e) { float ms = -1.f; if (e && hipEventElapsedTime(&ms, a->ev_t0, e) != hipSuccess) ms = -1.f; return ms; };
      std::fprintf(stderr, "[analyzer] device: channeliser done %.2f  agc pre %.2f |", el(a->ev_tfir), el(a->ev_tpre));
      for (int j = 0; j < a->nsub; ++j) std::fprintf(stderr, " [%d] agc %.2f carrier %.2f clock %.2f", j, el(a->ev_tstage[0][j]), el(a->ev_tstage[1][j]), el(a->ev_tstage[2][j]));
      std::fprintf(stderr, " | handed off %.2f ms after the block was on the device\n", el(a->ev_tdone));
      std::fprintf(stderr, "[analyzer] block (one in flight while tracing): input issued %.2f  chains issued %.2f  next block read %.2f  chains done %.2f  delivered %.2f ms\n",
                   tmark[0], tmark[1], tmark[2], a->t_chains_done, tmark[5]);
    }
    consumed += a->block;
    a->position = consumed;
    // ---- rate bookkeeping / throttle ----
    auto now = std::chrono::steady_clock::now();
    const uint64_t thr = a->throttle;
    if (thr > 0) {
      const double due = (double)(consumed - thr_c0) / (double)thr;
      for (;;) {                                             // short slices: halt / destroy are never kept waiting
        const double el = std::chrono::duration<double>(std::chrono::steady_clock::now() - thr_t0).count();
        if (!(due > el) || a->halt) break;
        std::this_thread::sleep_for(std::chrono::duration<double>(std::min(due - el, 0.02)));
      }
      now = std::chrono::steady_clock::now();
    }
    const double dt = std::chrono::duration<double>(now - t_prev).count();
    t_prev = now;
    if (dt > 0) {
      const float inst = (float)((double)a->block / dt);
      const float prev = a->measured_rate;
      a->measured_rate = prev == 0.f ? inst : prev + 0.2f * (inst - prev);
    }
  }
  finish(flight);
  bus_close(a);                                              // ... and before this worker's HALT
  free_device(a);
  push(a, SUSCAN_WORKER_MSG_TYPE_HALT, nullptr);
}

SUBOOL post(suscan_analyzer *a, Request &&r)
{
  if (!a) return SU_FALSE;
  {
    std::lock_guard<std::mutex> lk(a->req_m);
    a->requests.push_back(std::move(r));
  }
  if (a->shard > 0 && a->bus) a->bus->cv.notify_all();         // a GPU shard waiting for the next block looks at its requests
  return SU_TRUE;
}

// the shard an inspector handle lives on (handles are dealt s, s + G, s + 2G ... by shard s)
suscan_analyzer *shard_of(suscan_analyzer *a, SUHANDLE h)
{
  if (!a || a->nshards <= 1 || h < 0) return a;
  const int s = (int)(h % a->nshards);
  return s == 0 ? a : a->secondaries[(size_t)s - 1];
}

}  // namespace

// ==========================================================================================
extern "C" {

// ---- analyzer ----
suscan_analyzer_t *suscan_analyzer_new(const struct suscan_analyzer_params *params, suscan_source_config_t *config,
                                       struct suscan_mq *mq)
{
  if (!params || !config || !mq || !mq->impl) return nullptr;
  // one hardware queue per inspector stream (the HIP default is 4); only effective when this is the
  // process's first use of the HIP runtime, and never overrides the user's choice
  setenv("GPU_MAX_HW_QUEUES", "8", 0);
  auto *a = new (std::nothrow) suscan_analyzer;
  if (!a) return nullptr;
  a->params = *params;
  a->source_cfg = *config;
  a->mq = mq;
  a->info.source_samp_rate = a->info.effective_samp_rate = config->samp_rate;
  a->info.frequency = config->freq;
  a->info.freq_min = -3e11; a->info.freq_max = 3e11;
  a->info.bandwidth = (SUFLOAT)config->samp_rate;
  a->info.seekable = config->type == "file" ? SU_TRUE : SU_FALSE;
  a->primary = a;
  // SUAMD_DEVICES="0,1,2,3": the inspectors are sharded over these GPUs (the first one also owns the source and the PSD)
  {
    std::vector<int> devs;
    if (const char *e = std::getenv("SUAMD_DEVICES")) {
      for (const char *q = e; *q;) {
        char *end = nullptr;
        const long v = std::strtol(q, &end, 10);
        if (end == q) break;
        if (v >= 0 && v < 64) devs.push_back((int)v);
        q = *end == ',' ? end + 1 : end;
      }
    }
    if (devs.empty()) devs.push_back(0);
    a->device = devs[0];
    a->nshards = (int)devs.size();
    if (devs.size() > 1) {
      a->bus = std::make_shared<BlockBus>();
      a->bus->done.assign(devs.size() - 1, 0);
      for (size_t i = 1; i < devs.size(); ++i) {
        auto *s = new (std::nothrow) suscan_analyzer;
        if (!s) break;
        s->params = *params; s->source_cfg = *config; s->mq = mq;
        s->device = devs[i]; s->shard = (int)i; s->nshards = a->nshards; s->primary = a; s->bus = a->bus;
        s->next_handle = (SUHANDLE)i;
        // test hook (tests/test_gpu_analyzer_fft.py: a shard that dies mid-broadcast), honoured only beside SUAMD_TEST_HOOKS=1:
        // nothing a deployment sets by accident.  INTEGRATION.md section 3.1.
        const char *hooks = std::getenv("SUAMD_TEST_HOOKS");
        if (const char *f = (hooks && std::atoi(hooks) == 1) ? std::getenv("SUAMD_ANALYZER_FAULT") : nullptr) {
          int fs = -1; long long fb = -1;
          if (std::sscanf(f, "shard_dies:%d:%lld", &fs, &fb) == 2) { s->fault_shard = fs; s->fault_block = fb; }
        }
        a->secondaries.push_back(s);
      }
      if (a->secondaries.size() != devs.size() - 1) {          // out of memory: one GPU then
        for (suscan_analyzer *s : a->secondaries) delete s;
        a->secondaries.clear(); a->bus.reset(); a->nshards = 1;
      }
      for (suscan_analyzer *s : a->secondaries)
        s->worker = std::thread([s] {
          try { secondary_main(s); }
          catch (const std::exception &e) {
            push_status(s, SUSCAN_ANALYZER_MESSAGE_TYPE_INTERNAL, -1, std::string("GPU shard worker: ") + e.what());
            { std::lock_guard<std::mutex> lk(s->bus->m); s->bus->done[(size_t)s->shard - 1] = ~0ull; }
            s->bus->cv.notify_all();
          }
        });
    }
  }
  a->worker = std::thread([a] {
    // an exception on this thread (std::bad_alloc from a per-block vector, a malformed header) must not take the host
    // process down with std::terminate: the reader gets READ_ERROR + HALT, as for any other source failure
    try { worker_main(a); }
    catch (const std::exception &e) {
      push_status(a, SUSCAN_ANALYZER_MESSAGE_TYPE_READ_ERROR, -1, std::string("analyzer worker: ") + e.what());
      push(a, SUSCAN_WORKER_MSG_TYPE_HALT, nullptr);
    }
    catch (...) {
      push_status(a, SUSCAN_ANALYZER_MESSAGE_TYPE_READ_ERROR, -1, "analyzer worker: unknown exception");
      push(a, SUSCAN_WORKER_MSG_TYPE_HALT, nullptr);
    }
  });
  return a;
}

void suscan_analyzer_destroy(suscan_analyzer_t *a)
{
  if (!a) return;
  a->halt = true;
  if (a->worker.joinable()) a->worker.join();                 // joins the other shards' workers on its way out
  for (suscan_analyzer *s : a->secondaries) {
    if (s->worker.joinable()) s->worker.join();
    for (auto &r : s->requests) if (r.config) suscan_config_destroy(r.config);
    delete s;
  }
  for (auto &r : a->requests) if (r.config) suscan_config_destroy(r.config);
  suscan_source_info_finalize(&a->info);
  delete a;
}

void *suscan_analyzer_read(suscan_analyzer_t *a, uint32_t *type) { return a ? suscan_mq_read(a->mq, type) : nullptr; }

void suscan_analyzer_dispose_message(uint32_t type, void *ptr)
{
  if (!ptr) return;
  switch (type) {
    case SUSCAN_ANALYZER_MESSAGE_TYPE_PSD:
      std::free(static_cast<suscan_analyzer_psd_msg *>(ptr)->psd_data);
      break;
    case SUSCAN_ANALYZER_MESSAGE_TYPE_SAMPLES:
      std::free(static_cast<suscan_analyzer_sample_batch_msg *>(ptr)->samples);
      break;
    case SUSCAN_ANALYZER_MESSAGE_TYPE_INSPECTOR: {
      auto *m = static_cast<suscan_analyzer_inspector_msg *>(ptr);
      std::free(m->class_name);
      if (m->config) suscan_config_destroy(m->config);
      std::free(m->spectrum_data);
      std::free(m->spectsrc_list);                           // the names themselves are static
      std::free(m->estimator_list);
      std::free(m->signal_name);
      break;
    }
    case SUSCAN_ANALYZER_MESSAGE_TYPE_SOURCE_INFO:
      suscan_source_info_finalize(static_cast<suscan_source_info *>(ptr));
      break;
    case SUSCAN_ANALYZER_MESSAGE_TYPE_CHANNEL: {
      auto *m = static_cast<suscan_analyzer_channel_msg *>(ptr);
      for (unsigned i = 0; i < m->channel_count; ++i) std::free(m->channel_list[i]);
      std::free(m->channel_list);
      break;
    }
    case SUSCAN_ANALYZER_MESSAGE_TYPE_SOURCE_INIT:
    case SUSCAN_ANALYZER_MESSAGE_TYPE_EOS:
    case SUSCAN_ANALYZER_MESSAGE_TYPE_READ_ERROR:
    case SUSCAN_ANALYZER_MESSAGE_TYPE_INTERNAL:
      std::free(static_cast<suscan_analyzer_status_msg *>(ptr)->err_msg);
      break;
    default: break;
  }
  std::free(ptr);
}

void suscan_analyzer_req_halt(suscan_analyzer_t *a) { if (a) a->halt = true; }

SUBOOL suscan_analyzer_set_params_async(suscan_analyzer_t *a, const struct suscan_analyzer_params *p, uint32_t req)
{
  if (!p) return SU_FALSE;
  Request r; r.kind = Request::SET_PARAMS; r.req_id = req; r.params = *p;
  return post(a, std::move(r));
}

SUBOOL suscan_analyzer_set_throttle_async(suscan_analyzer_t *a, SUSCOUNT rate, uint32_t req)
{
  Request r; r.kind = Request::SET_THROTTLE; r.req_id = req; r.value = rate;
  return post(a, std::move(r));
}

unsigned int suscan_analyzer_get_samp_rate(const suscan_analyzer_t *a) { return a ? a->source_cfg.samp_rate : 0; }
SUFLOAT suscan_analyzer_get_measured_samp_rate(const suscan_analyzer_t *a) { return a ? (SUFLOAT)a->measured_rate : 0; }
struct suscan_source_info *suscan_analyzer_get_source_info(const suscan_analyzer_t *a)
{
  return a ? const_cast<suscan_source_info *>(&a->info) : nullptr;
}

SUBOOL suscan_analyzer_open_ex_async(suscan_analyzer_t *a, const char *cls, const struct sigutils_channel *ch,
                                     SUBOOL precise, SUHANDLE, uint32_t req)
{
  if (!cls || !ch) return SU_FALSE;
  Request r; r.kind = Request::OPEN; r.req_id = req; r.cls = cls; r.channel = *ch; r.precise = precise != 0;
  // new inspectors go round the GPU shards (channel c -> GPU c mod G)
  return post(a && a->nshards > 1 ? shard_of(a, (SUHANDLE)(a->open_rr++ % (uint32_t)a->nshards)) : a, std::move(r));
}

SUBOOL suscan_analyzer_open_async(suscan_analyzer_t *a, const char *cls, const struct sigutils_channel *ch, uint32_t req)
{
  return suscan_analyzer_open_ex_async(a, cls, ch, SU_FALSE, -1, req);
}

SUBOOL suscan_analyzer_close_async(suscan_analyzer_t *a, SUHANDLE h, uint32_t req)
{
  Request r; r.kind = Request::CLOSE; r.req_id = req; r.handle = h;
  return post(shard_of(a, h), std::move(r));
}

SUBOOL suscan_analyzer_set_inspector_id_async(suscan_analyzer_t *a, SUHANDLE h, uint32_t id, uint32_t req)
{
  Request r; r.kind = Request::SET_ID; r.req_id = req; r.handle = h; r.inspector_id = id;
  return post(shard_of(a, h), std::move(r));
}

SUBOOL suscan_analyzer_set_inspector_config_async(suscan_analyzer_t *a, SUHANDLE h, const suscan_config_t *cfg, uint32_t req)
{
  if (!cfg) return SU_FALSE;
  Request r; r.kind = Request::SET_CONFIG; r.req_id = req; r.handle = h; r.config = suscan_config_dup(cfg);
  return post(shard_of(a, h), std::move(r));
}

SUBOOL suscan_analyzer_set_inspector_watermark_async(suscan_analyzer_t *a, SUHANDLE h, SUSCOUNT wm, uint32_t req)
{
  Request r; r.kind = Request::SET_WATERMARK; r.req_id = req; r.handle = h; r.value = wm;
  return post(shard_of(a, h), std::move(r));
}

const struct suscan_spectsrc_class *suscan_spectsrc_class_lookup(const char *name)
{
  static const struct suscan_spectsrc_class classes[] = {
    {"psd", "Power spectrum"}, {"cyclo", "Cyclostationary analysis"}, {"fmspect", "FM spectrum"},
    {"pmspect", "PM spectrum"}, {"timediff", "Time derivative"}, {"abstimediff", "Absolute value of the time derivative"},
    {"exp_2", "Signal exponentiation (2)"}, {"exp_4", "Signal exponentiation (4)"}, {"exp_8", "Signal exponentiation (8)"},
  };
  if (!name) return nullptr;
  for (const auto &c : classes) if (!std::strcmp(c.name, name)) return &c;
  return nullptr;
}

const struct suscan_estimator_class *suscan_estimator_class_lookup(const char *name)
{
  for (const auto &e : kEstimators) if (name && std::strcmp(name, e.name) == 0) return &e;
  return nullptr;
}

SUBOOL suscan_analyzer_inspector_set_spectrum_async(suscan_analyzer_t *a, SUHANDLE h, uint32_t spectsrc_id, uint32_t req)
{
  Request r; r.kind = Request::SET_SPECTRUM; r.req_id = req; r.handle = h; r.value = spectsrc_id;
  return post(shard_of(a, h), std::move(r));
}

SUBOOL suscan_analyzer_set_inspector_freq_overridable(suscan_analyzer_t *a, SUHANDLE h, SUFREQ f)
{
  Request r; r.kind = Request::SET_FREQ; r.handle = h; r.fvalue = f;
  return post(shard_of(a, h), std::move(r));
}

SUBOOL suscan_analyzer_set_inspector_bandwidth_overridable(suscan_analyzer_t *a, SUHANDLE h, SUFREQ bw)
{
  Request r; r.kind = Request::SET_BW; r.handle = h; r.fvalue = bw;
  return post(shard_of(a, h), std::move(r));
}

SUBOOL suscan_analyzer_inspector_estimator_cmd_async(suscan_analyzer_t *a, SUHANDLE h, uint32_t estimator_id, SUBOOL enabled, uint32_t req)
{
  Request r; r.kind = Request::ESTIMATOR; r.handle = h; r.value = estimator_id; r.fvalue = enabled ? 1 : 0; r.req_id = req;
  return post(shard_of(a, h), std::move(r));
}

SUBOOL suscan_analyzer_inspector_set_tle_async(suscan_analyzer_t *a, SUHANDLE h, const orbit_t *tle, uint32_t req)
{
  Request r; r.kind = Request::SET_TLE; r.handle = h; r.fvalue = tle ? 1 : 0; r.req_id = req;
  return post(shard_of(a, h), std::move(r));
}

SUBOOL suscan_analyzer_register_baseband_filter_with_prio(suscan_analyzer_t *a, suscan_analyzer_baseband_filter_func_t func, void *priv,
                                                          int64_t prio)
{
  if (!a || !func) return SU_FALSE;
  std::lock_guard<std::mutex> lk(a->req_m);
  a->filters.push_back(suscan_analyzer::Filter{func, priv, prio, a->filter_seq++});
  std::stable_sort(a->filters.begin(), a->filters.end(),
                   [](const suscan_analyzer::Filter &x, const suscan_analyzer::Filter &y) { return x.prio < y.prio; });
  return SU_TRUE;
}

SUBOOL suscan_analyzer_register_baseband_filter(suscan_analyzer_t *a, suscan_analyzer_baseband_filter_func_t func, void *priv)
{
  return suscan_analyzer_register_baseband_filter_with_prio(a, func, priv, 0);
}

static SUBOOL post_source_info(suscan_analyzer_t *a)
{
  Request r; r.kind = Request::SOURCE_INFO;
  return post(a, std::move(r));
}

SUBOOL suscan_analyzer_set_freq(suscan_analyzer_t *a, SUFREQ freq, SUFREQ lnb)
{
  if (!a) return SU_FALSE;
  {
    std::lock_guard<std::mutex> lk(a->req_m);
    a->source_cfg.freq = freq;                             // the PSD messages' fc from the next block on
    a->info.frequency = freq; a->info.lnb = lnb;
  }
  return post_source_info(a);
}

SUBOOL suscan_analyzer_set_gain(suscan_analyzer_t *a, const char *name, SUFLOAT value)
{
  if (!a || !name) return SU_FALSE;
  { std::lock_guard<std::mutex> lk(a->req_m); a->gains[name] = value; }
  return SU_TRUE;
}

SUBOOL suscan_analyzer_set_antenna(suscan_analyzer_t *a, const char *name)
{
  if (!a || !name) return SU_FALSE;
  {
    std::lock_guard<std::mutex> lk(a->req_m);
    a->antenna = name;
    std::free(a->info.antenna); a->info.antenna = strdup(name);
  }
  return SU_TRUE;
}

SUBOOL suscan_analyzer_set_bw(suscan_analyzer_t *a, SUFLOAT bw)
{
  if (!a || !(bw > 0)) return SU_FALSE;
  { std::lock_guard<std::mutex> lk(a->req_m); a->info.bandwidth = bw; }
  return post_source_info(a);
}

SUBOOL suscan_analyzer_set_ppm(suscan_analyzer_t *a, SUFLOAT ppm)
{
  if (!a) return SU_FALSE;
  { std::lock_guard<std::mutex> lk(a->req_m); a->info.ppm = ppm; }
  return post_source_info(a);
}

SUBOOL suscan_analyzer_set_agc(suscan_analyzer_t *a, SUBOOL enabled)
{
  if (!a) return SU_FALSE;
  { std::lock_guard<std::mutex> lk(a->req_m); a->info.agc = enabled ? SU_TRUE : SU_FALSE; }
  return post_source_info(a);
}

SUBOOL suscan_analyzer_set_dc_remove(suscan_analyzer_t *a, SUBOOL remove)
{
  if (!a) return SU_FALSE;
  a->dc_remove = remove != 0;
  { std::lock_guard<std::mutex> lk(a->req_m); a->info.dc_remove = remove ? SU_TRUE : SU_FALSE; }
  return post_source_info(a);
}

SUBOOL suscan_analyzer_set_iq_reverse(suscan_analyzer_t *a, SUBOOL reverse)
{
  if (!a) return SU_FALSE;
  a->iq_reverse = reverse != 0;
  { std::lock_guard<std::mutex> lk(a->req_m); a->info.iq_reverse = reverse ? SU_TRUE : SU_FALSE; }
  return post_source_info(a);
}

SUBOOL suscan_analyzer_seek(suscan_analyzer_t *a, const struct timeval *pos)
{
  if (!a || !pos || !a->info.seekable) return SU_FALSE;
  const double t = (double)pos->tv_sec + 1e-6 * (double)pos->tv_usec;
  if (t < 0) return SU_FALSE;
  Request r; r.kind = Request::SEEK; r.value = (SUSCOUNT)(t * (double)a->source_cfg.samp_rate);
  return post(a, std::move(r));
}

SUBOOL suscan_analyzer_set_history_size(suscan_analyzer_t *a, SUSCOUNT size)
{
  if (!a) return SU_FALSE;
  {
    std::lock_guard<std::mutex> lk(a->req_m);
    a->history_size = size;                                // a file is its own history: nothing to allocate
    a->info.history_length = size;
  }
  return SU_TRUE;
}

SUBOOL suscan_analyzer_replay(suscan_analyzer_t *a, SUBOOL replay)
{
  if (!a) return SU_FALSE;
  { std::lock_guard<std::mutex> lk(a->req_m); a->replay = replay != 0; a->info.replay = replay ? SU_TRUE : SU_FALSE; }
  return SU_TRUE;
}

void suscan_analyzer_get_source_time(const suscan_analyzer_t *a, struct timeval *tv)
{
  if (!tv) return;
  tv->tv_sec = 0; tv->tv_usec = 0;
  if (!a) return;
  const double t = (double)a->position.load() / (double)a->source_cfg.samp_rate;
  tv->tv_sec = a->info.source_start.tv_sec + (time_t)t;
  tv->tv_usec = (suseconds_t)((t - std::floor(t)) * 1e6);
}

// wide-spectrum controls: only meaningful for an analyzer created in that mode (a hopping tuner is not part of this path)
static bool wide(const suscan_analyzer_t *a) { return a && a->params.mode == SUSCAN_ANALYZER_MODE_WIDE_SPECTRUM; }

SUBOOL suscan_analyzer_set_sweep_stratrgy(suscan_analyzer_t *a, enum suscan_analyzer_sweep_strategy strategy)
{
  if (!wide(a) || (int)strategy < 0 || (int)strategy > 1) return SU_FALSE;
  { std::lock_guard<std::mutex> lk(a->req_m); a->sweep_strategy = (int)strategy; }
  return SU_TRUE;
}

SUBOOL suscan_analyzer_set_spectrum_partitioning(suscan_analyzer_t *a, enum suscan_analyzer_spectrum_partitioning p)
{
  if (!wide(a) || (int)p < 0 || (int)p > 1) return SU_FALSE;
  { std::lock_guard<std::mutex> lk(a->req_m); a->partitioning = (int)p; }
  return SU_TRUE;
}

SUBOOL suscan_analyzer_set_hop_range(suscan_analyzer_t *a, SUFREQ min, SUFREQ max)
{
  if (!wide(a) || max < min) return SU_FALSE;
  { std::lock_guard<std::mutex> lk(a->req_m); a->hop_min = min; a->hop_max = max; a->hop_k = 0; }   // the sweep starts over
  return SU_TRUE;
}

SUBOOL suscan_analyzer_set_rel_bandwidth(suscan_analyzer_t *a, SUFLOAT rel_bw)
{
  if (!wide(a) || !(rel_bw > 0) || rel_bw > 1) return SU_FALSE;
  { std::lock_guard<std::mutex> lk(a->req_m); a->rel_bw = rel_bw; }
  return SU_TRUE;
}

SUBOOL suscan_analyzer_set_buffering_size(suscan_analyzer_t *a, SUSCOUNT size)
{
  if (!wide(a)) return SU_FALSE;
  a->buffering_size = size;
  return SU_TRUE;
}

}  // extern "C"
