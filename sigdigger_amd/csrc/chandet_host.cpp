// chandet_host.cpp -- suamd_chandet_*: the object around chandet.hip (SPEC.md section O; row N1).  Conversion of the
// device's bin records to channels in Hz / dB happens here (a handful of values per update: formatting, not the path).
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cmath>
#include <new>
#include <vector>

#include "../../include/sigdigger_amd.h"
#include "kernels.hpp"

void suamd_set_error(const char *fmt, ...);                // capi.hip

struct suamd_chandet {
  suamd_ctx_t *ctx = nullptr;
  int n = 0;
  float alpha = 0, beta = 0, gamma = 0, snr = 0;
  bool first = true;
  float *d_S = nullptr, *d_N0 = nullptr;
  static constexpr unsigned CAP = 1024;
  sdk::ChanDetRecord *d_rec = nullptr;
  unsigned *d_count = nullptr;
  struct Landing { unsigned count; float N0; sdk::ChanDetRecord rec[CAP]; } *h[2] = {nullptr, nullptr};   // pinned
  struct Track { double fc; float S0; unsigned age; };
  std::vector<Track> tracks;                                   // the previous list: beta smooths a continuing channel's level
};

extern "C" {

suamd_chandet_t *suamd_chandet_new(suamd_ctx_t *ctx, unsigned n, SUFLOAT alpha, SUFLOAT beta, SUFLOAT gamma, SUFLOAT snr)
{
  if (!ctx) { suamd_set_error("null context"); return nullptr; }
  if (n < 512 || n > 16384 || (n & (n - 1))) { suamd_set_error("channel detector: %u bins unsupported (power of two, 512..16384)", n); return nullptr; }
  if (!(alpha > 0 && alpha <= 1) || !(gamma > 0 && gamma <= 1) || !(snr > 0)) { suamd_set_error("channel detector: alpha, gamma in (0, 1], snr > 0"); return nullptr; }
  if (hipSetDevice(suamd_ctx_device(ctx)) != hipSuccess) { suamd_set_error("hipSetDevice failed"); return nullptr; }
  auto *d = new (std::nothrow) suamd_chandet();
  if (!d) { suamd_set_error("out of memory"); return nullptr; }
  d->ctx = ctx; d->n = (int)n; d->alpha = alpha; d->beta = beta; d->gamma = gamma; d->snr = snr;
  bool ok = hipMalloc((void **)&d->d_S, n * sizeof(float)) == hipSuccess && hipMalloc((void **)&d->d_N0, sizeof(float)) == hipSuccess &&
            hipMalloc((void **)&d->d_rec, suamd_chandet::CAP * sizeof(sdk::ChanDetRecord)) == hipSuccess &&
            hipMalloc((void **)&d->d_count, sizeof(unsigned)) == hipSuccess;
  for (int p = 0; p < 2 && ok; ++p) ok = hipHostMalloc((void **)&d->h[p], sizeof(suamd_chandet::Landing), hipHostMallocDefault) == hipSuccess;
  if (!ok) { suamd_set_error("allocation failed"); suamd_chandet_destroy(d); return nullptr; }
  return d;
}

void suamd_chandet_destroy(suamd_chandet_t *d)
{
  if (!d) return;
  for (void *p : {(void *)d->d_S, (void *)d->d_N0, (void *)d->d_rec, (void *)d->d_count}) if (p) (void)hipFree(p);
  for (int p = 0; p < 2; ++p) if (d->h[p]) (void)hipHostFree(d->h[p]);
  delete d;
}

SUBOOL suamd_chandet_feed(suamd_chandet_t *d, const SUFLOAT *d_psd, void *stream)
{
  if (!d || !d_psd) { suamd_set_error("null argument"); return SU_FALSE; }
  const hipError_t e = sdk::chandet_feed(d->d_S, d_psd, d->n, d->alpha, d->gamma, d->first ? 1 : 0, d->d_N0, static_cast<hipStream_t>(stream));
  if (e != hipSuccess) { suamd_set_error("channel detector launch failed: %s", hipGetErrorString(e)); return SU_FALSE; }
  d->first = false;
  return SU_TRUE;
}

SUBOOL suamd_chandet_find(suamd_chandet_t *d, int slot, void *stream)
{
  if (!d || slot < 0 || slot > 1) { suamd_set_error("bad argument"); return SU_FALSE; }
  if (d->first) { d->h[slot]->count = 0; d->h[slot]->N0 = 0; return SU_TRUE; }
  hipStream_t s = static_cast<hipStream_t>(stream);
  hipError_t e = sdk::chandet_find(d->d_S, d->n, d->d_N0, d->snr, d->d_rec, d->d_count, suamd_chandet::CAP, s);
  if (e == hipSuccess) e = hipMemcpyAsync(&d->h[slot]->count, d->d_count, sizeof(unsigned), hipMemcpyDeviceToHost, s);
  if (e == hipSuccess) e = hipMemcpyAsync(&d->h[slot]->N0, d->d_N0, sizeof(float), hipMemcpyDeviceToHost, s);
  if (e == hipSuccess) e = hipMemcpyAsync(d->h[slot]->rec, d->d_rec, suamd_chandet::CAP * sizeof(sdk::ChanDetRecord), hipMemcpyDeviceToHost, s);
  if (e != hipSuccess) { suamd_set_error("channel detector: %s", hipGetErrorString(e)); return SU_FALSE; }
  return SU_TRUE;
}

int suamd_chandet_collect(suamd_chandet_t *d, int slot, SUFLOAT samp_rate, struct suamd_channel *out, unsigned cap)
{
  if (!d || slot < 0 || slot > 1 || (cap && !out)) { suamd_set_error("bad argument"); return -1; }
  const suamd_chandet::Landing &l = *d->h[slot];
  const unsigned n = std::min(l.count, suamd_chandet::CAP);
  std::vector<sdk::ChanDetRecord> rec(l.rec, l.rec + n);
  std::sort(rec.begin(), rec.end(), [](const sdk::ChanDetRecord &a, const sdk::ChanDetRecord &b) { return a.first < b.first; });
  const double df = (double)samp_rate / (double)d->n, half = 0.5 * (double)d->n;
  const float n0db = 10.0f * std::log10(l.N0 + 1e-8f);     // SU_POWER_DB
  unsigned k = 0;
  for (; k < n && k < cap; ++k) {
    const sdk::ChanDetRecord &r = rec[k];
    out[k].f_lo = ((double)r.first - half - 0.5) * df;
    out[k].f_hi = ((double)r.last - half + 0.5) * df;
    out[k].fc = (r.wsum / r.sum - half) * df;
    out[k].bw = (SUFLOAT)(out[k].f_hi - out[k].f_lo);
    out[k].S0 = 10.0f * std::log10(r.peak + 1e-8f);
    out[k].N0 = n0db;
    out[k].age = 0;
    // beta: the channel of the previous list whose centre lies inside this one is continued (the first such, in
    // frequency order)
    if (d->beta > 0.0f && d->beta < 1.0f) {
      for (const suamd_chandet::Track &t : d->tracks)
        if (t.fc >= out[k].f_lo && t.fc <= out[k].f_hi) {
          out[k].S0 = t.S0 + d->beta * (out[k].S0 - t.S0);
          out[k].age = t.age + 1;
          break;
        }
    }
    out[k].snr = out[k].S0 - n0db;
  }
  d->tracks.clear();
  for (unsigned i = 0; i < k; ++i) d->tracks.push_back(suamd_chandet::Track{out[i].fc, out[i].S0, out[i].age});
  return (int)k;
}

int suamd_chandet_channels(suamd_chandet_t *d, SUFLOAT samp_rate, struct suamd_channel *out, unsigned cap, void *stream)
{
  if (!suamd_chandet_find(d, 0, stream)) return -1;
  if (hipStreamSynchronize(static_cast<hipStream_t>(stream)) != hipSuccess) { suamd_set_error("device failure"); return -1; }
  return suamd_chandet_collect(d, 0, samp_rate, out, cap);
}

SUFLOAT suamd_chandet_noise_floor(suamd_chandet_t *d, void *stream)
{
  if (!d || d->first) return 0;
  float v = 0;
  if (hipMemcpyAsync(&v, d->d_N0, sizeof v, hipMemcpyDeviceToHost, static_cast<hipStream_t>(stream)) != hipSuccess ||
      hipStreamSynchronize(static_cast<hipStream_t>(stream)) != hipSuccess) return 0;
  return v;
}

}  // extern "C"
