// specttuner_host.cpp -- host side of the FFT channeliser (kernel: specttuner.hip; SPEC.md section C2):
//   suamd_specttuner_*   device-resident block interface (include/sigdigger_amd.h), used by the analyzer and the bench;
//   su_specttuner_*      the libsigutils names and callback contract (include/sigutils/specttuner.h) that
//                        Tasks/LPFTask.cpp:52-69,83-87,104-107,123 is written against -- it links unchanged.
// Channel geometry and the frequency response are designed here in double precision (not on the hot path); there is
// no CPU implementation of the channeliser itself: without a gfx950 device the constructors fail.
// phase clocks (STW_TSTAMP) and the wrong-result timing experiments (STP_UNSAFE_*) exist in the instrumented build only
#ifndef SUAMD_INSTRUMENT
#undef STW_TSTAMP
#undef STP_UNSAFE_NO_EXCHANGE
#undef STP_UNSAFE_NO_ALIAS_BARRIERS
#endif
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <cstdio>
#include <map>
#include <memory>
#include <new>
#include <vector>

#include "../../include/sigdigger_amd.h"
#include "kernels.hpp"
#include "tuning.hpp"

void suamd_set_error(const char *fmt, ...);                // capi.hip

namespace {

constexpr double kPi = 3.14159265358979323846;
struct c32 { float re, im; };

// iterative radix-2 transform in binary64 (design only), sign = -1 forward, +1 backward, unnormalised
void fft64(std::vector<double> &re, std::vector<double> &im, int sign)
{
  const size_t n = re.size();
  for (size_t i = 1, j = 0; i < n; ++i) {
    size_t bit = n >> 1;
    for (; j & bit; bit >>= 1) j ^= bit;
    j ^= bit;
    if (i < j) { std::swap(re[i], re[j]); std::swap(im[i], im[j]); }
  }
  for (size_t len = 2; len <= n; len <<= 1) {
    const size_t half = len >> 1;
    for (size_t k = 0; k < half; ++k) {
      // sincos() explicitly: glibc's sincos and its separate sin / cos differ in rare last bits, and compilers disagree on
      // merging the pair -- the response is bit-pinned (SPEC.md C2), so the call is part of its statement
      const double ang = (double)sign * 2.0 * kPi * (double)k / (double)len;
      double wr, wi;
      ::sincos(ang, &wi, &wr);
      for (size_t i = k; i < n; i += len) {
        const double ur = re[i], ui = im[i];
        const double vr = re[i + half] * wr - im[i + half] * wi, vi = re[i + half] * wi + im[i + half] * wr;
        re[i] = ur + vr; im[i] = ui + vi; re[i + half] = ur - vr; im[i + half] = ui - vi;
      }
    }
  }
}

struct Geom { unsigned size, halfsz, width, halfw, decimation; int center, log2s; uint32_t dphase; };

// SPEC.md C2: channel sizing (su_specttuner_open_channel's arithmetic as recollected in SURVEY.md Appendix C)
Geom design_geometry(unsigned W, double f0, double bw, double guard)
{
  Geom g{};
  double actual_bw = bw * guard;
  if (actual_bw > 2.0 * kPi) actual_bw = 2.0 * kPi;
  const double k = actual_bw / (2.0 * kPi);
  const unsigned min_size = (unsigned)std::ceil(k * (double)W - 1e-6);   // bw * guard = 2 pi / D must give W / D bins, not one more
  unsigned size = 1;
  while (size < min_size) size <<= 1;
  if (size < 2) size = 2;
  if (size > W) size = W;
  g.size = size; g.halfsz = size / 2;
  g.width = (unsigned)std::ceil((double)min_size / guard);
  if (g.width > size) g.width = size;
  g.halfw = std::max(1u, g.width >> 1);
  g.decimation = W / size;
  g.center = (int)(2.0 * std::floor(f0 / (4.0 * kPi) * (double)W + 0.5)) & (int)(W - 1);
  double f0w = std::fmod(f0, 2.0 * kPi);
  if (f0w < 0) f0w += 2.0 * kPi;
  double lo = f0w - (double)g.center * 2.0 * kPi / (double)W;
  if (lo > kPi) lo -= 2.0 * kPi;
  if (lo < -kPi) lo += 2.0 * kPi;
  g.dphase = (uint32_t)(int64_t)std::llround(-lo * (double)g.decimation / (2.0 * kPi) * 4294967296.0);
  g.log2s = 0;
  while ((1u << g.log2s) < size) ++g.log2s;
  return g;
}

// k h[i]: brick wall of 2 halfw bins -> time domain -> centred, Blackman-Harris, back -> frequency domain; k = 1/W
std::vector<c32> design_response(unsigned W, unsigned size, unsigned halfw)
{
  std::vector<double> re(size, 0.0), im(size, 0.0);
  const unsigned half = size / 2;
  for (unsigned i = 0; i < size; ++i) re[i] = (i < halfw || i >= size - halfw) ? 1.0 : 0.0;
  fft64(re, im, +1);
  for (unsigned i = 0; i < size; ++i) { re[i] /= (double)size; im[i] /= (double)size; }
  for (unsigned i = 0; i < half; ++i) { std::swap(re[i], re[i + half]); std::swap(im[i], im[i + half]); }
  for (unsigned i = 0; i < size; ++i) {
    const double t = 2.0 * kPi * (double)i / (double)(size - 1);
    const double w = 0.35875 - 0.48829 * std::cos(t) + 0.14128 * std::cos(2 * t) - 0.01168 * std::cos(3 * t);
    re[i] *= w; im[i] *= w;
  }
  for (unsigned i = 0; i < half; ++i) { std::swap(re[i], re[i + half]); std::swap(im[i], im[i + half]); }
  fft64(re, im, -1);
  std::vector<c32> hk(size);
  for (unsigned i = 0; i < size; ++i) {
    const bool pass = i < halfw || i >= size - halfw;
    hk[i].re = pass ? (float)(re[i] / (double)W) : 0.0f;
    hk[i].im = pass ? (float)(im[i] / (double)W) : 0.0f;
  }
  return hk;
}

std::vector<c32> twiddles(unsigned n)
{
  std::vector<c32> tw(n);
  for (unsigned i = 0; i < n; ++i) {
    const double ang = -2.0 * kPi * (double)i / (double)n;
    double c, s;
    ::sincos(ang, &s, &c);
    tw[i].re = (float)c; tw[i].im = (float)s;
  }
  return tw;
}

template <typename T> T *dev_upload_new(const std::vector<T> &v)
{
  void *p = nullptr;
  if (hipMalloc(&p, std::max<size_t>(1, v.size()) * sizeof(T)) != hipSuccess) return nullptr;
  if (!v.empty() && hipMemcpy(p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice) != hipSuccess) { (void)hipFree(p); return nullptr; }
  return static_cast<T *>(p);
}

struct Channel {
  bool open = false;
  Geom g{};
  bool precise = false;
  unsigned long long n_open = 0;       // the size group's output counter when the channel joined it
};

// all open channels of one inverse-transform size: one launch
struct SizeGroup {
  int log2s = 0;
  std::vector<int> members;            // channel indices, in table order
  sdk::StChan *d_chans = nullptr;
  c32 *d_hk = nullptr, *d_tw = nullptr;
  c32 *d_hkt = nullptr;                // per-channel responses, [block of 64 channels][bin][lane] (wavefront kernel)
  size_t hkt_blocks = 0;               // blocks in d_hkt (whole wavefronts' worth)
  c32 *d_handoff = nullptr;            // wavefront kernel: seam payload between consecutive runs, 16 KiB per run and block
  unsigned *d_flags = nullptr;         // one flag per run and block: the epoch of the launch that published it
  size_t ho_slots = 0;
  c32 *d_seam = nullptr;               // workgroup kernel: a run's last second half for its successor, [run][channel][size / 2]
  size_t seam_elems = 0;
  unsigned epoch = 0;                  // flag value of the next launch
  bool hk_uniform = false;             // one response for all members
  bool any_precise = false, all_precise = false;   // some / every member was opened `precise` (residual NCO on its outputs)
  int nsel = 0;                        // distinct responses of the members (tables in d_hk)
  float *d_win = nullptr;
  c32 *d_prev[2] = {nullptr, nullptr};
  int prev_cur = 0;
  unsigned long long nout = 0;         // outputs per channel emitted so far
  // membership changes are applied at the next feed (one table rebuild for any number of opens / closes); until then
  // the table the device last used and its cross-fade state are kept aside
  bool dirty = false;
  std::vector<int> snap_members;
  c32 *snap_prev = nullptr;
  void release()
  {
    for (void *p : {(void *)d_handoff, (void *)d_seam, (void *)d_flags, (void *)d_chans, (void *)d_hk, (void *)d_hkt, (void *)d_tw, (void *)d_win, (void *)d_prev[0], (void *)d_prev[1], (void *)snap_prev}) if (p) (void)hipFree(p);
    d_chans = nullptr; d_hk = d_tw = d_hkt = d_handoff = d_seam = nullptr; d_flags = nullptr; ho_slots = 0; seam_elems = 0; d_win = nullptr; d_prev[0] = d_prev[1] = nullptr; snap_prev = nullptr;
  }
};

}  // namespace

struct suamd_specttuner {
  suamd_ctx_t *ctx = nullptr;
  unsigned W = 4096, H = 2048;
  int log2w = 12;
  unsigned run = 0;                    // windows per workgroup of the workgroup kernel (0: planned per launch, sdk::st_plan_run)
  unsigned run_wave = 0;               // wavefront kernel: windows per wavefront (0: one round of 4 wavefronts per CU)
  unsigned slots = 0;                  // wavefront kernels: the launch's budget of the chip's 1024 window slots (0: SUAMD_ST_SLOTS, else 768)
  int seam_polls = 256;                // wavefront kernel: bounded wait for a run's successor (SUAMD_ST_SEAM_POLLS; 0: never wait)
  bool use_wave = true;                // sizes 8..64 go to specttuner_wave.hip (SUAMD_ST_KERNEL=wg keeps them on specttuner.hip)
  bool use_pair = true;                // 64-bin channels with one response: two wavefronts per window (specttuner_pair.hip; SUAMD_ST_KERNEL=wave keeps them on specttuner_wave.hip)
  c32 *d_tw_w = nullptr;
  c32 *d_hist[2] = {nullptr, nullptr};
  int hist_cur = 0;
  bool have_hist = false;
  std::vector<Channel> ch;
  std::vector<int> closed_pending;     // closed since the last feed: not handed out again before the tables are rebuilt
  std::map<int, SizeGroup> groups;
};

namespace {

// Order of a size group's channels in the kernel's table.  In the channel stage a group of TPI = size/16 lanes serves one
// channel and reads its bins straight from the LDS spectrum; the 32 lanes of half a wavefront (= 32/TPI channels) go
// through the 32 bank pairs together, so two channels of one half wavefront whose bins fall on the same bank pairs
// serialise (a dense raster like C4's -- centres 6 or 8 bins apart -- gave 4-way conflicts and +25 % kernel time in
// table order).  Which channels share a half wavefront is free: a greedy pass puts those together whose accesses
// collide least.  Per channel and operand slot the bank pairs its lanes touch are a 32-bit mask.
std::vector<int> bank_friendly_order(unsigned W, int log2s, const std::vector<int> &members, const std::vector<Channel> &ch)
{
  const unsigned S = 1u << log2s, HS = S / 2;
  const unsigned TPI = log2s == 5 ? 4 : (log2s < 4 ? 1 : S / 16), E = S / TPI;
  if (TPI >= 32 || members.size() < 2) return members;
  const unsigned per_half = 32 / TPI;
  // pass-0 operand geometry of fft_core's plan: i = tl + b*TPI + q*(S/R0); as a set over (b, q) it is {tl + TPI*e, e < E}
  std::vector<std::vector<uint32_t>> mask(members.size(), std::vector<uint32_t>(E, 0));
  for (size_t k = 0; k < members.size(); ++k) {
    const int center = ch[members[k]].g.center;
    for (unsigned e = 0; e < E; ++e)
      for (unsigned tl = 0; tl < TPI; ++tl) {
        const unsigned i = tl + TPI * e;
        const unsigned idx = ((unsigned)center + i + (i < HS ? 0 : W - S)) & (W - 1);
        mask[k][e] |= 1u << ((idx + (idx >> 4)) & 31);
      }
  }
  std::vector<char> used(members.size(), 0);
  std::vector<int> order;
  std::vector<size_t> half;
  size_t next_free = 0;
  while (order.size() < members.size()) {
    if (half.size() == per_half) half.clear();
    size_t best = members.size();
    if (half.empty()) {
      while (used[next_free]) ++next_free;
      best = next_free;
    } else {
      unsigned best_cost = ~0u;
      for (size_t k = 0; k < members.size(); ++k) {
        if (used[k]) continue;
        unsigned cost = 0;
        for (size_t o : half) for (unsigned e = 0; e < E; ++e) cost += (unsigned)__builtin_popcount(mask[k][e] & mask[o][e]);
        if (cost < best_cost) { best_cost = cost; best = k; if (!cost) break; }
      }
    }
    used[best] = 1; half.push_back(best); order.push_back(members[best]);
  }
  return order;
}

// The wavefront kernel (sizes 8..64) maps channel k of the table to lane k mod 64 and reads two bins per lane with
// ds_read_b128: the hardware serves that instruction in four groups of 16 lanes (MI355X_MICROARCH.md, LDS table), a
// lane covering the bank quad (centre/2 + i/2) mod 16.  Within each block of 64 channels the lanes are dealt so that a
// 16-lane group sees every residue centre/2 mod 16 as few times as possible.
bool wave_kernel_size(int log2s) { return log2s >= 3 && log2s <= 6; }

std::vector<int> wave_friendly_order(const std::vector<int> &members, const std::vector<Channel> &ch)
{
  static const int group_lanes[4][16] = {{0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27},
                                         {4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31},
                                         {32, 33, 34, 35, 44, 45, 46, 47, 52, 53, 54, 55, 56, 57, 58, 59},
                                         {36, 37, 38, 39, 40, 41, 42, 43, 48, 49, 50, 51, 60, 61, 62, 63}};
  std::vector<int> order(members.size());
  for (size_t b0 = 0; b0 < members.size(); b0 += 64) {
    const size_t nb = std::min<size_t>(64, members.size() - b0);
    // most frequent residues first: they are the ones that need spreading
    int freq[16] = {0};
    for (size_t k = 0; k < nb; ++k) ++freq[(ch[members[b0 + k]].g.center / 2) & 15];
    std::vector<size_t> idx(nb);
    for (size_t k = 0; k < nb; ++k) idx[k] = k;
    std::stable_sort(idx.begin(), idx.end(), [&](size_t x, size_t y) {
      return freq[(ch[members[b0 + x]].g.center / 2) & 15] > freq[(ch[members[b0 + y]].g.center / 2) & 15];
    });
    int fill[4] = {0, 0, 0, 0}, have[4][16] = {{0}};
    std::vector<int> lane_of(nb, -1);
    std::vector<char> lane_used(64, 0);
    for (size_t k : idx) {
      const int r = (ch[members[b0 + k]].g.center / 2) & 15;
      int best = -1;
      for (int g = 0; g < 4; ++g) {
        // only lanes < nb exist in a short last block
        int free_lane = -1;
        for (int l : group_lanes[g]) if ((size_t)l < nb && !lane_used[l]) { free_lane = l; break; }
        if (free_lane < 0) continue;
        if (best < 0 || have[g][r] < have[best][r] || (have[g][r] == have[best][r] && fill[g] < fill[best])) best = g;
      }
      for (int l : group_lanes[best]) if ((size_t)l < nb && !lane_used[l]) { lane_of[k] = l; lane_used[l] = 1; break; }
      ++have[best][r]; ++fill[best];
    }
    for (size_t k = 0; k < nb; ++k) order[b0 + (size_t)lane_of[k]] = members[b0 + k];
  }
  return order;
}

// (re)builds a group's device tables after its membership changed; surviving members keep their cross-fade state
bool rebuild_group(suamd_specttuner *st, SizeGroup &g, const std::vector<int> &old_members, c32 *old_prev)
{
  const unsigned S = 1u << g.log2s, HS = S / 2;
  const size_t n = g.members.size();
  const bool wave = st->use_wave && wave_kernel_size(g.log2s);
  g.members = wave ? wave_friendly_order(g.members, st->ch) : bank_friendly_order(st->W, g.log2s, g.members, st->ch);
  std::vector<sdk::StChan> tab(n);
  std::vector<unsigned> halfws;
  for (size_t k = 0; k < n; ++k) {
    const Channel &c = st->ch[g.members[k]];
    size_t sel = std::find(halfws.begin(), halfws.end(), c.g.halfw) - halfws.begin();
    if (sel == halfws.size()) halfws.push_back(c.g.halfw);
    tab[k].center = c.g.center; tab[k].hsel = (int)sel; tab[k].row = g.members[k]; tab[k].precise = c.precise ? 1 : 0;
    tab[k].dphase = c.g.dphase;
    tab[k].n_open = c.n_open;
  }
  std::vector<c32> hk;
  for (unsigned hw : halfws) { std::vector<c32> h = design_response(st->W, S, hw); hk.insert(hk.end(), h.begin(), h.end()); }
  g.hk_uniform = halfws.size() == 1;
  g.any_precise = false; g.all_precise = !tab.empty();
  for (const sdk::StChan &c : tab) { g.any_precise |= c.precise != 0; g.all_precise &= c.precise != 0; }
  g.nsel = (int)halfws.size();
  if (g.d_chans) (void)hipFree(g.d_chans);
  if (g.d_hk) (void)hipFree(g.d_hk);
  if (g.d_hkt) (void)hipFree(g.d_hkt);
  g.d_hkt = nullptr;
  g.d_chans = dev_upload_new(tab);
  g.d_hk = dev_upload_new(hk);
  if (wave) {
    // stw_kernel reads one [bin][lane] block per lane group: a wavefront of channels below 64 bins serves 64 / S groups of 64
    // channels, and EVERY group's block is loaded whether or not its channels exist (the buffer descriptor bounds one block,
    // not the table) -- so the table holds whole wavefronts' worth.  Rounds 2-3 sized it ceil(n / 64): a bank of fewer than
    // 64 * 64 / S narrow channels with per-channel responses was read up to 28 KiB beyond its end, a fault whenever the
    // table happened to end a mapped region (found with eight analyzer shards on one device, tools/live_fault_repro.py).
    const size_t per_wave = sdk::stw_channels_per_wave(g.log2s) / 64;
    const size_t nblk = std::max<size_t>(1, (n + 64 * per_wave - 1) / (64 * per_wave)) * per_wave;
    g.hkt_blocks = nblk;
    std::vector<c32> hkt(nblk * S * 64, c32{0.f, 0.f});
    for (size_t k = 0; k < n; ++k)
      for (unsigned i = 0; i < S; ++i) hkt[((k >> 6) * S + i) * 64 + (k & 63)] = hk[(size_t)tab[k].hsel * S + i];
    g.d_hkt = dev_upload_new(hkt);
    if (!g.d_hkt) return false;
  }
  if (!g.d_tw) g.d_tw = dev_upload_new(twiddles(S));
  if (!g.d_win) {
    std::vector<float> win(S);
    for (unsigned i = 0; i < S; ++i) { const double s = std::sin(kPi * (double)i / (double)S); win[i] = (float)(s * s); }
    g.d_win = dev_upload_new(win);
  }
  c32 *np[2] = {nullptr, nullptr};
  for (int p = 0; p < 2; ++p) {
    if (hipMalloc((void **)&np[p], std::max<size_t>(1, n) * HS * sizeof(c32)) != hipSuccess) return false;
    if (hipMemset(np[p], 0, std::max<size_t>(1, n) * HS * sizeof(c32)) != hipSuccess) return false;
  }
  if (old_prev) {
    for (size_t k = 0; k < n; ++k) {
      const auto it = std::find(old_members.begin(), old_members.end(), g.members[k]);
      if (it == old_members.end()) continue;
      const size_t ok = (size_t)(it - old_members.begin());
      if (hipMemcpy(np[0] + k * HS, old_prev + ok * HS, HS * sizeof(c32), hipMemcpyDeviceToDevice) != hipSuccess) return false;
    }
  }
  // the fills and copies above went to the null stream and may still be on their way; the feeds run on the caller's
  // stream, which need not wait for it (a non-blocking stream does not): a cross-fade state zeroed AFTER the first
  // windows had written it showed up as one wrong half window per rebuild, once in a few dozen runs with several
  // analyzer shards on one device
  if (hipStreamSynchronize(nullptr) != hipSuccess) return false;
  for (int p = 0; p < 2; ++p) { if (g.d_prev[p]) (void)hipFree(g.d_prev[p]); g.d_prev[p] = np[p]; }
  g.prev_cur = 0;
  g.dirty = false;
  return g.d_chans && g.d_hk && g.d_tw && g.d_win;
}

// first membership change since the last feed: set the device's view of the group aside
void touch_group(SizeGroup &g)
{
  if (g.dirty) return;
  (void)hipDeviceSynchronize();                               // a feed still in flight reads the old tables
  g.snap_members = g.members;
  g.snap_prev = g.d_prev[g.prev_cur];
  g.d_prev[g.prev_cur] = nullptr;
  g.dirty = true;
}

}  // namespace

extern "C" {

suamd_specttuner_t *suamd_specttuner_new(suamd_ctx_t *ctx, unsigned window_size)
{
  if (!ctx) { suamd_set_error("null context"); return nullptr; }
  if (window_size != 4096) { suamd_set_error("specttuner window_size %u unsupported (4096, su_specttuner's default)", window_size); return nullptr; }
  if (hipSetDevice(suamd_ctx_device(ctx)) != hipSuccess) { suamd_set_error("hipSetDevice failed"); return nullptr; }
  auto *st = new (std::nothrow) suamd_specttuner();
  if (!st) { suamd_set_error("out of memory"); return nullptr; }
  st->ctx = ctx; st->W = window_size; st->H = window_size / 2; st->log2w = 12;
  {
    const sdk::Tuning &tn = sdk::tuning();                    // (a tuner keeps the plan it was made under)
    if (tn.st_run >= 1) st->run = st->run_wave = (unsigned)tn.st_run;
    st->use_wave = tn.st_kernel != 2;
    st->use_pair = tn.st_kernel == 0;
    if (tn.st_seam_polls >= 0) st->seam_polls = (int)tn.st_seam_polls;
  }
  st->d_tw_w = dev_upload_new(twiddles(st->W));
  bool ok = st->d_tw_w != nullptr;
  for (int p = 0; p < 2 && ok; ++p) ok = hipMalloc((void **)&st->d_hist[p], st->H * sizeof(c32)) == hipSuccess;
  if (!ok) { suamd_set_error("device allocation failed"); suamd_specttuner_destroy(st); return nullptr; }
  return st;
}

void suamd_specttuner_destroy(suamd_specttuner_t *st)
{
  if (!st) return;
  for (auto &kv : st->groups) kv.second.release();
  if (st->d_tw_w) (void)hipFree(st->d_tw_w);
  for (int p = 0; p < 2; ++p) if (st->d_hist[p]) (void)hipFree(st->d_hist[p]);
  delete st;
}

int suamd_specttuner_open_channel(suamd_specttuner_t *st, double f0, double bw, double guard, SUBOOL precise)
{
  if (!st) { suamd_set_error("null specttuner"); return -1; }
  if (!(bw > 0) || !(guard >= 1) || !std::isfinite(f0)) { suamd_set_error("bad channel parameters (bw > 0, guard >= 1)"); return -1; }
  if (hipSetDevice(suamd_ctx_device(st->ctx)) != hipSuccess) { suamd_set_error("hipSetDevice failed"); return -1; }
  Channel c;
  c.open = true; c.g = design_geometry(st->W, f0, bw, guard); c.precise = precise != 0;
  int idx = -1;
  for (size_t i = 0; i < st->ch.size(); ++i)
    if (!st->ch[i].open && std::find(st->closed_pending.begin(), st->closed_pending.end(), (int)i) == st->closed_pending.end()) { idx = (int)i; break; }
  if (idx < 0) { st->ch.push_back(Channel()); idx = (int)st->ch.size() - 1; }
  SizeGroup &g = st->groups[c.g.log2s];
  g.log2s = c.g.log2s;
  c.n_open = g.nout;
  st->ch[idx] = c;
  touch_group(g);
  g.members.push_back(idx);
  return idx;
}

SUBOOL suamd_specttuner_close_channel(suamd_specttuner_t *st, int channel)
{
  if (!st || channel < 0 || (size_t)channel >= st->ch.size() || !st->ch[channel].open) { suamd_set_error("no such channel"); return SU_FALSE; }
  if (hipSetDevice(suamd_ctx_device(st->ctx)) != hipSuccess) { suamd_set_error("hipSetDevice failed"); return SU_FALSE; }
  SizeGroup &g = st->groups[st->ch[channel].g.log2s];
  touch_group(g);
  g.members.erase(std::find(g.members.begin(), g.members.end(), channel));
  st->ch[channel].open = false;
  st->closed_pending.push_back(channel);
  return SU_TRUE;
}

unsigned suamd_specttuner_channel_size(const suamd_specttuner_t *st, int c)
{
  return (st && c >= 0 && (size_t)c < st->ch.size() && st->ch[c].open) ? st->ch[c].g.size : 0;
}
unsigned suamd_specttuner_channel_decimation(const suamd_specttuner_t *st, int c)
{
  return (st && c >= 0 && (size_t)c < st->ch.size() && st->ch[c].open) ? st->ch[c].g.decimation : 0;
}

unsigned suamd_specttuner_channel_capacity(const suamd_specttuner_t *st) { return st ? (unsigned)st->ch.size() : 0; }

SUBOOL suamd_specttuner_reset(suamd_specttuner_t *st, void *stream)
{
  if (!st) { suamd_set_error("null specttuner"); return SU_FALSE; }
  if (hipSetDevice(suamd_ctx_device(st->ctx)) != hipSuccess) { suamd_set_error("hipSetDevice failed"); return SU_FALSE; }
  hipStream_t s = static_cast<hipStream_t>(stream);
  st->have_hist = false;
  for (auto &kv : st->groups) {
    SizeGroup &g = kv.second;
    const size_t HS = ((size_t)1 << g.log2s) / 2;
    if (g.dirty) {
      // the tables are rebuilt at the next feed from the snapshot: drop what the rebuild would carry over (it zero-fills
      // a group that has no snapshot).  Not a memset on `s`: rebuild_group copies with blocking calls on the null stream,
      // which a non-blocking caller stream is not ordered with -- the snapshot's last writer was synchronised
      // when it was taken, so freeing it here is safe
      if (g.snap_prev) { (void)hipFree(g.snap_prev); g.snap_prev = nullptr; }
      g.snap_members.clear();
    } else if (g.d_prev[g.prev_cur] && !g.members.empty() &&
               hipMemsetAsync(g.d_prev[g.prev_cur], 0, g.members.size() * HS * sizeof(c32), s) != hipSuccess) { suamd_set_error("reset failed"); return SU_FALSE; }
  }
  return SU_TRUE;
}

SUBOOL suamd_specttuner_design(unsigned window_size, double f0, double bw, double guard, uint32_t geom[6], SUFLOAT *hk)
{
  if (window_size != 4096 || !(bw > 0) || !(guard >= 1) || !std::isfinite(f0) || !geom) { suamd_set_error("bad channel parameters (window 4096, bw > 0, guard >= 1)"); return SU_FALSE; }
  const Geom g = design_geometry(window_size, f0, bw, guard);
  geom[0] = g.size; geom[1] = g.halfsz; geom[2] = g.halfw; geom[3] = g.decimation; geom[4] = (uint32_t)g.center; geom[5] = g.dphase;
  if (hk) {
    const std::vector<c32> h = design_response(window_size, g.size, g.halfw);
    std::memcpy(hk, h.data(), h.size() * sizeof(c32));
  }
  return SU_TRUE;
}

SUBOOL suamd_specttuner_set_slots(suamd_specttuner_t *st, unsigned slots)
{
  if (!st || (slots != 0 && (slots < 64 || slots > 4096))) { suamd_set_error("slots out of range (0, or 64 .. 4096)"); return SU_FALSE; }
  st->slots = slots;
  return SU_TRUE;
}

SUBOOL suamd_specttuner_set_run(suamd_specttuner_t *st, unsigned run)
{
  if (!st || run < 1 || run > 4096) { suamd_set_error("run out of range"); return SU_FALSE; }
  st->run = st->run_wave = run;
  return SU_TRUE;
}

// (mix_y: channels of at most mix_max_size bins go through the view {mix_y, mix_view}, the others through d_y / d_rows)
static SUBOOL st_feed(suamd_specttuner_t *st, const suamd_complex *d_x, SUSCOUNT len, suamd_complex *d_y, suamd_view view,
                      suamd_complex *const *d_rows, SUSCOUNT *counts, void *stream, size_t rows_span = 0,
                      suamd_complex *mix_y = nullptr, suamd_view mix_view = suamd_view{0, 1}, unsigned mix_max_size = 0);

SUBOOL suamd_specttuner_feed(suamd_specttuner_t *st, const suamd_complex *d_x, SUSCOUNT len, suamd_complex *d_y, suamd_view view,
                             SUSCOUNT *counts, void *stream)
{
  return st_feed(st, d_x, len, d_y, view, nullptr, counts, stream);
}

SUBOOL suamd_specttuner_feed_rows(suamd_specttuner_t *st, const suamd_complex *d_x, SUSCOUNT len, suamd_complex *const *d_rows,
                                  SUSCOUNT *counts, void *stream)
{
  if (!d_rows) { suamd_set_error("null row table"); return SU_FALSE; }
  return st_feed(st, d_x, len, nullptr, suamd_view{0, 1}, d_rows, counts, stream);
}

SUBOOL suamd_specttuner_feed_rows_near(suamd_specttuner_t *st, const suamd_complex *d_x, SUSCOUNT len, suamd_complex *const *d_rows,
                                       const void *d_base, size_t span_bytes, SUSCOUNT *counts, void *stream)
{
  if (!d_rows) { suamd_set_error("null row table"); return SU_FALSE; }
  if (!d_base || span_bytes == 0 || span_bytes >= ((size_t)1 << 31))
    return st_feed(st, d_x, len, nullptr, suamd_view{0, 1}, d_rows, counts, stream);          // no usable promise: 64-bit addressing
  // (d_y carries the base: with a row table the kernels take row starts from it and, rows_span set, offsets from d_y)
  return st_feed(st, d_x, len, static_cast<suamd_complex *>(const_cast<void *>(d_base)), suamd_view{0, 1}, d_rows, counts, stream, span_bytes);
}

SUBOOL suamd_specttuner_feed_mixed(suamd_specttuner_t *st, const suamd_complex *d_x, SUSCOUNT len, suamd_complex *d_y, suamd_view view,
                                   unsigned view_max_size, suamd_complex *const *d_rows, const void *d_base, size_t span_bytes,
                                   SUSCOUNT *counts, void *stream)
{
  if (!d_y || !d_rows) { suamd_set_error("null output"); return SU_FALSE; }
  const bool near = d_base && span_bytes && span_bytes < ((size_t)1 << 31);
  return st_feed(st, d_x, len, near ? static_cast<suamd_complex *>(const_cast<void *>(d_base)) : nullptr, suamd_view{0, 1}, d_rows, counts, stream,
                 near ? span_bytes : 0, d_y, view, view_max_size);
}

static SUBOOL st_feed(suamd_specttuner_t *st, const suamd_complex *d_x, SUSCOUNT len, suamd_complex *d_y_all, suamd_view view_all,
                      suamd_complex *const *d_rows_all, SUSCOUNT *counts, void *stream, size_t rows_span_all,
                      suamd_complex *mix_y, suamd_view mix_view, unsigned mix_max_size)
{
  if (!st || (len && !d_x)) { suamd_set_error("null argument"); return SU_FALSE; }
  if (len % st->H) { suamd_set_error("len must be a multiple of half a window (%u)", st->H); return SU_FALSE; }
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (counts) for (size_t i = 0; i < st->ch.size(); ++i) counts[i] = 0;
  if (len == 0) return SU_TRUE;
  const long long nwin = (long long)(len / st->H) - (st->have_hist ? 0 : 1);
  bool hist_written = false;
  if (nwin > 0) {
    for (auto &kv : st->groups) {
      SizeGroup &g = kv.second;
      if (g.dirty) {
        const bool ok = g.members.empty() || rebuild_group(st, g, g.snap_members, g.snap_prev);
        if (g.snap_prev) (void)hipFree(g.snap_prev);
        g.snap_prev = nullptr; g.snap_members.clear(); g.dirty = false;
        if (!ok) { suamd_set_error("device allocation failed"); return SU_FALSE; }
      }
      if (g.members.empty()) continue;
      // where this size group's samples go: the mixed feed's view for the narrow sizes, the caller's rows / view otherwise
      const bool to_mix = mix_y && (1u << g.log2s) <= mix_max_size;
      suamd_complex *d_y = to_mix ? mix_y : d_y_all;
      const suamd_view view = to_mix ? mix_view : view_all;
      suamd_complex *const *d_rows = to_mix ? nullptr : d_rows_all;
      const size_t rows_span = to_mix ? 0 : rows_span_all;
      if (!d_y && !d_rows) { suamd_set_error("null output"); return SU_FALSE; }
      sdk::StArgs a{};
      a.x = d_x; a.hist = st->d_hist[st->hist_cur]; a.have_hist = st->have_hist ? 1 : 0;
      a.nwin = nwin; a.run = st->run ? (int)st->run : 3;
      a.tw_w = st->d_tw_w; a.tw_s = g.d_tw;
      a.chans = g.d_chans; a.nchan = (int)g.members.size();
      a.hk = g.d_hk; a.win = g.d_win;
      a.prev_in = g.d_prev[g.prev_cur]; a.prev_out = g.d_prev[g.prev_cur ^ 1];
      a.n0 = g.nout;
      a.y = d_y; a.yv = sdk::View{(long long)view.chan_stride, (long long)view.time_stride};
      a.rows = reinterpret_cast<const void *const *>(d_rows);
      hipError_t e;
      if (g.d_hkt) {
        if (!hist_written) { a.hist_out = st->d_hist[st->hist_cur ^ 1]; hist_written = true; }   // the kernel carries the history over
        // One wavefront per window and SIMD (it takes the whole register file), 1024 slots on the chip.  A launch must
        // fit ONE round with room to spare: the recurrence kernels of earlier blocks hold a few SIMDs for milliseconds, and
        // a wavefront that finds no free SIMD waits for a whole run of another -- 1024 wavefronts of 2 windows measured
        // 32 us on an idle chip and 56 us inside the pipeline.  3/4 of the slots: 683 wavefronts of 3 windows for a
        // 4 Mi-sample block.
        a.hkt = g.d_hkt;
        {
          const size_t cpw = (size_t)sdk::stw_channels_per_wave(g.log2s);
          if (g.hkt_blocks < (g.members.size() + cpw - 1) / cpw * (cpw / 64)) { suamd_set_error("internal: response table smaller than the launch"); return SU_FALSE; }
        }
        a.hk_uniform = g.hk_uniform ? 1 : 0;
        a.any_precise = !g.any_precise ? 0 : g.all_precise ? 2 : 1;
        a.nsel = g.nsel;
        {
          // 32-bit buffer addressing of the outputs when the whole view of this feed lies below 2 GiB
          const long long hs = (1ll << g.log2s) / 2;
          const long long last = ((long long)st->ch.size() * (long long)view.chan_stride + (nwin * hs + hs) * (long long)view.time_stride) * 8;
          const bool no_y32 = sdk::tuning().st_y32 == 0;          // debug: 64-bit addressing everywhere
          a.y32 = (!d_rows && last < (1ll << 31) && !no_y32) ? 1 : 0;
          // rows promised to start within rows_span bytes of d_y: offsets from there, if the feed's own extent fits too
          if (d_rows && rows_span && !no_y32 && (long long)rows_span + (nwin * hs + hs) * 8 < (1ll << 31)) a.y32 = 1;
        }
        if (st->run_wave) a.run = (int)st->run_wave;
        else {
          const long long ny = ((long long)g.members.size() + sdk::stw_channels_per_wave(g.log2s) - 1) / sdk::stw_channels_per_wave(g.log2s);
          // (SUAMD_ST_SLOTS: the wavefront budget of a launch, default 768; a long block rounds up to whole windows per
          // wavefront far below the budget anyway, and there a larger budget is pure gain)
          const long long slots = sdk::tuning().st_slots >= 64 ? sdk::tuning().st_slots : 0;
          const long long budget = st->slots ? st->slots : (slots ? slots : 768);
          a.run = (int)std::max<long long>(1, (nwin * ny + budget - 1) / budget);
        }
        {
          const long long ny = ((long long)g.members.size() + sdk::stw_channels_per_wave(g.log2s) - 1) / sdk::stw_channels_per_wave(g.log2s);
          const size_t slots = (size_t)(((nwin + a.run - 1) / a.run) * ny);
          if (slots > g.ho_slots) {
            (void)hipDeviceSynchronize();                      // an earlier feed may still use the smaller buffers
            if (g.d_handoff) (void)hipFree(g.d_handoff);
            if (g.d_flags) (void)hipFree(g.d_flags);
            g.d_handoff = nullptr; g.d_flags = nullptr; g.ho_slots = 0;
            if (hipMalloc((void **)&g.d_handoff, slots * 2048 * sizeof(c32)) != hipSuccess ||
                hipMalloc((void **)&g.d_flags, slots * sizeof(unsigned)) != hipSuccess ||
                hipMemset(g.d_flags, 0, slots * sizeof(unsigned)) != hipSuccess || hipStreamSynchronize(nullptr) != hipSuccess) { suamd_set_error("device allocation failed"); return SU_FALSE; }
            g.ho_slots = slots;
          }
          a.handoff = g.d_handoff; a.flags = g.d_flags;
          if (++g.epoch == 0) g.epoch = 1;
          a.epoch = g.epoch;
          a.seam_polls = st->seam_polls;
        }
#ifdef STW_TSTAMP
        static unsigned long long *d_ts = nullptr;
        const size_t nts = (size_t)((nwin + a.run - 1) / a.run) * (size_t)a.run * 16;
        if (!d_ts) (void)hipMalloc((void **)&d_ts, (1 << 20) * sizeof(unsigned long long));
        (void)hipMemsetAsync(d_ts, 0, nts * sizeof(unsigned long long), s);
        a.tstamp = d_ts;
#endif
        // (with 64-bit output addressing -- row pointers, views beyond 2 GiB -- the 8- and 16-bin instantiations of the
        // two-wavefront kernel spill registers and measured 36-37 against 35 us per 4 Mi-sample block: those stay)
        const bool pair = st->use_pair && a.nsel >= 1 && a.nsel <= sdk::stp_max_responses() && a.run >= 2 && (a.y32 || g.log2s >= 5);
        e = pair ? sdk::specttuner_feed_pair(g.log2s, a, s) : sdk::specttuner_feed_wave(g.log2s, a, s);
#ifdef STW_TSTAMP
        {
          (void)hipStreamSynchronize(s);
          std::vector<unsigned long long> ts(nts);
          (void)hipMemcpy(ts.data(), d_ts, nts * sizeof(unsigned long long), hipMemcpyDeviceToHost);
          const size_t nr = (size_t)((nwin + a.run - 1) / a.run);
          unsigned long long tmin = 0;
          {
            double pro = 0, tail = 0, life = 0; size_t nw = 0;
            unsigned long long lo[8], hi[8];
            for (int q = 0; q < 8; ++q) { lo[q] = ~0ull; hi[q] = 0; }
            for (size_t r = 0; r < nr; ++r) {
              const unsigned long long *p0 = &ts[(r * a.run) * 16];
              unsigned long long e1 = 0;
              for (int k = 0; k < a.run; ++k) e1 = std::max(e1, ts[(r * a.run + k) * 16 + 9]);
              if (!p0[10] || !p0[11]) continue;
              pro += (double)(p0[0] - p0[10]); tail += (double)(p0[11] - e1); life += (double)(p0[11] - p0[10]); ++nw;
              lo[r & 7] = std::min(lo[r & 7], p0[10]); hi[r & 7] = std::max(hi[r & 7], p0[11]);
            }
            double span = 0; for (int q = 0; q < 8; ++q) span += (double)(hi[q] - lo[q]) / 8;
            std::fprintf(stderr, "stw tstamp: %zu runs x %d; per wave: entry -> first window %.0f, last window -> exit %.0f, life %.0f ticks; per XCD first entry -> last exit %.0f ticks\n",
                         nr, a.run, pro / nw, tail / nw, life / nw, span);
          }
          for (int k = 0; k < a.run; ++k) {
            double acc[10] = {0}; double start = 0; size_t cnt = 0;
            for (size_t r = 0; r < nr; ++r) { const unsigned long long *p = &ts[(r * a.run + k) * 16]; if (!p[0]) continue; ++cnt; start += 0.0 * (double)tmin; for (int n = 1; n < 10; ++n) acc[n] += (double)(p[n] - p[n - 1]); }
            std::fprintf(stderr, "  window %d: start %.0f | dft1 %.0f issue %.0f twid %.0f (ts3) transp %.0f dft2 %.0f hk+spec %.0f gather %.0f ifft %.0f xfade+store %.0f\n", k, start / cnt,
                         acc[1] / cnt, acc[2] / cnt, acc[3] / cnt, acc[4] / cnt, acc[5] / cnt, acc[6] / cnt, acc[7] / cnt, acc[8] / cnt, acc[9] / cnt);
          }
        }
#endif
      } else {
        if (!st->run) a.run = sdk::st_plan_run(g.log2s, a.nchan, nwin);
        {
          // seam buffer of the runs: no warm-up window, the first block of every later run completed by st_seam_kernel.
          // Banks that fit one workgroup side by side only (measured per 4 Mi samples, tools/st_wide.py: one channel of 256
          // bins 49.1 -> 44.7 us, of 128 bins 46.6 -> 40.3; 64 x 256 bins 136 -> 143, 128 x 256 267 -> 284: there a run is
          // long and the warm-up window a ninth of it, less than the second launch costs).  SUAMD_ST_SEAM=0 / 1: never / always.
          // (sdk::tuning().st_seam, read on every feed: a test that flips it inside one process gets the other path)
          const int seam_env = (int)sdk::tuning().st_seam;
          const size_t nruns = (size_t)((nwin + a.run - 1) / a.run);
          // st_seam_kernel takes the runs in grid.y: beyond 65535 of them (one-window runs of a narrow size on a very long
          // feed) the launch keeps its warm-up windows instead
          const bool seam_off = seam_env == 0 || (seam_env < 0 && a.nchan > sdk::st_channels_per_group(g.log2s)) || nruns - 1 > 65535;
          const size_t need = nruns > 1 ? (nruns - 1) * (size_t)a.nchan * ((size_t)1 << (g.log2s - 1)) : 0;
          a.handoff = nullptr;
          if (!seam_off && need && g.log2s >= 1 && need * sizeof(c32) <= ((size_t)256 << 20)) {
            if (need > g.seam_elems) {
              if (g.d_seam) (void)hipFree(g.d_seam);
              g.d_seam = nullptr; g.seam_elems = 0;
              if (hipMalloc((void **)&g.d_seam, need * sizeof(c32)) == hipSuccess) g.seam_elems = need;
              else (void)hipGetLastError();
            }
            a.handoff = g.d_seam;                                 // (null: the launch falls back to warm-up windows)
          }
        }
        e = sdk::specttuner_feed(st->log2w, g.log2s, a, s);
      }
      if (e != hipSuccess) { suamd_set_error("specttuner launch failed: %s", hipGetErrorString(e)); return SU_FALSE; }
      g.prev_cur ^= 1;
      const unsigned HS = (1u << g.log2s) / 2;
      g.nout += (unsigned long long)nwin * HS;
      if (counts) for (int c : g.members) counts[c] = (SUSCOUNT)nwin * HS;
    }
    st->closed_pending.clear();                               // every group's table has been rebuilt
  }
  // the last half window is the next feed's history
  if (!hist_written &&
      hipMemcpyAsync(st->d_hist[st->hist_cur ^ 1], reinterpret_cast<const c32 *>(d_x) + (len - st->H), st->H * sizeof(c32),
                     hipMemcpyDeviceToDevice, s) != hipSuccess) { suamd_set_error("history copy failed"); return SU_FALSE; }
  st->hist_cur ^= 1;
  st->have_hist = true;
  return SU_TRUE;
}

}  // extern "C"

// ==========================================================================================================
// libsigutils front end: su_specttuner_* (include/sigutils/specttuner.h)
// ==========================================================================================================
#include <complex>
#define SUCOMPLEX std::complex<float>
struct sigutils_specttuner_params { SUSCOUNT window_size; SUBOOL early_windowing; };
struct sigutils_specttuner_channel;
struct sigutils_specttuner_channel_params {
  SUFLOAT f0, delta_f, bw, guard; SUBOOL precise; void *privdata;
  SUBOOL (*on_data)(const struct sigutils_specttuner_channel *channel, void *privdata, const SUCOMPLEX *data, SUSCOUNT size);
};
struct sigutils_specttuner;
struct sigutils_specttuner_channel {
  sigutils_specttuner_channel_params params;
  sigutils_specttuner *owner;
  int index;
};

struct sigutils_specttuner {
  suamd_ctx_t *ctx = nullptr;
  suamd_specttuner_t *st = nullptr;
  unsigned W = 0, H = 0;
  static constexpr unsigned CAP_HALVES = 64;           // staging capacity, half windows
  c32 *h_in = nullptr, *d_in = nullptr;                // pinned staging / device input
  c32 *h_out = nullptr, *d_out = nullptr;              // [channel][row_len]
  size_t out_rows = 0, row_len = 0;
  size_t fill = 0;
  hipStream_t stream = nullptr;
  std::vector<std::unique_ptr<sigutils_specttuner_channel>> channels;

  bool ensure_out(size_t rows)
  {
    // a channel yields size/2 <= W/2 samples per half window fed: CAP_HALVES * H per flush at most
    if (rows <= out_rows) return true;
    if (h_out) (void)hipHostFree(h_out);
    if (d_out) (void)hipFree(d_out);
    h_out = d_out = nullptr;
    out_rows = std::max<size_t>(rows, 4); row_len = (size_t)CAP_HALVES * H + 16;
    if (hipHostMalloc((void **)&h_out, out_rows * row_len * sizeof(c32), hipHostMallocDefault) != hipSuccess) return false;
    if (hipMalloc((void **)&d_out, out_rows * row_len * sizeof(c32)) != hipSuccess) return false;
    return true;
  }

  // runs every complete half window in the staging buffer through the device and hands the results out
  bool flush()
  {
    const size_t nproc = (fill / H) * H;
    if (nproc == 0) return true;
    size_t rows = 0;
    for (auto &c : channels) if (c) rows = std::max<size_t>(rows, (size_t)c->index + 1);
    if (!ensure_out(std::max<size_t>(rows, 1))) { suamd_set_error("allocation failed"); return false; }
    if (hipMemcpyAsync(d_in, h_in, nproc * sizeof(c32), hipMemcpyHostToDevice, stream) != hipSuccess) return false;
    // one entry per slot of the tuner's channel table (it never shrinks: closing the highest channel leaves its slot)
    std::vector<SUSCOUNT> counts(std::max<size_t>({rows, (size_t)suamd_specttuner_channel_capacity(st), (size_t)1}), 0);
    const suamd_view v{(SUSCOUNT)row_len, 1};
    if (!suamd_specttuner_feed(st, reinterpret_cast<const suamd_complex *>(d_in), nproc, reinterpret_cast<suamd_complex *>(d_out), v,
                               counts.data(), stream)) return false;
    for (auto &c : channels) {
      if (!c || counts[c->index] == 0) continue;
      if (hipMemcpyAsync(h_out + (size_t)c->index * row_len, d_out + (size_t)c->index * row_len, counts[c->index] * sizeof(c32),
                         hipMemcpyDeviceToHost, stream) != hipSuccess) return false;
    }
    if (hipStreamSynchronize(stream) != hipSuccess) { suamd_set_error("device failure in the channeliser"); return false; }
    std::memmove(h_in, h_in + nproc, (fill - nproc) * sizeof(c32));
    fill -= nproc;
    bool ok = true;
    for (auto &c : channels) {
      if (!c || counts[c->index] == 0 || !c->params.on_data) continue;
      // the pointer stays valid until the next feed (Tasks/LPFTask.cpp:32)
      if (!c->params.on_data(c.get(), c->params.privdata, reinterpret_cast<const SUCOMPLEX *>(h_out + (size_t)c->index * row_len),
                             counts[c->index])) ok = false;
    }
    return ok;
  }
};

extern "C" {

SUAMD_API sigutils_specttuner *su_specttuner_new(const struct sigutils_specttuner_params *params)
{
  if (!params) { suamd_set_error("null parameters"); return nullptr; }
  auto *t = new (std::nothrow) sigutils_specttuner();
  if (!t) return nullptr;
  const char *dev = std::getenv("SUAMD_DEVICE");
  t->ctx = suamd_ctx_new(dev ? std::atoi(dev) : 0);
  if (t->ctx) t->st = suamd_specttuner_new(t->ctx, (unsigned)params->window_size);
  bool ok = t->st != nullptr;
  if (ok) {
    t->W = (unsigned)params->window_size; t->H = t->W / 2;
    const size_t cap = (size_t)sigutils_specttuner::CAP_HALVES * t->H;
    ok = hipStreamCreateWithFlags(&t->stream, hipStreamNonBlocking) == hipSuccess &&
         hipHostMalloc((void **)&t->h_in, cap * sizeof(c32), hipHostMallocDefault) == hipSuccess &&
         hipMalloc((void **)&t->d_in, cap * sizeof(c32)) == hipSuccess;
    if (!ok) suamd_set_error("allocation failed");
  }
  if (!ok) {
    if (t->st) suamd_specttuner_destroy(t->st);
    if (t->ctx) suamd_ctx_destroy(t->ctx);
    if (t->h_in) (void)hipHostFree(t->h_in);
    if (t->d_in) (void)hipFree(t->d_in);
    if (t->stream) (void)hipStreamDestroy(t->stream);
    delete t;
    return nullptr;
  }
  return t;
}

SUAMD_API void su_specttuner_destroy(sigutils_specttuner *t)
{
  if (!t) return;
  if (t->stream) (void)hipStreamSynchronize(t->stream);
  if (t->st) suamd_specttuner_destroy(t->st);
  if (t->h_in) (void)hipHostFree(t->h_in);
  if (t->d_in) (void)hipFree(t->d_in);
  if (t->h_out) (void)hipHostFree(t->h_out);
  if (t->d_out) (void)hipFree(t->d_out);
  if (t->stream) (void)hipStreamDestroy(t->stream);
  if (t->ctx) suamd_ctx_destroy(t->ctx);
  delete t;
}

SUAMD_API sigutils_specttuner_channel *su_specttuner_open_channel(sigutils_specttuner *t, const struct sigutils_specttuner_channel_params *p)
{
  if (!t || !p) { suamd_set_error("null argument"); return nullptr; }
  const int idx = suamd_specttuner_open_channel(t->st, (double)p->f0, (double)p->bw, (double)p->guard, p->precise);
  if (idx < 0) return nullptr;
  std::unique_ptr<sigutils_specttuner_channel> c(new sigutils_specttuner_channel{*p, t, idx});
  if ((size_t)idx >= t->channels.size()) t->channels.resize((size_t)idx + 1);
  t->channels[idx] = std::move(c);
  return t->channels[idx].get();
}

SUAMD_API SUBOOL su_specttuner_close_channel(sigutils_specttuner *t, sigutils_specttuner_channel *c)
{
  if (!t || !c || c->owner != t) { suamd_set_error("no such channel"); return SU_FALSE; }
  const int idx = c->index;
  if (!suamd_specttuner_close_channel(t->st, idx)) return SU_FALSE;
  t->channels[idx].reset();
  return SU_TRUE;
}

SUAMD_API SUBOOL su_specttuner_feed_bulk(sigutils_specttuner *t, const SUCOMPLEX *buf, SUSCOUNT size)
{
  if (!t || (size && !buf)) { suamd_set_error("null argument"); return SU_FALSE; }
  const size_t cap = (size_t)sigutils_specttuner::CAP_HALVES * t->H;
  while (size > 0) {
    const size_t n = std::min<size_t>(size, cap - t->fill);
    std::memcpy(static_cast<void *>(t->h_in + t->fill), buf, n * sizeof(c32));
    t->fill += n; buf += n; size -= n;
    if (t->fill == cap && !t->flush()) return SU_FALSE;
  }
  return t->flush() ? SU_TRUE : SU_FALSE;
}

SUAMD_API SUFLOAT su_specttuner_channel_get_decimation(const sigutils_specttuner_channel *c)
{
  return c ? (SUFLOAT)suamd_specttuner_channel_decimation(c->owner->st, c->index) : 0;
}
SUAMD_API SUFLOAT su_specttuner_channel_get_bw(const sigutils_specttuner_channel *c) { return c ? c->params.bw : 0; }
SUAMD_API SUFLOAT su_specttuner_channel_get_f0(const sigutils_specttuner_channel *c) { return c ? c->params.f0 : 0; }
SUAMD_API unsigned su_specttuner_channel_get_size(const sigutils_specttuner_channel *c)
{
  return c ? suamd_specttuner_channel_size(c->owner->st, c->index) : 0;
}

}  // extern "C"
