// capi_gangs.hip -- the C ABI of the gangs: many 1-channel banks (the live analyzer's inspectors) side by side in one launch per
// loop type -- on rows of their own (gather -> packed slab -> scatter) and on rows that are columns of the caller's time-major
// slab already (suamd_*_gang_*_slab).  include/sigdigger_amd.h documents every entry; kernels: gangs.hip.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstring>
#include <algorithm>
#include <map>
#include <vector>

#include "capi_internal.hpp"
#include "tuning.hpp"

extern "C" {

// ---- gangs: many 1-channel banks, each with its own parameters, side by side in one launch -----------
SUBOOL suamd_costas_gang_feed(suamd_ctx_t *ctx, suamd_costas_bank_t *const *banks, unsigned n, const suamd_complex *const *d_x,
                              suamd_complex *const *d_y, const SUSCOUNT *len, void *stream)
{
  if (!ctx || (n && (!banks || !d_x || !d_y || !len))) { set_err("null argument"); return SU_FALSE; }
  hipStream_t st = as_stream(stream);
  // items sorted by loop type (kind, arm-filter order): the type is compiled in, everything else is per lane
  std::map<int, std::vector<sdk::CostasGangItem>> types;
  for (unsigned i = 0; i < n; ++i) {
    if (!banks[i] || banks[i]->nchan != 1) { set_err("gang members must be 1-channel banks"); return SU_FALSE; }
    if (len[i] == 0) continue;
    if (!d_x[i] || !d_y[i]) { set_err("null row"); return SU_FALSE; }
    types[banks[i]->p.kind * 8 + banks[i]->p.order].push_back(sdk::CostasGangItem{banks[i]->p, banks[i]->s, d_x[i], d_y[i], (long long)len[i]});
  }
  // one workgroup per <= 64 items of one type; all of them in one launch (<= 448 items per descriptor slot)
  static_assert(448 * sizeof(sdk::CostasGangItem) <= suamd_ctx::GANG_SLOT_BYTES, "a descriptor slot holds 448 Costas items");
  std::vector<sdk::CostasGangItem> items;
  std::vector<sdk::GangGroup> groups;
  auto flush = [&]() -> SUBOOL {
    if (items.empty()) return SU_TRUE;
    const SUBOOL ok = gang_tm(ctx, items, &groups, 8, offsetof(sdk::CostasGangItem, x), (long long)offsetof(sdk::CostasGangItem, y),
                              offsetof(sdk::CostasGangItem, len), st,
                              [&](sdk::CostasGangItem *d, sdk::GangGroup *dg, void *tm, long long slab) {
                                return sdk::costas_gang(d, dg, (int)groups.size(), tm, slab, st);
                              });
    items.clear(); groups.clear();
    return ok;
  };
  for (auto &kv : types) {
    for (size_t o = 0; o < kv.second.size(); o += 64) {
      const size_t cnt = std::min<size_t>(64, kv.second.size() - o);
      if (items.size() + cnt > 448 && !flush()) return SU_FALSE;
      bool unit = true;                                       // x * 1.0f is exact: skipping the multiply keeps the bits
      for (size_t q = 0; q < cnt; ++q) unit = unit && kv.second[o + q].p.gain == 1.0f;
      groups.push_back(sdk::GangGroup{(int)items.size(), (int)cnt, kv.first / 8, kv.first % 8, unit ? 1 : 0});
      items.insert(items.end(), kv.second.begin() + o, kv.second.begin() + o + cnt);
    }
  }
  return flush();
}

SUBOOL suamd_pll_gang_feed(suamd_ctx_t *ctx, suamd_pll_bank_t *const *banks, unsigned n, const suamd_complex *const *d_x,
                           suamd_complex *const *d_y, const SUSCOUNT *len, void *stream)
{
  if (!ctx || (n && (!banks || !d_x || !d_y || !len))) { set_err("null argument"); return SU_FALSE; }
  hipStream_t st = as_stream(stream);
  std::vector<sdk::PllGangItem> items;
  for (unsigned i = 0; i < n; ++i) {
    if (!banks[i] || banks[i]->nchan != 1) { set_err("gang members must be 1-channel banks"); return SU_FALSE; }
    if (len[i] == 0) continue;
    if (!d_x[i] || !d_y[i]) { set_err("null row"); return SU_FALSE; }
    items.push_back(sdk::PllGangItem{banks[i]->alpha, banks[i]->beta, banks[i]->s, d_x[i], d_y[i], (long long)len[i]});
  }
  for (size_t o = 0; o < items.size(); o += 512) {
    std::vector<sdk::PllGangItem> part(items.begin() + o, items.begin() + std::min(items.size(), o + 512));
    if (!gang_tm(ctx, part, nullptr, 8, offsetof(sdk::PllGangItem, x), (long long)offsetof(sdk::PllGangItem, y), offsetof(sdk::PllGangItem, len), st,
                 [&](sdk::PllGangItem *d, sdk::GangGroup *, void *tm, long long slab) { return sdk::pll_gang(d, (int)part.size(), tm, slab, st); }))
      return SU_FALSE;
  }
  return SU_TRUE;
}

SUBOOL suamd_cma_gang_feed(suamd_ctx_t *ctx, suamd_cma_bank_t *const *banks, unsigned n, const suamd_complex *const *d_x,
                           const uint32_t *const *d_count, const SUSCOUNT *fixed_len, suamd_complex *const *d_y, void *stream)
{
  if (!ctx || (n && (!banks || !d_x || !d_y))) { set_err("null argument"); return SU_FALSE; }
  hipStream_t st = as_stream(stream);
  std::map<unsigned, std::vector<sdk::CmaGangItem>> groups;                    // one launch per equalizer length
  for (unsigned i = 0; i < n; ++i) {
    suamd_cma_bank *b = banks[i];
    if (!b || b->nchan != 1) { set_err("gang members must be 1-channel banks"); return SU_FALSE; }
    if (!d_x[i] || !d_y[i]) { set_err("null row"); return SU_FALSE; }
    const uint32_t *cnt = d_count ? d_count[i] : nullptr;
    if (!cnt && !fixed_len) { set_err("neither counts nor lengths"); return SU_FALSE; }
    groups[b->n].push_back(sdk::CmaGangItem{b->mu, b->locked, b->d_w, b->d_dl, d_x[i], d_y[i], cnt, fixed_len ? (long long)fixed_len[i] : 0});
  }
  for (auto &kv : groups) {
    for (size_t o = 0; o < kv.second.size(); o += 512) {
      std::vector<sdk::CmaGangItem> part(kv.second.begin() + o, kv.second.begin() + std::min(kv.second.size(), o + 512));
      sdk::CmaGangItem *d = gang_upload(ctx, part, st);
      if (!d) return SU_FALSE;
      HIP_TRY(sdk::cma_gang(d, (int)part.size(), (int)kv.first, st), SU_FALSE);
    }
  }
  return SU_TRUE;
}

// per group of 64 items (one wavefront): the schedule.  Round by round (loops.hip clock_ring) when there are enough lanes for
// their crossings to spread over the samples and every item's half cycle is short enough for one round's advance steps;
// sdk::tuning().clock_mode 0 / 1 keeps the crossing-by-crossing form (A / B)
static void clock_gang_schedule(std::vector<sdk::ClockGangItem> &items)
{
  for (size_t g0 = 0; g0 < items.size(); g0 += 64) {
    const size_t g1 = std::min(items.size(), g0 + 64);
    int steps = 0;
    bool same = true;
    for (size_t q = g0; q < g1; ++q) {
      steps = std::max(steps, (int)std::ceil(0.5f / (2.0f * items[q].p.bmin)) + 1);
      same = same && std::memcmp(&items[q].p, &items[g0].p, sizeof(sdk::ClockParams)) == 0;
    }
    const long long forced = sdk::tuning().clock_mode;
    const bool ring = forced >= 0 ? forced == 2 : (g1 - g0 >= 3 && steps <= 25);
    steps = std::min(30, 3 * ((steps + 2) / 3));
    for (size_t q = g0; q < g1; ++q) { items[q].steps = ring ? steps : 0; items[q].uniform = same ? 1 : 0; }
  }
}

SUBOOL suamd_clock_gang_feed(suamd_ctx_t *ctx, suamd_clock_bank_t *const *banks, unsigned n, const suamd_complex *const *d_x,
                             const SUSCOUNT *len, suamd_complex *const *d_sym, uint32_t *const *d_count, void *stream)
{
  if (!ctx || (n && (!banks || !d_x || !len || !d_sym || !d_count))) { set_err("null argument"); return SU_FALSE; }
  hipStream_t st = as_stream(stream);
  std::vector<sdk::ClockGangItem> items;
  for (unsigned i = 0; i < n; ++i) {
    if (!banks[i] || banks[i]->nchan != 1) { set_err("gang members must be 1-channel banks"); return SU_FALSE; }
    if (len[i] == 0) continue;
    if (!d_x[i] || !d_sym[i] || !d_count[i]) { set_err("null row"); return SU_FALSE; }
    items.push_back(sdk::ClockGangItem{banks[i]->p, banks[i]->s, d_x[i], (long long)len[i], d_sym[i], d_count[i], 0, 0});
  }
  clock_gang_schedule(items);
  for (size_t o = 0; o < items.size(); o += 512) {
    std::vector<sdk::ClockGangItem> part(items.begin() + o, items.begin() + std::min(items.size(), o + 512));
    if (!gang_tm(ctx, part, nullptr, 8, offsetof(sdk::ClockGangItem, x), -1, offsetof(sdk::ClockGangItem, len), st,
                 [&](sdk::ClockGangItem *d, sdk::GangGroup *, void *tm, long long slab) { return sdk::clock_gang(d, (int)part.size(), tm, slab, st); }))
      return SU_FALSE;
  }
  return SU_TRUE;
}

// The AGC of a gang in its four steps, so that a caller can pipeline sub-ranges of a block through the
// level trackers and the stages behind them: pre (|x|^2 in dB and its sliding maximum, whole block),
// level (recurrence, any sub-range in order), apply (gain on the delayed input, same sub-range),
// finish (history / delay-line carry, whole block, after the last apply).
SUBOOL suamd_agc_gang_pre(suamd_ctx_t *ctx, suamd_agc_bank_t *const *banks, unsigned n, const suamd_complex *const *d_x,
                          const SUSCOUNT *len, void *stream)
{
  if (!ctx || (n && (!banks || !d_x || !len))) { set_err("null argument"); return SU_FALSE; }
  hipStream_t st = as_stream(stream);
  std::vector<sdk::AgcPreItem> items;
  long long span = 0;
  for (unsigned i = 0; i < n; ++i) {
    suamd_agc_bank *b = banks[i];
    if (!b || b->nchan != 1) { set_err("gang members must be 1-channel banks"); return SU_FALSE; }
    if (len[i] == 0) continue;
    if (!d_x[i]) { set_err("null row"); return SU_FALSE; }
    if (!b->scratch.reserve(2 * sizeof(float) * (size_t)len[i])) { set_err("scratch allocation failed"); return SU_FALSE; }
    float *db = static_cast<float *>(b->scratch.p);
    items.push_back(sdk::AgcPreItem{d_x[i], b->s.mag_history, db, db + len[i], (long long)len[i], (int)b->p.mag_history_size});
    span = std::max(span, (long long)len[i]);
  }
  for (size_t o = 0; o < items.size(); o += 512) {
    std::vector<sdk::AgcPreItem> part(items.begin() + o, items.begin() + std::min(items.size(), o + 512));
    sdk::AgcPreItem *d = gang_upload(ctx, part, st);
    if (!d) return SU_FALSE;
    HIP_TRY(sdk::agc_pre_items(d, (int)part.size(), span, st), SU_FALSE);
  }
  return SU_TRUE;
}

SUBOOL suamd_agc_gang_level(suamd_ctx_t *ctx, suamd_agc_bank_t *const *banks, unsigned n, const SUSCOUNT *len,
                            const SUSCOUNT *m0, const SUSCOUNT *m1, void *stream)
{
  if (!ctx || (n && (!banks || !len || !m0 || !m1))) { set_err("null argument"); return SU_FALSE; }
  hipStream_t st = as_stream(stream);
  std::vector<sdk::AgcGangItem> items;
  for (unsigned i = 0; i < n; ++i) {
    suamd_agc_bank *b = banks[i];
    if (!b || m1[i] > len[i] || m0[i] > m1[i]) { set_err("bad sub-range"); return SU_FALSE; }
    if (m1[i] == m0[i]) continue;
    items.push_back(sdk::AgcGangItem{b->p, b->s, static_cast<float *>(b->scratch.p) + len[i] + m0[i], (long long)(m1[i] - m0[i])});
  }
  for (size_t o = 0; o < items.size(); o += 512) {
    std::vector<sdk::AgcGangItem> part(items.begin() + o, items.begin() + std::min(items.size(), o + 512));
    if (!gang_tm(ctx, part, nullptr, 4, offsetof(sdk::AgcGangItem, peak), (long long)offsetof(sdk::AgcGangItem, peak), offsetof(sdk::AgcGangItem, len), st,
                 [&](sdk::AgcGangItem *d, sdk::GangGroup *, void *tm, long long slab) { return sdk::agc_level_gang(d, (int)part.size(), tm, slab, st); }))
      return SU_FALSE;
  }
  return SU_TRUE;
}

SUBOOL suamd_agc_gang_apply(suamd_ctx_t *ctx, suamd_agc_bank_t *const *banks, unsigned n, const suamd_complex *const *d_x,
                            suamd_complex *const *d_y, const SUSCOUNT *len, const SUSCOUNT *m0, const SUSCOUNT *m1, void *stream)
{
  if (!ctx || (n && (!banks || !d_x || !d_y || !len || !m0 || !m1))) { set_err("null argument"); return SU_FALSE; }
  hipStream_t st = as_stream(stream);
  std::vector<sdk::AgcApplyItem> items;
  long long span = 0;
  for (unsigned i = 0; i < n; ++i) {
    suamd_agc_bank *b = banks[i];
    if (!b || m1[i] > len[i] || m0[i] > m1[i]) { set_err("bad sub-range"); return SU_FALSE; }
    if (m1[i] == m0[i]) continue;
    if (!d_x[i] || !d_y[i] || d_x[i] == d_y[i]) { set_err("null or aliased row"); return SU_FALSE; }
    items.push_back(sdk::AgcApplyItem{b->p, b->s.delay_line, d_x[i], d_y[i], static_cast<const float *>(b->scratch.p) + len[i],
                                      (long long)m0[i], (long long)m1[i]});
    span = std::max(span, (long long)(m1[i] - m0[i]));
  }
  for (size_t o = 0; o < items.size(); o += 512) {
    std::vector<sdk::AgcApplyItem> part(items.begin() + o, items.begin() + std::min(items.size(), o + 512));
    sdk::AgcApplyItem *d = gang_upload(ctx, part, st);
    if (!d) return SU_FALSE;
    HIP_TRY(sdk::agc_apply_items(d, (int)part.size(), span, st), SU_FALSE);
  }
  return SU_TRUE;
}

SUBOOL suamd_rows_deliver_strided(suamd_ctx_t *ctx, unsigned n, const suamd_complex *const *d_src, const SUSCOUNT *src_stride,
                                  uint32_t *const *d_count, const SUSCOUNT *fixed_len, suamd_complex *const *dst,
                                  uint32_t *const *count_out, void *stream)
{
  if (!ctx || (n && (!d_src || !dst || !count_out || (!d_count && !fixed_len)))) { set_err("null argument"); return SU_FALSE; }
  hipStream_t st = as_stream(stream);
  std::vector<sdk::DeliverItem> items;
  for (unsigned i = 0; i < n; ++i) {
    uint32_t *cnt = d_count ? d_count[i] : nullptr;
    if (!d_src[i] || !dst[i] || !count_out[i] || (!cnt && !fixed_len)) { set_err("null row"); return SU_FALSE; }
    if (!cnt && fixed_len[i] > 0xffffffffull) { set_err("row too long"); return SU_FALSE; }
    const SUSCOUNT stride = src_stride ? src_stride[i] : 1;
    if (stride < 1 || stride > 0xffffffffull) { set_err("bad row stride"); return SU_FALSE; }
    items.push_back(sdk::DeliverItem{d_src[i], dst[i], cnt, cnt ? 0u : (unsigned)fixed_len[i], count_out[i], (unsigned)stride});
  }
  for (size_t o = 0; o < items.size(); o += 512) {
    std::vector<sdk::DeliverItem> part(items.begin() + o, items.begin() + std::min(items.size(), o + 512));
    sdk::DeliverItem *d = gang_upload(ctx, part, st);
    if (!d) return SU_FALSE;
    HIP_TRY(sdk::rows_deliver(d, (int)part.size(), st), SU_FALSE);
  }
  return SU_TRUE;
}

SUBOOL suamd_rows_deliver(suamd_ctx_t *ctx, unsigned n, const suamd_complex *const *d_src, uint32_t *const *d_count,
                          const SUSCOUNT *fixed_len, suamd_complex *const *dst, uint32_t *const *count_out, void *stream)
{
  return suamd_rows_deliver_strided(ctx, n, d_src, nullptr, d_count, fixed_len, dst, count_out, stream);
}

SUBOOL suamd_agc_gang_finish(suamd_ctx_t *ctx, suamd_agc_bank_t *const *banks, unsigned n, const suamd_complex *const *d_x,
                             const SUSCOUNT *len, void *stream)
{
  if (!ctx || (n && (!banks || !d_x || !len))) { set_err("null argument"); return SU_FALSE; }
  hipStream_t st = as_stream(stream);
  std::vector<sdk::AgcStateItem> items;
  for (unsigned i = 0; i < n; ++i) {
    suamd_agc_bank *b = banks[i];
    if (!b || len[i] == 0) continue;
    if (!d_x[i] || !b->scratch.p) { set_err("null row"); return SU_FALSE; }
    items.push_back(sdk::AgcStateItem{b->s.delay_line, b->s.mag_history, d_x[i], static_cast<const float *>(b->scratch.p), (long long)len[i],
                                      (int)b->p.delay_line_size, (int)b->p.mag_history_size, 1, 1});
    b->n_fed += len[i];
  }
  for (size_t o = 0; o < items.size(); o += 512) {
    std::vector<sdk::AgcStateItem> part(items.begin() + o, items.begin() + std::min(items.size(), o + 512));
    sdk::AgcStateItem *d = gang_upload(ctx, part, st);
    if (!d) return SU_FALSE;
    HIP_TRY(sdk::agc_state_items(d, (int)part.size(), st), SU_FALSE);
  }
  return SU_TRUE;
}

SUBOOL suamd_agc_gang_feed(suamd_ctx_t *ctx, suamd_agc_bank_t *const *banks, unsigned n, const suamd_complex *const *d_x,
                           suamd_complex *const *d_y, const SUSCOUNT *len, void *stream)
{
  if (!ctx || (n && (!banks || !d_x || !d_y || !len))) { set_err("null argument"); return SU_FALSE; }
  for (unsigned i = 0; i < n; ++i)
    if (len[i] && (!d_x[i] || !d_y[i] || d_x[i] == d_y[i])) { set_err("null or aliased row"); return SU_FALSE; }
  const std::vector<SUSCOUNT> zero(n, 0);                    // the whole block as one sub-range: five launches in all
  return suamd_agc_gang_pre(ctx, banks, n, d_x, len, stream) && suamd_agc_gang_level(ctx, banks, n, len, zero.data(), len, stream) &&
         suamd_agc_gang_apply(ctx, banks, n, d_x, d_y, len, zero.data(), len, stream) &&
         suamd_agc_gang_finish(ctx, banks, n, d_x, len, stream) ? SU_TRUE : SU_FALSE;
}

// ---- gangs on time-major slabs (kernels.hpp GangSlab) ------------------------------------------------------------
// The items' rows are columns of one slab per side already: nothing is gathered or scattered, the recurrence wavefronts
// stream the slab where it lies.  The base of a side is the lowest row start of the call; every other one must lie less
// than 4 GiB (less the chunk look-ahead) above it.
extern "C++" {
namespace {
template <typename Item>
bool slab_bases(const std::vector<Item> &items, const void *Item::*x, void *Item::*y, long long px, long long py, int elem, sdk::GangSlab *io)
{
  uintptr_t lx = ~(uintptr_t)0, hx = 0, ly = ~(uintptr_t)0, hy = 0;
  for (const Item &it : items) {
    const uintptr_t a = reinterpret_cast<uintptr_t>(it.*x);
    lx = std::min(lx, a); hx = std::max(hx, a);
    if (y) { const uintptr_t b = reinterpret_cast<uintptr_t>(it.*y); ly = std::min(ly, b); hy = std::max(hy, b); }
  }
  const unsigned long long lim = (1ull << 32) - 1;
  if ((unsigned long long)(hx - lx) + 64ull * (unsigned long long)px * elem > lim) return false;
  if (y && (unsigned long long)(hy - ly) + 64ull * (unsigned long long)py * elem > lim) return false;
  io->in = reinterpret_cast<const void *>(lx);
  io->out = y ? reinterpret_cast<void *>(ly) : nullptr;
  io->pitch_in = px; io->pitch_out = py;
  return true;
}
}  // namespace
}  // extern "C++"

SUBOOL suamd_costas_gang_feed_slab(suamd_ctx_t *ctx, suamd_costas_bank_t *const *banks, unsigned n, const suamd_complex *const *d_x,
                                   SUSCOUNT x_pitch, suamd_complex *const *d_y, SUSCOUNT y_pitch, const SUSCOUNT *len, void *stream)
{
  if (!ctx || (n && (!banks || !d_x || !d_y || !len))) { set_err("null argument"); return SU_FALSE; }
  if (x_pitch < 1 || y_pitch < 1 || x_pitch > (1u << 20) || y_pitch > (1u << 20)) { set_err("bad slab pitch"); return SU_FALSE; }
  hipStream_t st = as_stream(stream);
  std::map<int, std::vector<sdk::CostasGangItem>> types;
  for (unsigned i = 0; i < n; ++i) {
    if (!banks[i] || banks[i]->nchan != 1) { set_err("gang members must be 1-channel banks"); return SU_FALSE; }
    if (len[i] == 0) continue;
    if (!d_x[i] || !d_y[i]) { set_err("null row"); return SU_FALSE; }
    types[banks[i]->p.kind * 8 + banks[i]->p.order].push_back(sdk::CostasGangItem{banks[i]->p, banks[i]->s, d_x[i], d_y[i], (long long)len[i]});
  }
  std::vector<sdk::CostasGangItem> items;
  std::vector<sdk::GangGroup> groups;
  auto flush = [&]() -> SUBOOL {
    if (items.empty()) return SU_TRUE;
    sdk::GangSlab io{};
    if (!slab_bases(items, &sdk::CostasGangItem::x, &sdk::CostasGangItem::y, (long long)x_pitch, (long long)y_pitch, 8, &io)) { set_err("rows further than 4 GiB apart"); return SU_FALSE; }
    sdk::CostasGangItem *d = gang_upload(ctx, items, st);
    sdk::GangGroup *dg = d ? gang_upload(ctx, groups, st) : nullptr;
    if (!d || !dg) return SU_FALSE;
    HIP_TRY(sdk::costas_gang_slab(d, dg, (int)groups.size(), io, st), SU_FALSE);
    items.clear(); groups.clear();
    return SU_TRUE;
  };
  for (auto &kv : types) {
    for (size_t o = 0; o < kv.second.size(); o += 64) {
      const size_t cnt = std::min<size_t>(64, kv.second.size() - o);
      if (items.size() + cnt > 448 && !flush()) return SU_FALSE;
      bool unit = true;
      for (size_t q = 0; q < cnt; ++q) unit = unit && kv.second[o + q].p.gain == 1.0f;
      groups.push_back(sdk::GangGroup{(int)items.size(), (int)cnt, kv.first / 8, kv.first % 8, unit ? 1 : 0});
      items.insert(items.end(), kv.second.begin() + o, kv.second.begin() + o + cnt);
    }
  }
  return flush();
}

SUBOOL suamd_pll_gang_feed_slab(suamd_ctx_t *ctx, suamd_pll_bank_t *const *banks, unsigned n, const suamd_complex *const *d_x,
                                SUSCOUNT x_pitch, suamd_complex *const *d_y, SUSCOUNT y_pitch, const SUSCOUNT *len, void *stream)
{
  if (!ctx || (n && (!banks || !d_x || !d_y || !len))) { set_err("null argument"); return SU_FALSE; }
  if (x_pitch < 1 || y_pitch < 1 || x_pitch > (1u << 20) || y_pitch > (1u << 20)) { set_err("bad slab pitch"); return SU_FALSE; }
  hipStream_t st = as_stream(stream);
  std::vector<sdk::PllGangItem> items;
  for (unsigned i = 0; i < n; ++i) {
    if (!banks[i] || banks[i]->nchan != 1) { set_err("gang members must be 1-channel banks"); return SU_FALSE; }
    if (len[i] == 0) continue;
    if (!d_x[i] || !d_y[i]) { set_err("null row"); return SU_FALSE; }
    items.push_back(sdk::PllGangItem{banks[i]->alpha, banks[i]->beta, banks[i]->s, d_x[i], d_y[i], (long long)len[i]});
  }
  for (size_t o = 0; o < items.size(); o += 512) {
    std::vector<sdk::PllGangItem> part(items.begin() + o, items.begin() + std::min(items.size(), o + 512));
    sdk::GangSlab io{};
    if (!slab_bases(part, &sdk::PllGangItem::x, &sdk::PllGangItem::y, (long long)x_pitch, (long long)y_pitch, 8, &io)) { set_err("rows further than 4 GiB apart"); return SU_FALSE; }
    sdk::PllGangItem *d = gang_upload(ctx, part, st);
    if (!d) return SU_FALSE;
    HIP_TRY(sdk::pll_gang_slab(d, (int)part.size(), io, st), SU_FALSE);
  }
  return SU_TRUE;
}

SUBOOL suamd_clock_gang_feed_slab(suamd_ctx_t *ctx, suamd_clock_bank_t *const *banks, unsigned n, const suamd_complex *const *d_x,
                                  SUSCOUNT x_pitch, const SUSCOUNT *len, suamd_complex *const *d_sym, uint32_t *const *d_count, void *stream)
{
  if (!ctx || (n && (!banks || !d_x || !len || !d_sym || !d_count))) { set_err("null argument"); return SU_FALSE; }
  if (x_pitch < 1 || x_pitch > (1u << 20)) { set_err("bad slab pitch"); return SU_FALSE; }
  hipStream_t st = as_stream(stream);
  std::vector<sdk::ClockGangItem> items;
  for (unsigned i = 0; i < n; ++i) {
    if (!banks[i] || banks[i]->nchan != 1) { set_err("gang members must be 1-channel banks"); return SU_FALSE; }
    if (len[i] == 0) continue;
    if (!d_x[i] || !d_sym[i] || !d_count[i]) { set_err("null row"); return SU_FALSE; }
    items.push_back(sdk::ClockGangItem{banks[i]->p, banks[i]->s, d_x[i], (long long)len[i], d_sym[i], d_count[i], 0, 0});
  }
  clock_gang_schedule(items);
  for (size_t o = 0; o < items.size(); o += 512) {
    std::vector<sdk::ClockGangItem> part(items.begin() + o, items.begin() + std::min(items.size(), o + 512));
    sdk::GangSlab io{};
    if (!slab_bases<sdk::ClockGangItem>(part, &sdk::ClockGangItem::x, nullptr, (long long)x_pitch, 0, 8, &io)) { set_err("rows further than 4 GiB apart"); return SU_FALSE; }
    sdk::ClockGangItem *d = gang_upload(ctx, part, st);
    if (!d) return SU_FALSE;
    HIP_TRY(sdk::clock_gang_slab(d, (int)part.size(), io, st), SU_FALSE);
  }
  return SU_TRUE;
}

// The AGC's four steps on slabs.  Item i is column d_x[i] - d_x_slab of the input slab and of the two work slabs
// (d_work: 2 * work_rows * x_pitch floats -- magnitudes in dB, then their sliding maxima which become the levels).
namespace {
bool agc_slab_items(suamd_agc_bank_t *const *banks, unsigned n, const suamd_complex *d_x_slab, SUSCOUNT x_pitch, const suamd_complex *const *d_x,
                    const suamd_complex *d_y_slab, SUSCOUNT y_pitch, suamd_complex *const *d_y, const SUSCOUNT *len, const SUSCOUNT *m0,
                    const SUSCOUNT *m1, SUSCOUNT work_rows, bool skip_empty_range, std::vector<sdk::AgcSlabItem> *items, std::vector<unsigned> *which)
{
  for (unsigned i = 0; i < n; ++i) {
    suamd_agc_bank *b = banks[i];
    if (!b || b->nchan != 1) { set_err("gang members must be 1-channel banks"); return false; }
    if (len[i] == 0) continue;
    if (len[i] > work_rows) { set_err("work slabs shorter than the row"); return false; }
    if (!d_x[i] || d_x[i] < d_x_slab || (SUSCOUNT)(d_x[i] - d_x_slab) >= x_pitch) { set_err("row is not a column of the slab"); return false; }
    long long ly = 0;
    if (d_y) {
      if (!d_y[i] || d_y[i] < d_y_slab || (SUSCOUNT)(d_y[i] - d_y_slab) >= y_pitch) { set_err("row is not a column of the slab"); return false; }
      ly = d_y[i] - d_y_slab;
    }
    long long a = 0, e = (long long)len[i];
    if (m0) {
      if (m1[i] > len[i] || m0[i] > m1[i]) { set_err("bad sub-range"); return false; }
      if (skip_empty_range && m1[i] == m0[i]) continue;
      a = (long long)m0[i]; e = (long long)m1[i];
    }
    if (b->p.mag_history_size < 1 || b->p.mag_history_size > 64 || b->p.delay_line_size > 64) { set_err("history beyond the slab kernels' tiles"); return false; }
    items->push_back(sdk::AgcSlabItem{b->p, b->s, (int)(d_x[i] - d_x_slab), (int)ly, (long long)len[i], a, e});
    if (which) which->push_back(i);
  }
  return true;
}
}  // namespace

SUBOOL suamd_agc_gang_pre_slab(suamd_ctx_t *ctx, suamd_agc_bank_t *const *banks, unsigned n, const suamd_complex *d_x_slab, SUSCOUNT x_pitch,
                               const suamd_complex *const *d_x, const SUSCOUNT *len, float *d_work, SUSCOUNT work_rows, void *stream)
{
  if (!ctx || (n && (!banks || !d_x_slab || !d_x || !len || !d_work))) { set_err("null argument"); return SU_FALSE; }
  hipStream_t st = as_stream(stream);
  std::vector<sdk::AgcSlabItem> items;
  if (!agc_slab_items(banks, n, d_x_slab, x_pitch, d_x, nullptr, 0, nullptr, len, nullptr, nullptr, work_rows, false, &items, nullptr)) return SU_FALSE;
  for (size_t o = 0; o < items.size(); o += 512) {
    std::vector<sdk::AgcSlabItem> part(items.begin() + o, items.begin() + std::min(items.size(), o + 512));
    long long span = 0;
    int halo = 0;
    for (const sdk::AgcSlabItem &it : part) { span = std::max(span, it.len); halo = std::max(halo, (int)it.p.mag_history_size - 1); }
    sdk::AgcSlabItem *d = gang_upload(ctx, part, st);
    if (!d) return SU_FALSE;
    HIP_TRY(sdk::agc_pre_slab(d, (int)part.size(), d_x_slab, (long long)x_pitch, d_work, d_work + (size_t)work_rows * x_pitch, (long long)x_pitch, span, halo, st), SU_FALSE);
  }
  return SU_TRUE;
}

SUBOOL suamd_agc_gang_level_slab(suamd_ctx_t *ctx, suamd_agc_bank_t *const *banks, unsigned n, const suamd_complex *d_x_slab, SUSCOUNT x_pitch,
                                 const suamd_complex *const *d_x, const SUSCOUNT *len, const SUSCOUNT *m0, const SUSCOUNT *m1, float *d_work,
                                 SUSCOUNT work_rows, void *stream)
{
  if (!ctx || (n && (!banks || !d_x_slab || !d_x || !len || !m0 || !m1 || !d_work))) { set_err("null argument"); return SU_FALSE; }
  hipStream_t st = as_stream(stream);
  std::vector<sdk::AgcSlabItem> sl;
  if (!agc_slab_items(banks, n, d_x_slab, x_pitch, d_x, nullptr, 0, nullptr, len, m0, m1, work_rows, true, &sl, nullptr)) return SU_FALSE;
  float *peak = d_work + (size_t)work_rows * x_pitch;
  std::vector<sdk::AgcGangItem> items;
  for (const sdk::AgcSlabItem &it : sl) items.push_back(sdk::AgcGangItem{it.p, it.s, peak + (size_t)it.m0 * x_pitch + it.lane, it.m1 - it.m0});
  for (size_t o = 0; o < items.size(); o += 512) {
    std::vector<sdk::AgcGangItem> part(items.begin() + o, items.begin() + std::min(items.size(), o + 512));
    sdk::GangSlab io{};
    uintptr_t lo = ~(uintptr_t)0, hi = 0;
    for (const sdk::AgcGangItem &it : part) { const uintptr_t a = reinterpret_cast<uintptr_t>(it.peak); lo = std::min(lo, a); hi = std::max(hi, a); }
    if ((unsigned long long)(hi - lo) + 64ull * x_pitch * 4 > (1ull << 32) - 1) { set_err("rows further than 4 GiB apart"); return SU_FALSE; }
    io.in = reinterpret_cast<const void *>(lo); io.out = reinterpret_cast<void *>(lo); io.pitch_in = io.pitch_out = (long long)x_pitch;
    sdk::AgcGangItem *d = gang_upload(ctx, part, st);
    if (!d) return SU_FALSE;
    HIP_TRY(sdk::agc_level_gang_slab(d, (int)part.size(), io, st), SU_FALSE);
  }
  return SU_TRUE;
}

SUBOOL suamd_agc_gang_apply_slab(suamd_ctx_t *ctx, suamd_agc_bank_t *const *banks, unsigned n, const suamd_complex *d_x_slab, SUSCOUNT x_pitch,
                                 const suamd_complex *const *d_x, suamd_complex *d_y_slab, SUSCOUNT y_pitch, suamd_complex *const *d_y,
                                 const SUSCOUNT *len, const SUSCOUNT *m0, const SUSCOUNT *m1, float *d_work, SUSCOUNT work_rows, void *stream)
{
  if (!ctx || (n && (!banks || !d_x_slab || !d_x || !d_y_slab || !d_y || !len || !m0 || !m1 || !d_work))) { set_err("null argument"); return SU_FALSE; }
  if ((const void *)d_x_slab == (const void *)d_y_slab) { set_err("aliased slabs"); return SU_FALSE; }
  hipStream_t st = as_stream(stream);
  std::vector<sdk::AgcSlabItem> items;
  if (!agc_slab_items(banks, n, d_x_slab, x_pitch, d_x, d_y_slab, y_pitch, d_y, len, m0, m1, work_rows, true, &items, nullptr)) return SU_FALSE;
  for (size_t o = 0; o < items.size(); o += 512) {
    std::vector<sdk::AgcSlabItem> part(items.begin() + o, items.begin() + std::min(items.size(), o + 512));
    long long mlo = part[0].m0, mhi = part[0].m1;
    for (const sdk::AgcSlabItem &it : part) { mlo = std::min(mlo, it.m0); mhi = std::max(mhi, it.m1); }
    sdk::AgcSlabItem *d = gang_upload(ctx, part, st);
    if (!d) return SU_FALSE;
    HIP_TRY(sdk::agc_apply_slab(d, (int)part.size(), d_x_slab, (long long)x_pitch, d_y_slab, (long long)y_pitch,
                                d_work + (size_t)work_rows * x_pitch, (long long)x_pitch, mlo, mhi, st), SU_FALSE);
  }
  return SU_TRUE;
}

SUBOOL suamd_agc_gang_finish_slab(suamd_ctx_t *ctx, suamd_agc_bank_t *const *banks, unsigned n, const suamd_complex *d_x_slab, SUSCOUNT x_pitch,
                                  const suamd_complex *const *d_x, const SUSCOUNT *len, float *d_work, SUSCOUNT work_rows, void *stream)
{
  if (!ctx || (n && (!banks || !d_x_slab || !d_x || !len || !d_work))) { set_err("null argument"); return SU_FALSE; }
  hipStream_t st = as_stream(stream);
  std::vector<sdk::AgcSlabItem> sl;
  std::vector<unsigned> which;
  if (!agc_slab_items(banks, n, d_x_slab, x_pitch, d_x, nullptr, 0, nullptr, len, nullptr, nullptr, work_rows, false, &sl, &which)) return SU_FALSE;
  std::vector<sdk::AgcStateItem> items;
  for (size_t q = 0; q < sl.size(); ++q) {
    const sdk::AgcSlabItem &it = sl[q];
    items.push_back(sdk::AgcStateItem{it.s.delay_line, it.s.mag_history, d_x_slab + it.lane, d_work + it.lane, it.len,
                                      (int)it.p.delay_line_size, (int)it.p.mag_history_size, (long long)x_pitch, (long long)x_pitch});
    banks[which[q]]->n_fed += (uint64_t)it.len;
  }
  for (size_t o = 0; o < items.size(); o += 512) {
    std::vector<sdk::AgcStateItem> part(items.begin() + o, items.begin() + std::min(items.size(), o + 512));
    sdk::AgcStateItem *d = gang_upload(ctx, part, st);
    if (!d) return SU_FALSE;
    HIP_TRY(sdk::agc_state_items(d, (int)part.size(), st), SU_FALSE);
  }
  return SU_TRUE;
}

}  // extern "C"
