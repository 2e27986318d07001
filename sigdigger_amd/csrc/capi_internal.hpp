// capi_internal.hpp -- what the translation units of the C-ABI layer share (capi.hip: contexts, plans, banks, element-wise
// calls; capi_gangs.hip: the gangs of 1-channel banks, on rows and on slabs): the error text, the launch-error macro, the
// handles' layouts and the descriptor / slab rings of a context.  Not installed; the boundary is include/sigdigger_amd.h.
#pragma once
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstddef>
#include <cstdint>
#include <deque>
#include <vector>

#include "../../include/sigdigger_amd.h"
#include "kernels.hpp"

// thread-local error text behind suamd_last_error() (defined in capi.hip)
void suamd_set_error(const char *fmt, ...);
#define set_err suamd_set_error

#define HIP_TRY(expr, ret)                                                          \
  do {                                                                              \
    hipError_t e__ = (expr);                                                        \
    if (e__ != hipSuccess) {                                                        \
      set_err("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e__), __FILE__, __LINE__); \
      return ret;                                                                   \
    }                                                                               \
  } while (0)

namespace {
inline hipStream_t as_stream(void *s) { return static_cast<hipStream_t>(s); }
inline sdk::View as_view(suamd_view v) { return sdk::View{(long long)v.chan_stride, (long long)v.time_stride}; }

// grow-only device scratch owned by a plan / bank
struct Scratch {
  void *p = nullptr;
  size_t bytes = 0;
  bool reserve(size_t need)
  {
    if (need <= bytes) return true;
    if (p) { (void)hipFree(p); p = nullptr; bytes = 0; }
    if (hipMalloc(&p, need) != hipSuccess) return false;
    bytes = need;
    return true;
  }
  void release() { if (p) (void)hipFree(p); p = nullptr; bytes = 0; }
};
}  // namespace

// ==========================================================================================
struct suamd_ctx {
  int device;
  // descriptor tables of the gang launches: a ring of slots in device memory.  The launches that read a slot are
  // enqueued right after its upload, so the next upload marks the slot's stream with the slot's event; when the
  // slot comes round again (128 uploads later) on another stream, that stream waits for the event on the device
  static constexpr int GANG_SLOTS = 128;
  static constexpr size_t GANG_SLOT_BYTES = 64 * 1024;
  float *fix_partial = nullptr;                             // suamd_source_fix: block sums
  char *gang_ring = nullptr;
  int gang_next = 0, gang_open = -1;                        // gang_open: the slot whose launches are being enqueued
  hipStream_t gang_user[GANG_SLOTS] = {};
  hipEvent_t gang_ev[GANG_SLOTS] = {};
  bool gang_marked[GANG_SLOTS] = {};
  // time-major slabs of the gang launches: a stream-ordered ring.  A region is handed out again only behind the event
  // its previous user recorded when it was done with it (a device-side wait on the new user's stream: the host
  // never blocks; hipMallocAsync / hipFreeAsync cost ~240 us per pair here)
  struct SlabUse { size_t off, size; hipEvent_t ev; };
  char *slab_base = nullptr;
  size_t slab_size = 0, slab_head = 0;
  std::deque<SlabUse> slab_live;
  std::vector<hipEvent_t> slab_spare;
  void *slab_take(size_t bytes, hipStream_t st, size_t *off_out)
  {
    bytes = (bytes + 255) & ~(size_t)255;
    if (bytes > slab_size) {                                 // grow: rare, and the only place that waits for the device
      (void)hipDeviceSynchronize();
      for (SlabUse &u : slab_live) slab_spare.push_back(u.ev);
      slab_live.clear();
      if (slab_base) (void)hipFree(slab_base);
      slab_base = nullptr; slab_size = 0; slab_head = 0;
      const size_t want = std::max<size_t>((size_t)256 << 20, 8 * bytes);
      if (hipMalloc((void **)&slab_base, want) != hipSuccess) { slab_base = nullptr; return nullptr; }
      slab_size = want;
    }
    if (slab_head + bytes > slab_size) slab_head = 0;
    const size_t off = slab_head;
    slab_head += bytes;
    for (auto it = slab_live.begin(); it != slab_live.end();) {
      if (it->off < off + bytes && off < it->off + it->size) {
        (void)hipStreamWaitEvent(st, it->ev, 0);
        slab_spare.push_back(it->ev);
        it = slab_live.erase(it);
      } else ++it;
    }
    *off_out = off;
    return slab_base + off;
  }
  void slab_give(size_t off, size_t bytes, hipStream_t st)
  {
    bytes = (bytes + 255) & ~(size_t)255;
    hipEvent_t ev = nullptr;
    if (!slab_spare.empty()) { ev = slab_spare.back(); slab_spare.pop_back(); }
    else if (hipEventCreateWithFlags(&ev, hipEventDisableTiming) != hipSuccess) { (void)hipStreamSynchronize(st); return; }
    (void)hipEventRecord(ev, st);
    slab_live.push_back(SlabUse{off, bytes, ev});
  }
  ~suamd_ctx()
  {
    for (SlabUse &u : slab_live) (void)hipEventDestroy(u.ev);
    for (hipEvent_t ev : slab_spare) (void)hipEventDestroy(ev);
    if (slab_base) (void)hipFree(slab_base);
    if (fix_partial) (void)hipFree(fix_partial);
    if (gang_ring) (void)hipFree(gang_ring);
    for (hipEvent_t ev : gang_ev) if (ev) (void)hipEventDestroy(ev);
  }
};


struct suamd_costas_bank {
  suamd_ctx *ctx;
  unsigned nchan;
  sdk::CostasParams p;
  sdk::CostasState  s;
};

struct suamd_pll_bank {
  suamd_ctx *ctx;
  unsigned nchan;
  float alpha, beta;
  sdk::PllState s;
};

struct suamd_clock_bank {
  suamd_ctx *ctx;
  unsigned nchan;
  sdk::ClockParams p;
  sdk::ClockState  s;
};

struct suamd_agc_bank {
  suamd_ctx *ctx;
  unsigned nchan;
  uint64_t n_fed;        // samples fed so far (history ring position = n_fed mod mag_history_size)
  sdk::AgcParams p;
  sdk::AgcState  s;
  Scratch scratch;       // 2 x [len][nchan] floats: magnitudes in dB; their sliding maximum, then levels
  hipEvent_t ev[2] = {nullptr, nullptr};   // suamd_agc_bank_feed_split: the two hops between its streams
};
struct suamd_cma_bank { suamd_ctx *ctx; unsigned nchan, n; float mu; int locked; suamd_complex *d_w, *d_dl; };

template <typename Item>
static Item *gang_upload(suamd_ctx *ctx, const std::vector<Item> &items, hipStream_t st)
{
  const size_t bytes = items.size() * sizeof(Item);
  if (bytes > suamd_ctx::GANG_SLOT_BYTES) { set_err("gang too large (%zu items)", items.size()); return nullptr; }
  if (!ctx->gang_ring && hipMalloc((void **)&ctx->gang_ring, suamd_ctx::GANG_SLOTS * suamd_ctx::GANG_SLOT_BYTES) != hipSuccess) {
    set_err("device allocation failed"); return nullptr;
  }
  const int si = ctx->gang_next;
  char *slot = ctx->gang_ring + (size_t)si * suamd_ctx::GANG_SLOT_BYTES;
  ctx->gang_next = (ctx->gang_next + 1) % suamd_ctx::GANG_SLOTS;
  if (ctx->gang_open >= 0) {                                  // the previous slot's launches are all enqueued by now
    const int po = ctx->gang_open;
    if (!ctx->gang_ev[po] && hipEventCreateWithFlags(&ctx->gang_ev[po], hipEventDisableTiming) != hipSuccess) ctx->gang_ev[po] = nullptr;
    ctx->gang_marked[po] = ctx->gang_ev[po] && hipEventRecord(ctx->gang_ev[po], ctx->gang_user[po]) == hipSuccess;
    if (!ctx->gang_marked[po]) (void)hipGetLastError();
  }
  if (ctx->gang_marked[si] && ctx->gang_user[si] != st) (void)hipStreamWaitEvent(st, ctx->gang_ev[si], 0);
  ctx->gang_marked[si] = false;
  ctx->gang_open = si;
  ctx->gang_user[si] = st;
  if (hipMemcpyAsync(slot, items.data(), bytes, hipMemcpyHostToDevice, st) != hipSuccess) { set_err("descriptor upload failed"); return nullptr; }
  return reinterpret_cast<Item *>(slot);
}

// One gang launch on its time-major slab: upload the items, gather their rows, run the recurrence, scatter the results.
// The slab comes from the context's stream-ordered ring (no host synchronisation, reused across calls).
template <typename Item, typename Launch>
static SUBOOL gang_tm(suamd_ctx *ctx, const std::vector<Item> &part, const std::vector<sdk::GangGroup> *groups, int elem_bytes, size_t off_src,
                      long long off_dst, size_t off_len, hipStream_t st, Launch launch)
{
  long long maxlen = 0;
  for (const Item &it : part) maxlen = std::max(maxlen, (long long)it.len);
  if (part.empty() || maxlen <= 0) return SU_TRUE;
  Item *d = gang_upload(ctx, part, st);
  if (!d) return SU_FALSE;
  sdk::GangGroup *dg = nullptr;
  if (groups) { dg = gang_upload(ctx, *groups, st); if (!dg) return SU_FALSE; }
  const long long slab = ((maxlen + 63) / 64 + 1) * 64 * 64;           // whole tiles + one of slack for the prefetch
  const size_t ngroups = groups ? groups->size() : (part.size() + 63) / 64;
  const size_t tm_bytes = ngroups * (size_t)slab * (size_t)elem_bytes;
  size_t tm_off = 0;
  void *tm = ctx->slab_take(tm_bytes, st, &tm_off);
  if (!tm) { set_err("device allocation failed (%zu B of gang slabs)", tm_bytes); return SU_FALSE; }
  hipError_t e = sdk::rows_tm_gather(d, (int)sizeof(Item), (int)off_src, (int)off_len, (int)part.size(), dg, (int)ngroups, elem_bytes, tm, slab, maxlen, st);
  if (e == hipSuccess) e = launch(d, dg, tm, slab);
  if (e == hipSuccess && off_dst >= 0)
    e = sdk::rows_tm_scatter(d, (int)sizeof(Item), (int)off_dst, (int)off_len, (int)part.size(), dg, (int)ngroups, elem_bytes, tm, slab, maxlen, st);
  ctx->slab_give(tm_off, tm_bytes, st);
  if (e != hipSuccess) { set_err("%s", hipGetErrorString(e)); return SU_FALSE; }
  return SU_TRUE;
}
