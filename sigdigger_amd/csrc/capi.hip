// capi.hip -- the C ABI of libsigdigger_amd.so (include/sigdigger_amd.h): contexts, plans,
// per-bank device state, parameter design (host, double precision) and kernel dispatch.
// There is deliberately no CPU implementation of any hot-path operation in this file: if no
// gfx950 device is usable every constructor fails and says why.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <algorithm>
#include <atomic>
#include <map>
#include <mutex>
#include <new>
#include <string>
#include <deque>
#include <vector>

#include "../../include/sigdigger_amd.h"
#include "kernels.hpp"
#include "tuning.hpp"
#include "design.hpp"
#include "capi_internal.hpp"

namespace {

thread_local std::string g_err;


}  // namespace

// the other translation units of the library report through the same thread-local message
void suamd_set_error(const char *fmt, ...)
{
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  g_err = buf;
}

namespace {


constexpr double kPi = 3.14159265358979323846;

template <typename T> T *dev_alloc(size_t n)
{
  void *p = nullptr;
  if (hipMalloc(&p, (n ? n : 1) * sizeof(T)) != hipSuccess) return nullptr;
  return static_cast<T *>(p);
}

template <typename T> bool dev_upload(T *dst, const T *src, size_t n)
{
  return hipMemcpy(dst, src, n * sizeof(T), hipMemcpyHostToDevice) == hipSuccess;
}

template <typename T> T *dev_from_host(const std::vector<T> &v)
{
  T *p = dev_alloc<T>(v.size());
  if (p && !v.empty() && !dev_upload(p, v.data(), v.size())) { hipFree(p); return nullptr; }
  return p;
}

template <typename T> T *dev_zeros(size_t n)
{
  T *p = dev_alloc<T>(n);
  // (hipMemset on device memory may return before the fill has run, and the callers' launches go to non-blocking streams,
  // which do not wait for the null stream: the fill must be over before the object is handed out)
  if (p && (hipMemset(p, 0, (n ? n : 1) * sizeof(T)) != hipSuccess || hipStreamSynchronize(nullptr) != hipSuccess)) { hipFree(p); return nullptr; }
  return p;
}



// ---- parameter design (host, double precision; not on the hot path) ----------------------

using sdk_design::butter_lp;                                // design.hpp (shared with sigutils_host.cpp)

void make_window(int type, std::vector<float> &w)
{
  const size_t n = w.size();
  const double d = (double)(n - 1);
  for (size_t i = 0; i < n; ++i) {
    const double t = 2.0 * kPi * (double)i / d;
    double v = 1.0;
    switch (type) {
      case SUAMD_WINDOW_HAMMING: v = 0.54 - 0.46 * std::cos(t); break;
      case SUAMD_WINDOW_HANN:    v = 0.5 - 0.5 * std::cos(t); break;
      case SUAMD_WINDOW_FLAT_TOP:
        v = 0.21557895 - 0.41663158 * std::cos(t) + 0.277263158 * std::cos(2 * t)
          - 0.083578947 * std::cos(3 * t) + 0.006947368 * std::cos(4 * t);
        break;
      case SUAMD_WINDOW_BLACKMANN_HARRIS:
        v = 0.35875 - 0.48829 * std::cos(t) + 0.14128 * std::cos(2 * t) - 0.01168 * std::cos(3 * t);
        break;
      default: v = 1.0;
    }
    w[i] = (float)v;
  }
}

}  // namespace

// ==========================================================================================
struct suamd_psd {
  suamd_ctx *ctx;
  unsigned n, log2n;
  float *d_window;
  void  *d_twiddle;      // float2[n]
  void  *d_tw_row = nullptr;   // frames beyond the LDS (psd_large.hip): W_N1 table of the row transforms, float2[N1]
  void  *d_tw_half = nullptr;  // 32768-point frames in one trip (psd_kernel HALVES): W_16384 table
  Scratch partial;       // split-frame partial sums
  int split_target = 0;  // suamd_psd_set_split_target (0: the kernel family's default)
};

struct suamd_chanbank {
  suamd_ctx *ctx;
  unsigned nchan, D, ntaps;
  uint64_t n_total;      // samples consumed so far (absolute index of the next x[0])
  float    *d_taps;      // real prototype [ntaps]
  void     *d_g;         // float4 [nchan][ntaps]: modulated taps as (re, re, -im, im)
  void     *d_g2;        // float2 [nchan][ntaps]: the same as (re, im) pairs (chan_stream.hip); 64 spare entries on either side
  void     *d_g2_base;   // the allocation behind d_g2
  uint32_t *d_dphase, *d_phase0;
  void     *d_hist[2];   // float2 [ntaps-1], ping-pong: d_hist[hist_cur] precedes the next block
  int       hist_cur;
  int       exclusive = 0;   // suamd_chanbank_set_exclusive
};



// ---- kernel timer -------------------------------------------------------------------------------------------------
namespace {
struct TimedLaunch { const char *name; hipEvent_t e0, e1; int dev; };
std::mutex g_timing_mu;
std::vector<TimedLaunch> g_timed;                              // pairs in flight, in launch order
// recycled pairs, per device: a HIP event belongs to the device that was current when it was created, and the sharded
// analyzer launches from one worker thread per GPU (a process-wide pool handed GPU 0's events to GPU 1's launches)
std::vector<std::pair<hipEvent_t, hipEvent_t>> g_timing_pool[64];
std::atomic<bool> g_timing{false};
}  // namespace
namespace {
// the gate of a timed launch (kernels.hpp): one lane polls a word of pinned host memory, 20 ms at most
__global__ void timing_gate_kernel(const volatile unsigned *flag)
{
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();            // 100 MHz
  while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) == 0u &&
         __builtin_amdgcn_s_memrealtime() - t0 < 2000000ull) __builtin_amdgcn_s_sleep(16);
}
constexpr unsigned kGateRing = 8192;
unsigned *g_gate_ring = nullptr;                               // pinned, mapped; a word per timed launch, reused round the ring
unsigned g_gate_next = 0;
}  // namespace
namespace sdk {
volatile unsigned *timing_gate(hipStream_t st)
{
  unsigned *w = nullptr;
  {
    std::lock_guard<std::mutex> lk(g_timing_mu);
    if (!g_gate_ring && hipHostMalloc((void **)&g_gate_ring, kGateRing * sizeof(unsigned), hipHostMallocMapped | hipHostMallocPortable) != hipSuccess) {
      g_gate_ring = nullptr;
      (void)hipGetLastError();
      return nullptr;
    }
    w = g_gate_ring + (g_gate_next++ % kGateRing);
  }
  __atomic_store_n(w, 0u, __ATOMIC_RELEASE);
  hipLaunchKernelGGL(timing_gate_kernel, dim3(1), dim3(1), 0, st, (const volatile unsigned *)w);
  if (hipGetLastError() != hipSuccess) { __atomic_store_n(w, 1u, __ATOMIC_RELEASE); return nullptr; }
  return w;
}
bool timing_on() { return g_timing.load(std::memory_order_relaxed); }
void timing_pair(const char *name, hipEvent_t *start, hipEvent_t *stop)
{
  int dev = 0;
  (void)hipGetDevice(&dev);                                    // the launching thread's device: the launch follows on it
  dev &= 63;
  std::lock_guard<std::mutex> lk(g_timing_mu);
  auto &pool = g_timing_pool[dev];
  if (!pool.empty()) { *start = pool.back().first; *stop = pool.back().second; pool.pop_back(); }
  else { (void)hipEventCreate(start); (void)hipEventCreate(stop); }
  g_timed.push_back({name, *start, *stop, dev});
}
}  // namespace sdk

extern "C" {

void suamd_kernel_timing(SUBOOL enable) { g_timing.store(enable != SU_FALSE); }

SUBOOL suamd_tuning_set(const char *name, long long value)
{
  if (sdk::tuning_set(name, value)) return SU_TRUE;
  set_err("suamd_tuning_set: no field '%s', or %lld outside its range", name ? name : "(null)", value);
  return SU_FALSE;
}
SUBOOL suamd_tuning_get(const char *name, long long *value) { return sdk::tuning_get(name, value) ? SU_TRUE : SU_FALSE; }
SUBOOL suamd_tuning_describe(unsigned index, const char **name, const char **env, long long *def, long long *lo, long long *hi, const char **doc)
{
  unsigned n = 0;
  const sdk::TuningField *f = sdk::tuning_fields(&n);
  if (index >= n) return SU_FALSE;
  if (name) *name = f[index].name;
  if (env) *env = f[index].env;
  if (def) *def = f[index].def;
  if (lo) *lo = f[index].lo;
  if (hi) *hi = f[index].hi;
  if (doc) *doc = f[index].doc;
  return SU_TRUE;
}
void suamd_tuning_reset(void) { sdk::tuning_reset(); }

SUBOOL suamd_kernel_timing_read(const char *kernel, double *sum_ms, double *min_ms, double *max_ms, unsigned *launches)
{
  std::vector<TimedLaunch> mine;
  {
    std::lock_guard<std::mutex> lk(g_timing_mu);
    size_t k = 0;
    for (const TimedLaunch &t : g_timed) {
      if (!kernel || std::strcmp(kernel, t.name) == 0) mine.push_back(t); else g_timed[k++] = t;
    }
    g_timed.resize(k);
  }
  double sum = 0, lo = 0, hi = 0; unsigned n = 0; bool ok = true;
  for (const TimedLaunch &t : mine) {
    float ms = 0;
    if (hipEventSynchronize(t.e1) != hipSuccess || hipEventElapsedTime(&ms, t.e0, t.e1) != hipSuccess) { ok = false; continue; }
    sum += ms; lo = n ? std::min(lo, (double)ms) : ms; hi = n ? std::max(hi, (double)ms) : ms; ++n;
  }
  {
    std::lock_guard<std::mutex> lk(g_timing_mu);
    for (const TimedLaunch &t : mine) g_timing_pool[t.dev].push_back({t.e0, t.e1});
  }
  if (sum_ms) *sum_ms = sum;
  if (min_ms) *min_ms = lo;
  if (max_ms) *max_ms = hi;
  if (launches) *launches = n;
  if (!ok) { set_err("kernel timer: an event pair could not be read"); return SU_FALSE; }
  return SU_TRUE;
}

const char *suamd_last_error(void) { return g_err.c_str(); }
const char *suamd_version(void) { return "sigdigger_amd 0.1 (gfx950)"; }

suamd_ctx_t *suamd_ctx_new(int device_ordinal)
{
  int count = 0;
  hipError_t e = hipGetDeviceCount(&count);
  if (e != hipSuccess || count <= 0) {
    set_err("no HIP device available (%s); libsigdigger_amd has no CPU fallback",
            e == hipSuccess ? "device count is 0" : hipGetErrorString(e));
    return nullptr;
  }
  if (device_ordinal < 0 || device_ordinal >= count) {
    set_err("device ordinal %d out of range (0..%d)", device_ordinal, count - 1);
    return nullptr;
  }
  HIP_TRY(hipSetDevice(device_ordinal), nullptr);
  hipDeviceProp_t prop;
  HIP_TRY(hipGetDeviceProperties(&prop, device_ordinal), nullptr);
  if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
    set_err("device %d is %s; this library only carries gfx950 code objects", device_ordinal, prop.gcnArchName);
    return nullptr;
  }
  suamd_ctx *ctx = new (std::nothrow) suamd_ctx;
  if (!ctx) { set_err("out of memory"); return nullptr; }
  ctx->device = device_ordinal;
  return ctx;
}

void suamd_ctx_destroy(suamd_ctx_t *ctx) { delete ctx; }
int  suamd_ctx_device(const suamd_ctx_t *ctx) { return ctx ? ctx->device : -1; }

SUBOOL suamd_chanbank_set_exclusive(suamd_chanbank_t *bank, SUBOOL exclusive)
{
  if (!bank) { set_err("null argument"); return SU_FALSE; }
  bank->exclusive = exclusive ? 1 : 0;
  return SU_TRUE;
}

SUBOOL suamd_psd_set_split_target(suamd_psd_t *psd, unsigned workgroups)
{
  if (!psd || workgroups > 65536) { set_err("bad argument"); return SU_FALSE; }
  psd->split_target = (int)workgroups;
  return SU_TRUE;
}

// ---- CU-masked streams ---------------------------------------------------------------------
unsigned suamd_ctx_cu_count(const suamd_ctx_t *ctx)
{
  if (!ctx) return 0;
  int n = 0;
  if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, ctx->device) != hipSuccess) return 0;
  return n > 0 ? (unsigned)n : 0;
}

void *suamd_stream_new_cu_mask(suamd_ctx_t *ctx, const uint32_t *cu_mask, unsigned nwords)
{
  if (!ctx || !cu_mask || !nwords) { set_err("null context or empty CU mask"); return nullptr; }
  uint32_t any = 0;
  for (unsigned i = 0; i < nwords; ++i) any |= cu_mask[i];
  if (!any) { set_err("a CU mask without a compute unit"); return nullptr; }
  HIP_TRY(hipSetDevice(ctx->device), nullptr);
  hipStream_t st = nullptr;
  HIP_TRY(hipExtStreamCreateWithCUMask(&st, nwords, cu_mask), nullptr);
  return st;
}

SUBOOL suamd_stream_destroy(suamd_ctx_t *ctx, void *stream)
{
  if (!ctx || !stream) { set_err("null context or stream"); return SU_FALSE; }
  HIP_TRY(hipSetDevice(ctx->device), SU_FALSE);
  HIP_TRY(hipStreamDestroy((hipStream_t)stream), SU_FALSE);
  return SU_TRUE;
}

namespace {
__global__ void __launch_bounds__(64) placement_kernel(uint32_t *where, unsigned spin)
{
  unsigned hw, xcc;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  const unsigned long long t0 = __builtin_readcyclecounter();
  while (__builtin_readcyclecounter() - t0 < spin) __builtin_amdgcn_s_sleep(8);
  if (threadIdx.x == 0)                                        // HW_ID: CU_ID [11:8], SH_ID [12], SE_ID [15:13]
    where[blockIdx.x] = (xcc & 0xf) << 16 | ((hw >> 12) & 0xf) << 8 | ((hw >> 8) & 0xf);
}
}

SUBOOL suamd_probe_placement(suamd_ctx_t *ctx, void *stream, unsigned nblocks, unsigned spin_ticks, uint32_t *h_where)
{
  if (!ctx || !h_where || !nblocks) { set_err("null argument"); return SU_FALSE; }
  HIP_TRY(hipSetDevice(ctx->device), SU_FALSE);
  uint32_t *d = nullptr;
  HIP_TRY(hipMalloc(&d, sizeof(uint32_t) * nblocks), SU_FALSE);
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(placement_kernel, dim3(nblocks), dim3(64), 0, st, d, spin_ticks);
  hipError_t e = hipGetLastError();
  if (e == hipSuccess) e = hipMemcpyAsync(h_where, d, sizeof(uint32_t) * nblocks, hipMemcpyDeviceToHost, st);
  if (e == hipSuccess) e = hipStreamSynchronize(st);
  (void)hipFree(d);
  if (e != hipSuccess) { set_err("placement probe: %s", hipGetErrorString(e)); return SU_FALSE; }
  return SU_TRUE;
}

// ---- PSD -----------------------------------------------------------------------------------
suamd_psd_t *suamd_psd_new(suamd_ctx_t *ctx, unsigned n, int window_type)
{
  if (!ctx) { set_err("null context"); return nullptr; }
  unsigned log2n = 0;
  while ((1u << log2n) < n) ++log2n;
  if ((1u << log2n) != n || log2n < 9 || log2n > 20) {
    set_err("window_size %u unsupported (power of two, 512..1048576)", n);
    return nullptr;
  }
  if (window_type < SUAMD_WINDOW_NONE || window_type > SUAMD_WINDOW_BLACKMANN_HARRIS) {
    set_err("unknown window type %d", window_type);
    return nullptr;
  }
  HIP_TRY(hipSetDevice(ctx->device), nullptr);
  std::vector<float> w(n);
  make_window(window_type, w);
  const unsigned ntw = n;                            // W_N table (the in-LDS kernels' passes; the column pass of psd_large.hip)
  std::vector<float> tw(2 * (size_t)ntw);
  for (unsigned i = 0; i < ntw; ++i) {
    const double ang = -2.0 * kPi * (double)i / (double)n;
    tw[2 * i] = (float)std::cos(ang);
    tw[2 * i + 1] = (float)std::sin(ang);
  }
  suamd_psd *p = new (std::nothrow) suamd_psd();
  if (!p) { set_err("out of memory"); return nullptr; }
  p->ctx = ctx; p->n = n; p->log2n = log2n;
  p->d_window = dev_from_host(w);
  p->d_twiddle = dev_from_host(tw);
  if (log2n > 14) {                                  // rows of N1 = N / N2 points: their own table
    const unsigned n1 = n >> sdk::psd_large_log2n2((int)log2n);
    std::vector<float> tr(2 * (size_t)n1);
    for (unsigned i = 0; i < n1; ++i) {
      const double ang = -2.0 * kPi * (double)i / (double)n1;
      tr[2 * i] = (float)std::cos(ang);
      tr[2 * i + 1] = (float)std::sin(ang);
    }
    p->d_tw_row = dev_from_host(tr);
    if (log2n == 15) {
      std::vector<float> th(2 * (size_t)(n / 2));
      for (unsigned i = 0; i < n / 2; ++i) {
        const double ang = -2.0 * kPi * (double)i / (double)(n / 2);
        th[2 * i] = (float)std::cos(ang);
        th[2 * i + 1] = (float)std::sin(ang);
      }
      p->d_tw_half = dev_from_host(th);
    }
  }
  if (!p->d_window || !p->d_twiddle || (log2n > 14 && !p->d_tw_row)) {
    set_err("device allocation failed");
    suamd_psd_destroy(p);
    return nullptr;
  }
  return p;
}

void suamd_psd_destroy(suamd_psd_t *p)
{
  if (!p) return;
  if (p->d_window) hipFree(p->d_window);
  if (p->d_twiddle) hipFree(p->d_twiddle);
  if (p->d_tw_row) hipFree(p->d_tw_row);
  if (p->d_tw_half) hipFree(p->d_tw_half);
  p->partial.release();
  delete p;
}

SUBOOL suamd_psd_feed(suamd_psd_t *p, const suamd_complex *d_x, SUSCOUNT nframes, SUSCOUNT hop, unsigned navg,
                      SUFLOAT scale, int mode, SUFLOAT *d_out, void *stream)
{
  if (!p || !d_x || !d_out) { set_err("null argument"); return SU_FALSE; }
  if (navg == 0) { set_err("navg must be >= 1"); return SU_FALSE; }
  if (mode != SUAMD_PSD_LINEAR && mode != SUAMD_PSD_DB_SHIFTED) { set_err("bad mode %d", mode); return SU_FALSE; }
  const long long nout = (long long)(nframes / navg);
  if (p->log2n > 14) {
    // FFTWidget offers 2^9..2^20 (Default/FFT/FFTWidget.cpp:350-351) and the scanner uses
    // nextPow2(fs / 1 kHz) (Panoramic/Scanner.cpp:323): frames beyond the LDS go pass by pass through HBM
    // (batches of up to 16 Mi points = 128 MiB per ping-pong buffer, the whole job if it is smaller)
    // SUAMD_PSD_LARGE (read per call: the tests switch it): "passes" = round 2's radix-16 passes through HBM, "twotrip" =
    // psd_large.hip for 32768 points too
    const bool passes = sdk::tuning().psd_large == 0, two_trip = sdk::tuning().psd_large == 1;
    if (p->log2n == 15 && !passes && !two_trip && p->d_tw_half) {
      // 32768 points (the scanner at 20 MS/s): ONE trip through HBM -- two workgroups per output on the 16384-point kernel,
      // each forms one of the two interleaved half spectra's operands from the whole frame on the way in (psd.hip, HALVES)
      const int S = sdk::psd_split(2 * nout, (int)navg, 14);
      float *partial = nullptr;
      if (S > 1) {
        if (!p->partial.reserve(sizeof(float) * (size_t)nout * S * p->n)) { set_err("scratch allocation failed"); return SU_FALSE; }
        partial = static_cast<float *>(p->partial.p);
      }
      HIP_TRY(sdk::psd_frames_32k(d_x, (long long)hop, (int)navg, p->d_window, p->d_tw_half, p->d_twiddle, scale, mode, d_out, nout,
                                  partial, as_stream(stream)), SU_FALSE);
      return SU_TRUE;
    }
    if (!passes) {
      // psd_large.hip: two trips through HBM (column transforms on registers, row transforms in LDS)
      const long long batch_points = 1ll << sdk::tuning().psd_large_points;
      long long batch = batch_points / (long long)p->n;         // 128 Mi points = 1 GiB of intermediate (measured: small batches that would fit the last-level cache lose more to launch tails than they gain)
      if (batch > 32768) batch = 32768;                         // grid.y of the column pass
      if (batch > nout * (long long)navg) batch = nout * (long long)navg;
      if (const long long v = sdk::tuning().psd_large_batch; v >= 1 && v < batch) batch = v;   // tests: awkward batch boundaries
      if (batch < 1) batch = 1;
      const int ch = sdk::psd_large_chunk((int)navg), cpo = ((int)navg + ch - 1) / ch;
      // ring of chunk sums: everything between the first chunk of the batch's first (possibly still open) output and the
      // batch's last chunk is live at once.  A batch of `batch` frames touches at most (batch + navg - 2) / navg + 1 outputs of
      // cpo = ceil(navg / ch) chunks each -- NOT batch / ch chunks: with navg no multiple of ch every output's last chunk is
      // short (round 3 sized the ring by batch / ch and let chunks of one batch share slots: ADVICE r3)
      const long long pring_ll = ((batch + (long long)navg - 2) / (long long)navg + 1) * cpo + 2;
      if (pring_ll > 0x7fffffff) { set_err("navg too small for this batch"); return SU_FALSE; }
      const int pring = (int)pring_ll;
      const size_t ab = sizeof(suamd_complex) * (size_t)p->n * (size_t)batch;
      if (!p->partial.reserve(ab + sizeof(float) * (size_t)p->n * (size_t)pring)) { set_err("scratch allocation failed"); return SU_FALSE; }
      char *base = static_cast<char *>(p->partial.p);
      HIP_TRY(sdk::psd_frames_large2((int)p->log2n, d_x, (long long)hop, (int)navg, p->d_window, p->d_twiddle, p->d_tw_row, scale, mode,
                                     d_out, nout, base, reinterpret_cast<float *>(base + ab), pring, (int)batch, as_stream(stream)), SU_FALSE);
      return SU_TRUE;
    }
    long long batch = (1ll << 24) / (long long)p->n;
    if (batch > nout * (long long)navg) batch = nout * (long long)navg;
    if (const long long v = sdk::tuning().psd_large_batch; v >= 1 && v < batch) batch = v;   // tests: awkward batch boundaries
    if (batch < 1) batch = 1;
    const size_t cb = sizeof(suamd_complex) * (size_t)p->n * (size_t)batch;
    if (!p->partial.reserve(2 * cb + sizeof(float) * (size_t)p->n)) { set_err("scratch allocation failed"); return SU_FALSE; }
    char *base = static_cast<char *>(p->partial.p);
    HIP_TRY(sdk::psd_frames_large((int)p->log2n, d_x, (long long)hop, (int)navg, p->d_window, scale, mode, d_out, nout,
                                  base, base + cb, reinterpret_cast<float *>(base + 2 * cb), (int)batch, as_stream(stream)), SU_FALSE);
    return SU_TRUE;
  }
  struct Plan { Plan(int t) { sdk::psd_split_target(t); } ~Plan() { sdk::psd_split_target(0); } } plan(p->split_target);
  const int S = sdk::psd_split(nout, (int)navg, (int)p->log2n);
  float *partial = nullptr;
  if (S > 1) {
    // NOTE: growing the scratch frees the old one; callers that enqueue on several streams must
    // size the plan with the largest request first
    if (!p->partial.reserve(sizeof(float) * (size_t)nout * S * p->n)) { set_err("scratch allocation failed"); return SU_FALSE; }
    partial = static_cast<float *>(p->partial.p);
  }
  HIP_TRY(sdk::psd_frames((int)p->log2n, d_x, (long long)hop, (int)navg, p->d_window, p->d_twiddle, scale, mode,
                          d_out, nout, partial, as_stream(stream)), SU_FALSE);
  return SU_TRUE;
}

SUBOOL suamd_psd_shift_db(suamd_ctx_t *ctx, SUFLOAT *d_psd, SUSCOUNT n, SUSCOUNT nframes, void *stream)
{
  if (!ctx || !d_psd) { set_err("null argument"); return SU_FALSE; }
  HIP_TRY(sdk::psd_shift_db(d_psd, (long long)n, (long long)nframes, as_stream(stream)), SU_FALSE);
  return SU_TRUE;
}

SUBOOL suamd_averager_feed(suamd_ctx_t *ctx, SUFLOAT *d_last, const SUFLOAT *d_x, SUSCOUNT n, SUFLOAT alpha,
                           SUBOOL blend, void *stream)
{
  if (!ctx || !d_last || !d_x) { set_err("null argument"); return SU_FALSE; }
  HIP_TRY(sdk::averager_feed(d_last, d_x, (long long)n, alpha, blend ? 1 : 0, as_stream(stream)), SU_FALSE);
  return SU_TRUE;
}

SUBOOL suamd_inspector_spectrum_db_shift(suamd_ctx_t *ctx, SUFLOAT *d_data, SUSCOUNT len, SUSCOUNT nspectra, void *stream)
{
  if (!ctx || !d_data) { set_err("null argument"); return SU_FALSE; }
  HIP_TRY(sdk::insp_spectrum_db_shift(d_data, (long long)len, (long long)nspectra, as_stream(stream)), SU_FALSE);
  return SU_TRUE;
}

// ---- NCO -----------------------------------------------------------------------------------
uint32_t suamd_fnor_to_dphase(double fnor)
{
  const long long v = std::llrint(fnor * 2147483648.0);
  return (uint32_t)(v & 0xFFFFFFFFll);
}

SUBOOL suamd_xlate_bulk(suamd_ctx_t *ctx, const suamd_complex *d_x, suamd_complex *d_y, SUSCOUNT len,
                        uint32_t phase0, uint32_t dphase, SUSCOUNT n0, void *stream)
{
  if (!ctx || !d_x || !d_y) { set_err("null argument"); return SU_FALSE; }
  HIP_TRY(sdk::xlate_bulk(d_x, d_y, (long long)len, phase0, dphase, n0, as_stream(stream)), SU_FALSE);
  return SU_TRUE;
}

// ---- channel bank ----------------------------------------------------------------------------
void suamd_lpf_design(SUFLOAT *taps, unsigned ntaps, double fc)
{
  std::vector<double> d(ntaps);
  double sum = 0;
  for (unsigned i = 0; i < ntaps; ++i) {
    const double t = (double)i - 0.5 * (double)(ntaps - 1);
    const double a = kPi * fc * t;
    const double sinc = std::fabs(a) < 1e-12 ? 1.0 : std::sin(a) / a;
    const double w = ntaps > 1 ? 0.54 - 0.46 * std::cos(2.0 * kPi * (double)i / (double)(ntaps - 1)) : 1.0;
    d[i] = fc * sinc * w;
    sum += d[i];
  }
  for (unsigned i = 0; i < ntaps; ++i) taps[i] = (float)(d[i] / sum);
}

suamd_chanbank_t *suamd_chanbank_new(suamd_ctx_t *ctx, unsigned nchan, const double *fnor, unsigned decimation,
                                     const SUFLOAT *taps, unsigned ntaps)
{
  if (!ctx || !fnor || !taps) { set_err("null argument"); return nullptr; }
  if (nchan == 0 || decimation == 0 || ntaps == 0) { set_err("nchan, decimation and ntaps must be > 0"); return nullptr; }
  HIP_TRY(hipSetDevice(ctx->device), nullptr);
  suamd_chanbank *b = new (std::nothrow) suamd_chanbank;
  if (!b) { set_err("out of memory"); return nullptr; }
  std::memset(b, 0, sizeof *b);
  b->ctx = ctx; b->nchan = nchan; b->D = decimation; b->ntaps = ntaps; b->n_total = 0;
  std::vector<uint32_t> dp(nchan), p0(nchan, 0u);
  // translate by -fc: Tasks/CarrierXlator.cpp:36 initialises the NCO with -relFreq
  for (unsigned c = 0; c < nchan; ++c) dp[c] = suamd_fnor_to_dphase(-fnor[c]);
  b->d_taps   = dev_alloc<float>(ntaps);
  b->d_g      = dev_alloc<float>(4 * (size_t)nchan * ntaps);
  b->d_g2_base = dev_zeros<float>(2 * ((size_t)nchan * ntaps + 128));   // the stream kernels read up to D taps in front of a row and two runs behind it
  b->d_g2     = b->d_g2_base ? static_cast<float *>(b->d_g2_base) + 2 * 64 : nullptr;
  b->d_dphase = dev_from_host(dp);
  b->d_phase0 = dev_from_host(p0);
  b->d_hist[0] = dev_zeros<float>(2 * (size_t)(ntaps > 1 ? ntaps - 1 : 1));
  b->d_hist[1] = dev_zeros<float>(2 * (size_t)(ntaps > 1 ? ntaps - 1 : 1));
  b->hist_cur = 0;
  if (!b->d_taps || !b->d_g || !b->d_g2 || !b->d_dphase || !b->d_phase0 || !b->d_hist[0] || !b->d_hist[1] ||
      !dev_upload(b->d_taps, taps, ntaps)) {
    set_err("device allocation failed");
    suamd_chanbank_destroy(b);
    return nullptr;
  }
  hipError_t e = sdk::chan_modulate_taps(b->d_taps, (int)ntaps, b->d_dphase, (int)nchan, b->d_g, b->d_g2, nullptr);
  if (e == hipSuccess) e = hipStreamSynchronize(nullptr);
  if (e != hipSuccess) {
    set_err("tap modulation kernel failed: %s", hipGetErrorString(e));
    suamd_chanbank_destroy(b);
    return nullptr;
  }
  return b;
}

void suamd_chanbank_destroy(suamd_chanbank_t *b)
{
  if (!b) return;
  if (b->d_taps) hipFree(b->d_taps);
  if (b->d_g) hipFree(b->d_g);
  if (b->d_g2_base) hipFree(b->d_g2_base);
  if (b->d_dphase) hipFree(b->d_dphase);
  if (b->d_phase0) hipFree(b->d_phase0);
  if (b->d_hist[0]) hipFree(b->d_hist[0]);
  if (b->d_hist[1]) hipFree(b->d_hist[1]);
  delete b;
}

static void chan_out_range(const suamd_chanbank *b, SUSCOUNT len, uint64_t *m_first, SUSCOUNT *n_out)
{
  const uint64_t n0 = b->n_total, D = b->D;
  const uint64_t mf = (n0 + D - 1) / D;
  *m_first = mf;
  if (len == 0 || mf * D >= n0 + len) { *n_out = 0; return; }
  const uint64_t ml = (n0 + len - 1) / D;
  *n_out = ml - mf + 1;
}

SUSCOUNT suamd_chanbank_output_count(const suamd_chanbank_t *b, SUSCOUNT len)
{
  if (!b) return 0;
  uint64_t mf; SUSCOUNT n;
  chan_out_range(b, len, &mf, &n);
  return n;
}

SUBOOL suamd_chanbank_feed(suamd_chanbank_t *b, const suamd_complex *d_x, SUSCOUNT len, suamd_complex *d_y,
                           suamd_view yv, SUSCOUNT *n_out, void *stream)
{
  if (!b) { set_err("null argument"); return SU_FALSE; }
  if (len == 0) { if (n_out) *n_out = 0; return SU_TRUE; }          // an empty block is a no-op
  if (!d_x || !d_y) { set_err("null argument"); return SU_FALSE; }
  uint64_t mf; SUSCOUNT no;
  chan_out_range(b, len, &mf, &no);
  if (yv.time_stride == 1 && b->nchan > 1 && no > yv.chan_stride) {
    set_err("chan_stride %llu smaller than output count %llu", (unsigned long long)yv.chan_stride, (unsigned long long)no);
    return SU_FALSE;
  }
  sdk::ChanFeedArgs a;
  a.x = d_x; a.hist = b->d_hist[b->hist_cur]; a.hist_next = b->d_hist[b->hist_cur ^ 1];
  a.len = (long long)len; a.n0 = b->n_total;
  a.g = b->d_g; a.g2 = b->d_g2; a.dphase = b->d_dphase; a.phase0 = b->d_phase0;
  a.ntaps = (int)b->ntaps; a.nchan = (int)b->nchan; a.D = b->D;
  a.m_first = mf; a.n_out = (long long)no; a.y = d_y; a.yv = as_view(yv);
  a.exclusive = b->exclusive;
  HIP_TRY(sdk::chan_feed(a, as_stream(stream)), SU_FALSE);
  b->hist_cur ^= 1;
  b->n_total += len;
  if (n_out) *n_out = no;
  return SU_TRUE;
}

SUBOOL suamd_chanbank_gang_feed(suamd_ctx_t *ctx, suamd_chanbank_t *const *banks, unsigned n, const suamd_complex *d_x, SUSCOUNT len,
                                suamd_complex *const *d_y, SUSCOUNT *n_out, void *stream)
{
  if (!ctx || (n && (!banks || !d_y))) { set_err("null argument"); return SU_FALSE; }
  if (len == 0) { for (unsigned i = 0; n_out && i < n; ++i) n_out[i] = 0; return SU_TRUE; }
  if (!d_x) { set_err("null argument"); return SU_FALSE; }
  hipStream_t st = as_stream(stream);
  std::vector<sdk::ChanGangItem> items(n);
  unsigned max_tiles = 0, max_lds = 0;
  for (unsigned i = 0; i < n; ++i) {
    suamd_chanbank *b = banks[i];
    if (!b || b->nchan != 1) { set_err("gang members must be 1-channel banks"); return SU_FALSE; }
    if (!d_y[i]) { set_err("null row"); return SU_FALSE; }
    uint64_t mf; SUSCOUNT no;
    chan_out_range(b, len, &mf, &no);
    sdk::ChanFeedArgs a;
    a.x = d_x; a.hist = b->d_hist[b->hist_cur]; a.hist_next = b->d_hist[b->hist_cur ^ 1];
    a.len = (long long)len; a.n0 = b->n_total;
    a.g = b->d_g; a.g2 = nullptr; a.dphase = b->d_dphase; a.phase0 = b->d_phase0;
    a.ntaps = (int)b->ntaps; a.nchan = 1; a.D = b->D;
    a.m_first = mf; a.n_out = (long long)no; a.y = d_y[i]; a.yv = sdk::View{0, 1};
    if (sdk::chan_gang_plan(a, &items[i]) != hipSuccess) { set_err("bank %u does not fit the gang kernel (decimation %u, %u taps)", i, b->D, b->ntaps); return SU_FALSE; }
    max_tiles = std::max(max_tiles, items[i].ntiles);
    max_lds = std::max(max_lds, items[i].lds);
    if (n_out) n_out[i] = no;
  }
  for (size_t o = 0; o < items.size(); o += 256) {
    std::vector<sdk::ChanGangItem> part(items.begin() + o, items.begin() + std::min(items.size(), o + 256));
    sdk::ChanGangItem *d = gang_upload(ctx, part, st);
    if (!d) return SU_FALSE;
    HIP_TRY(sdk::chan_gang_feed(d, (int)part.size(), max_tiles, max_lds, st), SU_FALSE);
  }
  for (unsigned i = 0; i < n; ++i) { banks[i]->hist_cur ^= 1; banks[i]->n_total += len; }
  return SU_TRUE;
}

SUBOOL suamd_chanbank_reset(suamd_chanbank_t *b, void *stream)
{
  if (!b) { set_err("null argument"); return SU_FALSE; }
  b->n_total = 0;
  if (b->ntaps > 1)
    HIP_TRY(hipMemsetAsync(b->d_hist[b->hist_cur], 0, 2 * sizeof(float) * (b->ntaps - 1), as_stream(stream)), SU_FALSE);
  return SU_TRUE;
}

// ---- element-wise ------------------------------------------------------------------------------
SUBOOL suamd_quad_demod_batch(suamd_ctx_t *ctx, const suamd_complex *d_x, suamd_view xv, suamd_complex *d_y,
                              suamd_view yv, unsigned nchan, SUSCOUNT len, const suamd_complex *d_prev,
                              SUBOOL first, suamd_complex *d_prev_out, void *stream)
{
  if (!ctx || !d_x || !d_y) { set_err("null argument"); return SU_FALSE; }
  if (!first && !d_prev) { set_err("d_prev required when first == SU_FALSE"); return SU_FALSE; }
  HIP_TRY(sdk::quad_demod_batch(d_x, as_view(xv), d_y, as_view(yv), (int)nchan, (long long)len,
                                d_prev, first ? 1 : 0, d_prev_out, as_stream(stream)), SU_FALSE);
  return SU_TRUE;
}

SUBOOL suamd_delayed_conj_bulk(suamd_ctx_t *ctx, const suamd_complex *d_x, suamd_complex *d_y, SUSCOUNT len,
                               SUSCOUNT delay, void *stream)
{
  if (!ctx || !d_x || !d_y) { set_err("null argument"); return SU_FALSE; }
  if (d_x == d_y) { set_err("delayed_conj cannot run in place"); return SU_FALSE; }
  HIP_TRY(sdk::delayed_conj_bulk(d_x, d_y, (long long)len, (long long)delay, as_stream(stream)), SU_FALSE);
  return SU_TRUE;
}

SUBOOL suamd_histogram_feed_bulk(suamd_ctx_t *ctx, const suamd_complex *d_x, SUSCOUNT len, int space,
                                 SUFLOAT *d_out, void *stream)
{
  if (!ctx || !d_x || !d_out) { set_err("null argument"); return SU_FALSE; }
  if (space < 0 || space > 2) { set_err("bad space %d", space); return SU_FALSE; }
  HIP_TRY(sdk::histogram_feed_bulk(d_x, (long long)len, space, d_out, as_stream(stream)), SU_FALSE);
  return SU_TRUE;
}

SUBOOL suamd_sample_manual_bulk(suamd_ctx_t *ctx, const suamd_complex *d_data, SUSCOUNT length, double symbol_count,
                                SUSCOUNT symbol_sync, int space, suamd_complex *d_out, SUSCOUNT nout, void *stream)
{
  if (!ctx || !d_data || !d_out) { set_err("null argument"); return SU_FALSE; }
  if (!(symbol_count > 0) || space < 0 || space > 2) { set_err("bad symbol_count / space"); return SU_FALSE; }
  HIP_TRY(sdk::sample_manual_bulk(d_data, (long long)length, symbol_count, (double)symbol_sync, space, d_out,
                                  (long long)nout, as_stream(stream)), SU_FALSE);
  return SU_TRUE;
}

SUBOOL suamd_ingest_iq(suamd_ctx_t *ctx, int format, const void *d_raw, SUSCOUNT nsamples, suamd_complex *d_out, void *stream)
{
  if (!ctx) { set_err("null context"); return SU_FALSE; }
  if (nsamples == 0) return SU_TRUE;
  if (!d_raw || !d_out) { set_err("null argument"); return SU_FALSE; }
  if (format < SUAMD_FORMAT_RAW_FLOAT32 || format > SUAMD_FORMAT_RAW_SIGNED16) { set_err("unsupported sample format %d", format); return SU_FALSE; }
  if ((format == SUAMD_FORMAT_RAW_UNSIGNED8 || format == SUAMD_FORMAT_RAW_SIGNED8 || format == SUAMD_FORMAT_RAW_SIGNED16) &&
      (reinterpret_cast<uintptr_t>(d_raw) & 15)) { set_err("d_raw must be 16-byte aligned"); return SU_FALSE; }
  HIP_TRY(sdk::ingest_iq(format, d_raw, (long long)nsamples, d_out, as_stream(stream)), SU_FALSE);
  return SU_TRUE;
}

SUBOOL suamd_source_fix(suamd_ctx_t *ctx, suamd_complex *d_x, SUSCOUNT nsamples, SUBOOL iq_reverse, SUFLOAT *d_dc, SUFLOAT alpha,
                        SUBOOL first, void *stream)
{
  if (!ctx) { set_err("null argument"); return SU_FALSE; }
  if (nsamples == 0 || (!iq_reverse && !d_dc)) return SU_TRUE;
  if (!d_x) { set_err("null argument"); return SU_FALSE; }
  if (d_dc && !(alpha > 0.0f && alpha <= 1.0f)) { set_err("alpha %g out of (0, 1]", (double)alpha); return SU_FALSE; }
  if (d_dc && !ctx->fix_partial && hipMalloc((void **)&ctx->fix_partial, 512 * sizeof(float)) != hipSuccess) {
    ctx->fix_partial = nullptr; set_err("device allocation failed"); return SU_FALSE;
  }
  HIP_TRY(sdk::source_fix(d_x, (long long)nsamples, iq_reverse ? 1 : 0, d_dc, alpha, first ? 1 : 0, ctx->fix_partial, as_stream(stream)), SU_FALSE);
  return SU_TRUE;
}

// A5: UIMediator::feedPSD's expiry rule (UIMediator/SpectrumMediator.cpp:35-85, :127); the lag warning for remote
// analyzers (:87-124) is GUI and not part of it
SUBOOL suamd_psd_ttl_accept(suamd_psd_ttl_t *s, double now_s, double rt_time_s, double ttl_ms, SUBOOL looped)
{
  if (!s) return SU_TRUE;
  bool expired = false;
  const double max_delta = ttl_ms * 1e-3;
  double delta = now_s - rt_time_s;                        // :50-51
  if (s->rt_calibrations++ == 0) s->rt_delta_real = delta; // :58-59
  else s->rt_delta_real += (1.0 - std::exp(-1.0 / 10.0)) * (delta - s->rt_delta_real);   // SU_SPLPF_FEED(.., SU_SPLPF_ALPHA(CAL_LEN)) :65-68
  if (!s->have_rt_delta) {
    if (++s->rt_calibrations > 10) s->have_rt_delta = SU_TRUE;   // :76-78
  } else {
    delta -= s->rt_delta_real;                             // :81-82
    expired = delta > max_delta;
  }
  return (!expired || looped) ? SU_TRUE : SU_FALSE;        // :127
}

unsigned suamd_format_bytes_per_sample(int format)
{
  switch (format) {
    case SUAMD_FORMAT_RAW_FLOAT32: return 8;
    case SUAMD_FORMAT_RAW_UNSIGNED8: case SUAMD_FORMAT_RAW_SIGNED8: return 2;
    case SUAMD_FORMAT_RAW_SIGNED16: return 4;
    default: return 0;
  }
}

SUBOOL suamd_conj_prev_bulk(suamd_ctx_t *ctx, const suamd_complex *d_x, suamd_complex *d_y, SUSCOUNT len,
                            SUFLOAT prev_re, SUFLOAT prev_im, void *stream)
{
  if (!ctx || !d_x || !d_y) { set_err("null argument"); return SU_FALSE; }
  HIP_TRY(sdk::conj_prev_bulk(d_x, d_y, (long long)len, prev_re, prev_im, as_stream(stream)), SU_FALSE);
  return SU_TRUE;
}

SUSDIFF suamd_sample_zero_crossing_bulk(suamd_ctx_t *ctx, const suamd_complex *d_data, SUSCOUNT length, SUFLOAT bnor,
                                        int space, SUBOOL amplitude, SUFLOAT thr_re, SUFLOAT thr_im, SUFLOAT ang_re,
                                        SUFLOAT ang_im, unsigned char *d_symbols, SUSCOUNT capacity, void *stream)
{
  if (!ctx) { set_err("null context"); return -1; }
  if (space < 0 || space > 2) { set_err("bad space %d", space); return -1; }
  if (length == 0) return 0;
  if (!d_data || !d_symbols) { set_err("null argument"); return -1; }
  const long long nblocks = (long long)((length + 4095) / 4096);
  if (capacity < (SUSCOUNT)nblocks * 4096) { set_err("d_symbols must hold 4096 * ceil(length / 4096) symbols"); return -1; }
  hipStream_t st = as_stream(stream);
  HIP_TRY(hipSetDevice(ctx->device), -1);
  // scratch: var[length] | last_pos[nblocks] | offset[nblocks] | count[nblocks] | seg[nblocks * 4096]
  const size_t o_var = 0, o_last = o_var + ((sizeof(float) * length + 15) & ~(size_t)15);
  const size_t o_off = o_last + sizeof(long long) * nblocks, o_cnt = o_off + sizeof(unsigned long long) * nblocks;
  const size_t o_seg = (o_cnt + sizeof(unsigned) * nblocks + 15) & ~(size_t)15;
  char *d = nullptr;
  HIP_TRY(hipMalloc(&d, o_seg + (size_t)nblocks * 4096), -1);
  float *d_var = reinterpret_cast<float *>(d + o_var);
  long long *d_last = reinterpret_cast<long long *>(d + o_last);
  unsigned long long *d_off = reinterpret_cast<unsigned long long *>(d + o_off);
  unsigned *d_cnt = reinterpret_cast<unsigned *>(d + o_cnt);
  unsigned char *d_seg = reinterpret_cast<unsigned char *>(d + o_seg);
  SUSDIFF total = -1;
  std::vector<unsigned> cnt((size_t)nblocks);
  std::vector<unsigned long long> off((size_t)nblocks);
  hipError_t e = sdk::zc_var(d_data, (long long)length, space, amplitude ? 1 : 0, thr_re, thr_im, ang_re, ang_im, d_var, st);
  if (e == hipSuccess) e = sdk::zc_scan(d_var, (long long)length, nblocks, d_last, st);
  if (e == hipSuccess) e = sdk::zc_emit(d_var, (long long)length, nblocks, bnor, d_last, d_seg, d_cnt, st);
  if (e == hipSuccess) e = hipMemcpyAsync(cnt.data(), d_cnt, sizeof(unsigned) * nblocks, hipMemcpyDeviceToHost, st);
  if (e == hipSuccess) e = hipStreamSynchronize(st);
  if (e == hipSuccess) {
    unsigned long long acc = 0;
    for (long long b = 0; b < nblocks; ++b) { off[(size_t)b] = acc; acc += cnt[(size_t)b]; }
    total = (SUSDIFF)acc;
    e = hipMemcpyAsync(d_off, off.data(), sizeof(unsigned long long) * nblocks, hipMemcpyHostToDevice, st);
    if (e == hipSuccess) e = sdk::zc_compact(d_seg, d_cnt, d_off, d_symbols, nblocks, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
  }
  (void)hipFree(d);
  if (e != hipSuccess) { set_err("zero-crossing sampler: %s", hipGetErrorString(e)); return -1; }
  return total;
}

// ---- A7 stages: fixed gain, manual carrier offset, matched filter, CMA equalizer ---------------------
SUBOOL suamd_rows_scale(suamd_ctx_t *ctx, const suamd_complex *d_x, suamd_view xv, suamd_complex *d_y, suamd_view yv,
                        unsigned nchan, SUSCOUNT len, SUFLOAT gain, void *stream)
{
  if (!ctx || !d_x || !d_y) { set_err("null argument"); return SU_FALSE; }
  HIP_TRY(sdk::rows_scale(d_x, as_view(xv), d_y, as_view(yv), (int)nchan, (long long)len, gain, as_stream(stream)), SU_FALSE);
  return SU_TRUE;
}

// ---- section 8f #3 --------------------------------------------------------------------------------------
SUBOOL suamd_decision_space(suamd_ctx_t *ctx, const suamd_complex *d_x, SUSCOUNT len, int mode, SUFLOAT *d_out, void *stream)
{
  if (!ctx) { set_err("null context"); return SU_FALSE; }
  if (mode != SUAMD_DECIDER_MODULUS && mode != SUAMD_DECIDER_ARGUMENT) { set_err("bad decision mode %d", mode); return SU_FALSE; }
  if (len == 0) return SU_TRUE;
  if (!d_x || !d_out) { set_err("null argument"); return SU_FALSE; }
  HIP_TRY(sdk::decision_space(d_x, (long long)len, mode, d_out, as_stream(stream)), SU_FALSE);
  return SU_TRUE;
}

SUBOOL suamd_decide(suamd_ctx_t *ctx, const suamd_complex *d_x, SUSCOUNT len, int mode, unsigned bps, SUFLOAT vmin, SUFLOAT vmax,
                    unsigned char *d_sym, void *stream)
{
  if (!ctx) { set_err("null context"); return SU_FALSE; }
  if (mode != SUAMD_DECIDER_MODULUS && mode != SUAMD_DECIDER_ARGUMENT) { set_err("bad decision mode %d", mode); return SU_FALSE; }
  if (bps < 1 || bps > 8 || !(vmax > vmin)) { set_err("bad bits per symbol / range"); return SU_FALSE; }
  if (len == 0) return SU_TRUE;
  if (!d_x || !d_sym) { set_err("null argument"); return SU_FALSE; }
  const int intervals = 1 << bps;
  HIP_TRY(sdk::decide(d_x, (long long)len, mode, intervals, vmin, (vmax - vmin) / (float)intervals, d_sym, as_stream(stream)), SU_FALSE);
  return SU_TRUE;
}

SUBOOL suamd_symbol_histogram(suamd_ctx_t *ctx, const suamd_complex *d_x, SUSCOUNT len, int mode, SUFLOAT vmin, SUFLOAT vmax,
                              unsigned nbins, unsigned *d_hist, void *stream)
{
  if (!ctx) { set_err("null context"); return SU_FALSE; }
  if (mode != SUAMD_DECIDER_MODULUS && mode != SUAMD_DECIDER_ARGUMENT) { set_err("bad decision mode %d", mode); return SU_FALSE; }
  if (nbins < 1 || nbins > 8192 || !(vmax > vmin)) { set_err("bad bin count / range"); return SU_FALSE; }
  if (len == 0) return SU_TRUE;
  if (!d_x || !d_hist) { set_err("null argument"); return SU_FALSE; }
  HIP_TRY(sdk::symbol_histogram(d_x, (long long)len, mode, vmin, (vmax - vmin) / (float)nbins, (int)nbins, d_hist, as_stream(stream)), SU_FALSE);
  return SU_TRUE;
}

struct suamd_snr_estimator { suamd_ctx *ctx; unsigned bps, intervals; float alpha; float *d_state, *d_model; };

suamd_snr_estimator_t *suamd_snr_estimator_new(suamd_ctx_t *ctx, unsigned bps, SUFLOAT alpha)
{
  if (!ctx || bps < 1 || bps > 8) { set_err("bad argument"); return nullptr; }
  HIP_TRY(hipSetDevice(ctx->device), nullptr);
  suamd_snr_estimator *e = new (std::nothrow) suamd_snr_estimator;
  if (!e) { set_err("out of memory"); return nullptr; }
  e->ctx = ctx; e->bps = bps; e->intervals = 1u << bps; e->alpha = alpha;
  e->d_state = dev_from_host(std::vector<float>{1.f / 8.f, 0.f, INFINITY});    // SNR_ESTIMATOR_DEFAULT_SIGMA
  e->d_model = dev_zeros<float>(4096);
  if (!e->d_state || !e->d_model) { set_err("device allocation failed"); suamd_snr_estimator_destroy(e); return nullptr; }
  return e;
}

void suamd_snr_estimator_destroy(suamd_snr_estimator_t *e)
{
  if (!e) return;
  if (e->d_state) hipFree(e->d_state);
  if (e->d_model) hipFree(e->d_model);
  delete e;
}

SUBOOL suamd_snr_estimator_feed(suamd_snr_estimator_t *e, const unsigned *d_history, unsigned length, void *stream)
{
  if (!e || !d_history) { set_err("null argument"); return SU_FALSE; }
  if (length < 1 || length > 4096) { set_err("history length %u unsupported (1..4096)", length); return SU_FALSE; }
  HIP_TRY(sdk::snr_feed(d_history, (int)length, (int)e->intervals, e->alpha, e->d_state, e->d_model, as_stream(stream)), SU_FALSE);
  return SU_TRUE;
}

SUBOOL suamd_snr_estimator_get(suamd_snr_estimator_t *e, SUFLOAT *sigma, SUFLOAT *snr, SUFLOAT *mse_sum, void *stream)
{
  if (!e) { set_err("null argument"); return SU_FALSE; }
  float st[3];
  HIP_TRY(hipMemcpyAsync(st, e->d_state, sizeof st, hipMemcpyDeviceToHost, as_stream(stream)), SU_FALSE);
  HIP_TRY(hipStreamSynchronize(as_stream(stream)), SU_FALSE);
  if (sigma) *sigma = st[0];
  if (snr) *snr = 1.f / (e->intervals * st[0]);              // getSNR()
  if (mse_sum) *mse_sum = st[2];
  return SU_TRUE;
}

SUFLOAT *suamd_snr_estimator_model(suamd_snr_estimator_t *e) { return e ? e->d_model : nullptr; }

static const char *const kSpectsrcNames[] = {"psd", "cyclo", "fmspect", "pmspect", "timediff", "abstimediff", "exp_2", "exp_4", "exp_8"};

unsigned suamd_spectsrc_count(void) { return (unsigned)(sizeof kSpectsrcNames / sizeof kSpectsrcNames[0]); }
const char *suamd_spectsrc_name(unsigned id) { return id >= 1 && id <= suamd_spectsrc_count() ? kSpectsrcNames[id - 1] : nullptr; }

SUBOOL suamd_spectsrc_preproc(suamd_ctx_t *ctx, unsigned id, const suamd_complex *d_x, SUSCOUNT len, SUFLOAT prev_re,
                              SUFLOAT prev_im, suamd_complex *d_y, void *stream)
{
  if (!ctx) { set_err("null context"); return SU_FALSE; }
  if (id < 1 || id > suamd_spectsrc_count()) { set_err("unknown spectrum source %u", id); return SU_FALSE; }
  if (len == 0) return SU_TRUE;
  if (!d_x || !d_y) { set_err("null argument"); return SU_FALSE; }
  if (d_x == d_y && id != 1 && id != 4 && id < 7) { set_err("sources that look at the previous sample cannot run in place"); return SU_FALSE; }
  HIP_TRY(sdk::spectsrc_preproc((int)id, d_x, (long long)len, prev_re, prev_im, nullptr, d_y, as_stream(stream)), SU_FALSE);
  return SU_TRUE;
}

SUBOOL suamd_spectsrc_preproc_from(suamd_ctx_t *ctx, unsigned id, const suamd_complex *d_x, SUSCOUNT len, const suamd_complex *d_prev,
                                   suamd_complex *d_y, void *stream)
{
  if (!ctx) { set_err("null context"); return SU_FALSE; }
  if (id < 1 || id > suamd_spectsrc_count()) { set_err("unknown spectrum source %u", id); return SU_FALSE; }
  if (len == 0) return SU_TRUE;
  if (!d_x || !d_y) { set_err("null argument"); return SU_FALSE; }
  if (d_x == d_y && id != 1 && id != 4 && id < 7) { set_err("sources that look at the previous sample cannot run in place"); return SU_FALSE; }
  HIP_TRY(sdk::spectsrc_preproc((int)id, d_x, (long long)len, 0.0f, 0.0f, d_prev, d_y, as_stream(stream)), SU_FALSE);
  return SU_TRUE;
}

struct suamd_nco_bank { suamd_ctx *ctx; unsigned nchan; uint32_t *d_dphase, *d_phase0; uint64_t n; };

suamd_nco_bank_t *suamd_nco_bank_new(suamd_ctx_t *ctx, unsigned nchan, const double *fnor)
{
  if (!ctx || !fnor || nchan == 0) { set_err("bad argument"); return nullptr; }
  HIP_TRY(hipSetDevice(ctx->device), nullptr);
  suamd_nco_bank *b = new (std::nothrow) suamd_nco_bank;
  if (!b) { set_err("out of memory"); return nullptr; }
  b->ctx = ctx; b->nchan = nchan; b->n = 0;
  std::vector<uint32_t> dp(nchan);
  for (unsigned c = 0; c < nchan; ++c) dp[c] = suamd_fnor_to_dphase(fnor[c]);
  b->d_dphase = dev_from_host(dp);
  b->d_phase0 = dev_zeros<uint32_t>(nchan);
  if (!b->d_dphase || !b->d_phase0) { set_err("device allocation failed"); suamd_nco_bank_destroy(b); return nullptr; }
  return b;
}

void suamd_nco_bank_destroy(suamd_nco_bank_t *b)
{
  if (!b) return;
  if (b->d_dphase) hipFree(b->d_dphase);
  if (b->d_phase0) hipFree(b->d_phase0);
  delete b;
}

SUBOOL suamd_nco_bank_feed(suamd_nco_bank_t *b, const suamd_complex *d_x, suamd_view xv, suamd_complex *d_y, suamd_view yv,
                           SUSCOUNT len, void *stream)
{
  if (!b) { set_err("null argument"); return SU_FALSE; }
  if (len == 0) return SU_TRUE;                                      // an empty block is a no-op
  if (!d_x || !d_y) { set_err("null argument"); return SU_FALSE; }
  HIP_TRY(sdk::rows_xlate(d_x, as_view(xv), d_y, as_view(yv), (int)b->nchan, (long long)len, b->d_dphase, b->d_phase0,
                          b->n, as_stream(stream)), SU_FALSE);
  b->n += len;
  return SU_TRUE;
}

unsigned suamd_rrc_ntaps(double sps) { return 2u * (unsigned)std::ceil(3.0 * sps) + 1u; }

void suamd_rrc_design(SUFLOAT *taps, unsigned ntaps, double sps, double beta)
{
  std::vector<double> d(ntaps);
  double sum = 0;
  for (unsigned i = 0; i < ntaps; ++i) {
    const double t = ((double)i - 0.5 * (double)(ntaps - 1)) / sps;     // in symbols
    double v;
    if (std::fabs(t) < 1e-12) {
      v = 1.0 - beta + 4.0 * beta / kPi;
    } else if (beta > 0 && std::fabs(std::fabs(4.0 * beta * t) - 1.0) < 1e-9) {
      v = beta / std::sqrt(2.0) * ((1.0 + 2.0 / kPi) * std::sin(kPi / (4.0 * beta)) + (1.0 - 2.0 / kPi) * std::cos(kPi / (4.0 * beta)));
    } else {
      const double a = kPi * t;
      v = (std::sin(a * (1.0 - beta)) + 4.0 * beta * t * std::cos(a * (1.0 + beta))) / (a * (1.0 - 16.0 * beta * beta * t * t));
    }
    d[i] = v;
    sum += v;
  }
  for (unsigned i = 0; i < ntaps; ++i) taps[i] = (float)(d[i] / sum);
}

struct suamd_fir_bank { suamd_ctx *ctx; unsigned nchan, ntaps; float *d_taps; suamd_complex *d_hist[2]; int cur; };

suamd_fir_bank_t *suamd_fir_bank_new(suamd_ctx_t *ctx, unsigned nchan, const SUFLOAT *taps, unsigned ntaps)
{
  if (!ctx || !taps || nchan == 0 || ntaps == 0) { set_err("bad argument"); return nullptr; }
  if (ntaps > 8192) { set_err("ntaps %u unsupported (<= 8192)", ntaps); return nullptr; }
  HIP_TRY(hipSetDevice(ctx->device), nullptr);
  suamd_fir_bank *b = new (std::nothrow) suamd_fir_bank;
  if (!b) { set_err("out of memory"); return nullptr; }
  std::memset(b, 0, sizeof *b);
  b->ctx = ctx; b->nchan = nchan; b->ntaps = ntaps;
  b->d_taps = dev_from_host(std::vector<float>(taps, taps + ntaps));
  const size_t hn = (size_t)(ntaps - 1) * nchan;
  b->d_hist[0] = dev_zeros<suamd_complex>(hn);
  b->d_hist[1] = dev_zeros<suamd_complex>(hn);
  if (!b->d_taps || !b->d_hist[0] || !b->d_hist[1]) { set_err("device allocation failed"); suamd_fir_bank_destroy(b); return nullptr; }
  return b;
}

void suamd_fir_bank_destroy(suamd_fir_bank_t *b)
{
  if (!b) return;
  if (b->d_taps) hipFree(b->d_taps);
  if (b->d_hist[0]) hipFree(b->d_hist[0]);
  if (b->d_hist[1]) hipFree(b->d_hist[1]);
  delete b;
}

SUBOOL suamd_fir_bank_feed(suamd_fir_bank_t *b, const suamd_complex *d_x, suamd_view xv, suamd_complex *d_y, suamd_view yv,
                           SUSCOUNT len, void *stream)
{
  if (!b) { set_err("null argument"); return SU_FALSE; }
  if (len == 0) return SU_TRUE;                                      // an empty block is a no-op
  if (!d_x || !d_y) { set_err("null argument"); return SU_FALSE; }
  if (d_x == d_y) { set_err("in-place filtering is not supported"); return SU_FALSE; }
  if (len == 0) return SU_TRUE;
  HIP_TRY(sdk::rows_fir(d_x, as_view(xv), d_y, as_view(yv), (int)b->nchan, (long long)len, b->d_taps, (int)b->ntaps,
                        b->d_hist[b->cur], b->d_hist[b->cur ^ 1], as_stream(stream)), SU_FALSE);
  b->cur ^= 1;
  return SU_TRUE;
}


suamd_cma_bank_t *suamd_cma_bank_new(suamd_ctx_t *ctx, unsigned nchan, unsigned ntaps, SUFLOAT mu)
{
  if (!ctx || nchan == 0) { set_err("bad argument"); return nullptr; }
  if (ntaps < 1 || ntaps > 16) { set_err("equalizer length %u unsupported (1..16)", ntaps); return nullptr; }
  HIP_TRY(hipSetDevice(ctx->device), nullptr);
  suamd_cma_bank *b = new (std::nothrow) suamd_cma_bank;
  if (!b) { set_err("out of memory"); return nullptr; }
  b->ctx = ctx; b->nchan = nchan; b->n = ntaps; b->mu = mu; b->locked = 0;
  std::vector<suamd_complex> w((size_t)ntaps * nchan);
  for (size_t i = 0; i < w.size(); ++i) { w[i].re = i < nchan ? 1.0f : 0.0f; w[i].im = 0.0f; }   // w[0] = 1
  b->d_w = dev_from_host(w);
  b->d_dl = dev_zeros<suamd_complex>((size_t)ntaps * nchan);
  if (!b->d_w || !b->d_dl) { set_err("device allocation failed"); suamd_cma_bank_destroy(b); return nullptr; }
  return b;
}

void suamd_cma_bank_destroy(suamd_cma_bank_t *b)
{
  if (!b) return;
  if (b->d_w) hipFree(b->d_w);
  if (b->d_dl) hipFree(b->d_dl);
  delete b;
}

void suamd_cma_bank_set_locked(suamd_cma_bank_t *b, SUBOOL locked) { if (b) b->locked = locked ? 1 : 0; }
void suamd_cma_bank_set_rate(suamd_cma_bank_t *b, SUFLOAT mu) { if (b) b->mu = mu; }

SUBOOL suamd_cma_bank_feed(suamd_cma_bank_t *b, const suamd_complex *d_x, SUSCOUNT x_stride, const uint32_t *d_count,
                           SUSCOUNT fixed_len, suamd_complex *d_y, SUSCOUNT y_stride, void *stream)
{
  if (!b || !d_x || !d_y) { set_err("null argument"); return SU_FALSE; }
  HIP_TRY(sdk::cma_feed((int)b->n, b->mu, b->locked, b->d_w, b->d_dl, (int)b->nchan, d_x, (long long)x_stride, d_count,
                        (long long)fixed_len, d_y, (long long)y_stride, as_stream(stream)), SU_FALSE);
  return SU_TRUE;
}

SUBOOL suamd_cma_bank_get_weights(suamd_cma_bank_t *b, suamd_complex *weights, void *stream)
{
  if (!b || !weights) { set_err("null argument"); return SU_FALSE; }
  HIP_TRY(hipMemcpyAsync(weights, b->d_w, sizeof(suamd_complex) * (size_t)b->n * b->nchan, hipMemcpyDeviceToHost, as_stream(stream)), SU_FALSE);
  HIP_TRY(hipStreamSynchronize(as_stream(stream)), SU_FALSE);
  return SU_TRUE;
}

// ---- Costas --------------------------------------------------------------------------------------
suamd_costas_bank_t *suamd_costas_bank_new(suamd_ctx_t *ctx, unsigned nchan, int kind, SUFLOAT fhint, SUFLOAT arm_bw,
                                           unsigned arm_order, SUFLOAT loop_bw)
{
  if (!ctx || nchan == 0) { set_err("bad argument"); return nullptr; }
  if (kind < SUAMD_COSTAS_BPSK || kind > SUAMD_COSTAS_8PSK) { set_err("unsupported Costas kind %d", kind); return nullptr; }
  if (arm_order == 0) arm_order = 1;
  if (arm_order - 1 > 4) { set_err("arm_order %u unsupported (<= 5)", arm_order); return nullptr; }
  HIP_TRY(hipSetDevice(ctx->device), nullptr);
  suamd_costas_bank *b = new (std::nothrow) suamd_costas_bank;
  if (!b) { set_err("out of memory"); return nullptr; }
  std::memset(b, 0, sizeof *b);
  b->ctx = ctx; b->nchan = nchan;
  b->p.kind = kind; b->p.order = (int)arm_order - 1;
  b->p.a = (float)(kPi * (double)loop_bw);
  b->p.b = 0.5f * b->p.a * b->p.a;
  b->p.gain = 1.0f;
  butter_lp(b->p.order, (double)arm_bw, b->p.fb, b->p.fa);
  std::vector<float> om(nchan, (float)(kPi * (double)fhint));
  b->s.phase = dev_zeros<uint32_t>(nchan);
  b->s.omega = dev_from_host(om);
  b->s.xh = dev_zeros<float>(8 * (size_t)nchan);
  b->s.yh = dev_zeros<float>(8 * (size_t)nchan);
  if (!b->s.phase || !b->s.omega || !b->s.xh || !b->s.yh) {
    set_err("device allocation failed");
    suamd_costas_bank_destroy(b);
    return nullptr;
  }
  return b;
}

void suamd_costas_bank_destroy(suamd_costas_bank_t *b)
{
  if (!b) return;
  if (b->s.phase) hipFree(b->s.phase);
  if (b->s.omega) hipFree(b->s.omega);
  if (b->s.xh) hipFree(b->s.xh);
  if (b->s.yh) hipFree(b->s.yh);
  delete b;
}

SUBOOL suamd_costas_bank_feed(suamd_costas_bank_t *b, const suamd_complex *d_x, suamd_view xv, suamd_complex *d_y,
                              suamd_view yv, SUSCOUNT len, void *stream)
{
  if (!b) { set_err("null argument"); return SU_FALSE; }
  if (len == 0) return SU_TRUE;                                      // an empty block is a no-op
  if (!d_x || !d_y) { set_err("null argument"); return SU_FALSE; }
  HIP_TRY(sdk::costas_feed(b->p, b->s, (int)b->nchan, d_x, as_view(xv), d_y, as_view(yv),
                           (long long)len, as_stream(stream)), SU_FALSE);
  return SU_TRUE;
}

SUBOOL suamd_costas_bank_get_state(suamd_costas_bank_t *b, SUFLOAT *omega, uint32_t *phase, void *stream)
{
  if (!b) { set_err("null argument"); return SU_FALSE; }
  HIP_TRY(hipStreamSynchronize(as_stream(stream)), SU_FALSE);
  if (omega) HIP_TRY(hipMemcpy(omega, b->s.omega, sizeof(float) * b->nchan, hipMemcpyDeviceToHost), SU_FALSE);
  if (phase) HIP_TRY(hipMemcpy(phase, b->s.phase, sizeof(uint32_t) * b->nchan, hipMemcpyDeviceToHost), SU_FALSE);
  return SU_TRUE;
}

// ---- PLL -----------------------------------------------------------------------------------------
suamd_pll_bank_t *suamd_pll_bank_new(suamd_ctx_t *ctx, unsigned nchan, SUFLOAT fhint, SUFLOAT fc)
{
  if (!ctx || nchan == 0) { set_err("bad argument"); return nullptr; }
  HIP_TRY(hipSetDevice(ctx->device), nullptr);
  suamd_pll_bank *b = new (std::nothrow) suamd_pll_bank;
  if (!b) { set_err("out of memory"); return nullptr; }
  std::memset(b, 0, sizeof *b);
  b->ctx = ctx; b->nchan = nchan;
  const double w = kPi * (double)fc;
  const double dinv = 1.0 / (1.0 + 2.0 * 0.707 * w + w * w);
  b->alpha = (float)(4.0 * w * w * dinv);
  b->beta  = (float)(4.0 * 0.707 * w * dinv);
  std::vector<float> om(nchan, (float)(kPi * (double)fhint));
  b->s.phase = dev_zeros<uint32_t>(nchan);
  b->s.omega = dev_from_host(om);
  if (!b->s.phase || !b->s.omega) { set_err("device allocation failed"); suamd_pll_bank_destroy(b); return nullptr; }
  return b;
}

void suamd_pll_bank_destroy(suamd_pll_bank_t *b)
{
  if (!b) return;
  if (b->s.phase) hipFree(b->s.phase);
  if (b->s.omega) hipFree(b->s.omega);
  delete b;
}

SUBOOL suamd_pll_bank_feed(suamd_pll_bank_t *b, const suamd_complex *d_x, suamd_view xv, suamd_complex *d_y,
                           suamd_view yv, SUSCOUNT len, void *stream)
{
  if (!b) { set_err("null argument"); return SU_FALSE; }
  if (len == 0) return SU_TRUE;                                      // an empty block is a no-op
  if (!d_x || !d_y) { set_err("null argument"); return SU_FALSE; }
  HIP_TRY(sdk::pll_feed(b->alpha, b->beta, b->s, (int)b->nchan, d_x, as_view(xv), d_y, as_view(yv),
                        (long long)len, as_stream(stream)), SU_FALSE);
  return SU_TRUE;
}

SUBOOL suamd_pll_bank_get_state(suamd_pll_bank_t *b, SUFLOAT *omega, uint32_t *phase, void *stream)
{
  if (!b) { set_err("null argument"); return SU_FALSE; }
  HIP_TRY(hipStreamSynchronize(as_stream(stream)), SU_FALSE);
  if (omega) HIP_TRY(hipMemcpy(omega, b->s.omega, sizeof(float) * b->nchan, hipMemcpyDeviceToHost), SU_FALSE);
  if (phase) HIP_TRY(hipMemcpy(phase, b->s.phase, sizeof(uint32_t) * b->nchan, hipMemcpyDeviceToHost), SU_FALSE);
  return SU_TRUE;
}

// ---- clock recovery --------------------------------------------------------------------------------
suamd_clock_bank_t *suamd_clock_bank_new(suamd_ctx_t *ctx, unsigned nchan, SUFLOAT loop_gain, SUFLOAT bhint)
{
  if (!ctx || nchan == 0) { set_err("bad argument"); return nullptr; }
  if (!(bhint > 0.0f)) { set_err("bhint must be > 0"); return nullptr; }
  HIP_TRY(hipSetDevice(ctx->device), nullptr);
  suamd_clock_bank *b = new (std::nothrow) suamd_clock_bank;
  if (!b) { set_err("out of memory"); return nullptr; }
  std::memset(b, 0, sizeof *b);
  b->ctx = ctx; b->nchan = nchan;
  b->p.alpha = 2e-1f; b->p.beta = 1.2e-4f; b->p.gain = loop_gain;
  b->p.bmin = 0.5f * bhint;
  b->p.bmax = bhint > 0.5f ? 1.0f : 2.0f * bhint;
  std::vector<float> phi(nchan, 0.25f), bn(nchan, bhint);
  b->s.phi = dev_from_host(phi);
  b->s.bnor = dev_from_host(bn);
  b->s.halfcycle = dev_zeros<int>(nchan);
  b->s.prev = dev_zeros<float>(2 * (size_t)nchan);
  b->s.x0 = dev_zeros<float>(2 * (size_t)nchan);
  b->s.x1 = dev_zeros<float>(2 * (size_t)nchan);
  b->s.x2 = dev_zeros<float>(2 * (size_t)nchan);
  if (!b->s.phi || !b->s.bnor || !b->s.halfcycle || !b->s.prev || !b->s.x0 || !b->s.x1 || !b->s.x2) {
    set_err("device allocation failed");
    suamd_clock_bank_destroy(b);
    return nullptr;
  }
  return b;
}

SUBOOL suamd_clock_bank_set_phase(suamd_clock_bank_t *b, SUFLOAT phi, void *stream)
{
  if (!b) { set_err("null argument"); return SU_FALSE; }
  std::vector<float> v(b->nchan, phi);
  HIP_TRY(hipMemcpyAsync(b->s.phi, v.data(), sizeof(float) * b->nchan, hipMemcpyHostToDevice, as_stream(stream)), SU_FALSE);
  HIP_TRY(hipStreamSynchronize(as_stream(stream)), SU_FALSE);
  return SU_TRUE;
}

void suamd_clock_bank_destroy(suamd_clock_bank_t *b)
{
  if (!b) return;
  if (b->s.phi) hipFree(b->s.phi);
  if (b->s.bnor) hipFree(b->s.bnor);
  if (b->s.halfcycle) hipFree(b->s.halfcycle);
  if (b->s.prev) hipFree(b->s.prev);
  if (b->s.x0) hipFree(b->s.x0);
  if (b->s.x1) hipFree(b->s.x1);
  if (b->s.x2) hipFree(b->s.x2);
  delete b;
}

SUBOOL suamd_clock_bank_feed(suamd_clock_bank_t *b, const suamd_complex *d_x, suamd_view xv, SUSCOUNT len,
                             suamd_complex *d_sym, SUSCOUNT sym_stride, uint32_t *d_count, void *stream)
{
  if (!b) { set_err("null argument"); return SU_FALSE; }
  if (len == 0) return SU_TRUE;                                      // an empty block is a no-op
  if (!d_x || !d_sym || !d_count) { set_err("null argument"); return SU_FALSE; }
  HIP_TRY(sdk::clock_feed(b->p, b->s, (int)b->nchan, d_x, as_view(xv), (long long)len, d_sym,
                          (long long)sym_stride, d_count, as_stream(stream)), SU_FALSE);
  return SU_TRUE;
}

SUBOOL suamd_clock_bank_get_state(suamd_clock_bank_t *b, SUFLOAT *bnor, SUFLOAT *phi, void *stream)
{
  if (!b) { set_err("null argument"); return SU_FALSE; }
  HIP_TRY(hipStreamSynchronize(as_stream(stream)), SU_FALSE);
  if (bnor) HIP_TRY(hipMemcpy(bnor, b->s.bnor, sizeof(float) * b->nchan, hipMemcpyDeviceToHost), SU_FALSE);
  if (phi)  HIP_TRY(hipMemcpy(phi, b->s.phi, sizeof(float) * b->nchan, hipMemcpyDeviceToHost), SU_FALSE);
  return SU_TRUE;
}

// ---- AGC -------------------------------------------------------------------------------------------
void suamd_agc_params_from_tau(struct suamd_agc_params *p, SUFLOAT tau)
{
  const struct suamd_agc_params def = suamd_agc_params_INITIALIZER;
  const double rise = 2 * 3.9062e-1;
  *p = def;
  p->fast_rise_t = (float)(tau * rise);
  p->fast_fall_t = (float)(tau * 2 * rise);
  p->slow_rise_t = (float)(tau * 10 * rise);
  p->slow_fall_t = (float)(tau * 10 * 2 * rise);
  p->hang_max    = (unsigned)(tau * rise * 5);
}

suamd_agc_bank_t *suamd_agc_bank_new(suamd_ctx_t *ctx, unsigned nchan, const struct suamd_agc_params *pp)
{
  if (!ctx || nchan == 0 || !pp) { set_err("bad argument"); return nullptr; }
  if (pp->delay_line_size == 0 || pp->delay_line_size > 64 || pp->mag_history_size == 0 || pp->mag_history_size > 64) {
    set_err("delay_line_size / mag_history_size must be in 1..64");
    return nullptr;
  }
  HIP_TRY(hipSetDevice(ctx->device), nullptr);
  suamd_agc_bank *b = new (std::nothrow) suamd_agc_bank();
  if (!b) { set_err("out of memory"); return nullptr; }
  b->ctx = ctx; b->nchan = nchan; b->n_fed = 0;
  b->p.knee = pp->threshold;
  b->p.gain_slope = pp->slope_factor * 1e-2f;
  b->p.hang_max = pp->hang_max;
  b->p.delay_line_size = pp->delay_line_size;
  b->p.mag_history_size = pp->mag_history_size;
  b->p.fast_alpha_rise = (float)(1.0 - std::exp(-1.0 / (double)pp->fast_rise_t));
  b->p.fast_alpha_fall = (float)(1.0 - std::exp(-1.0 / (double)pp->fast_fall_t));
  b->p.slow_alpha_rise = (float)(1.0 - std::exp(-1.0 / (double)pp->slow_rise_t));
  b->p.slow_alpha_fall = (float)(1.0 - std::exp(-1.0 / (double)pp->slow_fall_t));
  b->s.delay_line = dev_zeros<float>(64 * 2 * (size_t)nchan);
  b->s.mag_history = dev_zeros<float>(64 * (size_t)nchan);
  b->s.hang_n = dev_zeros<unsigned>(nchan);
  b->s.fast_level = dev_zeros<float>(nchan);
  b->s.slow_level = dev_zeros<float>(nchan);
  if (!b->s.delay_line || !b->s.mag_history || !b->s.hang_n || !b->s.fast_level || !b->s.slow_level) {
    set_err("device allocation failed");
    suamd_agc_bank_destroy(b);
    return nullptr;
  }
  return b;
}

void suamd_agc_bank_destroy(suamd_agc_bank_t *b)
{
  if (!b) return;
  if (b->s.delay_line) hipFree(b->s.delay_line);
  if (b->s.mag_history) hipFree(b->s.mag_history);
  if (b->s.hang_n) hipFree(b->s.hang_n);
  if (b->s.fast_level) hipFree(b->s.fast_level);
  if (b->s.slow_level) hipFree(b->s.slow_level);
  b->scratch.release();
  for (hipEvent_t e : b->ev) if (e) (void)hipEventDestroy(e);
  delete b;
}

SUBOOL suamd_agc_bank_feed(suamd_agc_bank_t *b, const suamd_complex *d_x, suamd_view xv, suamd_complex *d_y,
                           suamd_view yv, SUSCOUNT len, void *stream)
{
  if (!b) { set_err("null argument"); return SU_FALSE; }
  if (len == 0) return SU_TRUE;                                      // an empty block is a no-op
  if (!d_x || !d_y) { set_err("null argument"); return SU_FALSE; }
  if (d_x == d_y) { set_err("the AGC bank cannot run in place (the output is the input delayed)"); return SU_FALSE; }
  if (len == 0) return SU_TRUE;
  if (!b->scratch.reserve(2 * sizeof(float) * (size_t)len * b->nchan)) { set_err("scratch allocation failed"); return SU_FALSE; }
  HIP_TRY(sdk::agc_feed(b->p, b->s, (int)b->nchan, d_x, as_view(xv), d_y, as_view(yv), (long long)len,
                        static_cast<float *>(b->scratch.p), as_stream(stream)), SU_FALSE);
  b->n_fed += len;
  return SU_TRUE;
}

SUBOOL suamd_agc_bank_feed_split(suamd_agc_bank_t *b, const suamd_complex *d_x, suamd_view xv, suamd_complex *d_y,
                                 suamd_view yv, SUSCOUNT len, void *stream_level, void *stream_wide)
{
  if (!b) { set_err("null argument"); return SU_FALSE; }
  if (len == 0) return SU_TRUE;
  if (!d_x || !d_y) { set_err("null argument"); return SU_FALSE; }
  if (d_x == d_y) { set_err("the AGC bank cannot run in place (the output is the input delayed)"); return SU_FALSE; }
  hipStream_t sl = as_stream(stream_level), sw = as_stream(stream_wide);
  if (sl == sw) return suamd_agc_bank_feed(b, d_x, xv, d_y, yv, len, stream_wide);
  if (!b->scratch.reserve(2 * sizeof(float) * (size_t)len * b->nchan)) { set_err("scratch allocation failed"); return SU_FALSE; }
  for (hipEvent_t &e : b->ev)
    if (!e) HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming), SU_FALSE);
  float *scratch = static_cast<float *>(b->scratch.p);
  // wide: magnitudes + sliding maximum -> level: the trackers -> wide: gain on the delayed input, state carry.  Both hops
  // are device-side waits; the scratch needs no more: the next call's first step sits behind this call's last on `wide`
  HIP_TRY(sdk::agc_feed_pre(b->p, b->s, (int)b->nchan, d_x, as_view(xv), (long long)len, scratch, sw), SU_FALSE);
  HIP_TRY(hipEventRecord(b->ev[0], sw), SU_FALSE);
  HIP_TRY(hipStreamWaitEvent(sl, b->ev[0], 0), SU_FALSE);
  HIP_TRY(sdk::agc_feed_level(b->p, b->s, (int)b->nchan, (long long)len, scratch, sl), SU_FALSE);
  HIP_TRY(hipEventRecord(b->ev[1], sl), SU_FALSE);
  HIP_TRY(hipStreamWaitEvent(sw, b->ev[1], 0), SU_FALSE);
  HIP_TRY(sdk::agc_feed_post(b->p, b->s, (int)b->nchan, d_x, as_view(xv), d_y, as_view(yv), (long long)len, scratch, sw), SU_FALSE);
  b->n_fed += len;
  return SU_TRUE;
}

// ---- whole-capture FFT tasks ------------------------------------------------------------------------
SUBOOL suamd_fft_forward_bulk(suamd_ctx_t *ctx, const suamd_complex *d_in, suamd_complex *d_out, suamd_complex *d_work,
                              unsigned log2n, void *stream)
{
  if (!ctx || !d_in || !d_out || !d_work) { set_err("null argument"); return SU_FALSE; }
  if (log2n < 4 || log2n > 24) { set_err("log2n %u unsupported (4..24)", log2n); return SU_FALSE; }
  hipStream_t st = as_stream(stream);
  const size_t bytes = sizeof(suamd_complex) << log2n;
  // arrange the ping-pong so that the last pass lands in d_out
  const int npass = ((int)log2n + 3) / 4;
  void *a = (npass & 1) ? (void *)d_work : (void *)d_out;
  void *b = (npass & 1) ? (void *)d_out : (void *)d_work;
  HIP_TRY(hipMemcpyAsync(a, d_in, bytes, hipMemcpyDeviceToDevice, st), SU_FALSE);
  void *res = nullptr;
  HIP_TRY(sdk::fft_forward(a, b, (int)log2n, &res, st), SU_FALSE);
  if (res != (void *)d_out) HIP_TRY(hipMemcpyAsync(d_out, res, bytes, hipMemcpyDeviceToDevice, st), SU_FALSE);
  return SU_TRUE;
}

// ---- FAC (section 8f #4) ------------------------------------------------------------------------------
struct suamd_fac { suamd_ctx *ctx; unsigned log2n; float alpha; void *a, *b; float *d_abs, *d_fac; unsigned *d_mx, *d_mn; };

static SUBOOL fac_reset_state(suamd_fac *f, hipStream_t st)
{
  const unsigned init[2] = {0u, 0x7f800000u};               // max: nothing seen; min: +inf
  HIP_TRY(hipMemsetAsync(f->d_fac, 0, sizeof(float) << (f->log2n - 1), st), SU_FALSE);
  HIP_TRY(hipMemcpyAsync(f->d_mx, &init[0], 4, hipMemcpyHostToDevice, st), SU_FALSE);
  HIP_TRY(hipMemcpyAsync(f->d_mn, &init[1], 4, hipMemcpyHostToDevice, st), SU_FALSE);
  HIP_TRY(hipStreamSynchronize(st), SU_FALSE);
  return SU_TRUE;
}

suamd_fac_t *suamd_fac_new(suamd_ctx_t *ctx, unsigned size, SUFLOAT alpha)
{
  if (!ctx) { set_err("null context"); return nullptr; }
  unsigned l2 = 0;
  while ((1u << l2) < size && l2 < 31) ++l2;
  if ((1u << l2) != size || l2 < 4 || l2 > 24) { set_err("FAC size must be a power of two, 16..2^24"); return nullptr; }
  HIP_TRY(hipSetDevice(ctx->device), nullptr);
  suamd_fac *f = new (std::nothrow) suamd_fac;
  if (!f) { set_err("out of memory"); return nullptr; }
  std::memset(f, 0, sizeof *f);
  f->ctx = ctx; f->log2n = l2; f->alpha = alpha;
  const size_t n = (size_t)1 << l2;
  bool ok = hipMalloc(&f->a, n * 8) == hipSuccess && hipMalloc(&f->b, n * 8) == hipSuccess &&
            hipMalloc((void **)&f->d_abs, n * 2) == hipSuccess && hipMalloc((void **)&f->d_fac, n * 2) == hipSuccess &&
            hipMalloc((void **)&f->d_mx, 4) == hipSuccess && hipMalloc((void **)&f->d_mn, 4) == hipSuccess;
  if (!ok || !fac_reset_state(f, nullptr)) { set_err("device allocation failed"); suamd_fac_destroy(f); return nullptr; }
  return f;
}

void suamd_fac_destroy(suamd_fac_t *f)
{
  if (!f) return;
  for (void *p : {f->a, f->b, (void *)f->d_abs, (void *)f->d_fac, (void *)f->d_mx, (void *)f->d_mn}) if (p) (void)hipFree(p);
  delete f;
}

void suamd_fac_set_alpha(suamd_fac_t *f, SUFLOAT alpha) { if (f) f->alpha = alpha; }
SUBOOL suamd_fac_reset(suamd_fac_t *f, void *stream) { if (!f) { set_err("null argument"); return SU_FALSE; } return fac_reset_state(f, as_stream(stream)); }
SUFLOAT *suamd_fac_array(suamd_fac_t *f) { return f ? f->d_fac : nullptr; }

SUBOOL suamd_fac_feed(suamd_fac_t *f, const suamd_complex *d_data, SUSCOUNT nbuffers, SUSDIFF view_start, SUSDIFF view_end,
                      void *stream)
{
  if (!f || !d_data) { set_err("null argument"); return SU_FALSE; }
  hipStream_t st = as_stream(stream);
  const size_t n = (size_t)1 << f->log2n;
  for (SUSCOUNT k = 0; k < nbuffers; ++k) {                  // buffers are sequential: the EMA and the maximum carry over
    HIP_TRY(hipMemcpyAsync(f->a, d_data + k * n, n * 8, hipMemcpyDeviceToDevice, st), SU_FALSE);
    HIP_TRY(sdk::fac_feed(f->a, f->b, (int)f->log2n, f->alpha, (long long)view_start, (long long)view_end, f->d_abs, f->d_fac,
                          f->d_mx, f->d_mn, st), SU_FALSE);
  }
  return SU_TRUE;
}

SUBOOL suamd_fac_get_range(suamd_fac_t *f, SUFLOAT *min, SUFLOAT *max, void *stream)
{
  if (!f) { set_err("null argument"); return SU_FALSE; }
  unsigned v[2];
  HIP_TRY(hipMemcpyAsync(&v[0], f->d_mx, 4, hipMemcpyDeviceToHost, as_stream(stream)), SU_FALSE);
  HIP_TRY(hipMemcpyAsync(&v[1], f->d_mn, 4, hipMemcpyDeviceToHost, as_stream(stream)), SU_FALSE);
  HIP_TRY(hipStreamSynchronize(as_stream(stream)), SU_FALSE);
  float mx, mn;
  std::memcpy(&mx, &v[0], 4); std::memcpy(&mn, &v[1], 4);
  if (v[0] == 0u) mx = -INFINITY;
  if (max) *max = mx;
  if (min) *min = mn;
  return SU_TRUE;
}

namespace {
struct CaptureFft {                      // scratch of one whole-capture task
  void *a = nullptr, *b = nullptr, *res = nullptr;
  float *blk_max = nullptr; long long *blk_idx = nullptr; double *d_res = nullptr;
  long long alloc = 1; int log2n = 0;
  static constexpr int NBLK = 512;
  bool init(SUSCOUNT len)
  {
    while ((SUSCOUNT)alloc < len) { alloc <<= 1; ++log2n; }
    if (log2n < 4) { alloc = 16; log2n = 4; }
    const size_t bytes = sizeof(suamd_complex) * (size_t)alloc;
    return hipMalloc(&a, bytes) == hipSuccess && hipMalloc(&b, bytes) == hipSuccess &&
           hipMalloc((void **)&blk_max, NBLK * sizeof(float)) == hipSuccess &&
           hipMalloc((void **)&blk_idx, NBLK * sizeof(long long)) == hipSuccess &&
           hipMalloc((void **)&d_res, 8 * sizeof(double)) == hipSuccess;
  }
  ~CaptureFft()
  {
    for (void *p : {a, b, (void *)blk_max, (void *)blk_idx, (void *)d_res}) if (p) (void)hipFree(p);
  }
};
}  // namespace

// ---- "power" inspector class ----------------------------------------------------------------------------------
struct suamd_power_bank { suamd_ctx *ctx; uint64_t N, cnt; double *d_acc = nullptr; int cur = 0; };

suamd_power_bank_t *suamd_power_bank_new(suamd_ctx_t *ctx, SUSCOUNT integrate_samples)
{
  if (!ctx) { set_err("null context"); return nullptr; }
  if (integrate_samples == 0) { set_err("integrate_samples must be > 0"); return nullptr; }
  auto *b = new (std::nothrow) suamd_power_bank;
  if (!b) { set_err("out of memory"); return nullptr; }
  b->ctx = ctx; b->N = integrate_samples; b->cnt = 0;
  if (hipMalloc((void **)&b->d_acc, 2 * sizeof(double)) != hipSuccess || hipMemset(b->d_acc, 0, 2 * sizeof(double)) != hipSuccess ||
      hipStreamSynchronize(nullptr) != hipSuccess) {
    set_err("device allocation failed"); suamd_power_bank_destroy(b); return nullptr;
  }
  return b;
}

void suamd_power_bank_destroy(suamd_power_bank_t *b)
{
  if (!b) return;
  if (b->d_acc) (void)hipFree(b->d_acc);
  delete b;
}

SUBOOL suamd_power_bank_set_integrate(suamd_power_bank_t *b, SUSCOUNT integrate_samples, void *stream)
{
  if (!b || integrate_samples == 0) { set_err("bad argument"); return SU_FALSE; }
  b->N = integrate_samples; b->cnt = 0;                      // RMSInspector::updateMaxSamples -> checkMaxSamples: start over
  HIP_TRY(hipMemsetAsync(b->d_acc, 0, 2 * sizeof(double), as_stream(stream)), SU_FALSE);
  return SU_TRUE;
}

SUSCOUNT suamd_power_bank_output_count(const suamd_power_bank_t *b, SUSCOUNT len) { return b ? (b->cnt + len) / b->N : 0; }

SUBOOL suamd_power_bank_feed(suamd_power_bank_t *b, const suamd_complex *d_x, SUSCOUNT len, suamd_complex *d_out, SUSCOUNT *n_out,
                             void *stream)
{
  if (!b) { set_err("null argument"); return SU_FALSE; }
  if (n_out) *n_out = 0;
  if (len == 0) return SU_TRUE;
  const uint64_t K = (b->cnt + len) / b->N;
  if (!d_x || (K && !d_out)) { set_err("null argument"); return SU_FALSE; }
  HIP_TRY(sdk::power_integrate(d_x, (long long)len, (long long)b->N, (long long)b->cnt, b->d_acc + b->cur, b->d_acc + (b->cur ^ 1), d_out,
                               as_stream(stream)), SU_FALSE);
  b->cur ^= 1;
  b->cnt = (b->cnt + len) % b->N;
  if (n_out) *n_out = K;
  return SU_TRUE;
}

// ---- baud estimators (SPEC.md section M) ---------------------------------------------
struct suamd_baud_estimator {
  suamd_ctx *ctx; int kind; unsigned n;
  CaptureFft w;                          // nonlinear: window, transform, arg-max + centroid scratch
  void *y = nullptr;                     // nonlinear: the transformed block
  suamd_fac *fac = nullptr;              // fac: the running autocorrelation
  float *d_val = nullptr;                // fac: {lag, 1 / lag}; nonlinear: {centroid / n}
  struct Landing { float value; int fed; } *pin = nullptr;   // host landing zone of the plain feed()
};

suamd_baud_estimator_t *suamd_baud_estimator_new(suamd_ctx_t *ctx, int kind, unsigned size)
{
  if (!ctx) { set_err("null context"); return nullptr; }
  if (kind != SUAMD_BAUD_ESTIMATOR_FAC && kind != SUAMD_BAUD_ESTIMATOR_NONLINEAR && kind != SUAMD_ESTIMATOR_CARRIER) { set_err("unknown estimator kind %d", kind); return nullptr; }
  if (size < 512 || size > (1u << 20) || (size & (size - 1))) { set_err("size %u unsupported (power of two, 512..1048576)", size); return nullptr; }
  auto *e = new (std::nothrow) suamd_baud_estimator;
  if (!e) { set_err("out of memory"); return nullptr; }
  e->ctx = ctx; e->kind = kind; e->n = size;
  bool ok = hipHostMalloc((void **)&e->pin, sizeof *e->pin, hipHostMallocMapped) == hipSuccess;
  if (ok) std::memset(e->pin, 0, sizeof *e->pin);
  if (ok && kind == SUAMD_BAUD_ESTIMATOR_NONLINEAR) ok = e->w.init(size) && hipMalloc(&e->y, sizeof(suamd_complex) * (size_t)size) == hipSuccess;
  if (ok && kind == SUAMD_ESTIMATOR_CARRIER) ok = e->w.init(size);
  if (ok && kind == SUAMD_BAUD_ESTIMATOR_FAC) {
    e->fac = suamd_fac_new(ctx, size, 0.25f);
    ok = e->fac != nullptr;
  }
  ok = ok && hipMalloc((void **)&e->d_val, 2 * sizeof(float)) == hipSuccess;
  if (!ok) { if (g_err.empty()) set_err("device allocation failed"); suamd_baud_estimator_destroy(e); return nullptr; }
  return e;
}

void suamd_baud_estimator_destroy(suamd_baud_estimator_t *e)
{
  if (!e) return;
  if (e->fac) suamd_fac_destroy(e->fac);
  if (e->y) (void)hipFree(e->y);
  if (e->d_val) (void)hipFree(e->d_val);
  if (e->pin) (void)hipHostFree(e->pin);
  delete e;
}

unsigned suamd_baud_estimator_size(const suamd_baud_estimator_t *e) { return e ? e->n : 0; }

SUBOOL suamd_baud_estimator_feed(suamd_baud_estimator_t *e, const suamd_complex *d_x, SUSCOUNT len, void *stream)
{
  if (!e) { set_err("null argument"); return SU_FALSE; }
  if (len < e->n) return SU_TRUE;                            // not a whole analysis window: the estimate stands
  if (!suamd_baud_estimator_feed_to(e, d_x, len, &e->pin->value, stream)) return SU_FALSE;
  e->pin->fed = 1;
  return SU_TRUE;
}

SUBOOL suamd_baud_estimator_feed_to(suamd_baud_estimator_t *e, const suamd_complex *d_x, SUSCOUNT len, SUFLOAT *h_value, void *stream)
{
  if (!e || !h_value) { set_err("null argument"); return SU_FALSE; }
  if (len < e->n) { set_err("block shorter than the analysis window (%llu < %u)", (unsigned long long)len, e->n); return SU_FALSE; }
  if (!d_x) { set_err("null argument"); return SU_FALSE; }
  hipStream_t st = as_stream(stream);
  const long long n = e->n;
  if (e->kind == SUAMD_BAUD_ESTIMATOR_NONLINEAR) {
    HIP_TRY(sdk::baud_nl_transform(d_x, n, e->y, st), SU_FALSE);
    HIP_TRY(sdk::window_pad(e->y, n, e->w.alloc, e->w.a, st), SU_FALSE);
    HIP_TRY(sdk::fft_forward(e->w.a, e->w.b, e->w.log2n, &e->w.res, st), SU_FALSE);
    // the lowest strong line outside the DC notch (1 % of the band), power centroid over 9 bins
    const int skip = std::max(4, (int)(0.01 * (double)n));
    HIP_TRY(sdk::baud_line(e->w.res, (int)n, skip, e->w.d_res, e->d_val, st), SU_FALSE);
    HIP_TRY(hipMemcpyAsync(h_value, e->d_val, sizeof(float), hipMemcpyDeviceToHost, st), SU_FALSE);
  } else if (e->kind == SUAMD_ESTIMATOR_CARRIER) {
    // Tasks/CarrierDetector.cpp:99-137 on the window's samples -- Blackman-Harris, transform, the strongest bin, the power-
    // weighted phasor sum over the `bins` around it -- with avgRelBw = 1/2 (half the channel around the peak: a modulated
    // carrier's whole main lobe, not the data-dependent crest inside it) and no DC notch (the channel IS the baseband)
    HIP_TRY(sdk::window_pad(d_x, n, e->w.alloc, e->w.a, st), SU_FALSE);
    HIP_TRY(sdk::fft_forward(e->w.a, e->w.b, e->w.log2n, &e->w.res, st), SU_FALSE);
    const int bins = static_cast<int>((double)e->w.alloc * 0.5) + 1, delta = (bins - 1) / 2;
    HIP_TRY(sdk::spectrum_centroid(e->w.res, e->w.alloc, 0, e->w.alloc, nullptr, bins, delta, 0, e->w.blk_max, e->w.blk_idx,
                                   CaptureFft::NBLK, reinterpret_cast<float *>(e->w.d_res), st), SU_FALSE);
    HIP_TRY(sdk::carrier_norm(reinterpret_cast<const float *>(e->w.d_res), e->d_val, st), SU_FALSE);
    HIP_TRY(hipMemcpyAsync(h_value, e->d_val, sizeof(float), hipMemcpyDeviceToHost, st), SU_FALSE);
  } else {
    if (!suamd_fac_feed(e->fac, d_x, 1, 0, (SUSDIFF)(n / 2), st)) return SU_FALSE;
    HIP_TRY(sdk::fac_first_valley(suamd_fac_array(e->fac), (int)(n / 2), e->d_val, st), SU_FALSE);
    HIP_TRY(hipMemcpyAsync(h_value, e->d_val + 1, sizeof(float), hipMemcpyDeviceToHost, st), SU_FALSE);
  }
  return SU_TRUE;
}

SUFLOAT suamd_baud_estimator_get(const suamd_baud_estimator_t *e)
{
  return e && e->pin->fed ? e->pin->value : 0.0f;
}

SUBOOL suamd_carrier_detect(suamd_ctx_t *ctx, const suamd_complex *d_data, SUSCOUNT len, SUFLOAT avgRelBw,
                            SUFLOAT dcNotchRelBw, SUFLOAT *peak, void *stream)
{
  if (!ctx || !d_data || !peak || len < 2) { set_err("bad argument"); return SU_FALSE; }
  if (len > (1ull << 24)) { set_err("capture longer than 2^24 samples"); return SU_FALSE; }
  hipStream_t st = as_stream(stream);
  CaptureFft w;
  if (!w.init(len)) { set_err("device allocation failed"); return SU_FALSE; }
  HIP_TRY(sdk::window_pad(d_data, (long long)len, w.alloc, w.a, st), SU_FALSE);
  HIP_TRY(sdk::fft_forward(w.a, w.b, w.log2n, &w.res, st), SU_FALSE);
  // Tasks/CarrierDetector.cpp:99-104
  const int bins = static_cast<int>((double)w.alloc * (double)avgRelBw) + 1;
  const int delta = (bins - 1) / 2;
  const int skipLen = static_cast<int>(.5 * (double)dcNotchRelBw * (double)w.alloc);
  HIP_TRY(sdk::spectrum_centroid(w.res, w.alloc, skipLen, w.alloc - skipLen, nullptr, bins, delta, 0, w.blk_max,
                                 w.blk_idx, CaptureFft::NBLK, reinterpret_cast<float *>(w.d_res), st), SU_FALSE);
  float res[5];                                                        // binary32, as the reference's SUCOMPLEX acc
  HIP_TRY(hipMemcpyAsync(res, w.d_res, sizeof res, hipMemcpyDeviceToHost, st), SU_FALSE);
  HIP_TRY(hipStreamSynchronize(st), SU_FALSE);
  float p = std::atan2(res[1], res[0]);                                // SU_C_ARG (:134)
  if (p > (float)M_PI) p -= (float)(2 * M_PI);
  *peak = p;
  return SU_TRUE;
}

SUSCOUNT suamd_doppler_alloc_size(SUSCOUNT len)
{
  SUSCOUNT a = 1;
  while (a < len) a <<= 1;
  return a < 16 ? 16 : a;
}

SUBOOL suamd_doppler_calc(suamd_ctx_t *ctx, const suamd_complex *d_data, SUSCOUNT len, SUFLOAT fs, SUFREQ f0,
                          SUFLOAT *d_spectrum, SUFLOAT *peak, SUFLOAT *sigma, SUFLOAT *max, void *stream)
{
  if (!ctx || !d_data || len < 2) { set_err("bad argument"); return SU_FALSE; }
  if (len > (1ull << 24)) { set_err("capture longer than 2^24 samples"); return SU_FALSE; }
  hipStream_t st = as_stream(stream);
  CaptureFft w;
  if (!w.init(len)) { set_err("device allocation failed"); return SU_FALSE; }
  HIP_TRY(sdk::window_pad(d_data, (long long)len, w.alloc, w.a, st), SU_FALSE);
  HIP_TRY(sdk::fft_forward(w.a, w.b, w.log2n, &w.res, st), SU_FALSE);
  const long long bins = w.alloc, delta = bins / 2;                   // DopplerCalculator.cpp:107-108
  HIP_TRY(sdk::spectrum_centroid(w.res, w.alloc, 0, w.alloc, d_spectrum, bins, delta, 1, w.blk_max, w.blk_idx,
                                 CaptureFft::NBLK, reinterpret_cast<float *>(w.d_res), st), SU_FALSE);
  float res[5];                                                        // binary32 sums in the reference's order (fft.hip)
  HIP_TRY(hipMemcpyAsync(res, w.d_res, sizeof res, hipMemcpyDeviceToHost, st), SU_FALSE);
  HIP_TRY(hipStreamSynchronize(st), SU_FALSE);
  // DopplerCalculator.cpp:158-173
  const float lambda = static_cast<float>(299792458.0 / f0);
  float pk = std::atan2(res[1], res[0]);
  if (pk > (float)M_PI) pk -= (float)(2 * M_PI);
  pk = fs * (pk / (float)M_PI) * .5f;                                  // SU_NORM2ABS_FREQ(fs, SU_ANG2NORM_FREQ(pk))
  if (peak) *peak = -lambda * pk;
  if (sigma) *sigma = fs * std::sqrt(res[2]) * .5f;
  if (max) *max = res[3];
  return SU_TRUE;
}

// ---- SpectrumView ----------------------------------------------------------------------------------
struct suamd_specview {
  suamd_ctx *ctx;
  double freqMin, freqMax, freqRange, fftBandwidth;
  float fftRelBw;
  unsigned spectrumSize;
  float *d_psd, *d_accum, *d_count;
  unsigned long long *d_reset = nullptr;      // interpolate()'s count-cap resets, one mask per 64 bins
  sdk::SpecViewLinear *d_geom = nullptr;      // per-frame geometry of a batched sweep
  sdk::SpecViewLinear *h_geom = nullptr;      // its pinned staging copy (the upload must not make the host wait for the stream)
  hipEvent_t geom_ev = nullptr;               // recorded after the upload: h_geom may be rewritten once it has passed
  size_t geom_cap = 0;
};

static unsigned next_pow2_u(unsigned n) { unsigned i = 1; while (i < n) i <<= 1; return i; }

static SUBOOL specview_reset(suamd_specview *v, hipStream_t st)
{
  const size_t bytes = sizeof(float) * SUAMD_SCANNER_SPECTRUM_SIZE;
  HIP_TRY(hipMemsetAsync(v->d_psd, 0, bytes, st), SU_FALSE);
  HIP_TRY(hipMemsetAsync(v->d_accum, 0, bytes, st), SU_FALSE);
  HIP_TRY(hipMemsetAsync(v->d_count, 0, bytes, st), SU_FALSE);
  return SU_TRUE;
}

suamd_specview_t *suamd_specview_new(suamd_ctx_t *ctx)
{
  if (!ctx) { set_err("null context"); return nullptr; }
  HIP_TRY(hipSetDevice(ctx->device), nullptr);
  suamd_specview *v = new (std::nothrow) suamd_specview();
  if (!v) { set_err("out of memory"); return nullptr; }
  v->ctx = ctx;
  v->spectrumSize = SUAMD_SCANNER_SPECTRUM_SIZE;
  v->fftRelBw = .5f;
  v->d_psd = dev_zeros<float>(SUAMD_SCANNER_SPECTRUM_SIZE);
  v->d_accum = dev_zeros<float>(SUAMD_SCANNER_SPECTRUM_SIZE);
  v->d_count = dev_zeros<float>(SUAMD_SCANNER_SPECTRUM_SIZE);
  v->d_reset = dev_zeros<unsigned long long>(1024);
  if (!v->d_psd || !v->d_accum || !v->d_count || !v->d_reset) { set_err("device allocation failed"); suamd_specview_destroy(v); return nullptr; }
  return v;
}

void suamd_specview_destroy(suamd_specview_t *v)
{
  if (!v) return;
  if (v->d_psd) hipFree(v->d_psd);
  if (v->d_accum) hipFree(v->d_accum);
  if (v->d_count) hipFree(v->d_count);
  if (v->d_reset) hipFree(v->d_reset);
  if (v->d_geom) hipFree(v->d_geom);
  if (v->h_geom) hipHostFree(v->h_geom);
  if (v->geom_ev) hipEventDestroy(v->geom_ev);
  delete v;
}

SUBOOL suamd_specview_set_range(suamd_specview_t *v, SUFREQ fmin, SUFREQ fmax, void *stream)
{
  if (!v) { set_err("null argument"); return SU_FALSE; }
  // Panoramic/Scanner.cpp:41-54
  v->freqMin = fmin; v->freqMax = fmax; v->freqRange = fmax - fmin;
  v->spectrumSize = next_pow2_u((unsigned)(v->freqRange / 1000.0));
  if (v->spectrumSize > SUAMD_SCANNER_SPECTRUM_SIZE) v->spectrumSize = SUAMD_SCANNER_SPECTRUM_SIZE;
  return specview_reset(v, as_stream(stream));
}

void suamd_specview_set_fft(suamd_specview_t *v, SUFREQ bw, SUFLOAT rel)
{
  if (!v) return;
  v->fftBandwidth = bw;
  v->fftRelBw = rel;
}

unsigned suamd_specview_spectrum_size(const suamd_specview_t *v) { return v ? v->spectrumSize : 0; }

SUFLOAT *suamd_specview_array(suamd_specview_t *v, int which)
{
  if (!v) return nullptr;
  return which == 0 ? v->d_psd : (which == 1 ? v->d_accum : (which == 2 ? v->d_count : nullptr));
}

static bool specview_is_linear(const suamd_specview *v, double freqMin, double freqMax)
{
  const double fftCount = (freqMax - freqMin) / v->freqRange;          // Scanner.cpp:247
  return fftCount * v->spectrumSize >= 2;
}

// feedLinearMode, Scanner.cpp:126-151 (host side: the per-frame geometry)
static sdk::SpecViewLinear specview_linear_geom(const suamd_specview *v, SUSCOUNT psdSize, double freqMin, double freqMax,
                                                SUBOOL adjustSides)
{
  double inpBw, bw, freqSkip, fftCount, bins, pos;
  int skip;
  sdk::SpecViewLinear g;
  inpBw = freqMax - freqMin;
  skip = adjustSides ? static_cast<int>(.5f * (1 - v->fftRelBw) * psdSize) : 0;
  freqSkip = static_cast<double>(skip) / psdSize * inpBw;
  bw = inpBw - 2 * freqSkip;
  fftCount = static_cast<double>(v->freqRange / bw);
  bins = v->spectrumSize / fftCount;
  g.srcBinW = static_cast<double>(inpBw) / psdSize;
  g.dstBinW = static_cast<double>(v->freqRange) / v->spectrumSize;
  g.delta = g.dstBinW / g.srcBinW;
  pos = static_cast<double>(freqSkip + freqMin - v->freqMin) / (v->freqRange);
  pos *= v->spectrumSize;
  g.j0 = pos > 0 ? static_cast<int>(pos) : 0;
  g.k = pos + bins < v->spectrumSize ? static_cast<int>(pos + bins) : (int)v->spectrumSize;
  g.viewFreqMin = v->freqMin; g.freqMin = freqMin; g.psdSize = (int)psdSize;
  return g;
}

SUBOOL suamd_specview_feed(suamd_specview_t *v, const SUFLOAT *d_psd, const SUFLOAT *d_count, SUSCOUNT psdSize,
                           SUFREQ freqMin, SUFREQ freqMax, SUBOOL adjustSides, void *stream)
{
  if (!v || !d_psd) { set_err("null argument"); return SU_FALSE; }
  if (psdSize == 0 || psdSize > 0x7fffffff || !(v->freqRange > 0)) { set_err("bad frame size / range not set"); return SU_FALSE; }
  hipStream_t st = as_stream(stream);
  if (specview_is_linear(v, freqMin, freqMax)) {
    const sdk::SpecViewLinear g = specview_linear_geom(v, psdSize, freqMin, freqMax, adjustSides);
    HIP_TRY(sdk::specview_feed_linear(g, d_psd, d_count, v->d_accum, v->d_count, st), SU_FALSE);
  } else {
    // feedHistogramMode, Scanner.cpp:194-236
    double relBw = (freqMax - freqMin) / v->freqRange;
    double fStart = (freqMin - v->freqMin) / v->freqRange;
    double fEnd = (freqMax - v->freqMin) / v->freqRange;
    sdk::SpecViewHist g;
    fStart *= v->spectrumSize; fEnd *= v->spectrumSize; relBw *= v->spectrumSize;
    unsigned j = static_cast<unsigned>(fStart);
    if (j > v->spectrumSize - 1) j = v->spectrumSize - 1;
    g.psdSize = (int)psdSize;
    g.inv = (float)(1. / psdSize);
    g.j = j; g.spectrumSize = v->spectrumSize;
    g.split = std::floor(fStart) != std::floor(fEnd);
    g.t = g.split ? static_cast<float>((fStart - std::floor(fStart)) / relBw) : 0.0f;
    HIP_TRY(sdk::specview_feed_hist(g, d_psd, v->d_accum, v->d_count, st), SU_FALSE);
  }
  HIP_TRY(sdk::specview_interpolate(v->d_psd, v->d_accum, v->d_count, (int)v->spectrumSize, v->d_reset, st), SU_FALSE);
  return SU_TRUE;
}

SUBOOL suamd_specview_feed_sweep(suamd_specview_t *v, const SUFLOAT *d_psd, SUSCOUNT psdSize, SUSCOUNT nframes,
                                 const SUFREQ *center, SUBOOL adjustSides, void *stream)
{
  if (!v || !d_psd || !center) { set_err("null argument"); return SU_FALSE; }
  if (nframes == 0) return SU_TRUE;
  if (psdSize == 0 || psdSize > 0x7fffffff || nframes > 0x7fffffff || !(v->freqRange > 0)) {
    set_err("bad frame size / range not set"); return SU_FALSE;
  }
  // SpectrumView::feed(psd, count, size, center, adjust), Scanner.cpp:258-273, once per frame
  bool all_linear = true;
  for (SUSCOUNT f = 0; f < nframes && all_linear; ++f)
    all_linear = specview_is_linear(v, center[f] - v->fftBandwidth / 2, center[f] + v->fftBandwidth / 2);
  if (!all_linear) {
    for (SUSCOUNT f = 0; f < nframes; ++f) {
      if (!suamd_specview_feed(v, d_psd + f * psdSize, nullptr, psdSize, center[f] - v->fftBandwidth / 2,
                               center[f] + v->fftBandwidth / 2, adjustSides, stream))
        return SU_FALSE;
    }
    return SU_TRUE;
  }
  // all frames are linear-mode feeds: one launch replays feed + count-cap for every frame in order,
  // then interpolate() once (its output after the earlier frames would have been overwritten anyway)
  hipStream_t st = as_stream(stream);
  HIP_TRY(hipSetDevice(v->ctx->device), SU_FALSE);
  // The geometry goes up through a pinned staging buffer owned by the view, so the call only enqueues: no host-side wait
  // for the stream (the frames of the sweep are typically still being computed on it).
  if (!v->geom_ev) HIP_TRY(hipEventCreateWithFlags(&v->geom_ev, hipEventDisableTiming), SU_FALSE);
  else HIP_TRY(hipEventSynchronize(v->geom_ev), SU_FALSE);   // the previous sweep's upload (long done, normally)
  if (v->geom_cap < nframes) {
    if (v->d_geom) { HIP_TRY(hipStreamSynchronize(st), SU_FALSE); hipFree(v->d_geom); v->d_geom = nullptr; }
    if (v->h_geom) { hipHostFree(v->h_geom); v->h_geom = nullptr; }
    v->geom_cap = 0;
    HIP_TRY(hipMalloc(&v->d_geom, nframes * sizeof(sdk::SpecViewLinear)), SU_FALSE);
    HIP_TRY(hipHostMalloc(reinterpret_cast<void **>(&v->h_geom), nframes * sizeof(sdk::SpecViewLinear), hipHostMallocDefault), SU_FALSE);
    v->geom_cap = nframes;
  }
  for (SUSCOUNT f = 0; f < nframes; ++f)
    v->h_geom[f] = specview_linear_geom(v, psdSize, center[f] - v->fftBandwidth / 2, center[f] + v->fftBandwidth / 2, adjustSides);
  HIP_TRY(hipMemcpyAsync(v->d_geom, v->h_geom, nframes * sizeof(sdk::SpecViewLinear), hipMemcpyHostToDevice, st), SU_FALSE);
  HIP_TRY(hipEventRecord(v->geom_ev, st), SU_FALSE);
  // d_psd doubles as the snapshot of the counts before the sweep (neighbour validity)
  HIP_TRY(hipMemcpyAsync(v->d_psd, v->d_count, sizeof(float) * v->spectrumSize, hipMemcpyDeviceToDevice, st), SU_FALSE);
  HIP_TRY(sdk::specview_sweep_linear(v->d_geom, (int)nframes, d_psd, (long long)psdSize, v->d_psd, v->d_accum, v->d_count,
                                     (int)v->spectrumSize, st), SU_FALSE);
  HIP_TRY(sdk::specview_interpolate(v->d_psd, v->d_accum, v->d_count, (int)v->spectrumSize, v->d_reset, st), SU_FALSE);
  return SU_TRUE;
}

}  // extern "C"
