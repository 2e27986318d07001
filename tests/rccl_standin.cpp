// tests/rccl_standin.cpp -- TEST INFRASTRUCTURE ONLY: a stand-in for librccl that lets the one-GPU test box run the
// analyzer's RCCL branch (csrc/analyzer.cpp: setup_rccl, the root's ncclBroadcast per block, the shards' matching calls)
// with the device list 0,0,... .  Real RCCL refuses two ranks on one device; this library exports the two entry points
// the analyzer dlsym()s -- ncclCommInitAll, ncclBroadcast (+ ncclCommDestroy) -- with RCCL's single-process semantics:
// one communicator per rank, one thread per rank, a broadcast completes on a rank's stream once that rank has the
// root's bytes, and the root's stream does not pass its call before every rank has them (so the root may overwrite its
// buffer afterwards, as it does with the next block).  Same device: the "transfer" is a device-to-device copy.
//
// A MISSING rank (round 5).  With librccl a call returns at once and the collective lives on the stream: if a rank never
// calls, the root's stream sits in a kernel that waits for it -- for ever, or until ncclCommAbort.  Ranks that share one
// device cannot wait for each other on the device (their streams share hardware queues: the waiting kernel would sit in
// front of the work it waits for), so the stand-in waits for the other ranks on the HOST for a grace period
// (STANDIN_GRACE_MS, default 1000) and only when a rank has not shown up by then puts a kernel on the root's stream that
// spins until ncclCommAbort(root's communicator) releases it (or STANDIN_SPIN_LIMIT_S, default 20 s, have passed: a test
// must fail, not hang the box) -- what csrc/analyzer.cpp's broadcast watchdog is there for.
//
// Built by tests/test_gpu_analyzer_fft.py with hipcc into a temporary directory and handed to the analyzer through
// SUAMD_RCCL_LIB together with SUAMD_RCCL_ALLOW_SAME_DEVICE=1.  Nothing under sigdigger_amd/ refers to it.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdlib>
#include <condition_variable>
#include <map>
#include <mutex>
#include <vector>

namespace {

struct Call { const void *src = nullptr; hipEvent_t ready = nullptr; std::vector<hipEvent_t> got; };
struct Group {
  int n = 0;
  std::mutex m;
  std::condition_variable cv;
  std::map<unsigned long long, Call> calls;                    // by call index (every rank issues its broadcasts in the same order)
};
struct Comm { Group *g; int rank; unsigned long long seq = 0; unsigned *abort_flag = nullptr; };   // abort_flag: pinned host memory
unsigned long long g_broadcasts = 0, g_bytes = 0, g_aborts = 0, g_orphans = 0;
std::mutex g_m;
constexpr auto kWait = std::chrono::seconds(20);               // a test must fail, not hang

std::chrono::milliseconds grace()
{
  const char *e = std::getenv("STANDIN_GRACE_MS");
  const long v = e ? std::atol(e) : 1000;
  return std::chrono::milliseconds(v > 0 ? v : 1000);
}

unsigned long long spin_limit_ticks()
{
  const char *e = std::getenv("STANDIN_SPIN_LIMIT_S");
  const double s = e ? std::atof(e) : 20.0;
  return (unsigned long long)((s > 0 ? s : 20.0) * 1e8);       // wall_clock64(): 100 MHz
}

// the root's stream with a rank missing: nothing gets past this before ncclCommAbort (or the limit)
__global__ void orphan_kernel(const unsigned *abort_flag, unsigned long long limit)
{
  if (threadIdx.x) return;
  const unsigned long long t0 = wall_clock64();
  while (!__hip_atomic_load(abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) && wall_clock64() - t0 < limit)
    __builtin_amdgcn_s_sleep(64);
}

size_t type_size(int t) { static const size_t sz[] = {1, 1, 4, 4, 8, 8, 2, 4, 8, 2}; return t >= 0 && t < 10 ? sz[t] : 0; }

}  // namespace

extern "C" {

__attribute__((visibility("default"))) int ncclCommInitAll(void **comms, int ndev, const int *)
{
  if (!comms || ndev < 1) return 4;                            // ncclInvalidArgument
  Group *g = new Group;
  g->n = ndev;
  for (int r = 0; r < ndev; ++r) {
    Comm *c = new Comm{g, r};
    if (hipHostMalloc((void **)&c->abort_flag, sizeof(unsigned), hipHostMallocDefault) != hipSuccess) return 1;
    *c->abort_flag = 0;
    comms[r] = c;
  }
  return 0;
}

// releases what this rank's stream is stuck in and retires the communicator (as ncclCommAbort does; the flag's page stays:
// the kernel it releases is still reading it)
__attribute__((visibility("default"))) int ncclCommAbort(void *c)
{
  Comm *cm = static_cast<Comm *>(c);
  if (!cm) return 4;
  __atomic_store_n(cm->abort_flag, 1u, __ATOMIC_RELEASE);
  { std::lock_guard<std::mutex> gl(g_m); ++g_aborts; }
  Group *g = cm->g;
  bool last;
  { std::lock_guard<std::mutex> lk(g->m); last = --g->n == 0; }
  delete cm;
  if (last) delete g;
  return 0;
}

__attribute__((visibility("default"))) int ncclCommDestroy(void *c)
{
  Comm *cm = static_cast<Comm *>(c);
  if (!cm) return 4;
  Group *g = cm->g;
  bool last;
  { std::lock_guard<std::mutex> lk(g->m); last = --g->n == 0; }
  delete cm;
  if (last) delete g;
  return 0;
}

__attribute__((visibility("default"))) int ncclBroadcast(const void *send, void *recv, size_t count, int dtype, int root, void *c, hipStream_t st)
{
  Comm *cm = static_cast<Comm *>(c);
  const size_t bytes = count * type_size(dtype);
  if (!cm || !bytes) return 4;
  Group *g = cm->g;
  const unsigned long long k = cm->seq++;
  std::unique_lock<std::mutex> lk(g->m);
  if (cm->rank == root) {
    Call &cl = g->calls[k];
    cl.src = send;
    if (hipEventCreateWithFlags(&cl.ready, hipEventDisableTiming) != hipSuccess || hipEventRecord(cl.ready, st) != hipSuccess) return 1;
    g->cv.notify_all();
    if (recv != send && hipMemcpyAsync(recv, send, bytes, hipMemcpyDeviceToDevice, st) != hipSuccess) return 1;
    // the root's stream passes this call once every other rank has its copy
    const int peers = g->n - 1;
    if (!g->cv.wait_for(lk, grace(), [&] { return (int)g->calls[k].got.size() >= peers; })) {
      // a rank is missing: the call itself succeeds (it only enqueues), the stream is what hangs
      static const unsigned long long limit = spin_limit_ticks();
      hipLaunchKernelGGL(orphan_kernel, dim3(1), dim3(64), 0, st, cm->abort_flag, limit);
      std::lock_guard<std::mutex> gl(g_m);
      ++g_orphans;
      return hipGetLastError() == hipSuccess ? 0 : 1;
    }
    for (hipEvent_t e : g->calls[k].got) { (void)hipStreamWaitEvent(st, e, 0); }
    // (the events are released with the call record; the stream wait has captured them)
    for (hipEvent_t e : g->calls[k].got) (void)hipEventDestroy(e);
    (void)hipEventDestroy(g->calls[k].ready);
    g->calls.erase(k);
    std::lock_guard<std::mutex> gl(g_m);
    ++g_broadcasts; g_bytes += bytes;
    return 0;
  }
  if (!g->cv.wait_for(lk, kWait, [&] { return g->calls.count(k) && g->calls[k].ready != nullptr; })) return 6;
  Call &cl = g->calls[k];
  hipEvent_t done = nullptr;
  if (hipStreamWaitEvent(st, cl.ready, 0) != hipSuccess || hipMemcpyAsync(recv, cl.src, bytes, hipMemcpyDeviceToDevice, st) != hipSuccess ||
      hipEventCreateWithFlags(&done, hipEventDisableTiming) != hipSuccess || hipEventRecord(done, st) != hipSuccess) return 1;
  cl.got.push_back(done);
  g->cv.notify_all();
  return 0;
}

// what the test reads: how many broadcasts completed through this library, and their payload
__attribute__((visibility("default"))) unsigned long long standin_broadcasts(void) { std::lock_guard<std::mutex> gl(g_m); return g_broadcasts; }
__attribute__((visibility("default"))) unsigned long long standin_bytes(void) { std::lock_guard<std::mutex> gl(g_m); return g_bytes; }
__attribute__((visibility("default"))) unsigned long long standin_aborts(void) { std::lock_guard<std::mutex> gl(g_m); return g_aborts; }
__attribute__((visibility("default"))) unsigned long long standin_orphans(void) { std::lock_guard<std::mutex> gl(g_m); return g_orphans; }

}  // extern "C"
