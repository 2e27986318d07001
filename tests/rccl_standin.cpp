// tests/rccl_standin.cpp -- TEST INFRASTRUCTURE ONLY: a stand-in for librccl that lets the one-GPU test box run the
// analyzer's RCCL branch (csrc/analyzer.cpp: setup_rccl, the root's ncclBroadcast per block, the shards' matching calls)
// with the device list 0,0,... .  Real RCCL refuses two ranks on one device; this library exports the two entry points
// the analyzer dlsym()s -- ncclCommInitAll, ncclBroadcast (+ ncclCommDestroy) -- with RCCL's single-process semantics:
// one communicator per rank, one thread per rank, a broadcast completes on a rank's stream once that rank has the
// root's bytes, and the root's stream does not pass its call before every rank has them (so the root may overwrite its
// buffer afterwards, as it does with the next block).  Same device: the "transfer" is a device-to-device copy.
//
// Built by tests/test_gpu_analyzer_fft.py with hipcc into a temporary directory and handed to the analyzer through
// SUAMD_RCCL_LIB together with SUAMD_RCCL_ALLOW_SAME_DEVICE=1.  Nothing under sigdigger_amd/ refers to it.
#include <hip/hip_runtime.h>
#include <chrono>
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
struct Comm { Group *g; int rank; unsigned long long seq = 0; };
unsigned long long g_broadcasts = 0, g_bytes = 0;
std::mutex g_m;
constexpr auto kWait = std::chrono::seconds(20);               // a test must fail, not hang

size_t type_size(int t) { static const size_t sz[] = {1, 1, 4, 4, 8, 8, 2, 4, 8, 2}; return t >= 0 && t < 10 ? sz[t] : 0; }

}  // namespace

extern "C" {

__attribute__((visibility("default"))) int ncclCommInitAll(void **comms, int ndev, const int *)
{
  if (!comms || ndev < 1) return 4;                            // ncclInvalidArgument
  Group *g = new Group;
  g->n = ndev;
  for (int r = 0; r < ndev; ++r) comms[r] = new Comm{g, r};
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
    if (!g->cv.wait_for(lk, kWait, [&] { return (int)g->calls[k].got.size() == g->n - 1; })) return 6;   // ncclRemoteError-like
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

}  // extern "C"
