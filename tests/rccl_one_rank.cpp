// tests/rccl_one_rank.cpp -- TEST INFRASTRUCTURE ONLY: the calls csrc/analyzer.cpp makes into librccl (setup_rccl, the root's
// ncclBroadcast per block, bus_close / the broadcast watchdog), made against the REAL library with one rank on one GPU -- all a
// one-GPU box allows (RCCL refuses two ranks on one device).  Same lookups, same function-pointer types, same argument
// shapes as the analyzer's: the library is dlopen()ed under the names the analyzer tries, the four symbols are resolved by
// name, ncclCommInitAll builds the communicator from a device list, the block goes out in place as bytes (datatype 0 =
// ncclInt8, count = bytes, root 0) on a stream of ours, ncclCommDestroy / ncclCommAbort end it.  Prints "OK ..." and
// returns 0 when every call returned ncclSuccess and the buffer still holds the block.
// `--symbols-only`: stop after the lookups (no GPU needed: the CPU suite runs this form).
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <cstdio>
#include <cstring>
#include <vector>

int main(int argc, char **argv)
{
  const bool symbols_only = argc > 1 && !std::strcmp(argv[1], "--symbols-only");
  void *lib = dlopen("librccl.so.1", RTLD_NOW | RTLD_LOCAL);
  if (!lib) lib = dlopen("librccl.so", RTLD_NOW | RTLD_LOCAL);
  if (!lib) { std::printf("NOLIB %s\n", dlerror()); return 3; }
  using InitAll = int (*)(void **, int, const int *);
  using Bcast = int (*)(const void *, void *, size_t, int, int, void *, hipStream_t);
  using End = int (*)(void *);
  auto init_all = reinterpret_cast<InitAll>(dlsym(lib, "ncclCommInitAll"));
  auto bcast = reinterpret_cast<Bcast>(dlsym(lib, "ncclBroadcast"));
  auto destroy = reinterpret_cast<End>(dlsym(lib, "ncclCommDestroy"));
  auto abort_ = reinterpret_cast<End>(dlsym(lib, "ncclCommAbort"));
  if (!init_all || !bcast || !destroy || !abort_) { std::printf("MISSING %d %d %d %d\n", !!init_all, !!bcast, !!destroy, !!abort_); return 4; }
  if (symbols_only) { std::printf("OK symbols\n"); return 0; }

  const int dev = 0;
  if (hipSetDevice(dev) != hipSuccess) { std::printf("NODEVICE\n"); return 5; }
  const size_t n = 2u << 20;                                   // a 2 Mi-sample block of complex float32: 16 MiB
  const size_t bytes = n * 8;
  std::vector<unsigned> h(bytes / 4), back(bytes / 4);
  for (size_t i = 0; i < h.size(); ++i) h[i] = (unsigned)(i * 2654435761u) ^ 0x5bd1e995u;
  void *d = nullptr;
  hipStream_t st = nullptr;
  if (hipMalloc(&d, bytes) != hipSuccess || hipStreamCreateWithFlags(&st, hipStreamNonBlocking) != hipSuccess) { std::printf("NOMEM\n"); return 6; }
  (void)hipMemcpy(d, h.data(), bytes, hipMemcpyHostToDevice);
  int rc = 0;
  for (int round = 0; round < 2 && rc == 0; ++round) {         // round 0 ends with ncclCommDestroy, round 1 with ncclCommAbort
    void *comm = nullptr;
    const int devs[1] = {dev};
    int e = init_all(&comm, 1, devs);
    if (e != 0 || !comm) { std::printf("INIT %d\n", e); rc = 7; break; }
    (void)hipSetDevice(dev);
    for (int k = 0; k < 3 && rc == 0; ++k) {
      e = bcast(d, d, bytes, 0, 0, comm, st);
      if (e != 0) { std::printf("BCAST %d\n", e); rc = 8; }
    }
    if (hipStreamSynchronize(st) != hipSuccess) { std::printf("SYNC\n"); rc = 9; }
    e = round == 0 ? destroy(comm) : abort_(comm);
    if (e != 0 && rc == 0) { std::printf("END %d (round %d)\n", e, round); rc = 10; }
  }
  (void)hipMemcpy(back.data(), d, bytes, hipMemcpyDeviceToHost);
  if (rc == 0 && std::memcmp(back.data(), h.data(), bytes) != 0) { std::printf("DATA\n"); rc = 11; }
  (void)hipFree(d);
  (void)hipStreamDestroy(st);
  if (rc == 0) std::printf("OK one rank, %zu bytes per broadcast\n", bytes);
  return rc;
}
