// LD_PRELOAD shim: logs every hipMalloc / hipFree / hipHostMalloc (address range, size, thread, time) to the file named by
// HIPALLOC_LOG, so that the address of a "Memory access fault by GPU" can be matched to the allocation it lies in or next to.
//   g++ -O2 -fPIC -shared tools/hipalloc_log.cpp -o hipalloc_log.so -ldl -lpthread
#include <dlfcn.h>
#include <stdio.h>
#include <stdlib.h>
#include <time.h>
#include <pthread.h>
#include <mutex>
#include <execinfo.h>
#include <string.h>
typedef int hipError_t;
static FILE *g_f;
static std::mutex g_m;
static double now() { timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec + 1e-9 * t.tv_nsec; }
static void *real_sym(const char *name)
{
  void *f = dlsym(RTLD_NEXT, name);
  for (const char *lib : {"libamdhip64.so.7", "libamdhip64.so.6", "libamdhip64.so"}) {
    if (f) break;
    void *h = dlopen(lib, RTLD_NOW | RTLD_NOLOAD);
    if (!h) h = dlopen(lib, RTLD_NOW | RTLD_GLOBAL);
    if (h) f = dlsym(h, name);
  }
  if (!f) { fprintf(stderr, "hipalloc_log: %s not found\n", name); abort(); }
  return f;
}
static FILE *out() { if (!g_f) { const char *p = getenv("HIPALLOC_LOG"); g_f = p ? fopen(p, "w") : stderr; } return g_f; }
extern "C" hipError_t hipMalloc(void **p, size_t n)
{
  static auto real = (hipError_t(*)(void **, size_t))real_sym("hipMalloc");
  hipError_t e = real(p, n);
  std::lock_guard<std::mutex> l(g_m);
  void *bt[10]; const int nb = backtrace(bt, 10);
  fprintf(out(), "%.6f M %p %zu %lx %d", now(), e ? nullptr : *p, n, (unsigned long)pthread_self(), e);
  for (int i = 1; i < nb; ++i) {
    Dl_info di;
    if (dladdr(bt[i], &di) && di.dli_fname) {
      const char *b = strrchr(di.dli_fname, '/');
      fprintf(out(), " %s+%lx(%s)", b ? b + 1 : di.dli_fname, (unsigned long)((char *)bt[i] - (char *)di.dli_fbase), di.dli_sname ? di.dli_sname : "?");
    }
  }
  fprintf(out(), "\n"); fflush(out());
  return e;
}
extern "C" hipError_t hipFree(void *p)
{
  static auto real = (hipError_t(*)(void *))real_sym("hipFree");
  { std::lock_guard<std::mutex> l(g_m); fprintf(out(), "%.6f F %p 0 %lx 0\n", now(), p, (unsigned long)pthread_self()); fflush(out()); }
  return real(p);
}
