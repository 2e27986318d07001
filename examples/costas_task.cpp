// examples/costas_task.cpp -- what Tasks/CostasRecoveryTask.cpp looks like on top of libsigdigger_amd:
// the per-sample su_costas_feed loop of CostasRecoveryTask::work (Tasks/CostasRecoveryTask.cpp:58-61)
// becomes one block call per work() slice; everything else (slice size, progress, the
// CancellableTask protocol) stays as it is.  Plain C++ against the C ABI and the HIP runtime, no
// Python, no torch:
//
//   hipcc --offload-arch=gfx950 examples/costas_task.cpp -Iinclude -Lsigdigger_amd -lsigdigger_amd \
//         -Wl,-rpath,$PWD/sigdigger_amd -o costas_task
//   ./costas_task in.raw out.raw <kind 1|2|3> <tau> <loopbw>
#include <hip/hip_runtime.h>
#include <sigdigger_amd.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define ATTEMPT(expr)                                                                        \
  do {                                                                                       \
    if (!(expr)) {                                                                           \
      std::fprintf(stderr, "%s:%d: %s failed: %s\n", __FILE__, __LINE__, #expr, suamd_last_error()); \
      return 1;                                                                              \
    }                                                                                        \
  } while (0)

// the task object: the members CostasRecoveryTask keeps, with the by-value su_costas_t replaced by a bank
struct CostasRecoveryTask {
  const suamd_complex *origin;        // host capture (TimeWindow's data)
  suamd_complex *destination;
  size_t length, p = 0;
  suamd_costas_bank_t *costas = nullptr;
  suamd_complex *d_in = nullptr, *d_out = nullptr;
  static constexpr size_t BLOCK = 0x10000;       // SIGDIGGER_COSTAS_RECOVERY_TASK_BLOCK_LENGTH-style slice

  bool init(suamd_ctx_t *ctx, int kind, float tau, float loopbw)
  {
    // su_costas_init(&costas, kind, 0, 1 / tau, 3, loopbw)   (Tasks/CostasRecoveryTask.cpp:36-42)
    costas = suamd_costas_bank_new(ctx, 1, kind, 0.0f, 1.0f / tau, 3, loopbw);
    return costas && hipMalloc(&d_in, BLOCK * sizeof(suamd_complex)) == hipSuccess &&
           hipMalloc(&d_out, BLOCK * sizeof(suamd_complex)) == hipSuccess;
  }

  // CostasRecoveryTask::work(): one slice per call, true while there is more to do
  bool work()
  {
    size_t amount = length - p;
    if (amount > BLOCK) amount = BLOCK;
    const suamd_view row = {BLOCK, 1};
    if (hipMemcpy(d_in, origin + p, amount * sizeof(suamd_complex), hipMemcpyHostToDevice) != hipSuccess) return false;
    if (!suamd_costas_bank_feed(costas, d_in, row, d_out, row, amount, nullptr)) return false;
    if (hipMemcpy(destination + p, d_out, amount * sizeof(suamd_complex), hipMemcpyDeviceToHost) != hipSuccess) return false;
    p += amount;
    return p < length;
  }

  ~CostasRecoveryTask()
  {
    if (costas) suamd_costas_bank_destroy(costas);
    if (d_in) (void)hipFree(d_in);
    if (d_out) (void)hipFree(d_out);
  }
};

int main(int argc, char **argv)
{
  if (argc < 6) { std::fprintf(stderr, "usage: %s in.raw out.raw kind tau loopbw\n", argv[0]); return 2; }
  FILE *fi = std::fopen(argv[1], "rb");
  if (!fi) { std::perror(argv[1]); return 1; }
  std::fseek(fi, 0, SEEK_END);
  const size_t n = (size_t)std::ftell(fi) / sizeof(suamd_complex);
  std::rewind(fi);
  std::vector<suamd_complex> in(n), out(n);
  if (std::fread(in.data(), sizeof(suamd_complex), n, fi) != n) { std::fclose(fi); return 1; }
  std::fclose(fi);

  suamd_ctx_t *ctx = suamd_ctx_new(0);
  ATTEMPT(ctx != nullptr);
  {
    CostasRecoveryTask task;
    task.origin = in.data(); task.destination = out.data(); task.length = n;
    ATTEMPT(task.init(ctx, std::atoi(argv[3]), (float)std::atof(argv[4]), (float)std::atof(argv[5])));
    while (task.work()) {}
    ATTEMPT(task.p == n);
  }
  suamd_ctx_destroy(ctx);
  FILE *fo = std::fopen(argv[2], "wb");
  if (!fo) { std::perror(argv[2]); return 1; }
  std::fwrite(out.data(), sizeof(suamd_complex), n, fo);
  std::fclose(fo);
  std::printf("%zu samples\n", n);
  return 0;
}
