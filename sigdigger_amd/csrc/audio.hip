// audio.hip -- the "audio" inspector class (row A6; SPEC.md section Q): what libsuscan's audio inspector does with the
// channel AudioProcessor opens (Default/Audio/AudioProcessor.cpp:94-169) and configures with audio.demodulator /
// audio.cutoff / audio.sample-rate / audio.volume / audio.squelch / audio.squelch-level (:251-270): demodulate the
// channel samples, low-pass at the cut-off and resample from the channel rate to the sound card's.  The consumer
// plays the REAL part of what it gets (Audio/AudioPlayback.cpp:603-604).
//
//   demodulator 1 AM:  a = |x|            2 FM:  a = arg(x conj(x_prev)) / pi
//               3 USB: a = Re(x e^{+j w n})   4 LSB: a = Re(x e^{-j w n}),  w = pi bw / efs -- the client has put the
//                  channel centre half a bandwidth off the carrier (AudioProcessor.cpp:201-228): this puts it back
//               5 RAW: a = x (both components)
//   resampler:  output k sits at t_k = t0 + k efs / fa input samples; y_k = sum_m a[n0 + m] g(m - frac) / sum_m g(m - frac),
//               n0 = floor(t_k), frac = t_k - n0, m = -M .. M + 1, g(u) = sinc(2 fc u) (0.5 + 0.5 cos(pi u / (M + 1))) -- a window that vanishes at the ends of the tap range, so the result does not
//               depend on which side of an integer a rounding of t_k falls --,
//               fc = min(cutoff, 0.45 fa) / efs, M = min(ceil(2 / fc), 128): a windowed sinc evaluated where it is
//               needed (audio rates: a few hundred thousand taps per block), one thread per output sample.
//   squelch:    the block is muted when its mean channel power is below audio.squelch-level.
#include <hip/hip_runtime.h>
#include <cmath>
#include <new>
#include <stdint.h>

#include "../../include/sigdigger_amd.h"
#include "kernels.hpp"
#include "sd_math.hpp"

void suamd_set_error(const char *fmt, ...);                // capi.hip

namespace {

typedef float cf __attribute__((ext_vector_type(2)));
constexpr int MAXM = 128, HIST = 2 * MAXM + 4;

// a[HIST + i] = demod(x[i]); the HIST samples before it are the previous block's tail (copied by audio_tail_kernel)
__global__ void audio_demod_kernel(const cf *x, long long n, int mode, const cf *xprev, uint32_t phase0, uint32_t dphase, cf *a)
{
  const long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x;
  if (i >= n) return;
  const cf v = x[i];
  cf o;
  if (mode == 1) o = cf{sqrtf(v.x * v.x + v.y * v.y), 0.f};
  else if (mode == 2) {
    const cf p = i > 0 ? x[i - 1] : xprev[0];
    const float re = v.x * p.x + v.y * p.y, im = v.y * p.x - v.x * p.y;
    o = cf{sd::atan2_(im, re) * 0.318309886183790671538f, 0.f};
  } else if (mode == 3 || mode == 4) {
    float c, s;
    sd::phasor_u32(phase0 + (uint32_t)i * dphase, c, s);
    o = cf{v.x * c - v.y * s, 0.f};
  } else o = v;
  a[HIST + i] = o;
}

__global__ void audio_tail_kernel(cf *a, const cf *x, long long n, cf *xprev)
{
  // after the resampler has run: the last HIST demodulated samples become the next block's history; x[n-1] its x_prev
  const int i = threadIdx.x;
  cf v = cf{0.f, 0.f};
  if (i < HIST) v = a[n + i];                                // = a[HIST + (n - HIST + i)]: the last HIST samples (old history included when n < HIST)
  __syncthreads();
  if (i < HIST) a[i] = v;
  if (i == 0 && n > 0) xprev[0] = x[n - 1];
}

__global__ void audio_power_kernel(const cf *x, long long n, float *acc)
{
  __shared__ float part[256];
  float s = 0.f;
  for (long long i = threadIdx.x; i < n; i += 256) { const cf v = x[i]; s += v.x * v.x + v.y * v.y; }
  part[threadIdx.x] = s;
  __syncthreads();
  for (int k = 128; k > 0; k >>= 1) { if ((int)threadIdx.x < k) part[threadIdx.x] += part[threadIdx.x + k]; __syncthreads(); }
  if (threadIdx.x == 0) acc[0] = part[0] / (float)(n > 0 ? n : 1);
}

__global__ void audio_resample_kernel(const cf *a, long long count, double t0, double ratio, float fc, int M, float volume,
                                      const float *power, float squelch_level, int squelch, cf *out)
{
  const long long k = blockIdx.x * (long long)blockDim.x + threadIdx.x;
  if (k >= count) return;
  if (squelch && power[0] < squelch_level) { out[k] = cf{0.f, 0.f}; return; }
  const double t = t0 + (double)k * ratio;
  const double fl = floor(t);
  const long long n0 = (long long)fl;
  const float frac = (float)(t - fl);
  float accx = 0.f, accy = 0.f, norm = 0.f;
  const float w0 = 3.14159265358979323846f / (float)(M + 1);
  for (int m = -M; m <= M + 1; ++m) {
    const float u = (float)m - frac;
    const float arg = 6.28318530717958647692f * fc * u;
    const float sinc = fabsf(arg) < 1e-6f ? 1.0f : sinf(arg) / arg;
    const float g = sinc * (0.5f + 0.5f * cosf(w0 * u));
    const cf v = a[HIST + n0 + m];
    accx += v.x * g; accy += v.y * g; norm += g;
  }
  const float sc = volume / norm;
  out[k] = cf{accx * sc, accy * sc};
}

}  // namespace

struct suamd_audio {
  suamd_ctx_t *ctx = nullptr;
  int mode = 2;
  double efs = 0, fa = 0, cutoff = 0, bw = 0;
  float volume = 1.f, squelch_level = 0.f;
  bool squelch = false;
  double t0 = 0;                  // time of the next output sample, in input samples relative to the next block's start
  uint64_t n_in = 0;              // channel samples consumed (phase of the SSB oscillator)
  cf *d_a = nullptr; size_t cap = 0;
  cf *d_xprev = nullptr;
  float *d_power = nullptr;
};

extern "C" {

suamd_audio_t *suamd_audio_new(suamd_ctx_t *ctx, SUFLOAT equiv_fs, SUFLOAT bandwidth)
{
  if (!ctx || !(equiv_fs > 0)) { suamd_set_error("bad argument"); return nullptr; }
  if (hipSetDevice(suamd_ctx_device(ctx)) != hipSuccess) { suamd_set_error("hipSetDevice failed"); return nullptr; }
  auto *au = new (std::nothrow) suamd_audio();
  if (!au) return nullptr;
  au->ctx = ctx; au->efs = equiv_fs; au->bw = bandwidth; au->fa = equiv_fs < 44100 ? equiv_fs : 44100; au->cutoff = 0.45 * au->fa;
  if (hipMalloc((void **)&au->d_xprev, sizeof(cf)) != hipSuccess || hipMalloc((void **)&au->d_power, sizeof(float)) != hipSuccess ||
      hipMemset(au->d_xprev, 0, sizeof(cf)) != hipSuccess || hipStreamSynchronize(nullptr) != hipSuccess) {
    suamd_set_error("allocation failed"); suamd_audio_destroy(au); return nullptr;      // (the fill is over before a feed on another stream)
  }
  return au;
}

void suamd_audio_destroy(suamd_audio_t *au)
{
  if (!au) return;
  for (void *p : {(void *)au->d_a, (void *)au->d_xprev, (void *)au->d_power}) if (p) (void)hipFree(p);
  delete au;
}

SUBOOL suamd_audio_configure(suamd_audio_t *au, int demodulator, SUFLOAT sample_rate, SUFLOAT cutoff, SUFLOAT volume,
                             SUBOOL squelch, SUFLOAT squelch_level)
{
  if (!au || demodulator < 1 || demodulator > 5 || !(sample_rate > 0)) { suamd_set_error("audio: bad configuration"); return SU_FALSE; }
  au->mode = demodulator;
  au->fa = sample_rate < au->efs ? sample_rate : au->efs;     // the channel is never interpolated above its own rate
  au->cutoff = cutoff > 0 ? cutoff : 0.45 * au->fa;
  au->volume = volume;
  au->squelch = squelch != 0; au->squelch_level = squelch_level;
  return SU_TRUE;
}

static void audio_geometry(const suamd_audio *au, float *fc, int *M, double *ratio)
{
  const double cut = au->cutoff < 0.45 * au->fa ? au->cutoff : 0.45 * au->fa;
  *fc = (float)(cut / au->efs);
  int m = (int)std::ceil(2.0 / (double)*fc);
  *M = m < 2 ? 2 : (m > MAXM ? MAXM : m);
  *ratio = au->efs / au->fa;
}

SUSCOUNT suamd_audio_output_count(const suamd_audio_t *au, SUSCOUNT len)
{
  if (!au) return 0;
  float fc; int M; double ratio;
  audio_geometry(au, &fc, &M, &ratio);
  // outputs whose last tap (n0 + M + 1) lies inside the block: floor(t0 + k ratio) + M + 1 <= len - 1
  const double lim = (double)len - 1 - (M + 1);
  if (au->t0 > lim + 1) return 0;
  SUSCOUNT k = (SUSCOUNT)std::floor((lim + 1 - au->t0) / ratio);
  while (std::floor(au->t0 + (double)k * ratio) <= lim) ++k;
  while (k > 0 && std::floor(au->t0 + (double)(k - 1) * ratio) > lim) --k;
  return k;
}

SUBOOL suamd_audio_feed(suamd_audio_t *au, const suamd_complex *d_x, SUSCOUNT len, suamd_complex *d_out, SUSCOUNT *n_out, void *stream)
{
  if (!au || (len && !d_x)) { suamd_set_error("null argument"); return SU_FALSE; }
  if (n_out) *n_out = 0;
  if (len == 0) return SU_TRUE;
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (len + 2 * HIST > au->cap) {
    cf *na = nullptr;
    const size_t cap = (size_t)len + 2 * HIST + 1024;
    if (hipMalloc((void **)&na, cap * sizeof(cf)) != hipSuccess) { suamd_set_error("allocation failed"); return SU_FALSE; }
    (void)hipStreamSynchronize(s);
    if (au->d_a) { (void)hipMemcpy(na, au->d_a, HIST * sizeof(cf), hipMemcpyDeviceToDevice); (void)hipStreamSynchronize(nullptr); (void)hipFree(au->d_a); }
    else { (void)hipMemset(na, 0, HIST * sizeof(cf)); (void)hipStreamSynchronize(nullptr); }
    au->d_a = na; au->cap = cap;
  }
  float fc; int M; double ratio;
  audio_geometry(au, &fc, &M, &ratio);
  const SUSCOUNT count = suamd_audio_output_count(au, len);
  if (count && !d_out) { suamd_set_error("null output"); return SU_FALSE; }
  const double w = 3.14159265358979323846 * au->bw / au->efs;                 // rad / sample: half the channel bandwidth
  const uint32_t dphase = (uint32_t)(int64_t)std::llround((au->mode == 3 ? w : -w) / (2 * 3.14159265358979323846) * 4294967296.0);
  const cf *x = reinterpret_cast<const cf *>(d_x);
  hipLaunchKernelGGL(audio_demod_kernel, dim3((unsigned)((len + 255) / 256)), dim3(256), 0, s, x, (long long)len, au->mode, au->d_xprev,
                     (uint32_t)au->n_in * dphase, dphase, au->d_a);
  if (au->squelch) hipLaunchKernelGGL(audio_power_kernel, dim3(1), dim3(256), 0, s, x, (long long)len, au->d_power);
  if (count)
    hipLaunchKernelGGL(audio_resample_kernel, dim3((unsigned)((count + 127) / 128)), dim3(128), 0, s, au->d_a, (long long)count, au->t0, ratio,
                       fc, M, au->volume, au->d_power, au->squelch_level, au->squelch ? 1 : 0, reinterpret_cast<cf *>(d_out));
  hipLaunchKernelGGL(audio_tail_kernel, dim3(1), dim3(512), 0, s, au->d_a, x, (long long)len, au->d_xprev);
  if (hipGetLastError() != hipSuccess) { suamd_set_error("audio launch failed"); return SU_FALSE; }
  au->t0 = au->t0 + (double)count * ratio - (double)len;
  au->n_in += len;
  if (n_out) *n_out = count;
  return SU_TRUE;
}

}  // extern "C"
