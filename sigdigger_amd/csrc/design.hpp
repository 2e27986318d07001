// design.hpp -- parameter design shared by the device banks (capi.hip) and the per-sample host calls (sigutils_host.cpp):
// binary64, not on the hot path.  One source, so that a su_costas_t on the host and a suamd_costas_bank_t on the GPU
// get the same arm-filter coefficients bit for bit.
#pragma once
#include <cmath>

namespace sdk_design {

// bilinear-transform Butterworth low-pass, order <= 4, cut-off fc (1 = Nyquist); b, a: 5 coefficients each, a[0] = 1
// (SPEC.md section E: "arm filter = Butterworth low-pass of order arm_order - 1 ... equals scipy.signal.butter")
inline void butter_lp(int order, double fc, float *b, float *a)
{
  constexpr double kPi = 3.14159265358979323846;
  const double wc = std::tan(0.5 * kPi * fc);
  double ar[5] = {1, 0, 0, 0, 0}, ai[5] = {0, 0, 0, 0, 0};
  const int n = order;
  for (int i = 0; i < n; ++i) {
    const double th = kPi * (2.0 * i + n + 1.0) / (2.0 * n);
    double cth, sth;
    ::sincos(th, &sth, &cth);                              // explicitly: compilers disagree on merging sin + cos, glibc's sincos differs from them in rare last bits
    const double pr = wc * cth, pi = wc * sth;
    const double dr = 1.0 - pr, di = -pi, nr = 1.0 + pr, ni = pi;
    const double den = dr * dr + di * di;
    const double zr = (nr * dr + ni * di) / den, zi = (ni * dr - nr * di) / den;
    for (int k = i + 1; k >= 1; --k) {
      const double tr = ar[k] - (zr * ar[k - 1] - zi * ai[k - 1]);
      const double ti = ai[k] - (zr * ai[k - 1] + zi * ar[k - 1]);
      ar[k] = tr; ai[k] = ti;
    }
  }
  double bn[5] = {1, 0, 0, 0, 0}, sa = 0, sb = 0;
  for (int i = 0; i < n; ++i)
    for (int k = i + 1; k >= 1; --k) bn[k] += bn[k - 1];
  for (int k = 0; k <= n; ++k) { sa += ar[k]; sb += bn[k]; }
  for (int k = 0; k <= 4; ++k) { b[k] = 0; a[k] = 0; }
  for (int k = 0; k <= n; ++k) { b[k] = (float)(bn[k] * sa / sb); a[k] = (float)ar[k]; }
}

}  // namespace sdk_design
