// Accuracy of mpp::fpow (mpp_amd/csrc/fastpow.hpp) on the host against powl (80-bit long double): prints the largest error in
// units of the last place over the sweeps the constitutive laws span, and over random arguments.  Built and run by
// tests/test_fastpow.py (host code only: no GPU call is made).
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <random>

#include "../mpp_amd/csrc/fastpow.hpp"

static double ulp_err(double got, long double want) {
  if (want == 0.0L) return got == 0.0 ? 0.0 : 1e9;
  int e;
  frexpl(want, &e);                       // want = f 2^e, f in [1/2, 1)
  const long double ulp = ldexpl(1.0L, e - 53);
  return (double)(fabsl((long double)got - want) / ulp);
}

int main() {
  const mpp::PowTab pt{mpp::host_pow_tables(), mpp::host_pow_tables() + 4 * mpp::kPowLogEntries};
  double worst = 0.0, wx = 0, wy = 0;
  long long n = 0;
  auto probe = [&](double x, double y) {
    const double got = mpp::fpow(x, y, pt);
    const long double want = powl((long double)x, (long double)y);
    if (!std::isfinite((double)want) || want < 1e-300L) return;
    const double e = ulp_err(got, want);
    ++n;
    if (e > worst) { worst = e; wx = x; wy = y; }
  };
  // the laws' arguments: t = -alpha pc in (1e-6, 1e4), exponents -lambda, n = 1/(1-m), -m, m, 2.5 + 2/lambda, -5/17
  const double ys[] = {-0.5, -0.37, -0.05, -1.9, 0.4, 0.5, 1.6666666666666667, 2.0, -0.4, 6.5, 7.9, 12.5, -5.0 / 17.0, 0.45, -0.55, 1.0, -1.0, 3.0};
  for (double y : ys)
    for (int k = 0; k < 400000; ++k) {
      const double lx = -6.0 + 10.0 * (k + 0.37) / 400000.0;
      probe(std::pow(10.0, lx), y);
    }
  for (double y : ys)                         // around 1, where log x cancels
    for (int k = -20000; k <= 20000; ++k) probe(1.0 + k * 1e-7, y), probe(1.0 + k * 0x1p-52, y);
  std::mt19937_64 rng(12345);
  std::uniform_real_distribution<double> ux(-300.0, 300.0), uy(-40.0, 40.0);
  for (int k = 0; k < 3000000; ++k) {
    const double x = std::exp2(ux(rng)), y = uy(rng) / (1.0 + std::fabs(std::log2(x)) / 20.0);
    probe(x, y);
  }
  probe(0x1p-1060, 0.5); probe(0x1p-1030, -0.25); probe(1.0, 123.0);
  // outside the domain: NaN; 0^y like libm's pow (0, inf, 1); results that over- or underflow saturate (no int conversion of a huge double)
  const bool special = std::isnan(mpp::fpow(-1.0, 0.5, pt)) && std::isnan(mpp::fpow(INFINITY, 0.5, pt)) &&
                       std::isnan(mpp::fpow(NAN, 0.5, pt)) && mpp::fpow(1.0, 3.7, pt) == 1.0 && mpp::fpow(4.0, 0.5, pt) == 2.0 &&
                       mpp::fpow(0.0, 0.5, pt) == 0.0 && std::isinf(mpp::fpow(0.0, -0.5, pt)) && mpp::fpow(0.0, 0.0, pt) == 1.0 &&
                       mpp::fpow(1e-300, 1e6, pt) == 0.0 && std::isinf(mpp::fpow(1e-300, -1e6, pt)) && mpp::fpow(1e300, -1e300, pt) == 0.0 &&
                       std::isinf(mpp::fpow(1e300, 1e300, pt)) && mpp::fpow(2.0, -1074.0, pt) == 0x1p-1074 && mpp::fpow(2.0, 1023.0, pt) == 0x1p1023;
  std::printf("%lld %.4f %.17g %.17g %d\n", n, worst, wx, wy, special ? 1 : 0);
  return 0;
}
