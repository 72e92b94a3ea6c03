// The shared-subexpression IFC-67 evaluation the coupled thermal-hydrology kernels use (ifc67_temperature_terms / ifc67_at_pressure,
// mpp_amd/csrc/th_physics.hpp) against the term-by-term functions of the same header evaluated with the math library's pow
// (density_ifc67_pt<true>, enthalpy_ifc67<true>: what the oracle's C restatement computes): largest relative differences of density,
// enthalpy and their four derivatives over the range soils live in.  Built and run by tests/test_host_laws.py (host code, no GPU).
#include <cmath>
#include <cstdio>

#include "../mpp_amd/csrc/th_physics.hpp"

int main() {
  const mpp::PowTab pt{mpp::host_pow_tables(), mpp::host_pow_tables() + 4 * mpp::kPowLogEntries};
  double worst[6] = {0, 0, 0, 0, 0, 0};
  long long n = 0;
  for (int it = 0; it <= 400; ++it) {
    const double t_c = 0.05 + 99.9 * it / 400.0;                 // degC
    const mpp::Ifc67Temp f = mpp::ifc67_temperature_terms(t_c, true);
    for (int ip = 0; ip <= 400; ++ip) {
      const double p = ip < 200 ? 100.0 + (101325.0 - 100.0) * ip / 200.0 : 101325.0 * std::pow(200.0, (ip - 200) / 200.0);   // 100 Pa .. 20 MPa
      double d0, dp0, dt0, h0, hp0, ht0, d1, dp1, dt1, h1, hp1, ht1;
      mpp::density_ifc67_pt<true>(t_c, p, d0, dp0, dt0, pt);
      mpp::enthalpy_ifc67<true>(t_c, p, h0, hp0, ht0, pt);
      mpp::ifc67_at_pressure<true>(f, p, d1, dp1, dt1, h1, hp1, ht1, pt);
      const double got[6] = {d1, dp1, dt1, h1, hp1, ht1}, want[6] = {d0, dp0, dt0, h0, hp0, ht0};
      // derivatives that pass through zero inside the range (dden/dT at 4 degC) are measured against the scale of their terms
      const double scale[6] = {std::fabs(d0), std::fabs(dp0), std::fabs(dt0) + 1e-3, std::fabs(h0) + 1e4, std::fabs(hp0) + 1e-3, std::fabs(ht0)};
      for (int k = 0; k < 6; ++k) {
        const double e = std::fabs(got[k] - want[k]) / scale[k];
        if (e > worst[k]) worst[k] = e;
      }
      double d2, dp2, dt2, u1, u2, u3;
      mpp::ifc67_at_pressure<false>(f, p, d2, dp2, dt2, u1, u2, u3, pt);
      if (d2 != d1 || dp2 != dp1 || dt2 != dt1) worst[0] = 1.0;   // the density-only instantiation gives the same bits
      ++n;
    }
  }
  std::printf("%lld %.3e %.3e %.3e %.3e %.3e %.3e\n", n, worst[0], worst[1], worst[2], worst[3], worst[4], worst[5]);
  return 0;
}
