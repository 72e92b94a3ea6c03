// fastpow.hpp -- x^y for x > 0 with two small tables: the real powers of the retention curves and of the IFC-67 formulation.
//
// Why: the library pow() is ~250 instructions (~150 of them fp64) on gfx950 and the constitutive laws other than van Genuchten
// with m = 1/2 spend most of their instructions in it (SaturationFunction.F90:747-857, 900-990, 1027-1140 call `**` with real
// exponents two to five times per cell).  This one is ~65 instructions and two LDS reads, with the same accuracy class (< 1 ulp;
// tests/test_fastpow.py measures it against a long-double reference on the host).
//
// Method (the classical table-driven scheme, e.g. Tang 1989/1990; as in the open "optimized routines" libm implementations):
//   log:  x = m 2^e, m in [1/2, 1); the top 7 mantissa bits pick c_i (the centre of m's sub-interval, as the reciprocal 1/c_i rounded
//         to double); r = m / c_i - 1 is exact in one fma, |r| <= 2^-8; log x = e ln2 + log c_i + log1p(r), the first two from
//         hi/lo pairs whose hi parts are multiples of 2^-43 (their sum is exact), log1p by its series through r^7; everything is
//         kept as an unevaluated sum hi + lo with an absolute error near 2^-64.
//   exp:  z = y log x in hi/lo; k = round(z 128 / ln2); 2^(k/128) from a table as hi/lo; exp of the remainder (|.| <= ln2/256)
//         by its series through the 5th power; scaled by 2^(k div 128).
// The tables (128 x {1/c, log c hi, log c lo, 0} and 128 x {2^(i/128) hi, lo}: 6 KB) are computed once on the host in 80-bit long
// double, copied to the device, and staged in LDS by every kernel that evaluates a retention curve (stage_pow_tables).
// Domain: x a positive finite number (subnormals included), |y log x| below ~700; anything else returns NaN -- the callers'
// arguments are bounded away from that by the branches of the laws themselves (pc < 0, pc_alpha > 1, ...).
#pragma once

#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>
#include <cstring>

namespace mpp {

constexpr int kPowLogEntries = 128, kPowExpEntries = 128;
constexpr int kPowTableDoubles = 4 * kPowLogEntries + 2 * kPowExpEntries;   // 768 doubles = 6 KB

struct PowTab {
  const double *logt;   // [128][4]: 1/c, log c (hi, lo), unused
  const double *expt;   // [128][2]: 2^(i/128) (hi, lo)
};

// host: fill `t` (kPowTableDoubles doubles) -- long double (x87, 64-bit significand) carries the ~11 extra bits the lo parts need
inline void build_pow_tables(double *t) {
  const long double two43 = 8796093022208.0L;
  for (int i = 0; i < kPowLogEntries; ++i) {
    const double c = 0.5 + (i + 0.5) / 256.0;
    const double invc = 1.0 / c;
    const long double logc = -logl((long double)invc);            // log of the c that 1/c (as rounded) stands for
    const double hi = (double)(nearbyintl(logc * two43) / two43);  // a multiple of 2^-43
    t[4 * i + 0] = invc;
    t[4 * i + 1] = hi;
    t[4 * i + 2] = (double)(logc - (long double)hi);
    t[4 * i + 3] = 0.0;
  }
  double *e = t + 4 * kPowLogEntries;
  for (int i = 0; i < kPowExpEntries; ++i) {
    const long double v = exp2l((long double)i / kPowExpEntries);
    const double hi = (double)v;
    e[2 * i + 0] = hi;
    e[2 * i + 1] = (double)(v - (long double)hi);
  }
}
inline const double *host_pow_tables() {
  static const struct Holder { double t[kPowTableDoubles]; Holder() { build_pow_tables(t); } } h;
  return h.t;
}

namespace fastpow_detail {
__host__ __device__ inline uint64_t bits_of(double d) { uint64_t u; memcpy(&u, &d, 8); return u; }
__host__ __device__ inline double from_bits(uint64_t u) { double d; memcpy(&d, &u, 8); return d; }
// s + e == a + b exactly (Knuth), any magnitudes
__host__ __device__ inline void two_sum(double a, double b, double &s, double &e) {
  s = a + b;
  const double bb = s - a;
  e = (a - (s - bb)) + (b - bb);
}
}  // namespace fastpow_detail

// x^y; tables as described above (host_pow_tables() on the host, the LDS copy on the device)
__host__ __device__ inline double fpow(double x, double y, const PowTab &pt) {
#pragma clang fp contract(off)   // every fused multiply-add below is written as one; nothing else may be fused
  using namespace fastpow_detail;
  // ln2 = Ln2hi + Ln2lo, Ln2hi a multiple of 2^-43; ln2/128 = hi + lo with 32 significant bits in hi (k * hi is exact)
  const double Ln2hi = 0x1.62e42fefa38p-1, Ln2lo = 0x1.ef35793c7673p-45;
  const double InvLn2N = 0x1.71547652b82fep+7, NegLn2hiN = -0x1.62e42ffp-8, NegLn2loN = 0x1.718432a1b0e26p-42;
  const bool ok = x > 0.0 && x < INFINITY;
  double xs = x;
  int eadj = 0;
  if (x < 0x1p-1022) { xs = x * 0x1p54; eadj = -54; }              // subnormal: renormalise
  const uint64_t ix = bits_of(xs);
  const uint32_t hi32 = (uint32_t)(ix >> 32);
  const int e = (int)((hi32 >> 20) & 0x7ffu) - 1022 + eadj;         // xs = m 2^(e - eadj), m in [1/2, 1)
  const double m = from_bits((ix & 0x800fffffffffffffull) | 0x3fe0000000000000ull);
  const int i = (int)((hi32 >> 13) & 127u);
  const double invc = pt.logt[4 * i], lchi = pt.logt[4 * i + 1], lclo = pt.logt[4 * i + 2];
  const double r = fma(m, invc, -1.0);
  const double kd = (double)e;
  const double t1 = fma(kd, Ln2hi, lchi);                           // exact: both multiples of 2^-43, |t1| < 2^10
  double lo = fma(kd, Ln2lo, lclo);
  double s1, e1, s2, e2;
  two_sum(t1, r, s1, e1);
  const double r2 = r * r;
  const double r2lo = fma(r, r, -r2);
  two_sum(s1, -0.5 * r2, s2, e2);
  // r^3 (1/3 - r/4 + r^2/5 - r^3/6 + r^4/7)
  double p = fma(r, 0x1.2492492492492p-3, -0x1.5555555555555p-3);
  p = fma(r, p, 0x1.999999999999ap-3);
  p = fma(r, p, -0.25);
  p = fma(r, p, 0x1.5555555555555p-2);
  p = p * (r * r2);
  lo = (((lo + e1) + e2) + -0.5 * r2lo) + p;                        // log x = s2 + lo
  // z = y log x as ehi + elo
  const double ehi = y * s2;
  const double elo = fma(y, s2, -ehi) + y * lo;
  const double z = ehi * InvLn2N;
  // |y log2 x| beyond 2^11: the result over- or underflows whatever the fraction is -- the multiple of ln2/128 is clamped (2^18 / 128 =
  // 2048 binades: ldexp saturates to inf / 0, as the library pow does) instead of converting an arbitrarily large double to int
  const bool sat = !(fabs(z) < 0x1p18);
  const double kz = sat ? copysign(0x1p18, z) : rint(z);
  const int ki = (int)kz;
  double rr = fma(kz, NegLn2hiN, ehi);
  rr = fma(kz, NegLn2loN, rr);
  rr = sat ? 0.0 : rr + elo;
  const int idx = ki & (kPowExpEntries - 1), top = ki >> 7;
  const double thi = pt.expt[2 * idx], tlo = pt.expt[2 * idx + 1];
  // e^rr - 1 = rr + rr^2/2 + rr^3/6 + rr^4/24 + rr^5/120
  const double q2 = rr * rr;
  double q = fma(rr, 0x1.1111111111111p-7, 0x1.5555555555555p-5);
  q = fma(rr, q, 0x1.5555555555555p-3);
  q = fma(rr, q, 0.5);
  q = fma(q2, q, rr);
  const double res = ldexp(thi + fma(thi, q, tlo), top);
  // 0^y as the library pow the host laws use: 0 (y > 0), inf (y < 0), 1 (y == 0) -- the van Genuchten branch reaches it when (alpha pc)^n
  // underflows (AA = 0: kr = (1 - 0^m)^2 ... = 1, as the reference computes it)
  if (x == 0.0) return y > 0.0 ? 0.0 : (y < 0.0 ? INFINITY : 1.0);
  return ok ? res : NAN;
}

// the tables of fpow (fastpow.hpp) from global memory into the block's LDS; called by every thread of a block, first thing in
// every kernel that evaluates a retention curve or the IFC-67 density (before any thread leaves)
__device__ __forceinline__ PowTab stage_pow_tables(const double *__restrict__ g, double *lds) {
  for (int q = threadIdx.x; q < kPowTableDoubles; q += blockDim.x) lds[q] = g[q];
  __syncthreads();
  return PowTab{lds, lds + 4 * kPowLogEntries};
}
#define MPP_STAGE_POW(gtab)                      \
  __shared__ double pow_lds_[kPowTableDoubles]; \
  const PowTab pt = stage_pow_tables(gtab, pow_lds_)

}  // namespace mpp
