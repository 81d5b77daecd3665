// Integer power used for Power{N} nodes; one definition for host and device so
// that every checker and every kernel runs the same instruction sequence.
//
// Reference: the generated Julia code evaluates `(g)^N` (src/backend/static.jl:45).
// Julia lowers a literal exponent through Base.literal_pow: N = 2 -> x*x,
// N = 3 -> x*x*x, N = -1 -> inv(x), N = -2 -> inv(x)^2; any other N reaches
// Base.Math.pow_body(x::Float64, n::Integer) -- Julia Base >= 1.8 (the
// reference's CI runs Julia 1.9; Julia Base is not part of /root/reference), a
// power-by-squaring with a compensated low word.  That published algorithm is
// restated below with fma() wherever Julia writes muladd/fma.  For |N| >= 4 the
// reference itself is not bit-pinned across Julia versions.
#pragma once
#include <cmath>
#include <cstdint>

#if defined(__HIPCC__) || defined(__HIP__)
#define FDG_HD __host__ __device__
#else
#define FDG_HD
#endif

FDG_HD inline double fdg_powi_impl(double x, int32_t n) {
  if (n == 2) return x * x;
  if (n == 3) return x * x * x;
  if (n == -1) return 1.0 / x;
  if (n == -2) { double r = 1.0 / x; return r * r; }
  if (n == 0) return 1.0;
  if (n == 1) return x;
  double y = 1.0, xnlo = 0.0, ynlo = 0.0;
  int64_t m = n;
  if (m < 0) {
    double rx = 1.0 / x;
    if (std::isfinite(x)) xnlo = -std::fma(x, rx, -1.0) * rx;
    x = rx;
    m = -m;
  }
  while (m > 1) {
    if (m & 1) {
      double err = std::fma(y, xnlo, x * ynlo);
      double yh = x * y;
      double yl = std::fma(x, y, -yh);
      y = yh;
      ynlo = yl + err;
    }
    double err = x * 2 * xnlo;
    double xh = x * x;
    double xl = std::fma(x, x, -xh);
    x = xh;
    xnlo = xl + err;
    m >>= 1;
  }
  double err = std::fma(y, xnlo, x * ynlo);
  return (std::isfinite(x) && std::isfinite(err)) ? std::fma(x, y, err) : x * y;
}
