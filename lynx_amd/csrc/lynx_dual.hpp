// Forward-mode dual numbers for the map builders.
//
// The reverse pass of the gradient (k_build_bwd) needs dM/dtheta for every element
// parameter.  Instead of hand-deriving the partials of base_rmatrix / dipole edges / the
// cavity's T566..T555 terms, the builders of lynx_maps.hpp are instantiated a second time
// with T = Dual<float|double>: one seeded evaluation per (element, parameter) yields the
// exact derivative of all 49 map entries and of the 8 cavity coefficients.
// Comparisons act on the value part, so the builders take the same branches as the forward
// pass (including `k1 == 0 -> 1e-12`, whose derivative is 0 exactly as in an autograd trace
// of the reference).
#pragma once

#include "lynx_maps.hpp"

namespace lynx {

template <typename T> struct Dual {
  T v, d;
  LYNX_HD Dual() {}
  template <typename U> LYNX_HD Dual(U x) : v(T(x)), d(T(0)) {}
  LYNX_HD Dual(T v_, T d_) : v(v_), d(d_) {}
};

template <typename T> LYNX_HD Dual<T> operator+(Dual<T> a, Dual<T> b) { return Dual<T>(a.v + b.v, a.d + b.d); }
template <typename T> LYNX_HD Dual<T> operator-(Dual<T> a, Dual<T> b) { return Dual<T>(a.v - b.v, a.d - b.d); }
template <typename T> LYNX_HD Dual<T> operator-(Dual<T> a) { return Dual<T>(-a.v, -a.d); }
template <typename T> LYNX_HD Dual<T> operator*(Dual<T> a, Dual<T> b) {
  return Dual<T>(a.v * b.v, a.d * b.v + a.v * b.d);
}
template <typename T> LYNX_HD Dual<T> operator/(Dual<T> a, Dual<T> b) {
  const T q = a.v / b.v;
  return Dual<T>(q, (a.d - q * b.d) / b.v);
}
template <typename T> LYNX_HD bool operator==(Dual<T> a, Dual<T> b) { return a.v == b.v; }
template <typename T> LYNX_HD bool operator!=(Dual<T> a, Dual<T> b) { return a.v != b.v; }
template <typename T> LYNX_HD bool operator>(Dual<T> a, Dual<T> b) { return a.v > b.v; }
template <typename T> LYNX_HD bool operator<(Dual<T> a, Dual<T> b) { return a.v < b.v; }

// d/dx: sqrt -> 1/(2 sqrt x); sin -> cos; cos -> -sin; tan -> 1 + tan^2; sinh -> cosh; cosh -> sinh; log -> 1/x.
// INLINE, on the library functions themselves: the plain t_sin / t_cos / ... of lynx_maps.hpp are out-of-line on purpose
// (the forward builders' code size), but a dual-number builder makes dozens of such calls, and every call on this
// target saves and restores the caller's live registers through scratch memory -- that, not the arithmetic, was what
// k_build_bwd's parameter-gradient phase spent its time on.
LYNX_HD float dual_sin(float x) { return sinf(x); }
LYNX_HD double dual_sin(double x) { return sin(x); }
LYNX_HD float dual_cos(float x) { return cosf(x); }
LYNX_HD double dual_cos(double x) { return cos(x); }
LYNX_HD float dual_tan(float x) { return tanf(x); }
LYNX_HD double dual_tan(double x) { return tan(x); }
LYNX_HD float dual_sinh(float x) { return sinhf(x); }
LYNX_HD double dual_sinh(double x) { return sinh(x); }
LYNX_HD float dual_cosh(float x) { return coshf(x); }
LYNX_HD double dual_cosh(double x) { return cosh(x); }
LYNX_HD float dual_log(float x) { return logf(x); }
LYNX_HD double dual_log(double x) { return log(x); }
#define LYNX_DUAL_ALL(R)                                                                                                   \
  template <> LYNX_HD Dual<R> t_sqrt<Dual<R>>(Dual<R> x) { const R f = t_sqrt<R>(x.v); return Dual<R>(f, (R(0.5) / f) * x.d); } \
  template <> LYNX_HD Dual<R> t_sin<Dual<R>>(Dual<R> x) { return Dual<R>(dual_sin(x.v), dual_cos(x.v) * x.d); }             \
  template <> LYNX_HD Dual<R> t_cos<Dual<R>>(Dual<R> x) { return Dual<R>(dual_cos(x.v), -dual_sin(x.v) * x.d); }            \
  template <> LYNX_HD Dual<R> t_tan<Dual<R>>(Dual<R> x) { const R f = dual_tan(x.v); return Dual<R>(f, (R(1) + f * f) * x.d); } \
  template <> LYNX_HD Dual<R> t_sinh<Dual<R>>(Dual<R> x) { return Dual<R>(dual_sinh(x.v), dual_cosh(x.v) * x.d); }          \
  template <> LYNX_HD Dual<R> t_cosh<Dual<R>>(Dual<R> x) { return Dual<R>(dual_cosh(x.v), dual_sinh(x.v) * x.d); }          \
  template <> LYNX_HD Dual<R> t_log<Dual<R>>(Dual<R> x) { return Dual<R>(dual_log(x.v), (R(1) / x.v) * x.d); }
LYNX_DUAL_ALL(float)
LYNX_DUAL_ALL(double)
#undef LYNX_DUAL_ALL

// cs_of (lynx_maps.hpp) for dual numbers, inline for the same reason
template <typename R> LYNX_HD void cs_of(Dual<R> k2, Dual<R> L, Dual<R> s_at_zero, Dual<R>& c, Dual<R>& s) {
  if (k2.v > R(0)) {
    const Dual<R> a = t_sqrt(k2), x = a * L;
    c = t_cos(x);
    s = t_sin(x) / a;
  } else if (k2.v < R(0)) {
    const Dual<R> a = t_sqrt(-k2), x = a * L;
    c = t_cosh(x);
    s = t_sinh(x) / a;
  } else {
    c = Dual<R>(R(1));
    s = s_at_zero;
  }
}

// cavity_r55_bracket (lynx_maps.hpp) without its cancellations: with a = 1/g0^2, b = 1/g1^2,
//   beta0 beta1 - 1 = ((1 - a)(1 - b) - 1) / (beta0 beta1 + 1) = -(a + b - a b) / (1 + beta0 beta1)          =: -eps
//   g0 g1 (a + b - a b) = g1/g0 + g0/g1 - 1/(g0 g1) = 2 + ((g0 - g1)^2 - 1) / (g0 g1)                          =: 2 + u
//   bracket = 1 - (2 + u) / (1 + beta0 beta1) = ((2 - eps) - (2 + u)) / (2 - eps) = -(eps + u) / (1 + beta0 beta1)
// eps and u are small and (for an energy change of more than one rest mass) of one sign.
template <typename R> LYNX_HD Dual<R> cavity_r55_bracket_dual(Dual<R> g0, Dual<R> g1, Dual<R> beta0, Dual<R> beta1) {
  const Dual<R> one(R(1));
  const Dual<R> a = one / (g0 * g0), b = one / (g1 * g1), sum = one + beta0 * beta1;
  const Dual<R> eps = (a + b - a * b) / sum, dg = g0 - g1, u = (dg * dg - one) / (g0 * g1);
  return -(eps + u) / sum;
}
template <> LYNX_HD Dual<float> cavity_r55_bracket<Dual<float>>(Dual<float> g0, Dual<float> g1, Dual<float> b0, Dual<float> b1) {
  return cavity_r55_bracket_dual<float>(g0, g1, b0, b1);
}
template <> LYNX_HD Dual<double> cavity_r55_bracket<Dual<double>>(Dual<double> g0, Dual<double> g1, Dual<double> b0, Dual<double> b1) {
  return cavity_r55_bracket_dual<double>(g0, g1, b0, b1);
}

template <> LYNX_HD Dual<float> t_fma<Dual<float>>(Dual<float> a, Dual<float> b, Dual<float> c) { return a * b + c; }
template <> LYNX_HD Dual<double> t_fma<Dual<double>>(Dual<double> a, Dual<double> b, Dual<double> c) { return a * b + c; }

}  // namespace lynx
