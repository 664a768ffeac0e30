// Per-element 7x7 transfer-map builders, written once for the gfx950 kernels.
//
// Every function cites the reference lines whose arithmetic (and operation order) it
// follows.  The header is `__host__ __device__` so that tests/harness/ can compile the
// very same source with g++ and check it against the oracle on a machine without a GPU;
// the shipped library only ever instantiates it inside HIP kernels.
//
// Matrices are 49 scalars, row-major (M[i*7+j]), addressed with compile-time offsets
// only, so the same code works on an LDS slot, on registers, and on host memory.
#pragma once

#include <math.h>
#include <stdint.h>

#if defined(__HIPCC__)
#define LYNX_HD __host__ __device__ __forceinline__
// Out-of-line on the device: the builders run once per workgroup, so what matters is their
// code size (instruction-cache misses dominated the prologue when everything was inlined:
// 50-70 KB per kernel against a 64 KB I-cache shared by two CUs), not call overhead.
#define LYNX_FN __host__ __device__ __noinline__
#else
#define LYNX_HD inline
#define LYNX_FN inline
#endif

#include "../../include/lynx_hip.h"

// The builders below update a matrix in place in LDS.  Without a compiler-level fence
// hipcc forwards every store to the later loads and keeps the whole 7x7 (98 VGPRs in
// fp64) live, which caps the occupancy of the fused streaming kernel.  The fence costs
// nothing at run time.
#if defined(__HIP_DEVICE_COMPILE__)
#define LYNX_FORGET() asm volatile("" ::: "memory")
#else
#define LYNX_FORGET() ((void)0)
#endif

namespace lynx {

// lynx/track_methods.py:9-11 (scipy 1.15 / CODATA 2022: m_e c^2 / e)
#define LYNX_REST_ENERGY 510998.9506917531
// lynx/accelerator/cavity.py:20 (physical_constants["electron mass energy equivalent in MeV"] * 1e6)
#define LYNX_ELECTRON_MASS_EV 510998.95069
#define LYNX_SPEED_OF_LIGHT 299792458.0
#define LYNX_PI 3.141592653589793

// number of scalars a step occupies in the per-sample "step table":
// 49 map entries + 8 cavity coefficients, padded to 64
#define LYNX_STEP_STRIDE 64
#define LYNX_COEF_OFFSET 49
// slots 57..60 of a cavity that is applied merged with the run in front of it (M = T_cav . T_run): the inverse
// of T_cav's (s, delta) block.  The cavity's map couples s and delta only with each other (cavity.py:311-323),
// so this 2x2 applied to components 4, 5 of M z returns the s and delta that ENTER the cavity and drive its
// kick -- 4 multiply-adds on scalars that arrive with the map, instead of two 7-term rows from a second slot
#define LYNX_ENTRY_OFFSET 57
// slot 62: the step's flags as they stood when THIS table was built (a small integer stored as T).  The
// cavity bits are whole-batch predicates of the beam energy, re-evaluated on the device before every
// build; the streaming kernel of call n reads them here and never sees call n+1's.
#define LYNX_FLAGS_OFFSET 62
// ... together with the step's kind and whether it is applied merged with the cavity behind it, so that
// the streaming kernel's step loop needs nothing but wave-uniform scalar loads from its table (reading
// `lat.steps[s]` there cost a vector load + readfirstlane per step: the pointer sits in a by-value struct
// and cannot be declared __restrict__)
#define LYNX_DESC_KIND_SHIFT 16
#define LYNX_DESC_PAIR (1 << 20)
// on the CAVITY's descriptor of a merged pair, per sample: the (s, delta) block of this sample's cavity map is too
// close to singular for its inverse to recover what entered the cavity (cavity_entry_inverse); the kernels then take
// the two from rows 4 and 5 of the run's map, parked in the run's slot, applied to the state that enters the pair
#define LYNX_DESC_ILL (1 << 21)
// slot 61 of a cavity step: sin(phi), next to the coefficient cos(phi) -- the float32 kick is evaluated from the two and
// d = -s beta0 k without the cancellation of cos(phi + d) - cos(phi) (device_cavity_kick, lynx_device.hpp)
#define LYNX_SINPHI_OFFSET 61
#define LYNX_STEP_SCALARS 62  // what a step's application may read of its row: map, coefficients, entry inverse, sin(phi)
// last step, slot 63: the beam energy behind the last step (published by the streaming kernel)
#define LYNX_ENERGY_OFFSET 63
// cavity coefficient slots (lynx/accelerator/cavity.py:141-226)
#define LYNX_C_DSCALE 0   // E*beta0 / (E_out*beta1)
#define LYNX_C_DKICK 1    // V*beta0 / (E_out*beta1)
#define LYNX_C_BK 2       // beta0 * k
#define LYNX_C_PHI 3      // phi [rad]
#define LYNX_C_COSPHI 4   // cos(phi)
#define LYNX_C_T566 5
#define LYNX_C_T556 6
#define LYNX_C_T555 7

template <typename T> LYNX_HD T t_sqrt(T x);
template <> LYNX_HD float t_sqrt<float>(float x) { return sqrtf(x); }
template <> LYNX_HD double t_sqrt<double>(double x) { return sqrt(x); }
template <typename T> LYNX_FN T t_sin(T x);
template <> LYNX_FN float t_sin<float>(float x) { return sinf(x); }
template <> LYNX_FN double t_sin<double>(double x) { return sin(x); }
template <typename T> LYNX_FN T t_cos(T x);
template <> LYNX_FN float t_cos<float>(float x) { return cosf(x); }
template <> LYNX_FN double t_cos<double>(double x) { return cos(x); }
template <typename T> LYNX_FN T t_tan(T x);
template <> LYNX_FN float t_tan<float>(float x) { return tanf(x); }
template <> LYNX_FN double t_tan<double>(double x) { return tan(x); }
template <typename T> LYNX_FN T t_sinh(T x);
template <> LYNX_FN float t_sinh<float>(float x) { return sinhf(x); }
template <> LYNX_FN double t_sinh<double>(double x) { return sinh(x); }
template <typename T> LYNX_FN T t_cosh(T x);
template <> LYNX_FN float t_cosh<float>(float x) { return coshf(x); }
template <> LYNX_FN double t_cosh<double>(double x) { return cosh(x); }
template <typename T> LYNX_FN T t_log(T x);
template <> LYNX_FN float t_log<float>(float x) { return logf(x); }
template <> LYNX_FN double t_log<double>(double x) { return log(x); }
template <typename T> LYNX_HD T t_fma(T a, T b, T c);
template <> LYNX_HD float t_fma<float>(float a, float b, float c) { return fmaf(a, b, c); }
template <> LYNX_HD double t_fma<double>(double a, double b, double c) { return fma(a, b, c); }

template <typename T> LYNX_FN void mat_identity(T* M) {
#pragma unroll 1
  for (int i = 0; i < 49; ++i) M[i] = (i % 8 == 0) ? T(1) : T(0);
}

// cos/sin of complex sqrt(k2)*L, reduced to real arithmetic
// (lynx/track_methods.py:72-79): k2 > 0 -> cos / sin(aL)/a ; k2 < 0 -> cosh / sinh(aL)/a
// with a = sqrt(|k2|).  k2 == 0: c = 1, s = s_at_zero (0/0 = NaN for the x plane, `length`
// for the y plane, track_methods.py:76-77).
template <typename T> LYNX_FN void cs_of(T k2, T L, T s_at_zero, T& c, T& s) {
  if (k2 > T(0)) {
    T a = t_sqrt(k2);
    T x = a * L;
    c = t_cos(x);
    s = t_sin(x) / a;
  } else if (k2 < T(0)) {
    T a = t_sqrt(-k2);
    T x = a * L;
    c = t_cosh(x);
    s = t_sinh(x) / a;
  } else {
    c = T(1);
    s = s_at_zero;
  }
}

// In-place  M <- rot(angle) . M   (rotation_matrix: lynx/track_methods.py:14-34).
// Only rows 0..3 change; term order = ascending k of the dense product.
template <typename T> LYNX_FN void rot_left(T* M, T cs, T sn) {
#pragma unroll 1
  for (int j = 0; j < 7; ++j) {
    LYNX_FORGET();
    T r0 = M[0 * 7 + j], r1 = M[1 * 7 + j], r2 = M[2 * 7 + j], r3 = M[3 * 7 + j];
    M[0 * 7 + j] = t_fma(sn, r2, cs * r0);
    M[1 * 7 + j] = t_fma(sn, r3, cs * r1);
    M[2 * 7 + j] = t_fma(cs, r2, (-sn) * r0);
    M[3 * 7 + j] = t_fma(cs, r3, (-sn) * r1);
  }
}

// In-place  M <- M . rot(angle).  Only columns 0..3 change.
template <typename T> LYNX_FN void rot_right(T* M, T cs, T sn) {
#pragma unroll 1
  for (int i = 0; i < 7; ++i) {
    LYNX_FORGET();
    T c0 = M[i * 7 + 0], c1 = M[i * 7 + 1], c2 = M[i * 7 + 2], c3 = M[i * 7 + 3];
    M[i * 7 + 0] = t_fma(c2, -sn, c0 * cs);
    M[i * 7 + 1] = t_fma(c3, -sn, c1 * cs);
    M[i * 7 + 2] = t_fma(c2, cs, c0 * sn);
    M[i * 7 + 3] = t_fma(c3, cs, c1 * sn);
  }
}

// lynx/accelerator/drift.py:44-62 (also the body of both correctors)
template <typename T> LYNX_FN void build_drift(T L, T energy, T* M) {
  T gamma = energy / T(LYNX_REST_ENERGY);
  T igamma2 = T(0);  // zeros where gamma == 0 (drift.py:53)
  if (gamma != T(0)) igamma2 = T(1) / (gamma * gamma);
  T beta = t_sqrt(T(1) - igamma2);
  mat_identity(M);
  M[0 * 7 + 1] = L;
  M[2 * 7 + 3] = L;
  M[4 * 7 + 5] = -L / (beta * beta) * igamma2;
}

// The scalars of lynx/track_methods.py:37-99 (`base_rmatrix` without the tilt rotation): what both the 7x7 builder below
// and the 16-entry builder of structured maps (build_entries_u) put into their matrices.
template <typename T> struct BaseScalars {
  T cx, sx, cy, sy, dx, r56, kx2, ky2, beta;
};
template <typename T> LYNX_HD void base_rmatrix_scalars(T L, T k1, T hx, T energy, BaseScalars<T>& o) {
  T gamma = energy / T(LYNX_REST_ENERGY);
  T igamma2 = T(1);  // ones where gamma == 0 (track_methods.py:61)
  if (gamma != T(0)) igamma2 = T(1) / (gamma * gamma);
  T beta = t_sqrt(T(1) - igamma2);

  if (k1 == T(0)) k1 = T(1e-12);  // :67-68

  T kx2 = k1 + hx * hx;
  T ky2 = -k1;
  T cx, sx, cy, sy;
  cs_of(kx2, L, T(0) / T(0) * L, cx, sx);  // kx == 0: sin(0)/0 -> NaN (:79)
  cs_of(ky2, L, L, cy, sy);                // ky == 0: sy = length   (:76)
  T dx = hx / kx2 * (T(1) - cx);
  T r56 = hx * hx * (L - sx) / kx2 / (beta * beta);
  r56 = r56 - L / (beta * beta) * igamma2;
  o.cx = cx; o.sx = sx; o.cy = cy; o.sy = sy; o.dx = dx; o.r56 = r56; o.kx2 = kx2; o.ky2 = ky2; o.beta = beta;
}

// lynx/track_methods.py:37-99 without the tilt rotation (applied by the caller)
template <typename T> LYNX_FN void build_base_rmatrix(T L, T k1, T hx, T energy, T* M) {
  BaseScalars<T> b;
  base_rmatrix_scalars(L, k1, hx, energy, b);
  const T cx = b.cx, sx = b.sx, cy = b.cy, sy = b.sy, dx = b.dx, r56 = b.r56, kx2 = b.kx2, ky2 = b.ky2, beta = b.beta;

  mat_identity(M);
  M[0 * 7 + 0] = cx;
  M[0 * 7 + 1] = sx;
  M[0 * 7 + 5] = dx / beta;
  M[1 * 7 + 0] = -kx2 * sx;
  M[1 * 7 + 1] = cx;
  M[1 * 7 + 5] = sx * hx / beta;
  M[2 * 7 + 2] = cy;
  M[2 * 7 + 3] = sy;
  M[3 * 7 + 2] = -ky2 * sy;
  M[3 * 7 + 3] = cy;
  M[4 * 7 + 0] = sx * hx / beta;
  M[4 * 7 + 1] = dx / beta;
  M[4 * 7 + 5] = r56;
}

// lynx/accelerator/quadrupole.py:66-80.  p = [L, k1, tilt, mx, my]
template <typename T> LYNX_FN void build_quadrupole(const T* p, int flags, T energy, T* M) {
  build_base_rmatrix(p[0], p[1], T(0), energy, M);
  if (flags & LYNX_FLAG_TILT) {  // any(tilt != 0) over the batch (track_methods.py:101-104)
    T tilt = p[2];
    // (rot(-tilt) . R) . rot(tilt)
    rot_left(M, t_cos(-tilt), t_sin(-tilt));
    rot_right(M, t_cos(tilt), t_sin(tilt));
  }
  if (flags & LYNX_FLAG_MISALIGNED) {  // not all(misalignment == 0) (quadrupole.py:75-80)
    T mx = p[3], my = p[4];
    // R_exit . R : rows 0 and 2 pick up the constant row 6 (track_methods.py:114-116)
#pragma unroll 1
    for (int j = 0; j < 7; ++j) {
      LYNX_FORGET();
      T r6 = M[6 * 7 + j];
      M[0 * 7 + j] = t_fma(mx, r6, M[0 * 7 + j]);
      M[2 * 7 + j] = t_fma(my, r6, M[2 * 7 + j]);
    }
    // (.) . R_entry : column 6 (track_methods.py:118-120)
#pragma unroll 1
    for (int i = 0; i < 7; ++i) {
      LYNX_FORGET();
      T acc = M[i * 7 + 0] * (-mx);
      acc = t_fma(M[i * 7 + 2], -my, acc);
      M[i * 7 + 6] = acc + M[i * 7 + 6];
    }
  }
}

// lynx/accelerator/dipole.py:96-181 (RBend's e1/e2 shift, rbend.py:79-80, is applied by
// the host when the element is constructed, as in the reference).
// p = [L, angle, e1, e2, tilt, fint, fintx, gap]
template <typename T> LYNX_FN void build_dipole(const T* p, int flags, T energy, T* M) {
  T L = p[0], angle = p[1], e1 = p[2], e2 = p[3], tilt = p[4];
  T fint = p[5], fintx = p[6], gap = p[7];
  T hx = T(0);
  if (L != T(0)) hx = angle / L;  // dipole.py:96-102

  if (flags & LYNX_FLAG_THICK) {  // any(length != 0) over the batch (dipole.py:119)
    build_base_rmatrix(L, T(0), hx, energy, M);
  } else {  // dipole.py:127-133
    mat_identity(M);
    M[0 * 7 + 1] = L;
    M[2 * 7 + 6] = angle;
    M[2 * 7 + 3] = L;
  }
  // edges, dipole.py:143-181
  T sec1 = T(1) / t_cos(e1);
  T s1 = t_sin(e1);
  T phi1 = fint * hx * gap * sec1 * (T(1) + s1 * s1);
  T a1 = hx * t_tan(e1);
  T b1 = -hx * t_tan(e1 - phi1);
  T sec2 = T(1) / t_cos(e2);
  T s2 = t_sin(e2);
  T phi2 = fintx * hx * gap * sec2 * (T(1) + s2 * s2);
  T a2 = hx * t_tan(e2);
  T b2 = -hx * t_tan(e2 - phi2);
  // R . R_enter : columns 0 and 2 (dipole.py:136 inner product)
#pragma unroll 1
  for (int i = 0; i < 7; ++i) {
    LYNX_FORGET();
    M[i * 7 + 0] = t_fma(M[i * 7 + 1], a1, M[i * 7 + 0]);
    M[i * 7 + 2] = t_fma(M[i * 7 + 3], b1, M[i * 7 + 2]);
  }
  // R_exit . (.) : rows 1 and 3
#pragma unroll 1
  for (int j = 0; j < 7; ++j) {
    LYNX_FORGET();
    M[1 * 7 + j] = a2 * M[0 * 7 + j] + M[1 * 7 + j];
    M[3 * 7 + j] = b2 * M[2 * 7 + j] + M[3 * 7 + j];
  }
  // rot(-tilt) . (R . rot(tilt)), always (dipole.py:138-140)
  LYNX_FORGET();
  rot_right(M, t_cos(tilt), t_sin(tilt));
  rot_left(M, t_cos(-tilt), t_sin(-tilt));
}

// lynx/accelerator/horizontal_corrector.py:52-67 / vertical_corrector.py:52-66. p = [L, angle]
template <typename T> LYNX_FN void build_corrector(const T* p, bool vertical, T energy, T* M) {
  build_drift(p[0], energy, M);
  M[(vertical ? 3 : 1) * 7 + 6] = p[1];
}

// The bracket of r55_cor (cavity.py:296-305), `g0 g1 (beta0 beta1 - 1) + 1`, as the reference evaluates it.  On an
// ultra-relativistic beam it cancels twice -- beta0 beta1 - 1 ~ -4e-5 at gamma ~ 150-180, times g0 g1 = 27 000 is
// -1.017, plus 1 --: in float32 the VALUE is off by 20-30 % there (r55_cor itself is ~1e-5 next to the 1 of M[4][4], so
// the forward pass does not notice: it repeats the reference's operations, this template).  A derivative taken through
// the same operations is off by as much, and it multiplies the whole cotangent of M[4][4]; the dual-number instantiation
// (lynx_dual.hpp) therefore specialises this function with an algebraically equal form that does not cancel.
template <typename T> LYNX_HD T cavity_r55_bracket(T g0, T g1, T beta0, T beta1) {
  return g0 * g1 * (beta0 * beta1 - T(1)) + T(1);
}

// lynx/accelerator/cavity.py:248-325 (`_cavity_rmatrix`) plus, when `coef` is non-null,
// the per-sample coefficients of the non-linear step (cavity.py:97-246, `_track_beam`).
// p = [L, V, phase_deg, f].  Returns the outgoing energy.
template <typename T> struct CavityScalars {
  T r11, r12, r21, r22, r55_cor, r56, r65, r66;
};
template <typename T> LYNX_HD T cavity_scalars(const T* p, int flags, T energy, CavityScalars<T>& o, T* coef) {
  const T me = T(LYNX_ELECTRON_MASS_EV);
  T L = p[0], V = p[1], f = p[3];
  T phi = p[2] * T(LYNX_PI / 180.0);  // deg2rad
  T cphi = t_cos(phi), sphi = t_sin(phi);
  T delta_energy = V * cphi;
  T Ei = energy / me;
  T Ef = (energy + delta_energy) / me;
  T Ep = (Ef - Ei) / L;

  T alpha = t_sqrt(T(1) / T(8)) / cphi * t_log(Ef / Ei);
  T ca = t_cos(alpha), sa = t_sin(alpha);
  T r11 = ca - t_sqrt(T(2)) * cphi * sa;
  T r12 = t_sqrt(T(8)) * Ei / Ep * cphi * sa;  // V == 0 -> inf*0 = NaN, as the reference
  T r21 = -Ep / Ef * (cphi / t_sqrt(T(2)) + t_sqrt(T(1) / T(8)) / cphi) * sa;
  T r22 = Ei / Ef * (ca + t_sqrt(T(2)) * cphi * sa);

  T r56 = T(0), beta0 = T(1), beta1 = T(1), r55_cor = T(0);
  T k = T(2) * T(LYNX_PI) * f / T(LYNX_SPEED_OF_LIGHT);
  if (flags & LYNX_FLAG_CAV_BETA) {  // any((V != 0) & (E != 0)) over the batch (cavity.py:290)
    beta0 = t_sqrt(T(1) - T(1) / (Ei * Ei));
    beta1 = t_sqrt(T(1) - T(1) / (Ef * Ef));
    r56 = -L / (Ef * Ef * Ei * beta1) * (Ef + Ei) / (beta1 + beta0);
    T g0 = Ei, g1 = Ef;
    r55_cor = k * L * beta0 * V / me * sphi * cavity_r55_bracket<T>(g0, g1, beta0, beta1) /
              (beta1 * g1 * ((g0 - g1) * (g0 - g1)));
  }
  T r66 = Ei / Ef * beta0 / beta1;
  T r65 = k * sphi * V / (Ef * beta1 * me);
  o.r11 = r11; o.r12 = r12; o.r21 = r21; o.r22 = r22; o.r55_cor = r55_cor; o.r56 = r56; o.r65 = r65; o.r66 = r66;

  T energy_out = energy;
  if (coef) {
    // `_track_beam` prologue, cavity.py:101-111
    T tb0 = T(1), tig2 = T(0), tg0 = T(1e10);
    if (energy != T(0)) {
      tg0 = energy / me;
      tig2 = T(1) / (tg0 * tg0);
      tb0 = t_sqrt(T(1) - tig2);
    }
    T T566 = T(1.5) * L * tig2 / (tb0 * tb0 * tb0);  // :124
    T T556 = T(0), T555 = T(0);
    T dscale = T(1), dkick = T(0), bk = T(0);
    int gain = 0;
    if (flags & LYNX_FLAG_CAV_GAIN) {  // any(E + dE > 0) over the batch (cavity.py:128)
      gain = 1;
      energy_out = energy + delta_energy;
      T g1 = energy_out / me;
      T tb1 = t_sqrt(T(1) - T(1) / (g1 * g1));
      dscale = energy * tb0 / (energy_out * tb1);  // cavity.py:141-143
      dkick = V * tb0 / (energy_out * tb1);        // :146-148
      bk = tb0 * k;                                // :150-156
      T dgamma = V / me;
      if (flags & LYNX_FLAG_CAV_T5XX) {  // any(dE > 0) over the batch (cavity.py:164)
        T b03 = tb0 * tb0 * tb0, b13 = tb1 * tb1 * tb1;
        T g03 = tg0 * tg0 * tg0, g13 = g1 * g1 * g1;
        T dg = tg0 - g1;
        T566 = L * (b03 * g03 - b13 * g13) / (T(2) * tb0 * b13 * tg0 * dg * g13);
        T556 = tb0 * k * L * dgamma * tg0 * (b13 * g13 + tb0 * (tg0 - g13)) * sphi /
               (b13 * g13 * (dg * dg));
        T555 = tb0 * tb0 * (k * k) * L * dgamma / T(2) *
               (dgamma * (T(2) * tg0 * g13 * (tb0 * b13 - T(1)) + tg0 * tg0 + T(3) * (g1 * g1) - T(2)) /
                    (b13 * g13 * (dg * dg * dg)) * (sphi * sphi) -
                (g1 * tg0 * (tb1 * tb0 - T(1)) + T(1)) / (tb1 * g1 * (dg * dg)) * cphi);
      }
    }
    coef[LYNX_C_DSCALE] = dscale;
    coef[LYNX_C_DKICK] = dkick;
    coef[LYNX_C_BK] = bk;
    coef[LYNX_C_PHI] = phi;
    coef[LYNX_C_COSPHI] = cphi;
    // without the GAIN branch the reference adds no second-order term at all
    coef[LYNX_C_T566] = gain ? T566 : T(0);
    coef[LYNX_C_T556] = gain ? T556 : T(0);
    coef[LYNX_C_T555] = gain ? T555 : T(0);
  }
  return energy_out;
}
template <typename T> LYNX_FN T build_cavity(const T* p, int flags, T energy, T* M, T* coef) {
  CavityScalars<T> c;
  const T energy_out = cavity_scalars(p, flags, energy, c, coef);
  mat_identity(M);
  M[0 * 7 + 0] = c.r11;
  M[0 * 7 + 1] = c.r12;
  M[1 * 7 + 0] = c.r21;
  M[1 * 7 + 1] = c.r22;
  M[2 * 7 + 2] = c.r11;
  M[2 * 7 + 3] = c.r12;
  M[3 * 7 + 2] = c.r21;
  M[3 * 7 + 3] = c.r22;
  M[4 * 7 + 4] = T(1) + c.r55_cor;
  M[4 * 7 + 5] = c.r56;
  M[5 * 7 + 4] = c.r65;
  M[5 * 7 + 5] = c.r66;
  return energy_out;
}

// lynx/accelerator/solenoid.py:61-105.  p = [L, k, mx, my]
template <typename T> LYNX_FN void build_solenoid(const T* p, int flags, T energy, T* M) {
  const T L = p[0], k = p[1];
  const T gamma = energy / T(LYNX_REST_ENERGY);
  const T c = t_cos(L * k), s = t_sin(L * k);
  T s_k = L;
  if (k != T(0)) s_k = s / k;  // :68-70
  T r56 = T(0);
  if (gamma != T(0)) {  // :73-76
    const T gamma2 = gamma * gamma;
    const T beta = t_sqrt(T(1) - T(1) / gamma2);
    r56 = r56 - L / (beta * beta * gamma2);
  }
  mat_identity(M);
  M[0 * 7 + 0] = c * c;
  M[0 * 7 + 1] = c * s_k;
  M[0 * 7 + 2] = s * c;
  M[0 * 7 + 3] = s * s_k;
  M[1 * 7 + 0] = -k * s * c;
  M[1 * 7 + 1] = c * c;
  M[1 * 7 + 2] = -k * (s * s);
  M[1 * 7 + 3] = s * c;
  M[2 * 7 + 0] = -s * c;
  M[2 * 7 + 1] = -s * s_k;
  M[2 * 7 + 2] = c * c;
  M[2 * 7 + 3] = c * s_k;
  M[3 * 7 + 0] = k * (s * s);
  M[3 * 7 + 1] = -s * c;
  M[3 * 7 + 2] = -k * s * c;
  M[3 * 7 + 3] = c * c;
  M[4 * 7 + 5] = r56;
  if (flags & LYNX_FLAG_MISALIGNED) {  // :98-102, same sandwich as the quadrupole's
    const T mx = p[2], my = p[3];
#pragma unroll 1
    for (int j = 0; j < 7; ++j) {
      LYNX_FORGET();
      const T r6 = M[6 * 7 + j];
      M[0 * 7 + j] = t_fma(mx, r6, M[0 * 7 + j]);
      M[2 * 7 + j] = t_fma(my, r6, M[2 * 7 + j]);
    }
#pragma unroll 1
    for (int i = 0; i < 7; ++i) {
      LYNX_FORGET();
      T acc = M[i * 7 + 0] * (-mx);
      acc = t_fma(M[i * 7 + 2], -my, acc);
      M[i * 7 + 6] = acc + M[i * 7 + 6];
    }
  }
}

// lynx/accelerator/undulator.py:48-60 (drift-like; note R56 = +L / gamma^2, no beta). p = [L]
template <typename T> LYNX_FN void build_undulator(T L, T energy, T* M) {
  const T gamma = energy / T(LYNX_REST_ENERGY);
  T igamma2 = T(0);
  if (gamma != T(0)) igamma2 = T(1) / (gamma * gamma);
  mat_identity(M);
  M[0 * 7 + 1] = L;
  M[2 * 7 + 3] = L;
  M[4 * 7 + 5] = L * igamma2;
}

// The map of an element whose kind and whole-batch flags give it the structure of class U (lynx_unit_record.hpp: rows
// 0, 1 <- columns {0, 1, 6}; rows 2, 3 <- {2, 3, 6}; rows 4, 5 <- {4, 5}; row 6 = e6) as its 16 entries, in the order of
// unit_entry_u(), straight into registers -- no 7x7 in memory.  Same scalars, same operations as the builders above
// (the misaligned quadrupole's sandwich, quadrupole.py:75-80, written out for the entries it can reach).  Returns false
// for every other kind (dipoles, tilted quadrupoles, solenoids, custom maps, the helper kinds): those take
// build_element.  Used where a map is needed per (element, parameter) and a 49-entry array per lane would live in
// scratch memory: the dual-number evaluation of k_build_bwd.
template <typename T>
LYNX_HD bool build_entries_u(int kind, int flags, const T* p, T energy, T (&m)[16], T* coef) {
#pragma unroll
  for (int k = 0; k < 16; ++k) m[k] = T(0);
  m[0] = T(1);   // [0][0]
  m[4] = T(1);   // [1][1]
  m[6] = T(1);   // [2][2]
  m[10] = T(1);  // [3][3]
  m[12] = T(1);  // [4][4]
  m[15] = T(1);  // [5][5]
  switch (kind) {
    case LYNX_KIND_IDENTITY: return true;
    case LYNX_KIND_DRIFT:
    case LYNX_KIND_HCOR:
    case LYNX_KIND_VCOR: {  // drift.py:44-62; *_corrector.py:52-67
      const T L = p[0];
      T gamma = energy / T(LYNX_REST_ENERGY);
      T igamma2 = T(0);
      if (gamma != T(0)) igamma2 = T(1) / (gamma * gamma);
      T beta = t_sqrt(T(1) - igamma2);
      m[1] = L;                                // [0][1]
      m[7] = L;                                // [2][3]
      m[13] = -L / (beta * beta) * igamma2;    // [4][5]
      if (kind == LYNX_KIND_HCOR) m[5] = p[1];   // [1][6]
      if (kind == LYNX_KIND_VCOR) m[11] = p[1];  // [3][6]
      return true;
    }
    case LYNX_KIND_UNDULATOR: {  // undulator.py:48-60
      const T L = p[0];
      const T gamma = energy / T(LYNX_REST_ENERGY);
      T igamma2 = T(0);
      if (gamma != T(0)) igamma2 = T(1) / (gamma * gamma);
      m[1] = L;
      m[7] = L;
      m[13] = L * igamma2;
      return true;
    }
    case LYNX_KIND_QUADRUPOLE: {
      if (flags & LYNX_FLAG_TILT) return false;
      BaseScalars<T> b;
      base_rmatrix_scalars(p[0], p[1], T(0), energy, b);
      m[0] = b.cx;
      m[1] = b.sx;
      m[3] = -b.kx2 * b.sx;
      m[4] = b.cx;
      m[6] = b.cy;
      m[7] = b.sy;
      m[9] = -b.ky2 * b.sy;
      m[10] = b.cy;
      m[13] = b.r56;
      if (flags & LYNX_FLAG_MISALIGNED) {  // R_exit . R . R_entry (quadrupole.py:75-80; track_methods.py:108-122)
        const T mx = p[3], my = p[4];
        m[2] = (m[0] * (-mx)) + mx;   // [0][6]
        m[5] = (m[3] * (-mx));        // [1][6]
        m[8] = (m[6] * (-my)) + my;   // [2][6]
        m[11] = (m[9] * (-my));       // [3][6]
      }
      return true;
    }
    case LYNX_KIND_CAVITY: {
      CavityScalars<T> c;
      (void)cavity_scalars(p, flags, energy, c, coef);
      m[0] = c.r11;
      m[1] = c.r12;
      m[3] = c.r21;
      m[4] = c.r22;
      m[6] = c.r11;
      m[7] = c.r12;
      m[9] = c.r21;
      m[10] = c.r22;
      m[12] = T(1) + c.r55_cor;
      m[13] = c.r56;
      m[14] = c.r65;
      m[15] = c.r66;
      return true;
    }
    default: return false;
  }
}

// One element -> M (49 scalars).  `p` points at the element's parameter row of this
// sample.  Cavity coefficients are produced only for cavity *steps* (coef != nullptr).
template <typename T>
LYNX_FN void build_element(int kind, int flags, const T* p, T energy, T* M, T* coef) {
  switch (kind) {
    case LYNX_KIND_DRIFT: build_drift(p[0], energy, M); break;
    case LYNX_KIND_QUADRUPOLE: build_quadrupole(p, flags, energy, M); break;
    case LYNX_KIND_DIPOLE: build_dipole(p, flags, energy, M); break;
    case LYNX_KIND_HCOR: build_corrector(p, false, energy, M); break;
    case LYNX_KIND_VCOR: build_corrector(p, true, energy, M); break;
    case LYNX_KIND_CAVITY: build_cavity(p, flags, energy, M, coef); break;
    case LYNX_KIND_BASE_RMATRIX:  // track_methods.py:37-105, p = [L, k1, hx, tilt]
      build_base_rmatrix(p[0], p[1], p[2], energy, M);
      if (flags & LYNX_FLAG_TILT) {
        rot_left(M, t_cos(-p[3]), t_sin(-p[3]));
        rot_right(M, t_cos(p[3]), t_sin(p[3]));
      }
      break;
    case LYNX_KIND_ROTATION: {  // track_methods.py:14-34
      const T cs = t_cos(p[0]), sn = t_sin(p[0]);
      mat_identity(M);
      M[0 * 7 + 0] = cs;
      M[0 * 7 + 2] = sn;
      M[1 * 7 + 1] = cs;
      M[1 * 7 + 3] = sn;
      M[2 * 7 + 0] = -sn;
      M[2 * 7 + 2] = cs;
      M[3 * 7 + 1] = -sn;
      M[3 * 7 + 3] = cs;
      break;
    }
    case LYNX_KIND_MISALIGNMENT:  // track_methods.py:108-122
      mat_identity(M);
      M[0 * 7 + 6] = p[2] * p[0];
      M[2 * 7 + 6] = p[2] * p[1];
      break;
    case LYNX_KIND_SOLENOID: build_solenoid(p, flags, energy, M); break;
    case LYNX_KIND_UNDULATOR: build_undulator(p[0], energy, M); break;
    case LYNX_KIND_CUSTOM:
#pragma unroll 1
      for (int i = 0; i < 49; ++i) M[i] = p[i];  // custom_transfer_map.py:87-88
      break;
    default: mat_identity(M); break;  // marker.py:32-35, bpm.py:43-46
  }
}

// Non-linear cavity step on one particle, cavity.py:141-161 and :219-226.
// `lin` = T_cav . z_in (already applied), `s_in`/`d_in` = incoming s and delta.
template <typename T>
LYNX_HD void cavity_kick(const T* coef, T s_in, T d_in, T& s_out, T& d_out) {
  d_out = d_in * coef[LYNX_C_DSCALE] +
          coef[LYNX_C_DKICK] * (t_cos(T(-1) * s_in * coef[LYNX_C_BK] + coef[LYNX_C_PHI]) - coef[LYNX_C_COSPHI]);
  s_out = s_out + (coef[LYNX_C_T566] * (d_in * d_in) + coef[LYNX_C_T556] * s_in * d_in +
                   coef[LYNX_C_T555] * (s_in * s_in));
}

}  // namespace lynx
