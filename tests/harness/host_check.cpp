// TEST INFRASTRUCTURE ONLY -- not part of lynx_amd, never loaded by it.
//
// Compiles the kernels' map-builder source (lynx_amd/csrc/lynx_maps.hpp) for the host so
// that the CPU-only test suite can check the arithmetic the GPU will run against the
// oracle before any GPU time is spent.  The shipped library has no host path.
#include "../../lynx_amd/csrc/lynx_maps.hpp"

template <typename T>
static void build(int kind, int flags, const T* p, T energy, T* M, T* coef, int want_coef) {
  lynx::build_element<T>(kind, flags, p, energy, M, want_coef ? coef : nullptr);
}

extern "C" {
void harness_build_f32(int kind, int flags, const float* p, float energy, float* M, float* coef, int want_coef) {
  build<float>(kind, flags, p, energy, M, coef, want_coef);
}
void harness_build_f64(int kind, int flags, const double* p, double energy, double* M, double* coef, int want_coef) {
  build<double>(kind, flags, p, energy, M, coef, want_coef);
}
void harness_kick_f32(const float* coef, float s_in, float d_in, float* s_io, float* d_out) {
  lynx::cavity_kick<float>(coef, s_in, d_in, *s_io, *d_out);
}
void harness_kick_f64(const double* coef, double s_in, double d_in, double* s_io, double* d_out) {
  lynx::cavity_kick<double>(coef, s_in, d_in, *s_io, *d_out);
}
}

// Dual-number evaluation of the same builders (lynx_dual.hpp): value and derivative w.r.t.
// parameter `seed` (seed == n_params: derivative w.r.t. the beam energy).
#include "../../lynx_amd/csrc/lynx_dual.hpp"
extern "C" void harness_build_dual_f64(int kind, int flags, const double* p, int n_params, double energy, int seed,
                                       double* M, double* dM, double* coef, double* dcoef, int want_coef) {
  lynx::Dual<double> dp[49], dm[49], dc[8];
  for (int q = 0; q < 49; ++q) dp[q] = lynx::Dual<double>(q < n_params ? p[q] : 0.0, q == seed ? 1.0 : 0.0);
  for (int q = 0; q < 8; ++q) dc[q] = lynx::Dual<double>(0.0);
  lynx::Dual<double> de(energy, seed == n_params ? 1.0 : 0.0);
  lynx::build_element<lynx::Dual<double>>(kind, flags, dp, de, dm, want_coef ? dc : nullptr);
  for (int q = 0; q < 49; ++q) { M[q] = dm[q].v; dM[q] = dm[q].d; }
  for (int q = 0; q < 8; ++q) { coef[q] = dc[q].v; dcoef[q] = dc[q].d; }
}
extern "C" void harness_build_dual_f32(int kind, int flags, const float* p, int n_params, float energy, int seed,
                                       float* M, float* dM, float* coef, float* dcoef, int want_coef) {
  lynx::Dual<float> dp[49], dm[49], dc[8];
  for (int q = 0; q < 49; ++q) dp[q] = lynx::Dual<float>(q < n_params ? p[q] : 0.0f, q == seed ? 1.0f : 0.0f);
  for (int q = 0; q < 8; ++q) dc[q] = lynx::Dual<float>(0.0f);
  lynx::Dual<float> de(energy, seed == n_params ? 1.0f : 0.0f);
  lynx::build_element<lynx::Dual<float>>(kind, flags, dp, de, dm, want_coef ? dc : nullptr);
  for (int q = 0; q < 49; ++q) { M[q] = dm[q].v; dM[q] = dm[q].d; }
  for (int q = 0; q < 8; ++q) { coef[q] = dc[q].v; dcoef[q] = dc[q].d; }
}

// The 16-entry builder of structured maps (lynx_maps.hpp: build_entries_u): returns 1 and the entries, or 0.
extern "C" int harness_entries_u_f32(int kind, int flags, const float* p, float energy, float* m16, float* coef) {
  float m[16];
  const bool ok = lynx::build_entries_u<float>(kind, flags, p, energy, m, coef);
  for (int k = 0; k < 16; ++k) m16[k] = m[k];
  return ok ? 1 : 0;
}
extern "C" int harness_entries_u_f64(int kind, int flags, const double* p, double energy, double* m16, double* coef) {
  double m[16];
  const bool ok = lynx::build_entries_u<double>(kind, flags, p, energy, m, coef);
  for (int k = 0; k < 16; ++k) m16[k] = m[k];
  return ok ? 1 : 0;
}
