// Round 4 experiment: what bounds k_build on ONE sample?  Back-to-back durations of
//   (1) an empty kernel, (2) a kernel that only follows the build's chain of dependent loads (kernarg -> element -> parameters),
//   (3) k_build<float> on a 13-element ARES-like lattice and k_build<double> on the 128-element FODO,
//   (4) the same builds right behind a 1024-workgroup launch of the same kernel (instruction caches of every CU warm).
// Build: hipcc -O3 --offload-arch=gfx950 -std=c++17 -ffp-contract=off -fno-slp-vectorize build_floor.hip -o build_floor
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

#include "../../../lynx_amd/csrc/lynx_device.hpp"

using namespace lynx;

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__global__ void k_empty() {}
template <typename T>
__global__ __launch_bounds__(256) void k_chain(LatticeDev lat, const T* energy_in, T* out) {
  const int t = threadIdx.x;
  if (t < lat.n_elems) {
    lynx_elem el = lat.elems[t];
    const T* p = static_cast<const T*>(lat.pool) + el.param_offset;
    out[t] = p[0] + energy_in[0];
  }
}

template <typename T>
struct Lattice {
  LatticeDev lat{};
  void *d_e, *d_s, *d_es, *d_p, *d_en, *d_out;
  int E;
  int make(const std::vector<lynx_elem>& elems, const std::vector<T>& pool) {
    E = (int)elems.size();
    lynx_step step{LYNX_STEP_RUN, 0, E, 0};
    std::vector<int32_t> elem_step(E, 0);
    CK(hipMalloc(&d_e, E * sizeof(lynx_elem))); CK(hipMalloc(&d_s, sizeof(lynx_step))); CK(hipMalloc(&d_es, E * 4));
    CK(hipMalloc(&d_p, pool.size() * sizeof(T))); CK(hipMalloc(&d_en, 2048 * sizeof(T))); CK(hipMalloc(&d_out, 2048 * 64 * sizeof(T)));
    CK(hipMemcpy(d_e, elems.data(), E * sizeof(lynx_elem), hipMemcpyHostToDevice));
    CK(hipMemcpy(d_s, &step, sizeof(step), hipMemcpyHostToDevice));
    CK(hipMemcpy(d_es, elem_step.data(), E * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_p, pool.data(), pool.size() * sizeof(T), hipMemcpyHostToDevice));
    std::vector<T> energy(2048, T(1e8));
    CK(hipMemcpy(d_en, energy.data(), 2048 * sizeof(T), hipMemcpyHostToDevice));
    lat.elems = (const lynx_elem*)d_e; lat.steps = (const lynx_step*)d_s; lat.elem_step = (const int32_t*)d_es; lat.pool = d_p;
    lat.batch = 1; lat.n_elems = E; lat.n_steps = 1;
    return 0;
  }
};

template <typename T>
int measure(const char* what, Lattice<T>& L, hipStream_t st, hipEvent_t e0, hipEvent_t e1) {
  const int chunk = 128, reps = 200;
  const size_t lds = build_scratch_bytes(chunk, sizeof(T)) + (64 + 2) * sizeof(T);
  CK(hipFuncSetAttribute((const void*)k_build<T>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  float ms;
  auto timed = [&](auto launch) -> float {
    for (int i = 0; i < 5; ++i) launch();
    hipStreamSynchronize(st);
    hipEventRecord(e0, st);
    for (int i = 0; i < reps; ++i) launch();
    hipEventRecord(e1, st); hipEventSynchronize(e1);
    float t; hipEventElapsedTime(&t, e0, e1); return t * 1e3f / reps;
  };
  const float t_chain = timed([&] { hipLaunchKernelGGL(k_chain<T>, dim3(1), dim3(256), 0, st, L.lat, (const T*)L.d_en, (T*)L.d_out); });
  const float t_one = timed([&] { hipLaunchKernelGGL(k_build<T>, dim3(1), dim3(256), lds, st, L.lat, (const T*)L.d_en, (T*)L.d_out, (T*)nullptr, chunk, 0); });
  const float t_wide = timed([&] { hipLaunchKernelGGL(k_build<T>, dim3(1024), dim3(256), lds, st, L.lat, (const T*)L.d_en, (T*)L.d_out, (T*)nullptr, chunk, 0); });
  const float t_both = timed([&] {
    hipLaunchKernelGGL(k_build<T>, dim3(1024), dim3(256), lds, st, L.lat, (const T*)L.d_en, (T*)L.d_out, (T*)nullptr, chunk, 0);
    hipLaunchKernelGGL(k_build<T>, dim3(1), dim3(256), lds, st, L.lat, (const T*)L.d_en, (T*)L.d_out, (T*)nullptr, chunk, 0); });
  ms = 0; (void)ms;
  printf("%-34s chain of loads only %.2f us | one workgroup %.2f us | 1024 workgroups %.2f us | 1024 then one %.2f us -> one behind warm instruction caches %.2f us\n",
         what, t_chain, t_one, t_wide, t_both, t_both - t_wide);
  return 0;
}

int main() {
  hipStream_t st; CK(hipStreamCreate(&st));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  {
    for (int i = 0; i < 5; ++i) hipLaunchKernelGGL(k_empty, dim3(1), dim3(64), 0, st);
    CK(hipStreamSynchronize(st));
    CK(hipEventRecord(e0, st));
    for (int i = 0; i < 500; ++i) hipLaunchKernelGGL(k_empty, dim3(1), dim3(64), 0, st);
    CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    printf("empty kernel, back to back: %.2f us per launch\n", ms * 1e3 / 500);
  }
  {  // BASELINE config 2's lattice: 13 elements, float32 (drift, quadrupole, correctors, markers)
    std::vector<lynx_elem> elems; std::vector<float> pool;
    const int kinds[13] = {LYNX_KIND_DRIFT, LYNX_KIND_QUADRUPOLE, LYNX_KIND_DRIFT, LYNX_KIND_QUADRUPOLE, LYNX_KIND_DRIFT, LYNX_KIND_VCOR,
                           LYNX_KIND_DRIFT, LYNX_KIND_QUADRUPOLE, LYNX_KIND_DRIFT, LYNX_KIND_HCOR, LYNX_KIND_DRIFT, LYNX_KIND_IDENTITY, LYNX_KIND_DRIFT};
    for (int e = 0; e < 13; ++e) {
      lynx_elem el{kinds[e], 0, (int)pool.size(), 0};
      if (kinds[e] == LYNX_KIND_QUADRUPOLE) { pool.push_back(0.122f); pool.push_back(e % 4 ? -4.2f : 4.2f); pool.push_back(0); pool.push_back(0); pool.push_back(0); }
      else if (kinds[e] == LYNX_KIND_IDENTITY) pool.push_back(0);
      else if (kinds[e] == LYNX_KIND_DRIFT) pool.push_back(0.2f);
      else { pool.push_back(0.02f); pool.push_back(1e-3f); }
      while (pool.size() % 8) pool.push_back(0);
      elems.push_back(el);
    }
    Lattice<float> L; if (L.make(elems, pool)) return 1;
    if (measure<float>("13 elements, float32 (config 2)", L, st, e0, e1)) return 1;
  }
  {
    std::vector<lynx_elem> elems; std::vector<double> pool;
    for (int e = 0; e < 128; ++e) {
      const bool quad = (e % 2) == 0;
      lynx_elem el{quad ? LYNX_KIND_QUADRUPOLE : LYNX_KIND_DRIFT, 0, (int)pool.size(), 0};
      if (quad) { pool.push_back(0.2); pool.push_back((e % 4) ? -4.2 : 4.2); pool.push_back(0); pool.push_back(0); pool.push_back(0); }
      else pool.push_back(0.5);
      elems.push_back(el);
    }
    Lattice<double> L; if (L.make(elems, pool)) return 1;
    if (measure<double>("128-element FODO, float64 (config 3)", L, st, e0, e1)) return 1;
  }
  return 0;
}
