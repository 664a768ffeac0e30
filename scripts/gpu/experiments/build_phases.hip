// Round 4 experiment: where do the 20 us of k_build<double> on ONE sample of a 128-element FODO go?
// A copy of build_compose_sample (lynx_device.hpp) with wall-clock probes at the phase boundaries, next to the real kernel.
// Build: hipcc -O3 --offload-arch=gfx950 -std=c++17 -ffp-contract=off -fno-slp-vectorize build_phases.hip -o build_phases
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

#include "../../../lynx_amd/csrc/lynx_device.hpp"

using namespace lynx;

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <typename T>
__global__ __launch_bounds__(1024) void k_probe(LatticeDev lat, const T* energy_in, T* steps_out, int chunk, unsigned long long* probes) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  T* s_steps = reinterpret_cast<T*>(smem_raw + build_scratch_bytes(chunk, sizeof(T)));
  T* s_energy = s_steps + (size_t)lat.n_steps * LYNX_STEP_STRIDE;
  const int tid = threadIdx.x;
  int np = 0;
  auto probe = [&]() { if (tid == 0) probes[np] = wall_clock64(); ++np; };
  probe();
  for (int i = tid; i < lat.n_steps * LYNX_STEP_STRIDE; i += blockDim.x) s_steps[i] = T(0);
  __syncthreads();
  probe();
  const int E = lat.n_elems;
  const T* pool = static_cast<const T*>(lat.pool);
  double* bufA = reinterpret_cast<double*>(smem_raw);
  T* built = sizeof(T) == 8 ? reinterpret_cast<T*>(bufA) : reinterpret_cast<T*>(bufA + (chunk + (chunk / 2 + 1) + (chunk / 4 + 1) + 2) * 49);
  const int waves = blockDim.x >> 6;
  if (tid == 0) { s_energy[0] = energy_in[0]; s_energy[1] = energy_in[0]; }
  __syncthreads();
  probe();
  {
    const int t = (tid & 63) * waves + (tid >> 6);
    if (t < E) {
      lynx_elem el = lat.elems[t];
      const T* p = pool + el.param_offset;
      probe(); --np;
      build_element<T>(el.kind, el.flags, p, s_energy[0], built + t * 49, nullptr);
    }
  }
  ++np;
  probe();   // thread 0's own element done
  __syncthreads();
  probe();   // every element done
  // one tree level by hand: products of neighbours
  const int my_prod = tid / 7, my_row = tid - my_prod * 7, per_round = blockDim.x / 7;
  double* bufB = bufA + chunk * 49;
  if (sizeof(T) == 8) {
    for (int pr = my_prod; pr < E / 2; pr += per_round) mat_product_row(bufA + (2 * pr + 1) * 49, bufA + (2 * pr) * 49, bufB + pr * 49, my_row);
  }
  __syncthreads();
  probe();
  if (tid < 49) steps_out[tid] = (T)bufB[tid];
}

int main() {
  const int E = 128;
  std::vector<lynx_elem> elems(E);
  std::vector<double> pool;
  for (int e = 0; e < E; ++e) {
    const bool quad = (e % 2) == 0;
    elems[e].kind = quad ? LYNX_KIND_QUADRUPOLE : LYNX_KIND_DRIFT;
    elems[e].flags = 0;
    elems[e].param_offset = (int)pool.size();
    elems[e].batch_stride = 0;
    if (quad) { pool.push_back(0.2); pool.push_back((e % 4) ? -4.2 : 4.2); pool.push_back(0); pool.push_back(0); pool.push_back(0); }
    else pool.push_back(0.5);
  }
  lynx_step step{LYNX_STEP_RUN, 0, E, 0};
  std::vector<int32_t> elem_step(E, 0);
  LatticeDev lat{};
  void *d_e, *d_s, *d_es, *d_p, *d_en, *d_out;
  unsigned long long* d_probes;
  CK(hipMalloc(&d_e, E * sizeof(lynx_elem))); CK(hipMalloc(&d_s, sizeof(lynx_step))); CK(hipMalloc(&d_es, E * 4));
  CK(hipMalloc(&d_p, pool.size() * 8)); CK(hipMalloc(&d_en, 8)); CK(hipMalloc(&d_out, 64 * 8)); CK(hipMalloc((void**)&d_probes, 64 * 8));
  CK(hipMemcpy(d_e, elems.data(), E * sizeof(lynx_elem), hipMemcpyHostToDevice));
  CK(hipMemcpy(d_s, &step, sizeof(step), hipMemcpyHostToDevice));
  CK(hipMemcpy(d_es, elem_step.data(), E * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(d_p, pool.data(), pool.size() * 8, hipMemcpyHostToDevice));
  const double energy = 1e8;
  CK(hipMemcpy(d_en, &energy, 8, hipMemcpyHostToDevice));
  lat.elems = (const lynx_elem*)d_e; lat.steps = (const lynx_step*)d_s; lat.elem_step = (const int32_t*)d_es; lat.pool = d_p;
  lat.batch = 1; lat.n_elems = E; lat.n_steps = 1;
  hipStream_t st; CK(hipStreamCreate(&st));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int threads : {1024, 512, 256, 128, 64}) {
    for (int chunk : {128, 64, 32}) {
      if (chunk > threads) continue;
      const size_t lds = build_scratch_bytes(chunk, 8) + (64 + 2) * 8;
      CK(hipFuncSetAttribute((const void*)k_build<double>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
      for (int i = 0; i < 5; ++i) hipLaunchKernelGGL(k_build<double>, dim3(1), dim3(threads), lds, st, lat, (const double*)d_en, (double*)d_out, (double*)nullptr, chunk, 0);
      CK(hipStreamSynchronize(st));
      CK(hipEventRecord(e0, st));
      const int reps = 200;
      for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(k_build<double>, dim3(1), dim3(threads), lds, st, lat, (const double*)d_en, (double*)d_out, (double*)nullptr, chunk, 0);
      CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      printf("k_build<double> threads %d chunk %d: %.2f us per launch (back to back)\n", threads, chunk, ms * 1e3 / reps);
    }
  }
  for (int threads : {1024, 256, 128}) {
    const int chunk = 128;
    const size_t lds = build_scratch_bytes(chunk, 8) + (64 + 2) * 8;
    printf("probe with %d threads\n", threads);
    CK(hipFuncSetAttribute((const void*)k_probe<double>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    for (int rep = 0; rep < 3; ++rep) {
      hipLaunchKernelGGL(k_probe<double>, dim3(1), dim3(threads), lds, st, lat, (const double*)d_en, (double*)d_out, chunk, d_probes);
      CK(hipStreamSynchronize(st));
      unsigned long long p[8];
      CK(hipMemcpy(p, d_probes, sizeof(p), hipMemcpyDeviceToHost));
      printf("probe (100 MHz ticks -> us): zero %.2f  energy %.2f  before build %.2f  thread 0's element %.2f  all elements %.2f  one tree level %.2f\n",
             (p[1] - p[0]) / 100.0, (p[2] - p[1]) / 100.0, (p[3] - p[2]) / 100.0, (p[4] - p[3]) / 100.0, (p[5] - p[4]) / 100.0, (p[6] - p[5]) / 100.0);
    }
  }
  // an empty kernel, for the floor
  return 0;
}
