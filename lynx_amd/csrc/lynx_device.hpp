// gfx950 kernels of the beam-tracking hot path.  Included only by lynx_hip.hip.
//
// Data layout in HBM
//   particles   [B][N][7]  array-of-structs, exactly the reference's `(…, N, 7)` array
//               (lynx/particles/particle_beam.py:24-45); scalar type T = float | double.
//   lattice     elems[E], steps[S], elem_step[E] (int32) + parameter pool (T), see
//               include/lynx_hip.h.
//   step table  [B][S][64] T: per sample and step the composed 7x7 map (49), the cavity
//               coefficients (8), a merged pair's entry inverse (4), the step's descriptor (slot 62)
//               and, in the last step, the outgoing energy (slot 63); lynx_maps.hpp.  Written by the
//               builders (k_build, or k_build_pieces / k_pair_products / k_emit_steps for large
//               batches), read with scalar loads by the streaming kernel (in the optional fused
//               variant it only ever lives in LDS).
//   partials    [B][chunks][36] double: per-workgroup moment records, each around a reference
//               point of its own (deterministic reduction in k_reduce_moments; no float atomics).
//
// Wavefront = 64 everywhere; workgroups are 256 threads (4 waves) for the streaming kernel, 64 to
// 1024 for the builders and the per-sample kernels.
#pragma once

#include <hip/hip_runtime.h>

#include <type_traits>

#include "lynx_maps.hpp"
#include "lynx_unit_record.hpp"

namespace lynx {

// clang native vectors (HIP's float4/double2 structs defeat SROA and land in scratch)
typedef float lynx_f32x4 __attribute__((ext_vector_type(4)));
typedef double lynx_f64x2 __attribute__((ext_vector_type(2)));
template <typename T, bool VEC> struct VecOf { using type = T; static constexpr int width = 1; };
template <> struct VecOf<float, true> { using type = lynx_f32x4; static constexpr int width = 4; };
template <> struct VecOf<double, true> { using type = lynx_f64x2; static constexpr int width = 2; };

struct LatticeDev {
  const lynx_elem* elems;
  const lynx_step* steps;
  const int32_t* elem_step;
  const void* pool;
  int64_t batch;
  int32_t n_elems;
  int32_t n_steps;
};

// The parameter pool of a SMALL lattice inside the kernel arguments: a parameter write (`quad.k1 = ...` between two
// `track` calls -- the optimisation loop) then is a memcpy on the host instead of a copy to HBM the stream has to be
// waited for (BASELINE config 2 with a setting changed before every call: 38.6 -> 23.8 us per call), and the pool in
// HBM follows when a kernel that reads it from there is next launched (lynx_hip.hip: sync_pool).
// Two sizes, because argument bytes cost latency: with the tables in there as well (3.5 KB) a call that is waited for
// took 4 us longer, and nothing was gained on the build's chain of dependent accesses (NOTES.md).
// The pool is the kernel's FIRST parameter and is read where it lies, at offset 0 of the kernel-argument segment:
// naming the parameter makes the compiler copy all of it into every lane's scratch as soon as a pointer into it
// reaches a function that is not inlined.
template <int BYTES>
struct InlinePool {
  uint4 q[BYTES / 16];
};
constexpr int kInlinePoolSmall = 256, kInlinePoolLarge = 1024;
__device__ __forceinline__ LatticeDev inline_pool_view(const LatticeDev& lat) {
  LatticeDev v = lat;
  v.pool = (const void*)__builtin_amdgcn_kernarg_segment_ptr();
  return v;
}

// a run directly in front of an active cavity can be applied together with it -- unless it is an
// observer (its step must stay visible to the streaming kernel)
__device__ __forceinline__ bool steps_pair_up(const lynx_step& run, const lynx_step& cav) {
  return run.kind == LYNX_STEP_RUN && cav.kind == LYNX_STEP_CAVITY && !(run.flags & LYNX_STEP_FLAG_OBSERVE);
}

// what a builder parks in slot LYNX_FLAGS_OFFSET of step s
__device__ __forceinline__ int step_descriptor(const LatticeDev& lat, int s, int merge_pairs) {
  const lynx_step st = lat.steps[s];
  int d = (st.flags & 0xffff) | (st.kind << LYNX_DESC_KIND_SHIFT);
  if (merge_pairs && s + 1 < lat.n_steps && steps_pair_up(st, lat.steps[s + 1])) d |= LYNX_DESC_PAIR;
  return d;
}

// LYNX_ENTRY_OFFSET: inverse of the (s, delta) block of a cavity's rounded map, formed in float64.  Returns whether the
// block is too ill-conditioned to be used that way (LYNX_DESC_ILL): the forward kernels recover s_in, delta_in from
// components 4 and 5 of M z, whose float32 rounding the inverse amplifies by about
// (|c44 c55| + |c45 c54|) / |det| -- 1.0 to 1.04 on ultra-relativistic beams (BASELINE config 5), unbounded at low
// energy off crest (the block is singular at E = 1 MeV, phi = -30 deg, V = 0.5247 MV).  Above kEntryConditionLimit, or with a det that is
// not finite or zero, the pair is applied in the rows form instead, which has no such factor.
constexpr double kEntryConditionLimit = 4.0;
template <typename T>
__device__ __forceinline__ bool cavity_entry_inverse(T c44, T c45, T c54, T c55, T (&ci)[4]) {
  const double a = (double)c44 * (double)c55, b = (double)c45 * (double)c54;
  const double det = a - b;
  ci[0] = (T)((double)c55 / det);
  ci[1] = (T)(-(double)c45 / det);
  ci[2] = (T)(-(double)c54 / det);
  ci[3] = (T)((double)c44 / det);
  return !(fabs(det) * kEntryConditionLimit >= fabs(a) + fabs(b)) || det == 0.0;  // NaN compares false: ill
}

constexpr int kBuildChunk = 64;   // elements built in parallel per compose round (k_build may use up to 128)
constexpr int kTrackThreads = 256;
constexpr int kPartialStride = 36;

template <typename T> __device__ __forceinline__ T shfl_t(T v, int src) { return __shfl(v, src, 64); }

// ---------------------------------------------------------------------------------------
// Build + compose for ONE batch sample, executed by a whole workgroup (blockDim.x threads,
// a multiple of 64).  Result: s_steps[S][64] in LDS, s_energy[S+1] in LDS.
//
//   phase 0  thread 0 walks the steps and accumulates the beam energy across active
//            cavities (cavity.py:130: E_out = E + V cos(phi)).
//   phase 1  up to `chunk` elements are built in parallel, one element per thread, in the
//            lattice's own precision T (the reference builds every element map in the beam's
//            dtype).  Lane l of wave w takes element l * waves + w, so neighbouring elements
//            -- usually of different kinds -- land on different waves and each wave runs as
//            few different builders as possible, side by side on the CU's SIMDs.
//   phase 2  the chunk's maps of each step are multiplied together by a pairwise tree IN
//            FLOAT64 (also for float32 lattices): level by level neighbours (M_{2i+1} . M_{2i})
//            are merged, seven lanes per product (one lane per output row), then the chunk
//            product is left-multiplied onto the step's running map, tm = P_chunk . tm, and the
//            finished step is rounded to T once.
//            The product is the reference's tm = M_n ... M_2 M_1 (segment.py:334-335).  The
//            reference multiplies left to right in T; a float32 tree instead measured 2e-5
//            (sigma) / 5e-5 (sigma_xx') away from that chain on BASELINE config 4 -- inside the
//            1e-4 tolerance, but half of it.  Accumulated in float64 the product is exact to
//            ~1e-15 whatever the association, so what is left against the reference's float32
//            chain is the chain's own rounding (6e-6 / 1e-5 on config 4, tests/test_gpu_parity.py),
//            and the build keeps log2 instead of linear depth: it is a pure latency chain.
//
// LDS scratch: see build_scratch_bytes().
// ---------------------------------------------------------------------------------------
__host__ __device__ inline int build_chunk(int n_elems, int limit = kBuildChunk) {
  return n_elems < limit ? (n_elems > 1 ? n_elems : 1) : limit;
}
// float64 maps: chunk + two ping-pong tree levels (c/2+1, c/4+1) + the running map + one temporary;
// float32 lattices add the staging area the builders write their T-typed maps to.  The chunk must
// not exceed the workgroup size (one element per thread in phase 1).
__host__ __device__ inline size_t build_scratch_bytes(int chunk, size_t elem_size) {
  size_t b = (size_t)(chunk + (chunk / 2 + 1) + (chunk / 4 + 1) + 2) * 49 * sizeof(double);
  if (elem_size == 4) b += (size_t)chunk * 49 * sizeof(float);
  return (b + 15) / 16 * 16;
}

// Row r of D = A . Bm (7x7, row-major, float64), ascending k -- one lane per output row
__device__ __forceinline__ void mat_product_row(const double* A, const double* Bm, double* D, int r) {
  double a[7], acc[7];
#pragma unroll
  for (int k = 0; k < 7; ++k) a[k] = A[r * 7 + k];
#pragma unroll
  for (int j = 0; j < 7; ++j) acc[j] = a[0] * Bm[j];
#pragma unroll
  for (int k = 1; k < 7; ++k)
#pragma unroll
    for (int j = 0; j < 7; ++j) acc[j] = fma(a[k], Bm[k * 7 + j], acc[j]);
#pragma unroll
  for (int j = 0; j < 7; ++j) D[r * 7 + j] = acc[j];
}

template <typename T>
__device__ __forceinline__ T mat_product_entry(const T* A, const T* Bm, int ij) {
  // (A . Bm)[i][j] = sum_k A[i][k] * Bm[k][j], ascending k
  const int i = ij / 7, j = ij - i * 7;
  T acc = A[i * 7] * Bm[j];
#pragma unroll
  for (int k = 1; k < 7; ++k) acc = t_fma(A[i * 7 + k], Bm[k * 7 + j], acc);
  return acc;
}

template <typename T>
__device__ void build_compose_sample(const LatticeDev& lat, int64_t b, T energy_in, T* s_steps,
                                     T* s_energy, unsigned char* s_scratch /* build_scratch_bytes(chunk, sizeof(T)) */,
                                     int chunk = kBuildChunk) {
  const int tid = threadIdx.x;
  const int E = lat.n_elems, S = lat.n_steps;
  const T* pool = static_cast<const T*>(lat.pool);
  double* bufA = reinterpret_cast<double*>(s_scratch);
  double* bufB = bufA + chunk * 49;
  double* bufC = bufB + (chunk / 2 + 1) * 49;
  double* carry = bufC + (chunk / 4 + 1) * 49;
  double* tmp = carry + 49;
  // where the builders write: straight into bufA for float64, into the staging area for float32
  T* built = sizeof(T) == 8 ? reinterpret_cast<T*>(bufA) : reinterpret_cast<T*>(tmp + 49);
  const int per_round = blockDim.x / 7;  // products per round, 7 row-lanes each
  const int per_round49 = blockDim.x / 49;  // ... or 49 entry-lanes each
  const int my_prod = tid / 7, my_row = tid - my_prod * 7;
  const int waves = blockDim.x >> 6;

  if (tid == 0) {
    T e = energy_in;
    for (int s = 0; s < S; ++s) {
      s_energy[s] = e;
      lynx_step st = lat.steps[s];
      if (st.kind == LYNX_STEP_CAVITY && (st.flags & LYNX_FLAG_CAV_GAIN)) {
        lynx_elem el = lat.elems[st.first];
        const T* p = pool + el.param_offset + b * (int64_t)el.batch_stride;
        T phi = p[2] * T(LYNX_PI / 180.0);
        T de = p[1] * t_cos(phi);
        e = e + de;
      }
    }
    s_energy[S] = e;
  }
  __syncthreads();

  int s_cur = 0;  // first step that may intersect the current chunk (uniform)
  for (int e0 = 0; e0 < E; e0 += chunk) {
    const int ne = (E - e0) < chunk ? (E - e0) : chunk;
    // phase 1
    {
      const int t = (tid & 63) * waves + (tid >> 6);
      if (t < ne) {
        const int e = e0 + t;
        lynx_elem el = lat.elems[e];
        const int s = lat.elem_step[e];
        const T* p = pool + el.param_offset + b * (int64_t)el.batch_stride;
        const bool cav_step = lat.steps[s].kind == LYNX_STEP_CAVITY;
        build_element<T>(el.kind, el.flags, p, s_energy[s], built + t * 49,
                         cav_step ? s_steps + s * LYNX_STEP_STRIDE + LYNX_COEF_OFFSET : nullptr);
      }
    }
    __syncthreads();
    if (sizeof(T) == 4) {
      for (int i = tid; i < ne * 49; i += blockDim.x) bufA[i] = (double)built[i];
      __syncthreads();
    }
    // phase 2: every step that has elements in [e0, e0 + ne)
    while (s_cur < S) {
      const lynx_step st = lat.steps[s_cur];
      if (st.first >= e0 + ne) break;
      const int lo = (st.first > e0 ? st.first : e0) - e0;
      const int hi = (st.last < e0 + ne ? st.last : e0 + ne) - e0;
      const bool starts_here = st.first >= e0;
      const bool ends_here = st.last <= e0 + ne;
      const bool raw = st.kind == LYNX_STEP_CAVITY || (st.flags & LYNX_STEP_FLAG_RAW);
      // pairwise tree over bufA[lo .. hi)
      const double* src = bufA + lo * 49;
      int count = hi - lo;
      int level = 0;
      while (count > 1) {
        double* dst = (level & 1) ? bufC : bufB;
        const int half = count >> 1;
        // Few products in the level: one lane per ENTRY (seven multiply-adds behind fourteen LDS reads, ~0.15 us) -- the
        // row form (one lane per output row: 56 reads and 49 multiply-adds in a lane, ~0.8 us whatever the count) pays
        // only when there are enough products to fill the workgroup with rows.  Same multiply-adds in the same order.
        if (half <= 2 * per_round49) {
          if (tid < per_round49 * 49)
            for (int pr = tid / 49; pr < half; pr += per_round49)
              dst[pr * 49 + tid % 49] = mat_product_entry<double>(src + (2 * pr + 1) * 49, src + (2 * pr) * 49, tid % 49);
        } else if (my_prod < per_round) {
          for (int pr = my_prod; pr < half; pr += per_round)
            mat_product_row(src + (2 * pr + 1) * 49, src + (2 * pr) * 49, dst + pr * 49, my_row);
        }
        if ((count & 1) && tid >= blockDim.x - 49) {
          const int q = tid - (blockDim.x - 49);
          dst[half * 49 + q] = src[(count - 1) * 49 + q];
        }
        __syncthreads();
        src = dst;
        count = half + (count & 1);
        ++level;
      }
      // running map of the step: tm = P . tm, starting from eye(7) (segment.py:331) unless raw
      if (starts_here && raw) {
        if (tid < 49) carry[tid] = src[tid];
      } else {
        if (starts_here) {
          if (tid < 49) carry[tid] = (tid % 8 == 0) ? 1.0 : 0.0;
          __syncthreads();
        }
        if (tid < 49) tmp[tid] = mat_product_entry<double>(src, carry, tid);
        __syncthreads();
        if (tid < 49) carry[tid] = tmp[tid];
      }
      __syncthreads();
      if (ends_here) {
        if (tid < 49) s_steps[s_cur * LYNX_STEP_STRIDE + tid] = (T)carry[tid];
        ++s_cur;
      } else {
        break;  // the step continues in the next chunk
      }
    }
    __syncthreads();
  }
}

// sin(phi) of every cavity step into its row of a step table in LDS (LYNX_SINPHI_OFFSET: device_cavity_kick's float32
// form); behind build_compose_sample, in front of whoever reads the rows.  The caller puts the barrier.
template <typename T>
__device__ __forceinline__ void table_sinphi(const LatticeDev& lat, T* s_steps) {
  for (int s = threadIdx.x; s < lat.n_steps; s += blockDim.x)
    if (lat.steps[s].kind == LYNX_STEP_CAVITY)
      s_steps[s * LYNX_STEP_STRIDE + LYNX_SINPHI_OFFSET] = t_sin<T>(s_steps[s * LYNX_STEP_STRIDE + LYNX_COEF_OFFSET + LYNX_C_PHI]);
}

// ---------------------------------------------------------------------------------------
// k_cavity_flags: the reference decides three things about a cavity for the WHOLE batch at once, in
// Python (`if any(...)`): whether beta0/beta1 enter the map (cavity.py:290: any(V != 0 & E != 0)), whether
// the beam gains energy there (cavity.py:128: any(E + dE > 0)) and whether the second-order path-length
// terms apply (cavity.py:164: any(dE > 0)).  They depend on the beam energy, which for the second cavity
// of a lattice is itself a result of the first.  One workgroup walks the cavities in lattice order,
// OR-reduces the predicates over the batch, writes the bits into the element / step flags the builders
// read, and carries every sample's energy forward -- so the host never needs the energy (round 1 read it
// back and redid this bookkeeping in NumPy).
// ---------------------------------------------------------------------------------------
constexpr int kCavMask = LYNX_FLAG_CAV_BETA | LYNX_FLAG_CAV_GAIN | LYNX_FLAG_CAV_T5XX;

constexpr int kFlagsPerThread = 8;  // samples whose energy a thread of k_cavity_flags keeps in registers

// `status` (host-mapped, two words): cavity.py:260 asserts that every sample reaches every cavity with Ei > 0.  A
// sample that does not sets word 0 and leaves the element's index in word 1; the host looks at it when it waits for
// the GPU next (lynx_sync, lynx_buf_d2h) -- no read-back per call, and a beam whose energy lives in HBM only (the
// output of an earlier cavity program) is checked like any other.
//
// Two kernels.  The walk is serial in the cavities only through `any(E + dE > 0)`: when it holds -- as good as always
// -- every sample simply moves on with E + dE.  k_cavity_flags_spec ASSUMES that: one lane per sample, any number of
// workgroups, every lane walks its own sample through the cavities and the predicates are OR-ed into one word per
// cavity (wave ballot + one atomic per wave).  k_cavity_flags (ONE workgroup of 256) then checks the assumption on
// those words; if every step cavity's batch gains energy it only has to copy the bits into the element / step flags,
// otherwise it redoes the walk the serial way.  (One workgroup of 1024 walking 8 cavities through four block-wide
// reductions each took 45 us alone and 500 us underneath BASELINE config 5's streaming kernel: sixteen waves on one
// CU do not find room next to it.  The pair takes a few microseconds and its waves fit anywhere.)
constexpr int kBadEnergy = 1 << 30;

template <typename T>
__device__ __forceinline__ T cavity_look(const T* pool, const lynx_elem& el, bool is_step, T energy, int64_t b, int& mine) {
  if (!(energy > T(0))) mine |= kBadEnergy;  // cavity.py:260 (NaN fails the assertion too)
  const T* p = pool + el.param_offset + b * (int64_t)el.batch_stride;
  const T voltage = p[1];
  const T d_energy = voltage * t_cos(p[2] * T(LYNX_PI / 180.0));
  if (voltage != T(0) && energy != T(0)) mine |= LYNX_FLAG_CAV_BETA;
  if (is_step) {
    if (energy + d_energy > T(0)) mine |= LYNX_FLAG_CAV_GAIN;
    if (d_energy > T(0)) mine |= LYNX_FLAG_CAV_T5XX;
  }
  return d_energy;
}

// `cavs`: the lattice's cavities in lattice order, (element, its step if it is a step of its own else -1) -- made by
// the host once per lattice, so that neither kernel has to walk the step and element tables to find them
template <typename T>
__global__ __launch_bounds__(256) void k_cavity_flags_spec(LatticeDev lat, const int2* __restrict__ cavs, int n_cavs,
                                                           const T* __restrict__ energy_in, int32_t* __restrict__ words,
                                                           T* __restrict__ e_steps /* or null */, int64_t Bp) {
  const T* pool = static_cast<const T*>(lat.pool);
  const int64_t b_raw = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const bool live = b_raw < lat.batch;
  const int64_t b = live ? b_raw : lat.batch - 1;
  T energy = energy_in[b];
  // the sample's energy behind 0, 1, 2, ... step cavities, for the lanes build (StepEnergies): what this walk computes anyway
  int k = 0;
  if (e_steps && live) e_steps[b] = energy;
  for (int c = 0; c < n_cavs; ++c) {
    const int2 cv = cavs[c];
    const lynx_elem el = lat.elems[cv.x];
    const bool is_step = cv.y >= 0;
    int mine = 0;
    const T d_energy = cavity_look<T>(pool, el, is_step, energy, b, mine);
    if (!live) mine = 0;
    int bits = 0;
    if (__builtin_amdgcn_ballot_w64((mine & LYNX_FLAG_CAV_BETA) != 0)) bits |= LYNX_FLAG_CAV_BETA;
    if (__builtin_amdgcn_ballot_w64((mine & LYNX_FLAG_CAV_GAIN) != 0)) bits |= LYNX_FLAG_CAV_GAIN;
    if (__builtin_amdgcn_ballot_w64((mine & LYNX_FLAG_CAV_T5XX) != 0)) bits |= LYNX_FLAG_CAV_T5XX;
    if (__builtin_amdgcn_ballot_w64((mine & kBadEnergy) != 0)) bits |= kBadEnergy;
    if ((threadIdx.x & 63) == 0 && bits) atomicOr(&words[c], bits);
    if (is_step) {
      energy = energy + d_energy;  // the assumption: this cavity's batch gains energy (cavity.py:128-130)
      ++k;
      if (e_steps && live) e_steps[(int64_t)k * Bp + b] = energy;
    }
  }
}

template <typename T>
__global__ __launch_bounds__(256) void k_cavity_flags(LatticeDev lat, lynx_elem* elems, lynx_step* steps,
                                                      const T* __restrict__ energy_in, T* __restrict__ e_run,
                                                      int32_t* __restrict__ status, const int2* __restrict__ cavs, int n_cavs,
                                                      int32_t* __restrict__ words, int32_t* __restrict__ spec_valid /* or null */) {
  const T* pool = static_cast<const T*>(lat.pool);
  const int64_t B = lat.batch;
  const int nt = blockDim.x;
  if (threadIdx.x == 0 && spec_valid) *spec_valid = 0;  // until k_cavity_flags_spec's assumption has been found to hold
  if (words) {
    // what k_cavity_flags_spec found: valid if every step cavity's batch gains energy.  One thread per cavity.
    int ok = 1, first_bad = 0x7fffffff;
    for (int c = threadIdx.x; c < n_cavs; c += nt) {
      const int w = words[c];
      if (cavs[c].y >= 0 && !(w & LYNX_FLAG_CAV_GAIN)) ok = 0;
      if ((w & kBadEnergy) && c < first_bad) first_bad = c;
    }
    const int valid = __syncthreads_and(ok);
    if (valid) {
      for (int c = threadIdx.x; c < n_cavs; c += nt) {
        const int2 cv = cavs[c];
        const int f = words[c] & kCavMask;
        elems[cv.x].flags = (elems[cv.x].flags & ~kCavMask) | f;
        if (cv.y >= 0) steps[cv.y].flags = (steps[cv.y].flags & ~kCavMask) | f;
      }
      // the first cavity (in lattice order) that a sample reached with energy <= 0, if any
      __shared__ int s_bad;
      if (threadIdx.x == 0) s_bad = 0x7fffffff;
      __syncthreads();
      if (first_bad != 0x7fffffff) atomicMin(&s_bad, first_bad);
      __syncthreads();
      if (threadIdx.x == 0 && s_bad != 0x7fffffff && status) {
        if (__hip_atomic_exchange(status, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) == 0)
          __hip_atomic_store(status + 1, cavs[s_bad].x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      }
    }
    __syncthreads();  // everybody has read its words
    for (int c = threadIdx.x; c < n_cavs; c += nt) words[c] = 0;  // ready for the next call
    if (valid) {
      if (threadIdx.x == 0 && spec_valid) *spec_valid = 1;  // ... and with it the energies that kernel left in e_steps
      return;
    }
  }
  // every sample's energy on its way through the cavities: in registers for batches of up to 8 per thread, in
  // `e_run` beyond (hundreds of thousands of ParameterBeam settings)
  const bool in_regs = B <= (int64_t)kFlagsPerThread * nt;
  T e_reg[kFlagsPerThread];
#pragma unroll
  for (int q = 0; q < kFlagsPerThread; ++q) {
    const int64_t b = threadIdx.x + (int64_t)q * nt;
    e_reg[q] = b < B ? energy_in[b] : T(1);
  }
  if (!in_regs)
    for (int64_t b = threadIdx.x; b < B; b += nt) e_run[b] = energy_in[b];
  __syncthreads();
  for (int s = 0; s < lat.n_steps; ++s) {
    const lynx_step st = steps[s];
    for (int e = st.first; e < st.last; ++e) {
      const lynx_elem el = elems[e];
      if (el.kind != LYNX_KIND_CAVITY) continue;  // uniform
      const bool is_step = st.kind == LYNX_STEP_CAVITY;
      int mine = 0;
      T de_reg[kFlagsPerThread];
      if (in_regs) {
#pragma unroll
        for (int q = 0; q < kFlagsPerThread; ++q) {
          const int64_t b = threadIdx.x + (int64_t)q * nt;
          de_reg[q] = b < B ? cavity_look<T>(pool, el, is_step, e_reg[q], b, mine) : T(0);
        }
      } else {
        for (int64_t b = threadIdx.x; b < B; b += nt) (void)cavity_look<T>(pool, el, is_step, e_run[b], b, mine);
      }
      int f = 0;
      if (__syncthreads_or(mine & LYNX_FLAG_CAV_BETA)) f |= LYNX_FLAG_CAV_BETA;
      if (__syncthreads_or(mine & LYNX_FLAG_CAV_GAIN)) f |= LYNX_FLAG_CAV_GAIN;
      if (__syncthreads_or(mine & LYNX_FLAG_CAV_T5XX)) f |= LYNX_FLAG_CAV_T5XX;
      if (__syncthreads_or(mine & kBadEnergy) && threadIdx.x == 0 && status) {
        if (__hip_atomic_exchange(status, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) == 0)
          __hip_atomic_store(status + 1, e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);  // the first one found
      }
      if (threadIdx.x == 0) {
        elems[e].flags = (el.flags & ~kCavMask) | f;
        if (is_step) steps[s].flags = (st.flags & ~kCavMask) | f;
      }
      if (is_step && (f & LYNX_FLAG_CAV_GAIN)) {  // cavity.py:130: the whole batch moves on with E + dE
        if (in_regs) {
#pragma unroll
          for (int q = 0; q < kFlagsPerThread; ++q) e_reg[q] = e_reg[q] + de_reg[q];
        } else {
          for (int64_t b = threadIdx.x; b < B; b += nt) {
            const T* p = pool + el.param_offset + b * (int64_t)el.batch_stride;
            e_run[b] = e_run[b] + p[1] * t_cos(p[2] * T(LYNX_PI / 180.0));
          }
        }
      }
      __syncthreads();
    }
  }
}

// ---------------------------------------------------------------------------------------
// Build + compose with LANES = SAMPLES (large batches).
//
// The per-sample workgroup above keeps one element per lane busy in phase 1 and seven lanes per
// product in phase 2: for a batch of thousands of samples that is ~10x more wave-instructions than
// the arithmetic needs, and a build that runs underneath the previous call's streaming kernel
// takes exactly that share of the GPU away from it.  Here a wave owns 64 consecutive samples and
// every lane does ITS sample's arithmetic in registers -- no divergence (all lanes build the same
// element), no LDS, no barrier:
//
//   k_build_pieces   task (64 samples, piece): a piece is a stretch of <= L consecutive elements of one
//                    step.  The lane builds each element's map in T (the reference's precision), widens
//                    it to float64 and left-multiplies it onto the piece's running product, column by
//                    column in place.  Result: slot `piece` of the product buffer, [slot][49][Bp]
//                    float64 with the sample index fastest (coalesced), plus the cavity coefficients.
//   k_pair_products  one level of the pairwise tree over the pieces of a step: dst = X[later] . X[earlier].
//                    The levels are planned on the host from the lattice structure (who multiplies whom
//                    into which slot) and launched one after the other on the build stream.
//   k_emit_steps     task (64 samples, step): the step's product times eye(7) unless raw (segment.py:331,
//                    explicit so that NaN / Inf spread as in the reference), the merged [run, cavity] form,
//                    rounding to T and the row-major step table the streaming kernel reads.
//
// Same mathematics and the same float64 accumulation as build_compose_sample; only the association of
// the products can differ (pieces instead of chunks), i.e. differences at the 1e-16 level before the
// final rounding to T.
// ---------------------------------------------------------------------------------------
struct BuildPiece {
  int32_t first, last;  // elements [first, last)
  int32_t step;
  int32_t pad;
};
struct PairTask {
  int32_t a, b, dst;  // dst = X[b] . X[a]  (b is the later stretch)
  int32_t pad;
};

// beam energy in front of step `s_target` for sample b: incoming energy plus the gain of every active
// cavity before it (cavity.py:130)
template <typename T>
__device__ __forceinline__ T energy_before_step(const LatticeDev& lat, int64_t b, T energy_in, int s_target) {
  const T* pool = static_cast<const T*>(lat.pool);
  T e = energy_in;
  for (int s = 0; s < s_target; ++s) {
    const lynx_step st = lat.steps[s];
    if (st.kind == LYNX_STEP_CAVITY && (st.flags & LYNX_FLAG_CAV_GAIN)) {
      const lynx_elem el = lat.elems[st.first];
      const T* p = pool + el.param_offset + b * (int64_t)el.batch_stride;
      e = e + p[1] * t_cos(p[2] * T(LYNX_PI / 180.0));
    }
  }
  return e;
}

// The same from what k_cavity_flags_spec left behind: e[k][b] = the energy behind k step cavities, `before[s]` = the number
// of step cavities in front of step s (made by the host once per lattice) -- ONE load instead of a walk through every
// earlier cavity's parameters (two dependent loads and a cosine each: up to half of k_build_pieces' time on BASELINE
// config 5, whose last pieces have seven cavities in front of them).  Valid if that kernel's assumption held (`*valid`,
// set by k_cavity_flags); lattices without cavities pass e = null.
template <typename T>
struct StepEnergies {
  const T* e;
  const int32_t* before;
  const int32_t* valid;
  int64_t Bp;
};
template <typename T>
__device__ __forceinline__ T energy_at_step(const LatticeDev& lat, const StepEnergies<T>& se, int64_t b, T energy_in, int s) {
  if (se.e && *se.valid) return se.e[(int64_t)se.before[s] * se.Bp + b];
  return energy_before_step<T>(lat, b, energy_in, s);
}

// LDS of k_build_pieces: the running product [49][64] float64 (sample index fastest: conflict-free) and
// every lane's element map [64][49] T (49 is odd: conflict-free); a lane only ever touches its own cells,
// so the wave needs no barrier.  Registers hold one element map and one column of the product -- the
// kernel has to fit next to the streaming kernel's waves to run underneath it.
template <typename T> constexpr size_t build_pieces_lds() { return 49 * 64 * sizeof(double) + 64 * 49 * sizeof(T); }

// (at most 128 registers per lane: the build runs underneath the previous call's streaming kernel, whose waves hold
// 96 of a SIMD's 512 registers each at five per SIMD -- one of them retiring must make room for a build wave.  At the
// 160 registers the compiler would take, a build wave waited for TWO to retire on the same SIMD and the kernel that
// takes 31 us alone took 710 us next to BASELINE config 5's streaming kernel: the whole step waited for the build.)
template <typename T>
__global__ __launch_bounds__(64, 4) void k_build_pieces(LatticeDev lat, const BuildPiece* __restrict__ pieces,
                                                     const T* __restrict__ energy_in, int64_t Bp,
                                                     double* __restrict__ products, T* __restrict__ coefs,
                                                     StepEnergies<T> se) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  double* P = reinterpret_cast<double*>(smem_raw) + threadIdx.x;                       // P[q * 64]
  T* M = reinterpret_cast<T*>(smem_raw + 49 * 64 * sizeof(double)) + threadIdx.x * 49;  // M[q]
  const int64_t b_raw = (int64_t)blockIdx.x * 64 + threadIdx.x;
  const bool live = b_raw < lat.batch;
  const int64_t b = live ? b_raw : lat.batch - 1;  // idle lanes shadow the last sample, write nothing
  const BuildPiece pc = pieces[blockIdx.y];
  const T* pool = static_cast<const T*>(lat.pool);
  const lynx_step st = lat.steps[pc.step];
  const T energy = energy_at_step<T>(lat, se, b, energy_in[b], pc.step);
  T coef[8];
#pragma unroll
  for (int q = 0; q < 8; ++q) coef[q] = T(0);
  for (int e = pc.first; e < pc.last; ++e) {
    const lynx_elem el = lat.elems[e];
    const T* p = pool + el.param_offset + b * (int64_t)el.batch_stride;
    build_element<T>(el.kind, el.flags, p, energy, M, st.kind == LYNX_STEP_CAVITY ? coef : nullptr);
    LYNX_FORGET();
    if (e == pc.first) {
#pragma unroll 7
      for (int q = 0; q < 49; ++q) P[q * 64] = (double)M[q];
    } else {
      T m[49];
#pragma unroll
      for (int q = 0; q < 49; ++q) m[q] = M[q];
      // P <- M . P, one column of P at a time (a column of the product needs only that column of P)
#pragma unroll 1
      for (int j = 0; j < 7; ++j) {
        double col[7], out[7];
#pragma unroll
        for (int k = 0; k < 7; ++k) col[k] = P[(k * 7 + j) * 64];
#pragma unroll
        for (int i = 0; i < 7; ++i) {
          double acc = (double)m[i * 7] * col[0];
#pragma unroll
          for (int k = 1; k < 7; ++k) acc = fma((double)m[i * 7 + k], col[k], acc);
          out[i] = acc;
        }
#pragma unroll
        for (int i = 0; i < 7; ++i) P[(i * 7 + j) * 64] = out[i];
      }
    }
    LYNX_FORGET();
  }
  if (live) {
    double* dst = products + (int64_t)blockIdx.y * 49 * Bp + b;
#pragma unroll 7
    for (int q = 0; q < 49; ++q) dst[q * Bp] = P[q * 64];
    if (st.kind == LYNX_STEP_CAVITY) {
      T* cd = coefs + (int64_t)pc.step * 8 * Bp + b;
#pragma unroll
      for (int q = 0; q < 8; ++q) cd[q * Bp] = coef[q];
    }
  }
}

// dst = X[b] . X[a] for one sample (the lane's)
__device__ __forceinline__ void pair_product(const PairTask t, int64_t b, int64_t Bp, double* products);

__global__ __launch_bounds__(64) void k_pair_products(const PairTask* __restrict__ tasks, int64_t B, int64_t Bp,
                                                      double* products) {
  const int64_t b = (int64_t)blockIdx.x * 64 + threadIdx.x;
  if (b >= B) return;
  pair_product(tasks[blockIdx.y], b, Bp, products);
}

// Every level of the tree in ONE launch: a workgroup of 4 waves per 64 samples, wave w takes tasks w, w + 4, ... of a
// level, a barrier between levels (the products of a level are read by this workgroup only).  For lattices whose
// levels are narrow (<= kPairLevelsMaxTasks tasks: BASELINE config 4 has 8, 4, 2, 1) the chain of launches was what
// a build took: ~20 us per level underneath a streaming kernel, against 2-3 us for the product itself.
constexpr int kPairLevelsMax = 8, kPairLevelsMaxTasks = 16;
struct PairLevels {
  int32_t n;
  int32_t first[kPairLevelsMax], count[kPairLevelsMax];
};
__global__ __launch_bounds__(256) void k_pair_levels(PairLevels lv, const PairTask* __restrict__ tasks, int64_t B, int64_t Bp,
                                                     double* products) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t b = (int64_t)blockIdx.x * 64 + lane;
  for (int l = 0; l < lv.n; ++l) {
    if (b < B)
      for (int t = wave; t < lv.count[l]; t += 4) pair_product(tasks[lv.first[l] + t], b, Bp, products);
    __syncthreads();  // workgroup scope: this level's products are in memory before the next level reads them
  }
}

__device__ __forceinline__ void pair_product(const PairTask t, int64_t b, int64_t Bp, double* products) {
  const double* xa = products + (int64_t)t.a * 49 * Bp + b;
  const double* xb = products + (int64_t)t.b * 49 * Bp + b;
  double* xd = products + (int64_t)t.dst * 49 * Bp + b;
  double A[49];  // the earlier stretch: right factor, resident
#pragma unroll
  for (int q = 0; q < 49; ++q) A[q] = xa[q * Bp];
#pragma unroll 1
  for (int i = 0; i < 7; ++i) {
    double r[7], out[7];
#pragma unroll
    for (int k = 0; k < 7; ++k) r[k] = xb[(i * 7 + k) * Bp];
#pragma unroll
    for (int j = 0; j < 7; ++j) {
      double acc = r[0] * A[j];
#pragma unroll
      for (int k = 1; k < 7; ++k) acc = fma(r[k], A[k * 7 + j], acc);
      out[j] = acc;
    }
#pragma unroll
    for (int j = 0; j < 7; ++j) xd[(i * 7 + j) * Bp] = out[j];
  }
}

// LDS of k_emit_steps: every lane's table row [64][68] T (pitch 68: 16-byte aligned rows, conflict-free
// 16-byte reads).  Maps are streamed row by row from the product buffer; only the run's map of a merged
// [run, cavity] pair is held in registers, already rounded to T.
template <typename T> constexpr size_t emit_steps_lds() { return 64 * 68 * sizeof(T); }

// row i of a step's final map: product . eye(7) unless raw (segment.py:331-335), multiplied out so that a
// NaN / Inf entry spreads along its row exactly as in the reference; finite entries come back unchanged
__device__ __forceinline__ void emit_row_of(const double (&x)[7], bool raw, double (&out)[7]) {
  if (raw) {
#pragma unroll
    for (int j = 0; j < 7; ++j) out[j] = x[j];
  } else {
#pragma unroll
    for (int j = 0; j < 7; ++j) {
      double acc = x[0] * (j == 0 ? 1.0 : 0.0);
#pragma unroll
      for (int k = 1; k < 7; ++k) acc = fma(x[k], (j == k ? 1.0 : 0.0), acc);
      out[j] = acc;
    }
  }
}
// rows [I0, I0 + N) of a product, all of their loads in flight at once: the product buffer is in HBM, a row at a time
// is seven dependent round trips per map (the kernel is one wave per SIMD when it runs alone: nothing hides them)
template <int I0, int N>
__device__ __forceinline__ void load_rows(const double* src, int64_t Bp, double (&x)[N][7]) {
#pragma unroll
  for (int i = 0; i < N; ++i)
#pragma unroll
    for (int k = 0; k < 7; ++k) x[i][k] = src[((I0 + i) * 7 + k) * Bp];
}

template <typename T>
__global__ __launch_bounds__(64, 4) void k_emit_steps(LatticeDev lat, const int32_t* __restrict__ step_slot,
                                                   const T* __restrict__ energy_in, int64_t Bp,
                                                   const double* __restrict__ products, const T* __restrict__ coefs,
                                                   int merge_pairs, T* __restrict__ steps_out, T* __restrict__ energy_out,
                                                   const int32_t* __restrict__ step_unit, int n_units,
                                                   float* __restrict__ units_out, float* __restrict__ extras_out,
                                                   StepEnergies<T> se) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  T* row = reinterpret_cast<T*>(smem_raw) + threadIdx.x * 68;
  const int64_t b_raw = (int64_t)blockIdx.x * 64 + threadIdx.x;
  const bool live = b_raw < lat.batch;
  const int64_t b = live ? b_raw : lat.batch - 1;
  const int s = blockIdx.y, S = lat.n_steps;
  const lynx_step st = lat.steps[s];
  const bool raw = st.kind == LYNX_STEP_CAVITY || (st.flags & LYNX_STEP_FLAG_RAW);
  const double* src = products + (int64_t)step_slot[s] * 49 * Bp + b;
  const bool pair_cavity = merge_pairs && s > 0 && steps_pair_up(lat.steps[s - 1], st);
  const bool pair_run = merge_pairs && s + 1 < S && steps_pair_up(st, lat.steps[s + 1]);
  bool entry_ill = false;
#pragma unroll
  for (int q = 49; q < LYNX_STEP_STRIDE; ++q) row[q] = T(0);
  if (pair_cavity) {
    // slot of the cavity <- T_cav . T_run: the product of the two ROUNDED maps, accumulated in float64
    const lynx_step sr = lat.steps[s - 1];
    const double* rsrc = products + (int64_t)step_slot[s - 1] * 49 * Bp + b;
    T R[49];
    {
      const bool run_raw = (sr.flags & LYNX_STEP_FLAG_RAW) != 0;
      const auto run_rows = [&](auto first, const auto& x) {  // rows first, first + 1, ... -> the row buffer, rounded
        constexpr int I0 = decltype(first)::value;
#pragma unroll
        for (int i = 0; i < (int)(sizeof(x) / sizeof(x[0])); ++i) {
          double o[7];
          emit_row_of(x[i], run_raw, o);
#pragma unroll
          for (int j = 0; j < 7; ++j) row[(I0 + i) * 7 + j] = (T)o[j];  // parked in the row buffer, then into registers
        }
      };
      double x[4][7];
      load_rows<0, 4>(rsrc, Bp, x);
      run_rows(std::integral_constant<int, 0>{}, x);
      double y[3][7];
      load_rows<4, 3>(rsrc, Bp, y);
      run_rows(std::integral_constant<int, 4>{}, y);
    }
    LYNX_FORGET();
#pragma unroll
    for (int q = 0; q < 49; ++q) R[q] = row[q];
    T c4[2] = {T(0), T(0)}, c5[2] = {T(0), T(0)};  // the cavity's (s, delta) block
    const auto cavity_row = [&](int i, const double (&x)[7]) {  // (the cavity's map is raw: its rows as they are)
      if (i == 4) {
        c4[0] = (T)x[4];
        c4[1] = (T)x[5];
      }
      if (i == 5) {
        c5[0] = (T)x[4];
        c5[1] = (T)x[5];
      }
#pragma unroll
      for (int j = 0; j < 7; ++j) {
        double acc = (double)(T)x[0] * (double)R[j];
#pragma unroll
        for (int k = 1; k < 7; ++k) acc = fma((double)(T)x[k], (double)R[k * 7 + j], acc);
        row[i * 7 + j] = (T)acc;
      }
    };
    {  // (two rows at a time: with the run's map in 49 registers there is no room for more in flight)
      double x[2][7];
      load_rows<0, 2>(src, Bp, x);
      cavity_row(0, x[0]);
      cavity_row(1, x[1]);
      load_rows<2, 2>(src, Bp, x);
      cavity_row(2, x[0]);
      cavity_row(3, x[1]);
      load_rows<4, 2>(src, Bp, x);
      cavity_row(4, x[0]);
      cavity_row(5, x[1]);
      double z[1][7];
      load_rows<6, 1>(src, Bp, z);
      cavity_row(6, z[0]);
    }
    T ci[4];
    entry_ill = cavity_entry_inverse<T>(c4[0], c4[1], c5[0], c5[1], ci);
#pragma unroll
    for (int q = 0; q < 4; ++q) row[LYNX_ENTRY_OFFSET + q] = ci[q];
    // the run's (s, delta) block, for the unit record (the row buffer's padding: not part of the table)
    row[64] = R[4 * 7 + 4];
    row[65] = R[4 * 7 + 5];
    row[66] = R[5 * 7 + 4];
    row[67] = R[5 * 7 + 5];
  } else {
    {
      const auto rows = [&](auto first, const auto& x) {
        constexpr int I0 = decltype(first)::value;
#pragma unroll
        for (int i = 0; i < (int)(sizeof(x) / sizeof(x[0])); ++i) {
          double o[7];
          emit_row_of(x[i], raw, o);
#pragma unroll
          for (int j = 0; j < 7; ++j) row[(I0 + i) * 7 + j] = (T)o[j];
        }
      };
      if constexpr (sizeof(T) == 4) {
        double x[4][7];
        load_rows<0, 4>(src, Bp, x);
        rows(std::integral_constant<int, 0>{}, x);
        double y[3][7];
        load_rows<4, 3>(src, Bp, y);
        rows(std::integral_constant<int, 4>{}, y);
      } else {  // (float64 rows cost twice the registers on their way out)
        double x[2][7];
        load_rows<0, 2>(src, Bp, x);
        rows(std::integral_constant<int, 0>{}, x);
        load_rows<2, 2>(src, Bp, x);
        rows(std::integral_constant<int, 2>{}, x);
        load_rows<4, 2>(src, Bp, x);
        rows(std::integral_constant<int, 4>{}, x);
        double z[1][7];
        load_rows<6, 1>(src, Bp, z);
        rows(std::integral_constant<int, 6>{}, z);
      }
    }
    if (pair_run) {  // rows 4 and 5 in front: they give the s and delta entering the cavity
      LYNX_FORGET();
      T t[14];
#pragma unroll
      for (int q = 0; q < 14; ++q) t[q] = row[28 + q];
#pragma unroll
      for (int q = 0; q < 14; ++q) row[q] = t[q];
    }
  }
  if (st.kind == LYNX_STEP_CAVITY) {
    const T* cs = coefs + (int64_t)s * 8 * Bp + b;
#pragma unroll
    for (int q = 0; q < 8; ++q) row[LYNX_COEF_OFFSET + q] = cs[q * Bp];
    row[LYNX_SINPHI_OFFSET] = t_sin<T>(row[LYNX_COEF_OFFSET + LYNX_C_PHI]);
  }
  row[LYNX_FLAGS_OFFSET] = (T)(step_descriptor(lat, s, merge_pairs) | (entry_ill ? LYNX_DESC_ILL : 0));
  if (s == S - 1) {
    const T e_out = energy_at_step<T>(lat, se, b, energy_in[b], S);
    row[LYNX_ENERGY_OFFSET] = e_out;
    if (energy_out && live) energy_out[b] = e_out;
  }
  LYNX_FORGET();
  if (live) {
    // the sample's row of the table: 64 scalars, contiguous, as 16-byte vectors
    using V = typename VecOf<T, true>::type;
    constexpr int W = VecOf<T, true>::width;
    V* dst = reinterpret_cast<V*>(steps_out + (b * (int64_t)S + s) * LYNX_STEP_STRIDE);
    const V* from = reinterpret_cast<const V*>(row);
#pragma unroll 4
    for (int v = 0; v < LYNX_STEP_STRIDE / W; ++v) dst[v] = from[v];
    // the compact record of the unit that applies this step's map (multi-step float32 programs, lynx_units.hpp):
    // packed here, from the row the lane still holds, instead of by a kernel of its own behind this one
    if constexpr (sizeof(T) == 4) {
      if (units_out) {
        const int code = step_unit[s];
        if (code >= 0) {
          const int64_t i = b * n_units + (code & 0xff);
          pack_unit_record<T>(row, row + 64, s, (code >> 8) & 3, (code >> 10) & 1, units_out + i * kUnitStride,
                              extras_out + i * kUnitExtraStride);
        }
      }
    }
  }
}

// Copy a precomputed step table of one sample from HBM into LDS (two-kernel mode).
template <typename T>
__device__ void load_steps_sample(const T* g_steps, int S, T* s_steps) {
  for (int i = threadIdx.x; i < S * LYNX_STEP_STRIDE; i += blockDim.x) s_steps[i] = g_steps[i];
  __syncthreads();
}

// ---------------------------------------------------------------------------------------
// Phase trigonometry of the cavity kick, float32.  The kick is cos(-s beta0 k + phi) - cos(phi)
// per particle and cavity (cavity.py:141-161); with 8 cavities it is ~40 % of the VALU work of
// BASELINE config 5.  The library's cosf/sincosf spend ~45 instructions on the common path
// because they must be ready for any argument (Payne-Hanek behind a branch).  For |x| <= 1000:
// 3-term Cody-Waite reduction by pi/2 with FMAs and the Cephes minimax polynomials on
// [-pi/4, pi/4] -- about 22 instructions, half of that per particle when two are packed.
// Measured on 2 M random arguments per range: max error 1.55 ulp up to |x| = 1000 (NumPy's
// float32 cos, the oracle's: 1.49), max absolute error 9.2e-8 (6e-8).  Larger arguments (a
// reference test sets phase = 48198468 degrees) and NaN/Inf take the library path.
// ---------------------------------------------------------------------------------------
constexpr float kFastTrigLimit = 1000.0f;
typedef float lynx_f32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ float vfma(float a, float b, float c) { return fmaf(a, b, c); }
__device__ __forceinline__ lynx_f32x2 vfma(lynx_f32x2 a, lynx_f32x2 b, lynx_f32x2 c) {
  return __builtin_elementwise_fma(a, b, c);
}
__device__ __forceinline__ float vrint(float a) { return __builtin_rintf(a); }
__device__ __forceinline__ lynx_f32x2 vrint(lynx_f32x2 a) {
  return lynx_f32x2{__builtin_rintf(a.x), __builtin_rintf(a.y)};
}

// quadrant q (as a float) of x, |x| <= kFastTrigLimit, and sin / cos of the reduced argument
template <typename V>
__device__ __forceinline__ void sincos_reduced(V x, V& q, V& sr, V& cr) {
  q = vrint(x * 0.636619772f);
  V r = vfma(q, V(-1.5707963705062866f), x);
  r = vfma(q, V(4.3711388286737929e-08f), r);
  r = vfma(q, V(1.7151245100059975e-15f), r);
  const V z = r * r;
  V ps = vfma(z, V(-1.9515295891e-4f), V(8.3321608736e-3f));
  ps = vfma(ps, z, V(-1.6666654611e-1f));
  sr = vfma(ps * z, r, r);
  V pc = vfma(z, V(2.443315711809948e-5f), V(-1.388731625493765e-3f));
  pc = vfma(pc, z, V(4.166664568298827e-2f));
  cr = vfma(pc * z, z, vfma(z, V(-0.5f), V(1.0f)));
}

__device__ __forceinline__ void quadrant_select(float q, float sr, float cr, float& s, float& c) {
  const int m = (int)q & 3;
  const float ss = (m & 1) ? cr : sr, cc = (m & 1) ? sr : cr;
  s = (m & 2) ? -ss : ss;
  c = ((m + 1) & 2) ? -cc : cc;
}

__device__ __forceinline__ void phase_sincos(float x, float& s, float& c) {
  if (__builtin_expect(!(__builtin_fabsf(x) <= kFastTrigLimit), 0)) {
    s = t_sin(x);
    c = t_cos(x);
    return;
  }
  float q, sr, cr;
  sincos_reduced<float>(x, q, sr, cr);
  quadrant_select(q, sr, cr, s, c);
}
__device__ __forceinline__ void phase_sincos(double x, double& s, double& c) {
  s = t_sin(x);
  c = t_cos(x);
}
__device__ __forceinline__ void phase_sincos(lynx_f32x2 x, lynx_f32x2& s, lynx_f32x2& c) {
  float s0, c0, s1, c1;
  const float big = __builtin_fmaxf(__builtin_fabsf(x.x), __builtin_fabsf(x.y));
  // whole wave inside the first quadrant's half width: quadrant 0 for everybody -- no reduction, no selection (the
  // general path's result for q = 0, bit for bit)
  if (__builtin_amdgcn_ballot_w64(!(big <= 0.75f)) == 0) {
    const lynx_f32x2 z = x * x;
    lynx_f32x2 ps = vfma(z, lynx_f32x2(-1.9515295891e-4f), lynx_f32x2(8.3321608736e-3f));
    ps = vfma(ps, z, lynx_f32x2(-1.6666654611e-1f));
    s = vfma(ps * z, x, x);
    lynx_f32x2 pc = vfma(z, lynx_f32x2(2.443315711809948e-5f), lynx_f32x2(-1.388731625493765e-3f));
    pc = vfma(pc, z, lynx_f32x2(4.166664568298827e-2f));
    c = vfma(pc * z, z, vfma(z, lynx_f32x2(-0.5f), lynx_f32x2(1.0f)));
    return;
  }
  if (__builtin_expect(!(big <= kFastTrigLimit), 0)) {
    phase_sincos(x.x, s0, c0);
    phase_sincos(x.y, s1, c1);
  } else {
    lynx_f32x2 q, sr, cr;
    sincos_reduced<lynx_f32x2>(x, q, sr, cr);
    quadrant_select(q.x, sr.x, cr.x, s0, c0);
    quadrant_select(q.y, sr.y, cr.y, s1, c1);
  }
  s = lynx_f32x2{s0, s1};
  c = lynx_f32x2{c0, c1};
}
// The forward kernels need the cosine alone: one polynomial on [-pi/2, pi/2] (cos r = 1 - z/2 + z^2 q(z),
// z = r^2) and the sign of the half-period, instead of both quadrant polynomials and a select -- 23 instead
// of 40 instructions per pair of particles in the cavity step.  Absolute error <= 7.5e-8 for |x| <= pi/2
// (no reduction; what rounding the result to float costs anyway), <= 1.3e-7 beyond (the reduced argument is
// itself rounded to float); the kick is DKICK (cos a - cos phi), so the absolute error is the one that counts.
template <typename V>
__device__ __forceinline__ V cos_half_period(V x, V& n) {
  n = vrint(x * 0.31830987334251404f);
  V r = vfma(n, V(-3.1415927410125732f), x);
  r = vfma(n, V(8.742277657347586e-08f), r);
  r = vfma(n, V(3.4302490200117637e-15f), r);
  const V z = r * r;
  V q = vfma(z, V(-2.654408035596134e-07f), V(2.478602073097136e-05f));
  q = vfma(q, z, V(-0.0013888812391087413f));
  q = vfma(q, z, V(0.0416666679084301f));
  return vfma(q * z, z, vfma(z, V(-0.5f), V(1.0f)));
}
__device__ __forceinline__ float negate_if_odd(float c, float n) {
  return __int_as_float(__float_as_int(c) ^ ((int)n << 31));
}
__device__ __forceinline__ float phase_cos(float x) {
  if (__builtin_expect(!(__builtin_fabsf(x) <= kFastTrigLimit), 0)) return t_cos(x);
  float n;
  const float c = cos_half_period<float>(x, n);
  return negate_if_odd(c, n);
}
__device__ __forceinline__ double phase_cos(double x) { return t_cos(x); }
__device__ __forceinline__ lynx_f32x2 phase_cos(lynx_f32x2 x) {
  const float big = __builtin_fmaxf(__builtin_fabsf(x.x), __builtin_fabsf(x.y));
  // Cavity phases are a few degrees and s beta0 k a few milliradians: when the WHOLE WAVE sits inside |x| <= 1.5 the
  // half-period index is 0 for everybody, so the reduction (x - n pi in three steps) and the sign (n's parity) drop
  // out -- bit for bit what the general path returns for n = 0, at 9 instead of 21 instructions per pair
  // (BASELINE config 5 forward: 0.765 -> 0.72 ms/step).
  if (__builtin_amdgcn_ballot_w64(!(big <= 1.5f)) == 0) {
    const lynx_f32x2 z = x * x;
    lynx_f32x2 q = vfma(z, lynx_f32x2(-2.654408035596134e-07f), lynx_f32x2(2.478602073097136e-05f));
    q = vfma(q, z, lynx_f32x2(-0.0013888812391087413f));
    q = vfma(q, z, lynx_f32x2(0.0416666679084301f));
    return vfma(q * z, z, vfma(z, lynx_f32x2(-0.5f), lynx_f32x2(1.0f)));
  }
  lynx_f32x2 out;
  if (__builtin_expect(!(big <= kFastTrigLimit), 0)) {
    out.x = phase_cos(x.x);
    out.y = phase_cos(x.y);
  } else {
    lynx_f32x2 n;
    const lynx_f32x2 c = cos_half_period<lynx_f32x2>(x, n);
    out.x = negate_if_odd(c.x, n.x);
    out.y = negate_if_odd(c.y, n.y);
  }
  return out;
}

// sin(d) and cos(d) - 1, the second WITHOUT the cancellation of `cos(d) - 1` (d is s beta0 k: milliradians)
__device__ __forceinline__ void sin_cosm1(double d, double& sd, double& cm1) {
  sd = t_sin(d);
  const double h = t_sin(0.5 * d);
  cm1 = -2.0 * h * h;
}
__device__ __forceinline__ void sin_cosm1(float d, float& sd, float& cm1) {
  if (__builtin_expect(__builtin_fabsf(d) <= 0.75f, 1)) {  // the polynomials of sincos_reduced, quadrant 0, without the "1 +"
    const float z = d * d;
    float ps = fmaf(z, -1.9515295891e-4f, 8.3321608736e-3f);
    ps = fmaf(ps, z, -1.6666654611e-1f);
    sd = fmaf(ps * z, d, d);
    float pc = fmaf(z, 2.443315711809948e-5f, -1.388731625493765e-3f);
    pc = fmaf(pc, z, 4.166664568298827e-2f);
    cm1 = fmaf(pc * z, z, z * -0.5f);
  } else {
    float sh, ch;
    phase_sincos(0.5f * d, sh, ch);
    sd = 2.0f * sh * ch;
    cm1 = -2.0f * sh * sh;
  }
}
__device__ __forceinline__ void sin_cosm1(lynx_f32x2 d, lynx_f32x2& sd, lynx_f32x2& cm1) {
  const float big = __builtin_fmaxf(__builtin_fabsf(d.x), __builtin_fabsf(d.y));
  if (__builtin_amdgcn_ballot_w64(!(big <= 0.75f)) == 0) {  // whole wave (as good as always): packed
    const lynx_f32x2 z = d * d;
    lynx_f32x2 ps = vfma(z, lynx_f32x2(-1.9515295891e-4f), lynx_f32x2(8.3321608736e-3f));
    ps = vfma(ps, z, lynx_f32x2(-1.6666654611e-1f));
    sd = vfma(ps * z, d, d);
    lynx_f32x2 pc = vfma(z, lynx_f32x2(2.443315711809948e-5f), lynx_f32x2(-1.388731625493765e-3f));
    pc = vfma(pc, z, lynx_f32x2(4.166664568298827e-2f));
    cm1 = vfma(pc * z, z, z * -0.5f);
    return;
  }
  float s0, c0, s1, c1;
  sin_cosm1(d.x, s0, c0);
  sin_cosm1(d.y, s1, c1);
  sd = lynx_f32x2{s0, s1};
  cm1 = lynx_f32x2{c0, c1};
}


// cos(phi + d) - cos(phi) from d and the sample's sin(phi), cos(phi): cos(phi)(cos d - 1) - sin(phi) sin d.
// The reference subtracts two float32 cosines (cavity.py:150-160).  On a short bunch -- d = -s beta0 k is milliradians --
// they agree to four or five digits, every particle of a sample sits at practically the same argument, and the
// cosines' rounding (6e-8, half an ulp at 1) is therefore COMMON to the sample: it moves the mean of delta by up to
// 5e-4 of |mu_p| + sigma_p on BASELINE config 5's bench beam, in the reference's own float32 arithmetic (float32 oracle
// against float64 oracle, all 4096 environments) as in any other float32 evaluation of that difference.  Formed this
// way the difference carries a RELATIVE error of 1e-7, the product's moments are within 1e-4 of the float64 chain in
// every environment, and with that as close to the float32 chain as that chain is to float64
// (tests/test_gpu_parity.py: test_config_5_on_the_bench_beam_...).  Same instruction count as the cosine it replaces.
// |d| <= 0.25 rad (9 mm of s at 1.3 GHz): Taylor polynomials -- sin d to d^5 (relative error d^6/5040 < 5e-8), cos d - 1 to
// d^6 (< 1e-8) -- in ten packed operations where the cosine they replace took eight and its argument and the
// subtraction two more.  (sin_cosm1's longer polynomials up to 0.75 rad: fourteen, BASELINE config 5's streaming kernel
// 0.521 -> 0.553 ms.)  Beyond: the two cosines as the reference subtracts them -- they no longer cancel there.
constexpr float kCosDifferenceSmall = 0.25f;
template <typename V>
__device__ __forceinline__ V cos_difference_small(V d, float sphi, float cphi) {
  const V z = d * d;
  const V p = vfma(z, V(8.3333333333e-3f), V(-1.6666666667e-1f));
  const V sd = vfma(p * z, d, d);                                               // sin d
  V q = vfma(z, V(-1.3888888889e-3f), V(4.1666666667e-2f));
  q = vfma(q, z, V(-0.5f));
  return vfma(q * z, V(cphi), sd * (-sphi));                                    // cos phi (cos d - 1) - sin phi sin d
}
__device__ __forceinline__ float cos_difference(float d, float phi, float sphi, float cphi) {
  if (__builtin_expect(__builtin_fabsf(d) <= kCosDifferenceSmall, 1)) return cos_difference_small<float>(d, sphi, cphi);
  return phase_cos(d + phi) - cphi;
}
__device__ __forceinline__ lynx_f32x2 cos_difference(lynx_f32x2 d, float phi, float sphi, float cphi) {
  const float big = __builtin_fmaxf(__builtin_fabsf(d.x), __builtin_fabsf(d.y));
  if (__builtin_amdgcn_ballot_w64(!(big <= kCosDifferenceSmall)) == 0) return cos_difference_small<lynx_f32x2>(d, sphi, cphi);  // whole wave
  // element by element (what a particle gets does not depend on its wave), one after the other: this path is rare and
  // must not cost the kernels around it registers -- with both elements' two forms alive at once the streaming kernel
  // of multi-step programs took 92 registers instead of 88, and the next call's build waves (136) then waited for TWO
  // of its waves to retire on a SIMD instead of one (k_build_pieces 97 -> 357 us underneath BASELINE config 5's kernel)
  lynx_f32x2 out;
  out.x = cos_difference(d.x, phi, sphi, cphi);
  __builtin_amdgcn_sched_barrier(0);
  out.y = cos_difference(d.y, phi, sphi, cphi);
  return out;
}

// What the float32 kick leaves in s and delta (cavity.py:151-161, :219-226), given the difference of cosines: the
// reference's sums as chains of fused multiply-adds like the linear maps' rows (deviation (iv)) --
// 7 operations where the expression as written takes 12, a ninth of a unit of BASELINE config 5.  ONE function for the
// scalar and the packed form, the step loop and the unit loops: they agree bit for bit.
template <typename V>
__device__ __forceinline__ void kick_outputs(const float* coef, V s_in, V d_in, V dcos, V& s_out, V& d_out) {
  d_out = vfma(dcos, V(coef[LYNX_C_DKICK]), d_in * coef[LYNX_C_DSCALE]);
  const V t = vfma(d_in, V(coef[LYNX_C_T566]), s_in * coef[LYNX_C_T556]);  // T566 delta + T556 s
  s_out = vfma(s_in * s_in, V(coef[LYNX_C_T555]), vfma(t, d_in, s_out));
}

// Non-linear cavity step on the device.  float64: the expression of cavity_kick<T> (lynx_maps.hpp, the one the host
// harness checks against the oracle).  float32: the same with the difference of cosines from cos_difference;
// `coef` points at the step-table row's coefficients, sin(phi) sits behind them (LYNX_SINPHI_OFFSET).
template <typename T>
__device__ __forceinline__ void device_cavity_kick(const T* coef, T s_in, T d_in, T& s_out, T& d_out) {
  if constexpr (sizeof(T) == 4) {
    const float dcos = cos_difference(T(-1) * s_in * coef[LYNX_C_BK], coef[LYNX_C_PHI], coef[LYNX_SINPHI_OFFSET - LYNX_COEF_OFFSET],
                                      coef[LYNX_C_COSPHI]);
    kick_outputs<float>(coef, s_in, d_in, dcos, s_out, d_out);
  } else {
    d_out = d_in * coef[LYNX_C_DSCALE] +
            coef[LYNX_C_DKICK] * (phase_cos(T(-1) * s_in * coef[LYNX_C_BK] + coef[LYNX_C_PHI]) - coef[LYNX_C_COSPHI]);
    s_out = s_out + (coef[LYNX_C_T566] * (d_in * d_in) + coef[LYNX_C_T556] * s_in * d_in +
                     coef[LYNX_C_T555] * (s_in * s_in));
  }
}

// Apply one step of the program to one particle held in registers.
//   run:    z <- T z                                  (element.py:85, `particles @ tm^T`)
//   cavity: z <- T z, then the non-linear delta / s update from the INCOMING s, delta
//           (cavity.py:141-161, 219-226)
template <typename T>
__device__ __forceinline__ void apply_step(const T* M /*49 + coef*/, int step_kind, int step_flags,
                                           T (&z)[7]) {
  T o[7];
#pragma unroll
  for (int i = 0; i < 7; ++i) {
    // float64: the map stays in LDS; fetch it row by row instead of letting the compiler hoist
    // all 49 entries into 98 VGPRs (that alone cost one of the three waves per SIMD)
    if (sizeof(T) == 8) LYNX_FORGET();
    T acc = z[0] * M[i * 7 + 0];
#pragma unroll
    for (int j = 1; j < 7; ++j) acc = t_fma(z[j], M[i * 7 + j], acc);
    o[i] = acc;
  }
  if (step_kind == LYNX_STEP_CAVITY && (step_flags & LYNX_FLAG_CAV_GAIN)) {
    device_cavity_kick<T>(M + LYNX_COEF_OFFSET, z[4], z[5], o[4], o[5]);
  }
#pragma unroll
  for (int i = 0; i < 7; ++i) z[i] = o[i];
}

// Two particles per lane as packed float pairs: every multiply-add of the step becomes one
// v_pk_fma_f32 with the map entry broadcast from an SGPR (op_sel), i.e. half the VALU issue
// slots of two scalar applications.  Only multi-step float32 programs use it: they are
// VALU-bound (BASELINE config 5: 16 maps + 8 cosines per particle), whereas the single-map
// stream is HBM-bound and measured 10 % slower with packed math.  Same operations in the same
// order per component as apply_step<float>, so the results are bit-identical.

__device__ __forceinline__ lynx_f32x2 pk_fma(lynx_f32x2 a, float b, lynx_f32x2 c) {
  return __builtin_elementwise_fma(a, (lynx_f32x2)(b), c);
}

// `entry`: where the s and delta that drive a cavity's kick come from --
//   kEntryOwn     the pair's own components 4, 5 (a cavity step on its own),
//   kEntryGiven   s_entry, d_entry (merged [run, cavity] pair, reverse pass: rows 4, 5 of T_run applied to z),
//   kEntryInverse merged pair, forward pass: the inverse of T_cav's (s, delta) block (LYNX_ENTRY_OFFSET) applied
//                 to components 4, 5 of M z, M = T_cav . T_run
//   kEntryStash   merged pair whose block is ill-conditioned (LYNX_DESC_ILL), forward pass: s_entry, d_entry as in
//                 kEntryGiven, parked in this lane's four floats of LDS while the 7x7 is applied (carrying them in
//                 registers cost the step loop 10 of them and the multi-step kernels spills)
enum { kEntryOwn = 0, kEntryGiven = 1, kEntryInverse = 2, kEntryStash = 3 };

__device__ __forceinline__ void apply_step_pair(const float* M /*49 + coef (+ 4)*/, int step_kind, int step_flags,
                                                lynx_f32x2 (&z)[7], int entry = kEntryOwn,
                                                lynx_f32x2 s_entry = lynx_f32x2{0.f, 0.f},
                                                lynx_f32x2 d_entry = lynx_f32x2{0.f, 0.f},
                                                const float* stash = nullptr) {
  lynx_f32x2 o[7];
#pragma unroll
  for (int i = 0; i < 7; ++i) {
    lynx_f32x2 acc = z[0] * M[i * 7 + 0];
#pragma unroll
    for (int j = 1; j < 7; ++j) acc = pk_fma(z[j], M[i * 7 + j], acc);
    o[i] = acc;
  }
  if (step_kind == LYNX_STEP_CAVITY && (step_flags & LYNX_FLAG_CAV_GAIN)) {
    const float* coef = M + LYNX_COEF_OFFSET;
    lynx_f32x2 s_in = z[4], d_in = z[5];
    if (entry == kEntryGiven) {
      s_in = s_entry;
      d_in = d_entry;
    } else if (entry == kEntryInverse) {
      const float* ci = M + LYNX_ENTRY_OFFSET;
      s_in = pk_fma(o[5], ci[1], o[4] * ci[0]);
      d_in = pk_fma(o[5], ci[3], o[4] * ci[2]);
    } else if (entry == kEntryStash) {
      const lynx_f32x4 v = *reinterpret_cast<const lynx_f32x4*>(stash);
      s_in = lynx_f32x2{v.x, v.y};
      d_in = lynx_f32x2{v.z, v.w};
    }
    const lynx_f32x2 dcos = cos_difference(-1.0f * s_in * coef[LYNX_C_BK], coef[LYNX_C_PHI], coef[LYNX_SINPHI_OFFSET - LYNX_COEF_OFFSET],
                                           coef[LYNX_C_COSPHI]);
    kick_outputs<lynx_f32x2>(coef, s_in, d_in, dcos, o[4], o[5]);
  }
#pragma unroll
  for (int i = 0; i < 7; ++i) z[i] = o[i];
}

// [run, cavity] pair in merged form, as the REVERSE pass walks it: `pre` = rows 4 and 5 of the run's map
// (the run's slot), `M` = T_cav . T_run with the cavity's coefficients.  One 7x7 application + two rows
// instead of two applications:
//   z_cav_in[4], z_cav_in[5] = pre . z (merged_pair_entry);  o = M z;  kick from those two.
// Same algebra as the two steps one after the other; the product is rounded once more in
// k_build and once less per particle.  (The forward kernel gets the two from M z itself: kEntryInverse.)
__device__ __forceinline__ void merged_pair_entry(const float* pre /*14*/, const lynx_f32x2 (&z)[7],
                                                  lynx_f32x2& s_in, lynx_f32x2& d_in) {
  s_in = z[0] * pre[0];
  d_in = z[0] * pre[7];
#pragma unroll
  for (int j = 1; j < 7; ++j) {
    s_in = pk_fma(z[j], pre[j], s_in);
    d_in = pk_fma(z[j], pre[7 + j], d_in);
  }
}

// ---------------------------------------------------------------------------------------
// k_build: standalone build+compose, one workgroup of 256 threads per sample; chunks of up to 64 elements
// when the batch fills the GPU, of up to 128 when it does not (a single sample's 128-element lattice is
// then 7 tree levels deep instead of two rounds of 6; lynx_hip.hip: build_shape).
// (Round 4 tried the tree's 7x7 products on the matrix cores -- v_mfma_f64_16x16x4_f64, two products per wave as the
// diagonal blocks of a 14x14 one: correct, 205 GPU tests green, and SLOWER: 13.8 us against 11.3 for one sample's
// 128-element float64 lattice.  A level is latency, not arithmetic: an LDS read, two dependent matrix instructions and
// an LDS write per pair of products, in a loop the waves walk serially; the vector form keeps seven independent
// multiply-add chains per lane in flight.  Where the 11 us go: launch 2.5, element maps 4.7 (one float64 quadrupole:
// sqrt, cos, sin, cosh, sinh and six divisions, one dependent chain), tree ~3, table write 0.5.)
// Depends on the lattice and the incoming energy only -- not on the particles -- so consecutive
// `track` calls can run it on a second stream underneath the previous call's streaming kernel.
// The outgoing energy is parked in the last step's row (LYNX_ENERGY_OFFSET) for the streaming
// kernel to publish; `energy_out` is only used by the synchronous lynx_build_compose entry.
// ---------------------------------------------------------------------------------------
template <typename T>
__device__ __forceinline__ void build_to_table(const LatticeDev& lat, const T* __restrict__ energy_in,
                                               T* __restrict__ steps_out, T* __restrict__ energy_out,
                                               int chunk, int merge_pairs) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  T* s_steps = reinterpret_cast<T*>(smem_raw + build_scratch_bytes(chunk, sizeof(T)));
  T* s_energy = s_steps + (size_t)lat.n_steps * LYNX_STEP_STRIDE;
  const int64_t b = blockIdx.x;
  // zero the coefficient slots so that run steps have defined padding
  for (int i = threadIdx.x; i < lat.n_steps * LYNX_STEP_STRIDE; i += blockDim.x) s_steps[i] = T(0);
  __syncthreads();
  build_compose_sample<T>(lat, b, energy_in[b], s_steps, s_energy, smem_raw, chunk);
  table_sinphi<T>(lat, s_steps);  // (its slot is nobody else's: the barriers below come before the rows leave)
  if (merge_pairs) {
    // [run, cavity] pairs for the streaming kernel: slot of the cavity <- T_cav . T_run (one
    // 7x7 application per pair instead of two), slot of the run <- rows 4 and 5 of T_run, which
    // give the s and delta that ENTER the cavity and drive its non-linear kick
    for (int s = 1; s < lat.n_steps; ++s) {
      if (!steps_pair_up(lat.steps[s - 1], lat.steps[s])) continue;  // uniform
      T* run = s_steps + (s - 1) * LYNX_STEP_STRIDE;
      T* cav = s_steps + s * LYNX_STEP_STRIDE;
      T v = T(0);
      if (threadIdx.x < 49) {  // product of the two rounded maps, accumulated in float64, rounded once
        const int i = threadIdx.x / 7, j = threadIdx.x - i * 7;
        double acc = (double)cav[i * 7] * (double)run[j];
#pragma unroll
        for (int k = 1; k < 7; ++k) acc = fma((double)cav[i * 7 + k], (double)run[k * 7 + j], acc);
        v = (T)acc;
      }
      T ci[4] = {T(0), T(0), T(0), T(0)};
      bool ill = false;
      if (threadIdx.x == 49) ill = cavity_entry_inverse<T>(cav[4 * 7 + 4], cav[4 * 7 + 5], cav[5 * 7 + 4], cav[5 * 7 + 5], ci);
      __syncthreads();
      if (threadIdx.x < 49) cav[threadIdx.x] = v;
      if (threadIdx.x == 49) {
#pragma unroll
        for (int q = 0; q < 4; ++q) cav[LYNX_ENTRY_OFFSET + q] = ci[q];
        cav[LYNX_FLAGS_OFFSET] = ill ? T(1) : T(0);  // until the descriptors are written below
      }
      if (threadIdx.x >= 64 && threadIdx.x < 78) run[threadIdx.x - 64] = run[28 + (threadIdx.x - 64)];
      __syncthreads();
    }
  }
  for (int s = threadIdx.x; s < lat.n_steps; s += blockDim.x) {
    const bool ill = s_steps[s * LYNX_STEP_STRIDE + LYNX_FLAGS_OFFSET] != T(0);  // zero unless set in the loop above
    s_steps[s * LYNX_STEP_STRIDE + LYNX_FLAGS_OFFSET] = (T)(step_descriptor(lat, s, merge_pairs) | (ill ? LYNX_DESC_ILL : 0));
  }
  if (lat.n_steps > 0 && threadIdx.x == 0)
    s_steps[(lat.n_steps - 1) * LYNX_STEP_STRIDE + LYNX_ENERGY_OFFSET] = s_energy[lat.n_steps];
  __syncthreads();
  T* dst = steps_out + b * (int64_t)lat.n_steps * LYNX_STEP_STRIDE;
  for (int i = threadIdx.x; i < lat.n_steps * LYNX_STEP_STRIDE; i += blockDim.x) dst[i] = s_steps[i];
  if (energy_out && threadIdx.x == 0) energy_out[b] = s_energy[lat.n_steps];
}

template <typename T>
__global__ __launch_bounds__(256) void k_build(LatticeDev lat, const T* __restrict__ energy_in,
                                                T* __restrict__ steps_out, T* __restrict__ energy_out,
                                                int chunk, int merge_pairs) {
  build_to_table<T>(lat, energy_in, steps_out, energy_out, chunk, merge_pairs);
}

// ... of a small lattice, whose parameters come with the arguments (InlinePool)
template <typename T, int BYTES>
__global__ __launch_bounds__(256) void k_build_inline(InlinePool<BYTES> /* read in place: inline_pool_view */, LatticeDev lat,
                                                       const T* __restrict__ energy_in, T* __restrict__ steps_out,
                                                       T* __restrict__ energy_out, int chunk, int merge_pairs) {
  build_to_table<T>(inline_pool_view(lat), energy_in, steps_out, energy_out, chunk, merge_pairs);
}


struct TrackArgs {
  int64_t n_particles;
  int32_t chunks;        // workgroups per sample
  int32_t tiles_per_wg;
  int32_t fused_build;   // 1: build+compose in the prologue, 0: read steps_in
  int32_t store;         // 1: write p_out
  int32_t lds_scratch_bytes;  // build scratch / moment-reduction slab in front of the step table
  int32_t build_chunk;   // fused prologue: elements per compose round
  int64_t in_stride;     // scalars between the samples of p_in: N*7, or 0 for one shared incoming beam
  int32_t merged_pairs;  // the step table holds [run, cavity] pairs in merged form (the builder marks them: LYNX_DESC_PAIR)
  int32_t n_observers;   // LYNX_STEP_FLAG_OBSERVE steps of the program (their sums live behind the step table in LDS)
  // "this kernel has reached its tail": workgroup `tail_wg` (one of the last to be dispatched) writes `tail_seq` to
  // `tail_flag` when it starts.  The build stream waits for the flag of the kernel BEFORE the one in front of it
  // (hipStreamWaitValue32), so that the next call's build starts where the GPU is running empty instead of at the
  // head of a streaming kernel, where its waves take slots from it for hundreds of microseconds.
  uint32_t tail_seq;
  int32_t tail_wg;
  unsigned int* tail_flag;
  int32_t stash_offset;  // merged tables: byte offset in LDS of the lanes' four-float stashes (kEntryStash), 16-byte aligned
  int32_t reversed;      // 1: the workgroups walk the batch from its end (see track_particles_t: every other long call)
};

__device__ __forceinline__ void announce_tail(const TrackArgs& a) {
  if (a.tail_flag && blockIdx.x == (unsigned)a.tail_wg && threadIdx.x == 0)
    __hip_atomic_store(a.tail_flag, a.tail_seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

__device__ __forceinline__ float uniform_value(float v) {
  return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(v)));
}
__device__ __forceinline__ double uniform_value(double v) { return v; }  // stays in VGPRs / LDS
// value of the first active lane, as a wave-uniform (scalar-register) value
__device__ __forceinline__ float wave_first(float v) { return uniform_value(v); }
__device__ __forceinline__ double wave_first(double v) {
  const long long bits = __double_as_longlong(v);
  const int lo = __builtin_amdgcn_readfirstlane((int)(bits & 0xffffffffll));
  const int hi = __builtin_amdgcn_readfirstlane((int)(bits >> 32));
  return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}

// ---------------------------------------------------------------------------------------
// Moment partial records.  A record holds sums over some particles of d = z - c and d d^T for a
// reference point c near the beam (so that float32 partial sums do not cancel), layout:
//   [0..5] sum d_i   [6] sum z_6   [7..27] sum d_i d_j (i <= j, row-major)   [28..33] c   [35] count
// Every wave takes ITS OWN first tracked particle as c -- no pass over the beam, no dependence of
// the map build on the particles -- and records are moved to a common reference point when they
// are added up (workgroup reduction, k_finalize_moments), exactly, in float64:
//   sum (z - c0)_i                = D_i + n e_i                           e = c - c0
//   sum (z - c0)_i (z - c0)_j     = DD_ij + e_i D_j + D_i e_j + n e_i e_j
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ double moment_slot_moved(const double* row, const double* c0, int j) {
  if (j == 6 || j >= 28) return row[j];
  const double n = row[35];
  if (j < 6) return row[j] + n * (row[28 + j] - c0[j]);
  int k = j - 7, i = 0, len = 6;  // upper-triangle index -> (i, jj)
  while (k >= len) { k -= len; --len; ++i; }
  const int jj = i + k;
  const double ei = row[28 + i] - c0[i], ej = row[28 + jj] - c0[jj];
  return row[j] + ei * row[jj] + row[i] * ej + n * ei * ej;
}

// Which second moments a pass accumulates.
//   FULL = false  the ones the reference's ParticleBeam exposes as properties (particle_beam.py:736-836):
//                 the six variances, sigma_xx' and sigma_yy' -- 8 sums;
//   FULL = true   the whole upper triangle of the 6x6 covariance -- 21 sums.
// Slot of sum k inside a record's [7, 28) block (row-major upper triangle, include/lynx_hip.h):
template <bool FULL> struct MomentSet {
  static constexpr int N = FULL ? 21 : 8;
  __host__ __device__ static constexpr int row(int k) { return FULL ? tri_row(k) : (k < 6 ? k : (k == 6 ? 0 : 2)); }
  __host__ __device__ static constexpr int col(int k) { return FULL ? tri_col(k) : (k < 6 ? k : (k == 6 ? 1 : 3)); }
  __host__ __device__ static constexpr int slot(int k) { return 7 + tri_index(row(k), col(k)); }
  __host__ __device__ static constexpr int tri_index(int i, int j) { return i * 6 - (i * (i - 1)) / 2 + (j - i); }
  __host__ __device__ static constexpr int tri_row(int k) {
    int i = 0, len = 6;
    while (k >= len) { k -= len; --len; ++i; }
    return i;
  }
  __host__ __device__ static constexpr int tri_col(int k) {
    int i = 0, len = 6;
    while (k >= len) { k -= len; --len; ++i; }
    return i + k;
  }
};

template <typename T, bool FULL> struct MomentAcc {
  using Set = MomentSet<FULL>;
  double d[6], dd[Set::N], one, cnt;
  __device__ __forceinline__ void clear() {
#pragma unroll
    for (int i = 0; i < 6; ++i) d[i] = 0.0;
#pragma unroll
    for (int i = 0; i < Set::N; ++i) dd[i] = 0.0;
    one = 0.0;
    cnt = 0.0;
  }
  template <int K> __device__ __forceinline__ void products(const double (&e)[6]) {
    if constexpr (K < Set::N) {
      constexpr int r = Set::row(K), c = Set::col(K);  // constant-evaluated here: as call operands they were not
      dd[K] = fma(e[r], e[c], dd[K]);
      products<K + 1>(e);
    }
  }
  __device__ __forceinline__ void add(const T (&z)[7], const T (&shift)[6]) {
    double e[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) {
      e[i] = (double)z[i] - (double)shift[i];
      d[i] += e[i];
    }
    one += (double)z[6];
    cnt += 1.0;
    products<0>(e);
  }
};

// ---------------------------------------------------------------------------------------
// k_track_direct: the streaming kernel.  grid.x = B * chunks; each 256-thread workgroup owns
// `tiles_per_wg` tiles of 256 * UNROLL particles of ONE sample and every lane carries UNROLL
// particles per iteration in registers.  There is no workgroup barrier in the loop, the map sits in
// SGPRs (fp32) or is re-read from LDS (fp64), so occupancy -- not a tile pipeline -- hides the memory
// latency; the loads of a tile are issued one iteration ahead of its use.
//
// Two ways from HBM to a lane's registers (XPOSE):
//
//   false  lane l of a wave owns particles (base + l + 256 u): it loads its 7 scalars straight from
//          HBM with two (fp32: 16 B + 12 B) or four (fp64: 3 x 16 B + 8 B) dword-aligned vector loads.
//          A wave instruction sweeps a 64 x 28 B (56 B) span in 16-byte pieces 28 (56) bytes apart, i.e.
//          every instruction touches every 128-B line of the span: traffic stays at the algorithmic
//          2 x 7 x sizeof(T) per particle, but the address path works 2x (fp32) / 4x (fp64) harder
//          per byte than a plain copy does.
//   true   a WAVE owns 64 * P consecutive particles (P = 16 / sizeof(T): 4 fp32, 2 fp64) = 7168
//          bytes = exactly seven full-width 1-KiB wave loads (global_load_dwordx4, every lane 16
//          contiguous bytes: the access shape of the fastest plain copy).  The tile goes through a
//          wave-PRIVATE 7-KiB LDS region as a flat array (ds_write_b128, conflict-free) and comes
//          back transposed: lane l reads bytes [112 l, 112 l + 112) = its P consecutive particles
//          (7 x ds_read_b128, 112-byte lane stride: conflict-free).  LDS operations of one wave
//          execute in order, so no barrier is involved -- only a compiler fence.  Results leave
//          the same way.  (Round 1's LDS-tiled kernel transposed per WORKGROUP, with three
//          barriers per tile, and lost to the direct form; this one has none.)
//          Needs UNROLL == P.  A wave tile that is cut by the end of the sample falls back to the
//          per-particle accesses of the other form.
//
// MOM: 0 = no moments, 1 = float64 accumulation per particle, 2 = per-iteration float32
// partial sums (UNROLL particles, shifted by the wave's reference point) folded into
// float64 accumulators once per iteration, 3 = float32 per-lane sums for the whole
// workgroup lifetime (only chosen when a lane sees <= 32 particles; 58 fewer VGPRs), the
// float64 part then starts at the workgroup reduction.
// ---------------------------------------------------------------------------------------
typedef float lynx_f32x4u __attribute__((ext_vector_type(4), aligned(4)));
typedef float lynx_f32x3u __attribute__((ext_vector_type(3), aligned(4)));
typedef double lynx_f64x2u __attribute__((ext_vector_type(2), aligned(8)));
typedef unsigned int lynx_u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int lynx_u32x4u __attribute__((ext_vector_type(4), aligned(4)));

// A particle is 7 scalars = 28 / 56 bytes: it starts on a 4- / 8-byte boundary only, hence
// the under-aligned vector types (global dwordx4 / dwordx3 / dwordx2 accesses need no more).
// Non-temporal variants of these accesses were measured 10-15 % slower and are gone.
__device__ __forceinline__ void load_particle(const float* p, float (&z)[7]) {
  const lynx_f32x4u a = *reinterpret_cast<const lynx_f32x4u*>(p);
  const lynx_f32x3u b = *reinterpret_cast<const lynx_f32x3u*>(p + 4);
  z[0] = a.x; z[1] = a.y; z[2] = a.z; z[3] = a.w; z[4] = b.x; z[5] = b.y; z[6] = b.z;
}
__device__ __forceinline__ void store_particle(float* p, const float (&z)[7]) {
  lynx_f32x4u a = {z[0], z[1], z[2], z[3]};
  lynx_f32x3u b = {z[4], z[5], z[6]};
  *reinterpret_cast<lynx_f32x4u*>(p) = a;
  *reinterpret_cast<lynx_f32x3u*>(p + 4) = b;
}
__device__ __forceinline__ void load_particle(const double* p, double (&z)[7]) {
  const lynx_f64x2u a = *reinterpret_cast<const lynx_f64x2u*>(p);
  const lynx_f64x2u b = *reinterpret_cast<const lynx_f64x2u*>(p + 2);
  const lynx_f64x2u c = *reinterpret_cast<const lynx_f64x2u*>(p + 4);
  z[0] = a.x; z[1] = a.y; z[2] = b.x; z[3] = b.y; z[4] = c.x; z[5] = c.y; z[6] = p[6];
}
__device__ __forceinline__ void store_particle(double* p, const double (&z)[7]) {
  lynx_f64x2u a = {z[0], z[1]}, b = {z[2], z[3]}, c = {z[4], z[5]};
  *reinterpret_cast<lynx_f64x2u*>(p) = a;
  *reinterpret_cast<lynx_f64x2u*>(p + 2) = b;
  *reinterpret_cast<lynx_f64x2u*>(p + 4) = c;
  p[6] = z[6];
}

// -- wave tiles (XPOSE) -------------------------------------------------------------------
constexpr int kWaveTileBytes = 7168;  // 64 lanes x 7 x 16 B = 64 * P particles of 7 scalars

__device__ __forceinline__ void wave_fence() { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); }

// seven full-width loads of the wave tile at `g` (dword-aligned)
__device__ __forceinline__ void wave_tile_fetch(const void* g, int lane, lynx_u32x4 (&v)[7]) {
  const lynx_u32x4u* src = reinterpret_cast<const lynx_u32x4u*>(g);
#pragma unroll
  for (int k = 0; k < 7; ++k) v[k] = src[k * 64 + lane];  // (non-temporal loads: -1.5 %)
}
__device__ __forceinline__ void wave_tile_store(void* g, int lane, const lynx_u32x4 (&v)[7]) {
  lynx_u32x4u* dst = reinterpret_cast<lynx_u32x4u*>(g);
#pragma unroll
  // non-temporal: the outgoing beam is not read again by this kernel, and a full-width store that does not
  // linger in L2 leaves the cache to the reads (BASELINE config 3 at 8 M particles: 5.02 -> 5.40 TB/s; float32
  // wave tiles 5.4 -> 5.8; per-particle stores, which fill a line in two instructions, gain nothing)
  for (int k = 0; k < 7; ++k) __builtin_nontemporal_store(v[k], dst + k * 64 + lane);
}
// flat tile (lane-interleaved 16-byte pieces) -> this lane's P consecutive particles, through the
// wave's private LDS region
template <typename T, int P>
__device__ __forceinline__ void wave_tile_to_particles(const lynx_u32x4 (&v)[7], unsigned char* lds, int lane,
                                                       T (&z)[P][7]) {
  static_assert(P * 7 * sizeof(T) == 112, "a lane owns 112 bytes of the tile");
  lynx_u32x4* l = reinterpret_cast<lynx_u32x4*>(lds);
  wave_fence();  // earlier reads of the region (previous tile's results) are done
#pragma unroll
  for (int k = 0; k < 7; ++k) l[k * 64 + lane] = v[k];
  wave_fence();
  lynx_u32x4 w[7];
#pragma unroll
  for (int k = 0; k < 7; ++k) w[k] = l[lane * 7 + k];
  constexpr int PER = 16 / sizeof(T);  // scalars per 16-byte piece
#pragma unroll
  for (int f = 0; f < P * 7; ++f) {
    const lynx_u32x4 q = w[f / PER];
    if constexpr (sizeof(T) == 4) {
      z[f / 7][f % 7] = __uint_as_float(q[f % PER]);
    } else {
      const unsigned long long bits = ((unsigned long long)q[(f % PER) * 2 + 1] << 32) | q[(f % PER) * 2];
      z[f / 7][f % 7] = __longlong_as_double((long long)bits);
    }
  }
}
template <typename T, int P>
__device__ __forceinline__ void wave_tile_from_particles(const T (&z)[P][7], unsigned char* lds, int lane,
                                                         lynx_u32x4 (&v)[7]) {
  lynx_u32x4* l = reinterpret_cast<lynx_u32x4*>(lds);
  constexpr int PER = 16 / sizeof(T);
  lynx_u32x4 w[7];
#pragma unroll
  for (int f = 0; f < P * 7; ++f) {
    if constexpr (sizeof(T) == 4) {
      w[f / PER][f % PER] = __float_as_uint(z[f / 7][f % 7]);
    } else {
      const unsigned long long bits = (unsigned long long)__double_as_longlong(z[f / 7][f % 7]);
      w[f / PER][(f % PER) * 2] = (unsigned int)(bits & 0xffffffffull);
      w[f / PER][(f % PER) * 2 + 1] = (unsigned int)(bits >> 32);
    }
  }
  wave_fence();
#pragma unroll
  for (int k = 0; k < 7; ++k) l[lane * 7 + k] = w[k];
  wave_fence();
#pragma unroll
  for (int k = 0; k < 7; ++k) v[k] = l[k * 64 + lane];
}

template <int MOM> struct MomScratch { using type = double; };
template <> struct MomScratch<2> { using type = float; };  // float32 partial sums
template <> struct MomScratch<3> { using type = float; };
constexpr int kMomSlabScalars = 4 * 29 * 65;  // per workgroup (whole covariance), in units of MomScratch<MOM>::type
constexpr int kMomSlabScalarsCompact = 4 * 16 * 65;

// Per-lane moment sums of the streaming kernel (see the MOM modes above).
template <typename T, int MOM, bool FULL>
struct LaneSums {
  using Set = MomentSet<FULL>;
  static constexpr int kRows = 8 + Set::N;  // slab rows: 6 d, one, the products, count
  MomentAcc<T, FULL> acc;                            // MOM 1, 2
  float f_d[6], f_dd[Set::N], f_one = 0.f, f_cnt = 0.f;  // MOM 2: per iteration, MOM 3: per workgroup
  __device__ __forceinline__ void clear_f() {
#pragma unroll
    for (int k = 0; k < 6; ++k) f_d[k] = 0.f;
#pragma unroll
    for (int k = 0; k < Set::N; ++k) f_dd[k] = 0.f;
    f_one = 0.f;
  }
  __device__ __forceinline__ void init() {
    if (MOM == 1 || MOM == 2) acc.clear();
    if (MOM == 3) clear_f();
  }
  __device__ __forceinline__ void begin_iteration() {
    if (MOM == 2) clear_f();
  }
  // spelled out by recursion: as nested loops with a running index the compiler once left `e` in
  // scratch memory
  template <int K> __device__ __forceinline__ void products(const float (&e)[6]) {
    if constexpr (K < Set::N) {
      // constant-evaluated here; as call operands the triangle walk ran per particle (whole covariance: 400
      // selects per iteration, the kernel at half its rate)
      constexpr int r = Set::row(K), c = Set::col(K);
      f_dd[K] = fmaf(e[r], e[c], f_dd[K]);
      products<K + 1>(e);
    }
  }
  __device__ __forceinline__ void add(const T (&z)[7], const T (&shift)[6]) {
    if (MOM == 1) acc.add(z, shift);
    if (MOM == 2 || MOM == 3) {
      float e[6];
#pragma unroll
      for (int k = 0; k < 6; ++k) {
        e[k] = (float)(z[k] - shift[k]);
        f_d[k] += e[k];
      }
      f_one += (float)z[6];
      products<0>(e);
      if (MOM == 2) acc.cnt += 1.0;
      else f_cnt += 1.f;
    }
  }
  __device__ __forceinline__ void end_iteration() {
    if (MOM == 2) {
#pragma unroll
      for (int k = 0; k < 6; ++k) acc.d[k] += (double)f_d[k];
#pragma unroll
      for (int k = 0; k < Set::N; ++k) acc.dd[k] += (double)f_dd[k];
      acc.one += (double)f_one;
    }
  }
  // rows of the reduction slab: 0..5 d, 6 one, 7.. the products, last the count
  template <typename R> __device__ __forceinline__ void park(R* slab, int lane) const {
    if (MOM == 3) {
#pragma unroll
      for (int i = 0; i < 6; ++i) slab[i * 65 + lane] = (R)f_d[i];
      slab[6 * 65 + lane] = (R)f_one;
#pragma unroll
      for (int i = 0; i < Set::N; ++i) slab[(7 + i) * 65 + lane] = (R)f_dd[i];
      slab[(7 + Set::N) * 65 + lane] = (R)f_cnt;
    } else {
#pragma unroll
      for (int i = 0; i < 6; ++i) slab[i * 65 + lane] = (R)acc.d[i];
      slab[6 * 65 + lane] = (R)acc.one;
#pragma unroll
      for (int i = 0; i < Set::N; ++i) slab[(7 + i) * 65 + lane] = (R)acc.dd[i];
      slab[(7 + Set::N) * 65 + lane] = (R)acc.cnt;
    }
  }
  // record slot of slab row r
  template <int K> __device__ static __forceinline__ int slot_scan(int r, int sl) {
    if constexpr (K < Set::N) {
      constexpr int slot = Set::slot(K);
      return slot_scan<K + 1>(r, r == 7 + K ? slot : sl);
    } else {
      return sl;
    }
  }
  __device__ static __forceinline__ int slot_of_row(int r) {
    if (r < 7) return r;
    if (r == 7 + Set::N) return 35;
    return slot_scan<0>(r, 0);
  }
};

// Workgroup reduction of the lane sums of a streaming kernel into ONE partial record (epilogue of k_track_direct and
// k_track_units; 256 threads, `s_scratch` >= the slab of kMomSlabScalars* scalars, reused).
template <typename T, int MOM, bool FULL>
__device__ __forceinline__ void workgroup_moment_record(const LaneSums<T, MOM, FULL>& sums, const T (&shift)[6],
                                                        unsigned char* s_scratch, double* __restrict__ out) {
  const int tid = threadIdx.x;
  const int wave = tid >> 6, lane = tid & 63;
  {
    // Workgroup reduction of the 29 sums through LDS: every lane parks its values in a
    // [29][65] slab of its wave (row pitch 65 -> conflict-free both ways), lanes 0..28 then
    // each add up one row in float64.  A 64-lane shuffle tree of 29 values costs ~350
    // ds_bpermute per wave; this costs 29 stores and 64 loads per lane and is the larger part
    // of a workgroup's fixed cost when it only owns a few thousand particles.
    using R = typename MomScratch<MOM>::type;
    using Sums = LaneSums<T, MOM, FULL>;
    constexpr int kRows = Sums::kRows;  // 29 with the whole covariance, 16 with the property set
    __syncthreads();  // the scratch (build scratch / wave tiles) is reused
    R* slab = reinterpret_cast<R*>(s_scratch) + wave * (kRows * 65);
    sums.park(slab, lane);
    __syncthreads();
    double tot = 0.0;
    if (lane < kRows) {
      const R* row = slab + lane * 65;
#pragma unroll 8
      for (int k = 0; k < 64; ++k) tot += (double)row[k];
    }
    __syncthreads();
    // one record per wave (layout of moment_slot_moved), then the four are moved to wave 0's
    // reference point and added in wave order
    double* s_red = reinterpret_cast<double*>(s_scratch);  // [4][36]
    if (lane < kPartialStride) s_red[wave * kPartialStride + lane] = 0.0;
    __syncthreads();
    if (lane < kRows) s_red[wave * kPartialStride + Sums::slot_of_row(lane)] = tot;
    // lane 0 publishes the reference point: it was assigned inside the loop, where lanes beyond the
    // end of the sample are no longer active and so never received it
    if (lane == 0) {
#pragma unroll
      for (int k = 0; k < 6; ++k) s_red[wave * kPartialStride + 28 + k] = (double)shift[k];
      s_red[wave * kPartialStride + 34] = FULL ? 1.0 : 0.0;  // which second moments the record carries
    }
    __syncthreads();
    if (tid < kPartialStride) {
      double v;
      if (tid >= 28 && tid < 35) {
        v = s_red[tid];
      } else {
        v = 0.0;
#pragma unroll
        for (int w = 0; w < kTrackThreads / 64; ++w) v += moment_slot_moved(s_red + w * kPartialStride, s_red + 28, tid);
      }
      out[tid] = v;
    }
  }
}

// the whole program on the UNROLL particles a lane holds in registers: steps outermost, so that each
// step's map is fetched once per iteration (fp32: 57 scalar loads into SGPRs; fp64: read from LDS) for
// all of them; fp32 with an even UNROLL runs the particles as packed pairs
template <typename T, int UNROLL, bool SCALAR_TABLE>
__device__ __forceinline__ void apply_program_lane(const LatticeDev& lat, int S, const T* g_steps, const T* s_steps,
                                                   T (&z)[UNROLL][7],
                                                   double* s_obs = nullptr /* this lane's [observer][x, y] sums, pitch 256 */,
                                                   const bool (*live)[UNROLL] = nullptr,
                                                   float* stash = nullptr /* this lane's 4 floats of LDS (kEntryStash) */) {
  constexpr bool kMapInRegs = sizeof(T) == 4;
  constexpr bool kPairs = kMapInRegs && UNROLL % 2 == 0;
  lynx_f32x2 zp[kPairs ? UNROLL / 2 : 1][7];
  if constexpr (kPairs) {
#pragma unroll
    for (int u = 0; u < UNROLL; u += 2)
#pragma unroll
      for (int c = 0; c < 7; ++c) {  // member-wise: the brace initialiser sent z to scratch memory here
        zp[u / 2][c].x = (float)z[u][c];
        zp[u / 2][c].y = (float)z[u + 1][c];
      }
  }
  int n_obs = 0;
  for (int sidx = 0; sidx < S; ++sidx) {
    // merged [run, cavity] pair (see k_build): the loop moves straight on to the cavity's slot, which holds
    // T_cav . T_run, the cavity's coefficients and the 2x2 that recovers the s and delta entering the cavity
    // from the product's own components (kEntryInverse) -- one scalar fetch and one 7x7 application per pair
    bool merged = false;
    // kind, flags and pairing of the step: one wave-uniform scalar from the table this kernel was given
    const T* table = SCALAR_TABLE ? g_steps : s_steps;
    int desc = (int)uniform_value(table[sidx * LYNX_STEP_STRIDE + LYNX_FLAGS_OFFSET]);
    if (s_obs && (desc & LYNX_STEP_FLAG_OBSERVE)) {  // uniform
      // an active BPM: x and y of the particles that enter it, added up per lane in float64 (LDS cells of
      // this lane only: no barrier); bpm.py:48-54 reads mu_x, mu_y of the incoming beam
#pragma unroll
      for (int u = 0; u < UNROLL; ++u) {
        double x, y;
        if constexpr (kPairs) {
          x = (u & 1) ? (double)zp[u / 2][0].y : (double)zp[u / 2][0].x;
          y = (u & 1) ? (double)zp[u / 2][2].y : (double)zp[u / 2][2].x;
        } else {
          x = (double)z[u][0];
          y = (double)z[u][2];
        }
        if (!live || (*live)[u]) {
          s_obs[(2 * n_obs + 0) * 256] += x;
          s_obs[(2 * n_obs + 1) * 256] += y;
        }
      }
      ++n_obs;
    }
    // (two particles per lane only: the plan's choice for multi-step programs, and the only form the host asks merged
    // tables for -- the four-per-lane kernels of single-map programs, BASELINE config 4's among them, stay free of it)
    if constexpr (kPairs && SCALAR_TABLE && UNROLL == 2) {
      if (desc & LYNX_DESC_PAIR) {  // uniform: this run is applied together with the cavity behind it
        merged = true;
        ++sidx;
        desc = (int)uniform_value(table[sidx * LYNX_STEP_STRIDE + LYNX_FLAGS_OFFSET]);
      }
    }
    const int skind = (desc >> LYNX_DESC_KIND_SHIFT) & 3, sflags = desc & 0xffff;
    if constexpr (kMapInRegs) {
      T m[LYNX_STEP_SCALARS];  // T = float here: map, coefficients, entry inverse, sin(phi)
      if (SCALAR_TABLE) {
        const T* tab = g_steps + sidx * LYNX_STEP_STRIDE;  // global: s_load
#pragma unroll
        for (int q = 0; q < LYNX_STEP_SCALARS; ++q) m[q] = uniform_value(tab[q]);
      } else {
        const T* tab = s_steps + sidx * LYNX_STEP_STRIDE;  // LDS
#pragma unroll
        for (int q = 0; q < LYNX_STEP_SCALARS; ++q) m[q] = uniform_value(tab[q]);
      }
      if constexpr (kPairs) {
        // an ill-conditioned pair (uniform and rare: cavity_entry_inverse): what enters the cavity comes from the run's
        // rows 4 and 5, worked out here and parked in LDS while the 7x7 is applied
        const bool rows_form = __builtin_expect(merged && (desc & LYNX_DESC_ILL), 0);
        const int entry = !merged ? kEntryOwn : (rows_form ? kEntryStash : kEntryInverse);
#pragma unroll
        for (int h = 0; h < UNROLL / 2; ++h) {
          if (rows_form) {
            lynx_f32x2 s_in = zp[h][0] * uniform_value(g_steps[(sidx - 1) * LYNX_STEP_STRIDE]);
            lynx_f32x2 d_in = zp[h][0] * uniform_value(g_steps[(sidx - 1) * LYNX_STEP_STRIDE + 7]);
#pragma unroll
            for (int j = 1; j < 7; ++j) {
              s_in = pk_fma(zp[h][j], uniform_value(g_steps[(sidx - 1) * LYNX_STEP_STRIDE + j]), s_in);
              d_in = pk_fma(zp[h][j], uniform_value(g_steps[(sidx - 1) * LYNX_STEP_STRIDE + 7 + j]), d_in);
            }
            *reinterpret_cast<lynx_f32x4*>(stash) = lynx_f32x4{s_in.x, s_in.y, d_in.x, d_in.y};
          }
          apply_step_pair(m, skind, sflags, zp[h], entry, lynx_f32x2{0.f, 0.f}, lynx_f32x2{0.f, 0.f}, stash);
        }
      } else {
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) apply_step<T>(m, skind, sflags, z[u]);
      }
    } else {
      const T* tab = s_steps + sidx * LYNX_STEP_STRIDE;
#pragma unroll
      for (int u = 0; u < UNROLL; ++u) apply_step<T>(tab, skind, sflags, z[u]);
    }
  }
  if constexpr (kPairs) {
#pragma unroll
    for (int u = 0; u < UNROLL; u += 2)
#pragma unroll
      for (int c = 0; c < 7; ++c) {
        z[u][c] = (T)zp[u / 2][c].x;
        z[u + 1][c] = (T)zp[u / 2][c].y;
      }
  }
}

// Occupancy asked of the compiler: wave tiles 3 waves/SIMD (their sweet spot: 4 and 5 ran 10 % slower, DESIGN.md
// section 7); the packed-pair form of multi-step float32 programs 5 (96 VGPRs, 32 bytes of scratch outside the step
// loop): those wait on scalar fetches of the step table, and a fifth wave hides more of that (BASELINE config 5:
// 0.985 -> 0.961 ms; 6 waves spill into the loop: 1.07).
template <typename T, int MOM, bool FULL, int UNROLL, bool FUSED, bool XPOSE> constexpr int track_waves_per_simd() {
  if (XPOSE) return 3;
  if (sizeof(T) == 4 && UNROLL == 2 && !FUSED && !FULL && (MOM == 0 || MOM == 3)) return 5;
  return 1;
}

template <typename T, int MOM, bool FULL, int UNROLL, bool FUSED, bool XPOSE>
__global__ __launch_bounds__(kTrackThreads, (track_waves_per_simd<T, MOM, FULL, UNROLL, FUSED, XPOSE>())) void k_track_direct(
    LatticeDev lat, TrackArgs a, const T* __restrict__ energy_in, const T* p_in, T* p_out,
    T* __restrict__ energy_out, const T* __restrict__ steps_in, double* __restrict__ partials,
    double* __restrict__ obs_partials) {
  static_assert(!XPOSE || (UNROLL * 7 * sizeof(T) == 112 && !FUSED), "XPOSE: a lane owns 112 bytes");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  announce_tail(a);
  unsigned char* s_scratch = smem_raw;                                 // build scratch / wave tiles / reduction
  T* s_steps = reinterpret_cast<T*>(smem_raw + a.lds_scratch_bytes);    // [S][64]
  float* stash = reinterpret_cast<float*>(smem_raw + a.stash_offset) + threadIdx.x * 4;  // used with merged tables only
  T* s_energy = s_steps + (size_t)lat.n_steps * LYNX_STEP_STRIDE;       // [S+1]
  // observers (active BPMs): every lane's float64 sums of x and y at each of them, [2 * n_observers][256]
  double* s_obs = reinterpret_cast<double*>(smem_raw + ((a.lds_scratch_bytes + ((size_t)lat.n_steps * (LYNX_STEP_STRIDE + 1) + 1) * sizeof(T) + 7) / 8 * 8)) + threadIdx.x;

  const int tid = threadIdx.x;
  const int wave = tid >> 6, lane = tid & 63;
  const unsigned wg = a.reversed ? gridDim.x - 1u - blockIdx.x : blockIdx.x;
  const int64_t b = wg / a.chunks;
  const int chunk = wg % a.chunks;
  const int64_t end = a.n_particles;
  const int S = lat.n_steps;
  // This workgroup's `a.tiles_per_wg` tiles of 256*UNROLL particles: one contiguous stretch (taking every
  // `chunks`-th tile instead, so that a sample's workgroups advance side by side, measured 2-5 % slower).
  constexpr int64_t kTile = (int64_t)kTrackThreads * UNROLL;
  constexpr int64_t kWaveSpan = 64 * UNROLL;  // XPOSE: particles of a wave tile
  // first particle of this lane within a tile, and the distance between its UNROLL particles
  const int64_t lane_first = XPOSE ? (int64_t)wave * kWaveSpan + (int64_t)lane * UNROLL : (int64_t)tid;
  constexpr int64_t kLaneStep = XPOSE ? 1 : kTrackThreads;
  unsigned char* s_wave = s_scratch + wave * kWaveTileBytes;  // XPOSE: this wave's private tile
#define LYNX_TILE_OF(IT) ((int64_t)chunk * a.tiles_per_wg + (IT))
#define LYNX_WAVE_BASE(TILE) ((TILE)*kTile + (int64_t)wave * kWaveSpan)

  // Software pipeline: the loads of a tile are issued one iteration ahead (the first ones
  // right here, before the step table is fetched), so a workgroup always has a tile of HBM
  // reads in flight while it computes and stores the previous one.
  const T* src = p_in + b * a.in_stride;
  T* dst = p_out + b * end * 7;
  T zn[XPOSE ? 1 : UNROLL][7];   // !XPOSE: the next tile's particles
  lynx_u32x4 vn[XPOSE ? 7 : 1];  //  XPOSE: the next wave tile, flat
  {
    const int64_t tile = LYNX_TILE_OF(0);
    if constexpr (XPOSE) {
      if (LYNX_WAVE_BASE(tile) + kWaveSpan <= end) wave_tile_fetch(src + LYNX_WAVE_BASE(tile) * 7, lane, vn);
    } else {
      const int64_t i0 = tile * kTile + lane_first;
#pragma unroll
      for (int u = 0; u < UNROLL; ++u) {
        const int64_t i = i0 + (int64_t)u * kLaneStep;
        load_particle(src + (i < end ? i : (i0 < end ? i0 : 0)) * 7, zn[u]);
      }
    }
  }

  const bool one_run = (S == 1) && (lat.steps[0].kind == LYNX_STEP_RUN) && a.n_observers == 0;
  for (int k = 0; k < 2 * a.n_observers; ++k) s_obs[k * 256] = 0.0;
  constexpr bool kMapInRegs = sizeof(T) == 4;
  // Pre-built single-run fp32 program: the map comes straight from the step table with
  // wave-uniform (scalar) loads -- no LDS staging, no barrier.
  constexpr bool kScalarTable = !FUSED && kMapInRegs;  // address space known at compile time
  const bool scalar_table = kScalarTable && S > 0;
  const T* g_steps = steps_in + b * (int64_t)S * LYNX_STEP_STRIDE;

  if (FUSED) {
    build_compose_sample<T>(lat, b, energy_in[b], s_steps, s_energy, s_scratch, a.build_chunk);
    table_sinphi<T>(lat, s_steps);
    for (int s = tid; s < S; s += kTrackThreads) s_steps[s * LYNX_STEP_STRIDE + LYNX_FLAGS_OFFSET] = (T)step_descriptor(lat, s, 0);
    __syncthreads();
    if (energy_out && chunk == 0 && tid == 0) energy_out[b] = s_energy[S];
  } else if (S > 0) {
    if (!scalar_table) load_steps_sample<T>(g_steps, S, s_steps);
    // the beam energy behind the last step was parked in the table by k_build
    if (energy_out && chunk == 0 && tid == 0) energy_out[b] = g_steps[(S - 1) * LYNX_STEP_STRIDE + LYNX_ENERGY_OFFSET];
  }

  T m0[kMapInRegs ? 49 : 1];
  if (one_run && kMapInRegs) {
    if (scalar_table) {
#pragma unroll
      for (int i = 0; i < 49; ++i) m0[i] = uniform_value(g_steps[i]);
    } else {
#pragma unroll
      for (int i = 0; i < 49; ++i) m0[i] = uniform_value(s_steps[i]);
    }
  }

  LaneSums<T, MOM, FULL> sums;
  sums.init();
  T shift[6];  // this wave's reference point: its first tracked particle (wave-uniform)
#pragma unroll
  for (int i = 0; i < 6; ++i) shift[i] = T(0);

  // float64 wave tiles stay in LDS and the lane's two particles are taken one after the other (two
  // float64 particles + the prefetched tile + 29 float64 sums do not fit the registers of 3 waves/SIMD)
  constexpr bool kLdsResident = XPOSE && sizeof(T) == 8;

  for (int it = 0; it < a.tiles_per_wg; ++it) {
    const int64_t tile = LYNX_TILE_OF(it);
    const int64_t i0 = tile * kTile + lane_first;
    if (XPOSE ? (LYNX_WAVE_BASE(tile) >= end) : (i0 >= end)) break;
    if (!kMapInRegs) LYNX_FORGET();  // fp64: keep the map in LDS, not in 98 hoisted VGPRs
    const bool full = XPOSE && (LYNX_WAVE_BASE(tile) + kWaveSpan <= end);  // wave-uniform
    sums.begin_iteration();

    if constexpr (kLdsResident) {
      T* mine = reinterpret_cast<T*>(s_wave) + lane * (UNROLL * 7);  // this lane's particles inside the tile
      if (full) {
        lynx_u32x4* l = reinterpret_cast<lynx_u32x4*>(s_wave);
        wave_fence();
#pragma unroll
        for (int k = 0; k < 7; ++k) l[k * 64 + lane] = vn[k];
        wave_fence();
      }
      if (it + 1 < a.tiles_per_wg) {
        const int64_t tn = LYNX_TILE_OF(it + 1);
        if (LYNX_WAVE_BASE(tn) + kWaveSpan <= end) wave_tile_fetch(src + LYNX_WAVE_BASE(tn) * 7, lane, vn);
      }
#pragma unroll 1
      for (int u = 0; u < UNROLL; ++u) {  // one particle at a time: register pressure, not issue rate, is what binds here
        const int64_t i = i0 + u;
        T z[1][7];
        if (full) {
#pragma unroll
          for (int c = 0; c < 7; ++c) z[0][c] = mine[u * 7 + c];
        } else {
          load_particle(src + (i < end ? i : end - 1) * 7, z[0]);
        }
        if (one_run) {
          apply_step<T>(s_steps, LYNX_STEP_RUN, 0, z[0]);
        } else {
          const bool alive[1] = {i < end};
          apply_program_lane<T, 1, false>(lat, S, g_steps, s_steps, z, a.n_observers ? s_obs : nullptr, &alive);
        }
        if (MOM && it == 0 && u == 0) {
#pragma unroll
          for (int k = 0; k < 6; ++k) shift[k] = wave_first(z[0][k]);
        }
        if (i < end) {
          if (MOM) sums.add(z[0], shift);
          if (a.store) {
            if (full) {
#pragma unroll
              for (int c = 0; c < 7; ++c) mine[u * 7 + c] = z[0][c];
            } else {
              store_particle(dst + i * 7, z[0]);
            }
          }
        }
      }
      if (a.store && full) {
        lynx_u32x4 vo[7];
        const lynx_u32x4* l = reinterpret_cast<const lynx_u32x4*>(s_wave);
        wave_fence();
#pragma unroll
        for (int k = 0; k < 7; ++k) vo[k] = l[k * 64 + lane];
        wave_tile_store(dst + LYNX_WAVE_BASE(tile) * 7, lane, vo);
      }
    } else {
      T z[UNROLL][7];
      if constexpr (XPOSE) {
        if (full) {
          wave_tile_to_particles<T, UNROLL>(vn, s_wave, lane, z);
        } else {
#pragma unroll
          for (int u = 0; u < UNROLL; ++u) load_particle(src + (i0 + u < end ? i0 + u : end - 1) * 7, z[u]);
        }
        if (it + 1 < a.tiles_per_wg) {
          const int64_t tn = LYNX_TILE_OF(it + 1);
          if (LYNX_WAVE_BASE(tn) + kWaveSpan <= end) wave_tile_fetch(src + LYNX_WAVE_BASE(tn) * 7, lane, vn);
        }
      } else {
#pragma unroll
        for (int u = 0; u < UNROLL; ++u)
#pragma unroll
          for (int c = 0; c < 7; ++c) z[u][c] = zn[u][c];
        if (it + 1 < a.tiles_per_wg) {  // prefetch the next tile of this workgroup
          const int64_t j0 = LYNX_TILE_OF(it + 1) * kTile + lane_first;
#pragma unroll
          for (int u = 0; u < UNROLL; ++u) {
            const int64_t j = j0 + (int64_t)u * kLaneStep;
            load_particle(src + (j < end ? j : i0) * 7, zn[u]);
          }
        }
      }
      if (!one_run) {
        bool alive[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) alive[u] = i0 + (int64_t)u * kLaneStep < end;
        apply_program_lane<T, UNROLL, kScalarTable>(lat, S, g_steps, s_steps, z,
                                                    a.n_observers ? s_obs : nullptr, &alive, stash);
      }
#pragma unroll
      for (int u = 0; u < UNROLL; ++u) {
        const int64_t i = i0 + (int64_t)u * kLaneStep;
        if (one_run) {
          if (kMapInRegs) apply_step<T>(m0, LYNX_STEP_RUN, 0, z[u]);
          else apply_step<T>(s_steps, LYNX_STEP_RUN, 0, z[u]);
        }
        if (MOM && it == 0 && u == 0) {
          // reference point of this wave's sums: the tracked particle of its first lane (which exists
          // whenever any of the wave's particles does: indices grow with the lane)
#pragma unroll
          for (int k = 0; k < 6; ++k) shift[k] = wave_first(z[0][k]);
        }
        if (i < end) {
          if (a.store && !(XPOSE && full)) store_particle(dst + i * 7, z[u]);
          if (MOM) sums.add(z[u], shift);
        }
      }
      if constexpr (XPOSE) {
        if (a.store && full) {
          lynx_u32x4 vo[7];
          wave_tile_from_particles<T, UNROLL>(z, s_wave, lane, vo);
          wave_tile_store(dst + LYNX_WAVE_BASE(tile) * 7, lane, vo);
        }
      }
    }
    sums.end_iteration();
  }
#undef LYNX_TILE_OF
#undef LYNX_WAVE_BASE

  if (a.n_observers) {
    // this workgroup's sums of x and y at every observer: lanes in lane order, float64 -- deterministic
    __syncthreads();
    if (tid < 2 * a.n_observers) {
      const double* row = s_obs - threadIdx.x + tid * 256;
      double tot = 0.0;
      for (int k = 0; k < 256; ++k) tot += row[k];
      obs_partials[((int64_t)b * a.chunks + chunk) * (2 * LYNX_MAX_OBSERVERS) + tid] = tot;
    }
  }

  if (MOM) workgroup_moment_record<T, MOM, FULL>(sums, shift, s_scratch, partials + ((int64_t)b * a.chunks + chunk) * kPartialStride);
}

// ---------------------------------------------------------------------------------------
// k_reduce_moments: adds up partial records in a fixed order -- no float atomics.
//   grid = B x groups, 256 threads.  Workgroup (b, g) owns the rows [g * rows_per_group, ...) of
//   sample b's `rows` input records; it stages up to 70 rows at a time in LDS with coalesced loads,
//   thread (q, j) (7 sets of 36 threads) moves value j of staged rows q, q + 7, ... to the reference
//   point of the group's first row (moment_slot_moved) and adds them in row order, the 7 set sums are
//   then added in set order.
//   FINAL = false: writes one partial record per group (same layout) -- the level of a reduction tree
//                  when a sample has more rows than one workgroup should walk (a B = 1 beam of 8 M
//                  particles has ~8000 of them);
//   FINAL = true : writes the moment record of the sample (include/lynx_hip.h); groups must be 1.
// ---------------------------------------------------------------------------------------
constexpr int kReduceStage = 70;        // rows staged per pass by the 256-thread shape: 10 per set
constexpr int kReduceStageWide = 448;   // ... by the 1024-thread shape: 16 per set
template <int THREADS, int STAGE> constexpr size_t reduce_lds_bytes() {
  return (size_t)(STAGE + THREADS / kPartialStride + 1) * kPartialStride * sizeof(double);
}

__device__ __forceinline__ void write_moment_record(const double* s, double* dst, int tid) {
  const double n = s[35];
  if (tid < 6) {
    dst[tid] = s[28 + tid] + s[tid] / n;
  } else if (tid == 6) {
    dst[6] = s[6] / n;
  } else if (tid < 28) {
    int k = tid - 7, i = 0, row = 6;  // upper-triangle index -> (i, j)
    while (k >= row) { k -= row; --row; ++i; }
    const int j = i + k;
    // a property-set record (s[34] == 0) carries the variances, cov(x, x') and cov(y, y') only
    const bool have = s[34] != 0.0 || i == j || (i == 0 && j == 1) || (i == 2 && j == 3);
    dst[tid] = have ? (s[tid] - s[i] * s[j] / n) / n : __longlong_as_double(0x7ff8000000000000ll);
  } else if (tid < 34) {
    dst[tid] = 0.0;
  } else if (tid == 34) {
    dst[34] = s[34];
  } else if (tid == 35) {
    dst[35] = n;
  }
}

// THREADS / 36 sets of 36 threads; STAGE rows staged per pass (STAGE / sets per set).  Two shapes are used:
//   <FINAL, 256, 70>    7 sets, 22 KB of LDS: many samples, or the groups of a level;
//   <true, 1024, 448>   28 sets, 137 KB: ONE workgroup per sample walks a few hundred rows in one pass -- beams of
//                       few samples (BASELINE configs 2 and 3: 391 and 977 rows), where a level in between costs a
//                       launch and a kernel boundary (~10 us) and there are no other samples to fill the CUs with.
template <bool FINAL, int THREADS, int STAGE>
__global__ __launch_bounds__(THREADS) void k_reduce_moments(const double* __restrict__ in, int rows, int rows_per_group,
                                                             int groups, double* __restrict__ out) {
  constexpr int kSets = THREADS / kPartialStride;
  static_assert(STAGE % kSets == 0, "every set takes the same number of staged rows");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  double* s_rows = reinterpret_cast<double*>(smem_raw);        // [STAGE][36]
  double* s_set = s_rows + STAGE * kPartialStride;              // [kSets][36]
  double* s = s_set + kSets * kPartialStride;                   // [36]
  const int64_t b = blockIdx.x / groups;
  const int g = blockIdx.x % groups;
  const int tid = threadIdx.x;
  const int q = tid / kPartialStride, j = tid - q * kPartialStride;
  const int lo = g * rows_per_group;
  const int hi = (lo + rows_per_group) < rows ? (lo + rows_per_group) : rows;
  const double* src = in + (b * (int64_t)rows + lo) * kPartialStride;
  double v = 0.0;
  int mi = 0, mj = 0;  // j in [7, 28): the pair (mi <= mj) of components of second moment j (row-major upper triangle)
  if (j >= 7 && j < 28) {
    int k = j - 7, len = 6;
    while (k >= len) { k -= len; --len; ++mi; }
    mj = mi + k;
  }
  for (int base = 0; base < hi - lo; base += STAGE) {
    const int n_stage = (hi - lo - base) < STAGE ? (hi - lo - base) : STAGE;
    __syncthreads();  // the previous pass is done with the staging area
    {
      // all of a thread's loads are issued before the first one is waited for
      constexpr int kPerThread = (STAGE * kPartialStride + THREADS - 1) / THREADS;
      double staged[kPerThread];
#pragma unroll
      for (int k = 0; k < kPerThread; ++k) {
        const int i = tid + k * THREADS;
        staged[k] = i < n_stage * kPartialStride ? src[(int64_t)base * kPartialStride + i] : 0.0;
      }
#pragma unroll
      for (int k = 0; k < kPerThread; ++k) {
        const int i = tid + k * THREADS;
        if (i < n_stage * kPartialStride) s_rows[i] = staged[k];
      }
    }
    __syncthreads();
    if (base == 0 && tid < 8) s[28 + tid] = tid < 7 ? s_rows[28 + tid] : 0.0;  // the group's reference point (s[35] set below)
    __syncthreads();
    if (q < kSets && (j < 28 || j == 35)) {
      // moment_slot_moved with everything that depends on j alone taken out of the row loop (which pair of
      // components a second moment belongs to, the reference point's two components): what is left per row
      // is six LDS reads and five multiply-adds, rows independent of each other
      if (j == 6 || j == 35) {
#pragma unroll 4
        for (int r = q; r < n_stage; r += kSets) v += s_rows[r * kPartialStride + j];
      } else if (j < 6) {
        const double c0j = s[28 + j];
#pragma unroll 4
        for (int r = q; r < n_stage; r += kSets) {
          const double* row = s_rows + r * kPartialStride;
          v += row[j] + row[35] * (row[28 + j] - c0j);
        }
      } else {
        const double c0i = s[28 + mi], c0j = s[28 + mj];
#pragma unroll 4
        for (int r = q; r < n_stage; r += kSets) {
          const double* row = s_rows + r * kPartialStride;
          const double ei = row[28 + mi] - c0i, ej = row[28 + mj] - c0j;
          v += row[j] + ei * row[mj] + row[mi] * ej + row[35] * ei * ej;
        }
      }
    }
  }
  if (q < kSets) s_set[q * kPartialStride + j] = v;
  __syncthreads();
  if (tid < kPartialStride && (tid < 28 || tid == 35)) {
    double t = 0.0;
#pragma unroll
    for (int k = 0; k < kSets; ++k) t += s_set[k * kPartialStride + tid];
    s[tid] = t;
  }
  __syncthreads();
  if (FINAL) {
    write_moment_record(s, out + b * LYNX_MOMENT_STRIDE, tid);
  } else if (tid < kPartialStride) {
    out[(b * (int64_t)groups + g) * kPartialStride + tid] = s[tid];
  }
}

// k_reduce_moments_ticket: both levels of the tree in ONE launch, for beams whose samples have more records than one
// workgroup should walk (a B = 1 beam of 1 M particles has 1954, of 8 M ~8000).  grid = B x groups, groups <= 64.
// Every workgroup adds its group's rows up exactly like the level form above, publishes the group's record and takes a
// ticket of its sample; the workgroup that draws the last ticket adds the group records in group order (so the result
// does not depend on who arrives last) and writes the sample's moment record.  A launch less per call: kernels of
// one queue follow each other without a gap, but every one of these small ones takes 5-6 us from dispatch to retirement
// (BASELINE config 3: 5.8 + 5.8 us -> one launch).
// Coherence: the eight XCDs' L2 caches are not coherent with each other.  Group records are written and read with
// agent-scope atomic accesses (they bypass the L2 of the writer and of the reader), the ticket is an agent-scope
// acquire-release read-modify-write; this kernel runs behind the streaming kernel's end-of-kernel write-back, so
// there is no dirty streaming data for the release to push out (the same protocol INSIDE the streaming kernel cost more
// than the launch it saved: round 2).
template <int THREADS, int STAGE>
__global__ __launch_bounds__(THREADS) void k_reduce_moments_ticket(const double* __restrict__ in, int rows, int rows_per_group,
                                                                    int groups, double* level /* [B][groups][36] */,
                                                                    unsigned int* tickets /* [B], zero between launches */,
                                                                    double* __restrict__ out) {
  constexpr int kSets = THREADS / kPartialStride;
  static_assert(STAGE % kSets == 0 && STAGE >= 64, "every set takes the same number of staged rows; the second level fits one pass");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  double* s_rows = reinterpret_cast<double*>(smem_raw);        // [STAGE][36]
  double* s_set = s_rows + STAGE * kPartialStride;              // [kSets][36]
  double* s = s_set + kSets * kPartialStride;                   // [36]
  __shared__ int s_last;
  const int64_t b = blockIdx.x / groups;
  const int g = blockIdx.x % groups;
  const int tid = threadIdx.x;
  const int q = tid / kPartialStride, j = tid - q * kPartialStride;
  int mi = 0, mj = 0;
  if (j >= 7 && j < 28) {
    int k = j - 7, len = 6;
    while (k >= len) { k -= len; --len; ++mi; }
    mj = mi + k;
  }
  // rows [0, n) staged in s_rows, reference point = the first one's: this thread's share of the sums (moment_slot_moved)
  const auto add_staged = [&](int n, double v) {
    if (q < kSets && (j < 28 || j == 35)) {
      if (j == 6 || j == 35) {
#pragma unroll 4
        for (int r = q; r < n; r += kSets) v += s_rows[r * kPartialStride + j];
      } else if (j < 6) {
        const double c0j = s[28 + j];
#pragma unroll 4
        for (int r = q; r < n; r += kSets) {
          const double* row = s_rows + r * kPartialStride;
          v += row[j] + row[35] * (row[28 + j] - c0j);
        }
      } else {
        const double c0i = s[28 + mi], c0j = s[28 + mj];
#pragma unroll 4
        for (int r = q; r < n; r += kSets) {
          const double* row = s_rows + r * kPartialStride;
          const double ei = row[28 + mi] - c0i, ej = row[28 + mj] - c0j;
          v += row[j] + ei * row[mj] + row[mi] * ej + row[35] * ei * ej;
        }
      }
    }
    return v;
  };
  const auto fold_sets = [&](double v) {
    if (q < kSets) s_set[q * kPartialStride + j] = v;
    __syncthreads();
    if (tid < kPartialStride && (tid < 28 || tid == 35)) {
      double t = 0.0;
#pragma unroll
      for (int k = 0; k < kSets; ++k) t += s_set[k * kPartialStride + tid];
      s[tid] = t;
    }
    __syncthreads();
  };
  // level one: this group's rows
  const int lo = g * rows_per_group;
  const int hi = (lo + rows_per_group) < rows ? (lo + rows_per_group) : rows;
  const double* src = in + (b * (int64_t)rows + lo) * kPartialStride;
  double v = 0.0;
  for (int base = 0; base < hi - lo; base += STAGE) {
    const int n_stage = (hi - lo - base) < STAGE ? (hi - lo - base) : STAGE;
    __syncthreads();
    constexpr int kPerThread = (STAGE * kPartialStride + THREADS - 1) / THREADS;
    double staged[kPerThread];
#pragma unroll
    for (int k = 0; k < kPerThread; ++k) {
      const int i = tid + k * THREADS;
      staged[k] = i < n_stage * kPartialStride ? src[(int64_t)base * kPartialStride + i] : 0.0;
    }
#pragma unroll
    for (int k = 0; k < kPerThread; ++k) {
      const int i = tid + k * THREADS;
      if (i < n_stage * kPartialStride) s_rows[i] = staged[k];
    }
    __syncthreads();
    if (base == 0 && tid < 8) s[28 + tid] = tid < 7 ? s_rows[28 + tid] : 0.0;
    __syncthreads();
    v = add_staged(n_stage, v);
  }
  fold_sets(v);
  // publish, take a ticket
  double* mine = level + (b * (int64_t)groups + g) * kPartialStride;
  if (tid < kPartialStride) __hip_atomic_store(mine + tid, s[tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __syncthreads();  // (s_waitcnt in front of the barrier: the stores have been issued by every thread)
  if (tid == 0) {
    const unsigned int ticket = __hip_atomic_fetch_add(tickets + b, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
    s_last = ticket == (unsigned int)(groups - 1);
  }
  __syncthreads();
  if (!s_last) return;  // uniform
  // level two: the group records, in group order
  const double* lv = level + b * (int64_t)groups * kPartialStride;
  for (int i = tid; i < groups * kPartialStride; i += THREADS)
    s_rows[i] = __hip_atomic_load(lv + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __syncthreads();
  if (tid < 8) s[28 + tid] = tid < 7 ? s_rows[28 + tid] : 0.0;
  __syncthreads();
  fold_sets(add_staged(groups, 0.0));
  write_moment_record(s, out + b * LYNX_MOMENT_STRIDE, tid);
  if (tid == 0) __hip_atomic_store(tickets + b, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // ready for the next launch
}

// k_reduce_observers: per-workgroup sums of x and y at the observers -> BPM readings [B][n_observers][2]
// (mean x, mean y of the beam entering each active BPM), chunks added in chunk order.
__global__ __launch_bounds__(64) void k_reduce_observers(const double* __restrict__ obs_partials, int chunks,
                                                         int n_observers, double n_particles, double* __restrict__ out) {
  const int64_t b = blockIdx.x;
  const int t = threadIdx.x;
  if (t >= 2 * n_observers) return;
  const double* src = obs_partials + b * (int64_t)chunks * (2 * LYNX_MAX_OBSERVERS) + t;
  double tot = 0.0;
  for (int c = 0; c < chunks; ++c) tot += src[(int64_t)c * (2 * LYNX_MAX_OBSERVERS)];
  out[b * 2 * n_observers + t] = tot / n_particles;
}

// ---------------------------------------------------------------------------------------
// k_track_moments: ParameterBeam path, one workgroup per sample (element.py:71-82,
// cavity.py:134-140, 202-218): one wave when the batch is large (hundreds of thousands of
// settings), four when it is small, where the build of a long lattice is the whole cost and
// parallelises over elements.  The moment propagation itself uses 49 lanes.
// ---------------------------------------------------------------------------------------
template <typename T>
__device__ __forceinline__ void track_moments_sample(const LatticeDev& lat, const T* __restrict__ energy_in,
                                                     const T* mu_in, const T* cov_in, T* mu_out,
                                                     T* cov_out, T* __restrict__ energy_out, int chunk) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  T* s_steps = reinterpret_cast<T*>(smem_raw + build_scratch_bytes(chunk, sizeof(T)));
  T* s_energy = s_steps + (size_t)lat.n_steps * LYNX_STEP_STRIDE;
  T* s_mu = s_energy + lat.n_steps + 1;  // 8
  T* s_cov = s_mu + 8;                   // 49
  T* s_x = s_cov + 49;                   // 49
  T* s_in = s_x + 49;                    // mu_in(7) + cov_in 44,45,55 for the cavity branch

  const int64_t b = blockIdx.x;
  const int lane = threadIdx.x;
  build_compose_sample<T>(lat, b, energy_in[b], s_steps, s_energy, smem_raw, chunk);
  table_sinphi<T>(lat, s_steps);

  if (lane < 7) s_mu[lane] = mu_in[b * 7 + lane];
  if (lane < 49) s_cov[lane] = cov_in[b * 49 + lane];
  __syncthreads();
  const int cl = lane < 49 ? lane : 48;
  const int i = cl / 7, j = cl % 7;

  for (int s = 0; s < lat.n_steps; ++s) {
    lynx_step st = lat.steps[s];
    const T* M = s_steps + s * LYNX_STEP_STRIDE;
    // keep what the cavity branch needs from the incoming beam
    if (lane < 7) s_in[lane] = s_mu[lane];
    if (lane == 7) s_in[7] = s_cov[4 * 7 + 4];
    if (lane == 8) s_in[8] = s_cov[4 * 7 + 5];
    if (lane == 9) s_in[9] = s_cov[5 * 7 + 5];
    // mu' = T mu
    T mu_new = T(0);
    if (lane < 7) {
      mu_new = M[lane * 7 + 0] * s_mu[0];
#pragma unroll
      for (int k = 1; k < 7; ++k) mu_new = t_fma(M[lane * 7 + k], s_mu[k], mu_new);
    }
    // X = cov . T^T
    T x = s_cov[i * 7 + 0] * M[j * 7 + 0];
#pragma unroll
    for (int k = 1; k < 7; ++k) x = t_fma(s_cov[i * 7 + k], M[j * 7 + k], x);
    __syncthreads();
    if (lane < 49) s_x[lane] = x;
    if (lane < 7) s_mu[lane] = mu_new;
    __syncthreads();
    // cov' = T . X
    T c = M[i * 7 + 0] * s_x[0 * 7 + j];
#pragma unroll
    for (int k = 1; k < 7; ++k) c = t_fma(M[i * 7 + k], s_x[k * 7 + j], c);
    __syncthreads();
    if (lane < 49) s_cov[lane] = c;
    __syncthreads();
    if (st.kind == LYNX_STEP_CAVITY && (st.flags & LYNX_FLAG_CAV_GAIN) && lane == 0) {
      const T* coef = M + LYNX_COEF_OFFSET;
      T s_o = s_mu[4], d_o;
      device_cavity_kick<T>(coef, s_in[4], s_in[5], s_o, d_o);  // cavity.py:134-140, 202-206
      s_mu[4] = s_o;
      s_mu[5] = d_o;
      const T c44 = s_in[7], c45 = s_in[8], c55 = s_in[9];
      s_cov[5 * 7 + 5] = c55;  // cavity.py:140
      const T v = coef[LYNX_C_T566] * (c55 * c55) + coef[LYNX_C_T556] * c45 * c55 +
                  coef[LYNX_C_T555] * (c44 * c44);  // cavity.py:207-218
      s_cov[4 * 7 + 4] = v;
      s_cov[4 * 7 + 5] = v;
      s_cov[5 * 7 + 4] = v;
    }
    __syncthreads();
  }
  if (lane < 7) mu_out[b * 7 + lane] = s_mu[lane];
  if (lane < 49) cov_out[b * 49 + lane] = s_cov[lane];
  if (energy_out && lane == 0) energy_out[b] = s_energy[lat.n_steps];
}

template <typename T>
__global__ __launch_bounds__(256) void k_track_moments(LatticeDev lat, const T* __restrict__ energy_in,
                                                        const T* mu_in, const T* cov_in, T* mu_out,
                                                        T* cov_out, T* __restrict__ energy_out, int chunk) {
  track_moments_sample<T>(lat, energy_in, mu_in, cov_in, mu_out, cov_out, energy_out, chunk);
}

// ... of a small lattice, whose parameters come with the arguments (InlinePool)
template <typename T, int BYTES>
__global__ __launch_bounds__(256) void k_track_moments_inline(InlinePool<BYTES> /* read in place: inline_pool_view */, LatticeDev lat,
                                                               const T* __restrict__ energy_in, const T* mu_in,
                                                               const T* cov_in, T* mu_out, T* cov_out,
                                                               T* __restrict__ energy_out, int chunk) {
  track_moments_sample<T>(inline_pool_view(lat), energy_in, mu_in, cov_in, mu_out, cov_out, energy_out, chunk);
}

// ---------------------------------------------------------------------------------------
// k_apply_moments_lanes: ParameterBeam path for LARGE batches (hundreds of thousands of settings in one
// call, tests/test_vectorized.py:298-321), lanes = samples: the step table comes from the lanes build,
// a wave stages the 64 table rows of a step in LDS with coalesced loads, every lane takes its own row
// and propagates ITS sample's moments in registers -- mu' = T mu, cov' = T cov T^T as X = cov T^T (row by
// row in place) and cov' = T X (column by column in place), then the cavity branch (cavity.py:134-140,
// 202-218).  Same operations in the same order per sample as k_track_moments, which keeps one
// workgroup per sample and spends most of its lanes waiting (3 x 100 000 settings of the ARES lattice:
// 1.56 ms there).
// ---------------------------------------------------------------------------------------
template <typename T> constexpr size_t apply_moments_lds() { return 64 * 68 * sizeof(T); }

template <typename T>
__global__ __launch_bounds__(64) void k_apply_moments_lanes(LatticeDev lat, const T* __restrict__ steps,
                                                            const T* __restrict__ mu_in, const T* __restrict__ cov_in,
                                                            T* __restrict__ mu_out, T* __restrict__ cov_out,
                                                            T* __restrict__ energy_out) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  using V = typename VecOf<T, true>::type;
  constexpr int W = VecOf<T, true>::width, kVecPerRow = LYNX_STEP_STRIDE / W;
  T* rows = reinterpret_cast<T*>(smem_raw);
  const int lane = threadIdx.x, S = lat.n_steps;
  const int64_t b0 = (int64_t)blockIdx.x * 64, B = lat.batch;
  const bool live = b0 + lane < B;
  const int64_t b = live ? b0 + lane : B - 1;
  T mu[7], C[49];
#pragma unroll
  for (int q = 0; q < 7; ++q) mu[q] = mu_in[b * 7 + q];
#pragma unroll
  for (int q = 0; q < 49; ++q) C[q] = cov_in[b * 49 + q];
  for (int s = 0; s < S; ++s) {
    wave_fence();
    // the wave's 64 rows of step s: [sample][64] scalars, 16-byte pieces, consecutive lanes -> consecutive pieces
#pragma unroll 4
    for (int v = lane; v < 64 * kVecPerRow; v += 64) {
      const int r = v / kVecPerRow, piece = v - r * kVecPerRow;
      const int64_t br = (b0 + r < B) ? b0 + r : B - 1;
      const V x = *reinterpret_cast<const V*>(steps + (br * S + s) * LYNX_STEP_STRIDE + piece * W);
      *reinterpret_cast<V*>(rows + r * 68 + piece * W) = x;
    }
    wave_fence();
    const T* M = rows + lane * 68;
    const int desc = (int)M[LYNX_FLAGS_OFFSET];
    const bool kick = ((desc >> LYNX_DESC_KIND_SHIFT) & 3) == LYNX_STEP_CAVITY && (desc & LYNX_FLAG_CAV_GAIN);
    const T s_in = mu[4], d_in = mu[5], c44 = C[4 * 7 + 4], c45 = C[4 * 7 + 5], c55 = C[5 * 7 + 5];
    // mu' = T mu
    {
      T out[7];
#pragma unroll
      for (int i = 0; i < 7; ++i) {
        T acc = M[i * 7] * mu[0];
#pragma unroll
        for (int k = 1; k < 7; ++k) acc = t_fma(M[i * 7 + k], mu[k], acc);
        out[i] = acc;
      }
#pragma unroll
      for (int i = 0; i < 7; ++i) mu[i] = out[i];
    }
    // X = cov . T^T, row i of X from row i of cov
#pragma unroll
    for (int i = 0; i < 7; ++i) {
      T out[7];
#pragma unroll
      for (int j = 0; j < 7; ++j) {
        T acc = C[i * 7] * M[j * 7];
#pragma unroll
        for (int k = 1; k < 7; ++k) acc = t_fma(C[i * 7 + k], M[j * 7 + k], acc);
        out[j] = acc;
      }
#pragma unroll
      for (int j = 0; j < 7; ++j) C[i * 7 + j] = out[j];
    }
    // cov' = T . X, column j of cov' from column j of X
#pragma unroll
    for (int j = 0; j < 7; ++j) {
      T out[7];
#pragma unroll
      for (int i = 0; i < 7; ++i) {
        T acc = M[i * 7] * C[j];
#pragma unroll
        for (int k = 1; k < 7; ++k) acc = t_fma(M[i * 7 + k], C[k * 7 + j], acc);
        out[i] = acc;
      }
#pragma unroll
      for (int i = 0; i < 7; ++i) C[i * 7 + j] = out[i];
    }
    if (kick) {
      const T* coef = M + LYNX_COEF_OFFSET;  // (the row itself: the kick reads sin(phi) behind the coefficients)
      T s_o = mu[4], d_o;
      device_cavity_kick<T>(coef, s_in, d_in, s_o, d_o);  // cavity.py:134-140, 202-206
      mu[4] = s_o;
      mu[5] = d_o;
      C[5 * 7 + 5] = c55;  // cavity.py:140
      const T v = coef[LYNX_C_T566] * (c55 * c55) + coef[LYNX_C_T556] * c45 * c55 + coef[LYNX_C_T555] * (c44 * c44);  // cavity.py:207-218
      C[4 * 7 + 4] = v;
      C[4 * 7 + 5] = v;
      C[5 * 7 + 4] = v;
    }
    if (s == S - 1 && energy_out && live) energy_out[b] = M[LYNX_ENERGY_OFFSET];
  }
  if (live) {
#pragma unroll
    for (int q = 0; q < 7; ++q) mu_out[b * 7 + q] = mu[q];
#pragma unroll
    for (int q = 0; q < 49; ++q) cov_out[b * 49 + q] = C[q];
  }
}

// ---------------------------------------------------------------------------------------
// k_histogram2d: Screen read-out of a ParticleBeam.  Edges are staged in LDS; every thread
// walks particles of one sample (grid.y... folded into blockIdx.x), finds both bins by
// bisection on the very edge arrays NumPy would use (bit-exact bin assignment) and adds 1
// with an int atomic (exact, order-independent).
// ---------------------------------------------------------------------------------------
template <typename T>
__device__ __forceinline__ int bin_of(const T* edges, int n, T v) {
  // number of edges <= v, minus one; the last edge belongs to the last bin (numpy.histogramdd)
  if (!(v >= edges[0]) || !(v <= edges[n])) return -1;
  int lo = 0, hi = n + 1;  // invariant: edges[lo] <= v, (hi == n + 1 or edges[hi] > v)
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (edges[mid] <= v) lo = mid; else hi = mid;
  }
  return lo == n ? n - 1 : lo;
}

template <typename T>
__global__ __launch_bounds__(256) void k_histogram2d(const T* __restrict__ p, int64_t N, int chunks,
                                                      const T* __restrict__ xedges, const T* __restrict__ yedges,
                                                      int nx, int ny, int32_t* __restrict__ image) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  T* s_x = reinterpret_cast<T*>(smem_raw);
  T* s_y = s_x + nx + 1;
  for (int i = threadIdx.x; i <= nx; i += blockDim.x) s_x[i] = xedges[i];
  for (int i = threadIdx.x; i <= ny; i += blockDim.x) s_y[i] = yedges[i];
  __syncthreads();
  const int64_t b = blockIdx.x / chunks;
  const int chunk = blockIdx.x % chunks;
  const int64_t per = (N + chunks - 1) / chunks;
  const int64_t lo = chunk * per, hi = (lo + per) < N ? (lo + per) : N;
  int32_t* img = image + b * (int64_t)nx * ny;
  for (int64_t i = lo + threadIdx.x; i < hi; i += blockDim.x) {
    const T* q = p + (b * N + i) * 7;
    const int ix = bin_of<T>(s_x, nx, q[0]);
    const int iy = bin_of<T>(s_y, ny, q[2]);
    if (ix >= 0 && iy >= 0) atomicAdd(img + (int64_t)(ny - 1 - iy) * nx + ix, 1);
  }
}

// k_gaussian_image: Screen read-out of a ParameterBeam, one thread per pixel.
template <typename T>
__global__ __launch_bounds__(256) void k_gaussian_image(const T* __restrict__ mu, const T* __restrict__ cov,
                                                         const T* __restrict__ xs, const T* __restrict__ ys, int nx,
                                                         int ny, T* __restrict__ image) {
  const int64_t b = blockIdx.y;
  const int64_t pix = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (pix >= (int64_t)nx * ny) return;
  const int i = (int)(pix / ny), j = (int)(pix % ny);
  const T mx = mu[b * 7 + 0], my = mu[b * 7 + 2];
  const T a = cov[b * 49 + 0], bb = cov[b * 49 + 2], c = cov[b * 49 + 2 * 7 + 2];
  const T det = a * c - bb * bb;
  const T dx = xs[i] - mx, dy = ys[j] - my;
  const T maha = (c * dx * dx - T(2) * bb * dx * dy + a * dy * dy) / det;
  // exp(log_prob) of MultivariateNormal: -maha/2 - log(2 pi) - log(det)/2
  const T logp = T(-0.5) * maha - T(1.8378770664093453) - T(0.5) * t_log(det);
  image[(b * nx + (nx - 1 - i)) * (int64_t)ny + j] = (T)exp((double)logp);
}

// ---------------------------------------------------------------------------------------
// Aperture (lynx/accelerator/aperture.py:69-108): which particles survive, and the stable
// compaction of one sample's survivors / lost particles.
//   k_aperture_mask     mask[b][n] and the survivor count of every 1024-particle chunk
//   k_aperture_scan     per sample: exclusive scan of the chunk counts, total survivors
//   k_aperture_compact  one sample: survivors to `kept` in their original order, the rest to
//                       `lost` in their original order (what boolean-mask indexing returns)
// ---------------------------------------------------------------------------------------
constexpr int kApertureChunk = 1024;  // 256 threads x 4 consecutive particles

template <typename T>
__device__ __forceinline__ bool aperture_survives(T x, T y, T xm, T ym, int elliptical) {
  if (elliptical) return (x * x) / (xm * xm) + (y * y) / (ym * ym) <= T(1);  // aperture.py:83-86
  return (x > -xm && x < xm) && (y > -ym && y < ym);                         // aperture.py:78-82
}

template <typename T>
__global__ __launch_bounds__(256) void k_aperture_mask(const T* __restrict__ p, int64_t N, int chunks,
                                                        const T* __restrict__ x_max, const T* __restrict__ y_max,
                                                        int param_stride, int elliptical,
                                                        unsigned char* __restrict__ mask, int32_t* __restrict__ counts) {
  __shared__ int s_count;
  const int64_t b = blockIdx.x / chunks;
  const int chunk = blockIdx.x % chunks;
  const T xm = x_max[b * param_stride], ym = y_max[b * param_stride];
  if (threadIdx.x == 0) s_count = 0;
  __syncthreads();
  int mine = 0;
  for (int q = 0; q < 4; ++q) {
    const int64_t n = (int64_t)chunk * kApertureChunk + threadIdx.x * 4 + q;
    if (n < N) {
      const T* z = p + (b * N + n) * 7;
      const bool ok = aperture_survives<T>(z[0], z[2], xm, ym, elliptical);
      mask[b * N + n] = ok ? 1 : 0;
      mine += ok ? 1 : 0;
    }
  }
  atomicAdd(&s_count, mine);  // integer: exact and order-independent
  __syncthreads();
  if (threadIdx.x == 0) counts[b * chunks + chunk] = s_count;
}

__global__ __launch_bounds__(256) void k_aperture_scan(const int32_t* __restrict__ counts, int chunks,
                                                        int64_t* __restrict__ offsets, int64_t* __restrict__ totals) {
  // one workgroup per sample; thread 0 walks the chunk counts (chunks = N / 1024: short)
  const int64_t b = blockIdx.x;
  if (threadIdx.x == 0) {
    int64_t run = 0;
    for (int c = 0; c < chunks; ++c) {
      offsets[b * chunks + c] = run;
      run += counts[b * chunks + c];
    }
    totals[b] = run;
  }
}

template <typename T>
__global__ __launch_bounds__(256) void k_aperture_compact(const T* __restrict__ p, int64_t N,
                                                           const unsigned char* __restrict__ mask,
                                                           const int64_t* __restrict__ offsets, T* __restrict__ kept,
                                                           T* __restrict__ lost) {
  __shared__ int s_scan[256];
  const int chunk = blockIdx.x, tid = threadIdx.x;
  const int64_t first = (int64_t)chunk * kApertureChunk + tid * 4;
  bool ok[4];
  int mine = 0;
  for (int q = 0; q < 4; ++q) {
    ok[q] = first + q < N && mask[first + q] != 0;
    mine += ok[q] ? 1 : 0;
  }
  s_scan[tid] = mine;
  __syncthreads();
  for (int off = 1; off < 256; off <<= 1) {  // inclusive scan of the 256 per-thread counts
    const int add = tid >= off ? s_scan[tid - off] : 0;
    __syncthreads();
    s_scan[tid] += add;
    __syncthreads();
  }
  int64_t k = offsets[chunk] + (s_scan[tid] - mine);  // survivors before this thread's particles
  for (int q = 0; q < 4; ++q) {
    const int64_t n = first + q;
    if (n >= N) break;
    T* dst = ok[q] ? kept + k * 7 : lost + (n - k) * 7;  // n - k = lost particles before n
#pragma unroll
    for (int c = 0; c < 7; ++c) dst[c] = p[n * 7 + c];
    k += ok[q] ? 1 : 0;
  }
}

// k_diag_phase_trig: phase_sincos on an array, scalar and packed-pair code paths (test hook for
// the accuracy statement above); bit 1 of `packed`: the cosine from phase_cos, the forward kernels' own.
__global__ __launch_bounds__(256) void k_diag_phase_trig(const float* __restrict__ x, int64_t n, int packed,
                                                          float* __restrict__ s_out, float* __restrict__ c_out) {
  const int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 2;
  if (i >= n) return;
  const int64_t j = i + 1 < n ? i + 1 : i;
  const bool cos_alone = (packed & 2) != 0;
  if (packed & 1) {
    lynx_f32x2 s, c;
    phase_sincos(lynx_f32x2{x[i], x[j]}, s, c);
    if (cos_alone) c = phase_cos(lynx_f32x2{x[i], x[j]});
    s_out[i] = s.x; c_out[i] = c.x;
    s_out[j] = s.y; c_out[j] = c.y;
  } else {
    float s, c;
    phase_sincos(x[i], s, c);
    s_out[i] = s; c_out[i] = cos_alone ? phase_cos(x[i]) : c;
    phase_sincos(x[j], s, c);
    s_out[j] = s; c_out[j] = cos_alone ? phase_cos(x[j]) : c;
  }
}

// ---------------------------------------------------------------------------------------
// k_diag_copy: plain 16-byte grid-stride copy.  Calibration only: what this box sustains for
// a read+write stream of the same size as a tracking pass (the practical HBM ceiling).
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_diag_copy(const lynx_f32x4* __restrict__ src,
                                                    lynx_f32x4* __restrict__ dst, int64_t n_vec, int vec_per_thread) {
  // vec_per_thread == 0: grid-stride; > 0: each workgroup copies one contiguous block of
  // 256 * vec_per_thread vectors, workgroups in linear order
  // vec_per_thread >= 100: the same with non-temporal stores (what the wave-tile form of k_track_direct uses)
  const bool nt = vec_per_thread >= 100;
  if (nt) vec_per_thread -= 100;
  if (vec_per_thread == 0) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_vec; i += stride) dst[i] = src[i];
  } else {
    const int64_t base = (int64_t)blockIdx.x * 256 * vec_per_thread + threadIdx.x;
    for (int k = 0; k < vec_per_thread; ++k) {
      const int64_t i = base + (int64_t)k * 256;
      if (i < n_vec) {
        if (nt) __builtin_nontemporal_store(src[i], dst + i);
        else dst[i] = src[i];
      }
    }
  }
}

// ---------------------------------------------------------------------------------------
// k_fill_gaussian: counter-based synthetic beam (splitmix64 -> Box-Muller), one scalar of
// the flat [B][N][7] array per thread, fully coalesced stores.
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ uint64_t splitmix64(uint64_t x) {
  x += 0x9E3779B97F4A7C15ull;
  x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
  x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
  return x ^ (x >> 31);
}

struct GaussArgs {
  double mu[6];
  double sigma[6];
};

template <typename T>
__global__ __launch_bounds__(256) void k_fill_gaussian(T* __restrict__ p, int64_t total, uint64_t seed,
                                                        GaussArgs g) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += stride) {
    const int c = (int)(idx % 7);
    T v;
    if (c == 6) {
      v = T(1);
    } else {
      const uint64_t h1 = splitmix64(seed ^ splitmix64((uint64_t)idx * 2 + 0));
      const uint64_t h2 = splitmix64(seed ^ splitmix64((uint64_t)idx * 2 + 1));
      const double u1 = ((double)(h1 >> 11) + 1.0) * (1.0 / 9007199254740993.0);  // (0,1)
      const double u2 = (double)(h2 >> 11) * (1.0 / 9007199254740992.0);          // [0,1)
      const double r = sqrt(-2.0 * log(u1));
      const double zn = r * cos(6.283185307179586 * u2);
      v = (T)(g.mu[c] + g.sigma[c] * zn);
    }
    p[idx] = v;
  }
}

}  // namespace lynx
