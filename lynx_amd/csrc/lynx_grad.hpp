// Reverse pass: gradient of a scalar function of the OUTPUT-BEAM MOMENTS with respect to
// every element parameter and the incoming beam energy (BASELINE
// config 5).  The reference has no implementation of this (its differentiability tests
// still assert torch `grad_fn`, tests/test_differentiable.py:28-51); the checker is central
// finite differences of the oracle's forward pass (tests/test_gpu_grad.py).
//
// Forward, per sample:   theta --build--> M_e --compose--> T_s, coef_s --stream--> z_out
//                        --reduce--> (mean, cov) --user--> L
// Reverse:
//   k_track_bwd   z_bar = dL/dz_out from (mean_bar, cov_bar); walks the steps backwards per
//                 particle (forward states parked in LDS), z_bar <- T_s^T z_bar (+ cavity
//                 terms), and accumulates  T_bar_s = sum_n z_bar_n (x) z_n^(s)  and the
//                 cavity-coefficient cotangents.  The 64-lane sum of outer products is done
//                 as a 7x64 . 64x7 product through a per-wave LDS exchange buffer (one lane
//                 per output entry), not as 57 separate lane reductions.
//                 (fp32 MFMA was evaluated for this 7 x K GEMM: it runs at the VALU rate and
//                 padding 7 -> 16 wastes 81 % of it.)
//   k_reduce_tbar partial sums over a sample's workgroups -> T_bar[B][S][64]
//   k_build_bwd   per sample: prefix products P_j = M_j ... M_1, reverse sweep
//                 M_bar_j = A P_{j-1}^T, A <- M_j^T A, then theta_bar = <M_bar, dM/dtheta>
//                 with dM/dtheta from a dual-number evaluation of the very same builders, and
//                 the energy cotangent chained back through the cavities' energy gains.
#pragma once

#include "lynx_device.hpp"
#include "lynx_dual.hpp"
#include "lynx_unit_record.hpp"

namespace lynx {

constexpr int kGradStride = 64;   // per (sample, step): 49 T_bar + 8 coef_bar + padding
constexpr int kGradParams = 8;    // gradient slots per element (kinds with <= 8 parameters)
// k_build_bwd's task list: every (element, parameter or energy) pair as element * 16 + (parameter | kGradParams for the
// energy), 0xffff = padding.  The host lists them KIND BY KIND, each kind padded to whole waves (bwd_tasks_sorted), once
// per lattice: a wave then runs ONE kind's builder on dual numbers, the kinds run side by side on different waves, and
// slots of parameters a kind does not have do not exist (BASELINE config 5: 192 slots of 3 kinds instead of 288 over
// mixed waves).  (Until round 4 thread 0 of every workgroup sorted the list itself, after the chains: 80 of the
// kernel's 380 us.)
__host__ __device__ inline int bwd_kind_params_dev(int kind) {
  switch (kind) {
    case LYNX_KIND_DRIFT: return 1;
    case LYNX_KIND_QUADRUPOLE: return 5;
    case LYNX_KIND_DIPOLE: return 8;
    case LYNX_KIND_HCOR:
    case LYNX_KIND_VCOR: return 2;
    case LYNX_KIND_CAVITY: return 4;
    case LYNX_KIND_BASE_RMATRIX: return 4;
    case LYNX_KIND_ROTATION: return 1;
    case LYNX_KIND_MISALIGNMENT: return 3;
    case LYNX_KIND_SOLENOID: return 4;
    case LYNX_KIND_UNDULATOR: return 1;
    default: return 0;  // identity; custom maps carry no differentiable parameters here
  }
}
inline int bwd_kind_params(int kind) { return bwd_kind_params_dev(kind); }

constexpr int kBwdGroup = 4;
constexpr int kBwdMaxGroups = 16;  // => at most 64 units

// The sweeps of k_track_bwd walk UNITS: a step of the program, or (float32 packed pairs) a [run, cavity]
// pair in the merged form the forward kernel uses (one 7x7 application instead of two, see k_build).
struct BwdArgs {
  int64_t n_particles;
  int32_t chunks;
  int32_t tiles_per_wg;  // tiles of 256 particles
  int32_t n_units;
  // table slot of each unit's map; a merged pair: the cavity's slot (T_cav . T_run + coefficients), the
  // run's slot in front of it holds the two rows that give the s and delta entering the cavity and
  // carries LYNX_DESC_PAIR in its descriptor
  unsigned char unit_slot[kBwdGroup * kBwdMaxGroups];
  // observer steps (active BPMs, LYNX_STEP_FLAG_OBSERVE): 0 = this unit is none, k + 1 = it is the program's k-th one.
  // Its reading is the mean x and y of the beam that ENTERS it (bpm.py:48-54): linear in the particles, so a cotangent
  // of the reading is one more term of every particle's cotangent at that point of the reverse sweep.
  unsigned char unit_observer[kBwdGroup * kBwdMaxGroups];
  int32_t n_observers;
  // rows per sample of the partial-sum table when that is more than this launch's `chunks` (0: `chunks`): k_track_bwd
  // launched with ONE workgroup per sample next to k_track_bwd_units (track_backward_t) writes row 0 and clears the rest
  int32_t out_chunks;
  int32_t n_work;     // k_track_bwd: pieces of work (sample, chunk) in this launch
  int32_t leave_odd;  // tests (LYNX_BWD_UNITS=2): every odd sample is left to k_track_bwd whatever class its units have
};

// ---------------------------------------------------------------------------------------
// k_track_bwd: grid = B * chunks workgroups of 256 threads, one particle per thread and tile.
//
// The reverse sweep needs the state that ENTERED every step.  Every kBwdGroup-th state is
// parked during the forward sweep; inside a group the K-1 missing states are recomputed into
// registers (0.75 extra step applications per step for K = 4).  The parked states live in a
// per-lane private array (scratch memory: 7 dwords written and read per 4 steps, served by
// L1/L2), not in LDS: history of this kernel on BASELINE config 5 (16 steps) --
//   every state in LDS (476 B/particle)          1 workgroup/CU   20.7 ms
//   every 4th state in LDS, exchange pitch 68    2 workgroups/CU  10.0 ms
//   parked states in scratch, 14-row exchange    4 workgroups/CU   7.2 ms
//   + k-blocked exchange reads with DPP finishing, coefficient sums on idle lanes  6.9 ms
//   + two float32 particles per lane (v_pk_fma_f32)                  (see LaneOf below)
//
// LDS: exchange [4][21][ExGeom::kPitch] T | accumulators [4][S][64] T
// ---------------------------------------------------------------------------------------
// Rows of the exchange buffer: 14 (o_lin, z_in) + 1 (packed pairs: the cotangent of the s that enters a
// merged pair's cavity), or 21 when the first 7 cavity-coefficient cotangents also travel through it
// (summed by the otherwise idle lanes 56..63).  Packed pairs double the row length; with 21 rows only
// two workgroups fit a CU, with 15 three.
template <int W> struct ExRows { static constexpr int value = W == 1 ? 21 : 15; };

// LDS hand-over inside one wave: make this wave's LDS writes visible to its own later reads
// (and keep the compiler from moving accesses across); no workgroup barrier is involved.
__device__ __forceinline__ void wave_lds_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// Cross-lane sums without LDS traffic (DPP).  quad_perm [1,0,3,2] = 0xB1 and [2,3,0,1] = 0x4E
// swap within quads, row_half_mirror 0x141 reverses each group of 8 lanes, row_mirror 0x140
// each row of 16: after the first three every lane of an aligned group of 8 holds the group's
// sum, after the fourth every lane of a row; the 4 rows of the wave are added through SGPRs.
template <int CTRL> __device__ __forceinline__ float dpp_take(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
template <int CTRL> __device__ __forceinline__ double dpp_take(double v) {
  const long long bits = __builtin_bit_cast(long long, v);
  const int lo = __builtin_amdgcn_update_dpp(0, (int)bits, CTRL, 0xF, 0xF, true);
  const int hi = __builtin_amdgcn_update_dpp(0, (int)(bits >> 32), CTRL, 0xF, 0xF, true);
  return __builtin_bit_cast(double, ((long long)hi << 32) | (unsigned int)lo);
}
template <typename T> __device__ __forceinline__ T sum_over_8_lanes(T v) {
  v += dpp_take<0xB1>(v);
  v += dpp_take<0x4E>(v);
  v += dpp_take<0x141>(v);
  return v;
}
__device__ __forceinline__ float lane_value(float v, int l) {
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), l));
}
__device__ __forceinline__ double lane_value(double v, int l) {
  const long long bits = __builtin_bit_cast(long long, v);
  const int lo = __builtin_amdgcn_readlane((int)bits, l), hi = __builtin_amdgcn_readlane((int)(bits >> 32), l);
  return __builtin_bit_cast(double, ((long long)hi << 32) | (unsigned int)lo);
}
// Wave sums of EIGHT values at once: lane c (c < 8) ends up with the sum of v[c] over the 64 lanes.  Three
// halving steps over lane bits 0, 1, 2 -- a lane keeps the half of its values whose index bit matches its lane bit
// and hands the other half to its partner, which keeps exactly those -- leave every lane with ONE value, the sum
// over its aligned group of 8 lanes of v[lane & 7]; lane bits 3, 4, 5 then take one row rotation and two
// ds_bpermute.  ~30 VALU instructions where eight separate butterflies cost ~100 (float only).
template <int CTRL, int BANK_MASK> __device__ __forceinline__ float dpp_take_banks(float old, float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, old), __builtin_bit_cast(int, v), CTRL,
                                                                0xF, BANK_MASK, false));
}
// ... in two parts: sum_eight_over_8_lanes leaves lane l with the sum over its aligned group of 8 lanes of v[l & 7] (a
// kernel that sums the same eight quantities tile after tile can add these up per lane and finish once),
// finish_over_lane_groups adds the eight groups of the wave.
__device__ __forceinline__ float sum_eight_over_8_lanes(const float (&v)[8], int lane) {
  const bool b0 = (lane & 1) != 0, b1 = (lane & 2) != 0, b2 = (lane & 4) != 0;
  float a[4], g[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {  // bit 0: partner lane ^ 1 (quad_perm [1,0,3,2]); keeps index 2k + b0
    a[k] = b0 ? v[2 * k + 1] : v[2 * k];
    g[k] = b0 ? v[2 * k] : v[2 * k + 1];
    a[k] += dpp_take<0xB1>(g[k]);
  }
  float c2[2], g2[2];
#pragma unroll
  for (int k = 0; k < 2; ++k) {  // bit 1: partner lane ^ 2 (quad_perm [2,3,0,1]); index 4k + 2 b1 + b0
    c2[k] = b1 ? a[2 * k + 1] : a[2 * k];
    g2[k] = b1 ? a[2 * k] : a[2 * k + 1];
    c2[k] += dpp_take<0x4E>(g2[k]);
  }
  // bit 2: partner lane ^ 4 = row_shl:4 for banks 0 and 2 of a row, row_shr:4 for banks 1 and 3
  float one = b2 ? c2[1] : c2[0];
  const float give = b2 ? c2[0] : c2[1];
  float got = dpp_take_banks<0x104, 0x5>(0.f, give);   // row_shl:4 -> lanes with bit 2 clear read lane + 4
  got = dpp_take_banks<0x114, 0xA>(got, give);         // row_shr:4 -> lanes with bit 2 set read lane - 4
  one += got;
  return one;
}
__device__ __forceinline__ float finish_over_lane_groups(float one) {
  one += dpp_take<0x128>(one);                          // bit 3: row_ror:8 = lane ^ 8 within the row of 16
  one += __shfl_xor(one, 16, 64);                       // bits 4, 5: the four rows
  one += __shfl_xor(one, 32, 64);
  return one;  // lanes with (lane & 7) == c hold the total of v[c]
}
__device__ __forceinline__ float sum_eight_over_wave(const float (&v)[8], int lane) {
  return finish_over_lane_groups(sum_eight_over_8_lanes(v, lane));
}

template <typename T> __device__ __forceinline__ T sum_over_wave(T v) {
  v = sum_over_8_lanes(v);
  v += dpp_take<0x140>(v);
  return (lane_value(v, 0) + lane_value(v, 16)) + (lane_value(v, 32) + lane_value(v, 48));
}

// kind and flags of step s from the table's descriptor slot (a wave-uniform scalar load; `lat.steps[s]`
// would be a vector load + readfirstlane per step, see LYNX_FLAGS_OFFSET)
template <typename T>
__device__ __forceinline__ lynx_step table_step(const T* g_steps, int s) {
  const int desc = (int)uniform_value(g_steps[s * LYNX_STEP_STRIDE + LYNX_FLAGS_OFFSET]);
  lynx_step st;
  st.kind = (desc >> LYNX_DESC_KIND_SHIFT) & 3;
  st.flags = desc & 0xffff;
  st.first = 0;
  st.last = 0;
  return st;
}

template <typename T>
__device__ __forceinline__ void load_step_map(const T* g_steps, int s, T (&m)[LYNX_STEP_SCALARS]) {
#pragma unroll
  for (int q = 0; q < 57; ++q) m[q] = g_steps[s * LYNX_STEP_STRIDE + q];  // wave-uniform: scalar loads
#pragma unroll
  for (int q = 57; q < LYNX_SINPHI_OFFSET; ++q) m[q] = T(0);              // (the entry inverse: merged pairs fetch their own copy)
  m[LYNX_SINPHI_OFFSET] = g_steps[s * LYNX_STEP_STRIDE + LYNX_SINPHI_OFFSET];
}

// does the unit whose map sits in `slot` apply a [run, cavity] pair in merged form?  (wave-uniform)
template <typename T>
__device__ __forceinline__ bool unit_is_pair(const T* g_steps, int slot) {
  return slot > 0 && (((int)uniform_value(g_steps[(slot - 1) * LYNX_STEP_STRIDE + LYNX_FLAGS_OFFSET])) & LYNX_DESC_PAIR) != 0;
}

// ---- lane values: one particle per lane (float64) or two as a packed pair (float32) -----
// The reverse sweep is VALU-bound; with two float32 particles per lane every multiply-add of
// the step applications, of T^T z_bar and of the exchange products is one v_pk_fma_f32 for
// both (map entries broadcast from SGPRs), i.e. about 0.6x the instructions per particle.
template <typename Z> struct LaneOf { static constexpr int W = 1; };
template <> struct LaneOf<lynx_f32x2> { static constexpr int W = 2; };

__device__ __forceinline__ float zfma(float a, float b, float c) { return t_fma(a, b, c); }
__device__ __forceinline__ double zfma(double a, double b, double c) { return t_fma(a, b, c); }
__device__ __forceinline__ lynx_f32x2 zfma(lynx_f32x2 a, float b, lynx_f32x2 c) { return pk_fma(a, b, c); }
__device__ __forceinline__ lynx_f32x2 zfma(lynx_f32x2 a, lynx_f32x2 b, lynx_f32x2 c) {
  return __builtin_elementwise_fma(a, b, c);
}
__device__ __forceinline__ float zget(float z, int) { return z; }
__device__ __forceinline__ double zget(double z, int) { return z; }
__device__ __forceinline__ float zget(lynx_f32x2 z, int w) { return w ? z.y : z.x; }
__device__ __forceinline__ void zset(float& z, int, float v) { z = v; }
__device__ __forceinline__ void zset(double& z, int, double v) { z = v; }
__device__ __forceinline__ void zset(lynx_f32x2& z, int w, float v) {
  if (w) z.y = v;
  else z.x = v;
}
__device__ __forceinline__ float zhsum(float z) { return z; }
__device__ __forceinline__ double zhsum(double z) { return z; }
__device__ __forceinline__ float zhsum(lynx_f32x2 z) { return z.x + z.y; }
// sine and cosine of the cavity phase: phase_sincos of lynx_device.hpp (the forward kernels' cos)
template <typename Z> __device__ __forceinline__ void zsincos(Z x, Z& s, Z& c) { phase_sincos(x, s, c); }
__device__ __forceinline__ void zapply(const float* m, int kind, int flags, float (&z)[7]) { apply_step<float>(m, kind, flags, z); }
__device__ __forceinline__ void zapply(const double* m, int kind, int flags, double (&z)[7]) { apply_step<double>(m, kind, flags, z); }
__device__ __forceinline__ void zapply(const float* m, int kind, int flags, lynx_f32x2 (&z)[7]) { apply_step_pair(m, kind, flags, z); }

// Cotangents of an active cavity's kick (cavity.py:141-161, 219-226), shared by the three reverse kernels:
//   o5 = z5 c0 + c1 (cos(a) - c4),  a = -z4 c2 + c3;   o4 = o4_lin + c5 z5^2 + c6 z4 z5 + c7 z4^2
// with z4, z5 the s and delta that ENTER the cavity, o4b / o5b the cotangents of its outputs.  -> the eight coefficient
// cotangents cc[LYNX_C_*] and the direct terms d/dz4, d/dz5.
// The forward value repeats the reference's operations, cancellation included.  Its DERIVATIVES need the differences
// sin(a) - sin(phi) (phase) and cos(a) - cos(phi) (voltage) once more, and those are formed here without the
// cancellation, from d = a - phi = -z4 c2 (milliradians) and the addition theorems:
//   sin a - sin phi = sin phi (cos d - 1) + cos phi sin d,   cos a - cos phi = cos phi (cos d - 1) - sin phi sin d.
// The phase cotangent is handed over COMPLETE -- d/dc3 and the path through c4 = cos(phi) in one number, c4's own slot
// stays zero --, because as two separate sums over the particles (sum -o5b c1 sin a and sum -o5b c1, the second times
// -sin phi in k_build_bwd) they cancel to 1/300 of their size on a 10 um bunch: float32 gradients w.r.t. the phase were
// 1e-2 away from the float64 pass that way, voltage 2e-3 (tests/test_gpu_grad.py, BASELINE config 5 at its shape).
template <typename T, typename Z>
__device__ __forceinline__ void kick_cotangents(const T* cf, T sphi, T cphi, Z z4, Z z5, Z o4b, Z o5b, Z (&cc)[8], Z& dir4,
                                                Z& dir5) {
  const Z d = T(-1) * z4 * cf[LYNX_C_BK];
  Z sd, cm1;
  sin_cosm1(d, sd, cm1);
  const Z dsin = zfma(sd, cphi, cm1 * sphi);            // sin(a) - sin(phi)
  const Z dcos = zfma(sd, -sphi, cm1 * cphi);           // cos(a) - cos(phi)
  const Z sa = dsin + sphi;
  const Z ab = -o5b * cf[LYNX_C_DKICK] * sa;            // cotangent of a through the cosine
  cc[LYNX_C_DSCALE] = o5b * z5;
  cc[LYNX_C_DKICK] = o5b * dcos;
  cc[LYNX_C_BK] = ab * (-z4);
  cc[LYNX_C_PHI] = -o5b * cf[LYNX_C_DKICK] * dsin;      // complete: d/dc3 - sin(phi) d/dc4
  cc[LYNX_C_COSPHI] = Z(T(0));
  cc[LYNX_C_T566] = o4b * (z5 * z5);
  cc[LYNX_C_T556] = o4b * (z4 * z5);
  cc[LYNX_C_T555] = o4b * (z4 * z4);
  // (as chains of fused multiply-adds: 4 operations each where the expressions as written take 7)
  dir4 = zfma(o4b, zfma(z5, Z(cf[LYNX_C_T556]), z4 * (T(2) * cf[LYNX_C_T555])), ab * (-cf[LYNX_C_BK]));
  dir5 = zfma(o4b, zfma(z5, Z(T(2) * cf[LYNX_C_T566]), z4 * cf[LYNX_C_T556]), o5b * cf[LYNX_C_DSCALE]);
}
// sine and cosine of the cavity's phase, once per unit (wave-uniform)
template <typename T> __device__ __forceinline__ void phase_of(T phi, T& sphi, T& cphi) {
  T s, c;
  phase_sincos(phi, s, c);
  sphi = uniform_value(s);
  cphi = uniform_value(c);
}

// one unit of the forward / recompute sweeps
template <typename T, typename Z>
__device__ __forceinline__ void apply_unit(const T* g_steps, int slot, Z (&z)[7]) {
  const lynx_step st = table_step<T>(g_steps, slot);
  if constexpr (LaneOf<Z>::W == 2) {
    if (unit_is_pair<T>(g_steps, slot)) {  // uniform: as the forward kernel applies it (kEntryInverse)
      T m[LYNX_STEP_SCALARS];
#pragma unroll
      for (int q = 0; q < LYNX_STEP_SCALARS; ++q) m[q] = g_steps[slot * LYNX_STEP_STRIDE + q];
      const int desc = (int)uniform_value(g_steps[slot * LYNX_STEP_STRIDE + LYNX_FLAGS_OFFSET]);
      if (desc & LYNX_DESC_ILL) {  // uniform: the rows form, like the forward kernel
        T pre[14];
#pragma unroll
        for (int q = 0; q < 14; ++q) pre[q] = uniform_value(g_steps[(slot - 1) * LYNX_STEP_STRIDE + q]);
        lynx_f32x2 s_in, d_in;
        merged_pair_entry(pre, z, s_in, d_in);
        apply_step_pair(m, st.kind, st.flags, z, kEntryGiven, s_in, d_in);
      } else {
        apply_step_pair(m, st.kind, st.flags, z, kEntryInverse);
      }
      return;
    }
  }
  T m[LYNX_STEP_SCALARS];
  load_step_map<T>(g_steps, slot, m);
  zapply(m, st.kind, st.flags, z);
}

// Exchange-buffer geometry: a row holds the 64 W values of one quantity; lane (i, kb) works on
// the 16-byte pieces l * 32 + kb * 4 (in scalars of 4 bytes; 8-byte scalars: l * 16 + kb * 2)
// of its rows, so that the 8 lanes of one row group read 128 consecutive bytes per access.
template <typename T, int W> struct ExGeom {
  static constexpr int kVW = 16 / (int)sizeof(T);          // scalars per 16-byte access
  static constexpr int kRow = 64 * W;                      // scalars per row
  static constexpr int kPitch = kRow + kVW;                // + one access: rows start on different banks
  static constexpr int kPieces = kRow / (8 * kVW);         // accesses per lane and row
};

// what one workgroup of k_track_bwd does: chunk `work % a.chunks` of sample `work / a.chunks`
template <typename T, typename Z>
__device__ __forceinline__ void track_bwd_workgroup(
    const LatticeDev& lat, const BwdArgs& a, uint32_t work, const T* __restrict__ p_in, const T* __restrict__ steps,
    const double* __restrict__ moments_fwd, const double* __restrict__ grad_moments,
    T* __restrict__ partials /* [B][chunks][S][64] */, T* __restrict__ grad_p /* [B][N][7] or null */,
    const float* __restrict__ units_skip, int units_stride, int units_class_shift, int units_class_u,
    const double* __restrict__ grad_observations) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  constexpr int K = kBwdGroup;
  constexpr int W = LaneOf<Z>::W;
  if (units_skip) {  // uniform
    const float* rec = units_skip + (work / a.chunks) * (int64_t)a.n_units * units_stride;
    bool other = false;  // (one descriptor per lane, 64 at a time: one round trip where a loop over the units takes n_units)
    for (int u0 = 0; u0 < a.n_units; u0 += 64) {
      const int u = u0 + (int)(threadIdx.x & 63);
      other = other || (u < a.n_units && ((__float_as_int((float)rec[u * units_stride]) >> units_class_shift) & 3) != units_class_u);
    }
    const bool all = __builtin_amdgcn_ballot_w64(other) == 0;
    if (all && !(a.leave_odd && ((work / a.chunks) & 1))) return;
  }
  using Geo = ExGeom<T, W>;
  using V = typename VecOf<T, true>::type;  // 16-byte LDS accesses
  constexpr int VW = Geo::kVW, P = Geo::kPitch;
  const int S = lat.n_steps;
  const int U = a.n_units;
  const int G = (U + K - 1) / K;
  constexpr int kExRows = ExRows<W>::value;
  constexpr bool kCoefRows = kExRows == 21;
  T* s_ex = reinterpret_cast<T*>(smem_raw);                  // [4][kExRows][P]
  T* s_acc = s_ex + 4 * kExRows * P;                         // [4][S][64]
  Z stack[kBwdMaxGroups * 7];                                // private: state entering step g*K

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int64_t b = work / a.chunks;
  const int chunk = work % a.chunks;
  const int64_t N = a.n_particles;
  const T* g_steps = steps + b * (int64_t)S * LYNX_STEP_STRIDE;
  T* ex = s_ex + wave * (kExRows * P);
  T* acc = s_acc + wave * (S * 64);

  for (int s = 0; s < S; ++s) acc[s * 64 + lane] = T(0);

  const double* rec = moments_fwd + b * LYNX_MOMENT_STRIDE;
  const double* gm = grad_moments + b * LYNX_MOMENT_STRIDE;
  const T* src = p_in + b * N * 7;
  constexpr int64_t kTile = (int64_t)kTrackThreads * W;
  for (int it = 0; it < a.tiles_per_wg; ++it) {
    const int64_t base = ((int64_t)chunk * a.tiles_per_wg + it) * kTile;
    if (base >= N) break;  // uniform
    bool live[W];
    Z z[7];
#pragma unroll
    for (int w = 0; w < W; ++w) {
      const int64_t i = base + (int64_t)w * kTrackThreads + tid;
      live[w] = i < N;
      T zw[7];
      load_particle(src + (live[w] ? i : 0) * 7, zw);
#pragma unroll
      for (int c = 0; c < 7; ++c) zset(z[c], w, zw[c]);
    }

    // forward sweep, parking the state that enters every K-th unit
    for (int u = 0; u < U; ++u) {
      if (u % K == 0) {
#pragma unroll
        for (int c = 0; c < 7; ++c) stack[(u / K) * 7 + c] = z[c];
      }
      apply_unit<T, Z>(g_steps, a.unit_slot[u], z);
    }

    // cotangent of the outgoing particle: (1/N) (mu_bar + G_hat (z - mean)), G_hat built from
    // the upper-triangle cotangent (diagonal counted twice); the per-sample constants are
    // fetched here, per tile, so that they do not stay live across the sweeps
    Z zb[7];
    {
      const T inv_n = (T)(1.0 / rec[35]);
      Z d[6];
#pragma unroll
      for (int k = 0; k < 6; ++k) d[k] = z[k] - (T)rec[k];
#pragma unroll
      for (int k = 0; k < 6; ++k) {
        Z accv = Z((T)gm[k]);
#pragma unroll
        for (int j = 0; j < 6; ++j) {
          const int r = k < j ? k : j, c = k < j ? j : k;
          const T g = (T)gm[7 + r * 6 - (r * (r - 1)) / 2 + (c - r)];
          accv = zfma(d[j], k == j ? g + g : g, accv);
        }
        zb[k] = accv * inv_n;
      }
      zb[6] = Z((T)gm[6] * inv_n);
#pragma unroll
      for (int w = 0; w < W; ++w)
#pragma unroll
        for (int c = 0; c < 7; ++c) zset(zb[c], w, live[w] ? zget(zb[c], w) : T(0));
    }

    for (int grp = G - 1; grp >= 0; --grp) {
      // states entering units grp*K .. grp*K + K-1
      Z zz[K][7];
#pragma unroll
      for (int c = 0; c < 7; ++c) zz[0][c] = stack[grp * 7 + c];
#pragma unroll
      for (int j = 1; j < K; ++j) {
        const int up = grp * K + j - 1;  // unit that produces zz[j]
#pragma unroll
        for (int c = 0; c < 7; ++c) zz[j][c] = zz[j - 1][c];
        if (up + 1 < U) apply_unit<T, Z>(g_steps, a.unit_slot[up], zz[j]);
      }

      for (int j = K - 1; j >= 0; --j) {
        const int u = grp * K + j;
        if (u >= U) continue;  // uniform
        const int s = a.unit_slot[u];
        const lynx_step st = table_step<T>(g_steps, s);
        Z zin[7];
#pragma unroll
        for (int c = 0; c < 7; ++c) {
          Z v = zz[0][c];
#pragma unroll
          for (int q = 1; q < K; ++q) v = (j == q) ? zz[q][c] : v;
          zin[c] = v;
        }
        // what drives the cavity's kick: s and delta of the unit's incoming state, or -- merged pair --
        // of the state behind the pair's run: the two entry rows (run's slot) applied to z_in
        bool paired = false;
        Z s_in = zin[4], d_in = zin[5];
        if constexpr (W == 2) {
          if (unit_is_pair<T>(g_steps, s)) {  // uniform
            paired = true;
            float pre[14];
#pragma unroll
            for (int q = 0; q < 14; ++q) pre[q] = g_steps[(s - 1) * LYNX_STEP_STRIDE + q];
            merged_pair_entry(pre, zin, s_in, d_in);
            LYNX_FORGET();
          }
        }
        T m[LYNX_STEP_SCALARS];
        load_step_map<T>(g_steps, s, m);

        const bool kick = st.kind == LYNX_STEP_CAVITY && (st.flags & LYNX_FLAG_CAV_GAIN);
        const bool entry_rows = paired && kick;  // cotangents of the two entry rows are due
        Z olin[7], cc[8], dir4 = Z(T(0)), dir5 = Z(T(0));
#pragma unroll
        for (int c = 0; c < 7; ++c) olin[c] = zb[c];
        if (kick) {
          // o5' = z5 c0 + c1 (cos(a) - c4), a = -z4 c2 + c3 ; o4' = o4 + c5 z5^2 + c6 z4 z5 + c7 z4^2
          const T* cf = m + LYNX_COEF_OFFSET;
          T sphi, cphi;
          phase_of<T>(cf[LYNX_C_PHI], sphi, cphi);
          kick_cotangents<T, Z>(cf, sphi, cphi, s_in, d_in, zb[4], zb[5], cc, dir4, dir5);
          olin[5] = Z(T(0));  // the linear delta was overwritten
        }

        // Sums over the wave's 64 W particles through its exchange buffer: rows 0..6 o_lin,
        // 7..13 z_in, 14..20 the first 7 cavity-coefficient cotangents (kick steps only).
        // Lane (i, kb) = (lane / 8, lane % 8) owns output row i and one eighth of the
        // particles: it reads its pieces of row i once and the same pieces of all z_in rows
        // (lanes that share kb read the same addresses: broadcast), forms 7 partial sums, and
        // an 8-lane DPP butterfly finishes them -- half the wide LDS reads of a
        // lane-per-output mapping.  The 8 lanes with i = 7 run the same instructions on ones
        // against rows 14..20, which sums 7 of the 8 coefficient cotangents for free; the 8th
        // takes one DPP wave sum.
        //
        // Merged pair with a kick: o_lin[5] is zero (the linear delta was overwritten), so row 5 carries the
        // cotangent of the delta that ENTERS the cavity instead and the idle lanes 56..63 take the one of
        // s from row 14: their sums against z_in are the cotangents of the run slot's two entry rows.
        Z* exz = reinterpret_cast<Z*>(ex);
#pragma unroll
        for (int c = 0; c < 7; ++c) exz[(c * P) / W + lane] = (c == 5 && entry_rows) ? dir5 : olin[c];
#pragma unroll
        for (int c = 0; c < 7; ++c) exz[((7 + c) * P) / W + lane] = zin[c];
        if (W == 2 && entry_rows) exz[(14 * P) / W + lane] = dir4;
        if (kCoefRows && kick) {
#pragma unroll
          for (int c = 0; c < 7; ++c) exz[((14 + c) * P) / W + lane] = cc[c];
        }
        wave_lds_sync();
        {
          constexpr int NP = Geo::kPieces;
          const int oi = lane >> 3, kb = lane & 7;
          const bool coef_lane = kCoefRows && oi == 7;
          const bool entry_lane = W == 2 && entry_rows && oi == 7;
          const T* arow = ex + (coef_lane ? 0 : (entry_lane ? 14 : oi)) * P + kb * VW;
          const T* rows = ex + (coef_lane ? 14 : 7) * P + kb * VW;
          V av[NP];
#pragma unroll
          for (int l = 0; l < NP; ++l) {
            const V va = *reinterpret_cast<const V*>(arow + l * 8 * VW);
#pragma unroll
            for (int q = 0; q < VW; ++q) av[l][q] = coef_lane ? T(1) : va[q];
          }
          T mine = T(0);
#pragma unroll
          for (int j = 0; j < 7; ++j) {
            V part;
#pragma unroll
            for (int q = 0; q < VW; ++q) part[q] = T(0);
#pragma unroll
            for (int l = 0; l < NP; ++l) {
              const V vb = *reinterpret_cast<const V*>(rows + j * P + l * 8 * VW);
              part = __builtin_elementwise_fma(av[l], vb, part);
            }
            T folded = part[0];
#pragma unroll
            for (int q = 1; q < VW; ++q) folded += part[q];
            const T total = sum_over_8_lanes(folded);
            mine = (kb == j) ? total : mine;
          }
          if (kb < 7) {
            if (coef_lane) {
              if (kick) acc[s * 64 + 49 + kb] += mine;
            } else if (entry_lane) {
              acc[(s - 1) * 64 + kb] += mine;      // run slot, row of s
            } else if (oi < 7) {
              if (entry_rows && oi == 5) acc[(s - 1) * 64 + 7 + kb] += mine;  // run slot, row of delta
              else acc[s * 64 + oi * 7 + kb] += mine;
            }
          }
          if (kick) {
            // coefficient cotangents that did not travel through the buffer: DPP wave sums
            T rest = T(0);
            if constexpr (!kCoefRows && std::is_same<T, float>::value) {
              float each[8];  // all eight at once (sum_eight_over_wave): lane c < 8 receives the total of cc[c]
#pragma unroll
              for (int c = 0; c < 8; ++c) each[c] = zhsum(cc[c]);
              rest = sum_eight_over_wave(each, lane);
            } else {
#pragma unroll
              for (int c = kCoefRows ? 7 : 0; c < 8; ++c) {
                const T total = sum_over_wave(zhsum(cc[c]));
                rest = (lane == c) ? total : rest;
              }
            }
            if (lane >= (kCoefRows ? 7 : 0) && lane < 8) acc[s * 64 + 49 + lane] += rest;
          }
        }
        wave_lds_sync();

        // z_bar_in = T^T o_lin (+ direct cavity terms)
#pragma unroll
        for (int jj = 0; jj < 7; ++jj) {
          Z v = olin[0] * m[0 * 7 + jj];
#pragma unroll
          for (int r = 1; r < 7; ++r) v = zfma(olin[r], m[r * 7 + jj], v);
          zb[jj] = v;
        }
        if (grad_observations != nullptr && a.unit_observer[u] != 0) {  // uniform: an active BPM read the beam entering this unit
          const double* ob = grad_observations + (b * a.n_observers + (a.unit_observer[u] - 1)) * 2;
          const T inv_n = (T)(1.0 / rec[35]), gx = (T)ob[0] * inv_n, gy = (T)ob[1] * inv_n;
#pragma unroll
          for (int w = 0; w < W; ++w) {
            if (live[w]) {
              zset(zb[0], w, zget(zb[0], w) + gx);
              zset(zb[2], w, zget(zb[2], w) + gy);
            }
          }
        }
        if (!paired) {
          zb[4] += dir4;
          zb[5] += dir5;
        } else if (kick) {
          if constexpr (W == 2) {  // through the two entry rows back to the unit's incoming state
            LYNX_FORGET();
            float pre[14];
#pragma unroll
            for (int q = 0; q < 14; ++q) pre[q] = g_steps[(s - 1) * LYNX_STEP_STRIDE + q];
#pragma unroll
            for (int jj = 0; jj < 7; ++jj) zb[jj] = zfma(dir5, pre[7 + jj], zfma(dir4, pre[jj], zb[jj]));
          }
        }
      }
    }
    // what is left after the last (= first) step is dL/d(incoming particle)
    if (grad_p != nullptr) {
#pragma unroll
      for (int w = 0; w < W; ++w) {
        const int64_t i = base + (int64_t)w * kTrackThreads + tid;
        if (live[w]) {
          T zw[7];
#pragma unroll
          for (int c = 0; c < 7; ++c) zw[c] = zget(zb[c], w);
          store_particle(grad_p + (b * N + i) * 7, zw);
        }
      }
    }
  }

  __syncthreads();
  const int rows = a.out_chunks > 0 ? a.out_chunks : a.chunks;
  T* out = partials + (((int64_t)b * rows + chunk) * S) * kGradStride;
  for (int idx = tid; idx < S * 64; idx += kTrackThreads) {
    out[idx] = ((s_acc[idx] + s_acc[S * 64 + idx]) + s_acc[2 * S * 64 + idx]) + s_acc[3 * S * 64 + idx];
  }
  if (a.out_chunks > a.chunks) {  // (one workgroup for the whole sample: the rows of the workgroups that are not there)
    for (int64_t idx = tid; idx < (int64_t)(rows - a.chunks) * S * kGradStride; idx += kTrackThreads)
      partials[(((int64_t)b * rows + a.chunks) * S) * kGradStride + idx] = T(0);
  }
}

// grid = a.n_work (= B * chunks) workgroups, or fewer that take several pieces of work each (track_backward_t: the launch
// next to k_track_bwd_units, whose workgroups mostly find nothing to do)
template <typename T, typename Z>
__global__ __launch_bounds__(kTrackThreads) void k_track_bwd(
    LatticeDev lat, BwdArgs a, const T* __restrict__ p_in, const T* __restrict__ steps,
    const double* __restrict__ moments_fwd, const double* __restrict__ grad_moments,
    T* __restrict__ partials /* [B][chunks][S][64] */, T* __restrict__ grad_p /* [B][N][7] or null */,
    const float* __restrict__ units_skip = nullptr /* unit records: samples whose units are all of class U belong to
                                                       k_track_bwd_units (lynx_grad_units.hpp) */,
    int units_stride = 0, int units_class_shift = 0, int units_class_u = 0,
    const double* __restrict__ grad_observations = nullptr /* [B][n_observers][2] or null: dL/d(reading) */) {
  for (uint32_t work = blockIdx.x; work < (uint32_t)a.n_work; work += gridDim.x) {
    track_bwd_workgroup<T, Z>(lat, a, work, p_in, steps, moments_fwd, grad_moments, partials, grad_p, units_skip, units_stride,
                              units_class_shift, units_class_u, grad_observations);
    __syncthreads();  // (the LDS sums of this piece have been read by everybody before the next one clears them)
  }
}

// ---------------------------------------------------------------------------------------
// k_moments_bwd: reverse pass of k_track_moments (ParameterBeam), one wave per sample.
//   forward per step:  mu' = M mu,  C' = M C M^T,  then for an active cavity
//                      mu'[4] += T566 d^2 + T556 s d + T555 s^2,  mu'[5] = d c0 + c1 (cos a - c4),
//                      C'[5][5] = c55,  C'[4][4] = C'[4][5] = C'[5][4] = T566 c55^2 + T556 c45 c55 + T555 c44^2
//                      with s, d, c44, c45, c55 of the INCOMING beam (cavity.py:134-140, 202-218)
//   reverse per step:  G = C'_bar without the overwritten entries, m = mu'_bar without [5]
//                      T_bar = m (x) mu + (G M) C^T + (G^T M) C,   mu_bar = M^T m,   C_bar = M^T G M
//                      plus the cavity's direct terms and its 8 coefficient cotangents.
// The states entering every step are parked in HBM by the forward sweep ([B][S+1][56]).
// T_bar and the coefficient cotangents leave in the layout k_build_bwd consumes.
// ---------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(64) void k_moments_bwd(LatticeDev lat, const T* __restrict__ steps,
                                                    const T* __restrict__ mu_in, const T* __restrict__ cov_in,
                                                    const T* __restrict__ mu_bar_out, const T* __restrict__ cov_bar_out,
                                                    T* __restrict__ states, T* __restrict__ tbar,
                                                    T* __restrict__ grad_mu_in, T* __restrict__ grad_cov_in) {
  __shared__ T s_mu[8], s_c[49], s_x[49], s_y[49], s_g[49], s_mb[8], s_m[64], s_k[16];
  const int64_t b = blockIdx.x;
  const int lane = threadIdx.x;
  const int cl = lane < 49 ? lane : 48;
  const int i = cl / 7, j = cl % 7;
  const int S = lat.n_steps;
  const T* g_steps = steps + b * (int64_t)S * LYNX_STEP_STRIDE;
  T* st = states + b * (int64_t)(S + 1) * 56;

  if (lane < 7) s_mu[lane] = mu_in[b * 7 + lane];
  if (lane < 49) s_c[lane] = cov_in[b * 49 + lane];
  __syncthreads();
  for (int s = 0; s < S; ++s) {
    const lynx_step stp = lat.steps[s];
    if (lane < 7) st[s * 56 + lane] = s_mu[lane];
    if (lane < 49) st[s * 56 + 7 + lane] = s_c[lane];
    s_m[lane] = g_steps[s * LYNX_STEP_STRIDE + lane];
    __syncthreads();
    T mu_new = T(0);
    if (lane < 7) {
      mu_new = s_m[lane * 7] * s_mu[0];
#pragma unroll
      for (int k = 1; k < 7; ++k) mu_new = t_fma(s_m[lane * 7 + k], s_mu[k], mu_new);
    }
    T x = s_c[i * 7] * s_m[j * 7];
#pragma unroll
    for (int k = 1; k < 7; ++k) x = t_fma(s_c[i * 7 + k], s_m[j * 7 + k], x);
    const T s_i = s_mu[4], d_i = s_mu[5], c44 = s_c[32], c45 = s_c[33], c55 = s_c[40];
    __syncthreads();
    if (lane < 49) s_x[lane] = x;
    if (lane < 7) s_mu[lane] = mu_new;
    __syncthreads();
    T c = s_m[i * 7] * s_x[j];
#pragma unroll
    for (int k = 1; k < 7; ++k) c = t_fma(s_m[i * 7 + k], s_x[k * 7 + j], c);
    if (lane < 49) s_c[lane] = c;
    __syncthreads();
    if (stp.kind == LYNX_STEP_CAVITY && (stp.flags & LYNX_FLAG_CAV_GAIN) && lane == 0) {
      const T* coef = s_m + LYNX_COEF_OFFSET;
      T s_o = s_mu[4], d_o;
      device_cavity_kick<T>(coef, s_i, d_i, s_o, d_o);
      s_mu[4] = s_o;
      s_mu[5] = d_o;
      const T v = coef[LYNX_C_T566] * (c55 * c55) + coef[LYNX_C_T556] * c45 * c55 + coef[LYNX_C_T555] * (c44 * c44);
      s_c[40] = c55;
      s_c[32] = v;
      s_c[33] = v;
      s_c[39] = v;
    }
    __syncthreads();
  }

  if (lane < 7) s_mb[lane] = mu_bar_out[b * 7 + lane];
  if (lane < 49) s_g[lane] = cov_bar_out[b * 49 + lane];
  __syncthreads();
  for (int s = S - 1; s >= 0; --s) {
    const lynx_step stp = lat.steps[s];
    if (lane < 7) s_mu[lane] = st[s * 56 + lane];
    if (lane < 49) s_c[lane] = st[s * 56 + 7 + lane];
    s_m[lane] = g_steps[s * LYNX_STEP_STRIDE + lane];
    if (lane < 16) s_k[lane] = T(0);
    __syncthreads();
    const bool kick = stp.kind == LYNX_STEP_CAVITY && (stp.flags & LYNX_FLAG_CAV_GAIN);
    if (kick && lane == 0) {
      const T* cf = s_m + LYNX_COEF_OFFSET;
      const T z4 = s_mu[4], z5 = s_mu[5], c44 = s_c[32], c45 = s_c[33], c55 = s_c[40];
      const T m4 = s_mb[4], m5 = s_mb[5];
      const T vb = s_g[32] + s_g[33] + s_g[39];
      T sphi, cphi, kc[8], d4, d5;
      phase_sincos(cf[LYNX_C_PHI], sphi, cphi);
      kick_cotangents<T, T>(cf, sphi, cphi, z4, z5, m4, m5, kc, d4, d5);
#pragma unroll
      for (int q = 0; q < 8; ++q) s_k[q] = kc[q];
      s_k[LYNX_C_T566] += vb * (c55 * c55);
      s_k[LYNX_C_T556] += vb * (c45 * c55);
      s_k[LYNX_C_T555] += vb * (c44 * c44);
      s_k[8] = d4;  // d/d mu_in[4]
      s_k[9] = d5;  // d/d mu_in[5]
      s_k[10] = vb * T(2) * cf[LYNX_C_T555] * c44;                                                   // d/d c44
      s_k[11] = vb * cf[LYNX_C_T556] * c55;                                                          // d/d c45
      s_k[12] = s_g[40] + vb * (T(2) * cf[LYNX_C_T566] * c55 + cf[LYNX_C_T556] * c45);               // d/d c55
      s_mb[5] = T(0);  // these outputs were overwritten by the kick
      s_g[32] = T(0);
      s_g[33] = T(0);
      s_g[39] = T(0);
      s_g[40] = T(0);
    }
    __syncthreads();
    // P = G M, Q = G^T M
    T pv = s_g[i * 7] * s_m[j], qv = s_g[i] * s_m[j];
#pragma unroll
    for (int k = 1; k < 7; ++k) {
      pv = t_fma(s_g[i * 7 + k], s_m[k * 7 + j], pv);
      qv = t_fma(s_g[k * 7 + i], s_m[k * 7 + j], qv);
    }
    if (lane < 49) {
      s_x[lane] = pv;
      s_y[lane] = qv;
    }
    __syncthreads();
    T tb = s_mb[i] * s_mu[j], cb = T(0), mbn = T(0);
#pragma unroll
    for (int k = 0; k < 7; ++k) {
      tb = t_fma(s_x[i * 7 + k], s_c[j * 7 + k], tb);
      tb = t_fma(s_y[i * 7 + k], s_c[k * 7 + j], tb);
      cb = t_fma(s_m[k * 7 + i], s_x[k * 7 + j], cb);
    }
    if (lane < 7) {
#pragma unroll
      for (int k = 0; k < 7; ++k) mbn = t_fma(s_m[k * 7 + lane], s_mb[k], mbn);
    }
    T* tb_out = tbar + (b * S + s) * (int64_t)kGradStride;
    if (lane < 49) tb_out[lane] = tb;
    else if (lane < 57) tb_out[lane] = s_k[lane - 49];
    else tb_out[lane] = T(0);
    __syncthreads();
    if (lane < 49) s_g[lane] = cb;
    if (lane < 7) s_mb[lane] = mbn;
    __syncthreads();
    if (kick && lane == 0) {
      s_mb[4] += s_k[8];
      s_mb[5] += s_k[9];
      s_g[32] += s_k[10];
      s_g[33] += s_k[11];
      s_g[40] += s_k[12];
    }
    __syncthreads();
  }
  if (lane < 7) grad_mu_in[b * 7 + lane] = s_mb[lane];
  if (lane < 49) grad_cov_in[b * 49 + lane] = s_g[lane];
}

// partials [B][chunks][S][64] -> tbar [B][S][64]; one workgroup per (sample, step): one wave when a sample has a few
// chunks (large batches), four waves that take every fourth chunk each -- eight loads in flight per lane -- when it has
// many (one sample of 100 000 particles has 196: the single wave's chain of dependent loads took 41 us of a 145 us
// optimisation step).  Float64 sums, the waves' shares added in wave order: deterministic.
template <typename T>
__global__ __launch_bounds__(256) void k_reduce_tbar(const T* __restrict__ partials, int chunks, int S,
                                                      T* __restrict__ tbar) {
  __shared__ double s_part[4][64];
  const int64_t b = blockIdx.x / S;
  const int s = blockIdx.x % S;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, waves = blockDim.x >> 6;
  const T* src = partials + ((b * chunks) * S + s) * (int64_t)kGradStride + lane;
  const int64_t pitch = (int64_t)S * kGradStride;
  double v = 0.0;
  int c = wave;
  for (; c + 7 * waves < chunks; c += 8 * waves) {
    T x[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) x[k] = src[(int64_t)(c + k * waves) * pitch];
#pragma unroll
    for (int k = 0; k < 8; ++k) v += (double)x[k];
  }
  for (; c < chunks; c += waves) v += (double)src[(int64_t)c * pitch];
  if (waves > 1) {
    s_part[wave][lane] = v;
    __syncthreads();
    if (wave != 0) return;
    v = s_part[0][lane];
    for (int w = 1; w < waves; ++w) v += s_part[w][lane];
  }
  tbar[(b * S + s) * (int64_t)kGradStride + lane] = (T)v;
}

// The same for few workgroups per sample (chunks < 16): a wave adds up R rows (sample, step) of `chunks` partial sums each,
// all loads of a pass in flight at once -- one wave per row was 65 536 waves of ten dependent-free loads for BASELINE
// config 5, eight rounds of resident waves at one memory latency each (18 us).
template <typename T, int R>
__global__ __launch_bounds__(256) void k_reduce_tbar_rows(const T* __restrict__ partials, int chunks, int S, int64_t rows,
                                                           T* __restrict__ tbar) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t row0 = ((int64_t)blockIdx.x * 4 + wave) * R;
  const T* src[R];
  double v[R];
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const int64_t row = row0 + r < rows ? row0 + r : rows - 1;  // (clamped: read twice, written once)
    src[r] = partials + (((row / S) * chunks) * S + row % S) * (int64_t)kGradStride + lane;
    v[r] = 0.0;
  }
  const int64_t pitch = (int64_t)S * kGradStride;
  int c = 0;
  for (; c + 1 < chunks; c += 2) {
    T x[R][2];
#pragma unroll
    for (int r = 0; r < R; ++r) {
      x[r][0] = src[r][(int64_t)c * pitch];
      x[r][1] = src[r][(int64_t)(c + 1) * pitch];
    }
#pragma unroll
    for (int r = 0; r < R; ++r) v[r] = (v[r] + (double)x[r][0]) + (double)x[r][1];
  }
  if (c < chunks) {
#pragma unroll
    for (int r = 0; r < R; ++r) v[r] += (double)src[r][(int64_t)c * pitch];
  }
#pragma unroll
  for (int r = 0; r < R; ++r)
    if (row0 + r < rows) tbar[(row0 + r) * (int64_t)kGradStride + lane] = (T)v[r];
}

// LDS of k_build_bwd in front of the (optional) maps: energies and their cotangents [S+1] each, the per-element
// energy terms [E], and for each of the four waves the running cotangent A and a temporary [49] each
template <typename T> inline size_t build_bwd_lds_fixed(int S, int E) {
  return ((size_t)2 * (S + 1) + E + 4 * 98) * sizeof(T);
}

// ---------------------------------------------------------------------------------------
// k_build_bwd: one 256-thread workgroup per sample.
//   scratch (HBM, per sample): maps [E][49] and prefix/M_bar [E+1][49]
// ---------------------------------------------------------------------------------------
template <typename T>
__device__ __forceinline__ void build_bwd_sample(const LatticeDev& lat, const T* __restrict__ energy_in,
                                                 T* tbar, T* __restrict__ scratch,
                                                 T* __restrict__ grad_params /* [B][E][8] */,
                                                 T* __restrict__ grad_energy /* [B] */, int merged_pairs,
                                                 int maps_in_lds, const unsigned short* __restrict__ tasks, int n_tasks) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  const int E = lat.n_elems, S = lat.n_steps;
  T* s_energy = reinterpret_cast<T*>(smem_raw);  // [S+1]
  T* s_ebar = s_energy + (S + 1);                // [S+1] energy cotangent per step
  T* s_econ = s_ebar + (S + 1);                  // [E]   per-element contribution to its step's energy cotangent
  T* s_a = s_econ + E;                           // [4][98] per wave: running cotangent A [49], temp [49]
  T* s_maps = s_a + 4 * 98;                      // element maps and prefix products, when they fit (maps_in_lds)

  const int tid = threadIdx.x;
  const int64_t b = blockIdx.x;
  const T* pool = static_cast<const T*>(lat.pool);
  // The element maps and prefix products are walked in a chain of small dependent steps (49 threads, one 7-term
  // product each, a barrier in between): kept in LDS when they fit (maps_in_lds; BASELINE config 5: 16 KB),
  // every hop is an LDS round trip instead of one to L2 -- the kernel was latency-bound on exactly that
  // (0.88 -> 0.3 ms).  Long lattices (a 1051-element one needs 412 KB) keep them in the HBM scratch.
  T* g_maps = maps_in_lds ? s_maps : scratch + b * (int64_t)(2 * E + S + 1) * 49;  // M_e
  // prefix products of step s live in slots [first+s .. last+s]: slot(first+s) = start
  // (identity), slot(e+s+1) = M_e ... M_first; the reverse sweep then overwrites slot(e+s+1)
  // with M_bar_e.  (The +s keeps neighbouring steps from sharing a slot.)
  T* g_pref = g_maps + (int64_t)E * 49;

  // energies (same walk as the forward build).  The steps' energy gains are fetched by one thread per step --
  // lattice tables and parameters sit behind two dependent loads from memory, which the serial walk used to pay
  // step after step -- and the walk itself only adds them up.
  if (tid < S) {
    const lynx_step st = lat.steps[tid];
    T gain = T(0);
    if (st.kind == LYNX_STEP_CAVITY && (st.flags & LYNX_FLAG_CAV_GAIN)) {
      const lynx_elem el = lat.elems[st.first];
      const T* p = pool + el.param_offset + b * (int64_t)el.batch_stride;
      gain = p[1] * t_cos(p[2] * T(LYNX_PI / 180.0));
    }
    s_ebar[tid] = gain;  // parked here until the walk has used it
  }
  __syncthreads();
  if (tid == 0) {
    T e = energy_in[b];
    for (int s = 0; s < S; ++s) {
      s_energy[s] = e;
      e = e + s_ebar[s];  // same operation as `e + V cos(phi)` in the forward build; zero for every other step
    }
    s_energy[S] = e;
    for (int s = 0; s <= S; ++s) s_ebar[s] = T(0);
  }
  __syncthreads();

  // 1. element maps -> HBM scratch
  for (int e = tid; e < E; e += blockDim.x) {
    lynx_elem el = lat.elems[e];
    const int s = lat.elem_step[e];
    const T* p = pool + el.param_offset + b * (int64_t)el.batch_stride;
    T coef_unused[8];
    build_element<T>(el.kind, el.flags, p, s_energy[s], g_maps + (int64_t)e * 49,
                     lat.steps[s].kind == LYNX_STEP_CAVITY ? coef_unused : nullptr);
  }
  __threadfence_block();
  __syncthreads();

  // 2. per step: prefix products forward, cotangent sweep backward (49 lanes, one per entry).  The steps do not depend
  //    on each other -- except a run and the cavity it is merged with, which hand T_cav_bar over through the cavity's
  //    tbar row -- so they are dealt out to the four WAVES, a [run, cavity] pair as one item: each wave walks its
  //    steps' chains of small dependent products on its own, with hand-overs inside the wave instead of workgroup
  //    barriers (BASELINE config 5: 130 barrier rounds of the whole workgroup -> 32 hand-overs per wave, four chains
  //    side by side).
  const int wave = tid >> 6, lane = tid & 63;
  T* sa = s_a + wave * 98;   // [49] running cotangent A of this wave's current step
  T* sm = sa + 49;           // [49] temp
  const auto hand_over = [] {
    __threadfence_block();   // the maps may live in HBM scratch: this wave's stores have landed
    __builtin_amdgcn_wave_barrier();
  };
  const auto process_step = [&](int s) {
    const lynx_step st = lat.steps[s];
    const T* tb = tbar + (b * S + s) * (int64_t)kGradStride;
    const bool raw = st.kind == LYNX_STEP_CAVITY || (st.flags & LYNX_STEP_FLAG_RAW);
    // prefix: slot(e) = product of the step's elements up to and including e; slot(first-1)
    // is the start (identity); a raw step's first element has no predecessor product
    if (lane < 49) g_pref[(int64_t)(st.first + s) * 49 + lane] = (lane % 8 == 0) ? T(1) : T(0);
    hand_over();
    for (int e = st.first; e < st.last; ++e) {
      T v = T(0);
      if (lane < 49) v = mat_product_entry<T>(g_maps + (int64_t)e * 49, g_pref + (int64_t)(e + s) * 49, lane);
      if (lane < 49) g_pref[(int64_t)(e + s + 1) * 49 + lane] = v;
      hand_over();
    }
    if (merged_pairs && s + 1 < S && steps_pair_up(st, lat.steps[s + 1])) {  // uniform
      // k_track_bwd walked this run and the cavity behind it as ONE unit, M = T_cav . T_run, with the kick
      // driven by rows 4 and 5 of T_run: slot s + 1 holds M_bar, this slot the cotangents of those two rows.
      //   T_run_bar = T_cav^T M_bar (+ the two rows),   T_cav_bar = M_bar T_run^T  (parked for the cavity's turn)
      T* mbar = tbar + (b * S + s + 1) * (int64_t)kGradStride;
      const T* Tc = g_maps + (int64_t)lat.steps[s + 1].first * 49;
      const T* Tr = g_pref + (int64_t)(st.last + s) * 49;
      T run_bar = T(0), cav_bar = T(0);
      if (lane < 49) {
        const int i = lane / 7, j = lane - i * 7;
        run_bar = Tc[i] * mbar[j];
        cav_bar = mbar[i * 7] * Tr[j * 7];
#pragma unroll
        for (int k = 1; k < 7; ++k) {
          run_bar = t_fma(Tc[k * 7 + i], mbar[k * 7 + j], run_bar);
          cav_bar = t_fma(mbar[i * 7 + k], Tr[j * 7 + k], cav_bar);
        }
        if (i == 4) run_bar += tb[j];
        if (i == 5) run_bar += tb[7 + j];
      }
      hand_over();  // every entry of M_bar has been read
      if (lane < 49) {
        sa[lane] = run_bar;
        mbar[lane] = cav_bar;
      }
    } else if (lane < 49) {
      sa[lane] = tb[lane];
    }
    hand_over();
    for (int e = st.last - 1; e >= st.first; --e) {
      // M_bar_e = A . P_{e-1}^T ;  A <- M_e^T . A
      T mb = T(0), an = T(0);
      if (lane < 49) {
        const int i = lane / 7, j = lane - i * 7;
        const T* P = g_pref + (int64_t)(e + s) * 49;
        const T* M = g_maps + (int64_t)e * 49;
        if (raw && e == st.first) {
          mb = sa[lane];  // the map itself is the step's product start
        } else {
          T accv = sa[i * 7] * P[j * 7];
#pragma unroll
          for (int k = 1; k < 7; ++k) accv = t_fma(sa[i * 7 + k], P[j * 7 + k], accv);
          mb = accv;
        }
        T acc2 = M[0 * 7 + i] * sa[0 * 7 + j];
#pragma unroll
        for (int k = 1; k < 7; ++k) acc2 = t_fma(M[k * 7 + i], sa[k * 7 + j], acc2);
        an = acc2;
      }
      hand_over();  // A and P_{e-1} have been read by every lane
      if (lane < 49) {
        sa[lane] = an;
        g_pref[(int64_t)(e + s + 1) * 49 + lane] = mb;  // this slot now holds M_bar_e
      }
      hand_over();
    }
  };
  {
    int item = 0;
    for (int s = 0; s < S; ++item) {  // uniform walk over the items; wave (item mod 4) takes it
      const int n = (merged_pairs && s + 1 < S && steps_pair_up(lat.steps[s], lat.steps[s + 1])) ? 2 : 1;
      if ((item & 3) == wave)
        for (int q = 0; q < n; ++q) process_step(s + q);
      s += n;
    }
  }
  __syncthreads();

  // 3. theta_bar = <M_bar, dM/dtheta> (+ <coef_bar, dcoef/dtheta>) by dual evaluation; one task
  //    per (element, parameter), parameter index np = derivative w.r.t. the step energy.
  //    A task evaluates its element's builder on dual numbers: a wave whose lanes hold elements of different
  //    kinds runs every one of those builders one after the other.  With the maps in LDS (short lattices) the
  //    tasks are therefore listed kind by kind, each kind padded to whole waves -- a wave then runs ONE builder,
  //    the kinds run side by side on different waves, and slots of parameters a kind does not have do not
  //    exist (BASELINE config 5: 120 tasks of 3 kinds instead of 288 slots over mixed waves).
  for (int e = tid; e < E; e += blockDim.x) s_econ[e] = T(0);
  __syncthreads();
  for (int task = tid; task < n_tasks; task += blockDim.x) {
    const unsigned code = tasks[task];
    if (code == 0xffffu) continue;
    const int e = (int)(code >> 4), pidx = (int)(code & 15u);
    lynx_elem el = lat.elems[e];
    const int np = bwd_kind_params_dev(el.kind);
    const bool energy_task = pidx == kGradParams;
    T g = T(0);
    if ((pidx < np || energy_task) && np > 0) {
      const int s = lat.elem_step[e];
      const T* p = pool + el.param_offset + b * (int64_t)el.batch_stride;
      Dual<T> dp[8];
      for (int q = 0; q < 8; ++q) dp[q] = Dual<T>(q < np ? p[q] : T(0), (q == pidx) ? T(1) : T(0));
      const Dual<T> de(s_energy[s], energy_task ? T(1) : T(0));
      Dual<T> dC[8];
      const bool cav_step = lat.steps[s].kind == LYNX_STEP_CAVITY;
      for (int q = 0; q < 8; ++q) dC[q] = Dual<T>(T(0));
      const T* mb = g_pref + (int64_t)(e + s + 1) * 49;
      // Kinds whose maps have the structure of class U (drifts, correctors, untilted quadrupoles, cavities: all of
      // BASELINE config 5) are differentiated through their 16 entries in registers; a 49-entry array of dual numbers
      // per lane lives in scratch memory (656 bytes), which is what this kernel used to wait for.  An entry outside
      // the pattern has derivative zero, so leaving its term out is exact.
      Dual<T> m16[16];
      if (build_entries_u<Dual<T>>(el.kind, el.flags, dp, de, m16, cav_step ? dC : nullptr)) {
#pragma unroll
        for (int k = 0; k < 16; ++k) {
          const T w = mb[unit_entry_u(k)];
          if (w != T(0)) g = t_fma(w, m16[k].d, g);  // 0 * (inf or NaN derivative) must not poison the sum
        }
      } else {
        Dual<T> dM[49];
        build_element<Dual<T>>(el.kind, el.flags, dp, de, dM, cav_step ? dC : nullptr);
        for (int q = 0; q < 49; ++q) {
          const T w = mb[q];
          if (w != T(0)) g = t_fma(w, dM[q].d, g);
        }
      }
      if (cav_step) {
        const T* cb = tbar + (b * S + s) * (int64_t)kGradStride + LYNX_COEF_OFFSET;
        for (int q = 0; q < 8; ++q) {
          const T w = cb[q];
          if (w != T(0)) g = t_fma(w, dC[q].d, g);
        }
      }
    }
    if (energy_task) s_econ[e] = g;
    else grad_params[(b * E + e) * (int64_t)kGradParams + pidx] = g;
  }
  __syncthreads();

  // 4. energy cotangent: sum per step (one thread per step over its elements, in element order), the carries by a
  //    serial walk over the S sums, then back through the cavities' energy gains (one thread per step again)
  if (tid < S) {
    const lynx_step st = lat.steps[tid];
    T sum = T(0);
    for (int e = st.first; e < st.last; ++e) sum += s_econ[e];
    s_ebar[tid] = sum;
  }
  __syncthreads();
  if (tid == 0) {
    T carry = T(0);  // cotangent of the energy leaving step s
    for (int s = S - 1; s >= 0; --s) {
      const T own = s_ebar[s];
      s_ebar[s] = carry;
      carry += own;
    }
    grad_energy[b] = carry;
  }
  __syncthreads();
  if (tid < S) {
    const lynx_step st = lat.steps[tid];
    if (st.kind == LYNX_STEP_CAVITY && (st.flags & LYNX_FLAG_CAV_GAIN)) {
      const T carry = s_ebar[tid];
      const lynx_elem el = lat.elems[st.first];
      const T* p = pool + el.param_offset + b * (int64_t)el.batch_stride;
      const T phi = p[2] * T(LYNX_PI / 180.0);
      T* gp = grad_params + (b * E + st.first) * (int64_t)kGradParams;
      gp[1] += carry * t_cos(phi);                                   // dE_out/dV
      gp[2] += carry * (-p[1] * t_sin(phi)) * T(LYNX_PI / 180.0);    // dE_out/dphase[deg]
    }
  }
}

// (float32: compiled for seven waves per SIMD -- 72 registers and one accumulation register, which makes SIX: at the 132
// the compiler takes by itself three workgroups fit a CU, 768 on the GPU, and BASELINE config 5's 4096 samples are 5.3
// rounds of them: 210 us.  Four per CU (96 registers): 173; six: 153; seven (67 registers): 150.  The workgroup is a
// latency chain, not arithmetic: 720 bytes of scratch per lane instead of 608 cost it nothing that shows.)
template <typename T>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(sizeof(T) == 4 ? 7 : 1))) void k_build_bwd(LatticeDev lat, const T* __restrict__ energy_in,
                                                    T* tbar, T* __restrict__ scratch, T* __restrict__ grad_params,
                                                    T* __restrict__ grad_energy, int merged_pairs, int maps_in_lds,
                                                    const unsigned short* __restrict__ tasks, int n_tasks) {
  build_bwd_sample<T>(lat, energy_in, tbar, scratch, grad_params, grad_energy, merged_pairs, maps_in_lds, tasks, n_tasks);
}

// ... of a small lattice, whose parameters come with the arguments (InlinePool, lynx_device.hpp): a reverse pass right behind
// a parameter write does not have to bring the pool in HBM up to date first
template <typename T, int BYTES>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(sizeof(T) == 4 ? 7 : 1))) void k_build_bwd_inline(InlinePool<BYTES> /* read in place: inline_pool_view */, LatticeDev lat,
                                                           const T* __restrict__ energy_in, T* tbar, T* __restrict__ scratch,
                                                           T* __restrict__ grad_params, T* __restrict__ grad_energy,
                                                           int merged_pairs, int maps_in_lds,
                                                           const unsigned short* __restrict__ tasks, int n_tasks) {
  build_bwd_sample<T>(inline_pool_view(lat), energy_in, tbar, scratch, grad_params, grad_energy, merged_pairs, maps_in_lds, tasks,
                      n_tasks);
}

}  // namespace lynx
