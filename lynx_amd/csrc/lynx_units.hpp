// Multi-step float32 programs (cavities between runs: BASELINE config 5), structure-aware.
// Included only by lynx_hip.hip, after lynx_device.hpp.
//
// k_track_direct applies every step as a dense 7x7 (49 multiply-adds per particle), whatever the step is.  The maps
// of the path are far from dense: a run of drifts, quadrupoles, correctors and misaligned quadrupoles couples x with
// x', y with y', s with delta and nothing else, plus the affine column (track_methods.py:86-98, quadrupole.py:75-79,
// *_corrector.py); a cavity's map is three 2x2 blocks (cavity.py:311-323); the last row of every map of the path is
// e6.  A [run, cavity] pair of such elements has 16 entries that are not structurally zero, a run with untilted
// dipoles 24.  Here the program is walked as UNITS (a step, or a merged [run, cavity] pair) whose maps come from a
// COMPACT record holding exactly those entries:
//
//   class U  (uncoupled)        rows 0,1: cols {0,1,6}    rows 2,3: cols {2,3,6}    rows 4,5: cols {4,5}    16 entries
//   class D  (+ x dispersion)   rows 0,1: cols {0,1,5,6}  rows 2,3: cols {2,3,6}    rows 4,5: cols {0,1,4,5,6}  24
//   dense                       the step table's 7x7, as k_track_direct applies it
//
// The class of a unit is proposed by the host from the kinds and whole-batch flags of its elements (the same for
// every sample: it selects code) and CHECKED against the numbers by k_pack_units per sample and unit: every entry
// outside the pattern exactly zero, row 6 = e6, all 49 entries finite -- otherwise that sample takes the dense form
// for that unit.  Skipping a term m * z with m = +-0 is exact for finite z (it adds +-0: the only trace is the sign of
// an exactly-zero sum), so on finite beams the results are those of the dense chain, bit for bit.  A NON-FINITE
// particle is different -- 0 * inf = NaN spreads along the row in the reference's `P @ T^T` (element.py:85) -- so a
// wave whose tile holds a non-finite value, going in or coming out, (re)does the tile with the dense form.
//
// k_pack_units: step table [B][S][64] -> unit records [B][U][32] (+ [B][U][8] for class D), on the build's stream.
// k_track_units: the streaming kernel.  Same launch geometry, particle accesses, moment sums and epilogue as the
// two-particles-per-lane form of k_track_direct; what differs is the step loop.  k_track_direct fetches every step's
// 61 scalars from the 256-byte records of the step table, per unit and tile, behind a first fetch of the descriptor
// that tells a merged pair from a step -- dependent trips to L2 (a sample's 4 KB of records does not stay in the 16 KB
// scalar cache next to nine other workgroups'), which is what that loop is bound by (measured: with structured maps but
// the same fetches it ran in the same time).  Here a unit is ONE group of wave-uniform scalar loads from a 128-byte
// record -- 1 KB per sample for BASELINE config 5, which the scalar cache keeps -- followed by 16 / 24 packed
// multiply-adds instead of 49, the 2x2 blocks updated in place.  (The records in LDS, read with wave-uniform 16-byte
// loads, were tried first: 0.75 ms against 0.67 -- eight 1-KB LDS returns per unit and wave keep the LDS pipe three
// quarters as busy as the vector unit.)
#pragma once

#include "lynx_device.hpp"
#include "lynx_unit_record.hpp"

namespace lynx {

// ---------------------------------------------------------------------------------------
// k_pack_units: one thread per (sample, unit) -- for step tables that were not written by k_emit_steps (the workgroup
// build of small batches); the lanes build packs the records while it writes the table.
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_pack_units(UnitPlan plan, int64_t B, int32_t S, const float* __restrict__ steps,
                                                    float* __restrict__ units, float* __restrict__ extras) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int U = plan.n_units;
  if (i >= B * U) return;
  const int64_t b = i / U;
  const int u = (int)(i - b * U);
  const int slot = plan.slot[u];
  float pre[4] = {1.f, 0.f, 0.f, 1.f};
  if (plan.pair[u]) {  // the run's slot holds its rows 4 and 5 in front (k_build)
    const float* run = steps + (b * S + slot - 1) * LYNX_STEP_STRIDE;
    pre[0] = run[4]; pre[1] = run[5]; pre[2] = run[7 + 4]; pre[3] = run[7 + 5];
  }
  pack_unit_record<float>(steps + (b * S + slot) * LYNX_STEP_STRIDE, pre, slot, plan.cls[u], plan.pair[u],
                          units + i * kUnitStride, extras + i * kUnitExtraStride);
}

// ---------------------------------------------------------------------------------------
// the step loop's pieces
// ---------------------------------------------------------------------------------------

// The cavity's non-linear kick on a pair of particles: the expressions of apply_step_pair (lynx_device.hpp), which
// are device_cavity_kick's (cavity.py:141-161, 219-226), operation for operation.
// COMPLETE = false: the difference of cosines from cos_difference_small alone, which holds for |d| <= 0.25 rad (d = -s
// beta0 k: 9 mm of s at 1.3 GHz), and `widest` keeps the largest |d| the lane has seen -- the streaming kernel looks
// at it once per tile and does a tile that went beyond again with COMPLETE = true, like a tile that met a non-finite
// value.  (The complete form's branch and its second cosine in the unit loop cost four registers: see cos_difference.)
template <bool COMPLETE>
__device__ __forceinline__ void unit_kick(const float (&coef)[8], float sphi, lynx_f32x2 s_in, lynx_f32x2 d_in, lynx_f32x2& o4,
                                          lynx_f32x2& o5, float& widest) {
  const lynx_f32x2 d = -1.0f * s_in * coef[LYNX_C_BK];
  lynx_f32x2 dcos;
  if constexpr (COMPLETE) {
    dcos = cos_difference(d, coef[LYNX_C_PHI], sphi, coef[LYNX_C_COSPHI]);
  } else {
    dcos = cos_difference_small<lynx_f32x2>(d, sphi, coef[LYNX_C_COSPHI]);
    widest = __builtin_fmaxf(__builtin_fmaxf(__builtin_fabsf(d.x), __builtin_fabsf(d.y)), widest);  // (v_max3_f32)
  }
  kick_outputs<lynx_f32x2>(coef, s_in, d_in, dcos, o4, o5);
}

// one 2-row block with up to four columns: (za, zb) <- rows (ra, rb) applied to the given columns, in ascending
// column order like the dense chain
template <int N>
__device__ __forceinline__ lynx_f32x2 row_dot(const float (&m)[N], const lynx_f32x2 (&v)[N]) {
  lynx_f32x2 acc = v[0] * m[0];
#pragma unroll
  for (int k = 1; k < N; ++k) acc = pk_fma(v[k], m[k], acc);
  return acc;
}

// true if any of the 14 values of the pair is NaN or Inf: 0 * x is +-0 for finite x and NaN otherwise
__device__ __forceinline__ bool pair_not_finite(const lynx_f32x2 (&z)[7]) {
  lynx_f32x2 acc = z[0] * 0.0f;
#pragma unroll
  for (int k = 1; k < 7; ++k) acc = pk_fma(z[k], 0.0f, acc);
  return (acc.x != acc.x) || (acc.y != acc.y);
}

// particle `i` of a sample whose first particle sits at `base`: a wave-uniform base and a 32-bit byte offset per lane --
// the address form global loads take without any 64-bit vector arithmetic (the host keeps samples of 4 GiB and more
// away from this kernel)
__device__ __forceinline__ const float* particle_at(const float* base, uint32_t i) {
  return reinterpret_cast<const float*>(reinterpret_cast<const char*>(base) + i * 28u);
}
__device__ __forceinline__ float* particle_at(float* base, uint32_t i) {
  return reinterpret_cast<float*>(reinterpret_cast<char*>(base) + i * 28u);
}

// A wave-uniform value the compiler must keep in a VECTOR register: the step loop needs every scalar register there is,
// and the six coordinates of the moments' reference point, left to the compiler, lived in scalar registers that were
// spilled to vector-register lanes and read back one v_readlane at a time wherever a particle entered the sums (12 per
// tile) -- a copy per lane costs six registers and nothing else.
__device__ __forceinline__ float in_vector_register(float uniform) {
  float v;
  asm volatile("v_mov_b32 %0, %1" : "=v"(v) : "s"(uniform));
  return v;
}

template <int MOM, bool FULL, int PAIRS> constexpr int units_waves_per_simd() {
  return PAIRS == 1 ? ((!FULL && (MOM == 0 || MOM == 3)) ? 5 : 1) : ((!FULL && (MOM == 0 || MOM == 3)) ? 4 : 1);
}

// half a unit record (16 scalars), fetched with wave-uniform loads: they land in scalar registers and are the
// broadcast operand of the packed multiply-adds
struct UnitHalf {
  float v[16];
};
__device__ __forceinline__ void unit_fetch(const float* __restrict__ half, UnitHalf& r) {
#pragma unroll
  for (int k = 0; k < 16; ++k) r.v[k] = uniform_value(half[k]);
}

// the linear map of a unit of class CLS on a pair of particles; `ex`: the class-D extras of this unit (fetched here:
// that class pays a second round trip)
template <int CLS>
__device__ __forceinline__ void unit_linear(const UnitHalf& map, const float* __restrict__ ex, lynx_f32x2 (&z)[7]) {
  const float* m = map.v;
  if constexpr (CLS == kClassU) {
    {
      const lynx_f32x2 v[3] = {z[0], z[1], z[6]};
      const float r0[3] = {m[0], m[1], m[2]}, r1[3] = {m[3], m[4], m[5]};
      const lynx_f32x2 o0 = row_dot<3>(r0, v), o1 = row_dot<3>(r1, v);
      z[0] = o0;
      z[1] = o1;
    }
    {
      const lynx_f32x2 v[3] = {z[2], z[3], z[6]};
      const float r2[3] = {m[6], m[7], m[8]}, r3[3] = {m[9], m[10], m[11]};
      const lynx_f32x2 o2 = row_dot<3>(r2, v), o3 = row_dot<3>(r3, v);
      z[2] = o2;
      z[3] = o3;
    }
    {
      const lynx_f32x2 v[2] = {z[4], z[5]};
      const float r4[2] = {m[12], m[13]}, r5[2] = {m[14], m[15]};
      const lynx_f32x2 o4 = row_dot<2>(r4, v), o5 = row_dot<2>(r5, v);
      z[4] = o4;
      z[5] = o5;
    }
  } else {
    static_assert(CLS == kClassD, "dense programs take units_dense_program");
    float md[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) md[k] = uniform_value(ex[k]);
    const lynx_f32x2 vx[4] = {z[0], z[1], z[5], z[6]};
    const lynx_f32x2 vs[5] = {z[0], z[1], z[4], z[5], z[6]};
    const float r0[4] = {m[0], m[1], md[0], m[2]}, r1[4] = {m[3], m[4], md[1], m[5]};
    const float r4[5] = {md[2], md[3], m[12], m[13], md[4]}, r5[5] = {md[5], md[6], m[14], m[15], md[7]};
    const lynx_f32x2 o0 = row_dot<4>(r0, vx), o1 = row_dot<4>(r1, vx);
    const lynx_f32x2 o4 = row_dot<5>(r4, vs), o5 = row_dot<5>(r5, vs);
    {
      const lynx_f32x2 v[3] = {z[2], z[3], z[6]};
      const float r2[3] = {m[6], m[7], m[8]}, r3[3] = {m[9], m[10], m[11]};
      const lynx_f32x2 o2 = row_dot<3>(r2, v), o3 = row_dot<3>(r3, v);
      z[2] = o2;
      z[3] = o3;
    }
    z[0] = o0;
    z[1] = o1;
    z[4] = o4;
    z[5] = o5;
  }
}

// the non-linear part of a unit, if it has one; (s_own, d_own): the s and delta that entered the unit
template <bool COMPLETE>
__device__ __forceinline__ void unit_nonlinear(const UnitHalf& kick, int bits, lynx_f32x2 s_own, lynx_f32x2 d_own,
                                               lynx_f32x2 (&z)[7], float& widest) {
  if (bits & kUnitKick) {  // uniform
    lynx_f32x2 s_in = s_own, d_in = d_own;
    if (bits & kUnitInverse) {
      const float* inv = kick.v + kUnitInv;
      s_in = pk_fma(z[5], inv[1], z[4] * inv[0]);
      d_in = pk_fma(z[5], inv[3], z[4] * inv[2]);
    }
    float coef[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) coef[k] = kick.v[kUnitCoef + k];
    unit_kick<COMPLETE>(coef, kick.v[kUnitSinPhi], s_in, d_in, z[4], z[5], widest);
  }
}

// the linear map of a unit in its dense form, from the step table's 7x7: one row of scalars at a time (this is the
// rarely taken path -- a sample whose map lacks the structure, or a tile with a non-finite particle -- and must not
// cost the structured loop its registers).  Same operations in the same order as apply_step_pair's.
__device__ __forceinline__ void unit_linear_dense(const float* __restrict__ tab, lynx_f32x2 (&z)[7]) {
  lynx_f32x2 o[7];
#pragma unroll
  for (int i = 0; i < 7; ++i) {
    float row[7];
#pragma unroll
    for (int j = 0; j < 7; ++j) row[j] = uniform_value(tab[i * 7 + j]);
    lynx_f32x2 acc = z[0] * row[0];
#pragma unroll
    for (int j = 1; j < 7; ++j) acc = pk_fma(z[j], row[j], acc);
    o[i] = acc;
  }
#pragma unroll
  for (int i = 0; i < 7; ++i) z[i] = o[i];
}

// The program on the PAIRS pairs of particles a lane holds.  `s_units` / `s_extras`: the sample's records;
// `all_dense`: every unit in its dense form.  One record fetch per unit serves all pairs.  DENSE_ONLY: the linear maps in
// their dense form whatever the records say (what k_track_unit_pairs keeps for samples that are not of its form).
template <int PAIRS, bool DENSE_ONLY>
__device__ __forceinline__ void units_program(int U, const float* __restrict__ s_units, const float* __restrict__ s_extras,
                                              const float* __restrict__ g_steps, bool all_dense, lynx_f32x2 (&z)[PAIRS][7],
                                              float& widest) {
#pragma unroll 1
  for (int u = 0; u < U; ++u) {
    const float* rec = s_units + u * kUnitStride;
    UnitHalf kick, map;
    unit_fetch(rec + kUnitMap, map);
    unit_fetch(rec, kick);
    const int bits = __builtin_amdgcn_readfirstlane(__float_as_int(kick.v[kUnitDesc]));  // wave-uniform: scalar branches
    const int cls = (DENSE_ONLY || all_dense) ? (int)kClassDense : ((bits >> kUnitClassShift) & 3);
    lynx_f32x2 s_own[PAIRS], d_own[PAIRS];
#pragma unroll
    for (int p = 0; p < PAIRS; ++p) {
      s_own[p] = z[p][4];
      d_own[p] = z[p][5];
    }
    if (bits & kUnitRows) {  // uniform and rare: what enters the cavity from the run's rows 4 and 5 (LYNX_DESC_ILL)
      const int slot = __builtin_amdgcn_readfirstlane(__float_as_int(kick.v[kUnitSlot]));
      float pre[14];
#pragma unroll
      for (int q = 0; q < 14; ++q) pre[q] = uniform_value(g_steps[(slot - 1) * LYNX_STEP_STRIDE + q]);
#pragma unroll
      for (int p = 0; p < PAIRS; ++p) merged_pair_entry(pre, z[p], s_own[p], d_own[p]);
    }
    if (!DENSE_ONLY && cls == kClassU) {
#pragma unroll
      for (int p = 0; p < PAIRS; ++p) unit_linear<kClassU>(map, nullptr, z[p]);
    } else if (!DENSE_ONLY && cls == kClassD) {
#pragma unroll
      for (int p = 0; p < PAIRS; ++p) unit_linear<kClassD>(map, s_extras + u * kUnitExtraStride, z[p]);
    } else {
      const int slot = __builtin_amdgcn_readfirstlane(__float_as_int(kick.v[kUnitSlot]));
#pragma unroll
      for (int p = 0; p < PAIRS; ++p) unit_linear_dense(g_steps + slot * LYNX_STEP_STRIDE, z[p]);
    }
    if (all_dense) {  // (uniform; also the pass that does a tile again whose kicks went beyond cos_difference_small's range)
#pragma unroll
      for (int p = 0; p < PAIRS; ++p) unit_nonlinear<true>(kick, bits, s_own[p], d_own[p], z[p], widest);
    } else {
#pragma unroll
      for (int p = 0; p < PAIRS; ++p) unit_nonlinear<false>(kick, bits, s_own[p], d_own[p], z[p], widest);
    }
  }
}

// The same program for a sample ALL of whose units are merged [run, cavity] pairs of class U with an active cavity driven
// by the entry inverse -- BASELINE config 5's [Drift, misaligned Quadrupole, Drift, Cavity] cells: the same functions in
// the same order as units_program, so the same bits, without the descriptor's scalar branches around them (a third of
// the generic loop's scalar instructions, and the register copies where its paths meet).
constexpr int kUnitFormMask = kUnitKick | kUnitInverse | kUnitRows | (3 << kUnitClassShift);
constexpr int kUnitFormPairU = kUnitKick | kUnitInverse | ((int)kClassU << kUnitClassShift);
template <int PAIRS>
__device__ __forceinline__ void units_program_pairs_u(int U, const float* __restrict__ s_units, lynx_f32x2 (&z)[PAIRS][7],
                                                      float& widest) {
#pragma unroll 1
  for (int u = 0; u < U; ++u) {
    const float* rec = s_units + u * kUnitStride;
    UnitHalf kick, map;
    unit_fetch(rec + kUnitMap, map);
    unit_fetch(rec, kick);
#pragma unroll
    for (int p = 0; p < PAIRS; ++p) {
      const lynx_f32x2 s_own = z[p][4], d_own = z[p][5];
      unit_linear<kClassU>(map, nullptr, z[p]);
      unit_nonlinear<false>(kick, kUnitFormPairU, s_own, d_own, z[p], widest);
    }
  }
}

// true if every unit of the sample has the form units_program_pairs_u is written for (wave-uniform: one descriptor per lane)
__device__ __forceinline__ bool units_all_pairs_u(int U, const float* __restrict__ s_units) {
  const int lane = threadIdx.x & 63;
  const int form = lane < U ? (__float_as_int(s_units[lane * kUnitStride + kUnitDesc]) & kUnitFormMask) : kUnitFormPairU;
  return __builtin_amdgcn_ballot_w64(form != kUnitFormPairU) == 0;
}

// ---------------------------------------------------------------------------------------
// k_track_units: grid.x = B * chunks, 256 threads, 2 PAIRS particles per lane (tid + 256 k of a tile of 512 PAIRS).
// ---------------------------------------------------------------------------------------
// (at most 88 registers per lane where the compiler would take 93 for five waves per SIMD: five of these waves then leave
// 72 of a SIMD's 512 registers free, and ONE of them retiring makes room for a wave of the next call's build -- 136
// registers.  At 96 the build waited for two: k_build_pieces 97 -> 357 us underneath BASELINE config 5's streaming
// kernel, 25 us on every step.)
template <int MOM, bool FULL, int PAIRS, bool PAIR_FORM>
__device__ __forceinline__ void track_units_body(
    const TrackArgs& a, int32_t U, int32_t S, const float* p_in, float* p_out, float* __restrict__ energy_out,
    const float* __restrict__ steps_in, const float* __restrict__ units_in, const float* __restrict__ extras_in,
    double* __restrict__ partials) {
  using T = float;
  constexpr int UNROLL = 2 * PAIRS;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  announce_tail(a);
  const int tid = threadIdx.x;
  // sample and chunk of this workgroup, in scalar registers (the division runs on the vector unit): everything that
  // is derived from them -- the record addresses of the step loop above all -- then stays scalar
  const int64_t b = __builtin_amdgcn_readfirstlane((int)(blockIdx.x / a.chunks));
  const int chunk = __builtin_amdgcn_readfirstlane((int)(blockIdx.x % a.chunks));
  const uint32_t end = (uint32_t)a.n_particles;  // < 2^32 / 28 (launch_units)
  constexpr uint32_t kTile = (uint32_t)kTrackThreads * UNROLL;
  const T* src = p_in + b * a.in_stride;
  T* dst = p_out + b * (int64_t)a.n_particles * 7;
  const float* g_steps = steps_in + b * (int64_t)S * LYNX_STEP_STRIDE;
  const float* g_units = units_in + b * (int64_t)U * kUnitStride;
  const float* g_extras = extras_in + b * (int64_t)U * kUnitExtraStride;

  // (dealing a sample's tiles out so that the grid is a whole number of rounds of resident workgroups was measured and
  // lost: C4 0.995 -> 1.07 ms, the 128-sample shard 0.155 -> 0.162)
  const uint32_t tile0 = (uint32_t)chunk * (uint32_t)a.tiles_per_wg;
  const int my_tiles = a.tiles_per_wg;
  // the first tile's loads go out before anything else
  T zn[UNROLL][7];
  {
    const uint32_t i0 = tile0 * kTile + tid;
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) {
      const uint32_t i = i0 + (uint32_t)u * kTrackThreads;
      load_particle(particle_at(src, i < end ? i : (i0 < end ? i0 : 0u)), zn[u]);
    }
  }
  if (energy_out && chunk == 0 && tid == 0 && S > 0) energy_out[b] = g_steps[(S - 1) * LYNX_STEP_STRIDE + LYNX_ENERGY_OFFSET];

  const bool pairs_u = PAIR_FORM && units_all_pairs_u(U, g_units);
  LaneSums<T, MOM, FULL> sums;
  sums.init();
  T shift[6];
#pragma unroll
  for (int i = 0; i < 6; ++i) shift[i] = T(0);

  for (int it = 0; it < my_tiles; ++it) {
    const uint32_t tile = tile0 + (uint32_t)it;
    const uint32_t i0 = tile * kTile + tid;
    // (uniform: the whole workgroup leaves together.  A per-lane `i0 >= end` here makes the loop's control flow
    // divergent, and the compiler then keeps every loop-carried scalar -- record registers, addresses -- in VECTOR
    // registers; lanes beyond the end of the sample ride along on a clamped address and are masked where it counts)
    if (tile * kTile >= end) break;
    sums.begin_iteration();
    lynx_f32x2 z[PAIRS][7];
#pragma unroll
    for (int p = 0; p < PAIRS; ++p)
#pragma unroll
      for (int c = 0; c < 7; ++c) {
        z[p][c].x = zn[2 * p][c];
        z[p][c].y = zn[2 * p + 1][c];
      }
    if (it + 1 < my_tiles) {  // prefetch the next tile of this workgroup
      const uint32_t j0 = (tile + 1) * kTile + tid;
#pragma unroll
      for (int u = 0; u < UNROLL; ++u) {
        const uint32_t j = j0 + (uint32_t)u * kTrackThreads;
        load_particle(particle_at(src, j < end ? j : 0u), zn[u]);
      }
    }
    // a non-finite particle anywhere in the wave: the dense chain, whose zero entries spread it like the reference's
    auto any_not_finite = [&]() {
      bool bad = false;
#pragma unroll
      for (int p = 0; p < PAIRS; ++p) bad = bad || pair_not_finite(z[p]);
      return __builtin_amdgcn_ballot_w64(bad) != 0;  // wave-uniform
    };
    // (only what comes OUT is looked at: a non-finite coordinate going in leaves its own block of the first unit
    // non-finite -- m * inf is inf or, for m = 0, NaN -- and nothing on the way makes it finite again, the kick's
    // cosine and squares included; so the check in front of the units was 10 instructions per tile for nothing)
    bool all_dense = false;
    for (;;) {
      float widest = 0.f;  // the largest |d| of a kick on this lane (unit_kick)
      if constexpr (PAIR_FORM) {
        if (__builtin_expect(pairs_u && !all_dense, 1)) units_program_pairs_u<PAIRS>(U, g_units, z, widest);
        else units_program<PAIRS, true>(U, g_units, g_extras, g_steps, all_dense, z, widest);
      } else {
        units_program<PAIRS, false>(U, g_units, g_extras, g_steps, all_dense, z, widest);
      }
      // what came out: a value that overflowed on the way (or met a non-finite coefficient) would have spread through
      // the dense chain's zero entries -- then the tile is done again, densely; so is a tile one of whose kicks left
      // the range of the short form of the difference of cosines (a NaN there is among the non-finite ones)
      if (all_dense || !(any_not_finite() || __builtin_amdgcn_ballot_w64(widest > kCosDifferenceSmall) != 0)) break;
      all_dense = true;
      // reload the tile; the outgoing beam has not been written yet, so in-place tracking is safe
#pragma unroll
      for (int u = 0; u < UNROLL; ++u) {
        const uint32_t i = i0 + (uint32_t)u * kTrackThreads;
        T t[7];
        load_particle(particle_at(src, i < end ? i : 0u), t);
#pragma unroll
        for (int c = 0; c < 7; ++c) {
          if (u % 2 == 0) z[u / 2][c].x = t[c];
          else z[u / 2][c].y = t[c];
        }
      }
    }

    T zo[UNROLL][7];
#pragma unroll
    for (int p = 0; p < PAIRS; ++p)
#pragma unroll
      for (int c = 0; c < 7; ++c) {
        zo[2 * p][c] = z[p][c].x;
        zo[2 * p + 1][c] = z[p][c].y;
      }
    if (MOM && it == 0) {
#pragma unroll
      for (int k = 0; k < 6; ++k) shift[k] = PAIR_FORM ? wave_first(zo[0][k]) : in_vector_register(wave_first(zo[0][k]));
    }
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) {
      const uint32_t i = i0 + (uint32_t)u * kTrackThreads;
      if (i < end) {
        if (a.store) store_particle(particle_at(dst, i), zo[u]);
        if (MOM) sums.add(zo[u], shift);
      }
    }
    sums.end_iteration();
  }
  if (MOM) workgroup_moment_record<T, MOM, FULL>(sums, shift, smem_raw, partials + ((int64_t)b * a.chunks + chunk) * kPartialStride);
}

template <int MOM, bool FULL, int PAIRS>
__global__ __launch_bounds__(kTrackThreads, (units_waves_per_simd<MOM, FULL, PAIRS>())) void k_track_units(
    TrackArgs a, int32_t U, int32_t S, const float* p_in, float* p_out, float* __restrict__ energy_out,
    const float* __restrict__ steps_in, const float* __restrict__ units_in, const float* __restrict__ extras_in,
    double* __restrict__ partials) {
  track_units_body<MOM, FULL, PAIRS, false>(a, U, S, p_in, p_out, energy_out, steps_in, units_in, extras_in, partials);
}

// The kernel for a lattice whose plan PROPOSES a merged [run, cavity] pair of class U for every unit: samples that keep the
// promise (units_all_pairs_u looks at the records) walk units_program_pairs_u, the others -- a cavity without voltage, a
// map that failed the class check -- every unit in its dense form.  Same bits as k_track_units either way.
// Five waves per SIMD and no more (the hot loop alone would fit seven at 72 registers: config 5 then ran 0.54-0.555 ms a step
// instead of 0.52-0.53, the next call's build waiting for registers underneath), at most 88 registers like k_track_units --
// which the dense loop, rarely walked, just fits once the moments' reference point lives in scalar registers here
// (LYNX_UNIT_PAIRS; instantiated for the two forms that fit: no moments, and moments of mode 3 without the full covariance).
template <int MOM, bool FULL>
__global__ __launch_bounds__(kTrackThreads) __attribute__((amdgpu_waves_per_eu(5, 5))) void k_track_unit_pairs(
    TrackArgs a, int32_t U, int32_t S, const float* p_in, float* p_out, float* __restrict__ energy_out,
    const float* __restrict__ steps_in, const float* __restrict__ units_in, const float* __restrict__ extras_in,
    double* __restrict__ partials) {
  track_units_body<MOM, FULL, 1, true>(a, U, S, p_in, p_out, energy_out, steps_in, units_in, extras_in, partials);
}

// LDS of k_track_units: the moment slab (TrackArgs.lds_scratch_bytes)
inline size_t units_lds_bytes(size_t slab_bytes, int /*n_units*/) { return slab_bytes; }

}  // namespace lynx
