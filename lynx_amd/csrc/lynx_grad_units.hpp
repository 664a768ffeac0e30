// Reverse pass of multi-step float32 programs whose units are all of class U (lynx_units.hpp): the structured twin of
// k_track_bwd.  Included only by lynx_hip.hip, after lynx_grad.hpp and lynx_units.hpp.
//
// A class-U unit (rows 0,1: columns {0,1,6}; rows 2,3: {2,3,6}; rows 4,5: {4,5}; row 6 = e6) moves the two transverse
// planes by AFFINE 2x3 maps that see neither each other nor (s, delta), and (s, delta) by a 2x2 block plus the cavity's
// kick, which sees s and delta only (cavity.py:141-161).  The reverse pass of a program of such units falls apart:
//
//   transverse planes   With X_u the x-block of unit u (homogeneous 3x3, last row e), the state entering unit u is
//                       C_u (x0, x0', z6),  C_u = X_{u-1} ... X_0,  and the cotangent leaving it is  A_u^T (xbar_f, xbar'_f),
//                       A_u = the linear 2x2 of X_{U-1} ... X_{u+1}.  So every unit's cotangent block
//                         Tbar_x(u) = sum_n obar_n(u) (x) z_in,n(u) = A_u^T S_x C_u^T,   S_x = sum_n (xbar_f, xbar'_f)_n (x) (x0, x0', z6)_n
//                       comes from ONE 2x3 sum over the particles -- 12 multiply-adds per particle for both planes and
//                       the whole lattice, where walking the units costs 12 products, their sums over the wave and a 2x3
//                       transposed application PER UNIT.  The per-unit blocks are formed afterwards, per sample, by
//                       k_finish_tbar_units (float64).
//   (s, delta)          stays a walk over the units, but of two components: forward with the kick, parking the (s, delta)
//                       that enters every unit (16 registers per pair of particles for 8 units: nothing is recomputed,
//                       nothing goes to scratch memory), backward with the kick's cotangents (lynx_grad.hpp:
//                       kick_cotangents) and 4 (+ 4 for a merged pair's entry block) (+ 8 coefficient) sums per unit,
//                       finished over the wave by halving butterflies (sum_eight_over_wave) -- no exchange buffer.
//
// Round 3's form of this kernel applied every unit in full three times over (forward, recomputation inside a group of
// four, transposed) and summed 16 + 4 products per unit through a 14-row LDS exchange buffer: 2834 vector instructions
// per wave and tile of 128 particles, 912 bytes of scratch per lane, 3 waves per SIMD, 2.37 ms on BASELINE config 5.
//
// The partial sums land in the [B][chunks][S][64] layout k_track_bwd writes, so k_reduce_tbar and k_build_bwd are
// shared: the (s, delta) block, entry rows and coefficient cotangents of every unit where k_track_bwd puts them, S_x
// and S_y in the transverse positions of unit 0's slot until k_finish_tbar_units has replaced them.  Which kernel takes
// a SAMPLE is decided from its unit records (k_pack_units / k_emit_steps checked the numbers): all units class U -> this
// one, anything else -> k_track_bwd; both are launched, a workgroup of the other kind leaves at once.  Same mathematics
// as the dense pass; sums over particles are associated differently.
#pragma once

#include "lynx_grad.hpp"
#include "lynx_units.hpp"

namespace lynx {

constexpr int kBwdUnitsMax = 16;  // units whose entering (s, delta) a lane parks in registers (BASELINE config 5: 8)

// all units of the sample's program have class U, and there are few enough of them (wave-uniform)
__device__ __forceinline__ bool sample_is_class_u(const float* __restrict__ g_units, int U) {
  bool all = U <= kBwdUnitsMax;
  for (int u = 0; u < U; ++u) {
    const int bits = __float_as_int(uniform_value(g_units[u * kUnitStride + kUnitDesc]));
    all = all && ((bits >> kUnitClassShift) & 3) == kClassU;
  }
  return all;
}

// what the (s, delta) walk needs of a unit: descriptor bits, the cavity's coefficients and sin(phi), the (s, delta) block
// of the unit's map and -- merged pair -- of the run's (which gives the s and delta that enter the cavity)
struct UnitSD {
  int bits;
  float coef[8], sphi, m44, m45, m54, m55, pre[4];
};
__device__ __forceinline__ void unit_sd_fetch(const float* __restrict__ g_units, const float* __restrict__ g_extras, int u,
                                              UnitSD& r) {
  const float* rec = g_units + u * kUnitStride;
  r.bits = __builtin_amdgcn_readfirstlane(__float_as_int(uniform_value(rec[kUnitDesc])));
  r.sphi = uniform_value(rec[kUnitSinPhi]);
#pragma unroll
  for (int k = 0; k < 8; ++k) r.coef[k] = uniform_value(rec[kUnitCoef + k]);
  r.m44 = uniform_value(rec[kUnitMap + 12]);
  r.m45 = uniform_value(rec[kUnitMap + 13]);
  r.m54 = uniform_value(rec[kUnitMap + 14]);
  r.m55 = uniform_value(rec[kUnitMap + 15]);
#pragma unroll
  for (int k = 0; k < 4; ++k) r.pre[k] = uniform_value(g_extras[u * kUnitExtraStride + kUnitPre + k]);
}

// Where a sample's S_x / S_y live until k_finish_tbar_units has turned them into the units' transverse blocks: the
// transverse positions of its first unit's slot -- [i][j] at i * 7 + j; i in {0, 1}: j in {0, 1, 6}; i in {2, 3}: j in {2, 3, 6}
__host__ __device__ constexpr int transverse_slot(int k /* 0..11 */) {
  constexpr int at[12] = {0 * 7 + 0, 0 * 7 + 1, 0 * 7 + 6, 1 * 7 + 0, 1 * 7 + 1, 1 * 7 + 6,
                          2 * 7 + 2, 2 * 7 + 3, 2 * 7 + 6, 3 * 7 + 2, 3 * 7 + 3, 3 * 7 + 6};
  return at[k];
}

// MAXU: how many units' entering (s, delta) a lane parks in registers -- 8 (BASELINE config 5: 32 registers, four waves
// per SIMD without scratch) or kBwdUnitsMax
// (three waves per SIMD: 168 registers and 32 bytes of scratch; compiled for four -- 128 registers -- it spills 192
// bytes per lane into the tile loop and BASELINE config 5's forward + reverse step takes 3.05 ms instead of 2.46)
#ifndef LYNX_BWD_UNITS_WAVES
#define LYNX_BWD_UNITS_WAVES 3
#endif
template <int MAXU>
__global__ __launch_bounds__(kTrackThreads, (MAXU <= 8 ? LYNX_BWD_UNITS_WAVES : 3)) void k_track_bwd_units(
    BwdArgs a, int32_t S, const float* __restrict__ p_in, const float* __restrict__ units, const float* __restrict__ extras,
    const double* __restrict__ moments_fwd, const double* __restrict__ grad_moments,
    float* __restrict__ partials /* [B][chunks][S][64] */, float* __restrict__ grad_p /* [B][N][7] or null */) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  using T = float;
  using Z = lynx_f32x2;
  constexpr int W = 2;
  const int U = a.n_units;
  T* s_acc = reinterpret_cast<T*>(smem_raw);  // [4][S][64]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int64_t b = __builtin_amdgcn_readfirstlane((int)(blockIdx.x / a.chunks));
  const int chunk = __builtin_amdgcn_readfirstlane((int)(blockIdx.x % a.chunks));
  const int64_t N = a.n_particles;
  const float* g_units = units + b * (int64_t)U * kUnitStride;
  const float* g_extras = extras + b * (int64_t)U * kUnitExtraStride;
  if (!sample_is_class_u(g_units, U) || (a.leave_odd && (b & 1))) return;  // this sample belongs to k_track_bwd

  T* acc = s_acc + wave * (S * 64);
  for (int s = 0; s < S; ++s) acc[s * 64 + lane] = T(0);

  // The transverse planes of the WHOLE program: (x, x') -> X (x, x', z6), (y, y') -> Y (y, y', z6), the product of the
  // units' 2x3 blocks (wave-uniform arithmetic, once per workgroup)
  T X[6] = {1.f, 0.f, 0.f, 0.f, 1.f, 0.f}, Y[6] = {1.f, 0.f, 0.f, 0.f, 1.f, 0.f};
  for (int u = 0; u < U; ++u) {
    T m[12];
#pragma unroll
    for (int k = 0; k < 12; ++k) m[k] = uniform_value(g_units[u * kUnitStride + kUnitMap + k]);
    const auto left = [](const T* q /* 2x3 of the unit */, T (&P)[6]) {  // P <- q . P (homogeneous)
      const T p0 = P[0], p1 = P[1], p2 = P[2], p3 = P[3], p4 = P[4], p5 = P[5];
      P[0] = fmaf(q[1], p3, q[0] * p0);
      P[1] = fmaf(q[1], p4, q[0] * p1);
      P[2] = fmaf(q[1], p5, fmaf(q[0], p2, q[2]));
      P[3] = fmaf(q[4], p3, q[3] * p0);
      P[4] = fmaf(q[4], p4, q[3] * p1);
      P[5] = fmaf(q[4], p5, fmaf(q[3], p2, q[5]));
    };
    left(m, X);
    left(m + 6, Y);
  }
#pragma unroll
  for (int k = 0; k < 6; ++k) {
    X[k] = uniform_value(X[k]);
    Y[k] = uniform_value(Y[k]);
  }

  const double* rec = moments_fwd + b * LYNX_MOMENT_STRIDE;
  const double* gm = grad_moments + b * LYNX_MOMENT_STRIDE;
  const T* src = p_in + b * N * 7;
  constexpr int64_t kTile = (int64_t)kTrackThreads * W;
  Z sx[6], sy[6];  // this lane's share of S_x and S_y, over all its tiles
#pragma unroll
  for (int k = 0; k < 6; ++k) sx[k] = sy[k] = Z(T(0));
  // ... and of the units' sums: per unit the eight of the (s, delta) block and entry rows, and the eight coefficient
  // cotangents, each folded over the lane's group of eight after every tile (lane l keeps value l & 7) and finished over
  // the wave once, behind the last tile -- the tile loop then touches neither LDS nor another lane group
  T fold_sd[MAXU], fold_cc[MAXU];
#pragma unroll
  for (int u = 0; u < MAXU; ++u) fold_sd[u] = fold_cc[u] = T(0);

  for (int it = 0; it < a.tiles_per_wg; ++it) {
    const int64_t base = ((int64_t)chunk * a.tiles_per_wg + it) * kTile;
    if (base >= N) break;  // uniform
    bool live[W];
    Z z0[7];
#pragma unroll
    for (int w = 0; w < W; ++w) {
      const int64_t i = base + (int64_t)w * kTrackThreads + tid;
      live[w] = i < N;
      T zw[7];
      load_particle(src + (live[w] ? i : 0) * 7, zw);
#pragma unroll
      for (int c = 0; c < 7; ++c) zset(z0[c], w, zw[c]);
    }

    // the outgoing particle: transverse planes in one step, (s, delta) through the units; what enters every unit is kept
    Z zf[6];
    zf[0] = pk_fma(z0[6], X[2], pk_fma(z0[1], X[1], z0[0] * X[0]));
    zf[1] = pk_fma(z0[6], X[5], pk_fma(z0[1], X[4], z0[0] * X[3]));
    zf[2] = pk_fma(z0[6], Y[2], pk_fma(z0[3], Y[1], z0[2] * Y[0]));
    zf[3] = pk_fma(z0[6], Y[5], pk_fma(z0[3], Y[4], z0[2] * Y[3]));
    Z park[MAXU][2];
    {
      Z z4 = z0[4], z5 = z0[5];
#pragma unroll
      for (int u = 0; u < MAXU; ++u) {
        if (u < U) {  // uniform
          UnitSD r;
          unit_sd_fetch(g_units, g_extras, u, r);
          park[u][0] = z4;
          park[u][1] = z5;
          Z s_in = z4, d_in = z5;
          if (r.bits & kUnitPair) {
            s_in = pk_fma(z5, r.pre[1], z4 * r.pre[0]);
            d_in = pk_fma(z5, r.pre[3], z4 * r.pre[2]);
          }
          const Z o4 = pk_fma(z5, r.m45, z4 * r.m44), o5 = pk_fma(z5, r.m55, z4 * r.m54);
          z4 = o4;
          z5 = o5;
          if (r.bits & kUnitKick) {
            float unused = 0.f;
            unit_kick<true>(r.coef, r.sphi, s_in, d_in, z4, z5, unused);
          }
        }
      }
      zf[4] = z4;
      zf[5] = z5;
    }

    // cotangent of the outgoing particle: (1/N) (mu_bar + G_hat (z - mean)), as in k_track_bwd
    Z zb[7];
    {
      const T inv_n = (T)(1.0 / rec[35]);
      Z d[6];
#pragma unroll
      for (int k = 0; k < 6; ++k) d[k] = zf[k] - (T)rec[k];
#pragma unroll
      for (int k = 0; k < 6; ++k) {
        Z accv = Z((T)gm[k]);
#pragma unroll
        for (int j = 0; j < 6; ++j) {
          const int r = k < j ? k : j, c = k < j ? j : k;
          const T gg = (T)gm[7 + r * 6 - (r * (r - 1)) / 2 + (c - r)];
          accv = zfma(d[j], k == j ? gg + gg : gg, accv);
        }
        zb[k] = accv * inv_n;
      }
      zb[6] = Z((T)gm[6] * inv_n);
#pragma unroll
      for (int w = 0; w < W; ++w)
#pragma unroll
        for (int c = 0; c < 7; ++c) zset(zb[c], w, live[w] ? zget(zb[c], w) : T(0));
    }

    // transverse planes: S_x += (xbar, xbar') (x) (x0, x0', z6), S_y likewise
    sx[0] = zfma(zb[0], z0[0], sx[0]);
    sx[1] = zfma(zb[0], z0[1], sx[1]);
    sx[2] = zfma(zb[0], z0[6], sx[2]);
    sx[3] = zfma(zb[1], z0[0], sx[3]);
    sx[4] = zfma(zb[1], z0[1], sx[4]);
    sx[5] = zfma(zb[1], z0[6], sx[5]);
    sy[0] = zfma(zb[2], z0[2], sy[0]);
    sy[1] = zfma(zb[2], z0[3], sy[1]);
    sy[2] = zfma(zb[2], z0[6], sy[2]);
    sy[3] = zfma(zb[3], z0[2], sy[3]);
    sy[4] = zfma(zb[3], z0[3], sy[4]);
    sy[5] = zfma(zb[3], z0[6], sy[5]);

    // (s, delta) backwards through the units
    Z b4 = zb[4], b5 = zb[5];
#pragma unroll
    for (int u = MAXU - 1; u >= 0; --u) {
      if (u < U) {  // uniform
        UnitSD r;
        unit_sd_fetch(g_units, g_extras, u, r);
        const bool paired = (r.bits & kUnitPair) != 0, kick = (r.bits & kUnitKick) != 0;
        const Z z4 = park[u][0], z5 = park[u][1];
        Z s_in = z4, d_in = z5;
        if (paired) {
          s_in = pk_fma(z5, r.pre[1], z4 * r.pre[0]);
          d_in = pk_fma(z5, r.pre[3], z4 * r.pre[2]);
        }
        Z o5 = b5, dir4 = Z(T(0)), dir5 = Z(T(0)), cc[8];
        if (kick) {
          kick_cotangents<T, Z>(r.coef, r.sphi, r.coef[LYNX_C_COSPHI], s_in, d_in, b4, b5, cc, dir4, dir5);
          o5 = Z(T(0));  // the linear delta was overwritten
        }
        // sums over the wave's 128 particles, eight at a time in one halving butterfly: the unit's (s, delta) block, then
        // the entry rows of s and of delta (merged pair: sums against the state that entered the RUN)
        {
          float each[8];
          each[0] = zhsum(b4 * z4);
          each[1] = zhsum(b4 * z5);
          each[2] = zhsum(o5 * z4);
          each[3] = zhsum(o5 * z5);
          each[4] = zhsum(dir4 * z4);
          each[5] = zhsum(dir4 * z5);
          each[6] = zhsum(dir5 * z4);
          each[7] = zhsum(dir5 * z5);
          fold_sd[u] += sum_eight_over_8_lanes(each, lane);
        }
        if (kick) {  // uniform: the eight coefficient cotangents
          float each[8];
#pragma unroll
          for (int c = 0; c < 8; ++c) each[c] = zhsum(cc[c]);
          fold_cc[u] += sum_eight_over_8_lanes(each, lane);
        }
        // through the unit's (s, delta) block (transposed), then the kick's direct terms -- through the entry block for
        // a merged pair
        const Z n4 = pk_fma(o5, r.m54, b4 * r.m44), n5 = pk_fma(o5, r.m55, b4 * r.m45);
        b4 = n4;
        b5 = n5;
        if (!paired) {
          b4 += dir4;
          b5 += dir5;
        } else if (kick) {
          b4 = pk_fma(dir5, r.pre[2], pk_fma(dir4, r.pre[0], b4));
          b5 = pk_fma(dir5, r.pre[3], pk_fma(dir4, r.pre[1], b5));
        }
      }
    }

    // dL/d(incoming particle): the transverse cotangents through the whole program's blocks (transposed), s and delta
    // from the walk, the constant component's from the affine columns
    if (grad_p != nullptr) {
      Z g[7];
      g[0] = pk_fma(zb[1], X[3], zb[0] * X[0]);
      g[1] = pk_fma(zb[1], X[4], zb[0] * X[1]);
      g[2] = pk_fma(zb[3], Y[3], zb[2] * Y[0]);
      g[3] = pk_fma(zb[3], Y[4], zb[2] * Y[1]);
      g[4] = b4;
      g[5] = b5;
      g[6] = pk_fma(zb[3], Y[5], pk_fma(zb[2], Y[2], pk_fma(zb[1], X[5], pk_fma(zb[0], X[2], zb[6]))));
#pragma unroll
      for (int w = 0; w < W; ++w) {
        const int64_t i = base + (int64_t)w * kTrackThreads + tid;
        if (live[w]) {
          T zw[7];
#pragma unroll
          for (int c = 0; c < 7; ++c) zw[c] = zget(g[c], w);
          store_particle(grad_p + (b * N + i) * 7, zw);
        }
      }
    }
  }

  // the units' sums: finish over the wave; lane c < 8 books value c -- [4][4], [4][5], [5][4], [5][5] of the unit's
  // slot; the entry rows in the RUN's slot, row of s at [0..6], of delta at [7..13] (where k_track_bwd puts them); the
  // coefficient cotangents behind the unit's 49
#pragma unroll
  for (int u = 0; u < MAXU; ++u) {
    if (u < U) {  // uniform
      const int s = a.unit_slot[u];
      const int bits = __builtin_amdgcn_readfirstlane(__float_as_int(uniform_value(g_units[u * kUnitStride + kUnitDesc])));
      const bool entry_rows = (bits & kUnitPair) && (bits & kUnitKick);  // cotangents of the run's (s, delta) block are due
      const T total = finish_over_lane_groups(fold_sd[u]);
      if (lane < 4) acc[s * 64 + (lane < 2 ? 32 : 37) + lane] += total;                              // 32, 33, 39, 40
      else if (lane < 8 && entry_rows) acc[(s - 1) * 64 + (lane < 6 ? 0 : 5) + lane] += total;  // 4, 5, 11, 12
      if (bits & kUnitKick) {
        const T coefs = finish_over_lane_groups(fold_cc[u]);
        if (lane < 8) acc[s * 64 + 49 + lane] += coefs;
      }
    }
  }
  // the workgroup's S_x and S_y: lanes -> wave (two butterflies of eight, the last four slots idle), into unit 0's slot
  {
    const int s0 = a.unit_slot[0];
    float each[8];
#pragma unroll
    for (int k = 0; k < 6; ++k) each[k] = zhsum(sx[k]);
    each[6] = zhsum(sy[0]);
    each[7] = zhsum(sy[1]);
    T total = sum_eight_over_wave(each, lane);
    if (lane < 8) acc[s0 * 64 + transverse_slot(lane)] += total;
#pragma unroll
    for (int k = 0; k < 4; ++k) each[k] = zhsum(sy[2 + k]);
#pragma unroll
    for (int k = 4; k < 8; ++k) each[k] = 0.f;
    total = sum_eight_over_wave(each, lane);
    if (lane < 4) acc[s0 * 64 + transverse_slot(8 + lane)] += total;
  }

  __syncthreads();
  T* out = partials + (((int64_t)b * a.chunks + chunk) * S) * kGradStride;
  for (int idx = tid; idx < S * 64; idx += kTrackThreads) {
    out[idx] = ((s_acc[idx] + s_acc[S * 64 + idx]) + s_acc[2 * S * 64 + idx]) + s_acc[3 * S * 64 + idx];
  }
}

inline size_t bwd_units_lds_bytes(int S) { return (size_t)4 * S * 64 * sizeof(float); }

// ---------------------------------------------------------------------------------------
// k_finish_tbar_units: S_x, S_y of a class-U sample (k_track_bwd_units, reduced over the workgroups by k_reduce_tbar) ->
// the transverse blocks of every unit's cotangent,  Tbar_x(u) = A_u^T S_x C_u^T  (see the top of this file), in float64.
// One wave per sample, lane u = unit u: every lane reads S before any of them writes (unit 0's slot is where S sits).
// A merged pair's entry rows have no transverse part; samples of other classes were walked densely and are left alone.
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void k_finish_tbar_units(BwdArgs a, int32_t S, const float* __restrict__ units,
                                                          float* __restrict__ tbar /* [B][S][64] */) {
  const int64_t b = blockIdx.x;
  const int U = a.n_units;
  const int u = threadIdx.x;
  const float* g_units = units + b * (int64_t)U * kUnitStride;
  // (sample_is_class_u with one descriptor per lane, and the units' transverse entries fetched side by side into LDS: the
  // loops below walked them one global round trip per unit and plane -- 35 us for a kernel of a few hundred operations)
  __shared__ float s_m[kBwdUnitsMax * 12];
  const bool other = u < U && ((__float_as_int(g_units[u * kUnitStride + kUnitDesc]) >> kUnitClassShift) & 3) != kClassU;
  if (U > kBwdUnitsMax || __builtin_amdgcn_ballot_w64(other) != 0 || (a.leave_odd && (b & 1))) return;  // uniform
  for (int i = u; i < U * 12; i += 64) s_m[i] = g_units[(i / 12) * kUnitStride + kUnitMap + i % 12];
  float* row0 = tbar + (b * S + a.unit_slot[0]) * (int64_t)kGradStride;
  double Sx[6], Sy[6];
#pragma unroll
  for (int k = 0; k < 6; ++k) {
    Sx[k] = (double)row0[transverse_slot(k)];
    Sy[k] = (double)row0[transverse_slot(6 + k)];
  }
  double out[12];
#pragma unroll
  for (int k = 0; k < 12; ++k) out[k] = 0.0;
  __syncthreads();  // s_m
  if (u < U) {
    for (int plane = 0; plane < 2; ++plane) {
      const double* Sp = plane ? Sy : Sx;
      // C = prefix (homogeneous 2x3) of the units in front of u, A = linear 2x2 of the units behind it
      double C[6] = {1, 0, 0, 0, 1, 0}, A[4] = {1, 0, 0, 1};
      for (int v = 0; v < U; ++v) {
        if (v == u) continue;
        const float* m = s_m + v * 12 + 6 * plane;
        const double q0 = m[0], q1 = m[1], q2 = m[2], q3 = m[3], q4 = m[4], q5 = m[5];
        if (v < u) {
          const double c0 = C[0], c1 = C[1], c2 = C[2], c3 = C[3], c4 = C[4], c5 = C[5];
          C[0] = q0 * c0 + q1 * c3;
          C[1] = q0 * c1 + q1 * c4;
          C[2] = q0 * c2 + q1 * c5 + q2;
          C[3] = q3 * c0 + q4 * c3;
          C[4] = q3 * c1 + q4 * c4;
          C[5] = q3 * c2 + q4 * c5 + q5;
        } else {
          const double a0 = A[0], a1 = A[1], a2 = A[2], a3 = A[3];
          A[0] = q0 * a0 + q1 * a2;
          A[1] = q0 * a1 + q1 * a3;
          A[2] = q3 * a0 + q4 * a2;
          A[3] = q3 * a1 + q4 * a3;
        }
      }
      // P = A^T S (2x3); Tbar = P Ct^T with Ct the homogeneous 3x3 of C (its last row picks the constant column)
      const double P0 = A[0] * Sp[0] + A[2] * Sp[3], P1 = A[0] * Sp[1] + A[2] * Sp[4], P2 = A[0] * Sp[2] + A[2] * Sp[5];
      const double P3 = A[1] * Sp[0] + A[3] * Sp[3], P4 = A[1] * Sp[1] + A[3] * Sp[4], P5 = A[1] * Sp[2] + A[3] * Sp[5];
      out[6 * plane + 0] = P0 * C[0] + P1 * C[1] + P2 * C[2];
      out[6 * plane + 1] = P0 * C[3] + P1 * C[4] + P2 * C[5];
      out[6 * plane + 2] = P2;
      out[6 * plane + 3] = P3 * C[0] + P4 * C[1] + P5 * C[2];
      out[6 * plane + 4] = P3 * C[3] + P4 * C[4] + P5 * C[5];
      out[6 * plane + 5] = P5;
    }
  }
  __syncthreads();  // one wave: every lane has read S
  if (u < U) {
    float* row = tbar + (b * S + a.unit_slot[u]) * (int64_t)kGradStride;
#pragma unroll
    for (int k = 0; k < 12; ++k) row[transverse_slot(k)] = (float)out[k];
  }
}

}  // namespace lynx
