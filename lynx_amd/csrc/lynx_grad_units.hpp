// Reverse pass of multi-step float32 programs whose units are all of class U (lynx_units.hpp): the structured twin of
// k_track_bwd.  Included only by lynx_hip.hip, after lynx_grad.hpp and lynx_units.hpp.
//
// k_track_bwd treats every unit's map as a dense 7x7 three times over -- forward sweep, recomputation inside a group,
// `M^T z_bar` -- and sums the full 7x7 outer product `z_bar (x) z_in` over the wave's particles through its exchange
// buffer.  For a class-U unit (rows 0,1: columns {0,1,6}; rows 2,3: {2,3,6}; rows 4,5: {4,5}; row 6 = e6) only 16 of
// those 49 cotangents can reach a parameter: every element of the unit is block-structured, so the cotangent of an
// element map's in-pattern entry is built from in-pattern entries of the unit's alone (k_build_bwd multiplies by the
// element maps, which have the same blocks), and `dM/dtheta` vanishes outside the pattern.  This kernel therefore
//   * applies units with 16 packed multiply-adds (unit_linear<kClassU>) in both sweeps,
//   * sums 16 (+ 4 for a merged pair's entry block) particle products per unit instead of 49 (+ 14): lane group g of 8
//     lanes owns one row of cotangents and at most three rows of states,
//   * forms `M^T z_bar` from 16 entries,
// and reads the unit records (128 bytes) instead of the 256-byte step-table rows.  The partial sums land in the same
// [B][chunks][S][64] layout k_track_bwd writes (zeros where nothing is summed), so k_reduce_tbar and k_build_bwd are
// shared.  Which kernel takes a SAMPLE is decided from its unit records (k_pack_units / k_emit_steps checked the
// numbers): all units class U -> this one, anything else -> k_track_bwd; both are launched, a workgroup of the other
// kind leaves at once.  Same mathematics as the dense pass; sums over particles are associated differently.
#pragma once

#include "lynx_grad.hpp"
#include "lynx_units.hpp"

namespace lynx {

// all units of the sample's program have class U (wave-uniform)
__device__ __forceinline__ bool sample_is_class_u(const float* __restrict__ g_units, int U) {
  bool all = true;
  for (int u = 0; u < U; ++u) {
    const int bits = __float_as_int(uniform_value(g_units[u * kUnitStride + kUnitDesc]));
    all = all && ((bits >> kUnitClassShift) & 3) == kClassU;
  }
  return all;
}

// one class-U unit, forward: linear map, then the kick driven by what entered the cavity -- the run's (s, delta) block
// `pre` applied to the unit's incoming state (merged pair), or that state's own s, delta
__device__ __forceinline__ void bwd_unit_forward(const UnitHalf& kick, const UnitHalf& map, const float (&pre)[4],
                                                 lynx_f32x2 (&z)[7]) {
  const int bits = __builtin_amdgcn_readfirstlane(__float_as_int(kick.v[kUnitDesc]));
  lynx_f32x2 s_in = z[4], d_in = z[5];
  if (bits & kUnitPair) {
    s_in = pk_fma(z[5], pre[1], z[4] * pre[0]);
    d_in = pk_fma(z[5], pre[3], z[4] * pre[2]);
  }
  unit_linear<kClassU>(map, nullptr, z);
  if (bits & kUnitKick) {
    float coef[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) coef[k] = kick.v[kUnitCoef + k];
    unit_kick(coef, s_in, d_in, z[4], z[5]);
  }
}

__device__ __forceinline__ void bwd_unit_fetch(const float* __restrict__ g_units, const float* __restrict__ g_extras, int u,
                                               UnitHalf& kick, UnitHalf& map, float (&pre)[4]) {
  unit_fetch(g_units + u * kUnitStride, kick);
  unit_fetch(g_units + u * kUnitStride + kUnitMap, map);
#pragma unroll
  for (int k = 0; k < 4; ++k) pre[k] = uniform_value(g_extras[u * kUnitExtraStride + kUnitPre + k]);
}

constexpr int kBwdUnitsPitch = 160;  // scalars per exchange row (k_track_bwd_units)
constexpr int kBwdUnitsRows = 14;  // exchange rows: 0..5 cotangents of the linear outputs, 6 the entering s', 7..13 the state

__global__ __launch_bounds__(kTrackThreads) void k_track_bwd_units(
    BwdArgs a, int32_t S, const float* __restrict__ p_in, const float* __restrict__ units, const float* __restrict__ extras,
    const double* __restrict__ moments_fwd, const double* __restrict__ grad_moments,
    float* __restrict__ partials /* [B][chunks][S][64] */, float* __restrict__ grad_p /* [B][N][7] or null */) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  using T = float;
  using Z = lynx_f32x2;
  constexpr int K = kBwdGroup;
  constexpr int W = 2;
  using Geo = ExGeom<T, W>;
  using V = lynx_f32x4;
  // Pitch and row order of the exchange buffer, chosen for ds_read_b128's lane groups on gfx950 ({0-3, 12-15, 20-27},
  // {4-11, 16-19, 28-31} and the same + 32; banks (a / 4) mod 64): with 160 scalars per row even and odd rows sit 32
  // banks apart, so the four row groups of a lane group read cotangent rows g .. g+3 without meeting, and with the
  // state rows stored in the order 0, 2, 1, 3, 4, 5, 6 the two rows a lane group reads at once (0 | 2, then 1 | 3) do
  // too.  (ExGeom's 132 with the rows in natural order: 108 LDS cycles per unit and wave where 64 are needed;
  // SQ_LDS_BANK_CONFLICT 23 % of SQ_LDS_IDX_ACTIVE.)
  constexpr int VW = Geo::kVW, P = kBwdUnitsPitch, NP = Geo::kPieces;
  constexpr int kStateRow[7] = {7, 9, 8, 10, 11, 12, 13};
  const int U = a.n_units;
  const int G = (U + K - 1) / K;
  T* s_ex = reinterpret_cast<T*>(smem_raw);       // [4][kBwdUnitsRows][P]
  T* s_acc = s_ex + 4 * kBwdUnitsRows * P;        // [4][S][64]
  Z stack[kBwdMaxGroups * 7];                     // private: state entering unit g*K

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int64_t b = __builtin_amdgcn_readfirstlane((int)(blockIdx.x / a.chunks));
  const int chunk = __builtin_amdgcn_readfirstlane((int)(blockIdx.x % a.chunks));
  const int64_t N = a.n_particles;
  const float* g_units = units + b * (int64_t)U * kUnitStride;
  const float* g_extras = extras + b * (int64_t)U * kUnitExtraStride;
  if (!sample_is_class_u(g_units, U)) return;  // this sample belongs to k_track_bwd

  T* ex = s_ex + wave * (kBwdUnitsRows * P);
  T* acc = s_acc + wave * (S * 64);
  for (int s = 0; s < S; ++s) acc[s * 64 + lane] = T(0);

  const double* rec = moments_fwd + b * LYNX_MOMENT_STRIDE;
  const double* gm = grad_moments + b * LYNX_MOMENT_STRIDE;
  const T* src = p_in + b * N * 7;
  constexpr int64_t kTile = (int64_t)kTrackThreads * W;
  // who sums what: lane group g = lane / 8 owns cotangent row g (0..5 the linear outputs' -- row 5 carries the
  // entering delta's for a kicked pair --, 6 the entering s') and the state rows its pattern pairs it with
  const int g = lane >> 3, kb = lane & 7;
  const int zr0 = g < 2 ? 0 : (g < 4 ? 2 : 4);   // first state row
  const int zr1 = zr0 + 1;                       // second
  const bool third = g < 4;                      // rows 0..3 also meet the constant component (column 6)
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
      UnitHalf kick, map;
      float pre[4];
      bwd_unit_fetch(g_units, g_extras, u, kick, map, pre);
      bwd_unit_forward(kick, map, pre, z);
    }

    // cotangent of the outgoing particle: (1/N) (mu_bar + G_hat (z - mean)), as in k_track_bwd
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
        if (up + 1 < U) {
          UnitHalf kick, map;
          float pre[4];
          bwd_unit_fetch(g_units, g_extras, up, kick, map, pre);
          bwd_unit_forward(kick, map, pre, zz[j]);
        }
      }

      for (int j = K - 1; j >= 0; --j) {
        const int u = grp * K + j;
        if (u >= U) continue;  // uniform
        const int s = a.unit_slot[u];
        Z zin[7];
#pragma unroll
        for (int c = 0; c < 7; ++c) {
          Z v = zz[0][c];
#pragma unroll
          for (int q = 1; q < K; ++q) v = (j == q) ? zz[q][c] : v;
          zin[c] = v;
        }
        UnitHalf kickr, map;
        float pre[4];
        bwd_unit_fetch(g_units, g_extras, u, kickr, map, pre);
        const int bits = __builtin_amdgcn_readfirstlane(__float_as_int(kickr.v[kUnitDesc]));
        const bool paired = (bits & kUnitPair) != 0, kick = (bits & kUnitKick) != 0;
        const bool entry_rows = paired && kick;  // cotangents of the run's (s, delta) block are due
        Z s_in = zin[4], d_in = zin[5];
        if (paired) {
          s_in = pk_fma(zin[5], pre[1], zin[4] * pre[0]);
          d_in = pk_fma(zin[5], pre[3], zin[4] * pre[2]);
        }
        Z olin[7], cc[8], dir4 = Z(T(0)), dir5 = Z(T(0));
#pragma unroll
        for (int c = 0; c < 7; ++c) olin[c] = zb[c];
        if (kick) {
          // o5' = z5 c0 + c1 (cos(a) - c4), a = -z4 c2 + c3 ; o4' = o4 + c5 z5^2 + c6 z4 z5 + c7 z4^2   (k_track_bwd)
          const T* cf = kickr.v + kUnitCoef;
          T sphi, cphi;
          phase_of<T>(cf[LYNX_C_PHI], sphi, cphi);
          kick_cotangents<T, Z>(cf, sphi, cphi, s_in, d_in, zb[4], zb[5], cc, dir4, dir5);
          olin[5] = Z(T(0));  // the linear delta was overwritten
        }

        // sums over the wave's 128 particles: the rows go through the exchange buffer, lane (g, kb) reads one eighth
        // of cotangent row g and of its two or three state rows, an 8-lane butterfly finishes each product
        Z* exz = reinterpret_cast<Z*>(ex);
#pragma unroll
        for (int c = 0; c < 6; ++c) exz[(c * P) / W + lane] = (c == 5 && entry_rows) ? dir5 : olin[c];
        exz[(6 * P) / W + lane] = dir4;
#pragma unroll
        for (int c = 0; c < 7; ++c) exz[(kStateRow[c] * P) / W + lane] = zin[c];
        wave_lds_sync();
        {
          const T* arow = ex + (g < 7 ? g : 1) * P + kb * VW;  // (group 7 has no row of its own: an odd one, see above)
          V av[NP];
#pragma unroll
          for (int l = 0; l < NP; ++l) av[l] = *reinterpret_cast<const V*>(arow + l * 8 * VW);
          const int zrows[3] = {zr0, zr1, 6};
          T tot[3];
#pragma unroll
          for (int k = 0; k < 3; ++k) {
            const T* zrow = ex + kStateRow[zrows[k]] * P + kb * VW;
            V part = {T(0), T(0), T(0), T(0)};
#pragma unroll
            for (int l = 0; l < NP; ++l) part = __builtin_elementwise_fma(av[l], *reinterpret_cast<const V*>(zrow + l * 8 * VW), part);
            tot[k] = sum_over_8_lanes((part[0] + part[1]) + (part[2] + part[3]));
          }
          // lane kb = k of the group books product k.  Where: unit's record rows 0..5 at [s][i * 7 + j]; a kicked pair's
          // entry block in the RUN's slot, rows of s (at [0..6]) and of delta (at [7..13]) like k_track_bwd
          if (kb < 3 && (kb < 2 || third)) {
            const T mine = kb == 0 ? tot[0] : (kb == 1 ? tot[1] : tot[2]);
            const int col = kb == 0 ? zr0 : (kb == 1 ? zr1 : 6);
            if (g < 5 || (g == 5 && !entry_rows)) acc[s * 64 + g * 7 + col] += mine;
            else if (g == 5) acc[(s - 1) * 64 + 7 + col] += mine;          // entry row of delta
            else if (g == 6 && entry_rows) acc[(s - 1) * 64 + col] += mine;  // entry row of s
          }
          if (kick) {  // the eight coefficient cotangents: one halving butterfly over the wave
            float each[8];
#pragma unroll
            for (int c = 0; c < 8; ++c) each[c] = zhsum(cc[c]);
            const T rest = sum_eight_over_wave(each, lane);
            if (lane < 8) acc[s * 64 + 49 + lane] += rest;
          }
        }
        wave_lds_sync();

        // z_bar_in = M^T o_lin, 16 entries (+ the constant component's own), then through the entry block
        {
          const float* m = map.v;
          const Z o0 = olin[0], o1 = olin[1], o2 = olin[2], o3 = olin[3], o4 = olin[4], o5 = olin[5];
          zb[0] = pk_fma(o1, m[3], o0 * m[0]);
          zb[1] = pk_fma(o1, m[4], o0 * m[1]);
          zb[2] = pk_fma(o3, m[9], o2 * m[6]);
          zb[3] = pk_fma(o3, m[10], o2 * m[7]);
          zb[4] = pk_fma(o5, m[14], o4 * m[12]);
          zb[5] = pk_fma(o5, m[15], o4 * m[13]);
          zb[6] = pk_fma(o3, m[11], pk_fma(o2, m[8], pk_fma(o1, m[5], pk_fma(o0, m[2], olin[6]))));
        }
        if (!paired) {
          zb[4] += dir4;
          zb[5] += dir5;
        } else if (kick) {
          zb[4] = pk_fma(dir5, pre[2], pk_fma(dir4, pre[0], zb[4]));
          zb[5] = pk_fma(dir5, pre[3], pk_fma(dir4, pre[1], zb[5]));
        }
      }
    }
    // what is left after the last (= first) unit is dL/d(incoming particle)
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
  T* out = partials + (((int64_t)b * a.chunks + chunk) * S) * kGradStride;
  for (int idx = tid; idx < S * 64; idx += kTrackThreads) {
    out[idx] = ((s_acc[idx] + s_acc[S * 64 + idx]) + s_acc[2 * S * 64 + idx]) + s_acc[3 * S * 64 + idx];
  }
}

inline size_t bwd_units_lds_bytes(int S) {
  return ((size_t)4 * kBwdUnitsRows * kBwdUnitsPitch + (size_t)4 * S * 64) * sizeof(float);
}

}  // namespace lynx
