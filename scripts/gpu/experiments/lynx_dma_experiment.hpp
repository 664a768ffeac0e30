// EXPERIMENT (round 3): the wave-tile form of k_track_direct<float, 3, false, 4, false, true> with
// the incoming tile brought in by LDS-DMA (global_load_lds_dwordx4: a wave's 1 KiB lands in LDS without passing through
// registers) instead of seven full-width loads into 28 prefetch registers + seven ds_write_b128.  Single composed map,
// float32, property-set moments in float32 lane sums -- exactly the headline configuration, nothing else.
// Two LDS regions per wave: the incoming tile of the NEXT iteration lands in one while the current tile is transposed
// out of the other and its results leave through it.  Enabled with LYNX_LDS_DMA=1; never the default.
#pragma once

#include "lynx_device.hpp"

namespace lynx {

__device__ __forceinline__ void dma_tile_fetch(const float* g, int lane, unsigned char* lds_region) {
  const lynx_u32x4u* src = reinterpret_cast<const lynx_u32x4u*>(g);
#pragma unroll
  for (int k = 0; k < 7; ++k)
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + k * 64 + lane),
                                     (__attribute__((address_space(3))) void*)(lds_region + k * 1024), 16, 0, 0);
}

__global__ __launch_bounds__(kTrackThreads, 2) void k_track_tiles_dma(TrackArgs a, const float* p_in, float* p_out,
                                                                      const float* __restrict__ steps_in,
                                                                      double* __restrict__ partials) {
  using T = float;
  constexpr int UNROLL = 4;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int64_t b = blockIdx.x / a.chunks;
  const int chunk = blockIdx.x % a.chunks;
  const int64_t end = a.n_particles;
  constexpr int64_t kTile = (int64_t)kTrackThreads * UNROLL, kWaveSpan = 64 * UNROLL;
  const T* src = p_in + b * a.in_stride;
  T* dst = p_out + b * end * 7;
  unsigned char* region[2] = {smem_raw + (wave * 2 + 0) * kWaveTileBytes, smem_raw + (wave * 2 + 1) * kWaveTileBytes};
  const T* g_steps = steps_in + b * (int64_t)LYNX_STEP_STRIDE;
  auto wave_base = [&](int it) { return ((int64_t)chunk * a.tiles_per_wg + it) * kTile + (int64_t)wave * kWaveSpan; };

  if (wave_base(0) + kWaveSpan <= end) dma_tile_fetch(src + wave_base(0) * 7, lane, region[0]);
  T m0[49];
#pragma unroll
  for (int i = 0; i < 49; ++i) m0[i] = uniform_value(g_steps[i]);
  LaneSums<T, 3, false> sums;
  sums.init();
  T shift[6];
#pragma unroll
  for (int i = 0; i < 6; ++i) shift[i] = T(0);

  for (int it = 0; it < a.tiles_per_wg; ++it) {
    const int64_t wb = wave_base(it);
    if (wb >= end) break;
    const bool full = wb + kWaveSpan <= end;  // wave-uniform
    unsigned char* cur = region[it & 1];
    const int64_t i0 = wb + (int64_t)lane * UNROLL;
    T z[UNROLL][7];
    if (full) {
      // the tile has landed: behind its seven DMA loads only the previous iteration's seven stores may still be out
      // (vector memory operations of a wave retire in order)
      if (it > 0 && a.store) __builtin_amdgcn_s_waitcnt(0x0f77);  // vmcnt(7)
      else __builtin_amdgcn_s_waitcnt(0x0f70);                    // vmcnt(0)
      wave_fence();
      const lynx_u32x4* l = reinterpret_cast<const lynx_u32x4*>(cur);
      lynx_u32x4 w[7];
#pragma unroll
      for (int k = 0; k < 7; ++k) w[k] = l[lane * 7 + k];
#pragma unroll
      for (int f = 0; f < UNROLL * 7; ++f) z[f / 7][f % 7] = __uint_as_float(w[f / 4][f % 4]);
    } else {
#pragma unroll
      for (int u = 0; u < UNROLL; ++u) load_particle(src + (i0 + u < end ? i0 + u : end - 1) * 7, z[u]);
    }
    if (it + 1 < a.tiles_per_wg && wave_base(it + 1) + kWaveSpan <= end) {
      wave_fence();  // the other region's last readers (its flat read-back two iterations ago) are done: LDS is in order
      dma_tile_fetch(src + wave_base(it + 1) * 7, lane, region[(it + 1) & 1]);
    }
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) {
      apply_step<T>(m0, LYNX_STEP_RUN, 0, z[u]);
      if (it == 0 && u == 0) {
#pragma unroll
        for (int k = 0; k < 6; ++k) shift[k] = wave_first(z[0][k]);
      }
      if (i0 + u < end) {
        if (a.store && !full) store_particle(dst + (i0 + u) * 7, z[u]);
        sums.add(z[u], shift);
      }
    }
    if (a.store && full) {
      lynx_u32x4 vo[7];
      wave_tile_from_particles<T, UNROLL>(z, cur, lane, vo);
      wave_tile_store(dst + wb * 7, lane, vo);
    }
  }
  workgroup_moment_record<T, 3, false>(sums, shift, smem_raw, partials + ((int64_t)b * a.chunks + chunk) * kPartialStride);
}

}  // namespace lynx
