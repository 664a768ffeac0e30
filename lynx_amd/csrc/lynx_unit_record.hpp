// Compact records of the units of a multi-step float32 program (lynx_units.hpp explains what they are for): layout,
// the structure classes, and the packing of one record from a row of the step table.  Included by lynx_device.hpp
// (k_emit_steps packs them while it writes the step table) and lynx_units.hpp.
#pragma once

#include <hip/hip_runtime.h>

#include "lynx_maps.hpp"

namespace lynx {

constexpr int kMaxUnits = 64;
// A unit's compact record: 32 floats = 128 bytes, every one of them used by class U.
constexpr int kUnitStride = 32;
// record layout (floats; integers are stored as raw bits)
// first half: what the kick needs; second half: the linear map
constexpr int kUnitDesc = 0;    // bits: kUnit* below
constexpr int kUnitSlot = 1;    // step-table slot of the unit's (dense) record
constexpr int kUnitSinPhi = 2;  // sin(phi) of an active cavity (the kick's difference of cosines and its cotangents; cos(phi) is a coefficient)
constexpr int kUnitCoef = 4;    // 8 cavity coefficients (LYNX_C_*)
constexpr int kUnitInv = 12;    // 4: inverse of the cavity's (s, delta) block (merged pairs)
constexpr int kUnitMap = 16;    // 16 entries of class U, in the order of unit_entry_u()
// second array [B][U][16]: [0..7] the 8 additional entries of class D (order of unit_entry_d()); [8..11] merged pairs:
// the (s, delta) block of the RUN's map, rows 4 and 5 -- what gives the s and delta that enter the cavity (the reverse
// pass drives the kick with them)
constexpr int kUnitExtraStride = 16;
constexpr int kUnitPre = 8;

constexpr int kUnitKick = 1;      // active cavity: non-linear kick behind the linear map
constexpr int kUnitInverse = 2;   // ... driven by the entry inverse (merged pair) instead of the unit's own s, delta
constexpr int kUnitPair = 4;      // merged [run, cavity] pair
constexpr int kUnitRows = 8;      // ... whose kick is driven by rows 4, 5 of the run's map instead (LYNX_DESC_ILL)
constexpr int kUnitClassShift = 4;  // bits 4-5: the class this sample's map was found to have (dense if the check failed)

enum { kClassDense = 0, kClassU = 1, kClassD = 2 };

struct UnitPlan {
  int32_t n_units;
  unsigned char slot[kMaxUnits];  // step-table slot whose record the unit applies (the cavity's, for a merged pair)
  unsigned char cls[kMaxUnits];   // proposed class (whole batch)
  unsigned char pair[kMaxUnits];  // 1: merged [run, cavity] pair
};

// (row, col) of compact entry k of class U and of the 8 additional ones of class D
__host__ __device__ constexpr int unit_entry_u(int k) {
  constexpr int rc[16] = {0 * 7 + 0, 0 * 7 + 1, 0 * 7 + 6, 1 * 7 + 0, 1 * 7 + 1, 1 * 7 + 6, 2 * 7 + 2, 2 * 7 + 3,
                          2 * 7 + 6, 3 * 7 + 2, 3 * 7 + 3, 3 * 7 + 6, 4 * 7 + 4, 4 * 7 + 5, 5 * 7 + 4, 5 * 7 + 5};
  return rc[k];
}
__host__ __device__ constexpr int unit_entry_d(int k) {
  constexpr int rc[8] = {0 * 7 + 5, 1 * 7 + 5, 4 * 7 + 0, 4 * 7 + 1, 4 * 7 + 6, 5 * 7 + 0, 5 * 7 + 1, 5 * 7 + 6};
  return rc[k];
}

__host__ __device__ inline bool unit_pattern_has(int cls, int q) {
  if (cls == kClassDense) return true;
  for (int k = 0; k < 16; ++k)
    if (unit_entry_u(k) == q) return true;
  if (cls == kClassD)
    for (int k = 0; k < 8; ++k)
      if (unit_entry_d(k) == q) return true;
  return false;
}

// One unit record from the 64 scalars of its step-table row (`rec`: map 49, coefficients, entry inverse, descriptor):
// checks the proposed class `cls` against the numbers (every entry outside the pattern exactly zero, row 6 = e6, all
// 49 finite -- otherwise dense) and writes the 32-float record and the 8 class-D extras.
template <typename R>
__device__ __forceinline__ void pack_unit_record(const R* rec, const R* pre /* merged pair: r44 r45 r54 r55 of the run */,
                                                 int slot, int cls, int pair, float* __restrict__ out, float* __restrict__ ex) {
  const int desc = (int)rec[LYNX_FLAGS_OFFSET];
  const bool kick = ((desc >> LYNX_DESC_KIND_SHIFT) & 3) == LYNX_STEP_CAVITY && (desc & LYNX_FLAG_CAV_GAIN);
  bool ok = cls != kClassDense;
  for (int q = 0; q < 49 && ok; ++q) {
    const float v = (float)rec[q];
    if (!(__builtin_fabsf(v) <= 3.4028234664e38f)) ok = false;  // NaN or Inf anywhere: dense
    if (q >= 42) {
      if (v != (q == 48 ? 1.0f : 0.0f)) ok = false;  // row 6 must be e6
    } else if (!unit_pattern_has(cls, q) && v != 0.0f) {
      ok = false;
    }
  }
  const bool rows = kick && pair && (desc & LYNX_DESC_ILL);
  const int bits = (kick ? kUnitKick : 0) | ((kick && pair && !rows) ? kUnitInverse : 0) | (pair ? kUnitPair : 0) |
                   (rows ? kUnitRows : 0) |
                   ((ok ? cls : (int)kClassDense) << kUnitClassShift);
  out[kUnitDesc] = __int_as_float(bits);
  out[kUnitSlot] = __int_as_float(slot);
  out[kUnitSinPhi] = kick ? (float)rec[LYNX_SINPHI_OFFSET] : 0.f;  // (the builders left it in the row: the same number the dense step loop reads)
  out[3] = 0.f;
#pragma unroll
  for (int k = 0; k < 8; ++k) out[kUnitCoef + k] = (float)rec[LYNX_COEF_OFFSET + k];
#pragma unroll
  for (int k = 0; k < 4; ++k) out[kUnitInv + k] = (float)rec[LYNX_ENTRY_OFFSET + k];
#pragma unroll
  for (int k = 0; k < 16; ++k) out[kUnitMap + k] = (float)rec[unit_entry_u(k)];
#pragma unroll
  for (int k = 0; k < 8; ++k) ex[k] = (float)rec[unit_entry_d(k)];
#pragma unroll
  for (int k = 0; k < 4; ++k) ex[kUnitPre + k] = pair ? (float)pre[k] : ((k == 0 || k == 3) ? 1.0f : 0.0f);
#pragma unroll
  for (int k = kUnitPre + 4; k < kUnitExtraStride; ++k) ex[k] = 0.f;
}

// what k_emit_steps is told about a step: -1 = no unit applies this step's record (the run half of a merged pair),
// else unit | proposed class << 8 | pair << 10
__host__ __device__ inline int step_unit_code(int unit, int cls, int pair) { return unit | (cls << 8) | (pair << 10); }

}  // namespace lynx
