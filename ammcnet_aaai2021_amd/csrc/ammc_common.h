// Shared device/host helpers for libammc_hip (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/ammc_hip.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define AMMC_WAVE 64

static inline int ammc_launch_status() {
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? AMMC_OK : (int)e;
}

// Bijective XCD-aware remap of a 1-D grid: blocks b and b+8 share an XCD (and
// its L2), so give each XCD a contiguous run of logical ids.  Speed only.
__device__ __forceinline__ int ammc_xcd_remap(int bid, int nblk) {
  const int q = nblk >> 3, r = nblk & 7, xcd = bid & 7;
  const int start = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return start + (bid >> 3);
}

// dispatch option "s16_mf" (capi_misc.hip): MFMA shape of the halo-patch kernel, -1 = per variant (the measured
// faster one, default), 1 = v_mfma_f32_16x16x32_f16, 0 = v_mfma_f32_32x32x16_f16; AMMC_S16_MF / ammc_set_option
int ammc_opt_s16_mf();

static inline int ammc_ilog2(int v) {
  int l = 0;
  while ((1 << l) < v) ++l;
  return l;
}
