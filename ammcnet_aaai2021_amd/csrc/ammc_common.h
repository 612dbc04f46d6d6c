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

// S16 encoding of two fp32 values, packed: hi = RNE fp16 of t, lo = RNE fp16 of (t - hi) * 2^11 - the same bits as the
// scalar expression `hv = (_Float16)t; lo = (_Float16)((t - (float)hv) * 2048.f)` (t - hi is exact in fp32, the scaling
// is a power of two), in five VALU operations per PAIR: v_cvt_pk_f16_f32, two v_fma_mix_f32 that read their half
// straight out of the packed register, v_fma_mixlo / mixhi_f16 that scale, round and pack.  (Left to the compiler the
// epilogues spent 10 per pair on this: cvt, cvt back, sub, mul, cvt for each element and a pack.)
typedef _Float16 ammc_h2 __attribute__((ext_vector_type(2)));
typedef float ammc_f2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void ammc_s16_split2(float t0, float t1, unsigned& hi, unsigned& lo) {
  const ammc_f2 t = {t0, t1};
  const ammc_h2 h = __builtin_convertvector(t, ammc_h2);
  hi = __builtin_bit_cast(unsigned, h);
  float d0, d1;
  asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(d0) : "v"(hi), "v"(t0));
  asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(d1) : "v"(hi), "v"(t1));
  const float s2048 = 2048.f;
  asm("v_fma_mixlo_f16 %0, %1, %2, 0 op_sel:[0,0,0] op_sel_hi:[0,0,0]" : "=v"(lo) : "v"(d0), "s"(s2048));
  asm("v_fma_mixhi_f16 %0, %1, %2, 0 op_sel:[0,0,0] op_sel_hi:[0,0,0]" : "+v"(lo) : "v"(d1), "s"(s2048));
}
// eight values -> the 16-byte hi and lo halves of one S16 group
typedef unsigned ammc_u4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void ammc_s16_split8(const float (&v)[8], ammc_u4& hi, ammc_u4& lo) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    unsigned h, l;
    ammc_s16_split2(v[2 * i], v[2 * i + 1], h, l);
    hi[i] = h, lo[i] = l;
  }
}

// dispatch option "s16_mf" (capi_misc.hip): MFMA shape of the halo-patch kernel, -1 = per variant (the measured
// faster one, default), 1 = v_mfma_f32_16x16x32_f16, 0 = v_mfma_f32_32x32x16_f16; AMMC_S16_MF / ammc_set_option
int ammc_opt_s16_mf();
// dispatch option "outc_stream": 1 = the output layer on conv_outc_s16.hip (default), 0 = on the halo-patch kernel;
// AMMC_OUTC_STREAM / ammc_set_option
int ammc_opt_outc_stream();
// dispatch option "memory_rt": rows per workgroup of memory_topk_s16, 0 = by size (default), 1 = 32, 2 = 64;
// AMMC_MEMORY_RT / ammc_set_option
int ammc_opt_memory_rt();
// dispatch option "memory_split": ammc_memory_topk_fwd_f16 as one fused launch (0), or as contraction launches on the
// caller's stream with the gather / commit of each chunk of rows on a second stream beside the next chunk's contraction
// (1: opt-in, measured 1 % faster at 262144 rows); -1 / 0 = fused (default); AMMC_MEMORY_SPLIT / ammc_set_option
int ammc_opt_memory_split();

static inline int ammc_ilog2(int v) {
  int l = 0;
  while ((1 << l) < v) ++l;
  return l;
}
