// Weight gradient of the implicit-GEMM convolutions on the fp32 MFMA pipe.
//
//   dWp[n][k] += sum over pixels m of  G[m][n] * A[m, k]          k = tap * Cin + c
//
// G is the gradient flowing into the conv's raw output (NHWC), A the conv's input gathered
// exactly as in the forward kernel (tap (r,s) of pixel m, channel c).  The result uses the
// forward kernel's packed weight layout [N][Kpad]; ammc_unpack_* turn it into the module's
// OIHW / IOHW parameter gradients.  The same kernel serves
//   3x3 conv      ntaps 9          (autograd of unet.py:11,14)
//   1x1 conv      ntaps 1          (enc / dec, unet.py:321-323)
//   4x4 conv      ntaps 16, a_step 1|2 (PixelDiscriminator, pix2pix_networks.py:604-621)
//   ConvTranspose ntaps 4, a_step 2: rows = input channels, columns = (dy,dx,c_out), "G" is the
//                 layer INPUT and "A" the output gradient gathered 2x2 stride 2 (unet.py:47)
//
// GEMM view: the contraction runs over pixels, so both LDS tiles are pixel-major
// [32 px][128 rows] / [32 px][128 cols]; a lane's MFMA operand is one dword of a pixel row
// (ds_read_b32, lanes along the contiguous dimension: conflict free), both tiles arrive by
// LDS-DMA (`global_load_lds_dwordx4`) in 16-B pieces, two stages.  A workgroup owns one
// 128x128 (or 64x128) tile of dWp and a slice of the pixels (split-M); partial tiles are
// combined with fp32 global atomics (rows of 128 B per wave instruction: the full-rate shape).
// Out-of-range pixels of the last chunk read G from a caller-supplied row of zeros.
// Roofline: MFMA fp32, same arithmetic intensity as the forward kernel.
#include "ammc_common.h"

namespace ammc_impl {

struct WgradArgs {
  AmmcWgradDesc d;
  int M, kpad, cin_log2;
  int row_tiles, col_tiles, msplit, chunks_per_block, nchunks;
};

template <int WGM, int WGN, int TM, int TN>
__global__ __launch_bounds__(256, 2) void wgrad_f32_kernel(WgradArgs a) {
  constexpr int BR = WGM * TM * 32;          // rows (n) of the dW tile
  constexpr int BC = WGN * TN * 32;          // columns (k)
  constexpr int G_STAGE = 32 * BR;           // floats
  constexpr int A_STAGE = 32 * BC;
  constexpr int GS = BR / 4;                 // 16-B slots per pixel row
  constexpr int AS = BC / 4;
  constexpr int GJ = 32 * GS / 256;          // pieces per thread per chunk
  constexpr int AJ = 32 * AS / 256;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* Gs = smem;                          // [2][32][BR]
  float* As = smem + 2 * G_STAGE;            // [2][32][BC]

  const AmmcWgradDesc& d = a.d;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WGN, wn = wave % WGN;
  const int h = lane >> 5, l31 = lane & 31;

  int bid = blockIdx.x;
  const int ms = bid % a.msplit;
  bid /= a.msplit;
  const int col0 = (bid % a.col_tiles) * BC;
  const int row0 = (bid / a.col_tiles) * BR;
  const int c_begin = ms * a.chunks_per_block;
  const int c_end = min(c_begin + a.chunks_per_block, a.nchunks);
  if (c_begin >= c_end) return;

  const int W = d.width, H = d.height;
  const int astep = d.a_step > 1 ? d.a_step : 1;

  // per-thread constant parts of the gather: which (tap, channel) its A pieces cover
  int64_t a_toff[AJ];
  int a_px[AJ];
#pragma unroll
  for (int j = 0; j < AJ; ++j) {
    const int p = j * 256 + tid;
    a_px[j] = p / AS;
    int k = col0 + 4 * (p % AS);
    k = k < a.kpad ? k : a.kpad - 4;                       // columns past Kpad: any valid address
    if (d.ntaps == 1) {
      a_toff[j] = k < d.cin ? k : 0;
    } else {
      int tap = k >> a.cin_log2;
      tap = tap < d.ntaps - 1 ? tap : d.ntaps - 1;
      int r, s;
      if (d.ntaps == 9) { r = (tap * 11) >> 5; s = tap - 3 * r; }
      else if (d.ntaps == 16) { r = tap >> 2; s = tap & 3; }
      else { r = tap >> 1; s = tap & 1; }
      a_toff[j] = (int64_t)r * d.a_rs + (int64_t)s * d.a_ps + (k & (d.cin - 1));
    }
  }
  int g_px[GJ], g_col[GJ];
#pragma unroll
  for (int j = 0; j < GJ; ++j) {
    const int p = j * 256 + tid;
    g_px[j] = p / GS;
    g_col[j] = row0 + 4 * (p % GS);
  }

  // (a macro, not a lambda: device lambdas inside a kernel template lose hipcc's host stub)
#define WG_ISSUE(chunk, stage)                                                                          \
  {                                                                                                     \
    float* gdst = Gs + (stage) * G_STAGE + wave * 256;                                                  \
    float* adst = As + (stage) * A_STAGE + wave * 256;                                                  \
    _Pragma("unroll") for (int j = 0; j < GJ; ++j) {                                                    \
      const int m = (chunk) * 32 + g_px[j];                                                             \
      const float* src;                                                                                 \
      if (m < a.M) {                                                                                    \
        const int x = m % W, t = m / W;                                                                 \
        const int y = t % H, b = t / H;                                                                 \
        src = d.g + ((int64_t)b * d.g_bs + (int64_t)y * d.g_rs + (int64_t)x * d.g_ps) + g_col[j];      \
      } else {                                                                                          \
        src = d.zeros + (g_col[j] - row0);                                                              \
      }                                                                                                 \
      __builtin_amdgcn_global_load_lds(src, gdst + j * 1024, 16, 0, 0);                                 \
    }                                                                                                   \
    _Pragma("unroll") for (int j = 0; j < AJ; ++j) {                                                    \
      int m = (chunk) * 32 + a_px[j];                                                                   \
      m = m < a.M ? m : a.M - 1;                                                                        \
      const int x = m % W, t = m / W;                                                                   \
      const int y = t % H, b = t / H;                                                                   \
      const float* src = d.a + ((int64_t)b * d.a_bs + (int64_t)(y * astep) * d.a_rs +                  \
                                (int64_t)(x * astep) * d.a_ps) + a_toff[j];                             \
      __builtin_amdgcn_global_load_lds(src, adst + j * 1024, 16, 0, 0);                                 \
    }                                                                                                   \
  }

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int g_lane = wm * TM * 32 + l31 + h * BR;        // + (2*s) * BR per k-step
  const int a_lane = wn * TN * 32 + l31 + h * BC;

  WG_ISSUE(c_begin, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  for (int c = c_begin; c < c_end; ++c) {
    const int stage = (c - c_begin) & 1;
    if (c + 1 < c_end) WG_ISSUE(c + 1, stage ^ 1);
    const float* Gc = Gs + stage * G_STAGE + g_lane;
    const float* Ac = As + stage * A_STAGE + a_lane;
#pragma unroll
    for (int s = 0; s < 16; ++s) {
      float gf[TM], af[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) gf[i] = Gc[2 * s * BR + i * 32];
#pragma unroll
      for (int j = 0; j < TN; ++j) af[j] = Ac[2 * s * BC + j * 32];
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(gf[i], af[j], acc[i][j], 0, 0, 0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }
#undef WG_ISSUE

  // ---- combine: fp32 atomics into the packed gradient -------------------------------
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int col = col0 + (wn * TN + j) * 32 + l31;
    if (col >= a.kpad) continue;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = row0 + (wm * TM + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
        if (row < d.n) unsafeAtomicAdd(d.dw + (int64_t)row * a.kpad + col, acc[i][j][r]);
      }
    }
  }
}

template <int WGM, int WGN, int TM, int TN>
int launch_wgrad(WgradArgs a, hipStream_t stream, int num_cu_hint) {
  constexpr int BR = WGM * TM * 32;
  constexpr int BC = WGN * TN * 32;
  constexpr size_t lds = (size_t)2 * 32 * (BR + BC) * sizeof(float);
  auto kern = wgrad_f32_kernel<WGM, WGN, TM, TN>;
  if (lds > 48 * 1024) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return (int)e;
  }
  a.row_tiles = (a.d.n + BR - 1) / BR;
  a.col_tiles = (a.kpad + BC - 1) / BC;
  a.nchunks = (a.M + 31) / 32;
  const int tiles = a.row_tiles * a.col_tiles;
  // enough workgroups for ~4 per CU, at least 8 chunks (256 pixels) each
  int msplit = (4 * num_cu_hint + tiles - 1) / tiles;
  const int max_split = (a.nchunks + 7) / 8;
  if (msplit > max_split) msplit = max_split;
  if (msplit < 1) msplit = 1;
  a.chunks_per_block = (a.nchunks + msplit - 1) / msplit;
  a.msplit = (a.nchunks + a.chunks_per_block - 1) / a.chunks_per_block;
  hipLaunchKernelGGL(kern, dim3(tiles * a.msplit), dim3(256), lds, stream, a);
  return ammc_launch_status();
}

// packed [cout][Kpad] (k = tap*cin_p + c) -> OIHW [cout][cin][ks][ks]
__global__ __launch_bounds__(256) void unpack_conv_wgrad_kernel(const float* __restrict__ p, int cout, int cin,
                                                                int ks2, int cin_p, int kpad,
                                                                float* __restrict__ out) {
  const int64_t total = (int64_t)cout * cin * ks2;
  const int64_t gid = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (gid >= total) return;
  const int tap = (int)(gid % ks2);
  const int c = (int)((gid / ks2) % cin);
  const int n = (int)(gid / ((int64_t)ks2 * cin));
  out[gid] = p[(int64_t)n * kpad + tap * cin_p + c];
}

// packed [cin][4*co] (k = g*co + c_out) -> IOHW [cin][co][2][2]
__global__ __launch_bounds__(256) void unpack_convt_wgrad_kernel(const float* __restrict__ p, int cin, int co,
                                                                 float* __restrict__ out) {
  const int64_t total = (int64_t)cin * co * 4;
  const int64_t gid = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (gid >= total) return;
  const int g = (int)(gid & 3);
  const int c_out = (int)((gid >> 2) % co);
  const int ci = (int)(gid / ((int64_t)4 * co));
  out[gid] = p[(int64_t)ci * 4 * co + g * co + c_out];
}

}  // namespace ammc_impl
using namespace ammc_impl;

extern "C" int ammc_conv_wgrad_f32(const AmmcWgradDesc* desc, void* stream) {
  if (!desc || !desc->g || !desc->a || !desc->dw || !desc->zeros) return AMMC_EINVAL;
  const AmmcWgradDesc& d = *desc;
  if (d.batch <= 0 || d.height <= 0 || d.width <= 0 || d.n <= 0 || (d.n % 32)) return AMMC_EINVAL;
  if (d.ntaps != 9 && d.ntaps != 4 && d.ntaps != 1 && d.ntaps != 16) return AMMC_EINVAL;
  if (d.ntaps != 1 && (d.cin < 4 || (d.cin & (d.cin - 1)))) return AMMC_EUNSUP;
  if (d.ntaps == 1 && (d.cin <= 0 || d.cin % 32)) return AMMC_EUNSUP;
  if (((uintptr_t)d.g | (uintptr_t)d.a | (uintptr_t)d.zeros) & 15) return AMMC_EINVAL;
  if ((d.g_bs | d.g_rs | d.g_ps | d.a_bs | d.a_rs | d.a_ps) & 3) return AMMC_EINVAL;
  const int64_t M = (int64_t)d.batch * d.height * d.width;
  if (M >= (1LL << 31)) return AMMC_EUNSUP;
  WgradArgs a;
  a.d = d;
  a.M = (int)M;
  a.kpad = ((d.ntaps * d.cin + 31) / 32) * 32;
  a.cin_log2 = ammc_ilog2(d.cin);
  hipStream_t s = (hipStream_t)stream;
  if (d.n % 128 == 0) return launch_wgrad<2, 2, 2, 2>(a, s, 256);   // 128 rows x 128 cols
  if (d.n % 64 == 0) return launch_wgrad<1, 4, 2, 1>(a, s, 256);    // 64 rows x 128 cols
  return launch_wgrad<1, 4, 1, 1>(a, s, 256);                       // 32 rows x 128 cols (outc, first layers)
}

extern "C" int ammc_unpack_conv_wgrad_f32(const float* packed, int32_t cout, int32_t cin, int32_t ksize,
                                          int32_t cin_p, float* out_oihw, void* stream) {
  if (!packed || !out_oihw || cout <= 0 || cin <= 0 || cin_p < cin || (ksize != 1 && ksize != 3 && ksize != 4)) return AMMC_EINVAL;
  const int ks2 = ksize * ksize;
  const int kpad = ((ks2 * cin_p + 31) / 32) * 32;
  const int64_t total = (int64_t)cout * cin * ks2;
  hipLaunchKernelGGL(unpack_conv_wgrad_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0,
                     (hipStream_t)stream, packed, cout, cin, ks2, cin_p, kpad, out_oihw);
  return ammc_launch_status();
}

extern "C" int ammc_unpack_convt_wgrad_f32(const float* packed, int32_t cin, int32_t co, float* out_iohw,
                                           void* stream) {
  if (!packed || !out_iohw || cin <= 0 || co <= 0) return AMMC_EINVAL;
  const int64_t total = (int64_t)cin * co * 4;
  hipLaunchKernelGGL(unpack_convt_wgrad_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0,
                     (hipStream_t)stream, packed, cin, co, out_iohw);
  return ammc_launch_status();
}
