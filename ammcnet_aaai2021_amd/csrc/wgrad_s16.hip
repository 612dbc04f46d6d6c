// Weight gradient of the convolutions with S16 operands on the fp16 MFMA pipe (training, see wgrad_f32.hip for
// the fp32 form and the GEMM view; 3x3 windows, and - round 4, this im2col form only - the 4x4 stride-1|2 windows of
// PixelDiscriminator and the 2x2 stride-2 window of a ConvTranspose's weight gradient):
//
//   dWp[n][k] += sum over pixels m of  G[m][n] * A[m + tap(k)][c(k)]          k = tap * Cin + c
//
// Both operands arrive pixel-major ([pixel][channels], channels in S16 groups of [8 hi | 8 lo]); the contraction index
// is the pixel, which is the ROW of those images, so the MFMA fragments (8 consecutive k per lane) are read with
// gfx950's transposing LDS load, ds_read_b64_tr_b16: a 16-lane group fetches 4 pixels x one 32-byte S16 group and
// every lane receives one 16-bit column of it.  An operand of 32 MFMA rows is therefore 2 channel groups = 16 channels
// x {hi, lo}, and ONE v_mfma_f32_32x32x16_f16 yields hi*hi, hi*lo, lo*hi and lo*lo of a 16 x 16 channel block at once
// (4 products for the 3 that matter); the epilogue folds them: planes of G sit 4 accumulator registers apart, planes
// of A 8 lanes apart.
//
// LDS: two stages of [32 pixels][BR] + [32 pixels][BC] 4-byte slots by LDS-DMA, as wgrad_f32.  The 32-byte group index
// is XORed with (pixel & 3) << 1 in the image, which makes the transposed reads (4 pixel rows x 2 groups per half
// wave) conflict free; the DMA applies the same permutation on its source side.
// A workgroup (8 waves as 4 x 2, each 32 rows x 64 columns) owns 128 rows (n) x 128 columns (k) of dWp and a slice of
// the pixels; fp32 atomics combine the slices.  The kernel is bound by its LDS-DMA traffic (16 MAC per staged byte).
// G may carry a power-of-two scale (ammc_split_rows_scaled_f32): `g_inv_scale` undoes it.
#include "ammc_common.h"
#include <hip/hip_fp16.h>

namespace ammc_s16 {

typedef _Float16 f16x8w __attribute__((ext_vector_type(8)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

struct WgradS16Args {
  AmmcWgradDesc d;
  const float* g_inv_scale;
  int M, kpad, cin_log2;
  int row_tiles, col_tiles, msplit, chunks_per_block, nchunks;
  int tap_w, a_step;             // window width (3, 4 or 2) and the stride of the window origin in `a` (1 or 2)
};

constexpr int WS_BR = 128, WS_BC = 128;         // tile of dWp: rows (n) x columns (k)
constexpr int WS_NT = 512;                      // 8 waves
constexpr int WS_GS = WS_BR / 4, WS_AS = WS_BC / 4;      // 16-byte slots per pixel row
constexpr int WS_GJ = 32 * WS_GS / WS_NT, WS_AJ = 32 * WS_AS / WS_NT;
constexpr int WS_GSTAGE = 32 * WS_BR, WS_ASTAGE = 32 * WS_BC;   // floats

__device__ __forceinline__ u32x2 ds_read_tr16(uint32_t addr) {
  u32x2 v;
  asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(v) : "v"(addr) : "memory");
  return v;
}

__global__ __launch_bounds__(WS_NT, 2) void wgrad_s16_kernel(WgradS16Args a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* Gs = smem;                              // [2][32][BR]
  float* As = smem + 2 * WS_GSTAGE;              // [2][32][BC]

  const AmmcWgradDesc& d = a.d;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int h = lane >> 5, l31 = lane & 31;
  const int wm = wave >> 1, wn = wave & 1;

  int bid = blockIdx.x;
  const int ms = bid % a.msplit;
  bid /= a.msplit;
  const int col0 = (bid % a.col_tiles) * WS_BC;
  const int row0 = (bid / a.col_tiles) * WS_BR;
  const int c_begin = ms * a.chunks_per_block;
  const int c_end = min(c_begin + a.chunks_per_block, a.nchunks);
  if (c_begin >= c_end) return;

  const int W = d.width, H = d.height;

  // DMA pieces: piece p -> pixel p / SLOTS of the chunk, physical slot p % SLOTS; the logical slot it fetches has its
  // group index (slot >> 1) XORed with (pixel & 3) << 1, i.e. slot ^ ((pixel & 3) << 2)
  int64_t a_toff[WS_AJ];
  int a_px[WS_AJ];
#pragma unroll
  for (int j = 0; j < WS_AJ; ++j) {
    const int p = j * WS_NT + tid;
    a_px[j] = p / WS_AS;
    const int ls = (p % WS_AS) ^ ((a_px[j] & 3) << 2);
    int k = col0 + 4 * ls;
    k = k < a.kpad ? k : a.kpad - 4;
    int tap = k >> a.cin_log2;
    tap = tap < d.ntaps - 1 ? tap : d.ntaps - 1;           // K padding: any valid address (those columns are never stored)
    const int r = a.tap_w == 3 ? (tap * 11) >> 5 : (a.tap_w == 4 ? tap >> 2 : tap >> 1), s = tap - a.tap_w * r;
    a_toff[j] = (int64_t)r * d.a_rs + (int64_t)s * d.a_ps + (k & (d.cin - 1));
  }
  int g_px[WS_GJ], g_col[WS_GJ];
#pragma unroll
  for (int j = 0; j < WS_GJ; ++j) {
    const int p = j * WS_NT + tid;
    g_px[j] = p / WS_GS;
    g_col[j] = row0 + 4 * ((p % WS_GS) ^ ((g_px[j] & 3) << 2));
  }

#define WS_ISSUE(chunk, stage)                                                                          \
  {                                                                                                     \
    float* gdst = Gs + (stage) * WS_GSTAGE + wave * 256;                                                \
    float* adst = As + (stage) * WS_ASTAGE + wave * 256;                                                \
    _Pragma("unroll") for (int j = 0; j < WS_GJ; ++j) {                                                 \
      const int m = (chunk) * 32 + g_px[j];                                                             \
      const float* src;                                                                                 \
      if (m < a.M) {                                                                                    \
        const int x = m % W, t = m / W;                                                                 \
        const int y = t % H, b = t / H;                                                                 \
        src = d.g + ((int64_t)b * d.g_bs + (int64_t)y * d.g_rs + (int64_t)x * d.g_ps) + g_col[j];      \
      } else {                                                                                          \
        src = d.zeros + (g_col[j] - row0);                                                              \
      }                                                                                                 \
      __builtin_amdgcn_global_load_lds(src, gdst + j * (WS_NT * 4), 16, 0, 0);                                 \
    }                                                                                                   \
    _Pragma("unroll") for (int j = 0; j < WS_AJ; ++j) {                                                 \
      int m = (chunk) * 32 + a_px[j];                                                                   \
      m = m < a.M ? m : a.M - 1;                                                                        \
      const int x = m % W, t = m / W;                                                                   \
      const int y = t % H, b = t / H;                                                                   \
      const float* src = d.a + ((int64_t)b * d.a_bs + (int64_t)(y * a.a_step) * d.a_rs + (int64_t)(x * a.a_step) * d.a_ps) + a_toff[j]; \
      __builtin_amdgcn_global_load_lds(src, adst + j * (WS_NT * 4), 16, 0, 0);                                 \
    }                                                                                                   \
  }

  // wave (wm, wn) owns G channels [32 wm, +32) = 4 groups = 2 operands and A columns [64 wn, +64) = 8 groups = 4 operands
  f32x16 acc[2][4];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // transposed-read addresses: lane 4q+p of a 16-lane group supplies pixel row q, bytes [8p, 8p+8) of the group
  const int l16 = lane & 15, q = l16 >> 2, p = l16 & 3;
  const int gi = l31 >> 4;                                  // which of the operand's two groups this lane reads
  // byte offset inside a pixel row of logical group g: ((2g + (p >> 1)) ^ (q << 2)) * 16 + 8 * (p & 1); the XOR only
  // touches the group bits 1..2, so it can be applied to g itself: g ^ (q << 1)
  uint32_t g_off[2], a_off[4];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int g = (wm * 4 + 2 * i + gi) ^ (q << 1);
    g_off[i] = (uint32_t)(g * 32 + 8 * p);
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int g = (wn * 8 + 2 * j + gi) ^ (q << 1);
    a_off[j] = (uint32_t)(g * 32 + 8 * p);
  }
  const uint32_t g_base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const void*)Gs;
  const uint32_t a_base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const void*)As;

  WS_ISSUE(c_begin, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  for (int c = c_begin; c < c_end; ++c) {
    const int stage = (c - c_begin) & 1;
    if (c + 1 < c_end) WS_ISSUE(c + 1, stage ^ 1);
    const uint32_t gst = g_base + (uint32_t)(stage * WS_GSTAGE * 4);
    const uint32_t ast = a_base + (uint32_t)(stage * WS_ASTAGE * 4);
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      // this lane's 8 pixels of the k-step: rows 16 s + 8 h + {0..3} and {4..7} (q picks the row inside each 4-block)
      const uint32_t row_lo = (uint32_t)(16 * s + 8 * h + q);
      const uint32_t row_hi = row_lo + 4;
      u32x2 g0[2], g1[2], a0[4], a1[4];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        g0[i] = ds_read_tr16(gst + row_lo * (WS_BR * 4) + g_off[i]);
        g1[i] = ds_read_tr16(gst + row_hi * (WS_BR * 4) + g_off[i]);
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        a0[j] = ds_read_tr16(ast + row_lo * (WS_BC * 4) + a_off[j]);
        a1[j] = ds_read_tr16(ast + row_hi * (WS_BC * 4) + a_off[j]);
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        u32x4 gv;
        gv[0] = g0[i][0]; gv[1] = g0[i][1]; gv[2] = g1[i][0]; gv[3] = g1[i][1];
        const f16x8w gf = __builtin_bit_cast(f16x8w, gv);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          u32x4 av;
          av[0] = a0[j][0]; av[1] = a0[j][1]; av[2] = a1[j][0]; av[3] = a1[j][1];
          const f16x8w af = __builtin_bit_cast(f16x8w, av);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(gf, af, acc[i][j], 0, 0, 0);
        }
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }
#undef WS_ISSUE

  // ---- fold the four (plane of G, plane of A) products and add to the packed gradient --------------------------
  // accumulator register r, half h  <->  G operand row (r & 3) + 8 (r >> 2) + 4 h:
  //     group r >> 3, plane (r >> 2) & 1, channel (r & 3) + 4 h;
  // lane l31  <->  A operand row l31: group l31 >> 4, plane (l31 >> 3) & 1, channel l31 & 7.
  const float inv = a.g_inv_scale ? a.g_inv_scale[0] : 1.f;
  constexpr float LO = 1.f / 2048.f;
  const bool a_hi = ((l31 >> 3) & 1) == 0;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int col = col0 + 64 * wn + 16 * j + 8 * (l31 >> 4) + (l31 & 7);
#pragma unroll
      for (int rr = 0; rr < 8; ++rr) {
        const int r = (rr & 3) + 8 * (rr >> 2);                       // registers with plane(G) = hi
        const float t = acc[i][j][r] + acc[i][j][r + 4] * LO;         // (hi + lo * 2^-11) of G against this lane's A plane
        const float other = __shfl_xor(t, 8);                         // the same against the other A plane
        if (a_hi) {
          const int row = row0 + wm * 32 + 16 * i + 8 * (r >> 3) + (r & 3) + 4 * h;
          if (row < d.n && col < a.kpad)
            unsafeAtomicAdd(d.dw + (int64_t)row * a.kpad + col, (t + other * LO) * inv);
        }
      }
    }
  }
}

// wgrad_tap_s16.hip: the halo-patch form; -12345 = not its case
int wgrad_tap_s16_try(const AmmcWgradDesc& d, const float* g_inv_scale, int kpad, hipStream_t stream);

}  // namespace ammc_s16
using namespace ammc_s16;

extern "C" int ammc_conv_wgrad_s16(const AmmcWgradDesc* desc, const float* g_inv_scale, void* stream) {
  if (!desc || !desc->g || !desc->a || !desc->dw || !desc->zeros) return AMMC_EINVAL;
  const AmmcWgradDesc& d = *desc;
  if (d.batch <= 0 || d.height <= 0 || d.width <= 0 || d.n <= 0 || (d.n % 32)) return AMMC_EINVAL;
  if ((d.ntaps != 9 && d.ntaps != 16 && d.ntaps != 4) || d.a_step < 0 || d.a_step > 2) return AMMC_EUNSUP;
  if (d.cin < 8 || (d.cin & (d.cin - 1))) return AMMC_EUNSUP;
  if (((uintptr_t)d.g | (uintptr_t)d.a | (uintptr_t)d.zeros) & 31) return AMMC_EINVAL;
  if ((d.g_bs | d.g_rs | d.g_ps | d.a_bs | d.a_rs | d.a_ps) & 7) return AMMC_EINVAL;
  const int64_t M = (int64_t)d.batch * d.height * d.width;
  if (M >= (1LL << 31)) return AMMC_EUNSUP;
  WgradS16Args a;
  a.d = d;
  a.g_inv_scale = g_inv_scale;
  a.M = (int)M;
  a.kpad = ((d.ntaps * d.cin + 31) / 32) * 32;
  a.cin_log2 = ammc_ilog2(d.cin);
  a.tap_w = d.ntaps == 9 ? 3 : (d.ntaps == 16 ? 4 : 2);
  a.a_step = d.a_step > 1 ? d.a_step : 1;
  if (d.ntaps == 9 && a.a_step == 1) {                    // the halo-patch forms: stride-1 3x3 layers
    const int rc = wgrad_tap_s16_try(d, g_inv_scale, a.kpad, (hipStream_t)stream);
    if (rc != -12345) return rc;
  }
  constexpr size_t lds = (size_t)2 * 32 * (WS_BR + WS_BC) * sizeof(float);
  static bool attr_done = false;
  if (!attr_done) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(wgrad_s16_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return (int)e;
    attr_done = true;
  }
  a.row_tiles = (d.n + WS_BR - 1) / WS_BR;
  a.col_tiles = (a.kpad + WS_BC - 1) / WS_BC;
  a.nchunks = (a.M + 31) / 32;
  const int tiles = a.row_tiles * a.col_tiles;
  int msplit = (2 * 256 + tiles - 1) / tiles;             // ~2 workgroups per CU, at least 8 chunks (256 pixels) each
  const int max_split = (a.nchunks + 7) / 8;
  if (msplit > max_split) msplit = max_split;
  if (msplit < 1) msplit = 1;
  a.chunks_per_block = (a.nchunks + msplit - 1) / msplit;
  a.msplit = (a.nchunks + a.chunks_per_block - 1) / a.chunks_per_block;
  hipLaunchKernelGGL(wgrad_s16_kernel, dim3(tiles * a.msplit), dim3(WS_NT), lds, (hipStream_t)stream, a);
  return ammc_launch_status();
}
