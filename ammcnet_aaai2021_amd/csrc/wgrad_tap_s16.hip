// 3x3 weight gradient, S16 operands, with the INPUT PATCH resident in LDS across the nine taps (training).
//
// wgrad_s16.hip contracts dWp[n][(tap, c)] over pixels with an im2col view of the layer input: every tap re-fetches the
// input, and the kernel is bound by that LDS-DMA traffic (16 MAC per staged byte).  Here a workgroup walks over spatial
// patches of 4 rows x 32 pixels; per patch it stages the output-gradient patch G [128 px][TN channels] and ONE halo
// patch of the input A [6 x 34 px][TC channels]; tap (r, s) is a row offset (r * 34 + s) into the A image - the
// contraction index IS the pixel row of both images, and the MFMA fragments are transposed reads
// (ds_read_b64_tr_b16) of them, so a shifted tap costs nothing.  ~30 MAC per staged byte.
//
// Accumulators: an MFMA tile holds the four plane products of a 16 x 16 channel block (see wgrad_s16.hip), and the
// nine taps are nine separate output blocks, so a wave keeps 9 tiles (144 registers): 8 waves = 72 tiles =
//   <8, 1>: 128 gradient channels (8 operands) x 16 input channels x 9 taps      (layers with >= 128 filters)
//   <4, 2>:  64 gradient channels (4 operands) x 32 input channels x 9 taps      (64 filters)
// LDS images, two stages each: G rows of TN * 4 bytes (group index XORed with (pixel & 3) << 1), A rows of 64 B (no
// swizzle needed: four consecutive rows already cover all banks) or 128 B (group ^ ((pixel >> 1) & 1) << 1).
// One workgroup per CU (154 / 116 KB); split over patches, fp32 atomics into the packed gradient.
// Needs H % 4 == 0, W % 32 == 0, Cin % (16 NA) == 0, N % (16 NG) == 0; ammc_conv_wgrad_s16 falls back otherwise.
#include "ammc_common.h"
#include <hip/hip_fp16.h>
#include <stdlib.h>

namespace ammc_s16 {

typedef _Float16 f16x8v __attribute__((ext_vector_type(8)));
typedef unsigned u32x2v __attribute__((ext_vector_type(2)));
typedef unsigned u32x4v __attribute__((ext_vector_type(4)));

struct WgradTapArgs {
  AmmcWgradDesc d;
  const float* g_inv_scale;
  int kpad, tiles_x, tiles_y, npatch, patches_per_block, msplit, row_tiles, col_tiles;
};

constexpr int WT_PH = 4, WT_PW = 32, WT_PX = WT_PH * WT_PW;            // 128 output pixels per patch
constexpr int WT_HW = WT_PW + 2, WT_HPX = (WT_PH + 2) * WT_HW;         // 204 halo pixels
constexpr int WT_NT = 512;

__device__ __forceinline__ u32x2v wt_read_tr16(uint32_t addr) {
  u32x2v v;
  asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(v) : "v"(addr) : "memory");
  return v;
}

template <int NG, int NA>
__global__ __launch_bounds__(WT_NT, 2) void wgrad_tap_s16_kernel(WgradTapArgs a) {
  static_assert(NG * NA == 8, "8 waves");
  constexpr int TN = 16 * NG, TC = 16 * NA;
  constexpr int G_RB = TN * 4, A_RB = TC * 4;                   // row bytes of the images
  constexpr int G_SLOTS = TN / 4, A_SLOTS = TC / 4;             // 16-byte slots per row
  constexpr int G_PIECES = WT_PX * G_SLOTS, A_PIECES = WT_HPX * A_SLOTS;
  constexpr int GJ = G_PIECES / WT_NT, AJ = (A_PIECES + WT_NT - 1) / WT_NT;
  constexpr int G_STAGE = WT_PX * TN;                           // floats
  constexpr int A_STAGE = AJ * WT_NT * 4;                       // floats (padded to whole rounds)
  static_assert(G_PIECES % WT_NT == 0, "G pieces");
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* Gs = smem;                                             // [2][128 px][TN]
  float* As = smem + 2 * G_STAGE;                               // [2][204 px][TC]

  const AmmcWgradDesc& d = a.d;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int h = lane >> 5, l31 = lane & 31;
  const int wg = wave % NG, wa = wave / NG;                     // this wave's G operand and A operand

  int bid = blockIdx.x;
  const int ms = bid % a.msplit;
  bid /= a.msplit;
  const int c0 = (bid % a.col_tiles) * TC;                      // first input channel of the tile
  const int row0 = (bid / a.col_tiles) * TN;                    // first gradient channel
  const int p_begin = ms * a.patches_per_block;
  const int p_end = min(p_begin + a.patches_per_block, a.npatch);
  if (p_begin >= p_end) return;

  // swizzles of the 32-byte group index by pixel row (see the header)
#define WT_GSW(px) (((px) & 3) << 1)
#define WT_ASW(px) (NA == 2 ? ((((px) >> 1) & 1) << 1) : 0)

  // DMA pieces: piece p -> image row p / SLOTS, physical slot p % SLOTS; the slot it fetches: group XORed back
  int g_off[GJ], a_off[AJ];
#pragma unroll
  for (int j = 0; j < GJ; ++j) {
    const int p = j * WT_NT + tid;
    const int px = p / G_SLOTS, ps = p % G_SLOTS;
    const int ls = ps ^ (WT_GSW(px) << 1);
    g_off[j] = (int)((int64_t)(px >> 5) * d.g_rs + (int64_t)(px & 31) * d.g_ps) + row0 + 4 * ls;
  }
#pragma unroll
  for (int j = 0; j < AJ; ++j) {
    int p = j * WT_NT + tid;
    p = p < A_PIECES ? p : A_PIECES - 1;
    const int hp = p / A_SLOTS, ps = p % A_SLOTS;
    const int ls = ps ^ (WT_ASW(hp) << 1);
    const int hy = hp / WT_HW, hx = hp - hy * WT_HW;
    a_off[j] = (int)((int64_t)hy * d.a_rs + (int64_t)hx * d.a_ps) + c0 + 4 * ls;
  }

#define WT_ISSUE(patch, stage)                                                                            \
  {                                                                                                       \
    int sp_ = (patch);                                                                                    \
    const int tx_ = sp_ % a.tiles_x;                                                                      \
    sp_ /= a.tiles_x;                                                                                     \
    const int ty_ = sp_ % a.tiles_y, b_ = sp_ / a.tiles_y;                                                \
    const float* gp_ = d.g + ((int64_t)b_ * d.g_bs + (int64_t)(ty_ * WT_PH) * d.g_rs + (int64_t)(tx_ * WT_PW) * d.g_ps); \
    const float* ap_ = d.a + ((int64_t)b_ * d.a_bs + (int64_t)(ty_ * WT_PH) * d.a_rs + (int64_t)(tx_ * WT_PW) * d.a_ps); \
    float* gdst_ = Gs + (stage) * G_STAGE + wave * 256;                                                   \
    float* adst_ = As + (stage) * A_STAGE + wave * 256;                                                   \
    _Pragma("unroll") for (int j = 0; j < GJ; ++j) {                                                      \
      const float* src_ = gp_ + g_off[j];                                                                 \
      __builtin_amdgcn_global_load_lds(src_, gdst_ + j * (WT_NT * 4), 16, 0, 0);                          \
    }                                                                                                     \
    _Pragma("unroll") for (int j = 0; j < AJ; ++j) {                                                      \
      const float* src_ = ap_ + a_off[j];                                                                 \
      __builtin_amdgcn_global_load_lds(src_, adst_ + j * (WT_NT * 4), 16, 0, 0);                          \
    }                                                                                                     \
  }

  f32x16 acc[9];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

  // transposed-read lane roles: lane 4q+p of a 16-lane group supplies image row q, bytes [8p, 8p+8) of its group
  const int l16 = lane & 15, q = l16 >> 2, p8 = (l16 & 3) * 8;
  const int gi = l31 >> 4;
  const int g_grp = wg * 2 + gi, a_grp = wa * 2 + gi;
  const uint32_t g_base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const void*)Gs;
  const uint32_t a_base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const void*)As;

  WT_ISSUE(p_begin, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  for (int pt = p_begin; pt < p_end; ++pt) {
    const int stage = (pt - p_begin) & 1;
    if (pt + 1 < p_end) WT_ISSUE(pt + 1, stage ^ 1);
    const uint32_t gst = g_base + (uint32_t)(stage * G_STAGE * 4);
    const uint32_t ast = a_base + (uint32_t)(stage * A_STAGE * 4);
#pragma unroll 1
    for (int ks = 0; ks < 8; ++ks) {                  // (rolled: unrolling it hoists 144 addresses and spills)
      // k-step: image row y = ks >> 1, pixels x = 16 (ks & 1) + 8 h + {q, q + 4}
      const int y = ks >> 1, xk = 16 * (ks & 1) + 8 * h + q;
      const int gpx = y * 32 + xk;                                          // (gpx & 3) == q, also for gpx + 4
      const uint32_t gaddr = gst + (uint32_t)(gpx * G_RB + ((g_grp ^ WT_GSW(q)) * 32) + p8);
      const u32x2v g0 = wt_read_tr16(gaddr);
      const u32x2v g1 = wt_read_tr16(gaddr + 4 * G_RB);
      u32x2v a0[9], a1[9];
#pragma unroll
      for (int t = 0; t < 9; ++t) {
        const int hp = (y + t / 3) * WT_HW + (t % 3) + xk;                  // halo pixel of this lane's first row
        a0[t] = wt_read_tr16(ast + (uint32_t)(hp * A_RB + ((a_grp ^ WT_ASW(hp)) * 32) + p8));
        a1[t] = wt_read_tr16(ast + (uint32_t)((hp + 4) * A_RB + ((a_grp ^ WT_ASW(hp + 4)) * 32) + p8));
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
      u32x4v gv;
      gv[0] = g0[0]; gv[1] = g0[1]; gv[2] = g1[0]; gv[3] = g1[1];
      const f16x8v gf = __builtin_bit_cast(f16x8v, gv);
#pragma unroll
      for (int t = 0; t < 9; ++t) {
        u32x4v av;
        av[0] = a0[t][0]; av[1] = a0[t][1]; av[2] = a1[t][0]; av[3] = a1[t][1];
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(gf, __builtin_bit_cast(f16x8v, av), acc[t], 0, 0, 0);
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }
#undef WT_ISSUE
#undef WT_GSW
#undef WT_ASW

  // ---- fold the four plane products (wgrad_s16.hip) and add to the packed gradient: column = tap * Cin + c ------
  const float inv = a.g_inv_scale ? a.g_inv_scale[0] : 1.f;
  constexpr float LO = 1.f / 2048.f;
  const bool a_hi = ((l31 >> 3) & 1) == 0;
  const int c = c0 + 16 * wa + 8 * (l31 >> 4) + (l31 & 7);
#pragma unroll
  for (int t = 0; t < 9; ++t) {
    const int col = t * d.cin + c;
#pragma unroll
    for (int rr = 0; rr < 8; ++rr) {
      const int r = (rr & 3) + 8 * (rr >> 2);
      const float v = acc[t][r] + acc[t][r + 4] * LO;
      const float other = __shfl_xor(v, 8);
      if (a_hi) {
        const int row = row0 + 16 * wg + 8 * (r >> 3) + (r & 3) + 4 * h;
        if (row < d.n && c < d.cin)
          unsafeAtomicAdd(d.dw + (int64_t)row * a.kpad + col, (v + other * LO) * inv);
      }
    }
  }
}

template <int NG, int NA>
static int launch_wgrad_tap(WgradTapArgs a, hipStream_t stream) {
  constexpr int TN = 16 * NG, TC = 16 * NA;
  constexpr int AJ = (WT_HPX * (TC / 4) + WT_NT - 1) / WT_NT;
  constexpr size_t lds = (size_t)(2 * WT_PX * TN + 2 * AJ * WT_NT * 4) * sizeof(float);
  static_assert(lds <= 160 * 1024, "LDS budget");
  auto kern = wgrad_tap_s16_kernel<NG, NA>;
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                     (int)lds);
  if (e != hipSuccess) return (int)e;
  a.row_tiles = (a.d.n + TN - 1) / TN;
  a.col_tiles = a.d.cin / TC;
  const int tiles = a.row_tiles * a.col_tiles;
  int msplit = (2 * 256 + tiles - 1) / tiles;                       // ~2 workgroups per CU, at least 4 patches each
  const int max_split = (a.npatch + 3) / 4;
  if (msplit > max_split) msplit = max_split;
  if (msplit < 1) msplit = 1;
  a.patches_per_block = (a.npatch + msplit - 1) / msplit;
  a.msplit = (a.npatch + a.patches_per_block - 1) / a.patches_per_block;
  hipLaunchKernelGGL(kern, dim3(tiles * a.msplit), dim3(WT_NT), lds, stream, a);
  return ammc_launch_status();
}

// wgrad_tap3_s16.hip: three MFMAs per product block, 128 x 64 channel tiles (N % 128 == 0, Cin % 64 == 0)
int wgrad_tap3_s16_try(const AmmcWgradDesc& d, const float* g_inv_scale, int kpad, hipStream_t stream, float* slabs, int query);

// Called by ammc_conv_wgrad_s16 after its argument checks; -12345 = not this kernel's case.
int wgrad_tap_s16_try(const AmmcWgradDesc& d, const float* g_inv_scale, int kpad, hipStream_t stream) {
  static const int mode = getenv("AMMC_WGRAD_TAP") ? atoi(getenv("AMMC_WGRAD_TAP")) : 1;
  if (!mode) return -12345;
  if (mode != 2) {                                         // AMMC_WGRAD_TAP=2: this file's kernels only (A/Bs)
    const int rc = wgrad_tap3_s16_try(d, g_inv_scale, kpad, stream, nullptr, 0);
    if (rc != -12345) return rc;
  }
  if (d.height % WT_PH || d.width % WT_PW) return -12345;
  const bool wide = d.n % 128 == 0 && d.cin % 16 == 0;
  const bool narrow = d.n % 64 == 0 && d.cin % 32 == 0;
  if (!wide && !narrow) return -12345;
  const int64_t gmax = (int64_t)(WT_PH - 1) * d.g_rs + (int64_t)(WT_PW - 1) * d.g_ps + d.n;
  const int64_t amax = (int64_t)(WT_PH + 1) * d.a_rs + (int64_t)(WT_PW + 1) * d.a_ps + d.cin;
  if (gmax >= (1LL << 30) || amax >= (1LL << 30)) return -12345;
  WgradTapArgs a;
  a.d = d;
  a.g_inv_scale = g_inv_scale;
  a.kpad = kpad;
  a.tiles_x = d.width / WT_PW;
  a.tiles_y = d.height / WT_PH;
  a.npatch = d.batch * a.tiles_x * a.tiles_y;
  return wide ? launch_wgrad_tap<8, 1>(a, stream) : launch_wgrad_tap<4, 2>(a, stream);
}

}  // namespace ammc_s16
