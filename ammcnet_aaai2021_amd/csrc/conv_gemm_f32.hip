// Implicit-GEMM convolution on the fp32 MFMA pipe of gfx950.
//
//   y[m, n] = act(scale[n] * sum_k A[m, k] * Wp[n, k] + shift[n]) + res[m, n]
//
// m = output pixel (b, y, x) of an NHWC tensor, k = tap * Cin + c.  A is never
// materialised: each 128-byte row of an A tile (32 channels of one window tap of one
// pixel) is fetched straight from the halo-padded NHWC input into LDS by
// `global_load_lds_dwordx4` (LDS-DMA, no VGPR round trip).  The zero halo makes
// every tap of every pixel a valid address, so the loader has no bounds checks.
//
// Tiling (one 256-thread workgroup = 4 waves, 2 workgroups per CU):
//   workgroup tile BM x BN x 32, wave tile (TM*32) x (TN*32) built from
//   v_mfma_f32_32x32x2_f32 (exact fp32, 64 FLOP/clk/SIMD).
//   LDS image of a tile: [row][32 floats] = 128-B rows, 16-B slots XOR-swizzled by
//   ((row >> 1) & 7): the DMA writes lane-linear (row = piece / 8, slot = piece % 8),
//   the swizzle is applied to the SOURCE slot and to the ds_read_b128 address, which
//   makes the fragment reads bank-conflict free.
//   A lane's ds_read_b128 yields 4 consecutive k of its row; lane half h takes slot
//   2*kk + h, so MFMA t of step kk contracts k = 8kk+t (h=0) and 8kk+4+t (h=1) -- the
//   same permutation for A and B, hence a plain sum over k.
//   Two LDS stages: the DMA for chunk c+1 is in flight while chunk c is contracted.
//
// Roofline: MFMA-bound for every 3x3 layer of the network (AI 96..1152 flop/B vs the
// fp32 ridge of ~20 flop/B); see DESIGN.md.
#include "ammc_common.h"

namespace ammc_impl {

struct ConvArgs {
  AmmcConvDesc d;
  int M;          // batch * height * width
  int kpad;       // roundup(ntaps * cin, 32)
  int nchunks;    // kpad / 32
  int cin_log2;
  int n_tiles;    // N / BN
};

// DMA of one K chunk (32 k-values of every tile row) into an LDS stage.  adst/bdst are the
// wave-uniform bases of this wave's 1-KiB pieces; the instruction adds lane * 16 B itself.
template <int AJ, int BJ>
__device__ __forceinline__ void issue_chunk(const AmmcConvDesc& d, int cin_log2, int sl, int c,
                                            const float* const (&a_src)[AJ], const float* const (&b_src)[BJ],
                                            float* adst, float* bdst) {
  const int k = c * 32 + 4 * sl;
  int64_t toff;
  if (d.ntaps == 9) {
    int tap = k >> cin_log2;
    tap = tap < 8 ? tap : 8;                     // K padding: weights there are zero
    const int r = (tap * 11) >> 5;               // tap / 3 for tap in [0, 8]
    const int s = tap - 3 * r;
    toff = (int64_t)r * d.x_rs + (int64_t)s * d.x_ps + (k & (d.cin - 1));
  } else if (d.ntaps == 4) {                    // 2x2 window on a 2x-resolution input (ConvTranspose dgrad)
    int tap = k >> cin_log2;
    tap = tap < 3 ? tap : 3;
    toff = (int64_t)(tap >> 1) * d.x_rs + (int64_t)(tap & 1) * d.x_ps + (k & (d.cin - 1));
  } else if (d.ntaps == 16) {                   // 4x4 window (PixelDiscriminator, pix2pix_networks.py:604-621)
    int tap = k >> cin_log2;
    tap = tap < 15 ? tap : 15;
    toff = (int64_t)(tap >> 2) * d.x_rs + (int64_t)(tap & 3) * d.x_ps + (k & (d.cin - 1));
  } else {
    toff = k;
  }
#pragma unroll
  for (int j = 0; j < AJ; ++j)
    __builtin_amdgcn_global_load_lds(a_src[j] + toff, adst + j * 1024, 16, 0, 0);
#pragma unroll
  for (int j = 0; j < BJ; ++j)
    __builtin_amdgcn_global_load_lds(b_src[j] + c * 32, bdst + j * 1024, 16, 0, 0);
}

template <int WGM, int WGN, int TM, int TN>
__global__ __launch_bounds__(256, 2) void conv_gemm_f32_kernel(ConvArgs a) {
  constexpr int BM = WGM * TM * 32;
  constexpr int BN = WGN * TN * 32;
  constexpr int A_STAGE = BM * 32;          // floats per A stage
  constexpr int B_STAGE = BN * 32;
  constexpr int AJ = BM / 32;               // 16-B pieces per thread per stage (A)
  constexpr int BJ = BN / 32;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* As = smem;                          // [2][BM][32]
  float* Bs = smem + 2 * A_STAGE;            // [2][BN][32]

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave / WGN;
  const int wn = wave % WGN;
  const int h = lane >> 5;
  const int l31 = lane & 31;

  const int logical = ammc_xcd_remap(blockIdx.x, gridDim.x);
  const int n0 = (logical % a.n_tiles) * BN;
  const int m0 = (logical / a.n_tiles) * BM;

  const AmmcConvDesc& d = a.d;
  const int W = d.width, H = d.height;

  // ---- per-thread source addresses of its DMA pieces --------------------------
  // piece p = j*256 + tid  ->  tile row j*32 + (tid >> 3), physical slot tid & 7;
  // the logical slot it must fetch is (tid & 7) ^ ((row >> 1) & 7) = (tid&7) ^ ((tid>>4)&7).
  const int sl = (tid & 7) ^ ((tid >> 4) & 7);
  const int xstep = d.x_step > 1 ? d.x_step : 1;
  const float* a_src[AJ];
#pragma unroll
  for (int j = 0; j < AJ; ++j) {
    int m = m0 + j * 32 + (tid >> 3);
    m = m < a.M ? m : a.M - 1;
    const int x = m % W;
    const int t = m / W;
    const int y = t % H;
    const int b = t / H;
    a_src[j] = d.x + ((int64_t)b * d.x_bs + (int64_t)(y * xstep) * d.x_rs + (int64_t)(x * xstep) * d.x_ps);
  }
  const float* b_src[BJ];
#pragma unroll
  for (int j = 0; j < BJ; ++j)
    b_src[j] = d.w + (int64_t)(n0 + j * 32 + (tid >> 3)) * a.kpad + 4 * sl;

#define ISSUE_CHUNK(c, stage) \
  issue_chunk<AJ, BJ>(d, a.cin_log2, sl, (c), a_src, b_src, As + (stage) * A_STAGE + wave * 256, \
                      Bs + (stage) * B_STAGE + wave * 256)

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // fragment read addresses (floats): row * 32 + ((2kk + h) ^ swz) * 4
  const int swz = (l31 >> 1) & 7;
  const int a_row = (wm * TM * 32 + l31) * 32;
  const int b_row = (wn * TN * 32 + l31) * 32;

  ISSUE_CHUNK(0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  for (int c = 0; c < a.nchunks; ++c) {
    const int stage = c & 1;
    if (c + 1 < a.nchunks) ISSUE_CHUNK(c + 1, stage ^ 1);
    const float* Ac = As + stage * A_STAGE + a_row;
    const float* Bc = Bs + stage * B_STAGE + b_row;
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      const int so = (((2 * kk + h) ^ swz) << 2);
      f32x4 af[TM], bf[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) af[i] = *reinterpret_cast<const f32x4*>(Ac + i * 1024 + so);
#pragma unroll
      for (int j = 0; j < TN; ++j) bf[j] = *reinterpret_cast<const f32x4*>(Bc + j * 1024 + so);
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i][t], bf[j][t], acc[i][j], 0, 0, 0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }

  // ---- epilogue -----------------------------------------------------------------
  // pixel -> output / residual offsets, once per tile row, in the (now idle) A stage
  int* tab_out = reinterpret_cast<int*>(smem);
  int* tab_res = tab_out + BM;
  for (int i = tid; i < BM; i += 256) {
    const int m = m0 + i;
    int o = -1, r = 0;
    if (m < a.M) {
      const int x = m % W;
      const int t = m / W;
      const int y = t % H;
      const int b = t / H;
      o = (int)((int64_t)b * d.y_bs + (int64_t)(y * d.up) * d.y_rs + (int64_t)(x * d.up) * d.y_ps);
      r = (int)((int64_t)b * d.r_bs + (int64_t)y * d.r_rs + (int64_t)x * d.r_ps);
    }
    tab_out[i] = o;
    tab_res[i] = r;
  }
  __syncthreads();

  const int b_first = (int)((int64_t)m0 / ((int64_t)H * W));
  float sq0 = 0.f;
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int ncol = n0 + (wn * TN + j) * 32 + l31;
    const float sc = d.scale ? d.scale[ncol] : 1.f;
    const float sh = d.shift ? d.shift[ncol] : 0.f;
    int co = ncol;
    const int nstore = d.n_store > 0 ? d.n_store : d.n;
    const int64_t ycs = d.y_cs > 0 ? d.y_cs : 1;
    int goff = 0;
    if (d.up == 2) {
      const int g = ncol / d.cgroup;
      co = ncol - g * d.cgroup;
      goff = (int)((g >> 1) * d.y_rs + (g & 1) * d.y_ps);
    }
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = (wm * TM + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
        const int o = tab_out[row];
        if (o >= 0 && co < nstore) {
          float v = acc[i][j][r] * sc + sh;
          if (d.act == AMMC_ACT_RELU) v = v > 0.f ? v : 0.f;
          else if (d.act == AMMC_ACT_TANH) v = tanhf(v);
          else if (d.act == AMMC_ACT_LRELU) v = v > 0.f ? v : 0.1f * v;
          if (d.res) v += d.res[tab_res[row] + co];
          const int64_t addr = o + goff + (int64_t)co * ycs;
          d.y[addr] = v;
          if (d.sq_target) {                       // fused squared error of `psnr_error` (outc only)
            const float df = 0.5f * (d.sq_target[addr] - v);
            const int bs = (int)(((int64_t)m0 + row) / ((int64_t)H * W));
            if (bs == b_first) sq0 += df * df;
            else unsafeAtomicAdd(d.sq_acc + bs, df * df);
          }
        }
      }
    }
  }
  if (d.sq_target) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) sq0 += __shfl_xor(sq0, off);
    if (lane == 0) unsafeAtomicAdd(d.sq_acc + b_first, sq0);
  }
}

template <int WGM, int WGN, int TM, int TN>
int launch(const ConvArgs& a, hipStream_t stream) {
  constexpr int BM = WGM * TM * 32;
  constexpr int BN = WGN * TN * 32;
  constexpr size_t lds = (size_t)(2 * BM * 32 + 2 * BN * 32) * sizeof(float);
  static bool attr_done = false;
  auto kern = conv_gemm_f32_kernel<WGM, WGN, TM, TN>;
  if (!attr_done && lds > 48 * 1024) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return (int)e;
    attr_done = true;
  }
  ConvArgs b = a;
  b.n_tiles = a.d.n / BN;
  const int m_tiles = (a.M + BM - 1) / BM;
  const int grid = m_tiles * b.n_tiles;
  hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, stream, b);
  return ammc_launch_status();
}

}  // namespace ammc_impl
using namespace ammc_impl;

extern "C" int ammc_conv_gemm_f32(const AmmcConvDesc* desc, void* stream) {
  if (!desc || !desc->x || !desc->w || !desc->y) return AMMC_EINVAL;
  const AmmcConvDesc& d = *desc;
  if (d.batch <= 0 || d.height <= 0 || d.width <= 0) return AMMC_EINVAL;
  if (d.ntaps != 9 && d.ntaps != 1 && d.ntaps != 4 && d.ntaps != 16) return AMMC_EINVAL;
  if (d.ntaps != 1 && (d.cin < 4 || (d.cin & (d.cin - 1)))) return AMMC_EUNSUP;   // power of two
  if (d.x_step < 0 || d.x_step > 2) return AMMC_EINVAL;
  if (d.ntaps == 1 && (d.cin <= 0 || d.cin % 32)) return AMMC_EUNSUP;
  if (d.n <= 0 || (d.n != 32 && (d.n % 64))) return AMMC_EUNSUP;
  if (d.n_store < 0 || d.n_store > d.n || d.y_cs < 0) return AMMC_EINVAL;
  if (d.up != 1 && d.up != 2) return AMMC_EINVAL;
  if (d.up == 2 && (d.cgroup <= 0 || d.cgroup % 32 || d.n != 4 * d.cgroup)) return AMMC_EINVAL;
  if (((uintptr_t)d.x | (uintptr_t)d.w) & 15) return AMMC_EINVAL;   // 16-B DMA pieces
  if ((d.x_bs | d.x_rs | d.x_ps) & 3) return AMMC_EINVAL;
  const int64_t M = (int64_t)d.batch * d.height * d.width;
  if (M >= (1LL << 31)) return AMMC_EUNSUP;
  // output / residual offsets are kept as int32 inside the kernel
  const int64_t ymax = (int64_t)d.batch * d.y_bs + (int64_t)d.height * d.up * d.y_rs;
  const int64_t rmax = (int64_t)d.batch * d.r_bs + (int64_t)d.height * d.r_rs;
  if (ymax >= (1LL << 31) || rmax >= (1LL << 31)) return AMMC_EUNSUP;

  ConvArgs a;
  a.d = d;
  a.M = (int)M;
  a.kpad = ((d.ntaps * d.cin + 31) / 32) * 32;
  a.nchunks = a.kpad / 32;
  a.cin_log2 = ammc_ilog2(d.cin);
  a.n_tiles = 0;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  if (d.n == 32) return launch<4, 1, 1, 1>(a, s);        // 128 x 32,  waves 4x1 of 32x32 (small-N layers)
  if (d.n % 128 == 0) return launch<2, 2, 2, 2>(a, s);   // 128 x 128, waves 2x2 of 64x64
  return launch<4, 1, 1, 2>(a, s);                        // 128 x 64,  waves 4x1 of 32x64
}
