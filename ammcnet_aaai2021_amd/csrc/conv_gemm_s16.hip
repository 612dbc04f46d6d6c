// Implicit-GEMM convolution with fp32-equivalent arithmetic on the fp16 MFMA pipe.
//
// Every fp32 value v is carried as a pair of halfs (hi, lo):  hi = half(v),
// lo = half((v - hi) * 2^11), i.e. v = hi + lo * 2^-11 to 22 significant bits (fp32 has 24).
// A product a*b is evaluated as  hi_a*hi_b + (hi_a*lo_b + lo_a*hi_b) * 2^-11  with three
// v_mfma_f32_32x32x16_f16 per 16-deep k-step and fp32 accumulation in two accumulator sets
// (the dropped lo*lo term is 2^-22 relative).  Measured against an fp64 evaluation of the whole
// network this is as accurate as native fp32 (1.6e-6 vs 1.7e-6 relative on the predicted
// frames), at 16/3 of the fp32 MFMA rate.  SURVEY.md section 7 names this scheme ("fp32 MFMA or a
// split-precision scheme") for the parity configs; the exact-fp32 kernel stays available.
//
// Storage ("S16"): NHWC with the channels in groups of 8, each group 32 bytes = [8 hi | 8 lo].
// 4 bytes per element like fp32, so strides, halos, the 128-byte LDS rows (32 k-values), the
// XOR swizzle and the LDS-DMA addressing are those of conv_gemm_f32.hip; only the fragment
// reads, the MFMAs and the epilogue differ.  A lane's A/B fragment for k-step s is channel
// group 2s + (lane >> 5): its hi and lo slots are two conflict-free ds_read_b128.
//
// Epilogue: the accumulators have channels on lanes, so a direct S16 store would be 2-byte
// scattered writes.  Instead the wave parks its tile in LDS (the DMA stages are idle by then)
// and every thread then finishes 8 consecutive channels of one pixel: scale/shift, ReLU,
// residual (S16), split, one 32-byte store.  fp32 outputs (`y_f32`: the encoder output that
// feeds the memory kernel, the NCHW `outc` frames) keep the direct per-lane store.
#include "ammc_common.h"
#include <hip/hip_fp16.h>

namespace ammc_s16 {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

struct ConvArgs {
  AmmcConvDesc d;
  int M, kpad, nchunks, cin_log2, n_tiles;
};

constexpr float LO_SCALE = 2048.f;
constexpr float LO_INV = 1.f / 2048.f;

__device__ __forceinline__ void split8(const float (&v)[8], f16x8& hi, f16x8& lo) {
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const _Float16 hv = (_Float16)v[i];
    hi[i] = hv;
    lo[i] = (_Float16)((v[i] - (float)hv) * LO_SCALE);
  }
}

__device__ __forceinline__ void join8(const f16x8& hi, const f16x8& lo, float (&v)[8]) {
#pragma unroll
  for (int i = 0; i < 8; ++i) v[i] = (float)hi[i] + (float)lo[i] * LO_INV;
}

template <int AJ, int BJ>
__device__ __forceinline__ void issue_chunk(const AmmcConvDesc& d, int cin_log2, int sl, int c,
                                            const float* const (&a_src)[AJ], const float* const (&b_src)[BJ],
                                            float* adst, float* bdst) {
  const int k = c * 32 + 4 * sl;
  int64_t toff;
  if (d.ntaps == 9) {
    int tap = k >> cin_log2;
    tap = tap < 8 ? tap : 8;
    const int r = (tap * 11) >> 5;
    const int s = tap - 3 * r;
    toff = (int64_t)r * d.x_rs + (int64_t)s * d.x_ps + (k & (d.cin - 1));
  } else if (d.ntaps == 4) {
    int tap = k >> cin_log2;
    tap = tap < 3 ? tap : 3;
    toff = (int64_t)(tap >> 1) * d.x_rs + (int64_t)(tap & 1) * d.x_ps + (k & (d.cin - 1));
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
__global__ __launch_bounds__(256, 2) void conv_gemm_s16_kernel(ConvArgs a) {
  constexpr int BM = WGM * TM * 32;
  constexpr int BN = WGN * TN * 32;
  constexpr int A_STAGE = BM * 32;
  constexpr int B_STAGE = BN * 32;
  constexpr int AJ = BM / 32;
  constexpr int BJ = BN / 32;
  constexpr int STAGES = 2 * A_STAGE + 2 * B_STAGE;              // floats
  constexpr int TILE = BM * BN;                                  // floats of the parked output tile
  constexpr int REGION = STAGES > TILE ? STAGES : TILE;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* As = smem;
  float* Bs = smem + 2 * A_STAGE;
  int* tab_out = reinterpret_cast<int*>(smem + REGION);          // [BM]
  int* tab_res = tab_out + BM;

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

  // output / residual pixel offsets of the tile rows (used by both epilogues)
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

#define S16_ISSUE(c, stage) \
  issue_chunk<AJ, BJ>(d, a.cin_log2, sl, (c), a_src, b_src, As + (stage) * A_STAGE + wave * 256, \
                      Bs + (stage) * B_STAGE + wave * 256)

  f32x16 hh[TM][TN], xx[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) { hh[i][j][r] = 0.f; xx[i][j][r] = 0.f; }

  const int swz = (l31 >> 1) & 7;
  const int a_row = (wm * TM * 32 + l31) * 32;
  const int b_row = (wn * TN * 32 + l31) * 32;

  S16_ISSUE(0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  for (int c = 0; c < a.nchunks; ++c) {
    const int stage = c & 1;
    if (c + 1 < a.nchunks) S16_ISSUE(c + 1, stage ^ 1);
    const float* Ac = As + stage * A_STAGE + a_row;
    const float* Bc = Bs + stage * B_STAGE + b_row;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      const int g = 2 * s + h;                                   // channel group of this lane half
      const int so_hi = ((2 * g) ^ swz) << 2;
      const int so_lo = ((2 * g + 1) ^ swz) << 2;
      f16x8 ah[TM], al[TM], bh[TN], bl[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        ah[i] = *reinterpret_cast<const f16x8*>(Ac + i * 1024 + so_hi);
        al[i] = *reinterpret_cast<const f16x8*>(Ac + i * 1024 + so_lo);
      }
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        bh[j] = *reinterpret_cast<const f16x8*>(Bc + j * 1024 + so_hi);
        bl[j] = *reinterpret_cast<const f16x8*>(Bc + j * 1024 + so_lo);
      }
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          hh[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bh[j], hh[i][j], 0, 0, 0);
          xx[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bl[j], xx[i][j], 0, 0, 0);
          xx[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[i], bh[j], xx[i][j], 0, 0, 0);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }
#undef S16_ISSUE

  const int nstore = d.n_store > 0 ? d.n_store : d.n;

  if (d.y_f32) {
    // ---- direct fp32 store (channels on lanes), NHWC or NCHW through y_cs ------------------
    const int64_t ycs = d.y_cs > 0 ? d.y_cs : 1;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int ncol = n0 + (wn * TN + j) * 32 + l31;
      const float sc = d.scale ? d.scale[ncol] : 1.f;
      const float sh = d.shift ? d.shift[ncol] : 0.f;
#pragma unroll
      for (int i = 0; i < TM; ++i) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = (wm * TM + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
          const int o = tab_out[row];
          if (o >= 0 && ncol < nstore) {
            float v = (hh[i][j][r] + xx[i][j][r] * LO_INV) * sc + sh;
            if (d.act == AMMC_ACT_RELU) v = v > 0.f ? v : 0.f;
            else if (d.act == AMMC_ACT_TANH) v = tanhf(v);
            d.y[o + (int64_t)ncol * ycs] = v;
          }
        }
      }
    }
    return;
  }

  // ---- S16 store: park the tile in LDS, then 8 channels of one pixel per thread ---------------
  float* T = smem;                                               // [BM][BN]
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int col = (wn * TN + j) * 32 + l31;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = (wm * TM + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
        T[row * BN + col] = hh[i][j][r] + xx[i][j][r] * LO_INV;
      }
    }
  }
  __syncthreads();
  constexpr int CG = BN / 8;                                     // channel groups per tile row
  for (int item = tid; item < BM * CG; item += 256) {
    const int row = item / CG;
    const int cg = item - row * CG;
    const int o = tab_out[row];
    if (o < 0) continue;
    const int ncol0 = n0 + cg * 8;
    float v[8];
    {
      const f32x4 t0 = *reinterpret_cast<const f32x4*>(T + row * BN + cg * 8);
      const f32x4 t1 = *reinterpret_cast<const f32x4*>(T + row * BN + cg * 8 + 4);
#pragma unroll
      for (int i = 0; i < 4; ++i) { v[i] = t0[i]; v[4 + i] = t1[i]; }
    }
    if (d.scale) {
      const f32x4 s0 = *reinterpret_cast<const f32x4*>(d.scale + ncol0);
      const f32x4 s1 = *reinterpret_cast<const f32x4*>(d.scale + ncol0 + 4);
#pragma unroll
      for (int i = 0; i < 4; ++i) { v[i] *= s0[i]; v[4 + i] *= s1[i]; }
    }
    if (d.shift) {
      const f32x4 s0 = *reinterpret_cast<const f32x4*>(d.shift + ncol0);
      const f32x4 s1 = *reinterpret_cast<const f32x4*>(d.shift + ncol0 + 4);
#pragma unroll
      for (int i = 0; i < 4; ++i) { v[i] += s0[i]; v[4 + i] += s1[i]; }
    }
    if (d.act == AMMC_ACT_RELU) {
#pragma unroll
      for (int i = 0; i < 8; ++i) v[i] = v[i] > 0.f ? v[i] : 0.f;
    }
    int co = ncol0, goff = 0;
    if (d.up == 2) {
      const int g = ncol0 / d.cgroup;
      co = ncol0 - g * d.cgroup;
      goff = (int)((g >> 1) * d.y_rs + (g & 1) * d.y_ps);
    }
    if (d.res) {
      const float* rp = d.res + tab_res[row] + co;
      float rv[8];
      join8(*reinterpret_cast<const f16x8*>(rp), *reinterpret_cast<const f16x8*>(rp + 4), rv);
#pragma unroll
      for (int i = 0; i < 8; ++i) v[i] += rv[i];
    }
    f16x8 hi, lo;
    split8(v, hi, lo);
    float* yp = d.y + o + goff + co;
    *reinterpret_cast<f16x8*>(yp) = hi;
    *reinterpret_cast<f16x8*>(yp + 4) = lo;
  }
}

template <int WGM, int WGN, int TM, int TN>
int launch(const ConvArgs& a, hipStream_t stream) {
  constexpr int BM = WGM * TM * 32;
  constexpr int BN = WGN * TN * 32;
  constexpr int STAGES = 2 * BM * 32 + 2 * BN * 32;
  constexpr int TILE = BM * BN;
  constexpr size_t lds = (size_t)((STAGES > TILE ? STAGES : TILE) + 2 * BM) * sizeof(float);
  auto kern = conv_gemm_s16_kernel<WGM, WGN, TM, TN>;
  if (lds > 48 * 1024) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return (int)e;
  }
  ConvArgs b = a;
  b.n_tiles = a.d.n / BN;
  const int m_tiles = (a.M + BM - 1) / BM;
  hipLaunchKernelGGL(kern, dim3(m_tiles * b.n_tiles), dim3(256), lds, stream, b);
  return ammc_launch_status();
}

// fp32 [rows][cols] (cols % 8 == 0) -> S16 groups [8 hi | 8 lo], same shape in bytes
__global__ __launch_bounds__(256) void split_rows_kernel(const float* __restrict__ src, int64_t ngroups,
                                                         float* __restrict__ dst) {
  const int64_t g = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (g >= ngroups) return;
  const f32x4 a0 = *reinterpret_cast<const f32x4*>(src + g * 8);
  const f32x4 a1 = *reinterpret_cast<const f32x4*>(src + g * 8 + 4);
  float v[8];
#pragma unroll
  for (int i = 0; i < 4; ++i) { v[i] = a0[i]; v[4 + i] = a1[i]; }
  f16x8 hi, lo;
  split8(v, hi, lo);
  *reinterpret_cast<f16x8*>(dst + g * 8) = hi;
  *reinterpret_cast<f16x8*>(dst + g * 8 + 4) = lo;
}

// NCHW fp32 -> S16 NHWC (explicit strides), channels c..cp-1 zero; one thread per (pixel, group of 8)
__global__ __launch_bounds__(256) void nchw_to_s16_kernel(const float* __restrict__ x, int B, int C, int H, int W,
                                                          float* __restrict__ y, int64_t y_bs, int64_t y_rs,
                                                          int64_t y_ps, int groups) {
  const int64_t total = (int64_t)B * groups * H * W;
  const int64_t gid = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (gid >= total) return;
  const int xw = (int)(gid % W);
  int64_t t = gid / W;
  const int yh = (int)(t % H);
  t /= H;
  const int g = (int)(t % groups);
  const int b = (int)(t / groups);
  float v[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int c = g * 8 + i;
    v[i] = c < C ? x[(((int64_t)b * C + c) * H + yh) * W + xw] : 0.f;
  }
  f16x8 hi, lo;
  split8(v, hi, lo);
  float* dst = y + (int64_t)b * y_bs + (int64_t)yh * y_rs + (int64_t)xw * y_ps + g * 8;
  *reinterpret_cast<f16x8*>(dst) = hi;
  *reinterpret_cast<f16x8*>(dst + 4) = lo;
}

// S16 NHWC (strided) -> fp32 NCHW (tests / debugging / views across the module boundary)
__global__ __launch_bounds__(256) void s16_to_nchw_kernel(const float* __restrict__ x, int64_t x_bs, int64_t x_rs,
                                                          int64_t x_ps, int B, int C, int H, int W,
                                                          float* __restrict__ y) {
  const int groups = C >> 3;
  const int64_t total = (int64_t)B * groups * H * W;
  const int64_t gid = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (gid >= total) return;
  const int xw = (int)(gid % W);
  int64_t t = gid / W;
  const int yh = (int)(t % H);
  t /= H;
  const int g = (int)(t % groups);
  const int b = (int)(t / groups);
  const float* src = x + (int64_t)b * x_bs + (int64_t)yh * x_rs + (int64_t)xw * x_ps + g * 8;
  float v[8];
  join8(*reinterpret_cast<const f16x8*>(src), *reinterpret_cast<const f16x8*>(src + 4), v);
#pragma unroll
  for (int i = 0; i < 8; ++i) y[(((int64_t)b * C + g * 8 + i) * H + yh) * W + xw] = v[i];
}

// MaxPool2d(2) on S16: compare the decoded values, copy the winner's (hi, lo) bits
__global__ __launch_bounds__(256) void maxpool2x2_s16_kernel(const float* __restrict__ x, int64_t x_bs, int64_t x_rs,
                                                             int64_t x_ps, float* __restrict__ y, int64_t y_bs,
                                                             int64_t y_rs, int64_t y_ps, int B, int h, int w, int G) {
  const int64_t total = (int64_t)B * h * w * G;
  const int64_t gid = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (gid >= total) return;
  const int g = (int)(gid % G);
  int64_t t = gid / G;
  const int xx = (int)(t % w);
  t /= w;
  const int yy = (int)(t % h);
  const int b = (int)(t / h);
  const float* s = x + (int64_t)b * x_bs + (int64_t)(2 * yy) * x_rs + (int64_t)(2 * xx) * x_ps + g * 8;
  f16x8 hi[4], lo[4];
  const int64_t offs[4] = {0, x_ps, x_rs, x_rs + x_ps};
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    hi[p] = *reinterpret_cast<const f16x8*>(s + offs[p]);
    lo[p] = *reinterpret_cast<const f16x8*>(s + offs[p] + 4);
  }
  f16x8 oh, ol;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    float best = (float)hi[0][i] + (float)lo[0][i] * LO_INV;
    _Float16 bh = hi[0][i], bl = lo[0][i];
#pragma unroll
    for (int p = 1; p < 4; ++p) {
      const float v = (float)hi[p][i] + (float)lo[p][i] * LO_INV;
      if (v > best) { best = v; bh = hi[p][i]; bl = lo[p][i]; }
    }
    oh[i] = bh;
    ol[i] = bl;
  }
  float* dst = y + (int64_t)b * y_bs + (int64_t)yy * y_rs + (int64_t)xx * y_ps + g * 8;
  *reinterpret_cast<f16x8*>(dst) = oh;
  *reinterpret_cast<f16x8*>(dst + 4) = ol;
}

inline unsigned nblk(int64_t total) { return (unsigned)((total + 255) / 256); }

}  // namespace ammc_s16
using namespace ammc_s16;

extern "C" int ammc_conv_gemm_s16(const AmmcConvDesc* desc, void* stream) {
  if (!desc || !desc->x || !desc->w || !desc->y) return AMMC_EINVAL;
  const AmmcConvDesc& d = *desc;
  if (d.batch <= 0 || d.height <= 0 || d.width <= 0) return AMMC_EINVAL;
  if (d.ntaps != 9 && d.ntaps != 1 && d.ntaps != 4) return AMMC_EINVAL;
  if (d.ntaps != 1 && (d.cin < 8 || (d.cin & (d.cin - 1)))) return AMMC_EUNSUP;    // whole groups of 8
  if (d.x_step < 0 || d.x_step > 2) return AMMC_EINVAL;
  if (d.ntaps == 1 && (d.cin <= 0 || d.cin % 32)) return AMMC_EUNSUP;
  if (d.n <= 0 || (d.n != 32 && (d.n % 64))) return AMMC_EUNSUP;
  if (d.n_store < 0 || d.n_store > d.n || d.y_cs < 0) return AMMC_EINVAL;
  if (d.up != 1 && d.up != 2) return AMMC_EINVAL;
  if (d.up == 2 && (d.cgroup <= 0 || d.cgroup % 32 || d.n != 4 * d.cgroup)) return AMMC_EINVAL;
  if (d.y_f32 && (d.res || d.up != 1)) return AMMC_EUNSUP;
  if (!d.y_f32 && (d.n_store || d.y_cs > 1 || d.act == AMMC_ACT_TANH)) return AMMC_EUNSUP;
  if (((uintptr_t)d.x | (uintptr_t)d.w) & 15) return AMMC_EINVAL;
  if (!d.y_f32 && (((uintptr_t)d.y & 31) || ((d.y_bs | d.y_rs | d.y_ps) & 7))) return AMMC_EINVAL;
  if (d.res && (((uintptr_t)d.res & 31) || ((d.r_bs | d.r_rs | d.r_ps) & 7))) return AMMC_EINVAL;
  if ((d.x_bs | d.x_rs | d.x_ps) & 7) return AMMC_EINVAL;
  const int64_t M = (int64_t)d.batch * d.height * d.width;
  if (M >= (1LL << 31)) return AMMC_EUNSUP;
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
  if (d.n == 32) return launch<4, 1, 1, 1>(a, s);
  if (d.n % 128 == 0) return launch<2, 2, 2, 2>(a, s);
  return launch<4, 1, 1, 2>(a, s);
}

extern "C" int ammc_split_rows_f32(const float* src, int64_t count, float* dst, void* stream) {
  if (!src || !dst || count <= 0 || (count & 7) || (((uintptr_t)src | (uintptr_t)dst) & 15)) return AMMC_EINVAL;
  hipLaunchKernelGGL(split_rows_kernel, dim3(nblk(count >> 3)), dim3(256), 0, (hipStream_t)stream, src, count >> 3, dst);
  return ammc_launch_status();
}

extern "C" int ammc_nchw_to_s16_f32(const float* x, int32_t batch, int32_t c, int32_t h, int32_t w, float* y,
                                    int64_t y_bs, int64_t y_rs, int64_t y_ps, int32_t cp, void* stream) {
  if (!x || !y || batch <= 0 || c <= 0 || h <= 0 || w <= 0 || cp < c || (cp & 7)) return AMMC_EINVAL;
  if (((uintptr_t)y & 31) || ((y_bs | y_rs | y_ps) & 7)) return AMMC_EINVAL;
  const int64_t total = (int64_t)batch * (cp >> 3) * h * w;
  hipLaunchKernelGGL(nchw_to_s16_kernel, dim3(nblk(total)), dim3(256), 0, (hipStream_t)stream, x, batch, c, h, w, y,
                     y_bs, y_rs, y_ps, cp >> 3);
  return ammc_launch_status();
}

extern "C" int ammc_s16_to_nchw_f32(const float* x, int64_t x_bs, int64_t x_rs, int64_t x_ps, int32_t batch, int32_t c,
                                    int32_t h, int32_t w, float* y, void* stream) {
  if (!x || !y || batch <= 0 || c <= 0 || (c & 7) || h <= 0 || w <= 0) return AMMC_EINVAL;
  const int64_t total = (int64_t)batch * (c >> 3) * h * w;
  hipLaunchKernelGGL(s16_to_nchw_kernel, dim3(nblk(total)), dim3(256), 0, (hipStream_t)stream, x, x_bs, x_rs, x_ps,
                     batch, c, h, w, y);
  return ammc_launch_status();
}

extern "C" int ammc_maxpool2x2_s16(const float* x, int64_t x_bs, int64_t x_rs, int64_t x_ps, float* y, int64_t y_bs,
                                   int64_t y_rs, int64_t y_ps, int32_t batch, int32_t h, int32_t w, int32_t c,
                                   void* stream) {
  if (!x || !y || batch <= 0 || h <= 0 || w <= 0 || c <= 0 || (c & 7)) return AMMC_EINVAL;
  if ((x_bs | x_rs | x_ps | y_bs | y_rs | y_ps) & 7) return AMMC_EINVAL;
  const int64_t total = (int64_t)batch * h * w * (c >> 3);
  hipLaunchKernelGGL(maxpool2x2_s16_kernel, dim3(nblk(total)), dim3(256), 0, (hipStream_t)stream, x, x_bs, x_rs, x_ps,
                     y, y_bs, y_rs, y_ps, batch, h, w, c >> 3);
  return ammc_launch_status();
}
