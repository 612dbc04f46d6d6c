// Memory addressing of the memory-consistency module, forward
// (`Quantize_topk.forward`, reference Code/models/unet.py:282-297, 310-313).
//
//   dist[r][s] = (|x_r|^2 - 2 x_r . E_s) + |E_s|^2        (same expression order as :284-288)
//   idx_topk[r] = the K smallest dist, nearest first       (:289, :293)
//   q_topk[r]  = [E_idx0 | E_idx1 | ...]                   (:294-297)
//   q_one[r]   = x_r + (E_idx0 - x_r)                      (:311)
//   diff       = mean((E_idx0 - x_r)^2)                    (:310)   -> per-block partial sums
//
// One fused kernel; the [n x m] distance matrix lives only in MFMA accumulators.
//   - a workgroup owns 128 feature rows: the x tile sits in LDS (16-B slots XOR-swizzled
//     by row so the ds_read_b128 fragment reads are conflict free);
//   - its 4 waves split the slots: wave w contracts slot tiles w, w+4, ... (32 slots each)
//     against all 128 rows with v_mfma_f32_32x32x2_f32.  Slots are the MFMA rows and
//     features the MFMA columns, so one lane sees 16 slots of ONE feature row per
//     accumulator tile and keeps that row's running top-K in registers;
//   - E is streamed from its native [d][m] layout (a lane reads E[dd][slot0 + lane%32]:
//     128-B coalesced segments), one dword per lane per four MFMAs, prefetched a group
//     ahead; nothing is staged twice;
//   - the 8 partial top-K lists of a row (4 waves x 2 lane halves) are merged through LDS
//     with ties broken on the lower slot index, then the K codebook rows are gathered
//     from the slot-major copy [m][d] with 16-B accesses.
// Roofline: MFMA (AI = 2*d*m / (4*(d + k*d + k)) flop/B per row, i.e. hundreds).
#include "ammc_common.h"
#include <math.h>

namespace ammc_impl {

constexpr int BR = 32;      // feature rows per workgroup (32: two workgroups per CU at batch 16, whose MFMA and top-k VALU phases overlap)
constexpr int RT = BR / 32;

template <int K>
__device__ __forceinline__ void topk_insert(float (&v)[K], int (&ix)[K], float c, int s) {
  // keep (v, ix) sorted ascending by (value, index)
  if (c < v[K - 1] || (c == v[K - 1] && s < ix[K - 1])) {
    v[K - 1] = c;
    ix[K - 1] = s;
#pragma unroll
    for (int j = K - 1; j > 0; --j) {
      const bool sw = v[j] < v[j - 1] || (v[j] == v[j - 1] && ix[j] < ix[j - 1]);
      const float tv = sw ? v[j - 1] : v[j];
      const int ti = sw ? ix[j - 1] : ix[j];
      v[j - 1] = sw ? v[j] : v[j - 1];
      ix[j - 1] = sw ? ix[j] : ix[j - 1];
      v[j] = tv;
      ix[j] = ti;
    }
  }
}

// The same insertion when candidates arrive in increasing slot order (one lane's walk over the slot tiles): a tie then
// always loses to the entry already held, so strict comparisons implement the (value, index) order - branch-free for
// the model's K = 2 (the branchy general form was 40 % of the kernel at 2000 slots).
template <int K>
__device__ __forceinline__ void topk_insert_ordered(float (&v)[K], int (&ix)[K], float c, int s) {
  if (K == 2) {
    const bool lt0 = c < v[0], lt1 = c < v[1];
    v[1] = lt0 ? v[0] : (lt1 ? c : v[1]);
    ix[1] = lt0 ? ix[0] : (lt1 ? s : ix[1]);
    v[0] = lt0 ? c : v[0];
    ix[0] = lt0 ? s : ix[0];
  } else {
    topk_insert<K>(v, ix, c, s);
  }
}

template <int K>
__global__ __launch_bounds__(256, 2) void memory_topk_kernel(
    const float* __restrict__ x, const float* __restrict__ e_dm, const float* __restrict__ e_md,
    const float* __restrict__ enorm, int n, int d, int m,
    int* __restrict__ idx_out, float* __restrict__ q_topk, float* __restrict__ q_one,
    float* __restrict__ diff_partial) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* xs = smem;                                   // [BR][d], swizzled 16-B slots
  float* xx = smem + BR * d;                          // [BR] |x|^2
  float* cand_v = xx + BR;                            // [BR][8][K]
  int* cand_i = reinterpret_cast<int*>(cand_v + BR * 8 * K);
  int* best = cand_i + BR * 8 * K;                    // [BR][K]
  float* red = reinterpret_cast<float*>(best + BR * K);   // [256]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int h = lane >> 5, l31 = lane & 31;
  const int r0 = blockIdx.x * BR;
  const int slots16 = d >> 2;                         // 16-B slots per row

  // ---- stage the x tile -------------------------------------------------------------
  for (int p = tid; p < BR * slots16; p += 256) {
    const int row = p / slots16, sl = p - row * slots16;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (r0 + row < n) v = *reinterpret_cast<const f32x4*>(x + (int64_t)(r0 + row) * d + sl * 4);
    *reinterpret_cast<f32x4*>(xs + row * d + ((sl ^ (row & 15)) << 2)) = v;
  }
  __syncthreads();
  if (tid < BR) {
    float s = 0.f;
    for (int sl = 0; sl < slots16; ++sl) {
      const f32x4 v = *reinterpret_cast<const f32x4*>(xs + tid * d + ((sl ^ (tid & 15)) << 2));
      s += v[0] * v[0];
      s += v[1] * v[1];
      s += v[2] * v[2];
      s += v[3] * v[3];
    }
    xx[tid] = s;
  }
  __syncthreads();

  float bv[RT][K];
  int bi[RT][K];
#pragma unroll
  for (int t = 0; t < RT; ++t)
#pragma unroll
    for (int j = 0; j < K; ++j) { bv[t][j] = INFINITY; bi[t][j] = 0x7fffffff; }
  float xnorm[RT];
#pragma unroll
  for (int t = 0; t < RT; ++t) xnorm[t] = xx[t * 32 + l31];

  const int ntile = (m + 31) >> 5;
  const int ngroup = d >> 3;                          // groups of 8 features (4 MFMAs)
  // accumulator reg r of this lane = slot s0 + (r&3) + 8*(r>>2) + 4*h, feature row t*32 + l31
#define TOPK_TILE_EPILOGUE(s0_, EN)                                                   \
  _Pragma("unroll") for (int r = 0; r < 16; ++r) {                                    \
    const int s = (s0_) + (r & 3) + 8 * (r >> 2) + 4 * h;                             \
    if (s < m) {                                                                      \
      const float en = EN;                                                            \
      _Pragma("unroll") for (int t = 0; t < RT; ++t) {                                \
        const float dist = (xnorm[t] - 2.f * acc[t][r]) + en;                         \
        topk_insert_ordered<K>(bv[t], bi[t], dist, s);                                \
      }                                                                               \
    }                                                                                 \
  }
  if (d == 64) {
    // The model's embedding width.  A lane needs 32 dwords of E per slot tile (features 8g + 4h + t); they come
    // straight from L2, and fetching one group of four MFMAs ahead (the generic loop below) hides 256 cycles of a
    // ~2000-cycle latency: the kernel sat at a quarter of the fp32 MFMA rate waiting for them.  Here the WHOLE next
    // tile is in flight during the current one (two register sets, swapped by unrolling the tile loop twice).
    // Same MFMAs in the same order: bit-identical distances.
    float e0[32], e1[32], n0[16], n1[16];            // ... and its 16 |E_s|^2 (they were 16 exposed loads per tile)
#define TOPK_LOAD_E(dst, ndst, tile_)                                                     \
    {                                                                                 \
      /* unconditional, clamped (a slot >= m contracts a copy of slot m - 1 and is never inserted; a tile past  \
         the end reloads the last one): a load behind a condition makes the compiler count its vmcnt waits as   \
         if it was not issued, and the waits for the current tile then cover the tile just requested */        \
      const int tc_ = (tile_) < ntile ? (tile_) : ntile - 1;                          \
      const int slot_ = (tc_ << 5) + l31;                                             \
      const float* ep_ = e_dm + (slot_ < m ? slot_ : m - 1) + (int64_t)(4 * h) * m;   \
      _Pragma("unroll") for (int i = 0; i < 32; ++i)                                  \
        dst[i] = ep_[(int64_t)(8 * (i >> 2) + (i & 3)) * m];                          \
      _Pragma("unroll") for (int r = 0; r < 16; ++r) {                                \
        const int s_ = (tc_ << 5) + (r & 3) + 8 * (r >> 2) + 4 * h;                   \
        ndst[r] = enorm[s_ < m ? s_ : m - 1];                                         \
      }                                                                               \
    }
#define TOPK_TILE64(src, nsrc, tile_)                                                     \
    {                                                                                 \
      f32x16 acc[RT];                                                                 \
      _Pragma("unroll") for (int t = 0; t < RT; ++t)                                  \
        _Pragma("unroll") for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;               \
      _Pragma("unroll") for (int g = 0; g < 8; ++g) {                                 \
        const int so = ((2 * g + h) ^ (l31 & 15)) << 2;                               \
        f32x4 xf[RT];                                                                 \
        _Pragma("unroll") for (int t = 0; t < RT; ++t)                                \
          xf[t] = *reinterpret_cast<const f32x4*>(xs + (t * 32 + l31) * 64 + so);     \
        _Pragma("unroll") for (int q = 0; q < 4; ++q)                                 \
          _Pragma("unroll") for (int t = 0; t < RT; ++t)                              \
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(src[4 * g + q], xf[t][q], acc[t], 0, 0, 0); \
      }                                                                               \
      TOPK_TILE_EPILOGUE((tile_) << 5, nsrc[r])                                       \
    }
    int tile = wave;
    TOPK_LOAD_E(e0, n0, tile)
    while (tile < ntile) {
      TOPK_LOAD_E(e1, n1, tile + 4)
      __builtin_amdgcn_sched_barrier(0);
      TOPK_TILE64(e0, n0, tile)
      tile += 4;
      if (tile >= ntile) break;
      TOPK_LOAD_E(e0, n0, tile + 4)
      __builtin_amdgcn_sched_barrier(0);
      TOPK_TILE64(e1, n1, tile)
      tile += 4;
    }
#undef TOPK_LOAD_E
#undef TOPK_TILE64
  } else {
  for (int tile = wave; tile < ntile; tile += 4) {
    const int s0 = tile << 5;
    const int slot = s0 + l31;
    const bool sv = slot < m;
    const float* ep = e_dm + (sv ? slot : 0) + (int64_t)(4 * h) * m;
    f32x16 acc[RT];
#pragma unroll
    for (int t = 0; t < RT; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

    float ea[4], eb[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) ea[t] = sv ? ep[(int64_t)t * m] : 0.f;
    for (int g = 0; g < ngroup; g += 2) {
      // group g uses ea, group g+1 uses eb (d % 16 == 0 is enforced on the host)
      const float* e1 = ep + (int64_t)(8 * (g + 1)) * m;
#pragma unroll
      for (int t = 0; t < 4; ++t) eb[t] = sv ? e1[(int64_t)t * m] : 0.f;
      {
        const int so = ((2 * g + h) ^ (l31 & 15)) << 2;
        f32x4 xf[RT];
#pragma unroll
        for (int t = 0; t < RT; ++t) xf[t] = *reinterpret_cast<const f32x4*>(xs + (t * 32 + l31) * d + so);
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
          for (int t = 0; t < RT; ++t)
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(ea[q], xf[t][q], acc[t], 0, 0, 0);
      }
      if (g + 2 < ngroup) {
        const float* e2 = ep + (int64_t)(8 * (g + 2)) * m;
#pragma unroll
        for (int t = 0; t < 4; ++t) ea[t] = sv ? e2[(int64_t)t * m] : 0.f;
      }
      {
        const int so = ((2 * (g + 1) + h) ^ (l31 & 15)) << 2;
        f32x4 xf[RT];
#pragma unroll
        for (int t = 0; t < RT; ++t) xf[t] = *reinterpret_cast<const f32x4*>(xs + (t * 32 + l31) * d + so);
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
          for (int t = 0; t < RT; ++t)
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(eb[q], xf[t][q], acc[t], 0, 0, 0);
      }
    }
    TOPK_TILE_EPILOGUE(s0, enorm[s])
  }
  }
#undef TOPK_TILE_EPILOGUE

  // ---- merge the 8 partial lists of every row ------------------------------------------
#pragma unroll
  for (int t = 0; t < RT; ++t)
#pragma unroll
    for (int j = 0; j < K; ++j) {
      const int o = ((t * 32 + l31) * 8 + wave * 2 + h) * K + j;
      cand_v[o] = bv[t][j];
      cand_i[o] = bi[t][j];
    }
  __syncthreads();
  if (tid < BR) {
    float v[K];
    int ix[K];
#pragma unroll
    for (int j = 0; j < K; ++j) { v[j] = INFINITY; ix[j] = 0x7fffffff; }
    for (int c = 0; c < 8 * K; ++c) topk_insert<K>(v, ix, cand_v[tid * 8 * K + c], cand_i[tid * 8 * K + c]);
#pragma unroll
    for (int j = 0; j < K; ++j) {
      best[tid * K + j] = ix[j];
      if (r0 + tid < n) idx_out[(int64_t)(r0 + tid) * K + j] = ix[j];
    }
  }
  __syncthreads();

  // ---- gather + commit distance -----------------------------------------------------
  float part = 0.f;
  for (int p = tid; p < BR * K * slots16; p += 256) {
    const int sl = p % slots16;
    const int rj = p / slots16;
    const int j = rj % K, row = rj / K;
    if (r0 + row >= n) continue;
    const int s = best[row * K + j];
    const f32x4 e = *reinterpret_cast<const f32x4*>(e_md + (int64_t)s * d + sl * 4);
    *reinterpret_cast<f32x4*>(q_topk + ((int64_t)(r0 + row) * K + j) * d + sl * 4) = e;
    if (j == 0) {
      const f32x4 xv = *reinterpret_cast<const f32x4*>(xs + row * d + ((sl ^ (row & 15)) << 2));
      f32x4 q1;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float df = e[i] - xv[i];
        part += df * df;
        q1[i] = xv[i] + df;
      }
      if (q_one) *reinterpret_cast<f32x4*>(q_one + (int64_t)(r0 + row) * d + sl * 4) = q1;
    }
  }
  red[tid] = part;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (tid < o) red[tid] += red[tid + o];
    __syncthreads();
  }
  if (tid == 0) diff_partial[blockIdx.x] = red[0];
}

template <int K>
int launch_topk(const float* x, const float* e_dm, const float* e_md, const float* enorm, int n, int d, int m,
                int* idx, float* q_topk, float* q_one, float* diff_partial, hipStream_t stream) {
  const size_t lds = sizeof(float) * ((size_t)BR * d + BR + (size_t)BR * 8 * K * 2 + BR * K + 256);
  auto kern = memory_topk_kernel<K>;
  if (lds > 48 * 1024) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return (int)e;
  }
  const int grid = (n + BR - 1) / BR;
  hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, stream, x, e_dm, e_md, enorm, n, d, m, idx, q_topk,
                     q_one, diff_partial);
  return ammc_launch_status();
}

}  // namespace ammc_impl
using namespace ammc_impl;

extern "C" int ammc_memory_topk_blocks(int32_t n) { return n <= 0 ? 0 : (n + BR - 1) / BR; }

extern "C" int ammc_memory_topk_fwd_f32(const float* x, const float* embed_dm, const float* embed_md,
                                        const float* enorm, int32_t n, int32_t d, int32_t m, int32_t k,
                                        int32_t* idx_topk, float* q_topk, float* q_one,
                                        float* diff_partial, void* stream) {
  if (!x || !embed_dm || !embed_md || !enorm || !idx_topk || !q_topk || !diff_partial) return AMMC_EINVAL;
  if (n <= 0 || m <= 0 || k <= 0 || k > m) return AMMC_EINVAL;
  if (d < 64 || (d % 64) || d > 256) return AMMC_EUNSUP;     // x tile must fit in LDS (128 x d fp32)
  if (k > 4) return AMMC_EUNSUP;
  hipStream_t s = (hipStream_t)stream;
  switch (k) {
    case 1: return launch_topk<1>(x, embed_dm, embed_md, enorm, n, d, m, idx_topk, q_topk, q_one, diff_partial, s);
    case 2: return launch_topk<2>(x, embed_dm, embed_md, enorm, n, d, m, idx_topk, q_topk, q_one, diff_partial, s);
    case 3: return launch_topk<3>(x, embed_dm, embed_md, enorm, n, d, m, idx_topk, q_topk, q_one, diff_partial, s);
    default: return launch_topk<4>(x, embed_dm, embed_md, enorm, n, d, m, idx_topk, q_topk, q_one, diff_partial, s);
  }
}
