// Memory addressing (`Quantize_topk.forward`, reference Code/models/unet.py:282-297, 310-313) with the distance GEMM
// on the fp16 MFMA pipe in fp32-EQUIVALENT arithmetic: features and slots are carried as (hi, lo) half pairs
// (v = hi + lo 2^-11, 22 significant bits) and x . E is evaluated as hi hi + (hi lo + lo hi) 2^-11 with three
// v_mfma_f32_32x32x16_f16 per 16 features and fp32 accumulation, exactly as in the S16 convolutions
// (conv_gemm_s16.hip) - as accurate against fp64 as an fp32 dot product, at 16/3 of the fp32 MFMA rate.
// Everything else is memory_topk.hip's: dist = (|x|^2 - 2 x.E) + |E|^2 with the fp32 norms, top-K with ties to the
// lower slot, rows gathered from the fp32 codebook, commit distance and q_one from the fp32 features.
//
// (memory_topk.hip, the exact-fp32 form, spends its time on v_mfma_f32_32x32x2_f32 at 1/16 of this pipe's rate:
// 85 us per stream at 2000 slots against a 27-us floor; config 5's fp16 kernel ranks with ROUNDED operands and is not
// a parity path.  This one is the inference default for the model's 64-d embeddings.)
//
//   - a workgroup (4 waves) owns 32 feature rows: fp32 copy + S16 image in LDS (256 B per row = 16 slots of 16 B,
//     XOR-swizzled by row: conflict-free ds_read_b128 of the B fragments);
//   - wave w contracts slot tiles in pairs (2 w, 2 w + 1), (2 w + 8, ...); the codebook is pre-packed
//     [D/8][hi | lo][Mpad][8] (ammc_pack_codebook_s16), so a lane's A fragments are two coalesced 16-byte loads from
//     L2; the NEXT pair's 16 loads are in flight during the current pair's 24 MFMAs (two register sets);
//   - running top-K per lane in registers (ordered insertion), 8 partial lists per row merged through LDS.
#include "ammc_common.h"
#include <hip/hip_fp16.h>
#include <math.h>
#include <stdlib.h>

namespace ammc_impl {

typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));

constexpr int SD = 64;           // embedding width this kernel is built for
constexpr float S_LO_SCALE = 2048.f;
constexpr float S_LO_INV = 1.f / 2048.f;

template <int K>
__device__ __forceinline__ void s_topk_insert(float (&v)[K], int (&ix)[K], float c, int s) {
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

// candidates arrive in increasing slot order within a lane: strict comparisons implement the (value, index) order
template <int K>
__device__ __forceinline__ void s_topk_insert_ordered(float (&v)[K], int (&ix)[K], float c, int s) {
  if (K == 2) {
    const bool lt0 = c < v[0], lt1 = c < v[1];
    v[1] = lt0 ? v[0] : (lt1 ? c : v[1]);
    ix[1] = lt0 ? ix[0] : (lt1 ? s : ix[1]);
    v[0] = lt0 ? c : v[0];
    ix[0] = lt0 ? s : ix[0];
  } else {
    s_topk_insert<K>(v, ix, c, s);
  }
}

// NL: the slot norms are read from LDS (memories of up to 4096 slots; 2048 for RT = 2).  A template argument, not a
// pointer chosen at run time: a pointer that may be LDS or global is a FLAT pointer, and every flat load makes the
// compiler wait vmcnt(0) - i.e. for the codebook fragments just requested for the next tile group.
// RT: 32-row tiles per workgroup.  RT = 1: 4 waves x 32 rows, two workgroups per CU, a wave contracts slot tiles in
// PAIRS.  RT = 2: 8 waves x 64 rows, one workgroup per CU, a wave contracts ONE slot tile against both row tiles - the
// same 24 MFMAs per step from half the codebook bytes: every CU pulls the whole S16 codebook (516 KB at 2000 slots)
// through its vector L1 once per workgroup, and with two 32-row workgroups per CU that fill path (64 B/clk) was the
// bound - 264 MB per launch at 16384 rows, ~13 us of the kernel's 33 with neither MFMAs nor epilogue (ablations).
// Results are bit-identical between the two (same accumulation order per output, same commit partial per 32 rows).
template <int K, bool NL, int RT>
__global__ __launch_bounds__(256 * RT, 2) void memory_topk_s16_kernel(
    const float* __restrict__ x, const h16x8* __restrict__ e_s16 /* [8 k-blocks][hi | lo][mpad] */, const float* __restrict__ e_md,
    const float* __restrict__ enorm, int n, int m, int mpad, int nparts, int* __restrict__ idx_out, float* __restrict__ q_topk,
    float* __restrict__ q_one, float* __restrict__ diff_partial) {
  constexpr int SBR = 32 * RT, NT = 256 * RT, NW = 4 * RT, U = 2 / RT;   // rows, threads, waves, slot tiles per step
  __shared__ __attribute__((aligned(16))) float xs[SBR * SD];            // fp32 features, swizzled 16-B slots
  __shared__ __attribute__((aligned(16))) _Float16 xh[SBR * 2 * SD];     // S16 image: row = 8 hi slots | 8 lo slots
  __shared__ float xx[SBR];
  // (candidate rows are 2 NW K + 1 words long: with an even length the 32 lanes of a half wave - consecutive feature
  // rows - hit two banks, the LDS bank conflicts the round-3 counters showed for this kernel)
  constexpr int CROW = 2 * NW * K + 1;
  __shared__ float cand_v[SBR * CROW];
  __shared__ int cand_i[SBR * CROW];
  __shared__ int best[SBR * K];
  __shared__ float red[NT];
  __shared__ float ens[RT == 2 ? 2048 : 4096];                           // |E_s|^2

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int h = lane >> 5, l31 = lane & 31;
  const int r0 = blockIdx.x * SBR;
  // the tile epilogues read 16 norms per lane and tile: from LDS (loaded once), not 16 exposed global loads
  if (NL) {
    for (int i = tid; i < m; i += NT) ens[i] = enorm[i];
  }

  // ---- stage the tile: thread = (row, group of 8 features) -------------------------------------------------------------
  {
    const int row = tid >> 3, kg = tid & 7;
    f32x4 a = {0.f, 0.f, 0.f, 0.f}, b = {0.f, 0.f, 0.f, 0.f};
    if (r0 + row < n) {
      a = *reinterpret_cast<const f32x4*>(x + (int64_t)(r0 + row) * SD + kg * 8);
      b = *reinterpret_cast<const f32x4*>(x + (int64_t)(r0 + row) * SD + kg * 8 + 4);
    }
    *reinterpret_cast<f32x4*>(xs + row * SD + (((2 * kg) ^ (row & 15)) << 2)) = a;
    *reinterpret_cast<f32x4*>(xs + row * SD + (((2 * kg + 1) ^ (row & 15)) << 2)) = b;
    h16x8 hi, lo;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const float v = i < 4 ? a[i] : b[i - 4];
      const _Float16 hv = (_Float16)v;
      hi[i] = hv;
      lo[i] = (_Float16)((v - (float)hv) * S_LO_SCALE);
    }
    *reinterpret_cast<h16x8*>(xh + row * 2 * SD + ((kg ^ (row & 15)) << 3)) = hi;
    *reinterpret_cast<h16x8*>(xh + row * 2 * SD + (((8 + kg) ^ (row & 15)) << 3)) = lo;
  }
  __syncthreads();
  if (tid < SBR) {
    float s = 0.f;
    for (int sl = 0; sl < SD / 4; ++sl) {
      const f32x4 v = *reinterpret_cast<const f32x4*>(xs + tid * SD + ((sl ^ (tid & 15)) << 2));
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
  float xnorm[RT];
#pragma unroll
  for (int rt = 0; rt < RT; ++rt) {
#pragma unroll
    for (int j = 0; j < K; ++j) { bv[rt][j] = INFINITY; bi[rt][j] = 0x7fffffff; }
    xnorm[rt] = xx[l31 + 32 * rt];
  }

  // this lane's B fragments (its feature rows, k-half h) for the four 16-feature steps: resident for the whole kernel
  h16x8 bh[RT][4], bx[RT][4], bl2[RT][4];
#pragma unroll
  for (int rt = 0; rt < RT; ++rt)
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int kg = 2 * t + h, row = l31 + 32 * rt;
      bh[rt][t] = *reinterpret_cast<const h16x8*>(xh + row * 2 * SD + ((kg ^ (row & 15)) << 3));
      const h16x8 bl = *reinterpret_cast<const h16x8*>(xh + row * 2 * SD + (((8 + kg) ^ (row & 15)) << 3));
      bx[rt][t] = bh[rt][t] * (_Float16)S_LO_INV;
      bl2[rt][t] = bl * (_Float16)S_LO_INV;
    }

  const int ntile = mpad >> 5;
  // A fragments of one step: [u][t] hi and lo of slot (tile + u) * 32 + l31, features 16 t + 8 h .. + 7
#define S_LOAD(dst_h, dst_l, tile_)                                                                    \
  _Pragma("unroll") for (int u = 0; u < U; ++u) {                                                      \
    const int tl_ = (tile_) + u < ntile ? (tile_) + u : ntile - 1;                                     \
    const h16x8* ep_ = e_s16 + ((int64_t)(2 * h) * mpad + (tl_ << 5) + l31);                           \
    _Pragma("unroll") for (int t = 0; t < 4; ++t) {                                                    \
      dst_h[u][t] = ep_[(int64_t)(4 * t) * mpad];                                                      \
      dst_l[u][t] = ep_[(int64_t)(4 * t + 1) * mpad];                                                  \
    }                                                                                                  \
  }
#define S_TILES(src_h, src_l, tile_)                                                                   \
  {                                                                                                    \
    f32x16 acc[U][RT];                                                                                 \
    _Pragma("unroll") for (int u = 0; u < U; ++u)                                                      \
      _Pragma("unroll") for (int rt = 0; rt < RT; ++rt)                                                \
        _Pragma("unroll") for (int r = 0; r < 16; ++r) acc[u][rt][r] = 0.f;                            \
    _Pragma("unroll") for (int t = 0; t < 4; ++t) {                                                    \
      _Pragma("unroll") for (int u = 0; u < U; ++u) _Pragma("unroll") for (int rt = 0; rt < RT; ++rt)  \
        acc[u][rt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(src_h[u][t], bh[rt][t], acc[u][rt], 0, 0, 0);  \
      _Pragma("unroll") for (int u = 0; u < U; ++u) _Pragma("unroll") for (int rt = 0; rt < RT; ++rt)  \
        acc[u][rt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(src_l[u][t], bx[rt][t], acc[u][rt], 0, 0, 0);  \
      _Pragma("unroll") for (int u = 0; u < U; ++u) _Pragma("unroll") for (int rt = 0; rt < RT; ++rt)  \
        acc[u][rt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(src_h[u][t], bl2[rt][t], acc[u][rt], 0, 0, 0); \
    }                                                                                                  \
    /* the 16 slot norms of a tile first (independent LDS reads, one wait), +inf for slots beyond m: such a candidate  \
       never wins (strict comparisons against a list that starts at +inf), so the insertion needs no per-candidate      \
       branch - the first form's `if (s < m)` around an LDS read cost an exposed LDS latency and an exec-mask branch   \
       per candidate, a quarter of this kernel's time */                                               \
    _Pragma("unroll") for (int u = 0; u < U; ++u) {                                                    \
      if ((tile_) + u < ntile) {                                                                       \
        const int s0_ = ((tile_) + u) << 5;                                                            \
        float en_[16];                                                                                 \
        _Pragma("unroll") for (int r = 0; r < 16; ++r) {                                               \
          const int s = s0_ + (r & 3) + 8 * (r >> 2) + 4 * h;                                          \
          const int sc = s < m ? s : m - 1;                                                            \
          const float e_ = NL ? ens[sc] : enorm[sc];                                                   \
          en_[r] = s < m ? e_ : INFINITY;                                                              \
        }                                                                                              \
        _Pragma("unroll") for (int rt = 0; rt < RT; ++rt)                                              \
          _Pragma("unroll") for (int r = 0; r < 16; ++r) {                                             \
            const float dist = (xnorm[rt] - 2.f * acc[u][rt][r]) + en_[r];                             \
            s_topk_insert_ordered<K>(bv[rt], bi[rt], dist, s0_ + (r & 3) + 8 * (r >> 2) + 4 * h);      \
          }                                                                                            \
      }                                                                                                \
    }                                                                                                  \
  }
  {
    h16x8 ah0[U][4], al0[U][4], ah1[U][4], al1[U][4];
    // (the prefetch loads are UNCONDITIONAL - S_LOAD clamps the tile index - because a load the compiler cannot be
    // sure was issued makes it count its vmcnt waits as if it was not: the waits of the current step then also cover
    // the step just requested and the prefetch is gone; seen in the ISA as vmcnt(15) .. vmcnt(0) inside the MFMAs)
    int tile = wave * U;                          // NW waves x U tiles = 8 tiles per round in both forms
    S_LOAD(ah0, al0, tile)
    while (tile < ntile) {
      S_LOAD(ah1, al1, tile + 8)
      __builtin_amdgcn_sched_barrier(0);         // (else the scheduler sinks these loads behind the MFMAs to save registers)
      S_TILES(ah0, al0, tile)
      tile += 8;
      if (tile >= ntile) break;
      S_LOAD(ah0, al0, tile + 8)
      __builtin_amdgcn_sched_barrier(0);
      S_TILES(ah1, al1, tile)
      tile += 8;
    }
  }
#undef S_LOAD
#undef S_TILES

  // ---- merge the 2 NW partial lists of every row ------------------------------------------------------------------------
#pragma unroll
  for (int rt = 0; rt < RT; ++rt)
#pragma unroll
    for (int j = 0; j < K; ++j) {
      const int o = (l31 + 32 * rt) * CROW + (wave * 2 + h) * K + j;
      cand_v[o] = bv[rt][j];
      cand_i[o] = bi[rt][j];
    }
  __syncthreads();
  if (tid < SBR) {
    float v[K];
    int ix[K];
#pragma unroll
    for (int j = 0; j < K; ++j) { v[j] = INFINITY; ix[j] = 0x7fffffff; }
    for (int c = 0; c < 2 * NW * K; ++c) s_topk_insert<K>(v, ix, cand_v[tid * CROW + c], cand_i[tid * CROW + c]);
#pragma unroll
    for (int j = 0; j < K; ++j) {
      best[tid * K + j] = ix[j];
      if (r0 + tid < n) idx_out[(int64_t)(r0 + tid) * K + j] = ix[j];
    }
  }
  __syncthreads();

  // ---- gather + commit distance (fp32 codebook, fp32 features): 256 threads per 32 rows, one partial per 32 rows ------------
  float part = 0.f;
  constexpr int slots16 = SD / 4;
  const int half = tid >> 8, t8 = tid & 255;                      // (RT = 1: half = 0)
  for (int p = t8; p < 32 * K * slots16; p += 256) {
    const int sl = p % slots16;
    const int rj = p / slots16;
    const int j = rj % K, row = 32 * half + rj / K;
    if (r0 + row >= n) continue;
    const int s = best[row * K + j];
    const f32x4 e = *reinterpret_cast<const f32x4*>(e_md + (int64_t)s * SD + sl * 4);
    *reinterpret_cast<f32x4*>(q_topk + ((int64_t)(r0 + row) * K + j) * SD + sl * 4) = e;
    if (j == 0) {
      const f32x4 xv = *reinterpret_cast<const f32x4*>(xs + row * SD + ((sl ^ (row & 15)) << 2));
      f32x4 q1;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float df = e[i] - xv[i];
        part += df * df;
        q1[i] = xv[i] + df;
      }
      if (q_one) *reinterpret_cast<f32x4*>(q_one + (int64_t)(r0 + row) * SD + sl * 4) = q1;
    }
  }
  red[tid] = part;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (t8 < o) red[tid] += red[tid + o];
    __syncthreads();
  }
  if (t8 == 0 && (int)blockIdx.x * RT + half < nparts) diff_partial[blockIdx.x * RT + half] = red[tid];
}

// [d][m] fp32 -> [d/8][hi | lo][mpad][8] halfs (slots >= m zero)
// `range_flag` (may be null) is raised when an entry does not fit the hi half (|v| > 65504 or not finite): such a slot would
// read as inf / NaN in the S16 image - a NaN distance is never inserted, a -inf distance WINS - so the caller must take
// the fp32 kernel for this codebook (engine.py does).  The reference's EMA update produces exactly that from its own
// initial state: a slot no row has hit yet sits at embed = 0.99^t e0 / ~1e-5 (models/unet.py:277-280, 298-309).
__global__ __launch_bounds__(256) void pack_codebook_s16_kernel(const float* __restrict__ e_dm, int d, int m, int mpad,
                                                                _Float16* __restrict__ out, int* __restrict__ range_flag) {
  const int64_t gid = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (gid >= (int64_t)(d >> 3) * mpad) return;
  const int s = (int)(gid % mpad), kb = (int)(gid / mpad);
  h16x8 hi, lo;
  bool bad = false;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const float v = s < m ? e_dm[(int64_t)(kb * 8 + i) * m + s] : 0.f;
    bad |= !(fabsf(v) <= 65504.f);
    const _Float16 hv = (_Float16)v;
    hi[i] = hv;
    lo[i] = (_Float16)((v - (float)hv) * S_LO_SCALE);
  }
  // plane-major within a k-block: [kb][hi | lo][mpad][8] - the 32 lanes of a half wave read 512 contiguous bytes per
  // load (interleaved [8 hi | 8 lo] rows made every load touch twice the cache lines it used)
  *reinterpret_cast<h16x8*>(out + ((int64_t)(2 * kb) * mpad + s) * 8) = hi;
  *reinterpret_cast<h16x8*>(out + ((int64_t)(2 * kb + 1) * mpad + s) * 8) = lo;
  if (bad && range_flag) atomicOr(range_flag, 1);
}

template <int K>
int launch_topk_s16(const float* x, const void* e_s16, const float* e_md, const float* enorm, int n, int m, int* idx,
                    float* q_topk, float* q_one, float* diff_partial, hipStream_t stream) {
  const int mpad = (m + 31) / 32 * 32;
  const int nparts = (n + 31) / 32;                                 // = ammc_memory_topk_blocks(n): one commit partial per 32 rows
  const h16x8* e = reinterpret_cast<const h16x8*>(e_s16);
  // 64-row workgroups (half the codebook traffic per CU) once they fill the chip; option "memory_rt" forces a form
  const int force = ammc_opt_memory_rt();
  const bool rt2 = K <= 2 && m <= 2048 && (force == 2 || (force != 1 && n >= 64 * 256));   // (K > 2: candidate lists outgrow the LDS)
  if (rt2)
    hipLaunchKernelGGL((memory_topk_s16_kernel<(K <= 2 ? K : 1), true, 2>), dim3((n + 63) / 64), dim3(512), 0, stream, x, e,
                       e_md, enorm, n, m, mpad, nparts, idx, q_topk, q_one, diff_partial);
  else if (m <= 4096)
    hipLaunchKernelGGL((memory_topk_s16_kernel<K, true, 1>), dim3(nparts), dim3(256), 0, stream, x, e, e_md, enorm, n, m,
                       mpad, nparts, idx, q_topk, q_one, diff_partial);
  else
    hipLaunchKernelGGL((memory_topk_s16_kernel<K, false, 1>), dim3(nparts), dim3(256), 0, stream, x, e, e_md, enorm, n, m,
                       mpad, nparts, idx, q_topk, q_one, diff_partial);
  return ammc_launch_status();
}


// ======================================================================================================================
// Round 6: the WHOLE memory block of the inference path as ONE launch (`enc_quan_dec_res_topk.forward`, reference
// Code/models/unet.py:318-331, 379-387): enc 1x1 (C -> 64, + bias) -> distance GEMM + top-2 (the kernel above) -> gather ->
// dec 1x1 (128 -> C, + bias) -> `out += x` for a 64-pixel tile, one workgroup (8 waves) per tile.  Until round 5 this was
// five launches per stream (conv_gemm_s16 1x1, memory_topk_s16, sum_partials, split_rows, conv_gemm_s16 1x1): ~95 us of
// kernel time for ~9 us of MFMAs, every stage latency-bound on its own (16384 rows = 256 tiles, one wave of workgroups).
// Now z, the N x M distances, the gathered rows and their S16 re-encoding never leave the CU:
//   A  z = x4 . Wenc: the pixel tile and the filters travel L2 -> LDS by LDS-DMA in 64-channel chunks (three stages, two
//      chunks in flight, one barrier per chunk; lanes of a DMA read consecutive 16-byte pieces - fragments fetched per lane
//      straight from L2, one cache line per lane and instruction, kept the CU's vector-memory path busy for 23 us); waves
//      0-3 contract a 32-pixel x 32-channel tile each, K sequential - the SAME k order, accumulator pair and epilogue
//      expression as conv_gemm_s16's 1x1 path, so z (and with it every lookup) is bit-identical to the five-launch chain;
//   B  the distance sweep / top-2 of memory_topk_s16_kernel<2, true, 2>, unchanged; the 32 partial lists of a row are
//      merged by eight lanes (the order is total - value, then slot - so the result does not depend on the merge tree);
//   C  gather: q_topk (optional), q_one, commit partial with the thread -> element map and the summation tree of the
//      kernel above (the same partials, bit for bit), plus the S16 image of the gathered rows in LDS;
//   D  every wave: out[64 pixels x 64 channels] = qk . Wdec, the filters pre-packed FRAGMENT-major
//      (ammc_pack_frag_rows_s16: a lane's fragment loads are 512 contiguous bytes per half wave); the accumulator tile
//      turns through a wave-private LDS tile so that bias + residual + split + store run with eight lanes per pixel on
//      256 contiguous bytes (one pixel per lane cost 16 cache lines per store instruction: 8 us);
//   E  the LAST workgroup to arrive sums the commit partials in sum_partials_kernel's order.  No release fence:
//      `__threadfence()` is `buffer_wbl2 sc1` on gfx950 - a write-back of the XCD's whole L2, by every wave of every
//      workgroup (the first form spent ~70 us there).  The partials are agent-scope atomic stores (write-through),
//      completed (vmcnt(0)) before the workgroup's barrier, behind which one thread bumps the arrival counter with a
//      relaxed agent-scope atomic; the last workgroup reads them with agent-scope atomic loads and leaves the counter at 0.
// (profiling only) AMMC_MB_STAMPS=1: workgroups 0 / 100 write the cycle counter at every phase boundary into this buffer
__device__ unsigned long long g_mb_stamps[2][16];
#define MB_STAMP(i)                                                                                  \
  if (a.stamps && (blockIdx.x == 0 || blockIdx.x == 100) && tid == 0)                               \
    g_mb_stamps[blockIdx.x ? 1 : 0][i] = __builtin_readcyclecounter();

struct MemBlockArgs {
  const float* x; int64_t x_bs, x_rs, x_ps;       // S16 activation [B][h][w][C] (interior pixel 0)
  float* y; int64_t y_bs, y_rs, y_ps;             // S16 output, same geometry
  const float* enc_w; const float* enc_b;         // S16 [64][C] (k-major rows), fp32 [64]
  const float* dec_wf; const float* dec_b;        // S16 [C][128] FRAGMENT-major (ammc_pack_frag_rows_s16), fp32 [C]
  const h16x8* e_s16; const float* e_md; const float* enorm;
  int n, hw, w, m, mpad, nparts;
  int* idx; float* q_topk; float* q_one;
  float* diff_partial; float* diff; int* counter; float inv_count;
  int* overflow_flag;
  int stamps;
};

// one 16-byte LDS-DMA per lane: global `src` (a full address per lane) -> LDS byte `lds_byte` (wave-uniform) + 16 lane.
// asm, because the builtin makes hipcc's waitcnt pass drain both counters in front of every later LDS read (DESIGN.md
// section 8, pitfall 9); M0 saved / restored inside the statement, s_nop 4: the SGPR may have just been written.
__device__ __forceinline__ void mb_dma16(const float* src, unsigned lds_byte) {
  unsigned keep;
  asm volatile("s_nop 4\n\ts_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(src), "s"(lds_byte) : "memory");
}

#ifndef AMMC_MB_STAGGER
#define AMMC_MB_STAGGER 0
#endif
constexpr int MB_STG = 8192;                       // floats per stage of phase A: x chunk [64 px][64 ch] | filter chunk [64][64]
constexpr int MB_DYN_FLOATS = 3 * MB_STG;          // 96 KB of dynamic LDS

template <int C>
__global__ __launch_bounds__(512, 2) void memory_block_s16_kernel(MemBlockArgs a) {
  constexpr int K = 2, RT = 2, SBR = 64, NT = 512, NW = 8;
  static_assert(C == 512, "8 waves x 64 output channels; 8 chunks of 64 input channels");
  extern __shared__ __attribute__((aligned(16))) float dyn[];
  // lives of the three 32-KB stages: A: DMA stages 0 / 1 / 2.  B: xs + xh in stage 2, the candidate lists in stage 0.
  // C, D: the gathered rows' S16 image in stage 1.  D epilogue: wave-private 8-KB tiles in stages 0 (waves 0-3) and 2 (4-7).
  float* xs = dyn + 2 * MB_STG;                                             // fp32 z [64][64], swizzled 16-B slots
  _Float16* xh = reinterpret_cast<_Float16*>(dyn + 2 * MB_STG + 4096);      // S16 image of z: row = 8 hi slots | 8 lo slots
  constexpr int CROW = 2 * NW * K + 1;
  float* cand_v = dyn;
  int* cand_i = reinterpret_cast<int*>(dyn + SBR * CROW);
  float* qs = dyn + MB_STG;                                                 // S16 image of the gathered rows [64][128 ch]
  __shared__ float xx[SBR];
  __shared__ int best[SBR * K];
  __shared__ float red[NT];
  __shared__ double dred[256];
  __shared__ float ens[2048];
  __shared__ int is_last;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int h = lane >> 5, l31 = lane & 31;
  const int r0 = blockIdx.x * SBR;
  const int n = a.n, m = a.m, mpad = a.mpad;
  const int pl31 = (l31 & 19) | ((l31 & 4) << 1) | ((l31 & 8) >> 1);     // MFMA row -> filter: bits 2 and 3 swapped (conv_gemm_s16.hip)
  MB_STAMP(0)
  for (int i = tid; i < m; i += NT) ens[i] = a.enorm[i];

  auto pix_of = [&](int row, int& pin, int& pout) {                        // (rows beyond n: the last row's pixel, results dropped)
    int r = r0 + row;
    r = r < n ? r : n - 1;
    const int b = r / a.hw, rem = r - b * a.hw, yy = rem / a.w, xq = rem - yy * a.w;
    pin = (int)((int64_t)b * a.x_bs + (int64_t)yy * a.x_rs + (int64_t)xq * a.x_ps);
    pout = (int)((int64_t)b * a.y_bs + (int64_t)yy * a.y_rs + (int64_t)xq * a.y_ps);
  };

  // ---- A: enc 1x1 ------------------------------------------------------------------------------------------------------------
  {
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) void*)dyn;
    const int uwave = __builtin_amdgcn_readfirstlane(wave);
    // a DMA instruction fills 4 rows x 16 pieces; lane = (row in the group, LDS position); the piece it FETCHES is
    // position ^ (row & 15), so that a fragment read (32 rows, one piece) meets every bank once per 16 lanes
    const int drow0 = (uwave * 2) * 4 + (lane >> 4), drow1 = drow0 + 4, dpos = lane & 15;
    int pin0, pin1, unused;
    pix_of(drow0, pin0, unused);
    pix_of(drow1, pin1, unused);
    const float* sx0 = a.x + pin0 + 4 * (dpos ^ (drow0 & 15));
    const float* sx1 = a.x + pin1 + 4 * (dpos ^ (drow1 & 15));
    const float* sw0 = a.enc_w + (int64_t)drow0 * C + 4 * (dpos ^ (drow0 & 15));
    const float* sw1 = a.enc_w + (int64_t)drow1 * C + 4 * (dpos ^ (drow1 & 15));
    const unsigned dst0 = lds0 + 4u * (unsigned)(uwave * 512);              // this wave's two 1-KB pieces of a 16-KB operand chunk
#define MB_ISSUE(c, st)                                                                             \
  {                                                                                                 \
    mb_dma16(sx0 + (c) * 64, dst0 + 4u * (unsigned)((st) * MB_STG));                                \
    mb_dma16(sx1 + (c) * 64, dst0 + 4u * (unsigned)((st) * MB_STG + 256));                          \
    mb_dma16(sw0 + (c) * 64, dst0 + 4u * (unsigned)((st) * MB_STG + 4096));                         \
    mb_dma16(sw1 + (c) * 64, dst0 + 4u * (unsigned)((st) * MB_STG + 4096 + 256));                   \
  }
    const int pt = wave & 1, ct = (wave >> 1) & 1;
    const int xrow = 32 * pt + l31, frow = 32 * ct + pl31;
    const int swz = xrow & 15, swzb = frow & 15;
    f32x16 hh, xa;
#pragma unroll
    for (int r = 0; r < 16; ++r) { hh[r] = 0.f; xa[r] = 0.f; }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                         // (the norms' loads: nothing of mine may be in flight)
    MB_ISSUE(0, 0)
    MB_ISSUE(1, 1)
#pragma unroll
    for (int c = 0; c < C / 64; ++c) {
      if (c + 1 < C / 64) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");   // chunk c has landed (chunk c + 1: four DMAs per wave)
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();                                                       // ... everybody's; and everybody is done with chunk c - 1
      if (c + 2 < C / 64) MB_ISSUE(c + 2, (c + 2) % 3)
      if (wave < 4) {
        const float* Xc = dyn + (c % 3) * MB_STG + xrow * 64;
        const float* Wc = dyn + (c % 3) * MB_STG + 4096 + frow * 64;
        h16x8 ah[4], al[4], bh[4], bl[4];                  // (all sixteen reads of the chunk in flight before the first MFMA)
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          const int g = 2 * s + h;
          ah[s] = *reinterpret_cast<const h16x8*>(Xc + (((2 * g) ^ swz) << 2));
          al[s] = *reinterpret_cast<const h16x8*>(Xc + (((2 * g + 1) ^ swz) << 2));
          bh[s] = *reinterpret_cast<const h16x8*>(Wc + (((2 * g) ^ swzb) << 2));
          bl[s] = *reinterpret_cast<const h16x8*>(Wc + (((2 * g + 1) ^ swzb) << 2));
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          hh = __builtin_amdgcn_mfma_f32_32x32x16_f16(bh[s], ah[s], hh, 0, 0, 0);
          xa = __builtin_amdgcn_mfma_f32_32x32x16_f16(bl[s], ah[s], xa, 0, 0, 0);
          xa = __builtin_amdgcn_mfma_f32_32x32x16_f16(bh[s], al[s], xa, 0, 0, 0);
        }
      }
    }
#undef MB_ISSUE
    // (stage 2 held chunk 5: every wave is past the barrier of chunk 7, i.e. done with chunk 6 - free to take z)
    if (wave < 4) {
      const bool live = r0 + xrow < n;
#pragma unroll
      for (int o = 0; o < 2; ++o) {
        const int c0 = ct * 32 + 8 * (2 * o + h);
        const f32x4 s0 = *reinterpret_cast<const f32x4*>(a.enc_b + c0), s1 = *reinterpret_cast<const f32x4*>(a.enc_b + c0 + 4);
        float v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = hh[8 * o + k] + xa[8 * o + k] * S_LO_INV;
#pragma unroll
        for (int k = 0; k < 4; ++k) { v[k] += s0[k]; v[4 + k] += s1[k]; }
        if (!live) {
#pragma unroll
          for (int k = 0; k < 8; ++k) v[k] = 0.f;
        }
        const int kg = c0 >> 3;
        *reinterpret_cast<f32x4*>(xs + xrow * SD + (((2 * kg) ^ (xrow & 15)) << 2)) = f32x4{v[0], v[1], v[2], v[3]};
        *reinterpret_cast<f32x4*>(xs + xrow * SD + (((2 * kg + 1) ^ (xrow & 15)) << 2)) = f32x4{v[4], v[5], v[6], v[7]};
        ammc_u4 hi, lo;
        ammc_s16_split8(v, hi, lo);
        *reinterpret_cast<ammc_u4*>(xh + xrow * 2 * SD + ((kg ^ (xrow & 15)) << 3)) = hi;
        *reinterpret_cast<ammc_u4*>(xh + xrow * 2 * SD + (((8 + kg) ^ (xrow & 15)) << 3)) = lo;
      }
    }
  }
  __syncthreads();
  MB_STAMP(1)
  if (tid < SBR) {
    float s = 0.f;
    for (int sl = 0; sl < SD / 4; ++sl) {
      const f32x4 v = *reinterpret_cast<const f32x4*>(xs + tid * SD + ((sl ^ (tid & 15)) << 2));
      s += v[0] * v[0];
      s += v[1] * v[1];
      s += v[2] * v[2];
      s += v[3] * v[3];
    }
    xx[tid] = s;
  }
  __syncthreads();
  MB_STAMP(2)

  // ---- B: distances + running top-2 (memory_topk_s16_kernel<2, true, 2>'s sweep) ---------------------------------------------
  {
    float bv[RT][K];
    int bi[RT][K];
    float xnorm[RT];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
#pragma unroll
      for (int j = 0; j < K; ++j) { bv[rt][j] = INFINITY; bi[rt][j] = 0x7fffffff; }
      xnorm[rt] = xx[l31 + 32 * rt];
    }
    h16x8 bh[RT][4], bx[RT][4], bl2[RT][4];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const int kg = 2 * t + h, row = l31 + 32 * rt;
        bh[rt][t] = *reinterpret_cast<const h16x8*>(xh + row * 2 * SD + ((kg ^ (row & 15)) << 3));
        const h16x8 bl = *reinterpret_cast<const h16x8*>(xh + row * 2 * SD + (((8 + kg) ^ (row & 15)) << 3));
        bx[rt][t] = bh[rt][t] * (_Float16)S_LO_INV;
        bl2[rt][t] = bl * (_Float16)S_LO_INV;
      }
    const int ntile = mpad >> 5;
    const h16x8* e_s16 = a.e_s16;
#define F_LOAD(dst_h, dst_l, tile_)                                                                    \
  {                                                                                                    \
    const int tl_ = (tile_) < ntile ? (tile_) : ntile - 1;                                             \
    const h16x8* ep_ = e_s16 + ((int64_t)(2 * h) * mpad + (tl_ << 5) + l31);                           \
    _Pragma("unroll") for (int t = 0; t < 4; ++t) {                                                    \
      dst_h[t] = ep_[(int64_t)(4 * t) * mpad];                                                         \
      dst_l[t] = ep_[(int64_t)(4 * t + 1) * mpad];                                                     \
    }                                                                                                  \
  }
#define F_TILES(src_h, src_l, tile_)                                                                   \
  {                                                                                                    \
    f32x16 acc[RT];                                                                                    \
    _Pragma("unroll") for (int rt = 0; rt < RT; ++rt)                                                  \
      _Pragma("unroll") for (int r = 0; r < 16; ++r) acc[rt][r] = 0.f;                                 \
    _Pragma("unroll") for (int t = 0; t < 4; ++t) {                                                    \
      _Pragma("unroll") for (int rt = 0; rt < RT; ++rt)                                                \
        acc[rt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(src_h[t], bh[rt][t], acc[rt], 0, 0, 0);       \
      _Pragma("unroll") for (int rt = 0; rt < RT; ++rt)                                                \
        acc[rt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(src_l[t], bx[rt][t], acc[rt], 0, 0, 0);       \
      _Pragma("unroll") for (int rt = 0; rt < RT; ++rt)                                                \
        acc[rt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(src_h[t], bl2[rt][t], acc[rt], 0, 0, 0);      \
    }                                                                                                  \
    if ((tile_) < ntile) {                                                                             \
      const int s0_ = (tile_) << 5;                                                                    \
      float en_[16];                                                                                   \
      _Pragma("unroll") for (int r = 0; r < 16; ++r) {                                                 \
        const int s = s0_ + (r & 3) + 8 * (r >> 2) + 4 * h;                                            \
        const int sc = s < m ? s : m - 1;                                                              \
        en_[r] = s < m ? ens[sc] : INFINITY;                                                           \
      }                                                                                                \
      _Pragma("unroll") for (int rt = 0; rt < RT; ++rt)                                                \
        _Pragma("unroll") for (int r = 0; r < 16; ++r) {                                               \
          const float dist = (xnorm[rt] - 2.f * acc[rt][r]) + en_[r];                                  \
          s_topk_insert_ordered<K>(bv[rt], bi[rt], dist, s0_ + (r & 3) + 8 * (r >> 2) + 4 * h);        \
        }                                                                                              \
    }                                                                                                  \
  }
    {
      h16x8 ah0[4], al0[4], ah1[4], al1[4];
      int tile = wave;
      F_LOAD(ah0, al0, tile)
#if AMMC_MB_STAGGER
      // the two waves of a SIMD (w and w + 4) run the same program in step: both contract (the matrix pipe serves both, 2 x
      // 768 cycles), then both insert (the VALU serves both, 2 x 1400) - the phases add up.  The second wave starts the
      // sweep half a period late, so that its insertions meet the first wave's MFMAs.
      if (wave >= 4) __builtin_amdgcn_s_sleep(AMMC_MB_STAGGER);
#endif
      while (tile < ntile) {
        F_LOAD(ah1, al1, tile + 8)
        __builtin_amdgcn_sched_barrier(0);
        F_TILES(ah0, al0, tile)
        tile += 8;
        if (tile >= ntile) break;
        F_LOAD(ah0, al0, tile + 8)
        __builtin_amdgcn_sched_barrier(0);
        F_TILES(ah1, al1, tile)
        tile += 8;
      }
    }
#undef F_LOAD
#undef F_TILES
    MB_STAMP(3)
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
      for (int j = 0; j < K; ++j) {
        const int o = (l31 + 32 * rt) * CROW + (wave * 2 + h) * K + j;
        cand_v[o] = bv[rt][j];
        cand_i[o] = bi[rt][j];
      }
  }
  __syncthreads();
  MB_STAMP(4)
  {
    // eight lanes per row, four candidates each, then three exchange rounds (the order (value, slot) is total: any merge
    // tree returns the two smallest)
    const int row = tid >> 3, part = tid & 7;
    float v[K];
    int ix[K];
#pragma unroll
    for (int j = 0; j < K; ++j) { v[j] = INFINITY; ix[j] = 0x7fffffff; }
#pragma unroll
    for (int c = 0; c < 4; ++c) s_topk_insert<K>(v, ix, cand_v[row * CROW + part * 4 + c], cand_i[row * CROW + part * 4 + c]);
#pragma unroll
    for (int o = 1; o < 8; o <<= 1) {
      float pv[K];
      int pi_[K];
#pragma unroll
      for (int j = 0; j < K; ++j) { pv[j] = __shfl_xor(v[j], o); pi_[j] = __shfl_xor(ix[j], o); }
#pragma unroll
      for (int j = 0; j < K; ++j) s_topk_insert<K>(v, ix, pv[j], pi_[j]);
    }
    if (part == 0) {
#pragma unroll
      for (int j = 0; j < K; ++j) {
        best[row * K + j] = ix[j];
        if (r0 + row < n) a.idx[(int64_t)(r0 + row) * K + j] = ix[j];
      }
    }
  }
  __syncthreads();
  MB_STAMP(5)

  // ---- C: gather + commit distance (the map of memory_topk_s16_kernel: the same partial sums), S16 image of the rows -------------
  float part = 0.f;
  constexpr int slots16 = SD / 4;
  const int half = tid >> 8, t8 = tid & 255;
  for (int p = t8; p < 32 * K * slots16; p += 256) {
    const int sl = p % slots16;
    const int rj = p / slots16;
    const int j = rj % K, row = 32 * half + rj / K;
    if (r0 + row >= n) continue;
    const int s = best[row * K + j];
    const f32x4 e = *reinterpret_cast<const f32x4*>(a.e_md + (int64_t)s * SD + sl * 4);
    if (a.q_topk) *reinterpret_cast<f32x4*>(a.q_topk + ((int64_t)(r0 + row) * K + j) * SD + sl * 4) = e;
    if (j == 0) {
      const f32x4 xv = *reinterpret_cast<const f32x4*>(xs + row * SD + ((sl ^ (row & 15)) << 2));
      f32x4 q1;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float df = e[i] - xv[i];
        part += df * df;
        q1[i] = xv[i] + df;
      }
      if (a.q_one) *reinterpret_cast<f32x4*>(a.q_one + (int64_t)(r0 + row) * SD + sl * 4) = q1;
    }
  }
  // (row, neighbour, group of 8 channels): 64 x 2 x 8 items; slot s of a 512-byte row lives at s ^ (row & 31)
  for (int p = tid; p < SBR * K * (SD / 8); p += NT) {
    const int g = p & 7, j = (p >> 3) & 1, row = p >> 4;
    const int s = r0 + row < n ? best[row * K + j] : 0;
    const float* ep = a.e_md + (int64_t)s * SD + g * 8;
    const f32x4 e0 = *reinterpret_cast<const f32x4*>(ep), e1 = *reinterpret_cast<const f32x4*>(ep + 4);
    const float v[8] = {e0[0], e0[1], e0[2], e0[3], e1[0], e1[1], e1[2], e1[3]};
    ammc_u4 hi, lo;
    ammc_s16_split8(v, hi, lo);
    bool bad = false;
#pragma unroll
    for (int k = 0; k < 8; ++k) bad |= !(fabsf(v[k]) <= 65504.f);
    if (bad && a.overflow_flag) atomicOr(a.overflow_flag, 1);            // (a gathered slot beyond the half range: split_rows' verdict)
    const int kg = j * 8 + g;                                             // channel group of the 128-channel row
    *reinterpret_cast<ammc_u4*>(qs + row * 2 * SD + (((2 * kg) ^ (row & 31)) << 2)) = hi;
    *reinterpret_cast<ammc_u4*>(qs + row * 2 * SD + (((2 * kg + 1) ^ (row & 31)) << 2)) = lo;
  }
  // the kernel above sums red[t] += red[t + o] for o = 128 ... 1 with a barrier per level; the same pairs here: two levels
  // through LDS, six inside wave 0 of each half (v + shfl_down(v, o) IS red[t] + red[t + o])
  red[tid] = part;
  __syncthreads();
  if (t8 < 128) red[tid] += red[tid + 128];
  __syncthreads();
  if (t8 < 64) {
    float v = red[tid] + red[tid + 64];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o);
    if (t8 == 0 && (int)blockIdx.x * RT + half < a.nparts)
      __hip_atomic_store(a.diff_partial + blockIdx.x * RT + half, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  MB_STAMP(6)

  // ---- D: dec 1x1 + bias + residual, S16 out: wave = 64 output channels x 64 pixels ------------------------------------------------
  {
    f32x16 hh[RT][2], xa[RT][2];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
      for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int r = 0; r < 16; ++r) { hh[rt][c][r] = 0.f; xa[rt][c][r] = 0.f; }
    // fragment-major filters: [group g][hi | lo][C rows, MFMA order][8 halfs]: lane l31 of tile (wave, c) reads row 64 wave + 32 c + l31
    const float* fw0 = a.dec_wf + (int64_t)(wave * 64 + l31) * 4;
#pragma unroll
    for (int t = 0; t < (2 * SD) / 16; ++t) {
      const int g = 2 * t + h;
      h16x8 ah[RT], al[RT], bh[2], bl[2];
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        bh[c] = *reinterpret_cast<const h16x8*>(fw0 + ((int64_t)(2 * g) * C + 32 * c) * 4);
        bl[c] = *reinterpret_cast<const h16x8*>(fw0 + ((int64_t)(2 * g + 1) * C + 32 * c) * 4);
      }
#pragma unroll
      for (int rt = 0; rt < RT; ++rt) {
        const int row = 32 * rt + l31;
        ah[rt] = *reinterpret_cast<const h16x8*>(qs + row * 2 * SD + (((2 * g) ^ (row & 31)) << 2));
        al[rt] = *reinterpret_cast<const h16x8*>(qs + row * 2 * SD + (((2 * g + 1) ^ (row & 31)) << 2));
      }
#pragma unroll
      for (int rt = 0; rt < RT; ++rt)
#pragma unroll
        for (int c = 0; c < 2; ++c) {
          hh[rt][c] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bh[c], ah[rt], hh[rt][c], 0, 0, 0);
          xa[rt][c] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bl[c], ah[rt], xa[rt][c], 0, 0, 0);
          xa[rt][c] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bh[c], al[rt], xa[rt][c], 0, 0, 0);
        }
    }
    MB_STAMP(7)
    // epilogue through a wave-private tile [32 px][64 ch] fp32 (16-B pieces XOR-swizzled by pixel): the accumulators go in
    // with one pixel per lane, come out with eight lanes per pixel = 256 contiguous bytes of residual and of output
    float* tile = dyn + (wave < 4 ? wave * 2048 : 2 * MB_STG + (wave - 4) * 2048);
    const int gq = lane & 7, psub = lane >> 3;
    const int cw = wave * 64 + gq * 8;                                      // this lane's eight output channels
    const f32x4 b0 = *reinterpret_cast<const f32x4*>(a.dec_b + cw), b1 = *reinterpret_cast<const f32x4*>(a.dec_b + cw + 4);
    bool bad = false;
    h16x8 rh[RT][4], rl[RT][4];                                            // the residual's S16 groups, a whole row tile at a time
    int pout[RT][4];
#define MB_RES(rt)                                                                                   \
  _Pragma("unroll") for (int it = 0; it < 4; ++it) {                                                 \
    int pin_;                                                                                        \
    pix_of(32 * (rt) + 8 * it + psub, pin_, pout[rt][it]);                                           \
    const float* rp_ = a.x + pin_ + cw;                                                              \
    rh[rt][it] = *reinterpret_cast<const h16x8*>(rp_);                                               \
    rl[rt][it] = *reinterpret_cast<const h16x8*>(rp_ + 4);                                           \
  }
    MB_RES(0)
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
#pragma unroll
      for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int o = 0; o < 2; ++o) {
          const int q = c * 8 + 2 * (2 * o + h);                            // 16-B piece of channels 32 c + 8 (2 o + h) ..
          float* tp = tile + l31 * 64;
          *reinterpret_cast<f32x4*>(tp + ((q ^ (l31 & 15)) << 2)) =
              f32x4{hh[rt][c][8 * o] + xa[rt][c][8 * o] * S_LO_INV, hh[rt][c][8 * o + 1] + xa[rt][c][8 * o + 1] * S_LO_INV,
                    hh[rt][c][8 * o + 2] + xa[rt][c][8 * o + 2] * S_LO_INV, hh[rt][c][8 * o + 3] + xa[rt][c][8 * o + 3] * S_LO_INV};
          *reinterpret_cast<f32x4*>(tp + (((q + 1) ^ (l31 & 15)) << 2)) =
              f32x4{hh[rt][c][8 * o + 4] + xa[rt][c][8 * o + 4] * S_LO_INV, hh[rt][c][8 * o + 5] + xa[rt][c][8 * o + 5] * S_LO_INV,
                    hh[rt][c][8 * o + 6] + xa[rt][c][8 * o + 6] * S_LO_INV, hh[rt][c][8 * o + 7] + xa[rt][c][8 * o + 7] * S_LO_INV};
        }
      if (rt == 0) MB_RES(1)                                               // (the next row tile's residuals fly behind this one's stores)
#pragma unroll
      for (int it = 0; it < 4; ++it) {
        const int px = 8 * it + psub, row = 32 * rt + px;
        const f32x4 t0 = *reinterpret_cast<const f32x4*>(tile + px * 64 + (((2 * gq) ^ (px & 15)) << 2));
        const f32x4 t1 = *reinterpret_cast<const f32x4*>(tile + px * 64 + (((2 * gq + 1) ^ (px & 15)) << 2));
        float v[8] = {t0[0], t0[1], t0[2], t0[3], t1[0], t1[1], t1[2], t1[3]};
#pragma unroll
        for (int k = 0; k < 4; ++k) { v[k] += b0[k]; v[4 + k] += b1[k]; }
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] += (float)rh[rt][it][k] + (float)rl[rt][it][k] * S_LO_INV;
        ammc_u4 hi, lo;
        ammc_s16_split8(v, hi, lo);
        if (r0 + row < n) {
#pragma unroll
          for (int k = 0; k < 8; ++k) bad |= !(fabsf(v[k]) <= 65504.f);
          float* yp = a.y + pout[rt][it] + cw;
          *reinterpret_cast<ammc_u4*>(yp) = hi;
          *reinterpret_cast<ammc_u4*>(yp + 4) = lo;
        }
      }
    }
#undef MB_RES
    if (bad && a.overflow_flag) atomicOr(a.overflow_flag, 1);
  }

  // ---- E: the last workgroup sums the commit partials (sum_partials_kernel's expressions, operation for operation) -------------
  MB_STAMP(8)
  __syncthreads();                                             // (every partial store of this workgroup has completed: see C)
  if (tid == 0) is_last = __hip_atomic_fetch_add(a.counter, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (int)gridDim.x - 1;
  __syncthreads();
  if (is_last) {
    if (tid < 256) {
      double s = 0.0;
      for (int i = tid; i < a.nparts; i += 256)
        s += (double)__hip_atomic_load(a.diff_partial + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      dred[tid] = s;
    }
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
      if (tid < o) dred[tid] += dred[tid + o];
      __syncthreads();
    }
    if (tid == 0) {
      a.diff[0] = (float)(dred[0] * (double)a.inv_count);
      __hip_atomic_store(a.counter, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
  MB_STAMP(9)
}

// S16 filter rows [N][K] (ammc_pack_conv_weight_f32 + ammc_split_rows_f32) -> FRAGMENT-major [K/8][hi | lo][N][8 halfs] with
// the rows of every 32-row tile in MFMA order (row l of a tile = filter pi(l), bits 2 and 3 of l swapped): the A operand
// of memory_block_s16_kernel's dec phase - 32 lanes read 512 contiguous bytes
__global__ __launch_bounds__(256) void pack_frag_rows_s16_kernel(const float* __restrict__ w, int n, int k, float* __restrict__ out) {
  const int64_t gid = (int64_t)blockIdx.x * 256 + threadIdx.x;           // one 16-byte piece
  const int kg = k >> 3;
  if (gid >= (int64_t)n * kg * 2) return;
  const int r = (int)(gid % n);
  const int gp = (int)(gid / n);                                          // 2 g + plane
  const int l = r & 31;
  const int src_row = (r & ~31) | (l & 19) | ((l & 4) << 1) | ((l & 8) >> 1);
  *reinterpret_cast<f32x4*>(out + gid * 4) = *reinterpret_cast<const f32x4*>(w + (int64_t)src_row * k + (gp >> 1) * 8 + (gp & 1) * 4);
}

}  // namespace ammc_impl
using namespace ammc_impl;

extern "C" int ammc_pack_frag_rows_s16(const float* w_s16, int32_t n, int32_t k, float* out, void* stream) {
  if (!w_s16 || !out || n <= 0 || (n % 32) || k <= 0 || (k % 8) || (((uintptr_t)w_s16 | (uintptr_t)out) & 15)) return AMMC_EINVAL;
  const int64_t pieces = (int64_t)n * (k >> 3) * 2;
  hipLaunchKernelGGL(pack_frag_rows_s16_kernel, dim3((unsigned)((pieces + 255) / 256)), dim3(256), 0, (hipStream_t)stream, w_s16, n, k,
                     out);
  return ammc_launch_status();
}

// (profiling only, not part of the ABI header) the s_memtime stamps of the last launch made with AMMC_MB_STAMPS=1
extern "C" int ammc_dbg_memory_block_stamps(unsigned long long* out32) {
  return (int)hipMemcpyFromSymbol(out32, HIP_SYMBOL(g_mb_stamps), sizeof(unsigned long long) * 32);
}

extern "C" int ammc_memory_block_s16(const float* x, int64_t x_bs, int64_t x_rs, int64_t x_ps, float* y, int64_t y_bs, int64_t y_rs,
                                     int64_t y_ps, int32_t batch, int32_t h, int32_t w, int32_t c, const float* enc_w,
                                     const float* enc_b, const void* e_s16, const float* embed_md, const float* enorm, int32_t d,
                                     int32_t m, int32_t k, const float* dec_wf, const float* dec_b, int32_t* idx_topk,
                                     float* q_topk, float* q_one, float* diff_partial, float* diff, int32_t* counter,
                                     int32_t* overflow_flag, void* stream) {
  if (!x || !y || !enc_w || !enc_b || !e_s16 || !embed_md || !enorm || !dec_wf || !dec_b || !idx_topk || !diff_partial || !diff ||
      !counter)
    return AMMC_EINVAL;
  if (batch <= 0 || h <= 0 || w <= 0 || m <= 0 || k <= 0 || k > m) return AMMC_EINVAL;
  if ((x_bs | x_rs | x_ps | y_bs | y_rs | y_ps) & 7) return AMMC_EINVAL;
  if ((((uintptr_t)x | (uintptr_t)y) & 31) || (((uintptr_t)enc_w | (uintptr_t)dec_wf | (uintptr_t)enc_b | (uintptr_t)dec_b) & 15))
    return AMMC_EINVAL;
  if (d != SD || k != 2 || c != 512 || m > 2048) return AMMC_EUNSUP;       // the shipped block's shape; else the five-launch chain
  const int64_t n64 = (int64_t)batch * h * w;
  if (n64 >= (1LL << 31)) return AMMC_EUNSUP;
  if ((int64_t)batch * x_bs >= (1LL << 31) || (int64_t)batch * y_bs >= (1LL << 31)) return AMMC_EUNSUP;      // 32-bit pixel offsets
  MemBlockArgs a;
  a.x = x; a.x_bs = x_bs; a.x_rs = x_rs; a.x_ps = x_ps;
  a.y = y; a.y_bs = y_bs; a.y_rs = y_rs; a.y_ps = y_ps;
  a.enc_w = enc_w; a.enc_b = enc_b; a.dec_wf = dec_wf; a.dec_b = dec_b;
  a.e_s16 = reinterpret_cast<const h16x8*>(e_s16); a.e_md = embed_md; a.enorm = enorm;
  a.n = (int)n64; a.hw = h * w; a.w = w; a.m = m; a.mpad = (m + 31) / 32 * 32; a.nparts = (a.n + 31) / 32;
  a.idx = idx_topk; a.q_topk = q_topk; a.q_one = q_one;
  a.diff_partial = diff_partial; a.diff = diff; a.counter = counter; a.inv_count = 1.f / ((float)a.n * (float)SD);
  a.overflow_flag = overflow_flag;
  static const int stamps = getenv("AMMC_MB_STAMPS") ? atoi(getenv("AMMC_MB_STAMPS")) : 0;
  a.stamps = stamps;
  constexpr int lds = MB_DYN_FLOATS * (int)sizeof(float);
  static const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void*>(memory_block_s16_kernel<512>),
                                                     hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  if (attr != hipSuccess) return (int)attr;
  hipLaunchKernelGGL(memory_block_s16_kernel<512>, dim3((a.n + 63) / 64), dim3(512), lds, (hipStream_t)stream, a);
  return ammc_launch_status();
}

extern "C" int ammc_pack_codebook_s16_guarded(const float* embed_dm, int32_t d, int32_t m, void* e_s16, int32_t* range_flag,
                                              void* stream) {
  if (!embed_dm || !e_s16 || d <= 0 || (d % 8) || m <= 0) return AMMC_EINVAL;
  const int mpad = (m + 31) / 32 * 32;
  const int64_t total = (int64_t)(d >> 3) * mpad;
  hipLaunchKernelGGL(pack_codebook_s16_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     embed_dm, d, m, mpad, reinterpret_cast<_Float16*>(e_s16), range_flag);
  return ammc_launch_status();
}

extern "C" int ammc_pack_codebook_s16(const float* embed_dm, int32_t d, int32_t m, void* e_s16, void* stream) {
  return ammc_pack_codebook_s16_guarded(embed_dm, d, m, e_s16, nullptr, stream);
}

// same workgroup geometry as ammc_memory_topk_fwd_f32: diff_partial has ammc_memory_topk_blocks(n) entries
extern "C" int ammc_memory_topk_fwd_s16(const float* x, const void* e_s16, const float* embed_md, const float* enorm,
                                        int32_t n, int32_t d, int32_t m, int32_t k, int32_t* idx_topk, float* q_topk,
                                        float* q_one, float* diff_partial, void* stream) {
  if (!x || !e_s16 || !embed_md || !enorm || !idx_topk || !q_topk || !diff_partial) return AMMC_EINVAL;
  if (n <= 0 || m <= 0 || k <= 0 || k > m) return AMMC_EINVAL;
  if (d != SD || k > 4) return AMMC_EUNSUP;                       // the model's embedding width; else the fp32 kernel
  hipStream_t s = (hipStream_t)stream;
  switch (k) {
    case 1: return launch_topk_s16<1>(x, e_s16, embed_md, enorm, n, m, idx_topk, q_topk, q_one, diff_partial, s);
    case 2: return launch_topk_s16<2>(x, e_s16, embed_md, enorm, n, m, idx_topk, q_topk, q_one, diff_partial, s);
    case 3: return launch_topk_s16<3>(x, e_s16, embed_md, enorm, n, m, idx_topk, q_topk, q_one, diff_partial, s);
    default: return launch_topk_s16<4>(x, e_s16, embed_md, enorm, n, m, idx_topk, q_topk, q_one, diff_partial, s);
  }
}
