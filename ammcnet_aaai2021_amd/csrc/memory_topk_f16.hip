// Memory addressing with fp16 MFMA operands and fp32 accumulation: the stress form of
// `Quantize_topk.forward` (reference Code/models/unet.py:282-297, 310-313) for large memories
// (BASELINE.json config 5: 8192 slots x 512-d features).  Same outputs as memory_topk.hip; the
// distance GEMM runs on v_mfma_f32_32x32x16_f16 (16x the fp32 MFMA rate), so the slot RANKING is
// computed from fp16-rounded features/slots, while the gathered rows, q_one and the commit
// distance use the fp32 codebook.  It is not the parity path: indices can differ from the
// fp32 result where two slots are closer than fp16 rounding noise (tests bound that).
//
//   - a workgroup (8 waves) owns 128 feature rows, staged once into LDS as fp16 with the 16-B
//     slots XOR-swizzled by row (conflict-free ds_read_b128 of the B fragments);
//   - the codebook is pre-packed k-blocked, [D/8][Mpad][8 halfs]: the A fragment of a lane
//     (8 consecutive features of one slot) is ONE coalesced 16-B global load, 512 B per half
//     wave; loads run a ring of PF steps ahead of the MFMAs - the codebook (8 MB at config 5)
//     streams from L2 / Infinity Cache and never touches LDS;
//   - wave w contracts slot tiles in PAIRS (2 w, 2 w + 1), (2 w + 16, ...): a feature fragment read from LDS feeds
//     two MFMAs.  With one tile per wave the kernel sat on the LDS ceiling - four 1-KB fragment reads per four MFMAs
//     and wave, eight waves: 256 B/clk/CU, all the LDS has - at 0.17 of the fp16 MFMA peak; pairs halve that.  Each
//     lane keeps the running top-K of its feature row in registers; the 16 partial lists per row are merged through LDS.
// Roofline: MFMA fp16 (2*d*m flop per row against ~2*d + 4*k*d bytes): AI in the thousands.
#include "ammc_common.h"
#include <hip/hip_fp16.h>
#include <math.h>
#include <algorithm>
#include <mutex>

namespace ammc_impl {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

constexpr int HBR = 128;          // feature rows per workgroup
constexpr int HRT = HBR / 32;
constexpr int HWAVES = 8;
constexpr int PF = 4;             // A-fragment prefetch depth (6: the same; 8: spills) (k-steps of 16; a step is 8 MFMAs = 256 cycles of the pipe)
constexpr int TS = 2;             // slot tiles per wave iteration

template <int K>
__device__ __forceinline__ void topk_insert16(float (&v)[K], int (&ix)[K], float c, int s) {
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

// The same insertion when candidates arrive in increasing slot order (one lane's walk over its slot tiles: tiles ascend,
// and so do the 16 slots of a tile's registers): a tie always loses to the entry already held, so strict comparisons
// implement the (value, index) order - branch-free for the model's K = 2 (as in memory_topk.hip).
template <int K>
__device__ __forceinline__ void topk_insert16_ordered(float (&v)[K], int (&ix)[K], float c, int s) {
  if (K == 2) {
    // (after the first tiles almost no candidate beats the second best of its row: the whole wave skips the update)
    if (c < v[1]) {
      const bool lt0 = c < v[0];
      v[1] = lt0 ? v[0] : c;
      ix[1] = lt0 ? ix[0] : s;
      v[0] = lt0 ? c : v[0];
      ix[0] = lt0 ? s : ix[0];
    }
  } else {
    topk_insert16<K>(v, ix, c, s);
  }
}

// NSTEP = d / 16, a template argument: the k-loop is fully unrolled (across the back edge of a run-time loop the
// compiler drains the register ring with vmcnt(0) every PF steps)
// FUSED = 1: one launch does everything (contraction, merge, gather / commit).  FUSED = 0 (launches of more than one
// round of workgroups): the kernel stops after writing the indices, and holds itself to SPLIT_VGPRS registers so that
// the waves of memory_gather_f16_kernel - which does the gather / commit of an EARLIER chunk of rows on a second
// stream - fit beside its two waves per SIMD (2 x 232 + 32 <= 512): the HBM-bound tail (3.2 GB per 262144 rows, 511 us
// of the 2717-us launch when it ran serially behind every workgroup's contraction) then streams
// while the matrix pipe works on the next rows.  `blk0`: first row block of this launch.
constexpr int SPLIT_VGPRS = 232;
template <int K, int NSTEP, int FUSED>
__device__ __forceinline__ void memory_topk_f16_body(
    const float* __restrict__ x, const f16x8* __restrict__ e_kblk /* [d/8][mpad] */,
    const float* __restrict__ e_md, const float* __restrict__ enorm16, int n, int d, int m, int mpad,
    int* __restrict__ idx_out, float* __restrict__ q_topk, float* __restrict__ q_one,
    float* __restrict__ diff_partial, int blk0) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  // region 0: the x tile [HBR][d] halfs (swizzled 16-B slots) during the contraction, then re-used for
  // the candidate lists [HBR][16][K] (value, index); region 1: |x|^2, final indices, reduction scratch
  // (candidate rows are 16 K + 1 words long: with 16 K the 32 lanes of a half wave - consecutive feature rows - wrote
  // and read words 128 B apart, i.e. two banks: 32-way conflicts, the 15 % of this kernel's LDS cycles that the round-2
  // counters showed as SQ_LDS_BANK_CONFLICT)
  constexpr int CROW = 16 * K + 1;
  const size_t region0 = max((size_t)HBR * d * 2, (size_t)HBR * CROW * 8);
  _Float16* xs = reinterpret_cast<_Float16*>(smem_raw);
  float* cand_v = reinterpret_cast<float*>(smem_raw);
  int* cand_i = reinterpret_cast<int*>(cand_v + HBR * CROW);
  float* xx = reinterpret_cast<float*>(smem_raw + region0);             // [HBR] (unused since the ranking dropped |x|^2)
  int* best = reinterpret_cast<int*>(xx + HBR);                         // [HBR][K]
  float* red = reinterpret_cast<float*>(best + HBR * K);                // [512]
  float* enl = red + 512;                                               // [HWAVES][TS * 32]: slot norms of a wave's tile pair

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int h = lane >> 5, l31 = lane & 31;
  const int r0 = (blk0 + (int)blockIdx.x) * HBR;
  const int slots = d >> 3;                                             // 16-B slots (8 halfs) per row

  // ---- stage x as fp16 -----------------------------------------------------------------
  for (int p = tid; p < HBR * slots; p += 512) {
    const int row = p / slots, sl = p - row * slots;
    f16x8 hv;
    if (r0 + row < n) {
      const f32x4 a = *reinterpret_cast<const f32x4*>(x + (int64_t)(r0 + row) * d + sl * 8);
      const f32x4 b = *reinterpret_cast<const f32x4*>(x + (int64_t)(r0 + row) * d + sl * 8 + 4);
#pragma unroll
      for (int i = 0; i < 4; ++i) { hv[i] = (_Float16)a[i]; hv[4 + i] = (_Float16)b[i]; }
    } else {
#pragma unroll
      for (int i = 0; i < 8; ++i) hv[i] = (_Float16)0.f;
    }
    *reinterpret_cast<f16x8*>(xs + (size_t)row * d + ((sl ^ (row & 15)) << 3)) = hv;
  }
  __syncthreads();
  float bv[HRT][K];
  int bi[HRT][K];
#pragma unroll
  for (int t = 0; t < HRT; ++t)
#pragma unroll
    for (int j = 0; j < K; ++j) { bv[t][j] = INFINITY; bi[t][j] = 0x7fffffff; }

  const int ntile = mpad >> 5;
  constexpr int nstep = NSTEP;                            // k-steps of 16 features (d == 16 NSTEP: the launcher checks)
  for (int tile = wave * TS; tile < ntile; tile += HWAVES * TS) {
    // lane (l31, h) at step t needs features [16t + 8h, +8) of slot s0 + l31: k-block 2t + h
    const f16x8* ep[TS];
    f32x16 acc[TS][HRT];
    f16x8 ring[TS][PF];
    // The codebook fragments run PF k-steps ahead in a register ring.  Every load of this loop is UNCONDITIONAL (the
    // steps past the end re-load the last one) and the k-loop is fully unrolled: a prefetch behind a condition is counted
    // by the compiler's waitcnt pass as not issued, and the back edge of a run-time k-loop drains the ring, so the first
    // form of this loop waited vmcnt(0) at EVERY step - a prefetch depth of one step instead of PF.  The request of step
    // t + PF is pinned before the MFMAs of step t (sched_barrier); the source offset advances incrementally behind an
    // opaque asm, like the swizzle of the LDS fragment addresses (hoisted out of the tile loop, the 32 x 4 addresses of
    // the unrolled steps spill).
    // |E_s|^2 of the 64 slots of this tile pair: one 4-byte load per lane, requested first (older than every ring load),
    // parked in a register and handed to the wave's row of `enl` right before the epilogue.  (It was an LDS-DMA: hipcc
    // books __builtin_amdgcn_global_load_lds as a FLAT access, after which every wait it inserts for an LDS read is
    // lgkmcnt(0) - the fragment reads of the k-loop could not be issued ahead of the MFMAs that do not need them.)
    float en_reg;
    {
      const int s_ = ((tile + (lane >> 5) < ntile ? tile + (lane >> 5) : ntile - 1) << 5) + l31;
      en_reg = enorm16[s_ < m ? s_ : m - 1];
    }
    const int64_t kstride = (int64_t)2 * mpad;                           // f16x8 elements between k-steps
    const int swz = l31 & 15;
    const _Float16* xrow = xs + (size_t)l31 * d;
#pragma unroll
    for (int u = 0; u < TS; ++u) {
      const int tl = tile + u < ntile ? tile + u : ntile - 1;           // (an odd tail tile is contracted twice, inserted once)
      ep[u] = e_kblk + (int64_t)h * mpad + (tl << 5) + l31;
#pragma unroll
      for (int t = 0; t < HRT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[u][t][r] = 0.f;
    }
#pragma unroll
    for (int p = 0; p < PF; ++p)                                         // nstep >= 8 > PF (d % 128 == 0)
#pragma unroll
      for (int u = 0; u < TS; ++u) ring[u][p] = ep[u][(int64_t)p * kstride];
    int64_t eoff = (int64_t)PF * kstride;        // wave uniform; opaque so that the 32 offsets are not pre-computed
    asm volatile("" : "+s"(eoff));               // (on the pointer itself the asm would turn it into a FLAT pointer)
    __builtin_amdgcn_sched_barrier(0);
    // feature fragments one k-step ahead in a second register set (left to itself hipcc reads each fragment right in
    // front of its two MFMAs and waits lgkmcnt(0): four exposed LDS latencies per k-step, 48 % of the matrix pipe)
    f16x8 bf[2][HRT];
#define MT_LOAD_B(t_, buf_)                                                                                \
  {                                                                                                        \
    int sw_ = swz;                                                       /* (opaque: the 32 x 4 fragment addresses of the */ \
    asm volatile("" : "+v"(sw_));                                        /*  unrolled steps must not be hoisted into registers) */ \
    const int so_ = (((2 * (t_) + h) ^ sw_) << 3);                                                         \
    _Pragma("unroll") for (int rt = 0; rt < HRT; ++rt)                                                     \
      bf[buf_][rt] = *reinterpret_cast<const f16x8*>(xrow + (size_t)(rt * 32) * d + so_);                  \
  }
    MT_LOAD_B(0, 0)
#pragma unroll
    for (int t = 0; t < nstep; ++t) {
      const int p = t % PF;
      f16x8 af[TS];
#pragma unroll
      for (int u = 0; u < TS; ++u) {
        af[u] = ring[u][p];
        ring[u][p] = ep[u][eoff];                                        // step t + PF (past the end: the last step again)
      }
      if (t + PF + 1 < nstep) {
        eoff += kstride;
        asm volatile("" : "+s"(eoff));
      }
      MT_LOAD_B((t + 1 < nstep ? t + 1 : t), (t + 1) & 1)
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_waitcnt(0xC07F | (HRT << 8));                   // lgkmcnt(HRT): everything but the reads just issued
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int rt = 0; rt < HRT; ++rt)
#pragma unroll
        for (int u = 0; u < TS; ++u) acc[u][rt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[u], bf[t & 1][rt], acc[u][rt], 0, 0, 0);
    }
#undef MT_LOAD_B
    enl[wave * (TS * 32) + lane] = en_reg;                               // (same wave writes and reads: no barrier)
    // The ranking needs |E_s|^2 - 2 x.E_s only (|x|^2 is the same for every slot of a row; the commit distance is
    // recomputed in fp32 by the gather phase): one FMA per candidate.  A tile's 16 candidates of a row are first reduced
    // to their minimum (v_min3: 8 operations); only a tile whose minimum beats the row's K-th best - after the first few
    // tiles almost none - runs the ordered insertion.  (The first form spent ~770 VALU operations per tile pair and wave
    // on this epilogue, a third of the pair's MFMA time; now ~200.)
#pragma unroll
    for (int u = 0; u < TS; ++u) {
      if (tile + u >= ntile) break;
      const int s0 = (tile + u) << 5;
      float en[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int o = (r & 3) + 8 * (r >> 2) + 4 * h;
        en[r] = s0 + o < m ? enl[wave * (TS * 32) + u * 32 + o] : INFINITY;       // slots beyond m never win
      }
#pragma unroll
      for (int rt = 0; rt < HRT; ++rt) {
        float dist[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) dist[r] = __builtin_fmaf(acc[u][rt][r], -2.f, en[r]);
        float mn = __builtin_fminf(__builtin_fminf(dist[0], dist[1]), dist[2]);
#pragma unroll
        for (int r = 3; r + 1 < 16; r += 2) mn = __builtin_fminf(__builtin_fminf(mn, dist[r]), dist[r + 1]);
        mn = __builtin_fminf(mn, dist[15]);
        if (mn < bv[rt][K - 1]) {
#pragma unroll
          for (int r = 0; r < 16; ++r) topk_insert16_ordered<K>(bv[rt], bi[rt], dist[r], s0 + (r & 3) + 8 * (r >> 2) + 4 * h);
        }
      }
    }
  }

  // ---- merge 16 partial lists per row ------------------------------------------------------
  __syncthreads();                     // every wave is done reading the x tile: region 0 changes hands
#pragma unroll
  for (int t = 0; t < HRT; ++t)
#pragma unroll
    for (int j = 0; j < K; ++j) {
      const int o = (t * 32 + l31) * CROW + (wave * 2 + h) * K + j;
      cand_v[o] = bv[t][j];
      cand_i[o] = bi[t][j];
    }
  __syncthreads();
  if (tid < HBR) {
    float v[K];
    int ix[K];
#pragma unroll
    for (int j = 0; j < K; ++j) { v[j] = INFINITY; ix[j] = 0x7fffffff; }
    for (int c = 0; c < 16 * K; ++c) topk_insert16<K>(v, ix, cand_v[tid * CROW + c], cand_i[tid * CROW + c]);
#pragma unroll
    for (int j = 0; j < K; ++j) {
      best[tid * K + j] = ix[j];
      if (r0 + tid < n) idx_out[(int64_t)(r0 + tid) * K + j] = ix[j];
    }
  }
  if (!FUSED) return;                  // (the gather / commit of these rows is memory_gather_f16_kernel's)
  __syncthreads();

  // ---- gather (fp32 codebook) + commit distance (fp32 features re-read from HBM) ------------
  float part = 0.f;
  const int slots4 = d >> 2;
  for (int p = tid; p < HBR * K * slots4; p += 512) {
    const int sl = p % slots4;
    const int rj = p / slots4;
    const int j = rj % K, row = rj / K;
    if (r0 + row >= n) continue;
    const int s = best[row * K + j];
    const f32x4 e = *reinterpret_cast<const f32x4*>(e_md + (int64_t)s * d + sl * 4);
    *reinterpret_cast<f32x4*>(q_topk + ((int64_t)(r0 + row) * K + j) * d + sl * 4) = e;
    if (j == 0) {
      const f32x4 xv = *reinterpret_cast<const f32x4*>(x + (int64_t)(r0 + row) * d + sl * 4);
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
  for (int o = 256; o > 0; o >>= 1) {
    if (tid < o) red[tid] += red[tid + o];
    __syncthreads();
  }
  if (tid == 0) diff_partial[blk0 + blockIdx.x] = red[0];
}

// (the register cap is an attribute, which takes literals only: two entry points around one body.  On gfx90a and later
// the backend DOUBLES the attribute's value - it budgets the unified file as architectural + accumulation registers - so
// amdgpu_num_vgpr(116) is what yields "NumVgprs: 232, ScratchSize: 0" in the ISA, and (16) caps the gather kernel at 32)
template <int K, int NSTEP>
__global__ __launch_bounds__(512, 1) void memory_topk_f16_kernel(
    const float* __restrict__ x, const f16x8* __restrict__ e_kblk, const float* __restrict__ e_md,
    const float* __restrict__ enorm16, int n, int d, int m, int mpad, int* __restrict__ idx_out, float* __restrict__ q_topk,
    float* __restrict__ q_one, float* __restrict__ diff_partial, int blk0) {
  memory_topk_f16_body<K, NSTEP, 1>(x, e_kblk, e_md, enorm16, n, d, m, mpad, idx_out, q_topk, q_one, diff_partial, blk0);
}
template <int K, int NSTEP>
__global__ __launch_bounds__(512, 1) __attribute__((amdgpu_num_vgpr(116))) void memory_topk_f16_split_kernel(
    const float* __restrict__ x, const f16x8* __restrict__ e_kblk, const float* __restrict__ e_md,
    const float* __restrict__ enorm16, int n, int d, int m, int mpad, int* __restrict__ idx_out, float* __restrict__ q_topk,
    float* __restrict__ q_one, float* __restrict__ diff_partial, int blk0) {
  static_assert(SPLIT_VGPRS == 232, "the attribute above takes a literal");
  memory_topk_f16_body<K, NSTEP, 0>(x, e_kblk, e_md, enorm16, n, d, m, mpad, idx_out, q_topk, q_one, diff_partial, blk0);
}

// The gather / commit phase as its own kernel (FUSED = 0 launches): one workgroup of 256 threads per 128-row block,
// <= 32 VGPRs and no LDS to speak of, so that it co-resides with the contraction kernel.  Row block b = blk0 + blockIdx.x;
// diff_partial[b] is summed in a fixed order (thread-strided partial sums, then a tree): deterministic.
template <int K>
__global__ __launch_bounds__(256) __attribute__((amdgpu_num_vgpr(16))) void memory_gather_f16_kernel(
    const float* __restrict__ x, const float* __restrict__ e_md, const int* __restrict__ idx, int n, int d,
    float* __restrict__ q_topk, float* __restrict__ q_one, float* __restrict__ diff_partial, int blk0) {
  __shared__ float red[256];
  __shared__ int best[HBR * K];
  const int tid = threadIdx.x;
  const int b = blk0 + (int)blockIdx.x;
  const int r0 = b * HBR;
  for (int i = tid; i < HBR * K; i += 256) best[i] = r0 + i / K < n ? idx[(int64_t)r0 * K + i] : 0;
  __syncthreads();
  float part = 0.f;
  const int slots4 = d >> 2;
  for (int p = tid; p < HBR * K * slots4; p += 256) {
    const int sl = p % slots4;
    const int rj = p / slots4;
    const int j = rj % K, row = rj / K;
    if (r0 + row >= n) continue;
    const int s = best[row * K + j];
    const f32x4 e = *reinterpret_cast<const f32x4*>(e_md + (int64_t)s * d + sl * 4);
    *reinterpret_cast<f32x4*>(q_topk + ((int64_t)(r0 + row) * K + j) * d + sl * 4) = e;
    if (j == 0) {
      const f32x4 xv = *reinterpret_cast<const f32x4*>(x + (int64_t)(r0 + row) * d + sl * 4);
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
  if (tid == 0) diff_partial[b] = red[0];
}

// [d][m] fp32 -> k-blocked fp16 [d/8][mpad][8] (slots >= m zero) and |half(E_s)|^2
__global__ __launch_bounds__(256) void pack_codebook_f16_kernel(const float* __restrict__ e_dm, int d, int m, int mpad,
                                                                _Float16* __restrict__ out, float* __restrict__ enorm16) {
  const int s = blockIdx.x * 256 + threadIdx.x;
  if (s >= mpad) return;
  float nrm = 0.f;
  for (int kb = 0; kb < (d >> 3); ++kb) {
    f16x8 hv;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const float v = s < m ? e_dm[(int64_t)(kb * 8 + i) * m + s] : 0.f;
      hv[i] = (_Float16)v;
      nrm += (float)hv[i] * (float)hv[i];
    }
    *reinterpret_cast<f16x8*>(out + ((int64_t)kb * mpad + s) * 8) = hv;
  }
  if (s < m) enorm16[s] = nrm;
}

// Second stream + events of the split form, made once per process (the only state the library keeps: a call that takes
// the split path records / waits events on the caller's stream, allocates nothing and never synchronises the host).
// Calls from several host threads serialise on the mutex for the few microseconds of their enqueue.
constexpr int SPLIT_MAX_CHUNKS = 8;
constexpr int SPLIT_ROUND_BLOCKS = 256;          // one workgroup per CU: a chunk is a whole number of rounds
struct SplitState {
  hipStream_t side = nullptr;
  hipEvent_t chunk_done[SPLIT_MAX_CHUNKS];
  hipEvent_t side_done = nullptr;
  int device = -1;
  bool ok = false;
};
static SplitState g_split;
static std::mutex g_split_mu;

static bool split_ready() {
  int dev = -1;
  if (hipGetDevice(&dev) != hipSuccess) return false;
  if (g_split.ok && g_split.device == dev) return true;
  if (g_split.ok) return false;                   // made for another device: that process keeps the fused form here
  if (hipStreamCreateWithFlags(&g_split.side, hipStreamNonBlocking) != hipSuccess) return false;
  for (int i = 0; i < SPLIT_MAX_CHUNKS; ++i)
    if (hipEventCreateWithFlags(&g_split.chunk_done[i], hipEventDisableTiming) != hipSuccess) return false;
  if (hipEventCreateWithFlags(&g_split.side_done, hipEventDisableTiming) != hipSuccess) return false;
  g_split.device = dev;
  g_split.ok = true;
  return true;
}

template <int K, int NSTEP>
int launch_topk16n(const float* x, const void* e_kblk, const float* e_md, const float* enorm16, int n, int d, int m,
                   int* idx, float* q_topk, float* q_one, float* diff_partial, hipStream_t stream) {
  const size_t region0 = std::max((size_t)HBR * d * 2, (size_t)HBR * (16 * K + 1) * 8);
  const size_t lds = region0 + sizeof(float) * (HBR + HBR * K + 512 + HWAVES * TS * 32);
  if (lds > 160 * 1024) return AMMC_EUNSUP;
  auto fused = memory_topk_f16_kernel<K, NSTEP>;
  auto split = memory_topk_f16_split_kernel<K, NSTEP>;
  if (lds > 48 * 1024) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(fused), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(split), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return (int)e;
  }
  const int mpad = (m + 31) / 32 * 32;
  const int nblk = (n + HBR - 1) / HBR;
  // "memory_split" (AMMC_MEMORY_SPLIT): 1 = split, else ONE fused launch (the default).  Measured at 262144 rows (round 4,
  // profiles/r04_stress_split_trace.txt): the trace shows the gather of chunk c running beside the contraction of chunk
  // c + 1 as designed, but the contraction of a chunk then takes 295 us instead of 282 (it shares L2 / HBM and the power
  // budget with the gather, which itself stretches from 154 to 240 us at the one wave per SIMD the register file leaves
  // it) and the last chunk's gather is exposed: 2560 us per call against 2580 fused - 1 %, not worth a second stream
  // by default.
  const int opt = ammc_opt_memory_split();
  hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
  (void)hipStreamIsCapturing(stream, &cap);
  bool use_split = opt == 1 && cap == hipStreamCaptureStatusNone;
  std::unique_lock<std::mutex> lock(g_split_mu, std::defer_lock);
  if (use_split) {
    lock.lock();
    use_split = split_ready();
  }
  if (!use_split) {
    hipLaunchKernelGGL(fused, dim3(nblk), dim3(512), lds, stream, x, reinterpret_cast<const f16x8*>(e_kblk), e_md, enorm16, n,
                       d, m, mpad, idx, q_topk, q_one, diff_partial, 0);
    return ammc_launch_status();
  }
  // chunks of whole rounds, at most SPLIT_MAX_CHUNKS: contraction of chunk c on the caller's stream, its gather / commit
  // on the side stream behind an event - i.e. beside the contraction of chunk c + 1; the caller's stream finally waits
  // for the side stream, so the call's outputs are complete for whatever the caller enqueues next
  int rounds = (nblk + SPLIT_ROUND_BLOCKS - 1) / SPLIT_ROUND_BLOCKS;
  int rounds_per_chunk = (rounds + SPLIT_MAX_CHUNKS - 1) / SPLIT_MAX_CHUNKS;
  const int chunk_blocks = rounds_per_chunk * SPLIT_ROUND_BLOCKS;
  int c = 0;
  for (int b0 = 0; b0 < nblk; b0 += chunk_blocks, ++c) {
    const int nb = std::min(chunk_blocks, nblk - b0);
    hipLaunchKernelGGL(split, dim3(nb), dim3(512), lds, stream, x, reinterpret_cast<const f16x8*>(e_kblk), e_md, enorm16, n, d,
                       m, mpad, idx, q_topk, q_one, diff_partial, b0);
    if (hipEventRecord(g_split.chunk_done[c], stream) != hipSuccess) return ammc_launch_status();
    if (hipStreamWaitEvent(g_split.side, g_split.chunk_done[c], 0) != hipSuccess) return ammc_launch_status();
    hipLaunchKernelGGL(memory_gather_f16_kernel<K>, dim3(nb), dim3(256), 0, g_split.side, x, e_md, idx, n, d, q_topk, q_one,
                       diff_partial, b0);
  }
  if (hipEventRecord(g_split.side_done, g_split.side) != hipSuccess) return ammc_launch_status();
  if (hipStreamWaitEvent(stream, g_split.side_done, 0) != hipSuccess) return ammc_launch_status();
  return ammc_launch_status();
}

template <int K>
int launch_topk16(const float* x, const void* e_kblk, const float* e_md, const float* enorm16, int n, int d, int m,
                  int* idx, float* q_topk, float* q_one, float* diff_partial, hipStream_t stream) {
  switch (d) {                                  // d % 128 == 0, d <= 512 (checked by the caller)
    case 128: return launch_topk16n<K, 8>(x, e_kblk, e_md, enorm16, n, d, m, idx, q_topk, q_one, diff_partial, stream);
    case 256: return launch_topk16n<K, 16>(x, e_kblk, e_md, enorm16, n, d, m, idx, q_topk, q_one, diff_partial, stream);
    case 384: return launch_topk16n<K, 24>(x, e_kblk, e_md, enorm16, n, d, m, idx, q_topk, q_one, diff_partial, stream);
    case 512: return launch_topk16n<K, 32>(x, e_kblk, e_md, enorm16, n, d, m, idx, q_topk, q_one, diff_partial, stream);
  }
  return AMMC_EUNSUP;
}

}  // namespace ammc_impl
using namespace ammc_impl;

extern "C" int ammc_pack_codebook_f16(const float* embed_dm, int32_t d, int32_t m, void* e_kblk_f16, float* enorm16,
                                      void* stream) {
  if (!embed_dm || !e_kblk_f16 || !enorm16 || d <= 0 || (d % 16) || m <= 0) return AMMC_EINVAL;
  const int mpad = (m + 31) / 32 * 32;
  hipLaunchKernelGGL(pack_codebook_f16_kernel, dim3((mpad + 255) / 256), dim3(256), 0, (hipStream_t)stream, embed_dm, d,
                     m, mpad, reinterpret_cast<_Float16*>(e_kblk_f16), enorm16);
  return ammc_launch_status();
}

extern "C" int ammc_memory_topk_f16_blocks(int32_t n) { return n <= 0 ? 0 : (n + HBR - 1) / HBR; }

extern "C" int ammc_memory_topk_fwd_f16(const float* x, const void* e_kblk_f16, const float* embed_md,
                                        const float* enorm16, int32_t n, int32_t d, int32_t m, int32_t k,
                                        int32_t* idx_topk, float* q_topk, float* q_one, float* diff_partial,
                                        void* stream) {
  if (!x || !e_kblk_f16 || !embed_md || !enorm16 || !idx_topk || !q_topk || !diff_partial) return AMMC_EINVAL;
  if (n <= 0 || m <= 0 || k <= 0 || k > m) return AMMC_EINVAL;
  if (d < 128 || (d % 128) || d > 512) return AMMC_EUNSUP;      // 16 swizzle positions per row need >= 16 slots
  if (k > 4) return AMMC_EUNSUP;
  hipStream_t s = (hipStream_t)stream;
  switch (k) {
    case 1: return launch_topk16<1>(x, e_kblk_f16, embed_md, enorm16, n, d, m, idx_topk, q_topk, q_one, diff_partial, s);
    case 2: return launch_topk16<2>(x, e_kblk_f16, embed_md, enorm16, n, d, m, idx_topk, q_topk, q_one, diff_partial, s);
    case 3: return launch_topk16<3>(x, e_kblk_f16, embed_md, enorm16, n, d, m, idx_topk, q_topk, q_one, diff_partial, s);
    default: return launch_topk16<4>(x, e_kblk_f16, embed_md, enorm16, n, d, m, idx_topk, q_topk, q_one, diff_partial, s);
  }
}
