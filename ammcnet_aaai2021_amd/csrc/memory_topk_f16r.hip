// Memory addressing with fp16 MFMA operands, ROWS RESIDENT IN REGISTERS (round 5): the form of
// `Quantize_topk.forward` (reference Code/models/unet.py:282-297, 310-313) for large memories (BASELINE.json config 5:
// 8192 slots x 512-d) that replaces memory_topk_f16.hip where it applies.  Same outputs, same arithmetic contract: the
// slot RANKING is computed from fp16-rounded features / slots with fp32 accumulation, the gathered rows, q_one and the
// commit distance come from the fp32 codebook and the fp32 features.
//
// Why: memory_topk_f16.hip keeps 128 feature rows in LDS and every workgroup streams the whole fp16 codebook (8 MB) from
// L2 into REGISTERS - 256 B per MFMA through the per-CU vector-memory path, 16.8 GB per 262144-row launch; that stream,
// not the matrix pipe, set its time (0.40 of the pipe busy).  Here the roles are swapped:
//   - a wave (one per SIMD, up to 512 VGPRs) keeps RT = 3 tiles of 32 feature rows x 512 features as fp16 B-FRAGMENTS IN
//     ITS REGISTERS (384 VGPRs) for a whole sweep of the codebook: 384 rows per workgroup and sweep instead of 128;
//   - the codebook, pre-packed tile by tile in fragment order (ammc_pack_codebook_f16_tiles), goes L2 -> LDS by LDS-DMA
//     (global_load_lds_dwordx4, a ring of four 33-KB tiles, counted vmcnt + one raw s_barrier per tile) ONCE per workgroup
//     and is read from LDS by the four waves: 85 B of L2 traffic and 341 B of LDS reads per MFMA (was 256 + 512);
//   - the distance epilogue costs nothing: the accumulators START at -|E_s|^2 / 2 (the first MFMA of a slot tile takes
//     them as its C operand from the 1 KB of constants that rides with the tile), so the ranking key x.E_s - |E_s|^2 / 2
//     comes out of the contraction itself and a tile's 16 candidates of a row are only reduced with v_max3 against the
//     row's K-th best;
//   - a lane owns a feature row: the running top-K lives in registers, the two halves of the wave (slots 4h .. 4h+3 of
//     every group of 8) are merged with one lane exchange at the end of the sweep - no LDS candidate lists, no
//     workgroup-wide merge;
//   - workgroups are persistent: each takes a contiguous run of row tiles and sweeps the codebook ceil(run / (4 RT))
//     times, the last sweep with fewer tiles per wave (dispatched on the count: no MFMA is issued for rows that do not
//     exist), so 262144 rows on 256 CUs are three sweeps of 3 + 3 + 2 tiles per wave with no tail round.
// The feature rows reach their fragment registers through LDS as well (the ring is idle between sweeps): whole 128-byte
// lines by LDS-DMA, 16-byte pieces XOR-swizzled by row on the SOURCE side so that the fragment reads are conflict free.
// Roofline: MFMA fp16, 2 * n * d * m flop.
#include "ammc_common.h"
#include <hip/hip_fp16.h>
#include <math.h>
#include <algorithm>
#include <stdlib.h>
#include <type_traits>

namespace ammc_f16r {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

constexpr int RW = 4;             // waves per workgroup (one per SIMD)
constexpr int NBUF = 4;           // codebook tiles in the LDS ring: one being read, one landed, two in flight
constexpr int NA = 4;             // A-fragment register ring (LDS reads NA - 1 k-steps ahead of their MFMAs)

struct F16rArgs {
  const float* x;                 // [n][d] fp32
  const unsigned char* tiles;     // ammc_pack_codebook_f16_tiles
  const float* e_md;              // [m][d] fp32
  int* idx_out;                   // [n][K]
  float* q_topk;                  // [n][K][d]
  float* q_one;                   // [n][d] or null
  float* diff_partial;            // [ceil(n / 32)]
  int n, d, m, ntile;             // ntile = ceil(m / 32)
  int t32;                        // row tiles of 32 rows: ceil(n / 32)
  int pw;                         // row tiles per wave
  int no_inline_tail;             // A/B (AMMC_F16R_TAIL_INLINE=0): the pipelined form writes every sweep's rows out serially
};

// LDS-DMA of 16 bytes per lane with the instruction hidden from the compiler (see conv_tap_s16.hip: hipcc books the
// builtin as a FLAT access, after which every LDS wait it inserts is lgkmcnt(0)); vmcnt waits are written by hand.
// `base` wave-uniform, `off` this lane's byte offset, `lds_byte` the wave-uniform LDS address of the wave's 1 KB.
__device__ __forceinline__ void r_dma16(const void* base, unsigned off, unsigned lds_byte) {
  unsigned keep;
  asm volatile("s_nop 4\n\ts_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(off), "s"(base), "s"(lds_byte) : "memory");
}
// the same with a full address per lane
__device__ __forceinline__ void r_dma16_ptr(const void* src, unsigned lds_byte) {
  unsigned keep;
  asm volatile("s_nop 4\n\ts_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(src), "s"(lds_byte) : "memory");
}
// 16 bytes per lane to global memory, hidden from the compiler like the DMAs (its vmcnt bookkeeping must not see one kind
// of vector-memory operation and miss the other); `base` wave-uniform
// AMMC_F16R_NT (compile time, A/B builds): 1 = the stores carry the non-temporal hint - 1.6 GB of gathered rows / q_one per
// 262144-row launch that nothing on the device reads again should not push the codebook out of the L2 / Infinity Cache.
// Measured (round 6, one box, A/B/A/B): 2.284 / 2.297 ms without, 2.276 / 2.281 with - 0.4 %, inside the noise: left off.
#ifndef AMMC_F16R_NT
#define AMMC_F16R_NT 0
#endif
__device__ __forceinline__ void r_store16(void* base, unsigned off, f32x4 v) {
#if AMMC_F16R_NT
  asm volatile("s_nop 4\n\tglobal_store_dwordx4 %0, %1, %2 nt\n\ts_nop 1" ::"v"(off), "v"(v), "s"(base) : "memory");
#else
  asm volatile("s_nop 4\n\tglobal_store_dwordx4 %0, %1, %2\n\ts_nop 1" ::"v"(off), "v"(v), "s"(base) : "memory");
#endif
}
#define R_VMCNT(n_) asm volatile("s_waitcnt vmcnt(" #n_ ")" ::: "memory")

// f(integral_constant<int, I>) for I = I0 .. N - 1, every call inlined: a k-loop whose step number is a constant
// expression (register arrays indexed by it stay in registers whatever the body holds; `#pragma unroll` gives up - and
// parks the fragment registers in scratch - once the body, with the top-K update inlined at two of its steps, is large)
template <int I, int N, class F>
__device__ __forceinline__ void r_static_for(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    r_static_for<I + 1, N>(f);
  }
}

// candidates arrive in increasing slot order within a lane: a tie loses to the entry already held (strict comparison =
// the (value, index) order).  Keys are MAXIMISED (x.E - |E|^2 / 2).
template <int K>
__device__ __forceinline__ void r_insert_ordered(float (&v)[K], int (&ix)[K], float c, int s) {
  if (c > v[K - 1]) {
    v[K - 1] = c;
    ix[K - 1] = s;
#pragma unroll
    for (int j = K - 1; j > 0; --j) {
      const bool sw = v[j] > v[j - 1];
      const float tv = sw ? v[j - 1] : v[j];
      const int ti = sw ? ix[j - 1] : ix[j];
      v[j - 1] = sw ? v[j] : v[j - 1];
      ix[j - 1] = sw ? ix[j] : ix[j - 1];
      v[j] = tv;
      ix[j] = ti;
    }
  }
}
// the model's K = 2, candidates in increasing slot order: one compare decides whether the WAVE does anything for this
// accumulator register (after the first tiles of a sweep almost no candidate beats the second best of its row), then four
// selects - no sorting network
__device__ __forceinline__ void r_insert_ordered2(float (&v)[2], int (&ix)[2], float c, int s) {
  if (c > v[1]) {
    const bool lt0 = c > v[0];
    v[1] = lt0 ? v[0] : c;
    ix[1] = lt0 ? ix[0] : s;
    v[0] = lt0 ? c : v[0];
    ix[0] = lt0 ? s : ix[0];
  }
}
// general insertion ((value desc, index asc) order): the merge of the two lane halves
template <int K>
__device__ __forceinline__ void r_insert(float (&v)[K], int (&ix)[K], float c, int s) {
  if (c > v[K - 1] || (c == v[K - 1] && s < ix[K - 1])) {
    v[K - 1] = c;
    ix[K - 1] = s;
#pragma unroll
    for (int j = K - 1; j > 0; --j) {
      const bool sw = v[j] > v[j - 1] || (v[j] == v[j - 1] && ix[j] < ix[j - 1]);
      const float tv = sw ? v[j - 1] : v[j];
      const int ti = sw ? ix[j - 1] : ix[j];
      v[j - 1] = sw ? v[j] : v[j - 1];
      ix[j - 1] = sw ? ix[j] : ix[j - 1];
      v[j] = tv;
      ix[j] = ti;
    }
  }
}

// running top-K of one row tile from one accumulator tile: register r of a lane is slot s0 + (r & 3) + 8 (r >> 2) + 4 h
// (ascending in r).  Two-level screen: the maximum of each group of four registers, then of the tile - 10 VALU operations
// and ONE branch that the whole wave takes when no candidate of the tile beats the K-th best of its row; else one branch
// per group and, only inside a group that holds a candidate, one per register (a sweep inserts ~2 candidates per tile
// and row tile: the flat form - sixteen compare / exec-mask / branch sequences whenever the tile held one - cost 19 %)
template <int K, int DBG>
__device__ __forceinline__ void r_update(const f32x16& acc, float (&bv)[K], int (&bi)[K], int s0, int h, int m) {
  float gm[4];
#pragma unroll
  for (int q = 0; q < 4; ++q)
    gm[q] = __builtin_fmaxf(__builtin_fmaxf(__builtin_fmaxf(acc[4 * q], acc[4 * q + 1]), acc[4 * q + 2]), acc[4 * q + 3]);
  const float mx = __builtin_fmaxf(__builtin_fmaxf(__builtin_fmaxf(gm[0], gm[1]), gm[2]), gm[3]);
  if (DBG & 1) {
    bv[0] = __builtin_fmaxf(bv[0], mx);
    bi[0] = s0 & (m - 1);
    if (K > 1) bi[K > 1 ? 1 : 0] = (s0 + 1) & (m - 1);
  } else if (mx > bv[K - 1]) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      if (gm[q] > bv[K - 1]) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int r = 4 * q + i;
          if constexpr (K == 2) r_insert_ordered2(bv, bi, acc[r], s0 + i + 8 * q + 4 * h);
          else r_insert_ordered<K>(bv, bi, acc[r], s0 + i + 8 * q + 4 * h);
        }
      }
    }
  }
}

// ---- branch-free running top-2 on PACKED keys (the pipelined form, K = 2) ----------------------------------------------
// A candidate's key x.E - |E|^2 / 2 carries its own accumulator register in its four lowest mantissa bits (15 - r), and the running
// (best, second) of a row is updated by TWO operations per candidate, no compare, no branch:
//     second = med3(key, best, second);  best = max(key, best)
// - 3 VALU operations per candidate with the packing, 16 candidates per tile and row tile, spread one candidate per
// k-step behind the NEXT tile's MFMAs (the compare-and-branch form, even hidden behind a second accumulator set, cost 15 %
// of the sweep: ~100 scalar / vector instructions per tile that the in-order wave issues instead of MFMAs).  The slot
// tile of the two survivors is tracked once per tile (6 operations).  What changes against the exact compare: candidates closer than 2^-19 of their
// magnitude - far inside the noise of the fp32 accumulation order, let alone of the fp16 operands - rank by register.
// Round 6 (advisor): keys x.E - |E|^2 / 2 are NEGATIVE for any feature that is not close to a slot (config 5's random
// features: -207 +- 16), and among negative floats the larger mantissa is the smaller value: two candidates equal in the
// upper 28 bits - duplicated codebook rows - then come out of the sweep as (higher slot, lower slot).  A sign-aware tag
// ((15 - r) ^ sign: two more VALU operations per candidate) fixed that and cost 6.7 % of the kernel (2.07 -> 2.22 ms,
// A/B/A/B/A/B on one box: the update issues in the shadow of the MFMAs and that shadow is full).  Instead the final pair of
// every row is put in slot order when its two keys tie in the upper 28 bits (`r_sweep`, behind the merge of the lane
// halves: once per row, not per candidate) - which also orders ties ACROSS lanes and slot tiles, where no tag could.
// What remains: three or more candidates with identical upper 28 bits (a codebook row present three times) may return
// two of them that are not the two lowest slots.
__device__ __forceinline__ float r_pack_key(float v, int r) {
  return __builtin_bit_cast(float, (__builtin_bit_cast(unsigned, v) & 0xFFFFFFF0u) | (unsigned)(15 - r));
}
__device__ __forceinline__ void r_top2_step(float key, float& b0, float& b1) {
  // (asm: from __builtin_fmaxf / fmed3f hipcc first canonicalises the bit-built key - v_max_f32 k, k, k - a quarter more
  // VALU work.  No key is a NaN: slots beyond m carry the finite -3e38, see pack_codebook_f16_tiles_kernel)
  float n1, n0;
  asm("v_med3_f32 %0, %1, %2, %3" : "=v"(n1) : "v"(key), "v"(b0), "v"(b1));
  asm("v_max_f32 %0, %1, %2" : "=v"(n0) : "v"(key), "v"(b0));
  b1 = n1;
  b0 = n0;
}
// after the 16 candidates of slot tile `tile`: which tile do the two survivors come from?  (o0, o1: the pair before)
__device__ __forceinline__ void r_top2_track(float b0, float b1, float o0, float o1, int& t0, int& t1, int tile) {
  const bool ch0 = b0 != o0;                        // a strictly better candidate arrived
  const float from = ch0 ? o0 : o1;                 // where an unchanged-looking second would have come from
  const int tfrom = ch0 ? t0 : t1;
  t1 = (b1 == from) ? tfrom : tile;
  t0 = ch0 ? tile : t0;
}

// what a sweep leaves for later: the lookups of its row tiles (lane r < 32 holds row r's slots), still to be written out
template <int K>
struct RCarry {
  int bi[2][K];
  int tile0, cnt, pending;
};

// gather (fp32 codebook rows), q_one, commit partial sum of ONE row tile, serially: UN (row, 256-float chunk) units per trip
template <int K, int D, bool Q1>
__device__ __forceinline__ void r_tail_tile(const F16rArgs& a, const int (&bi)[K], int rtile, int lane) {
  constexpr int NCH = (D + 255) / 256;
  constexpr int UN = 8;                              // 24 loads of 1 KB in flight per wave: the tail runs at HBM speed
  const int r0 = rtile * 32;
  if (r0 >= a.n) return;
  const int nrow = a.n - r0 < 32 ? a.n - r0 : 32;
  float part = 0.f;
  for (int u0 = 0; u0 < nrow * NCH; u0 += UN) {
    f32x4 e[UN][K], xv[UN];
    bool on[UN];
    int64_t xo[UN];
#pragma unroll
    for (int u = 0; u < UN; ++u) {
      const int uu = u0 + u < nrow * NCH ? u0 + u : nrow * NCH - 1;       // (a short last trip repeats its last unit, unwritten)
      const int r = uu / NCH, ch = uu - r * NCH;
      const int off = ch * 256 + lane * 4;
      on[u] = off < D && u0 + u < nrow * NCH;
      const int offc = off < D ? off : 0;
      xo[u] = (int64_t)(r0 + r) * D + offc;
#pragma unroll
      for (int j = 0; j < K; ++j) {
        const int s = __builtin_amdgcn_readlane(bi[j], r);
        e[u][j] = *reinterpret_cast<const f32x4*>(a.e_md + (int64_t)s * D + offc);
      }
      xv[u] = *reinterpret_cast<const f32x4*>(a.x + xo[u]);
    }
#pragma unroll
    for (int u = 0; u < UN; ++u) {
      if (!on[u]) continue;
      const int uu = u0 + u;
      const int r = uu / NCH, ch = uu - r * NCH;
      const int64_t qo = ((int64_t)(r0 + r) * K) * D + ch * 256 + lane * 4;
#pragma unroll
      for (int j = 0; j < K; ++j) *reinterpret_cast<f32x4*>(a.q_topk + qo + (int64_t)j * D) = e[u][j];
      f32x4 q1;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float df = e[u][0][i] - xv[u][i];
        part += df * df;
        q1[i] = xv[u][i] + df;
      }
      if (Q1) *reinterpret_cast<f32x4*>(a.q_one + xo[u]) = q1;
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) part += __shfl_xor(part, o, 64);
  if (lane == 0) a.diff_partial[rtile] = part;
}

// One sweep of the codebook for CNT row tiles of this wave (rows [row0 + 32 i, +32), i < CNT; rows >= n are clamped for
// the loads and never written).  Everything of the sweep: feature staging, contraction + running top-K, the merge of
// the lane halves, indices, gather / q_one / commit partials.
// DBG (measurement builds of the K = 2, d = 512 instance only, AMMC_F16R_DBG): 1 = no top-K update, 2 = no codebook DMA
// inside the sweep (the MFMAs read stale LDS), 4 = no gather / commit tail, 8 = no feature staging; results are then wrong
template <int K, int NSTEP, int CNT, bool Q1, int DBG = 0, bool PIPE = false>
__device__ __forceinline__ void r_sweep(const F16rArgs& a, unsigned char* smem, unsigned lds0, int lane, int uwave, int tile0,
                                        RCarry<K>& prev, RCarry<K>& cur) {
  constexpr int TB = (NSTEP + 1) * 1024;           // bytes of a codebook tile image
  constexpr int D = NSTEP * 16;
  constexpr int PH = D / 8;                        // 16-byte pieces of half a feature row (fp32)
  constexpr int BPH = PH / 16;                     // 256-byte blocks of half a feature row
  constexpr int DPT = NSTEP / 4 + 1;               // DMA instructions per tile and wave (the constants' KB by every wave)
  const int h = lane >> 5, l31 = lane & 31;

  // ---- features -> fp16 B fragments in registers, through this wave's quarter of the (idle) ring -------------------------
  // Row tiles 0 and 1 end up in AGPRs, tile 2 in VGPRs (see the contraction).  A value only STAYS in an AGPR if it is born
  // there - hipcc copies a VGPR-born value into a scratch AGPR in front of every asm that wants one - and the one
  // instruction that writes a 128-bit AGPR tuple in one go is an LDS read: the converted fragments of tiles 0 / 1 go back
  // to LDS (in place: into the 256-byte block of their own row they were just read from) and are read into AGPRs by asm.
  constexpr int CA = CNT < 2 ? CNT : 2;
  f16x8 xa[CA][NSTEP];                               // AGPR-resident ("a" operands only)
  f16x8 xv[CNT == 3 ? NSTEP : 1];                    // VGPR-resident third tile
  {
    // A chunk = 32 rows x PC pieces: half a feature row, or a quarter where that is still a whole number of 256-byte
    // blocks (d % 256 == 0) - then the wave's 32 KB hold TWO chunks and the DMA of chunk c + 1 is in flight while chunk c
    // is converted (one exposed HBM latency per sweep instead of one per chunk: 6 x ~10 us per sweep before)
    constexpr bool QUARTER = (D % 256) == 0;
    constexpr int NCH = QUARTER ? 4 : 2;             // chunks per row tile
    constexpr int PC = D / (4 * NCH);                // 16-byte pieces of a chunk row (fp32)
    constexpr int NI = PC / 2;                       // DMA instructions per chunk (32 rows x PC pieces / 64 lanes)
    constexpr int SPC = NSTEP / NCH;                 // k-steps per chunk
    const unsigned stage = (unsigned)uwave * (unsigned)(32 * PH * 16);       // this wave's 32 rows x half a row of LDS
    auto issue_chunk = [&](int c) {
      const int rt = c / NCH, kc = c % NCH;
      const int r0 = (tile0 + rt) * 32;
      const unsigned reg = stage + (QUARTER ? (unsigned)(c & 1) * (unsigned)(32 * PC * 16) : 0u);
      // 32 x PC pieces, piece g = (row, cc): LDS slot g linear; the SOURCE piece of slot (row, cc) is cc ^ (row & 15) within
      // its 256-byte block, so that the 16 lanes of a ds_read_b128 group (16 distinct rows mod 16) hit 16 distinct slots
#pragma unroll 2
      for (int i = 0; i < ((DBG & 8) ? 0 : NI); ++i) {
        const int g = i * 64 + lane;
        const int row = g / PC, cc = g % PC;
        const int srcp = (cc & ~15) | ((cc & 15) ^ (row & 15));
        int gr = r0 + row;
        gr = gr < a.n ? gr : a.n - 1;
        r_dma16_ptr(a.x + (int64_t)gr * D + kc * (D / NCH) + srcp * 4, lds0 + reg + (unsigned)i * 1024u);
      }
    };
    constexpr int NC = CNT * NCH;
    issue_chunk(0);
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      const int rt = c / NCH, kc = c % NCH;
      if (QUARTER && c + 1 < NC) {
        issue_chunk(c + 1);
        if (!(DBG & 8)) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NI) : "memory");       // chunk c landed, c + 1 in flight
      } else {
        if (!QUARTER && c > 0) issue_chunk(c);                                 // (one region: requested only now that chunk c - 1 is out)
        R_VMCNT(0);
      }
      unsigned char* sp = smem + stage + (QUARTER ? (unsigned)(c & 1) * (unsigned)(32 * PC * 16) : 0u);
      const unsigned sreg = lds0 + stage + (QUARTER ? (unsigned)(c & 1) * (unsigned)(32 * PC * 16) : 0u);
      // lane (row l31, half h), k-step t = kc SPC + tt: features 16 t + 8 h .. + 7 = pieces 4 tt + 2 h, + 1 of the chunk
      // row; four k-steps = one 256-byte block of the row
#pragma unroll
      for (int q = 0; q < SPC / 4; ++q) {
        f16x8 hv[4];
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) {
          const int c0 = 16 * q + 4 * s4 + 2 * h;                              // even: its partner is c0 ^ 1 after the swizzle too
          const int p0 = (c0 & ~15) | ((c0 & 15) ^ (l31 & 15));
          const f32x4 v0 = *reinterpret_cast<const f32x4*>(sp + ((unsigned)(l31 * PC + p0) << 4));
          const f32x4 v1 = *reinterpret_cast<const f32x4*>(sp + ((unsigned)(l31 * PC + (p0 ^ 1)) << 4));
#pragma unroll
          for (int j = 0; j < 4; ++j) { hv[s4][j] = (_Float16)v0[j]; hv[s4][4 + j] = (_Float16)v1[j]; }
        }
        const int tb_ = kc * SPC + 4 * q;
        if (rt == 2) {
#pragma unroll
          for (int s4 = 0; s4 < 4; ++s4) xv[CNT == 3 ? tb_ + s4 : 0] = hv[s4];
        } else {
          // (LDS operations of one wave execute in order: the block's fp32 pieces have been read by both of its lanes)
          unsigned wa[4];
#pragma unroll
          for (int s4 = 0; s4 < 4; ++s4) {
            const unsigned wo = ((unsigned)(l31 * PC + 16 * q + ((2 * s4 + h) ^ (l31 & 15))) << 4);
            *reinterpret_cast<f16x8*>(sp + wo) = hv[s4];
            wa[s4] = sreg + wo;
          }
          asm volatile("" ::: "memory");
          // (one statement, ending in its own wait: an asm output the compiler believes complete must BE complete - it is
          // free to copy the registers right behind the statement)
          asm volatile("ds_read_b128 %0, %4\n\tds_read_b128 %1, %5\n\tds_read_b128 %2, %6\n\tds_read_b128 %3, %7\n\t"
                       "s_waitcnt lgkmcnt(0)"
                       : "=&a"(xa[rt < CA ? rt : 0][tb_]), "=&a"(xa[rt < CA ? rt : 0][tb_ + 1]), "=&a"(xa[rt < CA ? rt : 0][tb_ + 2]),
                         "=&a"(xa[rt < CA ? rt : 0][tb_ + 3])
                       : "v"(wa[0]), "v"(wa[1]), "v"(wa[2]), "v"(wa[3])
                       : "memory");
        }
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                       // this chunk's reads are done before its region is refilled
    }
  }
  asm volatile("" ::: "memory");
  __builtin_amdgcn_s_barrier();                    // every wave is done with its staging region: the ring may be filled
  asm volatile("" ::: "memory");

  // ---- codebook ring ---------------------------------------------------------------------------------------------------
  const unsigned lane_off = (unsigned)lane * 16u + (unsigned)uwave * 1024u;
  auto issue_tile = [&](int tile) {
    const int tl = tile < a.ntile ? tile : a.ntile - 1;                        // (past the end: a valid source, a free slot, never read)
    const unsigned char* src = a.tiles + (int64_t)tl * TB;
    const unsigned dst = lds0 + (unsigned)(tile & (NBUF - 1)) * (unsigned)TB;
#pragma unroll
    for (int j = 0; j < NSTEP / 4; ++j)
      r_dma16(src + j * 4096, lane_off, dst + (unsigned)(j * 4096) + (unsigned)uwave * 1024u);
    r_dma16(src + NSTEP * 1024, (unsigned)lane * 16u, dst + (unsigned)(NSTEP * 1024));
  };
  float bv[CNT][K];
  int bi[CNT][K];
#pragma unroll
  for (int rt = 0; rt < CNT; ++rt)
#pragma unroll
    for (int j = 0; j < K; ++j) { bv[rt][j] = -INFINITY; bi[rt][j] = 0x7fffffff; }
#pragma unroll
  for (int p = 0; p < NBUF - 1; ++p) issue_tile(p);

  if constexpr (PIPE) {
    // ---- the pipelined form (CNT <= 2: all row fragments in AGPRs, ~130 VGPRs free) ---------------------------------------
    // TWO accumulator sets: the MFMAs of tile i write one while the top-K update of tile i - 1 reads the other, its VALU
    // instructions issued in the shadow of the MFMAs (a quarter of it per k-step) - with one set the matrix pipe drained
    // at every tile end (last MFMA's latency, the update, the barrier, the constants' LDS round trip: ~15 % of the sweep).
    // The tile loop is unrolled by two so that the set of a tile is a literal.
    static_assert(CNT <= 2, "the pipelined form keeps every row fragment in AGPRs");
    f32x16 acc[2][CNT];
#pragma unroll
    for (int rt = 0; rt < CNT; ++rt)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[1][rt][r] = -3.0e38f;                   // ("tile -1": nothing beats anything)
    constexpr bool PACKED = K == 2 && NSTEP >= 18;                               // (see r_top2_step; else r_update)
    // ---- the rows of the PREVIOUS sweep: their gather / q_one / commit write-out rides along with this sweep's first
    // tiles (K = 2, d = 512: a unit = one 256-float half of one row - two codebook rows and the feature row in, three rows
    // out - one unit per slot tile: its three LDS-DMAs a tile ahead into a 3-KB slot of this wave, read, summed and stored
    // behind the MFMAs of k-step 21).  Serially - every CU of the chip in its write-out at the same moments - those 2.7 GB
    // cost 0.28 ms of 2.15.  Carried tiles with rows past n, or more units than slot tiles, are written serially instead.
    constexpr bool TAIL_INLINE = PACKED && D == 512;
    constexpr int TSLOT = 3072, TWAVE = 2 * TSLOT;                               // LDS behind the ring: [wave][2 slots][e0 | e1 | x]
    const unsigned tl0 = (unsigned)(NBUF * TB) + (unsigned)uwave * (unsigned)TWAVE;
    int nunit = 0;
    float tpart0 = 0.f, tpart1 = 0.f;
    // (the carry's scalars are wave-uniform; readfirstlane tells the compiler, which would otherwise treat the loop over
    // the units - and every LDS address inside it - as divergent)
    const int pcnt = __builtin_amdgcn_readfirstlane(prev.cnt), ptile0 = __builtin_amdgcn_readfirstlane(prev.tile0);
    if (__builtin_amdgcn_readfirstlane(prev.pending)) {
      const bool full = (ptile0 + pcnt) * 32 <= a.n;
      if (TAIL_INLINE && !(DBG & 2) && !a.no_inline_tail && full && pcnt * 64 <= a.ntile) {
        nunit = pcnt * 64;
      } else {
        for (int rt = 0; rt < pcnt; ++rt) {
          if (rt == 0) r_tail_tile<K, D, Q1>(a, prev.bi[0], ptile0, lane);
          else r_tail_tile<K, D, Q1>(a, prev.bi[1], ptile0 + 1, lane);
        }
      }
      prev.pending = 0;
    }
    // the three LDS-DMAs of unit u (row tile u >> 6, row (u >> 1) & 31, half u & 1) into slot u & 1
    auto issue_unit = [&](int u) {
      // (everything here is wave-uniform - readfirstlane says so where the compiler cannot see it: "s" operands)
      const int uc = __builtin_amdgcn_readfirstlane(u < nunit ? u : nunit - 1);  // (the unit after the last: a valid re-load, never read)
      const int urt = uc >> 6, ur = (uc >> 1) & 31, uch = uc & 1;
      const int row = (ptile0 + urt) * 32 + ur;
      const unsigned dst = __builtin_amdgcn_readfirstlane(lds0 + tl0 + (unsigned)(u & 1) * (unsigned)TSLOT);
      const int s0_ = __builtin_amdgcn_readlane(urt ? prev.bi[1][0] : prev.bi[0][0], ur);
      const int s1_ = __builtin_amdgcn_readlane(urt ? prev.bi[1][K > 1 ? 1 : 0] : prev.bi[0][K > 1 ? 1 : 0], ur);
      r_dma16(a.e_md + (int64_t)s0_ * D + uch * 256, (unsigned)lane * 16u, (unsigned)__builtin_amdgcn_readfirstlane((int)dst));
      r_dma16(a.e_md + (int64_t)s1_ * D + uch * 256, (unsigned)lane * 16u, (unsigned)__builtin_amdgcn_readfirstlane((int)(dst + 1024u)));
      r_dma16(a.x + (int64_t)row * D + uch * 256, (unsigned)lane * 16u, (unsigned)__builtin_amdgcn_readfirstlane((int)(dst + 2048u)));
    };
    // unit u out of its slot: q_topk rows, q_one, the commit sum of its row tile
    auto drain_unit = [&](int u_) {
      const int u = __builtin_amdgcn_readfirstlane(u_);
      const int urt = u >> 6, ur = (u >> 1) & 31, uch = u & 1;
      const int row = (ptile0 + urt) * 32 + ur;
      const unsigned char* tp = smem + tl0 + (unsigned)(u & 1) * (unsigned)TSLOT + lane * 16;
      const f32x4 e0 = *reinterpret_cast<const f32x4*>(tp);
      const f32x4 e1 = *reinterpret_cast<const f32x4*>(tp + 1024);
      const f32x4 xr = *reinterpret_cast<const f32x4*>(tp + 2048);
      f32x4 q1;
      float sq = 0.f;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float df = e0[i] - xr[i];
        sq += df * df;
        q1[i] = xr[i] + df;
      }
      tpart0 += urt ? 0.f : sq;
      tpart1 += urt ? sq : 0.f;
      float* q0p = a.q_topk + ((int64_t)row * K) * D + uch * 256;
      r_store16(q0p, (unsigned)lane * 16u, e0);
      r_store16(q0p + D, (unsigned)lane * 16u, e1);
      if (Q1) r_store16(a.q_one + (int64_t)row * D + uch * 256, (unsigned)lane * 16u, q1);
    };
    float pb0[CNT], pb1[CNT], ob0[CNT], ob1[CNT];                                // packed keys: best, second; the pair a tile ago
    int pt0[CNT], pt1[CNT];                                                      // the slot tiles they come from
#pragma unroll
    for (int rt = 0; rt < CNT; ++rt) { pb0[rt] = pb1[rt] = ob0[rt] = ob1[rt] = -INFINITY; pt0[rt] = pt1[rt] = -1; }
#define R_BODY(tile_, B_, TI_)                                                                                         \
  {                                                                                                                     \
    /* with the write-out riding along (TI_): every tile issues 3 more LDS-DMAs and 2 - 3 stores; 2 DPT + 3 is what is  \
       younger than the tile's own codebook DMAs in the FIRST such tile (later ones: more), so the wait is never short */ \
    if (!(DBG & 2)) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * DPT + ((TI_) ? 3 : 0)) : "memory");                    \
    __builtin_amdgcn_s_barrier();                                                                                       \
    asm volatile("" ::: "memory");                                                                                      \
    const int nt_ = __builtin_amdgcn_readfirstlane((tile_) + NBUF - 1);                                                 \
    const int ntl_ = nt_ < a.ntile ? nt_ : a.ntile - 1;                                                                 \
    const unsigned char* nsrc = a.tiles + (int64_t)ntl_ * TB;                                                           \
    const unsigned ndst = lds0 + (unsigned)(nt_ & (NBUF - 1)) * (unsigned)TB;                                           \
    const unsigned char* tb = smem + (unsigned)((tile_) & (NBUF - 1)) * (unsigned)TB;                                   \
    const unsigned char* ap = tb + lane * 16;                                                                           \
    /* the accumulators START at -|E_s|^2 / 2: read straight into them (set B_ was last read a tile ago) */             \
    _Pragma("unroll") for (int q = 0; q < 4; ++q) {                                                                     \
      const f32x4 c4 = *reinterpret_cast<const f32x4*>(tb + NSTEP * 1024 + (8 * q + 4 * h) * 4);                       \
      _Pragma("unroll") for (int rt = 0; rt < CNT; ++rt)                                                                \
        _Pragma("unroll") for (int i = 0; i < 4; ++i) acc[B_][rt][4 * q + i] = c4[i];                                   \
    }                                                                                                                   \
    f16x8 ar[NA];                                                                                                       \
    _Pragma("unroll") for (int t = 0; t < NA - 1; ++t) ar[t] = *reinterpret_cast<const f16x8*>(ap + t * 1024);          \
    const int sp_ = ((tile_) - 1) << 5;                                                                                 \
    r_static_for<0, NSTEP>([&](auto T_) __attribute__((always_inline)) {                                                \
      constexpr int t = decltype(T_)::value;                                                                            \
      if constexpr (t + NA - 1 < NSTEP) ar[(t + NA - 1) % NA] = *reinterpret_cast<const f16x8*>(ap + (t + NA - 1) * 1024); \
      __builtin_amdgcn_sched_barrier(0);                                                                                \
      _Pragma("unroll") for (int rt = 0; rt < CNT; ++rt) {                                                              \
        asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc[B_][rt]) : "v"(ar[t % NA]), "a"(xa[rt][t]));   \
        if (!(DBG & 2) && rt == 0 && t % 3 == 1 && t / 3 < DPT) {                                                       \
          constexpr int j = t / 3;                                                                                      \
          if (j < NSTEP / 4) r_dma16(nsrc + j * 4096, lane_off, ndst + (unsigned)(j * 4096) + (unsigned)uwave * 1024u); \
          else r_dma16(nsrc + NSTEP * 1024, (unsigned)lane * 16u, ndst + (unsigned)(NSTEP * 1024));                     \
        }                                                                                                               \
      }                                                                                                                 \
      if constexpr ((TI_) && t == 2) issue_unit((tile_) + 1);                                                           \
      if constexpr ((TI_) && t == 21) {                                                                                 \
        /* unit tile_'s DMAs are older than DMA 0, the next unit's three and DMAs 1-6 of this tile: all but those 10 */ \
        asm volatile("s_waitcnt vmcnt(10)" ::: "memory");                                                               \
        drain_unit(tile_);                                                                                              \
      }                                                                                                                 \
      /* the previous tile's update behind this step's MFMAs: K = 2 - one candidate per k-step (packed keys, no      \
         branch); other K - the compare-and-branch form, one row tile per chosen k-step */                             \
      if constexpr (K == 2 && NSTEP >= 18) {                                                                            \
        if constexpr (t < 16 && !(DBG & 1)) {                                                                           \
          _Pragma("unroll") for (int rt = 0; rt < CNT; ++rt) {                                                          \
            if (t == 0) { ob0[rt] = pb0[rt]; ob1[rt] = pb1[rt]; }                                                       \
            r_top2_step(r_pack_key(acc[(B_) ^ 1][rt][t < 16 ? t : 0], t), pb0[rt], pb1[rt]);                            \
          }                                                                                                             \
        }                                                                                                               \
        if constexpr (t == 16 && !(DBG & 1)) {                                                                          \
          _Pragma("unroll") for (int rt = 0; rt < CNT; ++rt)                                                            \
            r_top2_track(pb0[rt], pb1[rt], ob0[rt], ob1[rt], pt0[rt], pt1[rt], (tile_) - 1);                            \
        }                                                                                                               \
      } else {                                                                                                          \
        if constexpr (t == NSTEP / 4) r_update<K, DBG>(acc[(B_) ^ 1][0], bv[0], bi[0], sp_, h, a.m);                    \
        if constexpr (CNT == 2 && t == (3 * NSTEP) / 4)                                                                 \
          r_update<K, DBG>(acc[(B_) ^ 1][CNT - 1], bv[CNT - 1], bi[CNT - 1], sp_, h, a.m);                              \
      }                                                                                                                 \
      __builtin_amdgcn_sched_barrier(0);                                                                                \
    });                                                                                                                 \
  }
    int tile_b = 0;
    if constexpr (TAIL_INLINE) {
      if (nunit > 0) {
        issue_unit(0);
        for (int tile = 0; tile < nunit; tile += 2) {                            // (nunit is even and <= ntile)
          R_BODY(tile, 0, true)
          R_BODY(tile + 1, 1, true)
        }
        tile_b = nunit;
        // the commit sums of the carried row tiles
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { tpart0 += __shfl_xor(tpart0, o, 64); tpart1 += __shfl_xor(tpart1, o, 64); }
        if (lane == 0) {
          a.diff_partial[ptile0] = tpart0;
          if (pcnt > 1) a.diff_partial[ptile0 + 1] = tpart1;
        }
      }
    }
    for (int tile = tile_b; tile < a.ntile; tile += 2) {
      R_BODY(tile, 0, false)
      if (tile + 1 < a.ntile) R_BODY(tile + 1, 1, false)
      else {                                                                   // (odd tile count: the set 1 of "tile + 1" stays unwritten)
#pragma unroll
        for (int rt = 0; rt < CNT; ++rt)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[1][rt][r] = -3.0e38f;
      }
    }
#undef R_BODY
    // the last tile's update (nothing left to hide it behind): wait out its MFMAs first
    {
      const bool odd = a.ntile & 1;
      if constexpr (CNT == 2) asm volatile("s_nop 15\n\ts_nop 7" : "+v"(acc[0][0]), "+v"(acc[0][1]), "+v"(acc[1][0]), "+v"(acc[1][1]));
      else asm volatile("s_nop 15\n\ts_nop 7" : "+v"(acc[0][0]), "+v"(acc[1][0]));
      const int sl = (a.ntile - 1) << 5;
#pragma unroll
      for (int rt = 0; rt < CNT; ++rt) {
        if constexpr (PACKED) {
          if (!(DBG & 1)) {
            const float o0 = pb0[rt], o1 = pb1[rt];
#pragma unroll
            for (int r = 0; r < 16; ++r) r_top2_step(r_pack_key(odd ? acc[0][rt][r] : acc[1][rt][r], r), pb0[rt], pb1[rt]);
            r_top2_track(pb0[rt], pb1[rt], o0, o1, pt0[rt], pt1[rt], a.ntile - 1);
          }
          // the survivors as (key, slot): register 15 - (key & 15) of tile pt, slot = 32 pt + (r & 3) + 8 (r >> 2) + 4 h
          const int r0_ = 15 - (int)(__builtin_bit_cast(unsigned, pb0[rt]) & 15u), r1_ = 15 - (int)(__builtin_bit_cast(unsigned, pb1[rt]) & 15u);
          bv[rt][0] = pb0[rt];
          bv[rt][K > 1 ? 1 : 0] = pb1[rt];
          bi[rt][0] = pt0[rt] < 0 ? 0x7fffffff : (pt0[rt] << 5) + (r0_ & 3) + 8 * (r0_ >> 2) + 4 * h;
          bi[rt][K > 1 ? 1 : 0] = pt1[rt] < 0 ? 0x7fffffff : (pt1[rt] << 5) + (r1_ & 3) + 8 * (r1_ >> 2) + 4 * h;
          if (DBG & 1) { bi[rt][0] = 0; bi[rt][K > 1 ? 1 : 0] = 1; }
        } else {
          if (odd) r_update<K, DBG>(acc[0][rt], bv[rt], bi[rt], sl, h, a.m);
          else r_update<K, DBG>(acc[1][rt], bv[rt], bi[rt], sl, h, a.m);
        }
      }
    }
  } else {
  for (int tile = 0; tile < a.ntile; ++tile) {
    // tile's own DMAs (issued three iterations ago) have landed for this wave once all but the two younger tiles' are
    // done; behind the barrier that holds for every wave, and everybody has finished reading tile - 1, whose slot the
    // tile after next then goes into
    if (!(DBG & 2)) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * DPT) : "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    // (the DMA requests of the tile after next are spread over the k-loop below, one every third k-step: issued in one
    // burst here they cost the wave ~9 x 60 cycles in which it feeds no MFMA)
    const int nt_ = tile + NBUF - 1;
    const int ntl_ = nt_ < a.ntile ? nt_ : a.ntile - 1;                        // (past the end: a valid source, a free slot, never read)
    const unsigned char* nsrc = a.tiles + (int64_t)ntl_ * TB;
    const unsigned ndst = lds0 + (unsigned)(nt_ & (NBUF - 1)) * (unsigned)TB;
    const unsigned char* tb = smem + (unsigned)(tile & (NBUF - 1)) * (unsigned)TB;
    const unsigned char* ap = tb + lane * 16;
    // C operand of the tile's first MFMAs: -|E_s|^2 / 2 of the 16 slots this lane's accumulator registers stand for
    f32x16 ci;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const f32x4 c4 = *reinterpret_cast<const f32x4*>(tb + NSTEP * 1024 + (8 * q + 4 * h) * 4);
#pragma unroll
      for (int i = 0; i < 4; ++i) ci[4 * q + i] = c4[i];
    }
    // The MFMAs are written in asm so that their register classes are OURS: the 3 x 128 fragment registers of the rows do
    // not fit the 256 architectural VGPRs, and hipcc, which keeps MFMA sources in VGPRs, parked two thirds of them in
    // AGPRs and copied four registers back per MFMA (v_accvgpr_read + the wait states behind it).  On gfx950 an MFMA takes
    // its B operand from an AGPR directly: row tiles 0 and 1 live in AGPRs ("a"), tile 2 and the accumulators in VGPRs
    // (the top-K screen below reads the accumulators with VALU instructions: no v_accvgpr_read either).
    f32x16 acc[CNT];
    f16x8 ar[NA];
#define R_MFMA_FIRST(rt_, a_, t_)                                                                                       \
  if (rt_ < 2) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %3" : "=&v"(acc[rt_]) : "v"(a_), "a"(xa[rt_ < CA ? rt_ : 0][t_]), "v"(ci)); \
  else asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %3" : "=&v"(acc[rt_]) : "v"(a_), "v"(xv[CNT == 3 ? t_ : 0]), "v"(ci));
#define R_MFMA_NEXT(rt_, a_, t_)                                                                                        \
  if (rt_ < 2) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc[rt_]) : "v"(a_), "a"(xa[rt_ < CA ? rt_ : 0][t_])); \
  else asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc[rt_]) : "v"(a_), "v"(xv[CNT == 3 ? t_ : 0]));
#pragma unroll
    for (int t = 0; t < NA - 1; ++t) ar[t] = *reinterpret_cast<const f16x8*>(ap + t * 1024);
#pragma unroll
    for (int t = 0; t < NSTEP; ++t) {
      if (t + NA - 1 < NSTEP) ar[(t + NA - 1) % NA] = *reinterpret_cast<const f16x8*>(ap + (t + NA - 1) * 1024);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int rt = 0; rt < CNT; ++rt) {
        if (t == 0) { R_MFMA_FIRST(rt, ar[t % NA], t) } else { R_MFMA_NEXT(rt, ar[t % NA], t) }
        if (!(DBG & 2) && rt == 0 && t % 3 == 1 && t / 3 < DPT) {               // behind the step's first MFMA: it runs meanwhile
          const int j = t / 3;
          if (j < NSTEP / 4) r_dma16(nsrc + j * 4096, lane_off, ndst + (unsigned)(j * 4096) + (unsigned)uwave * 1024u);
          else r_dma16(nsrc + NSTEP * 1024, (unsigned)lane * 16u, ndst + (unsigned)(NSTEP * 1024));
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
#undef R_MFMA_FIRST
#undef R_MFMA_NEXT
    // (nothing pads an asm statement: the VALU reads of the screen below must not issue sooner than 18 wait states after
    // the last MFMA that writes the registers they read - cdna4 ISA, XDL write VGPR -> VALU read)
    // (the accumulators are operands of the statement: without them hipcc moved the first reads ABOVE the nops)
    if constexpr (CNT == 3) asm volatile("s_nop 15\n\ts_nop 7" : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[CNT - 1]));
    else if constexpr (CNT == 2) asm volatile("s_nop 15\n\ts_nop 7" : "+v"(acc[0]), "+v"(acc[CNT - 1]));
    else asm volatile("s_nop 15\n\ts_nop 7" : "+v"(acc[0]));
    const int s0 = tile << 5;
#pragma unroll
    for (int rt = 0; rt < CNT; ++rt) r_update<K, DBG>(acc[rt], bv[rt], bi[rt], s0, h, a.m);
  }
  }
  R_VMCNT(0);                                      // (the clamped DMAs past the end)
  asm volatile("" ::: "memory");
  __builtin_amdgcn_s_barrier();                    // the ring changes hands: the next sweep stages features in it
  asm volatile("" ::: "memory");

  // ---- merge the two lane halves of a row (lanes l and l ^ 32), write the indices ----------------------------------------
#pragma unroll
  for (int rt = 0; rt < CNT; ++rt) {
    float ov[K];
    int oi[K];
#pragma unroll
    for (int j = 0; j < K; ++j) {
      ov[j] = __shfl_xor(bv[rt][j], 32, 64);
      oi[j] = __shfl_xor(bi[rt][j], 32, 64);
    }
#pragma unroll
    for (int j = 0; j < K; ++j) r_insert<K>(bv[rt], bi[rt], ov[j], oi[j]);
    if constexpr (PIPE && K == 2 && NSTEP >= 18) {   // (= the packed-key form, see r_pack_key: exact ties go to the lower slot)
      const unsigned k0_ = __builtin_bit_cast(unsigned, bv[rt][0]), k1_ = __builtin_bit_cast(unsigned, bv[rt][1]);
      const bool swap_ = ((k0_ ^ k1_) & 0xFFFFFFF0u) == 0u && bi[rt][0] > bi[rt][1];
      const int lo_ = swap_ ? bi[rt][1] : bi[rt][0], hi_ = swap_ ? bi[rt][0] : bi[rt][1];
      bi[rt][0] = lo_;
      bi[rt][1] = hi_;
    }
    const int row = (tile0 + rt) * 32 + l31;
    if (h == 0 && row < a.n) {
#pragma unroll
      for (int j = 0; j < K; ++j) a.idx_out[(int64_t)row * K + j] = bi[rt][j];
    }
  }

  // ---- gather / q_one / commit: right away (one accumulator set), or left to the NEXT sweep, which writes these rows out
  // behind its own MFMAs (pipelined form; the kernel flushes the last sweep's) ------------------------------------------------
  if constexpr (PIPE) {
#pragma unroll
    for (int rt = 0; rt < CNT; ++rt)
#pragma unroll
      for (int j = 0; j < K; ++j) cur.bi[rt][j] = bi[rt][j];
    cur.tile0 = tile0, cur.cnt = CNT, cur.pending = (DBG & 4) ? 0 : 1;
  } else {
#pragma unroll
    for (int rt = 0; rt < CNT; ++rt)
      if (!(DBG & 4)) r_tail_tile<K, D, Q1>(a, bi[rt], tile0 + rt, lane);
  }
}

template <int K, int NSTEP, int RT, bool Q1, int DBG = 0, bool PIPE = false>
__global__ __launch_bounds__(256, 1) void memory_topk_f16r_kernel(F16rArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_r[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int uwave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) void*)smem_r;
  // this wave's run of row tiles; every wave of the grid has `pw` of them (the last ones past the end of the rows)
  const int first = ((int)blockIdx.x * RW + uwave) * a.pw;
  // The sweeps of a workgroup: RT row tiles per wave at a time - but the FIRST sweep of every second and third workgroup
  // is one / two tiles short (3,3,2 | 2,3,3 | 1,3,3,1 for eight tiles per wave).  Workgroups of equal work run in lockstep:
  // all 256 CUs would stage their features and write their gathered rows - the HBM-bound quarter of the kernel, nothing of
  // which overlaps the contraction inside a workgroup (one wave per SIMD, every register taken) - at the same moments,
  // sharing the HBM bandwidth 256 ways while the matrix pipes of the whole chip wait.  Staggered, a third of the CUs is in
  // a memory phase at a time and the others' contractions run beside it.
  int cnt = RT - (int)(blockIdx.x % 3);
  if (cnt < 1 || a.pw <= RT) cnt = RT;
  if (cnt > a.pw) cnt = a.pw;
  RCarry<K> prev, cur;
  prev.pending = 0, prev.tile0 = 0, prev.cnt = 0;
  for (int p0 = 0; p0 < a.pw;) {
    int t0 = first + p0;
    // a wave whose tiles lie past the end still takes part in the ring (DMA shares, barriers): its loads clamp to the last
    // row and it writes nothing
    if (t0 >= a.t32) t0 = a.t32;
    cur.pending = 0;
    if constexpr (RT >= 3) {
      if (cnt == 3) r_sweep<K, NSTEP, 3, Q1, DBG, false>(a, smem_r, lds0, lane, uwave, t0, prev, cur);
      else if (cnt == 2) r_sweep<K, NSTEP, 2, Q1, DBG, false>(a, smem_r, lds0, lane, uwave, t0, prev, cur);
      else r_sweep<K, NSTEP, 1, Q1, DBG, false>(a, smem_r, lds0, lane, uwave, t0, prev, cur);
    } else {
      if (cnt == 2) r_sweep<K, NSTEP, 2, Q1, DBG, PIPE>(a, smem_r, lds0, lane, uwave, t0, prev, cur);
      else r_sweep<K, NSTEP, 1, Q1, DBG, PIPE>(a, smem_r, lds0, lane, uwave, t0, prev, cur);
    }
    prev = cur;
    p0 += cnt;
    cnt = a.pw - p0 < RT ? a.pw - p0 : RT;
  }
  if constexpr (PIPE) {                              // the last sweep's rows: nothing left to hide their write-out behind
    if (prev.pending) {
      for (int rt = 0; rt < prev.cnt; ++rt) {
        if (rt == 0) r_tail_tile<K, NSTEP * 16, Q1>(a, prev.bi[0], prev.tile0, lane);
        else r_tail_tile<K, NSTEP * 16, Q1>(a, prev.bi[1], prev.tile0 + 1, lane);
      }
    }
  }
}

// [d][m] fp32 -> tile images: tile T (slots 32 T .. 32 T + 31) = NSTEP KB of A fragments - KB t holds, for lane
// (l31, h), the 8 halfs of slot 32 T + l31, features 16 t + 8 h .. + 7 - and one KB of constants: float i < 32 =
// -|half(E_{32 T + i})|^2 / 2 (-3e38 for slots >= m: they never beat a real slot), the rest zero
__global__ __launch_bounds__(256) void pack_codebook_f16_tiles_kernel(const float* __restrict__ e_dm, int d, int m,
                                                                      unsigned char* __restrict__ out) {
  const int nstep = d >> 4;
  const int tile = blockIdx.x, tid = threadIdx.x;
  unsigned char* tb = out + (int64_t)tile * (nstep + 1) * 1024;
  for (int p = tid; p < nstep * 64; p += 256) {
    const int t = p >> 6, ln = p & 63;
    const int s = tile * 32 + (ln & 31), f0 = 16 * t + 8 * (ln >> 5);
    f16x8 hv;
#pragma unroll
    for (int i = 0; i < 8; ++i) hv[i] = (_Float16)(s < m ? e_dm[(int64_t)(f0 + i) * m + s] : 0.f);
    *reinterpret_cast<f16x8*>(tb + (int64_t)p * 16) = hv;
  }
  float* cn = reinterpret_cast<float*>(tb + (int64_t)nstep * 1024);
  if (tid < 32) {
    const int s = tile * 32 + tid;
    float nrm = 0.f;
    if (s < m)
      for (int f = 0; f < d; ++f) {
        const float v = (float)(_Float16)e_dm[(int64_t)f * m + s];
        nrm += v * v;
      }
    cn[tid] = s < m ? -0.5f * nrm : -3.0e38f;       // (finite: a packed key must never be a NaN; it never beats a real slot)
  } else {
    cn[tid] = 0.f;
  }
}

template <int K, int NSTEP, bool Q1>
int launch_f16r(const F16rArgs& a0, hipStream_t stream) {
  constexpr int RT = 3;
  F16rArgs a = a0;
  int dev = 0, cus = 256;
  if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
  if (cus <= 0) cus = 256;
  // one workgroup per CU at most; every wave gets the same number of row tiles
  const int waves = std::min(cus * RW, a.t32);
  a.pw = (a.t32 + waves - 1) / waves;
  const int grid = (a.t32 + a.pw * RW - 1) / (a.pw * RW);
  size_t lds = (size_t)NBUF * (NSTEP + 1) * 1024;
  auto kern = memory_topk_f16r_kernel<K, NSTEP, RT, Q1>;
  if constexpr (K == 2 && NSTEP == 32 && Q1) {          // measurement builds (wrong results): AMMC_F16R_DBG = 1 | 2 | 4 | 8 | 15
    static const int dbg = getenv("AMMC_F16R_DBG") ? atoi(getenv("AMMC_F16R_DBG")) : 0;
    if (dbg == 1) kern = memory_topk_f16r_kernel<K, NSTEP, RT, Q1, 1>;
    else if (dbg == 2) kern = memory_topk_f16r_kernel<K, NSTEP, RT, Q1, 2>;
    else if (dbg == 4) kern = memory_topk_f16r_kernel<K, NSTEP, RT, Q1, 4>;
    else if (dbg == 8) kern = memory_topk_f16r_kernel<K, NSTEP, RT, Q1, 8>;
    else if (dbg == 5) kern = memory_topk_f16r_kernel<K, NSTEP, RT, Q1, 5>;
    else if (dbg == 7) kern = memory_topk_f16r_kernel<K, NSTEP, RT, Q1, 7>;
    else if (dbg == 15) kern = memory_topk_f16r_kernel<K, NSTEP, RT, Q1, 15>;
  }
  if constexpr (K == 2 && NSTEP == 32) {
    // AMMC_F16R_FORM: 2 (default for the model's K = 2 at d = 512) = two row tiles per wave, all fragments in AGPRs, two
    // accumulator sets, the top-2 update of a tile branch-free on packed keys behind the next tile's MFMAs (2.15-2.2 ms
    // per 262144 rows); 3 = three row tiles per wave, one accumulator set, compare-and-branch update (2.35-2.4 ms; what
    // every other K / d runs)
    static const int form = getenv("AMMC_F16R_FORM") ? atoi(getenv("AMMC_F16R_FORM")) : 2;
    if (form == 2) {
      lds += (size_t)RW * 2 * 3072;                      // two 3-KB write-out slots per wave behind the ring
      kern = memory_topk_f16r_kernel<K, NSTEP, 2, Q1, 0, true>;
      if constexpr (Q1) {
        static const int dbg2 = getenv("AMMC_F16R_DBG") ? atoi(getenv("AMMC_F16R_DBG")) : 0;
        if (dbg2 == 1) kern = memory_topk_f16r_kernel<K, NSTEP, 2, Q1, 1, true>;
        else if (dbg2 == 4) kern = memory_topk_f16r_kernel<K, NSTEP, 2, Q1, 4, true>;
        else if (dbg2 == 5) kern = memory_topk_f16r_kernel<K, NSTEP, 2, Q1, 5, true>;
        else if (dbg2 == 15) kern = memory_topk_f16r_kernel<K, NSTEP, 2, Q1, 15, true>;
      }
    }
  }
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) return (int)e;
  hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, stream, a);
  return ammc_launch_status();
}

template <int K, bool Q1>
int launch_f16r_d(const F16rArgs& a, hipStream_t stream) {
  switch (a.d) {
    case 128: return launch_f16r<K, 8, Q1>(a, stream);
    case 256: return launch_f16r<K, 16, Q1>(a, stream);
    case 384: return launch_f16r<K, 24, Q1>(a, stream);
    case 512: return launch_f16r<K, 32, Q1>(a, stream);
  }
  return AMMC_EUNSUP;
}

}  // namespace ammc_f16r
using namespace ammc_f16r;

extern "C" int64_t ammc_codebook_f16_tiles_bytes(int32_t d, int32_t m) {
  if (d <= 0 || (d % 16) || m <= 0) return 0;
  return (int64_t)((m + 31) / 32) * ((d >> 4) + 1) * 1024;
}

extern "C" int ammc_pack_codebook_f16_tiles(const float* embed_dm, int32_t d, int32_t m, void* tiles, void* stream) {
  if (!embed_dm || !tiles || d <= 0 || (d % 16) || m <= 0) return AMMC_EINVAL;
  if ((uintptr_t)tiles & 15) return AMMC_EINVAL;
  hipLaunchKernelGGL(pack_codebook_f16_tiles_kernel, dim3((m + 31) / 32), dim3(256), 0, (hipStream_t)stream, embed_dm, d, m,
                     reinterpret_cast<unsigned char*>(tiles));
  return ammc_launch_status();
}

extern "C" int ammc_memory_topk_f16r_blocks(int32_t n) { return n <= 0 ? 0 : (n + 31) / 32; }

extern "C" int ammc_memory_topk_fwd_f16r(const float* x, const void* tiles, const float* embed_md, int32_t n, int32_t d,
                                         int32_t m, int32_t k, int32_t* idx_topk, float* q_topk, float* q_one,
                                         float* diff_partial, void* stream) {
  if (!x || !tiles || !embed_md || !idx_topk || !q_topk || !diff_partial) return AMMC_EINVAL;
  if (n <= 0 || m <= 0 || k <= 0 || k > m) return AMMC_EINVAL;
  if (d < 128 || (d % 128) || d > 512) return AMMC_EUNSUP;
  if (k > 4) return AMMC_EUNSUP;
  if (((uintptr_t)x | (uintptr_t)tiles | (uintptr_t)embed_md | (uintptr_t)q_topk | (uintptr_t)q_one) & 15) return AMMC_EINVAL;
  F16rArgs a;
  a.x = x, a.tiles = reinterpret_cast<const unsigned char*>(tiles), a.e_md = embed_md, a.idx_out = idx_topk;
  a.q_topk = q_topk, a.q_one = q_one, a.diff_partial = diff_partial;
  a.n = n, a.d = d, a.m = m, a.ntile = (m + 31) / 32, a.t32 = (n + 31) / 32, a.pw = 0;
  static const int inline_tail = getenv("AMMC_F16R_TAIL_INLINE") ? atoi(getenv("AMMC_F16R_TAIL_INLINE")) : 1;
  a.no_inline_tail = inline_tail == 0;
  hipStream_t s = (hipStream_t)stream;
#define F16R_K(K_)                                                                                   \
  case K_: return q_one ? launch_f16r_d<K_, true>(a, s) : launch_f16r_d<K_, false>(a, s);
  switch (k) {
    F16R_K(1) F16R_K(2) F16R_K(3) F16R_K(4)
  }
#undef F16R_K
  return AMMC_EUNSUP;
}
