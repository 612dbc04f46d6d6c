// 3x3 weight gradient, S16 operands, halo patch in LDS (see wgrad_tap_s16.hip), with THREE MFMAs per product block.
//
// wgrad_tap_s16.hip feeds an MFMA 16 channels x {hi, lo} per operand: one instruction yields the four plane products
// of a 16 x 16 channel block, of which lo*lo is wasted, and an accumulator tile covers only 16 x 16 outputs - nine taps
// x 144 registers hold 16 x 16 x 9 outputs per wave, 20 transposed LDS reads feed 9 MFMAs.
//
// Here an MFMA row is ONE plane of a channel and the planes are concatenated along K (K = pixels; the 16 K slots of
// an MFMA are two lane halves of 8):
//     acc += (GH[0-7]  | GH[8-15])  * (AH[0-7]  | AH[8-15])
//          + (GL'[0-7] | GH'[0-7])  * (AH[0-7]  | AL[0-7])             (X' = X * 2^-11, applied to the G fragment)
//          + (GH'[8-15] | GL'[8-15]) * (AL[8-15] | AH[8-15])
// = hi*hi + 2^-11 (hi*lo + lo*hi) over 16 pixels of a 32 x 32 channel block in 3 MFMAs (4 before), into ONE fp32 tile:
// a wave holds 32 x 32 x 9 outputs in the same 144 registers, a workgroup 128 gradient channels x 64 input channels.
// The A fragment of the hi*hi MFMA is a lane select of the two cross fragments (their hi halves are exactly AH[0-7] on
// the lower lanes and AH[8-15] on the upper ones - the K order of an MFMA is free as long as both operands agree), so
// a tap costs 4 transposed reads and a k-step of 16 pixels 6 + 36 reads for 27 MFMAs.  The pre-scaled G halves are
// exact down to the half-precision subnormals, i.e. an absolute 2^-25 of the operand scale (the argument of the
// single-accumulator forward kernel, conv_tap_s16.hip).
//
// ds_read_b64_tr_b16 takes a per-lane address: lane 4q+p of a 16-lane group supplies 8 bytes of pixel row q and gets
// back the four pixels of "column" l16.  Pointing p = 0,1 at the 16 hi (or lo) bytes of one S16 group and p = 2,3 at
// those of the next gives 16 channels of a single plane per lane group, 32 per wave.
//
// LDS: two stages of G [patch px][TN ch] and A [halo px][TC ch] (padded to whole DMA rounds).
// The 16-byte slot index s = 2 * group + plane of pixel row m is stored XOR-swizzled by m & 3 (w3_swz): a transposed
// read touches four consecutive rows x four groups of one plane, which the swizzle spreads over all sixteen 16-byte
// bank slots for any start row - conflict free for every tap.
// One workgroup of 8 waves per CU; split over patches; the partial tiles go out as slabs (round 5, WgradTap3Args::slabs)
// or as fp32 atomics into the packed gradient.
// Variants <NG, NA, NP, PH> (32-channel blocks of the gradient / of the input, row groups, patch rows):
//   <4, 2, 1, 2> 128 x 64 channel tiles (N % 128 == 0)          <2, 2, 2, 2> 64 x 64, the two patch rows on different waves
//   <1, 2, 4, 4> 32 gradient channels (the output layer)        <2, 1, 4, 4> Cin <= 32 (first layers, zero-padded block)
// Needs W % 32 == 0, H % 2 == 0 (% 4 for the 4-row patches); wgrad_tap_s16_try falls back otherwise.
//
// ROLL (round 5, the two-row patches): a workgroup walks DOWN a 32-pixel column of the image and keeps the input halo
// rows in a ring of four two-row groups - patch ty reads rows 2 ty - 1 .. 2 ty + 2 = groups ty and ty + 1, and while it is
// contracted only group ty + 2 (two NEW rows) is fetched, not the four rows of the next patch's halo: 33 KB instead of
// 51 KB of L2 -> LDS DMA per 64 x 64-channel patch, 49 instead of 67 KB per 128 x 64 one (PMC: 2.06 -> 1.38 GB and 509 ->
// 419 MB per launch).  Worth 1.5-4 % per layer: the kernel runs at the power cap and those bytes are that share of its
// energy - it was not waiting for them.  A run of patches that crosses into the next column loads that column's first two
// groups into the two free slots of the ring.
//
// G11 (round 5, opt-in: AMMC_WGRAD_G11=1): the gradient operand with its hi half only, (GH | GH) * (AH | AH) +
// (GH'[8-15] | GH'[0-7]) * (AL[8-15] | AL[0-7]) - the second product's fragments are lane selects of the cross fragments
// that are read anyway - two MFMAs per 16 pixels instead of three; see wgrad_tap3_s16_try for what it costs in accuracy
// (nothing measurable at the timed batch, 1-2e-4 on an isolated layer) and buys (15-19 % per layer).
//
// Measured and removed again (round 5, profiles/r05_wgrad_af_ab.txt; the code is in the history): the three taps of a
// filter row sharing SIX transposed reads (tap 1 = a 16-bit funnel shift of tap 0's registers, tap 2 = the same
// registers one further) instead of twelve - bit-identical results, 43 % fewer LDS read instructions, the same time
// within 1 % on every layer shape: the LDS read port is not what the kernel waits for.  Its matrix pipe is 0.65 busy in
// CYCLES (PMC) at a clock the power cap holds near 1.7 GHz; the rest is the per-patch `vmcnt(0)` + barrier and the
// epilogue (AMMC_WGRAD_DBG ablations).
#include "ammc_common.h"
#include <hip/hip_fp16.h>
#include <stdlib.h>

namespace ammc_s16 {

typedef _Float16 f16x8u __attribute__((ext_vector_type(8)));
typedef unsigned u32x2u __attribute__((ext_vector_type(2)));
typedef unsigned u32x4u __attribute__((ext_vector_type(4)));

struct WgradTap3Args {
  AmmcWgradDesc d;
  const float* g_inv_scale;
  int kpad, tiles_x, tiles_y, npatch, patches_per_block, msplit, col_tiles;
  // round 5: with `slabs` every workgroup STORES its tile of the packed gradient into the slab of its patch split
  // ([msplit x row groups][n][kpad], plain stores) and reduce_unpack_wgrad_kernel sums the slabs straight into the OIHW parameter
  // gradient - instead of fp32 atomics of all `msplit` tiles into one packed buffer (up to 75 MB of atomic traffic per
  // launch at the memory side's ~1.3 TB/s for a <= 9.4-MB result, a memset of that buffer and an unpack launch)
  float* slabs;
  int query;                       // 1: launch nothing, return msplit * n * kpad (floats of slabs the launch would need)
  int dbg;                         // AMMC_WGRAD_DBG (timing ablations, wrong results): 1 = no DMA inside the patch loop, 2 = no contraction, 4 = no epilogue
};

constexpr int W3_PW = 32, W3_HW = W3_PW + 2;                           // patch width, halo width
// slot swizzle by pixel row m & 3: rows of 256 / 512 bytes are bank aligned (16 slots of 16 B = all banks) - toggle the
// plane bit for odd rows and the next 128 B for rows 2, 3; rows of 128 bytes alternate bank halves by themselves, so
// only rows two apart have to part: toggle the plane bit by bit 1 of the row
template <int ROWB>
__device__ __forceinline__ int w3_swz(int m) { return ROWB >= 256 ? ((m & 1) | ((m & 2) << 2)) : ((m >> 1) & 1); }

template <int OFF>
__device__ __forceinline__ u32x2u w3_read_tr16(uint32_t addr) {
  u32x2u v;
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF) : "memory");
  return v;
}

__device__ __forceinline__ f16x8u w3_frag(u32x2u a, u32x2u b) {
  u32x4u v;
  v[0] = a[0]; v[1] = a[1]; v[2] = b[0]; v[3] = b[1];
  return __builtin_bit_cast(f16x8u, v);
}

// NG x NA x NP = 8 waves: NG 32-channel blocks of the gradient, NA of the input, NP groups of image rows of the PH-row
// patch (waves of different groups add their parts to the same outputs).  PH = 4 with one row per wave serves the thin
// operands: 32 gradient channels (the output layer) or 16 / 8 input channels (the first layers; channels beyond Cin
// are read from the caller's zero buffer).
// GW = 2 (round 4, four waves, ONE per SIMD with the whole 512-register file): a wave owns TWO 32-channel blocks of the
// gradient, so a transposed A fragment - four of the 4.7 LDS reads a step costs - feeds six MFMAs instead of three:
// 0.9 reads per MFMA instead of 1.4 (the kernel runs at 0.52 of the matrix pipe with its LDS reads at ~80 % of the MFMA
// time).  288 accumulator registers per lane; the same 128 x 64 channel workgroup tile as <4, 2, 1, 2>.
template <int NG, int NA, int NP, int PH, int GW = 1, int ROLL = 0, int G11 = 0>
__global__ __launch_bounds__(64 * NG * NA * NP, (GW == 2 ? 1 : 2)) void wgrad_tap3_s16_kernel(WgradTap3Args a) {
  static_assert((NG * NA * NP == 8 || (GW == 2 && NG * NA * NP == 4)) && (PH == 2 || PH == 4) && PH % NP == 0, "8 (4) waves");
  static_assert(!ROLL || PH == 2, "the ring holds two-row groups");
  constexpr int W3_NT = 64 * NG * NA * NP;
  constexpr int W3_PH = PH, W3_PX = PH * W3_PW, W3_HPX = (PH + 2) * W3_HW, RPW = PH / NP;
  constexpr int W3_TN = 32 * NG * GW, W3_TC = 32 * NA;
  constexpr int W3_GRB = W3_TN * 4, W3_ARB = W3_TC * 4;                  // row bytes
  constexpr int W3_GSLOTS = W3_TN / 4, W3_ASLOTS = W3_TC / 4;            // 16-byte slots per row
  constexpr int W3_GJ = W3_PX * W3_GSLOTS / W3_NT;                       // DMA rounds
  constexpr int W3_AJ = (W3_HPX * W3_ASLOTS + W3_NT - 1) / W3_NT;        // (the last one partly padding)
  constexpr int W3_GSTAGE = W3_PX * W3_TN;                               // floats
  constexpr int W3_ASTAGE = W3_AJ * W3_NT * 4;                           // floats
  // ROLL: a group = two halo rows = 68 pixels x TC channels, whole 64-piece wave instructions (68 ASLOTS % 64 == 0)
  constexpr int W3_RPIECES = 2 * W3_HW * W3_ASLOTS;
  constexpr int W3_RJ = (W3_RPIECES + W3_NT - 1) / W3_NT;                // DMA rounds of a group (the last: the first waves only)
  constexpr int W3_AGRP = W3_RPIECES * 4;                                // floats
  static_assert(!ROLL || W3_RPIECES % 64 == 0, "a group is whole wave instructions");
  static_assert(W3_PX * W3_GSLOTS % W3_NT == 0, "G pieces");
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* Gs = smem;                                             // [2][64 px][TN]
  float* As = smem + 2 * W3_GSTAGE;                             // [2][136 px (+pad)][TC]

  const AmmcWgradDesc& d = a.d;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int h = lane >> 5, l31 = lane & 31;
  const int wg = wave % NG, wa = (wave / NG) % NA, wp = wave / (NG * NA);   // this wave's blocks of G, of A, its row

  int bid = blockIdx.x;
  const int ms = bid % a.msplit;
  bid /= a.msplit;
  const int c0 = (bid % a.col_tiles) * W3_TC;                   // first input channel of the tile
  const int row0 = (bid / a.col_tiles) * W3_TN;                 // first gradient channel
  const int p_begin = ms * a.patches_per_block;
  const int p_end = min(p_begin + a.patches_per_block, a.npatch);
  if (p_begin >= p_end) return;

  // DMA pieces: piece p -> image row p / SLOTS, physical slot p % SLOTS, which holds logical slot ps ^ swz(row)
  int g_off[W3_GJ], a_off[W3_AJ];
  bool a_pad[W3_AJ];                                            // this piece lies beyond the layer's input channels
  int r_off[W3_RJ];                                             // ROLL: the pieces of a two-row group
#pragma unroll
  for (int j = 0; j < W3_GJ; ++j) {
    const int p = j * W3_NT + tid;
    const int px = p / W3_GSLOTS, ls = (p % W3_GSLOTS) ^ w3_swz<W3_GRB>(px & 3);
    g_off[j] = (int)((int64_t)(px >> 5) * d.g_rs + (int64_t)(px & 31) * d.g_ps) + row0 + 4 * ls;
  }
#pragma unroll
  for (int j = 0; j < W3_AJ; ++j) {
    int p = j * W3_NT + tid;
    p = p < W3_HPX * W3_ASLOTS ? p : W3_HPX * W3_ASLOTS - 1;
    const int hp = p / W3_ASLOTS, ls = (p % W3_ASLOTS) ^ w3_swz<W3_ARB>(hp & 3);
    const int hy = hp / W3_HW, hx = hp - hy * W3_HW;
    a_off[j] = (int)((int64_t)hy * d.a_rs + (int64_t)hx * d.a_ps) + c0 + 4 * ls;
    a_pad[j] = c0 + 4 * ls >= d.cin;                           // (slot = 4 elements: hi | lo halves of 8 channels per 2 slots)
  }

#pragma unroll
  for (int j = 0; j < W3_RJ; ++j) {
    int p = j * W3_NT + tid;
    p = p < W3_RPIECES ? p : W3_RPIECES - 1;
    const int hp = p / W3_ASLOTS, ls = (p % W3_ASLOTS) ^ w3_swz<W3_ARB>(hp & 3);     // (68 = 0 mod 4: the row's swizzle is the same in any group)
    const int hy = hp / W3_HW, hx = hp - hy * W3_HW;
    r_off[j] = (int)((int64_t)hy * d.a_rs + (int64_t)hx * d.a_ps) + c0 + 4 * ls;
  }

#define W3_ISSUE(patch, stage)                                                                            \
  {                                                                                                       \
    int sp_ = (patch);                                                                                    \
    const int tx_ = sp_ % a.tiles_x;                                                                      \
    sp_ /= a.tiles_x;                                                                                     \
    const int ty_ = sp_ % a.tiles_y, b_ = sp_ / a.tiles_y;                                                \
    const float* gp_ = d.g + ((int64_t)b_ * d.g_bs + (int64_t)(ty_ * W3_PH) * d.g_rs + (int64_t)(tx_ * W3_PW) * d.g_ps); \
    const float* ap_ = d.a + ((int64_t)b_ * d.a_bs + (int64_t)(ty_ * W3_PH) * d.a_rs + (int64_t)(tx_ * W3_PW) * d.a_ps); \
    float* gdst_ = Gs + (stage) * W3_GSTAGE + wave * 256;                                                 \
    float* adst_ = As + (stage) * W3_ASTAGE + wave * 256;                                                 \
    _Pragma("unroll") for (int j = 0; j < W3_GJ; ++j) {                                                   \
      const float* src_ = gp_ + g_off[j];                                                                 \
      __builtin_amdgcn_global_load_lds(src_, gdst_ + j * (W3_NT * 4), 16, 0, 0);                          \
    }                                                                                                     \
    _Pragma("unroll") for (int j = 0; j < W3_AJ; ++j) {                                                   \
      const float* src_ = a_pad[j] ? d.zeros : ap_ + a_off[j];                                            \
      __builtin_amdgcn_global_load_lds(src_, adst_ + j * (W3_NT * 4), 16, 0, 0);                          \
    }                                                                                                     \
  }

  // ROLL: patches are numbered ty fastest (down a column), then tx, then the image
#define W3_DECOMP(patch, b_, tx_, ty_)                                                                    \
  int b_, tx_, ty_;                                                                                       \
  {                                                                                                       \
    int sp_ = (patch);                                                                                    \
    ty_ = sp_ % a.tiles_y;                                                                                \
    sp_ /= a.tiles_y;                                                                                     \
    tx_ = sp_ % a.tiles_x, b_ = sp_ / a.tiles_x;                                                          \
  }
#define W3_ISSUE_G(b_, tx_, ty_, stage)                                                                   \
  {                                                                                                       \
    const float* gp_ = d.g + ((int64_t)(b_) * d.g_bs + (int64_t)((ty_) * W3_PH) * d.g_rs + (int64_t)((tx_) * W3_PW) * d.g_ps); \
    float* gdst_ = Gs + (stage) * W3_GSTAGE + wave * 256;                                                 \
    _Pragma("unroll") for (int j = 0; j < W3_GJ; ++j) {                                                   \
      const float* src_ = gp_ + g_off[j];                                                                 \
      __builtin_amdgcn_global_load_lds(src_, gdst_ + j * (W3_NT * 4), 16, 0, 0);                          \
    }                                                                                                     \
  }
  // group `grp` of a column = halo rows 2 grp, 2 grp + 1 (image rows 2 grp - 1, 2 grp); the last round: the waves it has pieces for
#define W3_ISSUE_GRP(b_, tx_, grp_, slot_)                                                                \
  {                                                                                                       \
    const float* ap_ = d.a + ((int64_t)(b_) * d.a_bs + (int64_t)((grp_) * 2) * d.a_rs + (int64_t)((tx_) * W3_PW) * d.a_ps); \
    float* adst_ = As + (slot_) * W3_AGRP + wave * 256;                                                   \
    _Pragma("unroll") for (int j = 0; j < W3_RJ; ++j)                                                     \
      if (j * W3_NT + wave * 64 < W3_RPIECES) {                                                           \
        const float* src_ = ap_ + r_off[j];                                                               \
        __builtin_amdgcn_global_load_lds(src_, adst_ + j * (W3_NT * 4), 16, 0, 0);                        \
      }                                                                                                   \
  }

  f32x16 acc[GW][9];
#pragma unroll
  for (int b = 0; b < GW; ++b)
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[b][t][r] = 0.f;

  // transposed-read lane roles: lane 4q+p of a 16-lane group supplies pixel row q; p >> 1 picks the S16 group of the
  // pair, p & 1 the 8-byte half of its 16 plane bytes
  const int l16 = lane & 15, q = l16 >> 2, p = l16 & 3, gi = l31 >> 4;
  const int sg = 8 * (wg * GW) + 4 * gi + 2 * (p >> 1);         // logical slot of this lane's hi bytes in a G row (block 0 of the wave)
  const int sa = 8 * wa + 4 * gi + 2 * (p >> 1);                // ... in an A row
  // G pixel rows of a read are y*32 + x16 + 4j + q: row & 3 == q, one swizzle per lane.  Fragments per lane half h:
  //   G_hi = GH[8h .. 8h+7]                                   A_x1 = (h ? AL : AH)[0-7]      A_x2 = (h ? AH : AL)[8-15]
  //   G_x1 = 2^-11 (h ? GH : GL)[0-7]                         A_hi = h ? A_x2 : A_x1 = AH[8h .. 8h+7]   (a lane select)
  //   G_x2 = 2^-11 (h ? GL : GH)[8-15]
  // K order is free as long as both operands agree: the hi*hi MFMA takes its A fragment from the halves of the two
  // cross fragments that hold hi, so a tap costs FOUR transposed reads, not six.  The 2^-11 of the cross terms goes on
  // the whole G cross fragments (hi and lo halves alike), once per k-step.
  uint32_t g_lane_hi[GW], g_lane_x1[GW], g_lane_x2[GW];       // (per block: the swizzle XORs a bit that adding 8 slots carries into)
#pragma unroll
  for (int b = 0; b < GW; ++b) {
    const int sb = sg + 8 * b;
    g_lane_hi[b] = (uint32_t)(((sb + 0) ^ w3_swz<W3_GRB>(q)) * 16 + (p & 1) * 8 + q * W3_GRB + h * (8 * W3_GRB));
    g_lane_x1[b] = (uint32_t)(((sb + (1 - h)) ^ w3_swz<W3_GRB>(q)) * 16 + (p & 1) * 8 + q * W3_GRB);
    g_lane_x2[b] = (uint32_t)(((sb + h) ^ w3_swz<W3_GRB>(q)) * 16 + (p & 1) * 8 + q * W3_GRB);
  }
  // A rows start anywhere: (row & 3) = (q + 2 (y + r) + s) & 3 for tap (r, s); the four possible swizzled slot offsets,
  // rotated by q, so that the index below is a compile-time constant
  uint32_t a_sw_x1[4], a_sw_x2[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    a_sw_x1[k] = (uint32_t)(((sa + h) ^ w3_swz<W3_ARB>((q + k) & 3)) * 16);
    a_sw_x2[k] = (uint32_t)(((sa + (1 - h)) ^ w3_swz<W3_ARB>((q + k) & 3)) * 16);
  }
  const uint32_t a_lane = (uint32_t)((p & 1) * 8 + q * W3_ARB);
  const _Float16 cg = (_Float16)(1.f / 2048.f);
  const bool upper = h != 0;
  const uint32_t g_base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const void*)Gs;
  const uint32_t a_base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const void*)As;

  // A wave's RPW image rows of a patch are ONE flat sequence of NU = 18 RPW steps (step U: row Y0 + U / 18, x half
  // (U / 9) & 1, tap U % 9; all literals): 4 transposed reads + 3 MFMAs per step, the A reads of steps U + 1 and U + 2 in
  // flight during the MFMAs of step U (three register sets) ACROSS half-row boundaries, and the six G reads of the next
  // half-row issued at tap 5 of the current one into a second raw set - a half-row no longer starts cold (14 reads, then
  // a full LDS round trip, four times per patch).  Waits are counted: LDS reads return in order, so step U needs
  // everything issued up to A(U) and tolerates what was issued after it.  Issue order around a boundary:
  //   ... A(t7) [step t5, before its wait], G(next) [step t5, after it], A(t8) [t6], A(next t0) [t7], A(next t1) [t8] ...
  u32x2u ar[3][4], graw[GW][2][6];
  f16x8u gf_hi[GW], gf_x1[GW], gf_x2[GW];
#define W3_AREAD(Y0, U, S)                                                                                    \
  {                                                                                                           \
    constexpr int y_ = (Y0) + (U) / 18, xh_ = ((U) / 9) & 1, t_ = (U) % 9;                                    \
    constexpr int hr_ = y_ + t_ / 3;                          /* halo row; ROLL: rows 2, 3 live in the next group */ \
    constexpr int off_ = ((ROLL ? (hr_ & 1) : hr_) * W3_HW + t_ % 3 + 16 * xh_) * W3_ARB;                     \
    constexpr int k_ = (2 * hr_ + t_ % 3) & 3;                                                                \
    static_assert(off_ + 12 * W3_ARB < 65536, "ds offset");                                                   \
    const uint32_t ab_ = (ROLL && hr_ >= 2) ? abase_hi : abase;                                               \
    const uint32_t a1_ = ab_ + a_sw_x1[k_], a2_ = ab_ + a_sw_x2[k_];                                          \
    ar[S][0] = w3_read_tr16<off_>(a1_);                                                                       \
    ar[S][1] = w3_read_tr16<off_ + 4 * W3_ARB>(a1_);                                                          \
    ar[S][2] = w3_read_tr16<off_ + 8 * W3_ARB>(a2_);                                                          \
    ar[S][3] = w3_read_tr16<off_ + 12 * W3_ARB>(a2_);                                                         \
  }
#define W3_GREAD(Y0, HH, S)                                                                                   \
  {                                                                                                           \
    constexpr int gp_ = (((Y0) + (HH) / 2) * 32 + 16 * ((HH) & 1)) * W3_GRB;                                  \
    static_assert(gp_ + 12 * W3_GRB < 65536, "ds offset");                                                    \
    _Pragma("unroll") for (int b_ = 0; b_ < GW; ++b_) {                                                       \
      graw[b_][S][0] = w3_read_tr16<gp_>(ghi[b_]);                                                            \
      graw[b_][S][1] = w3_read_tr16<gp_ + 4 * W3_GRB>(ghi[b_]);                                               \
      graw[b_][S][2] = w3_read_tr16<gp_>(g1[b_]);                                                             \
      graw[b_][S][3] = w3_read_tr16<gp_ + 4 * W3_GRB>(g1[b_]);                                                \
      graw[b_][S][4] = w3_read_tr16<gp_ + 8 * W3_GRB>(g2[b_]);                                                \
      graw[b_][S][5] = w3_read_tr16<gp_ + 12 * W3_GRB>(g2[b_]);                                               \
    }                                                                                                         \
  }
#define W3_STEP(Y0, U)                                                                                        \
  {                                                                                                           \
    constexpr int t_ = (U) % 9, hh_ = (U) / 9;                                                                \
    constexpr bool nxt_ = hh_ + 1 < NH;                                                                       \
    if ((U) + 2 < NU) W3_AREAD(Y0, ((U) + 2 < NU ? (U) + 2 : 0), (((U) + 2) % 3))                             \
    constexpr int cnt_ = ((U) + 1 < NU ? 4 : 0) + ((U) + 2 < NU ? 4 : 0) + (((t_ == 6 || t_ == 7) && nxt_) ? 6 * GW : 0); \
    __builtin_amdgcn_s_waitcnt(0xC07F | ((cnt_ < 15 ? cnt_ : 15) << 8));    /* (lgkmcnt is a 4-bit field: 15 = the most it can tolerate) */ \
    __builtin_amdgcn_sched_barrier(0);                                                                        \
    if (t_ == 0) {                                                                                            \
      _Pragma("unroll") for (int b_ = 0; b_ < GW; ++b_) {                                                     \
        gf_hi[b_] = w3_frag(graw[b_][hh_ & 1][0], graw[b_][hh_ & 1][1]);                                      \
        gf_x1[b_] = w3_frag(graw[b_][hh_ & 1][2], graw[b_][hh_ & 1][3]) * cg;                                 \
        gf_x2[b_] = w3_frag(graw[b_][hh_ & 1][4], graw[b_][hh_ & 1][5]) * cg;                                 \
        /* G11: the hi halves of the two cross fragments, GH'[8-15] on the lower lanes and GH'[0-7] on the upper ones */ \
        if (G11) gf_x1[b_] = upper ? gf_x1[b_] : gf_x2[b_];                                                   \
      }                                                                                                       \
    }                                                                                                         \
    if (t_ == 5 && nxt_) W3_GREAD(Y0, (nxt_ ? hh_ + 1 : 0), ((hh_ + 1) & 1))                                  \
    const f16x8u ax1_ = w3_frag(ar[(U) % 3][0], ar[(U) % 3][1]), ax2_ = w3_frag(ar[(U) % 3][2], ar[(U) % 3][3]); \
    const f16x8u ahi_ = upper ? ax2_ : ax1_;                                                                  \
    _Pragma("unroll") for (int b_ = 0; b_ < GW; ++b_)                                                         \
      acc[b_][t_] = __builtin_amdgcn_mfma_f32_32x32x16_f16(gf_hi[b_], ahi_, acc[b_][t_], 0, 0, 0);            \
    if (G11) {            /* ONE cross MFMA: GH' x AL over the 16 pixels (AL[8-15] on the lower lanes, AL[0-7] on the upper) */ \
      const f16x8u alo_ = upper ? ax1_ : ax2_;                                                                \
      _Pragma("unroll") for (int b_ = 0; b_ < GW; ++b_)                                                       \
        acc[b_][t_] = __builtin_amdgcn_mfma_f32_32x32x16_f16(gf_x1[b_], alo_, acc[b_][t_], 0, 0, 0);          \
    } else {                                                                                                  \
      _Pragma("unroll") for (int b_ = 0; b_ < GW; ++b_)                                                       \
        acc[b_][t_] = __builtin_amdgcn_mfma_f32_32x32x16_f16(gf_x1[b_], ax1_, acc[b_][t_], 0, 0, 0);          \
      _Pragma("unroll") for (int b_ = 0; b_ < GW; ++b_)                                                       \
        acc[b_][t_] = __builtin_amdgcn_mfma_f32_32x32x16_f16(gf_x2[b_], ax2_, acc[b_][t_], 0, 0, 0);          \
    }                                                                                                         \
    __builtin_amdgcn_sched_barrier(0);                                                                        \
  }
#define W3_STEP9(Y0, U) W3_STEP(Y0, U) W3_STEP(Y0, (U) + 1) W3_STEP(Y0, (U) + 2) W3_STEP(Y0, (U) + 3) W3_STEP(Y0, (U) + 4) \
  W3_STEP(Y0, (U) + 5) W3_STEP(Y0, (U) + 6) W3_STEP(Y0, (U) + 7) W3_STEP(Y0, (U) + 8)
#define W3_SEQ(Y0)                                                                                            \
  {                                                                                                           \
    W3_GREAD(Y0, 0, 0)                                                                                        \
    W3_AREAD(Y0, 0, 0)                                                                                        \
    W3_AREAD(Y0, 1, 1)                                                                                        \
    W3_STEP9(Y0, 0) W3_STEP9(Y0, 9)                                                                           \
    if (RPW == 2) { W3_STEP9(Y0, (RPW == 2 ? 18 : 0)) W3_STEP9(Y0, (RPW == 2 ? 27 : 9)) }                     \
  }
  constexpr int NH = 2 * RPW, NU = 18 * RPW;
  static_assert(RPW == 1 || RPW == 2, "rows per wave");

  int ring = 0;                                                 // ROLL: the slot of the current patch's first group
  if (ROLL) {
    W3_DECOMP(p_begin, b0, tx0, ty0)
    W3_ISSUE_G(b0, tx0, ty0, 0)
    W3_ISSUE_GRP(b0, tx0, ty0, 0)
    W3_ISSUE_GRP(b0, tx0, ty0 + 1, 1)
  } else {
    W3_ISSUE(p_begin, 0);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  for (int pt = p_begin; pt < p_end; ++pt) {
    const int stage = (pt - p_begin) & 1;
    int adv = 1;
    if (pt + 1 < p_end && !(a.dbg & 1)) {
      if (ROLL) {
        W3_DECOMP(pt + 1, b1, tx1, ty1)
        W3_ISSUE_G(b1, tx1, ty1, stage ^ 1)
        if (ty1 != 0) {                                  // the same column: two new rows
          W3_ISSUE_GRP(b1, tx1, ty1 + 1, (ring + 2) & 3)
        } else {                                         // the next column starts: its first two groups, into the two free slots
          W3_ISSUE_GRP(b1, tx1, 0, (ring + 2) & 3)
          W3_ISSUE_GRP(b1, tx1, 1, (ring + 3) & 3)
          adv = 2;
        }
      } else {
        W3_ISSUE(pt + 1, stage ^ 1);
      }
    }
    const uint32_t gst = g_base + (uint32_t)(stage * W3_GSTAGE * 4);
    const uint32_t abase = a_base + (uint32_t)((ROLL ? ring * W3_AGRP : stage * W3_ASTAGE) * 4) + a_lane;
    const uint32_t abase_hi = a_base + (uint32_t)((((ring + 1) & 3) * W3_AGRP) * 4) + a_lane;
    uint32_t ghi[GW], g1[GW], g2[GW];
#pragma unroll
    for (int b = 0; b < GW; ++b) { ghi[b] = gst + g_lane_hi[b]; g1[b] = gst + g_lane_x1[b]; g2[b] = gst + g_lane_x2[b]; }
    if (!(a.dbg & 2)) {
      if (0 == wp) { W3_SEQ(0) }                         // (uniform per wave; the rows are literals in the offsets)
      if (NP >= 2 && 1 == wp) { W3_SEQ((NP >= 2 ? RPW : 0)) }
      if (NP == 4) {
        if (2 == wp) { W3_SEQ((NP == 4 ? 2 * RPW : 0)) }
        if (3 == wp) { W3_SEQ((NP == 4 ? 3 * RPW : 0)) }
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    ring = (ring + adv) & 3;
  }
#undef W3_ISSUE
#undef W3_ISSUE_G
#undef W3_ISSUE_GRP
#undef W3_DECOMP
#undef W3_AREAD
#undef W3_GREAD
#undef W3_STEP
#undef W3_STEP9
#undef W3_SEQ

  // ---- add to the packed gradient: row = gradient channel (registers), column = tap * Cin + c (lanes) -------------
  if (a.dbg & 4) return;
  const float inv = a.g_inv_scale ? a.g_inv_scale[0] : 1.f;
  const int c = c0 + 32 * wa + l31;
#pragma unroll
  for (int b = 0; b < GW; ++b)
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      const int col = t * d.cin + c;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = row0 + 32 * (wg * GW + b) + (r & 3) + 8 * (r >> 2) + 4 * h;
        if (row < d.n && c < d.cin) {
          // (waves of different row groups hold parts of the SAME outputs - the atomics used to add them: each group its own slab)
          if (a.slabs) a.slabs[((int64_t)(ms * NP + wp) * d.n + row) * a.kpad + col] = acc[b][t][r] * inv;
          else unsafeAtomicAdd(d.dw + (int64_t)row * a.kpad + col, acc[b][t][r] * inv);
        }
      }
    }
}

template <int NG, int NA, int NP, int PH, int GW = 1, int ROLL = 0, int G11 = 0>
static int launch_wgrad_tap3(WgradTap3Args a, hipStream_t stream) {
  constexpr int W3_NT = 64 * NG * NA * NP;
  constexpr int TN = 32 * NG * GW, TC = 32 * NA;
  constexpr int AJ = ((PH + 2) * W3_HW * (TC / 4) + W3_NT - 1) / W3_NT;
  constexpr size_t lds = (size_t)(2 * PH * W3_PW * TN + (ROLL ? 4 * 2 * W3_HW * TC : 2 * AJ * W3_NT * 4)) * sizeof(float);
  static_assert(lds <= 160 * 1024, "LDS budget");
  auto kern = wgrad_tap3_s16_kernel<NG, NA, NP, PH, GW, ROLL, G11>;
  if (!a.query) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                       (int)lds);
    if (e != hipSuccess) return (int)e;
  }
  a.col_tiles = (a.d.cin + TC - 1) / TC;
  a.tiles_y = a.d.height / PH;
  a.npatch = a.d.batch * a.tiles_x * a.tiles_y;
  const int tiles = (a.d.n / TN) * a.col_tiles;
  int msplit = (256 + tiles - 1) / tiles;                           // one workgroup per CU, at least 8 patches each
  const int max_split = (a.npatch + 7) / 8;
  if (msplit > max_split) msplit = max_split;
  if (msplit < 1) msplit = 1;
  a.patches_per_block = (a.npatch + msplit - 1) / msplit;
  a.msplit = (a.npatch + a.patches_per_block - 1) / a.patches_per_block;       // (every split owns at least one patch)
  if (a.query) return a.msplit * NP;                               // slabs: one per patch split and row group of the workgroup
  hipLaunchKernelGGL(kern, dim3(tiles * a.msplit), dim3(W3_NT), lds, stream, a);
  return ammc_launch_status();
}

// Called by wgrad_tap_s16_try; -12345 = not this kernel's case.
// `slabs` / `query`: see WgradTap3Args (query: the return value is the number of patch splits, or -12345)
int wgrad_tap3_s16_try(const AmmcWgradDesc& d, const float* g_inv_scale, int kpad, hipStream_t stream, float* slabs, int query) {
  if (d.height % 2 || d.width % W3_PW) return -12345;
  const int64_t gmax = (int64_t)3 * d.g_rs + (int64_t)(W3_PW - 1) * d.g_ps + d.n;
  const int64_t amax = (int64_t)5 * d.a_rs + (int64_t)(W3_PW + 1) * d.a_ps + d.cin;
  if (gmax >= (1LL << 30) || amax >= (1LL << 30)) return -12345;
  WgradTap3Args a;
  a.d = d;
  a.g_inv_scale = g_inv_scale;
  a.kpad = kpad;
  a.slabs = slabs, a.query = query;
  static const int dbg = getenv("AMMC_WGRAD_DBG") ? atoi(getenv("AMMC_WGRAD_DBG")) : 0;
  a.dbg = dbg;
  a.tiles_x = d.width / W3_PW;
  a.tiles_y = a.npatch = 0;                                       // set by the launcher (patch height)
  if (d.cin % 64 == 0) {
    // AMMC_WGRAD_GW: 2 = the four-wave form with two gradient blocks per wave (A/B; round 4)
    static const int gw = getenv("AMMC_WGRAD_GW") ? atoi(getenv("AMMC_WGRAD_GW")) : 1;
    // AMMC_WGRAD_ROLL=0: every patch fetches its own four halo rows (the form before round 5; A/B)
    static const int roll = getenv("AMMC_WGRAD_ROLL") ? atoi(getenv("AMMC_WGRAD_ROLL")) : 1;
    // AMMC_WGRAD_G11=1 (opt-in, round 5): the GRADIENT operand enters the product with its 11-bit hi half only - two MFMAs
    // per 16 pixels of a channel block (GH x AH + GH' x AL) instead of three: 15-19 % off every layer, 3.7 % off the step
    // (59.4-59.8 -> 57.4 ms).  A weight gradient is a leaf, the 2^-12 relative rounding of g does not propagate, and against
    // the fp64 truth of the TIMED batch every per-tensor error is the same to three digits both ways (max 4.40e-3 / p90
    // 3.61e-3 / median 1.39e-3; the reference's own fp32 gradients 3.75e-3 / 3.03e-3 / 1.08e-3: profiles/r05_wgrad_g11_ab.txt).
    // NOT the default: an isolated layer is then 1-2e-4 from fp64 instead of 3e-6, and the mask-free fp64 tests of the
    // training path (tests/test_gpu_train.py: 1e-4 per tensor) see 2.1e-4 - the arithmetic would no longer be
    // fp32-equivalent, whatever the timed batch can resolve.
    static const int g11 = getenv("AMMC_WGRAD_G11") ? atoi(getenv("AMMC_WGRAD_G11")) : 0;
    if (g11 && roll && d.n % 128 == 0) return launch_wgrad_tap3<4, 2, 1, 2, 1, 1, 1>(a, stream);
    if (g11 && roll && d.n % 64 == 0) return launch_wgrad_tap3<2, 2, 2, 2, 1, 1, 1>(a, stream);
    if (d.n % 128 == 0 && gw == 2) return launch_wgrad_tap3<2, 2, 1, 2, 2>(a, stream);
    if (d.n % 128 == 0) return roll ? launch_wgrad_tap3<4, 2, 1, 2, 1, 1>(a, stream) : launch_wgrad_tap3<4, 2, 1, 2>(a, stream);
    if (d.n % 64 == 0) return roll ? launch_wgrad_tap3<2, 2, 2, 2, 1, 1>(a, stream) : launch_wgrad_tap3<2, 2, 2, 2>(a, stream);
    if (d.n == 32 && d.height % 4 == 0) return launch_wgrad_tap3<1, 2, 4, 4>(a, stream);    // the output layer
    return -12345;
  }
  // first layers: 8 / 16 / 32 input channels (one zero-padded 32-channel block)
  if (d.cin <= 32 && d.n % 64 == 0 && d.height % 4 == 0) return launch_wgrad_tap3<2, 1, 4, 4>(a, stream);
  return -12345;
}

// slabs [msplit][n][kpad] (k = tap * cin_p + c) -> OIHW [cout][cin][3][3]: 64 consecutive elements of the packed row x
// RU_G slab groups per workgroup; thread (e, g) sums the slabs m = g, g + RU_G, ... of element e with eight loads in
// flight, the groups are added through LDS in a fixed order (deterministic, unlike the atomics it replaces).
// (One thread per element walking all slabs - the first form - left the 64 -> 64 layers with 144 workgroups of serial
// 512-deep chains: 75 MB in 31 us.)
constexpr int RU_G = 4;
__global__ __launch_bounds__(64 * RU_G) void reduce_unpack_wgrad_kernel(const float* __restrict__ slabs, int msplit, int n, int kpad,
                                                                       int cout, int cin, int cin_p, float* __restrict__ out) {
  __shared__ float part[RU_G][64];
  const int64_t total = (int64_t)cout * cin * 9;
  const int e = threadIdx.x & 63, g = threadIdx.x >> 6;
  const int64_t gid = (int64_t)blockIdx.x * 64 + e;
  const bool live = gid < total;
  // threads of a wave walk the PACKED row (coalesced reads of every slab); the OIHW element is computed from it
  const int64_t gi = live ? gid : 0;
  const int o = (int)(gi / ((int64_t)cin * 9));
  const int k = (int)(gi - (int64_t)o * cin * 9);               // k = tap * cin + c over the TRUE channels
  const int tap = k / cin, c = k - tap * cin;
  const int64_t stride = (int64_t)n * kpad;
  const float* p = slabs + (int64_t)o * kpad + tap * cin_p + c + g * stride;
  const int64_t step = RU_G * stride;
  float s[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  int m = g;
  for (; m + 7 * RU_G < msplit; m += 8 * RU_G) {
#pragma unroll
    for (int u = 0; u < 8; ++u) s[u] += p[(int64_t)u * step];
    p += 8 * step;
  }
  for (; m < msplit; m += RU_G) {
    s[0] += *p;
    p += step;
  }
  part[g][e] = ((s[0] + s[1]) + (s[2] + s[3])) + ((s[4] + s[5]) + (s[6] + s[7]));
  __syncthreads();
  if (g == 0 && live) out[((int64_t)o * cin + c) * 9 + tap] = (part[0][e] + part[1][e]) + (part[2][e] + part[3][e]);
}

}  // namespace ammc_s16
using namespace ammc_s16;

extern "C" int64_t ammc_conv_wgrad_s16_slab_floats(const AmmcWgradDesc* desc) {
  if (!desc || desc->ntaps != 9 || desc->a_step > 1 || desc->n <= 0 || (desc->n % 32) || desc->cin < 8 ||
      (desc->cin & (desc->cin - 1)) || desc->batch <= 0 || desc->height <= 0 || desc->width <= 0)
    return 0;
  const int kpad = ((9 * desc->cin + 31) / 32) * 32;
  const int ms = wgrad_tap3_s16_try(*desc, nullptr, kpad, nullptr, nullptr, 1);
  return ms <= 0 ? 0 : (int64_t)ms * desc->n * kpad;
}

extern "C" int ammc_conv_wgrad_s16_slabs(const AmmcWgradDesc* desc, const float* g_inv_scale, float* slabs,
                                         int64_t slab_floats, float* dw_oihw, int32_t cout, int32_t cin, void* stream) {
  if (!desc || !desc->g || !desc->a || !desc->zeros || !slabs || !dw_oihw) return AMMC_EINVAL;
  const AmmcWgradDesc& d = *desc;
  if (cout <= 0 || cout > d.n || cin <= 0 || cin > d.cin) return AMMC_EINVAL;
  if (((uintptr_t)d.g | (uintptr_t)d.a | (uintptr_t)d.zeros) & 31) return AMMC_EINVAL;
  if ((d.g_bs | d.g_rs | d.g_ps | d.a_bs | d.a_rs | d.a_ps) & 7) return AMMC_EINVAL;
  const int64_t need = ammc_conv_wgrad_s16_slab_floats(desc);
  if (need <= 0) return AMMC_EUNSUP;
  if (slab_floats < need) return AMMC_EINVAL;
  const int kpad = ((9 * d.cin + 31) / 32) * 32;
  const int msplit = (int)(need / ((int64_t)d.n * kpad));
  const int rc = wgrad_tap3_s16_try(d, g_inv_scale, kpad, (hipStream_t)stream, slabs, 0);
  if (rc != AMMC_OK) return rc == -12345 ? AMMC_EUNSUP : rc;
  const int64_t total = (int64_t)cout * cin * 9;
  hipLaunchKernelGGL(reduce_unpack_wgrad_kernel, dim3((unsigned)((total + 63) / 64)), dim3(64 * RU_G), 0, (hipStream_t)stream,
                     slabs, msplit, d.n, kpad, cout, cin, d.cin, dw_oihw);
  return ammc_launch_status();
}
