// 3x3 convolution (S16 operands, see conv_gemm_s16.hip) with the input tile resident in LDS across the nine taps.
//
// Why: the implicit-GEMM kernel re-fetches the A operand once per tap (K order tap-major), and measurement says its
// time IS that traffic: with every MFMA removed it still takes 75-96 % of its time, at ~10-12 TB/s of L2 -> LDS DMA
// summed over the chip (`AMMC_S16_DBG=2`).  Here a workgroup owns a spatial patch of 8 x 32
// output pixels of one image and all (or 128) output channels.  For every block of 32 input channels the 10 x 34
// halo patch is DMA'd into LDS ONCE (43.5 KB instead of 9 x 32 KB); tap (r, s) is then just a shifted LDS address:
// an MFMA row tile is one image row of 32 pixels, whose A fragments for tap (r, s) are the 32 consecutive halo
// pixels starting at (y + r, s).  Only the filter slice (16 KB for 128 output channels) is streamed per tap.
// L2 -> LDS bytes per FLOP drop 2.3x (128 filters) to 3.7x (64 filters).
//
// LDS: halo patch, two stages (one per 32-channel block; the next block's patch arrives in six rounds spread over
// taps 0..5 of the current one), filter slices, three stages (two slices in flight, counted vmcnt waits).  Rows are 128 B (32 channels of one pixel / one filter);
// the 16-byte slots of row `p` are XOR-swizzled with (p >> 1) & 7 as in the GEMM kernels, which keeps both the
// lane-linear DMA writes and the ds_read_b128 fragment reads (32 consecutive rows, any start parity) conflict free.
// One workgroup (8 waves) per CU; accumulators: two fp32 sets (hi*hi, cross terms) per 32x32 tile.
//
// Serves: stride-1 3x3 convs with Cin % 32 == 0, W % 32 == 0, H % 8 == 0, N = 64 or a multiple of 128, S16 output
// (every double_conv layer of the network at 256x256 except the two first ones).  Everything else stays on
// conv_gemm_s16_kernel; ammc_conv_gemm_s16 dispatches.
#include "ammc_common.h"
#include <hip/hip_fp16.h>
#include <stdio.h>
#include <stdlib.h>

namespace ammc_s16 {

typedef _Float16 f16x8t __attribute__((ext_vector_type(8)));

// -DAMMC_TAP_STAMP (diagnostic builds only; tools/micro/tap_stamps.py): wave 0 of every tile records s_memrealtime
// (100 MHz) at the phase boundaries of its tile into a buffer that nothing else reads
#ifdef AMMC_TAP_STAMP
static unsigned long long* g_tap_stamps = nullptr;
extern "C" void ammc_debug_set_tap_stamps(void* p) { g_tap_stamps = (unsigned long long*)p; }
#define TAP_STAMP(slot) { if (a.stamps && threadIdx.x == 0 && (slot) < 16) a.stamps[(int64_t)vb * 16 + (slot)] = __builtin_amdgcn_s_memrealtime(); }
#else
#define TAP_STAMP(slot)
#endif

struct TapArgs {
  AmmcConvDesc d;
  int tiles_x, tiles_y, n_tiles, ncc, kpad, dbg;
  unsigned long long* stamps;       // diagnostic builds (AMMC_TAP_STAMP)
};

constexpr float T_LO_INV = 1.f / 2048.f;

// LDS-DMA of 16 bytes per lane, hidden from the compiler (KH loop): `lds_byte` = the wave-uniform LDS byte address the
// wave's 1 KB goes to, `off` = this lane's byte offset from the wave-uniform `base`.  Why not the builtin: hipcc books
// __builtin_amdgcn_global_load_lds as a FLAT access to both memories ("pending flat"), after which every wait it inserts
// for an LDS fragment read is lgkmcnt(0) until both counters have been drained - the reads of the NEXT chunk issued ahead
// of the MFMAs of this one would be waited for at once.  In asm the compiler sees neither the memory operation (so the
// vmcnt waits are written by hand, as asm as well: a builtin wait it believes redundant is dropped) nor M0, which is
// saved and restored in the statement (cdna_hip_programming.md section 5.7).
// the same with a full 64-bit address per lane (lanes of one instruction reading from unrelated allocations)
__device__ __forceinline__ void tap_dma16_ptr(const float* src, unsigned lds_byte) {
  unsigned keep;
  asm volatile("s_nop 4\n\ts_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(src), "s"(lds_byte) : "memory");
}
__device__ __forceinline__ void tap_dma16(const float* base, unsigned off, unsigned lds_byte) {
  unsigned keep;
  // (s_nop 4: `base` / `lds_byte` may have been written by the SALU instruction just before the statement, and a
  // vector-memory instruction must not read an SGPR sooner than five wait states after a scalar write - the compiler pads
  // its own instructions, never the inside of an asm statement; without it a launch now and then read a stale base)
  asm volatile("s_nop 4\n\ts_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(off), "s"(base), "s"(lds_byte) : "memory");
}
constexpr int T_TH = 8, T_TW = 32, T_HW = T_TW + 2, T_HP = (T_TH + 2) * T_HW;     // 340 halo pixels
constexpr int T_APIECES = T_HP * 8;                                              // 2720 16-byte pieces
// the patch stage is padded to whole rounds of the workgroup's threads so that EVERY wave issues every round (the
// counted vmcnt waits below need wave-uniform counts): 6 rounds of 512 threads, 11 of 256

// AS = halo-patch stages.  2: the next channel block's patch is prefetched during the current one (one workgroup per
// CU).  1: the patch is reloaded at every channel-block boundary and TWO workgroups share a CU (<= 80 KB of LDS,
// <= 128 VGPRs), so one's loads and epilogue overlap the other's MFMAs - for layers with little work per patch.
//
// MF = the MFMA shape.  0: v_mfma_f32_32x32x16_f16 (an MFMA row tile = one image row of 32 pixels, two k-steps per
// 32-channel block).  1: v_mfma_f32_16x16x32_f16 (16-pixel x 16-filter tiles, ONE MFMA per 32-channel block): the same
// LDS bytes, MFMA cycles and accumulator registers per wave tile, but the chip holds a higher clock on this shape
// under load (MI355X_MICROARCH.md, DVFS give-back item 7), and the 3-filter output layer wastes 13 of 16 MFMA rows
// instead of 29 of 32.  MF = 1 lays the 128-byte LDS rows out differently: lane l of a fragment read takes row l & 15,
// S16 group l >> 4, so the 16-byte slot of (group g, hi / lo) in row R is (2 g + (g & 1) ^ lo) ^ (R & 7) - found by
// exhaustive search: conflict free in all four 16-lane groups of ds_read_b128 for ANY 16 consecutive rows - and the
// filter rows of a stage are stored tile by tile, tile t row r = filter 32 (t >> 1) + 8 (r >> 2) + 4 (t & 1) + (r & 3),
// so that the four registers of filter tiles 2 u and 2 u + 1 of a lane are EIGHT CONSECUTIVE channels of one pixel:
// the 32-byte S16 store of the MF = 0 epilogue, unchanged.
// One output tile (8 x 32 pixels x BN filters) of one workgroup.  `vb` / `nvb` = the tile's block id and the number of
// tiles (what blockIdx.x / gridDim.x are when every tile is its own workgroup); `hiprio` = this workgroup belongs to the
// class that runs at the higher wave priority (see below).
//
// KH = 1 (4-wave, one patch stage, 32x32x16): the k-half-major software pipeline, see "KH" below.
// Sums of eight registers over the 32 lanes of each half of the wave (an accumulator register holds 32 pixels of one
// channel), valid in lanes 31 and 63: four row_shr steps leave each 16-lane row's total in its lane 15, row_bcast15
// adds it into the next row.  v_add_f32 with DPP modifiers - no LDS crossbar traffic, unlike __shfl - written in asm,
// one step of all eight registers after the other: from __builtin_amdgcn_update_dpp hipcc makes a v_mov_b32_dpp AND an
// add per step, and a DPP read needs two wait states behind the VALU write of its source, which eight independent
// chains hide (the s_nop covers the writes in front of the block).
#define TAP_DPP_STEP(op_, mod_)                                                                                        \
  op_ " %0, %0, %0 " mod_ "\n " op_ " %1, %1, %1 " mod_ "\n " op_ " %2, %2, %2 " mod_ "\n " op_ " %3, %3, %3 " mod_ "\n "  \
  op_ " %4, %4, %4 " mod_ "\n " op_ " %5, %5, %5 " mod_ "\n " op_ " %6, %6, %6 " mod_ "\n " op_ " %7, %7, %7 " mod_ "\n"
#define TAP_DPP_REDUCE(op_)                                                                                            \
  float v0 = a[0], v1 = a[1], v2 = a[2], v3 = a[3], v4 = b[0], v5 = b[1], v6 = b[2], v7 = b[3];                        \
  asm volatile("s_nop 1\n"                                                                                             \
               TAP_DPP_STEP(op_, "row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1")                                  \
               TAP_DPP_STEP(op_, "row_shr:2 row_mask:0xf bank_mask:0xf bound_ctrl:1")                                  \
               TAP_DPP_STEP(op_, "row_shr:4 row_mask:0xf bank_mask:0xf bound_ctrl:1")                                  \
               TAP_DPP_STEP(op_, "row_shr:8 row_mask:0xf bank_mask:0xf bound_ctrl:1")                                  \
               TAP_DPP_STEP(op_, "row_bcast:15 row_mask:0xa bank_mask:0xf")                                            \
               "s_nop 1"                                                                                               \
               : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "+v"(v4), "+v"(v5), "+v"(v6), "+v"(v7));                      \
  a = f32x4{v0, v1, v2, v3};                                                                                           \
  b = f32x4{v4, v5, v6, v7};
__device__ __forceinline__ void tap_half_sums32(f32x4& a, f32x4& b) { TAP_DPP_REDUCE("v_add_f32_dpp") }
// the same with max, for values >= 0 (the lanes a row_shr shifts in read as 0)
__device__ __forceinline__ void tap_half_maxs32(f32x4& a, f32x4& b) { TAP_DPP_REDUCE("v_max_f32_dpp") }
#undef TAP_DPP_REDUCE
#undef TAP_DPP_STEP

// BNB: the instance with the BatchNorm-backward statistics epilogue (AmmcConvDesc::bn_c) - its own kernel, so that the
// registers that epilogue needs (the saved tensor's values, four per-channel constants, four accumulators per channel
// quad) do not touch the allocation of the instances every other launch runs
template <int WGM, int WGN, int TM, int TN, int AS, int MF, int KH = 0, bool BNB = false>
__device__ __forceinline__ void conv_tap_s16_tile(const TapArgs& a, const int vb, const int nvb, const bool hiprio) {
  static_assert((WGM * WGN == 8 || WGM * WGN == 4) && WGM * TM == T_TH, "4 or 8 waves, 8 image rows");
  static_assert(!KH || (AS == 1 && MF == 0 && WGM == 4 && WGN == 1 && TM == 2 && (TN == 2 || TN == 4)), "KH: 4 waves of 64 x (64 | 128)");
  constexpr int NT = 64 * WGM * WGN;
  constexpr int T_AROUNDS = (T_APIECES + NT - 1) / NT;
  constexpr int T_ASTAGE = T_AROUNDS * NT * 4;             // floats
  static_assert(AS == 1 || T_AROUNDS <= 6, "two patch stages: one round per tap 0..5");
  constexpr int BM = T_TH * T_TW;           // 256 output pixels
  constexpr int BN = WGN * TN * 32;
  constexpr int BJ = BN * 8 >= NT ? BN * 8 / NT : 1;       // filter pieces per thread per tap (every wave issues:
  constexpr int B_STAGE = BJ * (NT / 8) * 32;                    // a 32-filter slice is padded to 64 rows; floats)
  // filter-slice stages: 3 = slices t+1 and t+2 fly during step t; the 4-wave 128-filter variant keeps 2 (one slice
  // ahead) so that two workgroups fit a CU
  constexpr int NB = KH ? 2 : (NT == 256 && (BN == 128 || BN == 32)) ? 2 : 3;     // (4-wave output layer: 52 KB, three workgroups per CU)
  constexpr int PD = NB - 1;
  // KH: the patch is two half-patches (channels 0-15 | 16-31 of the block), each 340 pixels x 64 B + 768 B that the last
  // DMA round of wave 1 spills into, and a 1-KB dump that the (all out of range) last round of waves 2, 3 is pointed at
  constexpr int K_HPIECES = T_HP * 4, K_HROUNDS = 6, K_HSTRIDE = (T_HP * 64 + 768) / 4, K_ADUMP = 2 * K_HSTRIDE;
  static_assert(K_HROUNDS * 256 >= K_HPIECES && (K_HROUNDS - 1) * 256 + 128 <= K_HPIECES + 48 && 5 * 256 + 64 <= K_HPIECES, "half-patch rounds");
  constexpr int A_FLOATS = KH ? K_ADUMP + 256 : AS * T_ASTAGE;
  constexpr int STAGES = A_FLOATS + NB * B_STAGE;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* As = smem;
  float* Bs = smem + A_FLOATS;
  // BatchNorm scale / shift of this tile's BN filters: [BN scale | BN shift] behind the stages (1 KB reserved), fetched
  // by one LDS-DMA per wave at the start of the tile (every wave writes the same bytes).  The epilogue reads them with
  // ds_read: its only vector-memory operations are then its stores.  Read with global loads they tied the epilogue to
  // its own stores - vector-memory operations retire in order, so the wait for the constants of channel group g + 1 was
  // a wait for the acknowledgement of the stores of group g - 1: eight store round trips per tile, 9.4 of the 92 us a
  // 128 -> 128 tile at 128 x 128 lasts (tools/micro/tap_stamps.py).
  float* SCs = smem + STAGES;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave / WGN;
  const int wn = wave % WGN;
  const int h = lane >> 5;
  const int l31 = lane & 31;
  const int l15 = lane & 15;           // MF = 1: pixel / filter row of a 16x16 tile
  const int g4 = lane >> 4;            //         S16 group of the 32-channel block (MFMA k = 8 g4 .. 8 g4 + 7)
  const AmmcConvDesc& d = a.d;

  const int logical = ammc_xcd_remap(vb, nvb);
  // Two workgroups share a CU (AS == 1).  Started together they stay in phase: both stream their patch, both run
  // their MFMAs and both store their tile at the same moments, so the chip alternates between an idle matrix pipe and
  // an idle memory system (the stores of one round of 512 tiles alone are 5-9 us at the full HBM rate).  Every other
  // group of 256 workgroups (the dispatcher fills the 256 CUs once, then a second time) runs at a higher wave priority:
  // it gets ahead of its CU-mate, finishes first, and from then on one workgroup's epilogue and prologue hide behind
  // the other's MFMAs.  Measured with the layer launched back to back (same box): 128x128 128->128 169 -> 155 us,
  // 256x256 64->64 224 -> 212 us; inside the model, where kernels of different shapes follow each other, the dispatch
  // order is less regular and the step time moves within its noise (AMMC_S16_DBG=-1 turns it off for A/Bs).
  if (AS == 1 && a.dbg != -1 && hiprio) __builtin_amdgcn_s_setprio(1);
  const int n0 = (logical % a.n_tiles) * BN;
  int sp = logical / a.n_tiles;
  const int tx = sp % a.tiles_x;
  sp /= a.tiles_x;
  const int ty = sp % a.tiles_y;
  const int b = sp / a.tiles_y;
  const int y0 = ty * T_TH, x0 = tx * T_TW;

  // d.x is the halo corner of pixel (0,0); the patch of this workgroup starts at image pixel (y0 - 1, x0 - 1)
  const float* xpatch = d.x + ((int64_t)b * d.x_bs + (int64_t)y0 * d.x_rs + (int64_t)x0 * d.x_ps);

  // halo-patch DMA pieces of this thread: piece p = j*NT + tid -> halo pixel p >> 3, physical slot p & 7
  int a_off[T_AROUNDS];
#pragma unroll
  for (int j = 0; j < T_AROUNDS; ++j) {
    int p = j * NT + tid;
    p = p < T_APIECES ? p : T_APIECES - 1;
    const int hp = p >> 3;
    int ls = (p & 7) ^ ((hp >> 1) & 7);
    if (MF) {                                       // physical slot q of row R holds source piece pi(q ^ (R & 7)),
      ls = (p & 7) ^ (hp & 7);                      // pi = swap 2 <-> 3 and 6 <-> 7 (an involution)
      ls ^= (ls >> 1) & 1;
    }
    const int hy = hp / T_HW;
    const int hx = hp - hy * T_HW;
    a_off[j] = (int)((int64_t)hy * d.x_rs + (int64_t)hx * d.x_ps) + 4 * ls;
  }
  int sl = (tid & 7) ^ ((tid >> 4) & 7);
  if (MF) {
    sl = (tid & 7) ^ ((tid >> 3) & 7);
    sl ^= (sl >> 1) & 1;
  }
  const float* b_src[BJ];
#pragma unroll
  for (int j = 0; j < BJ; ++j) {
    int row = j * (NT / 8) + (tid >> 3);                     // LDS row of the stage
    if (MF) row = (row & ~31) | (((row >> 2) & 3) << 3) | (((row >> 4) & 1) << 2) | (row & 3);   // tile-ordered rows
    row += n0;
    row = row < d.n ? row : d.n - 1;                         // padding rows of a 32-filter slice: any valid address
    b_src[j] = d.w + (int64_t)row * a.kpad + 4 * sl;
  }

  // (macros, not lambdas / dependent expressions: see the hipcc notes in DESIGN.md section 8)
#define TAP_ISSUE_A(j, cc, stage)                                                     \
  {                                                                                   \
    const float* src_ = xpatch + a_off[j] + (cc) * 32;                                \
    float* dst_ = As + (stage) * T_ASTAGE + ((j) * NT + wave * 64) * 4;               \
    __builtin_amdgcn_global_load_lds(src_, dst_, 16, 0, 0);                           \
  }
#define TAP_ISSUE_B(chunk, stage)                                                     \
  _Pragma("unroll") for (int j_ = 0; j_ < BJ; ++j_) {                                 \
    const float* src_ = b_src[j_] + (chunk) * 32;                                     \
    float* dst_ = Bs + (stage) * B_STAGE + (j_ * NT + wave * 64) * 4;                 \
    __builtin_amdgcn_global_load_lds(src_, dst_, 16, 0, 0);                           \
  }

  // SA (the two-workgroups-per-CU variants): ONE accumulator set.  The 2^-11 of the cross terms is applied to the
  // A-side fragments on the fly (hi * 2^-11 and lo * 2^-11 in half precision: exact above 2^-3, an absolute 2^-25
  // below, i.e. far under the fp32 rounding of the sum), which halves the accumulator registers.
  constexpr bool SA = AS == 1;
  constexpr int XM = SA ? 1 : TM, XN = SA ? 1 : TN;
  constexpr int PT = 2 * TM, FT = (WGN * TN * 32 == 32) ? 1 : 2 * TN;     // MF = 1: 16-pixel / 16-filter tiles of a wave
  constexpr int FC = FT > 4 ? 4 : (AS == 2 && FT > 2 ? 2 : FT);                                      //         filter tiles whose fragments are live at once
  f32x4 acc[MF ? PT : 1][MF ? FT : 1];                                     //         (the output layer uses ONE filter tile)
  f32x4 acx[MF && !SA ? PT : 1][MF && !SA ? FT : 1];                       //         cross terms (two-accumulator variants)
#pragma unroll
  for (int i = 0; i < (MF ? PT : 1); ++i)
#pragma unroll
    for (int j = 0; j < (MF ? FT : 1); ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int i = 0; i < (MF && !SA ? PT : 1); ++i)
#pragma unroll
    for (int j = 0; j < (MF && !SA ? FT : 1); ++j) acx[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
#define TAP_ACC16(pt, j, k) (SA ? acc[pt][j][k] : acc[pt][j][k] + acx[SA ? 0 : (pt)][SA ? 0 : (j)][k] * T_LO_INV)
  f32x16 hh[MF ? 1 : TM][MF ? 1 : TN], xx[MF ? 1 : XM][MF ? 1 : XN];
#pragma unroll
  for (int i = 0; i < (MF ? 1 : TM); ++i)
#pragma unroll
    for (int j = 0; j < (MF ? 1 : TN); ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) hh[i][j][r] = 0.f;
#pragma unroll
  for (int i = 0; i < (MF ? 1 : XM); ++i)
#pragma unroll
    for (int j = 0; j < (MF ? 1 : XN); ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) xx[i][j][r] = 0.f;
#define TAP_ACC(i, j, r) (SA ? hh[i][j][r] : hh[i][j][r] + xx[(SA ? 0 : (i))][(SA ? 0 : (j))][r] * T_LO_INV)

  // filter row of this lane: MFMA row l31 takes filter pi(l31), pi = swap bits 2 and 3, so that a lane's accumulator
  // registers are runs of eight consecutive output channels (see the epilogue)
  const int pl31 = (l31 & 19) | ((l31 & 4) << 1) | ((l31 & 8) >> 1);
  const int swzb = (pl31 >> 1) & 7;
  const int b_row = (wn * TN * 32 + pl31) * 32;
  int hpb[TM];                                  // halo pixel of this lane's output pixel for tap (0,0)
#pragma unroll
  for (int i = 0; i < TM; ++i) hpb[i] = (wm * TM + i) * T_HW + (MF ? l15 : l31);
  // MF = 1: logical slots of this lane's S16 group (hi, lo) and its filter row of a 16-row tile
  const int slot_hi = 2 * g4 + (g4 & 1), slot_lo = slot_hi ^ 1;
  const int b_row16 = (wn * TN * 32 + l15) * 32;
  const int swzb16 = l15 & 7;

#ifndef AMMC_TAP_SETPRIO
#define AMMC_TAP_SETPRIO 0
#endif
#if AMMC_TAP_SETPRIO
#define TAP_PRIO(v) __builtin_amdgcn_s_setprio(v)
#else
#define TAP_PRIO(v)
#endif
#define TAP_COMPUTE16(tap, astage, bstage)                                                                 \
  {                                                                                                        \
    /* pixel fragments of all PT tiles stay resident (SA: hi, hi 2^-11, lo 2^-11; else hi, lo); filter fragments in  \
       chunks of FC tiles */                                                                               \
    const float* Ac = As + (astage) * T_ASTAGE;                                                            \
    const float* Bc = Bs + (bstage) * B_STAGE + b_row16;                                                   \
    f16x8t ah[PT], ax_[PT], al2_[PT];      /* ax_: SA ? hi 2^-11 : lo */                                   \
    _Pragma("unroll") for (int pt = 0; pt < PT; ++pt) {                                                    \
      int hp_ = hpb[pt >> 1] + 16 * (pt & 1) + ((tap) / 3) * T_HW + ((tap) % 3);                           \
      asm volatile("" : "+v"(hp_));                                                                        \
      const float* ap_ = Ac + hp_ * 32;                                                                    \
      const int sw_ = hp_ & 7;                                                                             \
      ah[pt] = *reinterpret_cast<const f16x8t*>(ap_ + ((slot_hi ^ sw_) << 2));                             \
      const f16x8t al_ = *reinterpret_cast<const f16x8t*>(ap_ + ((slot_lo ^ sw_) << 2));                   \
      if (SA) {                                                                                            \
        ax_[pt] = ah[pt] * (_Float16)T_LO_INV;                                                             \
        al2_[pt] = al_ * (_Float16)T_LO_INV;                                                               \
      } else {                                                                                             \
        ax_[pt] = al_;                                                                                     \
      }                                                                                                    \
    }                                                                                                      \
    _Pragma("unroll") for (int f0 = 0; f0 < FT; f0 += FC) {                                                \
      f16x8t bh[FC], bl[FC];                                                                               \
      _Pragma("unroll") for (int j = 0; j < FC; ++j) {                                                     \
        bh[j] = *reinterpret_cast<const f16x8t*>(Bc + (f0 + j) * 512 + ((slot_hi ^ swzb16) << 2));         \
        bl[j] = *reinterpret_cast<const f16x8t*>(Bc + (f0 + j) * 512 + ((slot_lo ^ swzb16) << 2));         \
      }                                                                                                    \
      if (SA) {                                                                                            \
        _Pragma("unroll") for (int pt = 0; pt < PT; ++pt) _Pragma("unroll") for (int j = 0; j < FC; ++j)   \
          acc[pt][f0 + j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bh[j], ah[pt], acc[pt][f0 + j], 0, 0, 0);   \
        _Pragma("unroll") for (int pt = 0; pt < PT; ++pt) _Pragma("unroll") for (int j = 0; j < FC; ++j)   \
          acc[pt][f0 + j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bl[j], ax_[pt], acc[pt][f0 + j], 0, 0, 0);  \
        _Pragma("unroll") for (int pt = 0; pt < PT; ++pt) _Pragma("unroll") for (int j = 0; j < FC; ++j)   \
          acc[pt][f0 + j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bh[j], al2_[pt], acc[pt][f0 + j], 0, 0, 0); \
      } else {                                                                                             \
        _Pragma("unroll") for (int pt = 0; pt < PT; ++pt) _Pragma("unroll") for (int j = 0; j < FC; ++j)   \
          acc[pt][f0 + j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bh[j], ah[pt], acc[pt][f0 + j], 0, 0, 0);   \
        _Pragma("unroll") for (int pt = 0; pt < PT; ++pt) _Pragma("unroll") for (int j = 0; j < FC; ++j)   \
          acx[SA ? 0 : pt][SA ? 0 : f0 + j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bl[j], ah[pt], acx[SA ? 0 : pt][SA ? 0 : f0 + j], 0, 0, 0); \
        _Pragma("unroll") for (int pt = 0; pt < PT; ++pt) _Pragma("unroll") for (int j = 0; j < FC; ++j)   \
          acx[SA ? 0 : pt][SA ? 0 : f0 + j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bh[j], ax_[pt], acx[SA ? 0 : pt][SA ? 0 : f0 + j], 0, 0, 0); \
      }                                                                                                    \
    }                                                                                                      \
  }
#define TAP_COMPUTE(tap, astage, bstage)                                                                   \
  if constexpr (MF != 0) TAP_COMPUTE16(tap, astage, bstage) else                                           \
  {                                                                                                        \
    const float* Ac = As + (astage) * T_ASTAGE;                                                            \
    const float* Bc = Bs + (bstage) * B_STAGE + b_row;                                                     \
    int arow_[TM], aswz_[TM];                                                                              \
    _Pragma("unroll") for (int i = 0; i < TM; ++i) {                                                       \
      int hp_ = hpb[i] + ((tap) / 3) * T_HW + ((tap) % 3);                                                 \
      asm volatile("" : "+v"(hp_)); /* keep the per-tap addresses out of loop-invariant registers (else spills) */ \
      arow_[i] = hp_ * 32;                                                                                 \
      aswz_[i] = (hp_ >> 1) & 7;                                                                           \
    }                                                                                                      \
    _Pragma("unroll") for (int s = 0; s < 2; ++s) {                                                        \
      const int g = 2 * s + h;                                                                             \
      f16x8t ah[TM], al[TM], bh[TN], bl[TN];                                                               \
      _Pragma("unroll") for (int i = 0; i < TM; ++i) {                                                     \
        ah[i] = *reinterpret_cast<const f16x8t*>(Ac + arow_[i] + (((2 * g) ^ aswz_[i]) << 2));             \
        al[i] = *reinterpret_cast<const f16x8t*>(Ac + arow_[i] + (((2 * g + 1) ^ aswz_[i]) << 2));         \
      }                                                                                                    \
      _Pragma("unroll") for (int j = 0; j < TN; ++j) {                                                     \
        bh[j] = *reinterpret_cast<const f16x8t*>(Bc + j * 1024 + (((2 * g) ^ swzb) << 2));                 \
        bl[j] = *reinterpret_cast<const f16x8t*>(Bc + j * 1024 + (((2 * g + 1) ^ swzb) << 2));             \
      }                                                                                                    \
      if (SA) {                                                                                            \
        /* one accumulator set: three passes over the tiles, so that MFMAs on the same accumulator are TM*TN apart */ \
        f16x8t ah2_[TM], al2_[TM];                                                                         \
        _Pragma("unroll") for (int i = 0; i < TM; ++i) {                                                   \
          ah2_[i] = ah[i] * (_Float16)T_LO_INV;                                                            \
          al2_[i] = al[i] * (_Float16)T_LO_INV;                                                            \
        }                                                                                                  \
        TAP_PRIO(1);                                                                                       \
        _Pragma("unroll") for (int i = 0; i < TM; ++i) _Pragma("unroll") for (int j = 0; j < TN; ++j)      \
          hh[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bh[j], ah[i], hh[i][j], 0, 0, 0);             \
        _Pragma("unroll") for (int i = 0; i < TM; ++i) _Pragma("unroll") for (int j = 0; j < TN; ++j)      \
          hh[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bl[j], ah2_[i], hh[i][j], 0, 0, 0);           \
        _Pragma("unroll") for (int i = 0; i < TM; ++i) _Pragma("unroll") for (int j = 0; j < TN; ++j)      \
          hh[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bh[j], al2_[i], hh[i][j], 0, 0, 0);           \
        TAP_PRIO(0);                                                                                       \
      } else {                                                                                             \
        TAP_PRIO(1);                                                                                       \
        _Pragma("unroll") for (int i = 0; i < TM; ++i) _Pragma("unroll") for (int j = 0; j < TN; ++j) {    \
          hh[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bh[j], ah[i], hh[i][j], 0, 0, 0);             \
          xx[SA ? 0 : i][SA ? 0 : j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bl[j], ah[i], xx[SA ? 0 : i][SA ? 0 : j], 0, 0, 0); \
          xx[SA ? 0 : i][SA ? 0 : j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bh[j], al[i], xx[SA ? 0 : i][SA ? 0 : j], 0, 0, 0); \
        }                                                                                                  \
        TAP_PRIO(0);                                                                                       \
      }                                                                                                    \
    }                                                                                                      \
  }

  // one step = one tap of one 32-channel block.  In flight while step t is contracted: the filter slices of steps
  // t+1 and t+2 (three stages) and, during taps 0..5, one round of the next block's halo patch.  Every wave issues
  // the same instructions in the same order, so `vmcnt(n)` with n = what was issued after slice t+1 retires exactly
  // slice t+1 and everything older.  Literal counts only (a dependent asm operand loses the kernel's host stub).
#define TAP_WAIT(n)                                                              \
  {                                                                              \
    if ((n) == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");               \
    else if ((n) == 1) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");          \
    else if ((n) == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");          \
    else asm volatile("s_waitcnt vmcnt(3)" ::: "memory");                        \
  }
// the ablation switches of AMMC_S16_DBG (2 = no MFMA, 3 = no DMA, 4 = no DMA and no barriers) cost ~12 scalar branches
// per tap, which also cut the loop into small scheduling regions: compiled in only with -DAMMC_TAP_DEBUG
#ifdef AMMC_TAP_DEBUG
#define TAP_DBG a.dbg
#else
#define TAP_DBG 0
#endif
#define TAP_STEP(tap)                                                                                      \
  {                                                                                                        \
    if (AS == 2 && (tap) < T_AROUNDS && !lastcc && TAP_DBG < 3) { TAP_ISSUE_A((tap) < T_AROUNDS ? (tap) : 0, cc + 1, (cc + 1) & 1); } \
    const int bsp_ = bs + PD >= NB ? bs + PD - NB : bs + PD;                                               \
    if (TAP_DBG >= 3) {                                                                                    \
    } else if ((tap) + PD < 9) {                                                                           \
      TAP_ISSUE_B(((tap) + PD) * a.ncc + cc, bsp_);                                                        \
    } else if (!lastcc) {                                                                                  \
      TAP_ISSUE_B(((tap) + PD - 9) * a.ncc + cc + 1, bsp_);                                                \
    }                                                                                                      \
    if (TAP_DBG != 2) { TAP_COMPUTE(tap, (AS == 2 ? (cc & 1) : 0), bs); }                                    \
    if (TAP_DBG >= 3) {                                                                                    \
      TAP_WAIT(0);                                                                                         \
    } else if (!lastcc) {                                                                                  \
      TAP_WAIT((PD - 1) * BJ + (AS == 2 && (tap) < T_AROUNDS ? 1 : 0));                                    \
    } else {                                                                                               \
      TAP_WAIT(((PD - 1) < (7 - (tap)) ? (PD - 1) : ((7 - (tap)) > 0 ? (7 - (tap)) : 0)) * BJ);            \
    }                                                                                                      \
    if (TAP_DBG != 4) __builtin_amdgcn_s_barrier();                                                        \
    asm volatile("" ::: "memory");                                                                         \
    bs = bs + 1 == NB ? 0 : bs + 1;                                                                        \
  }

  // ---- KH: k-half-major software pipeline ---------------------------------------------------------------------------
  // What the tap-by-tap loop above leaves on the table (tools/micro/tap_stamps.py, 128 -> 128 at 128 x 128): a workgroup
  // ALONE on its CU reaches 46 % of the matrix pipe - every k-step begins with twelve fragment reads whose latency nothing
  // covers, every tap ends in a DMA wait and a barrier, every block in a patch reload (1.7 us) - and two workgroups
  // interleave to 66-71 %.  Here a block of 32 channels is swept as 18 STEPS of 16 channels, half-major: steps 0-8 =
  // the nine taps over channels 0-15, steps 9-17 = the taps over channels 16-31, so that
  //  * half 0 of the patch is dead after step 8 and is refilled with the NEXT block's channels while half 1 is swept
  //    (and half 1 after step 17): no exposed patch reload, no second stage (the two half-patches are separate LDS
  //    images of 64-byte rows: slot q of pixel p holds logical slot q ^ ((p >> 2) & 3), conflict free for ds_read_b128
  //    from any start pixel since 16 lanes of a group cover 16 distinct pixels mod 16);
  //  * a filter stage holds a PAIR of steps (same 128-byte rows and fragment addresses as a tap's two k-steps before;
  //    only the DMA source differs), two stages, one barrier per pair, placed before the pair's LAST chunk: behind it
  //    the next pair's slice has landed for everybody and this pair's stage is free, so the slice after next is
  //    requested there and has a whole pair to arrive;
  //  * fragments are read one CHUNK (12 MFMAs: two image rows x two filter tiles x three products) ahead into a second
  //    register set, pinned by sched_barrier (left alone hipcc sinks the reads to their use and waits lgkmcnt(0)).
  if constexpr (KH != 0) {
#ifndef AMMC_KH_JC4
#define AMMC_KH_JC4 1
#define AMMC_KH_JC2 1
#define AMMC_KH_PF4 1
#define AMMC_KH_PF2 2
#endif
    constexpr int JC = TN == 4 ? AMMC_KH_JC4 : AMMC_KH_JC2;   // filter tiles per chunk: 6 or 12 MFMAs
    constexpr int CS = TN / JC;                      // chunks per step
    constexpr int PF = TN == 4 ? AMMC_KH_PF4 : AMMC_KH_PF2;   // fragments are read PF chunks ahead (PF + 1 register sets)
    static_assert(PF == 1 || (PF == 2 && CS >= 2), "two chunks ahead needs a pair of at least four chunks");
    constexpr int LB = CS <= PF ? 2 : 1;             // buffers of the lo pixel fragments (only read by the scaling at a step's first chunk)
    // byte offsets of this lane's DMA pieces from wave-uniform bases (the patch of the tile; the filter matrix), kept
    // in no register at all between refills: piece p + 256 of a half-patch is halo pixel hp + 64 = one row down and 30 to the right, or two
    // rows down and 4 to the left (its slot swizzle, (hp >> 2) & 3, is the same), and the filter rows of a thread are
    // 32 rows apart - the offsets of rounds 1 .. 5 are rebuilt from round 0 at each refill (a few VALU operations per
    // round, twice per block) instead of living in registers through the contraction
    const unsigned ka_step_a = 4u * (unsigned)(int)(d.x_rs + 30 * d.x_ps), ka_step_b = 4u * (unsigned)(int)(2 * d.x_rs - 4 * d.x_ps);
    // the last piece of a half-patch (round 5 clamps to it): pixel 339 = (9, 33), its slot 3 ^ ((339 >> 2) & 3) = 3
    const unsigned ka_last = 4u * (unsigned)((int)(9 * d.x_rs + 33 * d.x_ps) + 4 * (3 ^ ((339 >> 2) & 3)));
    const int uwave = __builtin_amdgcn_readfirstlane(wave);
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) void*)smem;
    const int ksl = (tid & 7) ^ ((tid >> 4) & 7);    // logical slot of this thread's filter piece: (step of the pair, group of the half, hi / lo)
    const bool ku1 = (ksl >> 2) != 0;
    const unsigned kb_off0 = 4u * (unsigned)(((tid >> 3) + n0) * a.kpad + 4 * (ksl & 3));     // (KH serves n = 64 or n % 128 == 0: every row exists)
    const unsigned kb_rstep = 4u * 32u * (unsigned)a.kpad;
    // all six rounds of one half-patch
#define K_ISSUE_HALF(ccx, hf)                                                                              \
  {                                                                                                        \
    int t_ = tid;                                                                                          \
    asm volatile("" : "+v"(t_));     /* rebuilt here from the thread id: held in registers it was spilled */     \
    const int hp0_ = t_ >> 2;                                                                              \
    const int hy0_ = hp0_ / T_HW;                                                                          \
    int hx_ = hp0_ - hy0_ * T_HW;                                                                          \
    unsigned off_ = 4u * (unsigned)(hy0_ * (int)d.x_rs + hx_ * (int)d.x_ps + 4 * ((t_ & 3) ^ ((hp0_ >> 2) & 3))); \
    _Pragma("unroll") for (int j_ = 0; j_ < K_HROUNDS; ++j_) {                                             \
      const unsigned dst_ = lds0 + 4u * (unsigned)((j_ == K_HROUNDS - 1 && uwave >= 2) ? K_ADUMP : (hf) * K_HSTRIDE + (j_ * NT + uwave * 64) * 4); \
      const unsigned src_ = (j_ == K_HROUNDS - 1 && j_ * NT + tid >= K_HPIECES) ? ka_last : off_;          \
      tap_dma16(xpatch + (ccx) * 32 + (hf) * 16, src_, dst_);                                              \
      const bool wrap_ = hx_ >= 4;                                                                         \
      off_ += wrap_ ? ka_step_b : ka_step_a;                                                               \
      hx_ += wrap_ ? -4 : 30;                                                                              \
    }                                                                                                      \
  }
#define K_OFFB(sg, ccx) (((((sg) % 9) * a.ncc) + (ccx)) * 32 + ((sg) / 9) * 16)
#define K_ISSUE_B(k, ccx, stage)                                                                           \
  {                                                                                                        \
    const int o0_ = K_OFFB(2 * (k), ccx), o1_ = K_OFFB(2 * (k) + 1, ccx);                                  \
    const unsigned off_ = 4u * (unsigned)(ku1 ? o1_ : o0_);                                                \
    _Pragma("unroll") for (int j_ = 0; j_ < BJ; ++j_) {                                                    \
      const unsigned dst_ = lds0 + 4u * (unsigned)(A_FLOATS + (stage) * B_STAGE + (j_ * NT + uwave * 64) * 4); \
      tap_dma16(d.w, kb_off0 + j_ * kb_rstep + off_, dst_);                                                \
    }                                                                                                      \
  }
    f16x8t ra_h[2][TM], ra_l[LB][TM], sa_h[TM], sa_l[TM], rb_h[PF + 1][JC], rb_l[PF + 1][JC];
#define K_LOAD_A(sg, buf)                                                                                  \
  _Pragma("unroll") for (int i_ = 0; i_ < TM; ++i_) {                                                      \
    int hp_ = hpb[0] + (i_ + ((sg) % 9) / 3) * T_HW + (((sg) % 9) % 3);                                    \
    asm volatile("" : "+v"(hp_));                                                                          \
    const float* ap_ = As + ((sg) / 9) * K_HSTRIDE + hp_ * 16;                                             \
    const int q_ = (2 * h) ^ ((hp_ >> 2) & 3);                                                             \
    ra_h[buf][i_] = *reinterpret_cast<const f16x8t*>(ap_ + (q_ << 2));                                     \
    ra_l[LB == 2 ? (buf) : 0][i_] = *reinterpret_cast<const f16x8t*>(ap_ + ((q_ ^ 1) << 2));               \
  }
#define K_LOAD_B(sg, c, buf)                                                                               \
  {                                                                                                        \
    const float* Bc_ = Bs + ((((sg) >> 1) & 1) ^ bs0) * B_STAGE + b_row;                                   \
    const int ls_ = (((sg) & 1) << 2) | (h << 1);                                                          \
    _Pragma("unroll") for (int jj_ = 0; jj_ < JC; ++jj_) {                                                 \
      rb_h[buf][jj_] = *reinterpret_cast<const f16x8t*>(Bc_ + (JC * (c) + jj_) * 1024 + ((ls_ ^ swzb) << 2));       \
      rb_l[buf][jj_] = *reinterpret_cast<const f16x8t*>(Bc_ + (JC * (c) + jj_) * 1024 + (((ls_ | 1) ^ swzb) << 2)); \
    }                                                                                                      \
  }
    // chunk (sg, c): [reads of the fragments of the chunk PF ahead] | 6 JC MFMAs.  qn = the chunk's index inside the
    // block.  The pixel fragments of step sg + 1 are read in chunk (sg, CS - PF).  LDS reads return in order, so before
    // the MFMAs it is enough that nothing but the reads issued in this chunk and (PF = 2) in the previous one is
    // outstanding; written as a builtin wait, or hipcc inserts lgkmcnt(0).
#define K_CHUNK(sg, c)                                                                                     \
  {                                                                                                        \
    constexpr int qn_ = (sg) * CS + (c);                                                                   \
    constexpr int tq_ = qn_ + PF;                                                                          \
    constexpr int nsg_ = (tq_ / CS) % 18, nc_ = tq_ % CS;                                                  \
    if (tq_ / CS >= 18) {                       /* that chunk belongs to the next block: other stage parity */ \
      const int keep_ = bs0;                                                                               \
      bs0 ^= 1;                                                                                            \
      K_LOAD_B(nsg_, nc_, tq_ % (PF + 1))                                                                  \
      bs0 = keep_;                                                                                         \
    } else {                                                                                               \
      K_LOAD_B(nsg_, nc_, tq_ % (PF + 1))                                                                  \
    }                                                                                                      \
    if ((c) == CS - PF) { K_LOAD_A(((sg) + 1) % 18, ((sg) + 1) & 1) }                                      \
    __builtin_amdgcn_sched_barrier(0);                                                                     \
    __builtin_amdgcn_s_waitcnt(0xC07F | ((2 * JC + ((c) == CS - PF ? 2 * TM : 0) +                         \
                                          (PF == 2 ? 2 * JC + ((c) == CS - PF + 1 ? 2 * TM : 0) : 0)) << 8)); \
    __builtin_amdgcn_sched_barrier(0);                                                                     \
    if ((c) == 0) {                                                                                        \
      _Pragma("unroll") for (int i_ = 0; i_ < TM; ++i_) {                                                  \
        sa_h[i_] = ra_h[(sg) & 1][i_] * (_Float16)T_LO_INV;                                                \
        sa_l[i_] = ra_l[LB == 2 ? ((sg) & 1) : 0][i_] * (_Float16)T_LO_INV;                                \
      }                                                                                                    \
    }                                                                                                      \
    _Pragma("unroll") for (int i_ = 0; i_ < TM; ++i_) _Pragma("unroll") for (int jj_ = 0; jj_ < JC; ++jj_) \
      hh[i_][JC * (c) + jj_] = __builtin_amdgcn_mfma_f32_32x32x16_f16(rb_h[qn_ % (PF + 1)][jj_], ra_h[(sg) & 1][i_], hh[i_][JC * (c) + jj_], 0, 0, 0); \
    _Pragma("unroll") for (int i_ = 0; i_ < TM; ++i_) _Pragma("unroll") for (int jj_ = 0; jj_ < JC; ++jj_) \
      hh[i_][JC * (c) + jj_] = __builtin_amdgcn_mfma_f32_32x32x16_f16(rb_l[qn_ % (PF + 1)][jj_], sa_h[i_], hh[i_][JC * (c) + jj_], 0, 0, 0); \
    _Pragma("unroll") for (int i_ = 0; i_ < TM; ++i_) _Pragma("unroll") for (int jj_ = 0; jj_ < JC; ++jj_) \
      hh[i_][JC * (c) + jj_] = __builtin_amdgcn_mfma_f32_32x32x16_f16(rb_h[qn_ % (PF + 1)][jj_], sa_l[i_], hh[i_][JC * (c) + jj_], 0, 0, 0); \
    __builtin_amdgcn_sched_barrier(0);                                                                     \
  }
    // s_waitcnt through the builtin (gfx9 encoding: vmcnt[3:0] | expcnt << 4 | lgkmcnt << 8 | vmcnt[5:4] << 14), so that
    // the compiler's own wait insertion KNOWS the fragment reads are complete: behind an inline-asm wait it still
    // counts them as pending and puts lgkmcnt(0) - instead of lgkmcnt(reads of the next chunk) - in front of the MFMAs
#define K_WAIT(n)                                                                                          \
  {                                                                                                        \
    if ((n) == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                         \
    else asm volatile("s_waitcnt vmcnt(6)" ::: "memory");                                                  \
    __builtin_amdgcn_s_waitcnt(0xC07F);          /* lgkmcnt(0) */                                          \
  }
    // the point P(k) between the last two chunks of pair k: every fragment of this pair's stage has been read, the
    // slice of pair k + 1 (requested behind P(k - 1)) has landed; half-patch refills requested behind P(4) / P(8) may
    // still be in flight at P(5) / P(0) (six rounds, issued after that slice: counted)
#define K_POINT(k)                                                                                         \
  {                                                                                                        \
    if (((k) == 5 && !lastcc) || ((k) == 0 && cc > 0)) K_WAIT(6) else K_WAIT(0)                            \
    __builtin_amdgcn_s_barrier();                                                                          \
    if ((k) + 2 < 9) {                                                                                     \
      K_ISSUE_B((k) + 2, cc, ((k) & 1) ^ bs0)                                                              \
    } else if (!lastcc) {                                                                                  \
      K_ISSUE_B((k) + 2 - 9, cc + 1, ((k) & 1) ^ bs0)                                                      \
    }                                                                                                      \
    if ((k) == 4 && !lastcc) K_ISSUE_HALF(cc + 1, 0)                                                       \
    if ((k) == 8 && !lastcc) K_ISSUE_HALF(cc + 1, 1)                                                       \
  }
    // chunk x (0 .. 2 CS - 1) of pair k; P(k) sits in front of the pair's chunk 2 CS - PF: the chunks behind it read
    // the first fragments of the next pair
#define K_PCHUNK(k, x)                                                                                     \
  if ((x) < 2 * CS) {                                                                                      \
    if ((x) == 2 * CS - PF) K_POINT(k)                                                                     \
    K_CHUNK(2 * (k) + ((x) >= CS ? 1 : 0), ((x) >= CS ? (x) - CS : (x)))                                   \
  }
#define K_PAIR(k) { K_PCHUNK(k, 0) K_PCHUNK(k, 1) K_PCHUNK(k, 2) K_PCHUNK(k, 3) K_PCHUNK(k, 4) K_PCHUNK(k, 5) K_PCHUNK(k, 6) K_PCHUNK(k, 7) }
    TAP_STAMP(0)
    if (d.scale || d.shift) {
      const int pc = lane < BN / 4 ? lane : (lane < BN / 2 ? lane - BN / 4 : 0);
      const float* sbase = d.scale ? d.scale : d.shift;
      const float* mine = (lane < BN / 4 || lane >= BN / 2) ? sbase : (d.shift ? d.shift : d.scale);
      tap_dma16_ptr(mine + n0 + 4 * pc, lds0 + 4u * (unsigned)STAGES);      // (scale and shift are separate allocations)
    }
    K_ISSUE_HALF(0, 0)
    K_ISSUE_HALF(0, 1)
    {
      const int cc = 0;
      K_ISSUE_B(0, cc, 0)
      K_ISSUE_B(1, cc, 1)
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_waitcnt(0xC07F);
    __builtin_amdgcn_s_barrier();
    TAP_STAMP(1)
    int bs0 = 0;
    K_LOAD_A(0, 0)
    K_LOAD_B(0, 0, 0)
    if (PF == 2) { K_LOAD_B(1 / CS, 1 % CS, 1) }
    for (int cc = 0; cc < a.ncc; ++cc) {
      const bool lastcc = cc + 1 == a.ncc;
      K_PAIR(0) K_PAIR(1) K_PAIR(2) K_PAIR(3) K_PAIR(4) K_PAIR(5) K_PAIR(6) K_PAIR(7) K_PAIR(8)
      bs0 ^= 1;
      TAP_STAMP(2 + 2 * cc)
    }
#undef K_ISSUE_HALF
#undef K_ISSUE_B
#undef K_OFFB
#undef K_LOAD_A
#undef K_LOAD_B
#undef K_CHUNK
#undef K_WAIT
#undef K_POINT
#undef K_PCHUNK
#undef K_PAIR
  } else {
  TAP_STAMP(0)
  if (d.scale || d.shift) {
    const int pc = lane < BN / 4 ? lane : (lane < BN / 2 ? lane - BN / 4 : 0);
    const float* base = (lane < BN / 4 || lane >= BN / 2) ? (d.scale ? d.scale : d.shift) : (d.shift ? d.shift : d.scale);
    __builtin_amdgcn_global_load_lds(base + n0 + 4 * pc, SCs, 16, 0, 0);
  }
#pragma unroll
  for (int j = 0; j < T_AROUNDS; ++j) { TAP_ISSUE_A(j, 0, 0); }
  TAP_ISSUE_B(0, 0);
  if (PD == 2) { TAP_ISSUE_B(a.ncc, 1); }
  TAP_WAIT((PD - 1) * BJ);
  __syncthreads();
  TAP_STAMP(1)
  int bs = 0;
  for (int cc = 0; cc < a.ncc; ++cc) {
    const bool lastcc = cc + 1 == a.ncc;
    if (AS == 1 && cc > 0) {                     // single patch stage: everyone is past tap 8 of the previous block
      _Pragma("unroll") for (int j = 0; j < T_AROUNDS; ++j) { TAP_ISSUE_A(j, cc, 0); }
      TAP_WAIT(0);
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      TAP_STAMP(2 + 2 * cc - 1)
    }
    TAP_STEP(0) TAP_STEP(1) TAP_STEP(2) TAP_STEP(3) TAP_STEP(4) TAP_STEP(5) TAP_STEP(6) TAP_STEP(7) TAP_STEP(8)
    TAP_STAMP(2 + 2 * cc)
  }
  }                                              // (the tap-by-tap loop: !KH)
#undef TAP_WAIT
#undef TAP_STEP
#undef TAP_COMPUTE
#undef TAP_COMPUTE16
#undef TAP_ISSUE_A
#undef TAP_ISSUE_B

  // BatchNorm scale / shift of eight consecutive channels, from the tile's LDS copy (see SCs above); the NEXT group's
  // constants are read before the current group is stored.
#define TAP_LOAD_SCSH(c0_, sc_, sh_)                                                                       \
  {                                                                                                        \
    _Pragma("unroll") for (int k_ = 0; k_ < 8; ++k_) sc_[k_] = 1.f, sh_[k_] = 0.f;                         \
    if (d.scale) {                                                                                         \
      const f32x4 s0_ = *reinterpret_cast<const f32x4*>(SCs + (c0_) - n0), s1_ = *reinterpret_cast<const f32x4*>(SCs + (c0_) - n0 + 4); \
      _Pragma("unroll") for (int k_ = 0; k_ < 4; ++k_) sc_[k_] = s0_[k_], sc_[4 + k_] = s1_[k_];           \
    }                                                                                                      \
    if (d.shift) {                                                                                         \
      const f32x4 s0_ = *reinterpret_cast<const f32x4*>(SCs + BN + (c0_) - n0), s1_ = *reinterpret_cast<const f32x4*>(SCs + BN + (c0_) - n0 + 4); \
      _Pragma("unroll") for (int k_ = 0; k_ < 4; ++k_) sh_[k_] = s0_[k_], sh_[4 + k_] = s1_[k_];           \
    }                                                                                                      \
  }
  // ---- epilogue, straight from the accumulators ---------------------------------------------------------------
  // The MFMAs take the filter fragment as the row operand, so an accumulator tile has PIXELS on lanes (lane l31 =
  // pixel x0 + l31 of image row wm*TM + i) and CHANNELS on registers: register r of lane half h is MFMA row
  // (r & 3) + 8 (r >> 2) + 4 h, and the filter fragment presents filter pi(row) there (pi swaps bits 2 and 3, see
  // b_row), so registers 8 o .. 8 o + 7 of a lane are the EIGHT CONSECUTIVE channels of S16 group 2 o + h: a lane
  // stores whole 32-byte groups (16 B of hi halves, 16 B of lo halves), the two lane halves write neighbouring groups -
  // no LDS round trip, no barrier.  (The first form parked the tile in LDS to get 32-byte stores; with everything else
  // removed that epilogue was 10-35 % of the kernel.)
  if constexpr (MF != 0) {
    // ---- epilogue of the 16x16x32 form: lane = pixel l15 of a 16-pixel tile, registers = 4 filters of tile j; tiles
    // 2 u and 2 u + 1 together give the lane channels cb + 8 g4 .. + 7 (see the row order of the filter stage above)
    const int cbase = n0 + wn * TN * 32 + 8 * g4;
    if (d.y_f32) {
      const int nstore = d.n_store > 0 ? d.n_store : d.n;
      const int64_t ycs = d.y_cs > 0 ? d.y_cs : 1;
      float sq0 = 0.f;
#pragma unroll
      for (int j = 0; j < FT; ++j) {
        const int c0 = cbase + 32 * (j >> 1) + 4 * (j & 1);
        f32x4 sc = {1.f, 1.f, 1.f, 1.f}, sh = {0.f, 0.f, 0.f, 0.f};
        if (d.scale) sc = *reinterpret_cast<const f32x4*>(SCs + c0 - n0);
        if (d.shift) sh = *reinterpret_cast<const f32x4*>(SCs + BN + c0 - n0);
#pragma unroll
        for (int pt = 0; pt < PT; ++pt) {
          const int y = y0 + wm * TM + (pt >> 1), x = x0 + 16 * (pt & 1) + l15;
          const int64_t op = (int64_t)b * d.y_bs + (int64_t)y * d.y_rs + (int64_t)x * d.y_ps;
          f32x4 v;
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            float t = TAP_ACC16(pt, j, k) * sc[k] + sh[k];
            if (d.act == AMMC_ACT_RELU) t = t > 0.f ? t : 0.f;
            else if (d.act == AMMC_ACT_TANH) t = tanhf(t);
            v[k] = t;
          }
          if (d.res) {                                          // fp32 outputs take an fp32 NHWC residual
            const f32x4 rv = *reinterpret_cast<const f32x4*>(d.res + ((int64_t)b * d.r_bs + (int64_t)y * d.r_rs + (int64_t)x * d.r_ps) + c0);
#pragma unroll
            for (int k = 0; k < 4; ++k) v[k] += rv[k];
          }
          if (ycs == 1 && c0 + 4 <= nstore) {
            *reinterpret_cast<f32x4*>(d.y + op + c0) = v;
          } else {
#pragma unroll
            for (int k = 0; k < 4; ++k)
              if (c0 + k < nstore) d.y[op + (int64_t)(c0 + k) * ycs] = v[k];
          }
          if (d.sq_target) {
#pragma unroll
            for (int k = 0; k < 4; ++k)
              if (c0 + k < nstore) {
                const float df = 0.5f * (d.sq_target[op + (int64_t)(c0 + k) * ycs] - v[k]);
                sq0 += df * df;
              }
          }
        }
      }
      if (d.sq_target) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) sq0 += __shfl_xor(sq0, off);
        if (lane == 0) unsafeAtomicAdd(d.sq_acc + b, sq0);
      }
      return;
    }
    float vmax = 0.f;
    float scb[2][8], shb[2][8];
    TAP_LOAD_SCSH(cbase, scb[0], shb[0])
#pragma unroll
    for (int u = 0; u < FT / 2; ++u) {
      const int c0 = cbase + 32 * u;                            // this lane's S16 group
      if (u + 1 < FT / 2) TAP_LOAD_SCSH(cbase + 32 * (u + 1), scb[(u + 1) & 1], shb[(u + 1) & 1])
      const float (&sc)[8] = scb[u & 1];
      const float (&sh)[8] = shb[u & 1];
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        float pooled[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < TM; ++i) {
          const int pt = 2 * i + c;
          const int y = y0 + wm * TM + i, x = x0 + 16 * c + l15;
          float v[8];
#pragma unroll
          for (int k = 0; k < 8; ++k) {
            float t = TAP_ACC16(pt, 2 * u + (k >> 2), k & 3) * sc[k] + sh[k];
            if (d.act == AMMC_ACT_RELU) t = t > 0.f ? t : 0.f;
            else if (d.act == AMMC_ACT_LRELU) t = t > 0.f ? t : 0.1f * t;
            v[k] = t;
          }
          if (d.res) {
            const f16x8t* rp = reinterpret_cast<const f16x8t*>(d.res + ((int64_t)b * d.r_bs + (int64_t)y * d.r_rs + (int64_t)x * d.r_ps) + c0);
            const f16x8t rh = rp[0], rl = rp[1];
#pragma unroll
            for (int k = 0; k < 8; ++k) v[k] += (float)rh[k] + (float)rl[k] * T_LO_INV;
          }
          ammc_u4 hi, lo;
          ammc_s16_split8(v, hi, lo);
          // range check: two values per v_max3_f32 (|.| is an operand modifier); the pooling maximum only when asked for
#pragma unroll
          for (int k = 0; k < 8; k += 2) vmax = __builtin_fmaxf(__builtin_fmaxf(vmax, __builtin_fabsf(v[k])), __builtin_fabsf(v[k + 1]));
          if (TM == 2 && d.pool_y) {
#pragma unroll
            for (int k = 0; k < 8; ++k) pooled[k] = i == 0 ? v[k] : fmaxf(pooled[k], v[k]);
          }
          ammc_u4* yp = reinterpret_cast<ammc_u4*>(d.y + ((int64_t)b * d.y_bs + (int64_t)y * d.y_rs + (int64_t)x * d.y_ps) + c0);
          yp[0] = hi;
          yp[1] = lo;
        }
        if (TM == 2 && d.pool_y) {
#pragma unroll
          for (int k = 0; k < 8; ++k) pooled[k] = fmaxf(pooled[k], __shfl_xor(pooled[k], 1));
          if ((l15 & 1) == 0) {
            ammc_u4 hi, lo;
            ammc_s16_split8(pooled, hi, lo);
            ammc_u4* pp = reinterpret_cast<ammc_u4*>(d.pool_y + ((int64_t)b * d.pool_bs + (int64_t)((y0 >> 1) + wm) * d.pool_rs +
                                                              (int64_t)((x0 >> 1) + 8 * c + (l15 >> 1)) * d.pool_ps) + c0);
            pp[0] = hi;
            pp[1] = lo;
          }
        }
      }
    }
    if (d.overflow_flag && !(vmax <= 65504.f)) atomicOr(d.overflow_flag, 1);   // |v| beyond the half range (or NaN)
    TAP_STAMP(14)
#ifdef AMMC_TAP_STAMP
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    TAP_STAMP(15)
#endif
    return;
  }
  int o_pix[TM], r_pix[TM];
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    const int y = y0 + wm * TM + i, x = x0 + l31;
    o_pix[i] = (int)((int64_t)b * d.y_bs + (int64_t)y * d.y_rs + (int64_t)x * d.y_ps);
    r_pix[i] = (int)((int64_t)b * d.r_bs + (int64_t)y * d.r_rs + (int64_t)x * d.r_ps);
  }

  if (d.y_f32) {
    // fp32 output, NHWC (16-byte stores of four channels) or NCHW through y_cs (lanes = consecutive pixels); tanh and
    // the fused squared error of `psnr_error` for the output layer (a patch lies inside one sample)
    const int nstore = d.n_store > 0 ? d.n_store : d.n;
    const int64_t ycs = d.y_cs > 0 ? d.y_cs : 1;
    float sq0 = 0.f;
    // d.stats (training-mode BatchNorm): per-channel sum / sum of squares of the values this workgroup stores, so that
    // no pass re-reads the tensor for them.  A register is 32 pixels of one channel: the rows of the wave are added
    // in the lane, the lanes by DPP (tap_half_sum32), lanes 31 / 63 park the wave's totals in LDS - the stages are
    // dead by now: every wave has waited for its last DMA and is past its last fragment read once the barrier falls -
    // and after a second barrier the waves' rows are added in a fixed order and leave as one row of stats[patch][2][n].
    // d.bn_c (the dgrad of a training backward): the statistics are those of the BatchNorm BACKWARD of the unit this
    // gradient goes to - sum g, sum g xhat, max |g|, max |xhat| per channel with g = v [pre > 0], xhat and pre from that
    // unit's saved convolution output bn_c at the same pixels (chan_reduce_kernel<3>'s four rows, train_kernels.hip):
    // stats[patch][4][n].  The tensor the pass would read back (537 MB on the 256x256 level) is only written.
    float* const St = smem;                                     // [WGM][2 | 4][BN]
    constexpr int SQ = BNB ? 4 : 2;
    int b_pix[TM];
    if (d.stats) {
#pragma unroll
      for (int i = 0; i < TM; ++i)
        b_pix[i] = (int)((int64_t)b * d.bn_bs + (int64_t)(y0 + wm * TM + i) * d.bn_rs + (int64_t)(x0 + l31) * d.bn_ps);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
    }
    // BNB: the saved tensor's values of 32 channels (one j) are requested a whole j ahead of their use (a load per
    // quad, consumed at once, left eight to sixteen exposed round trips per tile: 455 -> 650 us on the 64 -> 64 layer
    // at 256x256), and the four per-channel constants come through the scalar cache: their addresses are uniform but for
    // the lane half h, so both halves' values are loaded and selected
    constexpr bool BN_AHEAD = TN <= 2;          // (128 accumulators leave room for one j of values, not two)
    f32x4 cvb[BN_AHEAD ? 2 : 1][4][TM];
    const float* const bcp = d.bn_c;
    auto bn_issue = [&](const int jj, const int buf) {
#pragma unroll
      for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int i = 0; i < TM; ++i)
          cvb[buf][q][i] = *reinterpret_cast<const f32x4*>(bcp + b_pix[i] + n0 + (wn * TN + jj) * 32 + 8 * (2 * (q >> 1) + h) + 4 * (q & 1));
    };
    if (BNB && BN_AHEAD) bn_issue(0, 0);
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      if (BNB && BN_AHEAD && j + 1 < TN) bn_issue(j + 1, (j + 1) & 1);
      if (BNB && !BN_AHEAD) bn_issue(j, 0);
#pragma unroll
      for (int q = 0; q < 4; ++q) {                             // register quad q: channels c0 .. c0 + 3
        const int c0 = n0 + (wn * TN + j) * 32 + 8 * (2 * (q >> 1) + h) + 4 * (q & 1);
        f32x4 sc = {1.f, 1.f, 1.f, 1.f}, sh = {0.f, 0.f, 0.f, 0.f};
        if (d.scale) sc = *reinterpret_cast<const f32x4*>(SCs + c0 - n0);
        if (d.shift) sh = *reinterpret_cast<const f32x4*>(SCs + BN + c0 - n0);
        f32x4 st1 = {0.f, 0.f, 0.f, 0.f}, st2 = {0.f, 0.f, 0.f, 0.f}, st3 = {0.f, 0.f, 0.f, 0.f}, st4 = {0.f, 0.f, 0.f, 0.f};
        f32x4 bmu, bis, bga, bbe, cv[TM];
        if (BNB) {
          const int cu = n0 + (wn * TN + j) * 32 + 16 * (q >> 1) + 4 * (q & 1);        // lane half 0's channels; half 1: + 8
#define TAP_BN_CONST(dst_, arr_)                                                                                       \
          {                                                                                                            \
            const f32x4 a0_ = *reinterpret_cast<const f32x4*>((arr_) + cu), a1_ = *reinterpret_cast<const f32x4*>((arr_) + cu + 8); \
            _Pragma("unroll") for (int k_ = 0; k_ < 4; ++k_) dst_[k_] = h ? a1_[k_] : a0_[k_];                        \
          }
          TAP_BN_CONST(bmu, d.bn_mean)
          TAP_BN_CONST(bis, d.bn_invstd)
          TAP_BN_CONST(bga, d.bn_scale)
          TAP_BN_CONST(bbe, d.bn_shift)
#undef TAP_BN_CONST
#pragma unroll
          for (int i = 0; i < TM; ++i) cv[i] = cvb[BN_AHEAD ? (j & 1) : 0][q][i];
        }
#pragma unroll
        for (int i = 0; i < TM; ++i) {
          f32x4 v;
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            float t = TAP_ACC(i, j, 4 * q + k) * sc[k] + sh[k];
            if (d.act == AMMC_ACT_RELU) t = t > 0.f ? t : 0.f;
            else if (d.act == AMMC_ACT_TANH) t = tanhf(t);
            v[k] = t;
            if (BNB) {                                       // chan_reduce_kernel<3>'s expressions, operation for operation
              const float xh = (cv[i][k] - bmu[k]) * bis[k];
              const float pre = cv[i][k] * bga[k] + bbe[k];
              const float gi = (!d.bn_relu || pre > 0.f) ? t : 0.f;
              st1[k] += gi;
              st2[k] += gi * xh;
              st3[k] = fmaxf(st3[k], fabsf(gi));
              st4[k] = fmaxf(st4[k], fabsf(xh));
            } else {
              st1[k] += t;
              st2[k] += t * t;
            }
          }
          if (d.res) {                                          // fp32 outputs take an fp32 NHWC residual
            const f32x4 rv = *reinterpret_cast<const f32x4*>(d.res + r_pix[i] + c0);
#pragma unroll
            for (int k = 0; k < 4; ++k) v[k] += rv[k];
          }
          if (ycs == 1 && c0 + 4 <= nstore) {
            *reinterpret_cast<f32x4*>(d.y + o_pix[i] + c0) = v;
          } else {
#pragma unroll
            for (int k = 0; k < 4; ++k)
              if (c0 + k < nstore) d.y[o_pix[i] + (int64_t)(c0 + k) * ycs] = v[k];
          }
          if (d.sq_target) {
#pragma unroll
            for (int k = 0; k < 4; ++k)
              if (c0 + k < nstore) {
                const float df = 0.5f * (d.sq_target[o_pix[i] + (int64_t)(c0 + k) * ycs] - v[k]);
                sq0 += df * df;
              }
          }
        }
        if (d.stats) {
          tap_half_sums32(st1, st2);
          if (BNB) tap_half_maxs32(st3, st4);
          if (l31 == 31) {
            *reinterpret_cast<f32x4*>(St + (wm * SQ) * BN + (c0 - n0)) = st1;
            *reinterpret_cast<f32x4*>(St + (wm * SQ + 1) * BN + (c0 - n0)) = st2;
            if (BNB) {
              *reinterpret_cast<f32x4*>(St + (wm * SQ + 2) * BN + (c0 - n0)) = st3;
              *reinterpret_cast<f32x4*>(St + (wm * SQ + 3) * BN + (c0 - n0)) = st4;
            }
          }
        }
      }
    }
    if (d.sq_target) {
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) sq0 += __shfl_xor(sq0, off);
      if (lane == 0) unsafeAtomicAdd(d.sq_acc + b, sq0);
    }
    if (d.stats) {
      __syncthreads();
      float* const row = d.stats + (int64_t)(logical / a.n_tiles) * SQ * d.n + n0;
      for (int t = tid; t < SQ * BN; t += NT) {
        const int which = t / BN, cl = t - which * BN;
        float acc = St[which * BN + cl];
#pragma unroll
        for (int w = 1; w < WGM; ++w) {
          const float o = St[(w * SQ + which) * BN + cl];
          acc = which < 2 ? acc + o : fmaxf(acc, o);
        }
        row[(int64_t)which * d.n + cl] = acc;
      }
    }
    return;
  }

  // S16 output [+ S16 residual] [+ the 2x2 max-pool of it as a second output: rows 2 wm, 2 wm + 1 are this wave's two
  // row tiles and the horizontal neighbour is the next lane, so the window never leaves the wave]
  float vmax = 0.f;
  float scb[2][8], shb[2][8];
  TAP_LOAD_SCSH(n0 + wn * TN * 32 + 8 * h, scb[0], shb[0])
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    TAP_STAMP(10 + j)
#pragma unroll
    for (int o = 0; o < 2; ++o) {
      const int c0 = n0 + (wn * TN + j) * 32 + 8 * (2 * o + h);  // this lane's S16 group
      const int g = 2 * j + o;                                   // group index; the next one is (j, 1) or (j + 1, 0)
      if (g + 1 < 2 * TN)
        TAP_LOAD_SCSH(n0 + (wn * TN + (g + 1) / 2) * 32 + 8 * (2 * ((g + 1) & 1) + h), scb[(g + 1) & 1], shb[(g + 1) & 1])
      const float (&sc)[8] = scb[g & 1];
      const float (&sh)[8] = shb[g & 1];
      float pooled[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        float v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          float t = TAP_ACC(i, j, 8 * o + k) * sc[k] + sh[k];
          if (d.act == AMMC_ACT_RELU) t = t > 0.f ? t : 0.f;
          else if (d.act == AMMC_ACT_LRELU) t = t > 0.f ? t : 0.1f * t;
          v[k] = t;
        }
        if (d.res) {
          const f16x8t* rp = reinterpret_cast<const f16x8t*>(d.res + r_pix[i] + c0);
          const f16x8t rh = rp[0], rl = rp[1];
#pragma unroll
          for (int k = 0; k < 8; ++k) v[k] += (float)rh[k] + (float)rl[k] * T_LO_INV;
        }
        ammc_u4 hi, lo;
#ifdef AMMC_TAP_STAMP
        if (a.dbg == 12) {          // ablation: no conversion (raw accumulator bits stored)
#pragma unroll
          for (int k = 0; k < 4; ++k) hi[k] = __float_as_uint(v[k]), lo[k] = __float_as_uint(v[4 + k]);
        } else
#endif
        ammc_s16_split8(v, hi, lo);
        // range check: two values per v_max3_f32 (|.| is an operand modifier); the pooling maximum only when asked for
#pragma unroll
        for (int k = 0; k < 8; k += 2) vmax = __builtin_fmaxf(__builtin_fmaxf(vmax, __builtin_fabsf(v[k])), __builtin_fabsf(v[k + 1]));
        if (TM == 2 && d.pool_y) {
#pragma unroll
          for (int k = 0; k < 8; ++k) pooled[k] = i == 0 ? v[k] : fmaxf(pooled[k], v[k]);
        }
        ammc_u4* yp = reinterpret_cast<ammc_u4*>(d.y + o_pix[i] + c0);
#ifdef AMMC_TAP_STAMP
        if (a.dbg == 11) {          // ablation: no stores (values kept alive)
          asm volatile("" :: "v"(hi), "v"(lo));
        } else
#endif
        {
        yp[0] = hi;
        yp[1] = lo;
        }
      }
      if (TM == 2 && d.pool_y) {
#pragma unroll
        for (int k = 0; k < 8; ++k) pooled[k] = fmaxf(pooled[k], __shfl_xor(pooled[k], 1));
        if ((l31 & 1) == 0) {
          ammc_u4 hi, lo;
          ammc_s16_split8(pooled, hi, lo);
          ammc_u4* pp = reinterpret_cast<ammc_u4*>(d.pool_y + ((int64_t)b * d.pool_bs + (int64_t)((y0 >> 1) + wm) * d.pool_rs +
                                                          (int64_t)((x0 >> 1) + (l31 >> 1)) * d.pool_ps) + c0);
          pp[0] = hi;
          pp[1] = lo;
        }
      }
    }
  }
  if (d.overflow_flag && !(vmax <= 65504.f)) atomicOr(d.overflow_flag, 1);   // |v| beyond the half range (or NaN)
  TAP_STAMP(14)
#ifdef AMMC_TAP_STAMP
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  TAP_STAMP(15)
#endif
}

// (hipcc: the second __launch_bounds__ argument is the minimum number of WAVES PER SIMD, i.e. 512 / it VGPRs)
#define TAP_BOUNDS __launch_bounds__(64 * WGM * WGN, (AS == 1 && WGM * WGN == 8 ? 4 : (WGN * TN == 1 ? 3 : 2)))
template <int WGM, int WGN, int TM, int TN, int AS, int MF, int KH, bool BNB = false>
__global__ TAP_BOUNDS void conv_tap_s16_kernel(TapArgs a) {
  conv_tap_s16_tile<WGM, WGN, TM, TN, AS, MF, KH, BNB>(a, blockIdx.x, gridDim.x, (blockIdx.x >> 8) & 1);
}

// (A persistent form of this kernel - 512 workgroups walking the tiles, blockIdx.x and blockIdx.x + 256 sharing a CU for
// the whole launch, class B started half a tile late so that one workgroup's memory phases run behind the other's MFMAs -
// was built and measured in round 3: 1-4 % SLOWER at every delay.  Removed; DESIGN.md section 5, round 3, has the numbers
// and tools/micro/census.hip the placement measurement.)

template <int WGM, int WGN, int TM, int TN, int AS = 2, int MF = 0, int KH = 0>
static int launch_tap(const TapArgs& a, hipStream_t stream, char* label, int label_len) {
  // the statistics epilogue exists in the 32x32x16 form's fp32 store of plain layers (what a training forward launches)
  if (a.d.stats && (MF != 0 || !a.d.y_f32 || a.d.act != AMMC_ACT_NONE || a.d.res || a.d.sq_target || a.d.n_store || a.d.y_cs > 1))
    return AMMC_EUNSUP;
  if (a.d.bn_c && (KH == 0 || MF != 0)) return AMMC_EUNSUP;      // (the k-half-major kernels: what a training dgrad of >= 512 patches runs)
  if (a.d.bn_c && (!a.d.stats || !a.d.bn_mean || !a.d.bn_invstd || !a.d.bn_scale || !a.d.bn_shift || ((uintptr_t)a.d.bn_c & 15) ||
                   ((a.d.bn_bs | a.d.bn_rs | a.d.bn_ps) & 3) ||
                   (int64_t)a.d.batch * a.d.bn_bs + (int64_t)a.d.height * a.d.bn_rs >= (1LL << 31)))
    return AMMC_EINVAL;
  if (label) {                                     // the name rocprofv3 prints for this instance
    if (KH) snprintf(label, label_len, "conv_tap_s16<%d, %d, %d, %d, %d, %d, %d>%s", WGM, WGN, TM, TN, AS, MF, KH, a.d.bn_c ? "+bnbwd" : (a.d.stats ? "+stats" : ""));
    else snprintf(label, label_len, "conv_tap_s16<%d, %d, %d, %d, %d, %d>%s", WGM, WGN, TM, TN, AS, MF, a.d.bn_c ? "+bnbwd" : (a.d.stats ? "+stats" : ""));
    return AMMC_OK;
  }
  constexpr int BN = WGN * TN * 32;
  constexpr int NT = 64 * WGM * WGN;
  constexpr int T_ASTAGE = (T_APIECES + NT - 1) / NT * NT * 4;
  constexpr int BJ = BN * 8 >= NT ? BN * 8 / NT : 1;
  constexpr int NB = KH ? 2 : (NT == 256 && (BN == 128 || BN == 32)) ? 2 : 3;
  constexpr int A_FLOATS = KH ? 2 * ((T_HP * 64 + 768) / 4) + 256 : AS * T_ASTAGE;       // as in conv_tap_s16_tile
  constexpr int STAGES = A_FLOATS + NB * BJ * (NT / 8) * 32 + 256;      // + 1 KB: scale / shift of the tile's filters
  static const size_t pad = getenv("AMMC_TAP_LDSPAD") ? (size_t)atoi(getenv("AMMC_TAP_LDSPAD")) : 0;   // occupancy experiments
  const size_t lds = (size_t)STAGES * sizeof(float) + pad;
  static_assert((size_t)STAGES * sizeof(float) <= 160 * 1024, "LDS budget");
  auto kern = conv_tap_s16_kernel<WGM, WGN, TM, TN, AS, MF, KH>;
  if constexpr (KH != 0 && MF == 0) {
    if (a.d.bn_c) kern = conv_tap_s16_kernel<WGM, WGN, TM, TN, AS, MF, KH, true>;
  }
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                     (int)lds);
  if (e != hipSuccess) return (int)e;
  TapArgs b = a;
  b.n_tiles = a.d.n / BN;
  const int grid = a.d.batch * a.tiles_y * a.tiles_x * b.n_tiles;
  hipLaunchKernelGGL(kern, dim3(grid), dim3(NT), lds, stream, b);
  return ammc_launch_status();
}

int conv_outc_s16_try(const AmmcConvDesc& d, int kpad, hipStream_t stream, char* label, int label_len);

// rows of AmmcConvDesc::stats: one per output patch
int conv_tap_s16_stat_rows(const AmmcConvDesc& d) { return d.batch * (d.height / T_TH) * (d.width / T_TW); }

// Called by ammc_conv_gemm_s16 (conv_gemm_s16.hip) after its argument checks.  Returns TAP_SKIP when the descriptor is
// not this kernel's case (the caller then runs the implicit-GEMM kernel), else the launch status.
int conv_tap_s16_try(const AmmcConvDesc& d, int kpad, hipStream_t stream, char* label, int label_len) {
  static const int mode = getenv("AMMC_S16_TAP") ? atoi(getenv("AMMC_S16_TAP")) : 1;
  static const int dbg = getenv("AMMC_S16_DBG") ? atoi(getenv("AMMC_S16_DBG")) : 0;
  constexpr int TAP_SKIP = -12345;
  if (!mode) return TAP_SKIP;
  if (d.ntaps != 9 || d.up != 1 || d.x_step > 1) return TAP_SKIP;
  // (round 6: LeakyReLU(0.1) in the epilogues - FlowNet2-SD's stride-1 3x3 layers at 128x128 ... 32x32 take the halo-patch
  // kernels; AMMC_TAP_LRELU=0 sends them back to the implicit GEMM for A/Bs)
  static const int tap_lrelu = getenv("AMMC_TAP_LRELU") ? atoi(getenv("AMMC_TAP_LRELU")) : 1;
  if (d.act == AMMC_ACT_LRELU && (!tap_lrelu || d.y_f32)) return TAP_SKIP;      // (the S16-output epilogues have it; fp32 outputs: the GEMM kernel)
  if (d.cin % 32 || d.width % T_TW || d.height % T_TH) return TAP_SKIP;
  if (d.n != 32 && d.n != 64 && d.n % 128) return TAP_SKIP;
  if (d.n == 32 && !d.y_f32) return TAP_SKIP;
  if (d.pool_y && (d.y_f32 || d.res || ((uintptr_t)d.pool_y & 31) || ((d.pool_bs | d.pool_rs | d.pool_ps) & 7))) return AMMC_EINVAL;
  const int64_t tiles = (int64_t)d.batch * (d.height / T_TH) * (d.width / T_TW) * (d.n <= 64 ? 1 : d.n / 128);
  if (tiles < 192) return TAP_SKIP;                       // cannot fill the chip: the split-K path of the GEMM kernel is better
  const int64_t patch = (int64_t)(T_TH + 1) * d.x_rs + (int64_t)(T_TW + 1) * d.x_ps;
  if (patch >= (1LL << 30)) return TAP_SKIP;
  TapArgs a;
  a.d = d;
  a.tiles_x = d.width / T_TW;
  a.tiles_y = d.height / T_TH;
  a.ncc = d.cin / 32;
  a.kpad = kpad;
  a.dbg = dbg;
  a.n_tiles = 0;
#ifdef AMMC_TAP_STAMP
  a.stamps = g_tap_stamps;
#else
  a.stamps = nullptr;
#endif
  // the MFMA shape per variant (option "s16_mf": -1 = the measured faster one, 0 / 1 = forced for A/Bs).  Measured at
  // batch 16, 256x256 on one MI355X: output layer 147 -> 123 us and 64-filter layers 306 -> 293 us
  // with 16x16x32, 128-filter layers 213 -> 220 us (twice the MFMA instructions leave the fragment reads and the
  // 2^-11 scaling half the issue slots), the 8-wave two-accumulator variant 165 -> 158 us
  const int mfo = d.s16_mf ? d.s16_mf - 1 : ammc_opt_s16_mf();      // the call's own choice, else the process default
  const int64_t ntiles = (int64_t)d.batch * (d.height / T_TH) * (d.width / T_TW) * (d.n <= 64 ? 1 : d.n / 128);
  const bool wide4 = d.n > 64 && (mode == 4 || (mode == 1 && ntiles >= 512));          // the 4-wave 128-filter variant
  const int mf = mfo < 0 ? (wide4 ? 0 : 1) : mfo;
  // the output layer (2-3 filters, fp32 NCHW + tanh): 4 waves, 52 KB of LDS, THREE workgroups per CU (124 us against
  // 128 for the 8-wave form at two per CU; the layer waits for its 45-KB patches, not for the matrix pipe)
  if (d.n == 32 && !d.stats) {
    const int rc = conv_outc_s16_try(d, kpad, stream, label, label_len);      // the streaming form (conv_outc_s16.hip)
    if (rc != TAP_SKIP) return rc;
  }
  if (d.n == 32) return mf ? launch_tap<4, 1, 2, 1, 1, 1>(a, stream, label, label_len)
                           : launch_tap<8, 1, 1, 1, 1, 0>(a, stream, label, label_len);
  // The 4-wave forms (two workgroups per CU) run the k-half-major software pipeline (KH, 32x32x16) by default
  // (AMMC_TAP_KH: 0 = the tap-by-tap loop everywhere, 1 = default, 2 = KH wherever it exists).  Measured per layer on
  // random operands at batch 16 (tools/conv_bench.py --net, one box, us; tap-by-tap 4-wave / KH / 8-wave 16x16x32):
  // 128x128 64->128 113 / 109 / 125, 128->128 202 / 192 / 203, 256->128 366 / 354 / 354; 64x64 (512 tiles = ONE round
  // of two workgroups per CU, nothing for a second workgroup to hide behind) 128->256 100 / 98 / 97.5, 256->256
  // 185 / 181 / 170, 512->256 351 / 340 / 314: from two rounds up the 4-wave KH form, below that the 8-wave form.
  static const int kh = getenv("AMMC_TAP_KH") ? atoi(getenv("AMMC_TAP_KH")) : 1;
  if (d.n == 64 && kh && mfo < 0) return launch_tap<4, 1, 2, 2, 1, 0, 1>(a, stream, label, label_len);
  if (d.n == 64) return mf ? launch_tap<4, 1, 2, 2, 1, 1>(a, stream, label, label_len)     // 4 waves of 64x64 (2 image rows x 64 filters), 2 workgroups per CU
                           : launch_tap<4, 1, 2, 2, 1, 0>(a, stream, label, label_len);
  // 4 waves of 64x128 (one accumulator set), two workgroups per CU: fewer LDS reads per MFMA and the neighbour's
  // MFMAs behind every prologue / epilogue - once there are two workgroups for every CU
  if (mfo < 0 && ((kh == 1 && mode == 1 && tiles >= 1024) || (kh == 2 && tiles >= 512) || (kh && mode == 4)))
    return launch_tap<4, 1, 2, 4, 1, 0, 1>(a, stream, label, label_len);
  if (mode == 4 || (mode == 1 && tiles >= 512 && (!kh || mfo >= 0)))
    return mf ? launch_tap<4, 1, 2, 4, 1, 1>(a, stream, label, label_len) : launch_tap<4, 1, 2, 4, 1, 0>(a, stream, label, label_len);
  if (mfo < 0) return launch_tap<4, 2, 2, 2, 2, 1>(a, stream, label, label_len);
  return mf ? launch_tap<4, 2, 2, 2, 2, 1>(a, stream, label, label_len) : launch_tap<4, 2, 2, 2, 2, 0>(a, stream, label, label_len);
}

}  // namespace ammc_s16
